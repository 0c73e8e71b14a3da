// train_tail_kernels.hpp -- backward passes of the G- / Q-sized tail of a training step (reference
// train_GENIE_model.py:1786-1861 around module.py:224-331): Bipartite_ReadIn.fc2, SpatialAggregation x3 (global edge-mean term
// included), SpatialDirect, SpatialAttention and TemporalAttention (applied twice). Included by genie_hip.hip (one translation
// unit: the kernels use its MFMA tile helpers, plans and partial-sum conventions).
//
// Conventions (those of the forward tail kernels and of k_train_b2 / b1 / b0):
//  * a wave owns 16 nodes, lane (j = lane & 15, q = lane >> 4) holds channels 16 t + 4 q + {0..3} of node j; per-edge phases run
//    in the ROW layout lane = 4 r + cq (node r, 16-B chunk cq);
//  * the training forward of the tail IS the inference tail (k_bip_out_m, k_sa_pre_m / k_sa_layer_m, k_ro_pre_m, k_readout_m);
//    it keeps only the layer inputs (r, bip, sa1, sa2, x_spatial) and every backward kernel recomputes the pre-activations it
//    needs from them with the forward's own weight images (G-sized work: cheaper than storing and re-reading them);
//  * dX = W^T dY chains are MFMAs with transposed A fragments (add_block_group_T); dW[out, in] = sum over nodes dY[out] X[in] is
//    an MFMA whose contraction runs over the 16 nodes of a tile (tr16 + outer16);
//  * every wave owns one slot of a partial buffer ([n_acc][256] blocks, [n_vec][16] vectors, 16 scalars), zero-fills it and adds
//    the contribution of each of its tiles; k_train_reduce sums the slots in a fixed two-level order: bitwise reproducible;
//  * scatter-shaped gradients (messages to their source nodes, attention edges to their grid nodes) are gathers over the
//    REVERSED graphs in a fixed edge order: no atomics anywhere.

// ---- partial-slot helpers --------------------------------------------------------------------------------------------------
struct TpSlot {
    float* base; int n_acc, n_vec;
    __device__ __forceinline__ float* acc(int k) const { return base + (size_t)k * 256; }
    __device__ __forceinline__ float* vec(int k) const { return base + (size_t)n_acc * 256 + (size_t)k * 16; }
    __device__ __forceinline__ float* scal() const { return base + (size_t)n_acc * 256 + (size_t)n_vec * 16; }
    __device__ __forceinline__ int floats() const { return n_acc * 256 + n_vec * 16 + 16; }
};
__device__ __forceinline__ TpSlot tp_open(float* part, int n_acc, int n_vec, int wave_id, int lane) {
    TpSlot s;
    s.n_acc = n_acc; s.n_vec = n_vec;
    s.base = part + (size_t)wave_id * (size_t)(n_acc * 256 + n_vec * 16 + 16);
    const int n4 = s.floats() / 4;
    for (int i = lane; i < n4; i += 64) ((f32x4*)s.base)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    return s;
}
__device__ __forceinline__ void tp_acc(const TpSlot& s, int k, int lane, f32x4 v) {
    f32x4* p = (f32x4*)(s.acc(k) + lane * 4);
    *p = *p + v;
}
// per-channel sums over the 16 nodes of the tile (MFMA layout: a DPP row = the 16 lanes of one q)
__device__ __forceinline__ void tp_vec(const TpSlot& s, int k, int j, int q, f32x4 v) {
    v.x = row_sum16(v.x); v.y = row_sum16(v.y); v.z = row_sum16(v.z); v.w = row_sum16(v.w);
    if (j == 0) {
        f32x4* p = (f32x4*)(s.vec(k) + 4 * q);
        *p = *p + v;
    }
}
// the same for a value held in the ROW layout (lane = 4 r + cq: channels 4 cq + {0..3} of node r): sum over r = lanes 4 apart
__device__ __forceinline__ void tp_vec_row(const TpSlot& s, int k, int lane, f32x4 v) {
#pragma unroll
    for (int d = 4; d < 64; d <<= 1) {
        v.x += __shfl_xor(v.x, d); v.y += __shfl_xor(v.y, d); v.z += __shfl_xor(v.z, d); v.w += __shfl_xor(v.w, d);
    }
    if (lane < 4) {
        f32x4* p = (f32x4*)(s.vec(k) + 4 * lane);
        *p = *p + v;
    }
}
__device__ __forceinline__ void tp_scal(const TpSlot& s, int k, int lane, float v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
    if (lane == 0) s.scal()[k] += v;
}
// row layout -> operand form of a node-contracting MFMA: V[ch 4 cq + i][node r] held by lane 4 r + cq -> t[s] = V[ch j][node 4 s + q]
__device__ __forceinline__ f32x4 tr16_row(f32x4 v, float* sc, int lane) {
    const int r = lane >> 2, cq = lane & 3, j = lane & 15, q = lane >> 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) sc[(4 * cq + i) * 17 + r] = v[i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x4 t;
#pragma unroll
    for (int s = 0; s < 4; ++s) t[s] = sc[j * 17 + 4 * s + q];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return t;
}
// MFMA layout <-> row layout of one 16-channel block through a per-wave [16][20] scratch
__device__ __forceinline__ f32x4 mfma_to_row(f32x4 v, float* sc, int lane) {
    const int j = lane & 15, q = lane >> 4, r = lane >> 2, cq = lane & 3;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    *(f32x4*)(sc + j * 20 + 4 * q) = v;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const f32x4 o = *(const f32x4*)(sc + r * 20 + 4 * cq);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return o;
}
__device__ __forceinline__ f32x4 row_to_mfma(f32x4 v, float* sc, int lane) {
    const int j = lane & 15, q = lane >> 4, r = lane >> 2, cq = lane & 3;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    *(f32x4*)(sc + r * 20 + 4 * cq) = v;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const f32x4 o = *(const f32x4*)(sc + j * 20 + 4 * q);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return o;
}
__device__ __forceinline__ f32x4 mask4(f32x4 v, bool ok) { return ok ? v : f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }

// group index maps of the transposed plans ---------------------------------------------------------------------------------
//  read-out heads (PL_TRO0 / PL_TRO1): proj_1^T (in block k of d proj_1), f_values_2^T / f_context_2^T (head h -> out block b),
//  f_values_1^T / f_context_1^T (out block b, in block k), FRONT^T (SpatialDirect.f_direct^T, or SpatialAttention.proj^T: b = 0)
#define GTR_P1(k) (k)
#define GTR_V2(h, b) (2 + (h) * 2 + (b))
#define GTR_C2(h, b) (12 + (h) * 2 + (b))
#define GTR_V1(b, k) (22 + (b) * 2 + (k))
#define GTR_C1(b, k) (26 + (b) * 2 + (k))
#define GTR_FR(b, k) (30 + (b) * 2 + (k))
#define GTR_GROUPS 34
//  SpatialAttention, grid-node side (PL_TSN): the x_j columns of f_context (m = 0) / f_values (m = 1), head h -> out block b
#define GTN(m, h, b) (((m) * 5 + (h)) * 2 + (b))
#define GTN_GROUPS 20
//  SpatialAggregation layer (PL_TSA1..3): fc2^T x-part / edge-mean part (out block b, in block t), fc1[:, 0:C]^T, fglobal^T
#define GTS_X(b, t) ((b) * 2 + (t))
#define GTS_A(b, t) (4 + (b) * 2 + (t))
#define GTS_PJ(b, t) (8 + (b) * 2 + (t))
#define GTS_FG(b) (12 + (b))
#define GTS_GROUPS 14
//  Bipartite_ReadIn.fc2^T (PL_TBIP): out block b of r
#define GTB(b) (b)
#define GTB_GROUPS 2

// accumulator / vector / scalar indices of the read-out backward (gradient maps: build_tail_grad_maps)
constexpr int RB_P1 = 0, RB_C1 = 2, RB_V1 = 6, RB_C2 = 10, RB_V2 = 20, RB_DQ = 30, RB_FR = 35, RB_EDGE = 39, RB_BQ = 54;
constexpr int RB_NACC0 = 39, RB_NACC1 = 59, RB_NVEC = 20;
constexpr int RBV_P1 = 0, RBV_P2W = 2, RBV_C1 = 4, RBV_V1 = 6, RBV_C2 = 8, RBV_V2 = 13, RBV_FR = 18;
constexpr int TQ_ROWS = 16;      // rows of the d(temporal query) block kept behind the gradient blob: [TQ_ROWS][75]

struct RbArgs {
    RoArgs ro;                   // the forward's arguments (img = the MODE's forward image)
    const float* timg;           // transposed image (PL_TRO0 / PL_TRO1)
    const float* d_out;          // [N][T] upstream gradient of y / x
    const float* d_lat;          // optional extra gradient on the head's latent (y_latent [G][30] / the SpatialAttention output [Q][30]:
                                 // consumers outside this kernel), or null
    float* dxs;                  // MODE 0: [G][32] gradient w.r.t. x_spatial through the y branch
    float* eb;                   // MODE 1: [Q * 10][12] per attention edge: alpha (5 heads), d(pre-activation) / sqrt(L) (5 heads)
    float* dxm;                  // MODE 1: [Q][16] gradient w.r.t. the aggregated attention vector (before proj)
    float* part; int n_acc, n_vec;
};

constexpr int RB_SCS = 68;       // floats per node of the score scratches (5 heads x 12 time slots + pad), as RO_SCS
constexpr int RB_LDS_FLOATS = GR_IMG_FLOATS + (GTR_GROUPS * 256 + 16) + 16 * 80 + 10 * 256 + 10 * 80 +
                              4 * (2 * 16 * RB_SCS + 16 * 17 + 16 * 17 + 16 * 20);

// Backward of a read-out head. MODE 0: y = TemporalAttention(SpatialDirect(x_spatial)); MODE 1: x = TemporalAttention(
// SpatialAttention(x_spatial, x_query, x_grid)) (module.py:251-331). MODE 1 leaves the grid-node side of SpatialAttention (the
// x_j columns of f_context / f_values and d x_spatial) to k_sat_node_bwd, handing it alpha and d(pre-activation) per edge.
template <int MODE>
__global__ __launch_bounds__(256, 1) void k_ro_bwd(RbArgs b) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const RoArgs& a = b.ro;
    const TlImg im = tl_stage_image(sm, a.img, GR_GROUPS, GR_BIAS);
    float* tw_ = sm + GR_IMG_FLOATS;                                   // transposed groups
    for (int i = threadIdx.x; i < (GTR_GROUPS * 256 + 16) / 4; i += blockDim.x) ((f32x4*)tw_)[i] = ((const f32x4*)b.timg)[i];
    const f32x4* tw = (const f32x4*)tw_;
    float* qtab = tw_ + GTR_GROUPS * 256 + 16;                         // [16][80]: query[t][16 h + l]
    float* qf = qtab + 16 * 80;                                        // [5][64][4] forward fragments, then [5][64][4] transposed ones
    float* qfT = qf + 5 * 256;
    float* et = qfT + 5 * 256;                                         // [10][80] (MODE 1) edge columns, as k_readout_m
    float* wscr = et + 10 * 80;
    {
        const float act3 = a.raw[a.o_a3];
        for (int i = threadIdx.x; i < 16 * 80; i += blockDim.x) {
            const int t = i / 80, rem = i - t * 80, h = rem >> 4, l = rem & 15;
            float v = 0.f;
            if (t < a.T && l < 15) {
                const int ch = 15 * h + l;
                const float tq = a.t_query[t] / a.scale_t;
                v = a.raw[a.o_q2b + ch];
                for (int k = 0; k < 30; ++k) v += a.raw[a.o_q2w + ch * 30 + k] * prelu1(a.raw[a.o_q1w + k] * tq + a.raw[a.o_q1b + k], act3);
            }
            qtab[i] = v;
        }
        if (MODE == 1) {
            for (int i = threadIdx.x; i < 10 * 80; i += blockDim.x) {
                const int m = i / 80, rem = i - m * 80, h = rem >> 4, l = rem & 15, ch = 15 * h + l;
                float v = 0.f;
                if (l < 15) {
                    if (m < 3) v = a.raw[a.o_sq_w + ch * 3 + m];
                    else if (m < 6) v = a.raw[a.o_sc_w + ch * 33 + 30 + (m - 3)];
                    else if (m < 9) v = a.raw[a.o_sv_w + ch * 33 + 30 + (m - 6)];
                    else v = a.raw[a.o_sq_b + ch];
                }
                et[i] = v;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 5 * 256; i += blockDim.x) {
        const int h = i >> 8, ln = (i & 255) >> 2, r = i & 3, ii = ln & 15, qq = ln >> 4;
        qf[i] = qtab[ii * 80 + h * 16 + 4 * qq + r];                   // A[i = t][k = l]
        qfT[i] = qtab[(4 * qq + r) * 80 + h * 16 + ii];                // A[i = l][k = t]
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    float* wsc = wscr + wave * (2 * 16 * RB_SCS + 16 * 17 + 16 * 17 + 16 * 20);
    float* sc_s = wsc;                       // scores  [16][RB_SCS]
    float* sc_d = wsc + 16 * RB_SCS;         // d score [16][RB_SCS]
    float* trs = sc_d + 16 * RB_SCS;         // tr16 scratch
    float* tre = trs + 16 * 17;              // operand form of the edge attributes (MODE 1)
    float* lsc = tre + 16 * 17;              // layout-change scratch [16][20]
    for (int i = lane; i < 2 * 16 * RB_SCS + 2 * 16 * 17; i += 64) wsc[i] = 0.f;
    if (MODE == 1 && lane < 16) tre[3 * 17 + lane] = 1.f;            // constant-one channel: column 3 of the f_queries blocks = its bias gradient
    __syncthreads();
    const float fa = im.scal[0], sa1 = im.scal[1], act1 = im.scal[2], act2 = im.scal[3], act4 = im.scal[4], act5 = im.scal[5];
    const float inv_sqrt_l = 1.f / sqrtf(15.f);
    const TpSlot ps = tp_open(b.part, b.n_acc, b.n_vec, blockIdx.x * 4 + wave, lane);
    float s_fa = 0.f, s_a1 = 0.f, s_a2 = 0.f, s_a4 = 0.f, s_a5 = 0.f, s_b2 = 0.f, s_sa1 = 0.f;
    float* ws = sc_s + j * RB_SCS;
    float* wd = sc_d + j * RB_SCS;
    const int ntiles = (a.N + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int n = tile * 16 + j;
        const bool ok = n < a.N;
        const int nc = ok ? n : a.N - 1;
        // ------------------------------------------------------------------ forward recompute up to xin
        f32x4 xb[2] = {tl_zero(), tl_zero()}, pre_f[2], xin[2];
        // MODE 1 state kept for the attention backward (row layout)
        const int jl = lane >> 2, ql = lane & 3;
        const int n_l = tile * 16 + jl;
        const bool okl = n_l < a.N;
        const int ncl = okl ? n_l : a.N - 1;
        int jn[RO_K];
        float e[RO_K][3];
        f32x4 xm = tl_zero();
        if (MODE == 0) {
            const float* row = a.x_spatial + (long long)nc * 30;
            xb[0] = tl_load30(row, 0, q); xb[1] = tl_load30(row, 1, q);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                pre_f[t] = mma_block(mma_block(tl_bias(im, t, q), TLW(im, GR_FRONT(t, 0)), xb[0]), TLW(im, GR_FRONT(t, 1)), xb[1]);
                xin[t] = prelu4(pre_f[t], fa);
            }
        } else {
            const float* cvw = a.cv + 4 * ql;
            const float xq0 = a.x_query[ncl * 3 + 0], xq1 = a.x_query[ncl * 3 + 1], xq2 = a.x_query[ncl * 3 + 2];
#pragma unroll
            for (int k = 0; k < RO_K; ++k) jn[k] = a.knn[(long long)ncl * RO_K + k];
#pragma unroll
            for (int k = 0; k < RO_K; ++k) {
                e[k][0] = (xq0 - a.x_grid[jn[k] * 3 + 0]) / a.scale_rel;
                e[k][1] = (xq1 - a.x_grid[jn[k] * 3 + 1]) / a.scale_rel;
                e[k][2] = (xq2 - a.x_grid[jn[k] * 3 + 2]) / a.scale_rel;
            }
#pragma unroll 1
            for (int h = 0; h < 5; ++h) {                                   // as k_readout_m<1>
                const float* eh = et + h * 16 + 4 * ql;
                const f32x4 bq = *(const f32x4*)(eh + 9 * 80);
                f32x4 wq_[3], wc_[3], wv_[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    wq_[d] = *(const f32x4*)(eh + d * 80); wc_[d] = *(const f32x4*)(eh + (3 + d) * 80); wv_[d] = *(const f32x4*)(eh + (6 + d) * 80);
                }
                float al[RO_K];
#pragma unroll
                for (int k = 0; k < RO_K; ++k) {
                    f32x4 q4 = bq, c4 = *(const f32x4*)(cvw + (long long)jn[k] * CVP + h * 16);
#pragma unroll
                    for (int d = 0; d < 3; ++d) { q4 += wq_[d] * e[k][d]; c4 += wc_[d] * e[k][d]; }
                    const f32x4 pr = q4 * c4;
                    float s = ((pr.x + pr.y) + pr.z) + pr.w;
                    s += __shfl_xor(s, 1);
                    s += __shfl_xor(s, 2);
                    al[k] = prelu1(s * inv_sqrt_l, sa1);
                }
                float mx = al[0];
#pragma unroll
                for (int k = 1; k < RO_K; ++k) mx = fmaxf(mx, al[k]);
                float ssum = 0.f;
#pragma unroll
                for (int k = 0; k < RO_K; ++k) { al[k] = expf(al[k] - mx); ssum += al[k]; }
                const float den = ssum + 1e-16f;
                f32x4 gh = tl_zero();
#pragma unroll
                for (int k = 0; k < RO_K; ++k) {
                    f32x4 v4 = *(const f32x4*)(cvw + (long long)jn[k] * CVP + 80 + h * 16);
#pragma unroll
                    for (int d = 0; d < 3; ++d) v4 += wv_[d] * e[k][d];
                    gh += v4 * (al[k] / den);
                }
                xm += gh;
            }
            xm *= 0.2f;
            xm = row_to_mfma(xm, lsc, lane);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                pre_f[t] = mma_block(tl_bias(im, t, q), TLW(im, GR_FRONT(t, 0)), xm);
                xin[t] = prelu4(pre_f[t], fa);
            }
        }
        // ------------------------------------------------------------------ TemporalAttention forward (kept: c1, v1, ctx, val, scores)
        f32x4 c1[2], v1[2], h1[2], h2[2], ctx[5], val[5];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            c1[t] = mma_block(mma_block(tl_bias(im, 2 + t, q), TLW(im, GR_C1(t, 0)), xin[0]), TLW(im, GR_C1(t, 1)), xin[1]);
            h1[t] = prelu4(c1[t], act1);
            v1[t] = mma_block(mma_block(tl_bias(im, 4 + t, q), TLW(im, GR_V1(t, 0)), xin[0]), TLW(im, GR_V1(t, 1)), xin[1]);
            h2[t] = prelu4(v1[t], act2);
        }
#pragma unroll
        for (int h = 0; h < 5; ++h) {
            ctx[h] = mma_block(mma_block(tl_bias(im, 6 + h, q), TLW(im, GR_C2(h, 0)), h1[0]), TLW(im, GR_C2(h, 1)), h1[1]);
            const f32x4 sc = mma_block(tl_zero(), ((const f32x4*)qf)[h * 64 + lane], ctx[h]) * inv_sqrt_l;
            if (q < 3) *(f32x4*)(ws + h * 12 + 4 * q) = sc;
            val[h] = mma_block(mma_block(tl_bias(im, 11 + h, q), TLW(im, GR_V2(h, 0)), h2[0]), TLW(im, GR_V2(h, 1)), h2[1]);
        }
        GSYNC();
        // ------------------------------------------------------------------ backward, time step by time step
        const f32x4 w2a = tl_bias(im, 18, q), w2b = tl_bias(im, 19, q);
        f32x4 dval[5], dw2a = tl_zero(), dw2b = tl_zero(), dbp1a = tl_zero(), dbp1b = tl_zero(), accp1a = tl_zero(), accp1b = tl_zero();
#pragma unroll
        for (int h = 0; h < 5; ++h) dval[h] = tl_zero();
#pragma unroll 1
        for (int t = 0; t < a.T; ++t) {
            const float dout = ok ? b.d_out[(long long)n * a.T + t] : 0.f;
            f32x4 zp = tl_zero();
#pragma unroll
            for (int h = 0; h < 5; ++h) zp += val[h] * ws[h * 12 + t];
            zp *= 0.2f;
            const f32x4 z4 = prelu4(zp, act4);
            const f32x4 ppa = mma_block(tl_bias(im, 16, q), TLW(im, GR_P1(0)), z4), ppb = mma_block(tl_bias(im, 17, q), TLW(im, GR_P1(1)), z4);
            const f32x4 pa = prelu4(ppa, act5), pb = prelu4(ppb, act5);
            if (q == 0) s_b2 += dout;
            dw2a += pa * dout; dw2b += pb * dout;
            const f32x4 dpa = w2a * dout, dpb = w2b * dout;
            s_a5 += negsum4(dpa, ppa) + negsum4(dpb, ppb);
            const f32x4 dp1a = dpa * dprelu4(ppa, act5), dp1b = dpb * dprelu4(ppb, act5);
            dbp1a += dp1a; dbp1b += dp1b;
            f32x4 dz4 = mma_block(tl_zero(), tw[GTR_P1(0) * 64 + lane], dp1a);
            dz4 = mma_block(dz4, tw[GTR_P1(1) * 64 + lane], dp1b);
            {
                const f32x4 zt = tr16(z4, trs, j, q);
                accp1a = outer16(accp1a, tr16(dp1a, trs, j, q), zt);
                accp1b = outer16(accp1b, tr16(dp1b, trs, j, q), zt);
            }
            s_a4 += negsum4(dz4, zp);
            const f32x4 dzp = dz4 * dprelu4(zp, act4) * 0.2f;
#pragma unroll
            for (int h = 0; h < 5; ++h) {
                float ds = dot4(dzp, val[h]);
                ds += __shfl_xor(ds, 16);
                ds += __shfl_xor(ds, 32);
                if (q == 0) wd[h * 12 + t] = ds;
                dval[h] += dzp * ws[h * 12 + t];
            }
        }
        tp_acc(ps, RB_P1 + 0, lane, accp1a); tp_acc(ps, RB_P1 + 1, lane, accp1b);
        tp_vec(ps, RBV_P1 + 0, j, q, dbp1a); tp_vec(ps, RBV_P1 + 1, j, q, dbp1b);
        tp_vec(ps, RBV_P2W + 0, j, q, dw2a); tp_vec(ps, RBV_P2W + 1, j, q, dw2b);
        GSYNC();
        // ------------------------------------------------------------------ heads: d ctx, d val -> d h1, d h2
        f32x4 dh1[2] = {tl_zero(), tl_zero()}, dh2[2] = {tl_zero(), tl_zero()};
        const f32x4 h1t[2] = {tr16(h1[0], trs, j, q), tr16(h1[1], trs, j, q)};
        const f32x4 h2t[2] = {tr16(h2[0], trs, j, q), tr16(h2[1], trs, j, q)};
#pragma unroll
        for (int h = 0; h < 5; ++h) {
            f32x4 dsc = tl_zero();
            if (q < 3) dsc = *(const f32x4*)(wd + h * 12 + 4 * q);
            dsc *= inv_sqrt_l;
            const f32x4 dctx = mma_block(tl_zero(), ((const f32x4*)qfT)[h * 64 + lane], dsc);
            const f32x4 dsct = tr16(dsc, trs, j, q);
            tp_acc(ps, RB_DQ + h, lane, outer16(tl_zero(), dsct, tr16(ctx[h], trs, j, q)));
            const f32x4 dct = tr16(dctx, trs, j, q), dvt = tr16(dval[h], trs, j, q);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                tp_acc(ps, RB_C2 + h * 2 + k, lane, outer16(tl_zero(), dct, h1t[k]));
                tp_acc(ps, RB_V2 + h * 2 + k, lane, outer16(tl_zero(), dvt, h2t[k]));
                dh1[k] = mma_block(dh1[k], tw[GTR_C2(h, k) * 64 + lane], dctx);
                dh2[k] = mma_block(dh2[k], tw[GTR_V2(h, k) * 64 + lane], dval[h]);
            }
            tp_vec(ps, RBV_C2 + h, j, q, dctx);
            tp_vec(ps, RBV_V2 + h, j, q, dval[h]);
        }
        f32x4 dc1[2], dv1[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            s_a1 += negsum4(dh1[t], c1[t]);
            s_a2 += negsum4(dh2[t], v1[t]);
            dc1[t] = dh1[t] * dprelu4(c1[t], act1);
            dv1[t] = dh2[t] * dprelu4(v1[t], act2);
            tp_vec(ps, RBV_C1 + t, j, q, dc1[t]);
            tp_vec(ps, RBV_V1 + t, j, q, dv1[t]);
        }
        f32x4 dxin[2];
        {
            const f32x4 xt[2] = {tr16(xin[0], trs, j, q), tr16(xin[1], trs, j, q)};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x4 dct = tr16(dc1[t], trs, j, q), dvt = tr16(dv1[t], trs, j, q);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    tp_acc(ps, RB_C1 + t * 2 + k, lane, outer16(tl_zero(), dct, xt[k]));
                    tp_acc(ps, RB_V1 + t * 2 + k, lane, outer16(tl_zero(), dvt, xt[k]));
                }
            }
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                f32x4 d = tl_zero();
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    d = mma_block(d, tw[GTR_C1(bb, k) * 64 + lane], dc1[k]);
                    d = mma_block(d, tw[GTR_V1(bb, k) * 64 + lane], dv1[k]);
                }
                dxin[bb] = d;
            }
        }
        if (b.d_lat != nullptr && ok) {
            dxin[0] += tl_load30(b.d_lat + (long long)n * 30, 0, q);
            dxin[1] += tl_load30(b.d_lat + (long long)n * 30, 1, q);
        }
        // ------------------------------------------------------------------ front: SpatialDirect / SpatialAttention.proj
        f32x4 dpf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            dxin[t] = mask4(dxin[t], ok);
            s_fa += negsum4(dxin[t], pre_f[t]);
            dpf[t] = dxin[t] * dprelu4(pre_f[t], fa);
            tp_vec(ps, RBV_FR + t, j, q, dpf[t]);
        }
        if (MODE == 0) {
            const f32x4 xt[2] = {tr16(xb[0], trs, j, q), tr16(xb[1], trs, j, q)};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x4 dt_ = tr16(dpf[t], trs, j, q);
                tp_acc(ps, RB_FR + t * 2 + 0, lane, outer16(tl_zero(), dt_, xt[0]));
                tp_acc(ps, RB_FR + t * 2 + 1, lane, outer16(tl_zero(), dt_, xt[1]));
            }
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                f32x4 d = mma_block(tl_zero(), tw[GTR_FR(bb, 0) * 64 + lane], dpf[0]);
                d = mma_block(d, tw[GTR_FR(bb, 1) * 64 + lane], dpf[1]);
                if (ok) *(f32x4*)(b.dxs + (long long)n * 32 + 16 * bb + 4 * q) = d;
            }
        } else {
            const f32x4 xmt = tr16(xm, trs, j, q);
            tp_acc(ps, RB_FR + 0, lane, outer16(tl_zero(), tr16(dpf[0], trs, j, q), xmt));
            tp_acc(ps, RB_FR + 2, lane, outer16(tl_zero(), tr16(dpf[1], trs, j, q), xmt));
            f32x4 dxm = mma_block(tl_zero(), tw[GTR_FR(0, 0) * 64 + lane], dpf[0]);
            dxm = mma_block(dxm, tw[GTR_FR(0, 1) * 64 + lane], dpf[1]);
            if (ok) *(f32x4*)(b.dxm + (long long)n * 16 + 4 * q) = dxm;
            const f32x4 dagg = mfma_to_row(dxm, lsc, lane) * 0.2f;              // d(agg_h) for every head: mean over heads  :285
            // ---- attention backward, head by head (row layout; forward values recomputed)
            const float* cvw = a.cv + 4 * ql;
#pragma unroll 1
            for (int h = 0; h < 5; ++h) {
                const float* eh = et + h * 16 + 4 * ql;
                const f32x4 bq = *(const f32x4*)(eh + 9 * 80);
                f32x4 wq_[3], wc_[3], wv_[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    wq_[d] = *(const f32x4*)(eh + d * 80); wc_[d] = *(const f32x4*)(eh + (3 + d) * 80); wv_[d] = *(const f32x4*)(eh + (6 + d) * 80);
                }
                float al[RO_K], u[RO_K], dal[RO_K];
#pragma unroll
                for (int k = 0; k < RO_K; ++k) {
                    f32x4 q4 = bq, c4 = *(const f32x4*)(cvw + (long long)jn[k] * CVP + h * 16);
                    f32x4 v4 = *(const f32x4*)(cvw + (long long)jn[k] * CVP + 80 + h * 16);
#pragma unroll
                    for (int d = 0; d < 3; ++d) { q4 += wq_[d] * e[k][d]; c4 += wc_[d] * e[k][d]; v4 += wv_[d] * e[k][d]; }
                    const f32x4 pr = q4 * c4;
                    float s = ((pr.x + pr.y) + pr.z) + pr.w;
                    s += __shfl_xor(s, 1);
                    s += __shfl_xor(s, 2);
                    u[k] = s * inv_sqrt_l;
                    al[k] = prelu1(u[k], sa1);
                    float da = dot4(dagg, v4);
                    da += __shfl_xor(da, 1);
                    da += __shfl_xor(da, 2);
                    dal[k] = da;
                }
                float mx = al[0];
#pragma unroll
                for (int k = 1; k < RO_K; ++k) mx = fmaxf(mx, al[k]);
                float ssum = 0.f;
#pragma unroll
                for (int k = 0; k < RO_K; ++k) { al[k] = expf(al[k] - mx); ssum += al[k]; }
                const float den = ssum + 1e-16f;
                float sdot = 0.f;
#pragma unroll
                for (int k = 0; k < RO_K; ++k) { al[k] = al[k] / den; sdot += al[k] * dal[k]; }
                f32x4 aq = tl_zero(), ac = tl_zero(), av = tl_zero();
#pragma unroll 1
                for (int k = 0; k < RO_K; ++k) {
                    const float dsk = okl ? al[k] * (dal[k] - sdot) : 0.f;            // softmax backward
                    if (ql == 0) s_sa1 += dsk * fminf(u[k], 0.f);
                    const float dui = dsk * (u[k] > 0.f ? 1.f : sa1) * inv_sqrt_l;
                    if (ql == 0 && okl) {
                        float* ep = b.eb + ((long long)n_l * RO_K + k) * 12;
                        ep[h] = al[k];
                        ep[5 + h] = dui;
                    }
                    f32x4 q4 = bq, c4 = *(const f32x4*)(cvw + (long long)jn[k] * CVP + h * 16);
#pragma unroll
                    for (int d = 0; d < 3; ++d) { q4 += wq_[d] * e[k][d]; c4 += wc_[d] * e[k][d]; }
                    const f32x4 dq4 = c4 * dui, dc4 = q4 * dui, dv4 = dagg * (okl ? al[k] : 0.f);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (ql == 0) { tre[0 * 17 + jl] = e[k][0]; tre[1 * 17 + jl] = e[k][1]; tre[2 * 17 + jl] = e[k][2]; }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    f32x4 et_;
#pragma unroll
                    for (int s = 0; s < 4; ++s) et_[s] = tre[j * 17 + 4 * s + q];
                    aq = outer16(aq, tr16_row(dq4, trs, lane), et_);
                    ac = outer16(ac, tr16_row(dc4, trs, lane), et_);
                    av = outer16(av, tr16_row(dv4, trs, lane), et_);
                }
                tp_acc(ps, RB_EDGE + 0 * 5 + h, lane, aq);
                tp_acc(ps, RB_EDGE + 1 * 5 + h, lane, ac);
                tp_acc(ps, RB_EDGE + 2 * 5 + h, lane, av);
                tp_acc(ps, RB_BQ + h, lane, aq);
            }
        }
        GSYNC();
    }
    tp_scal(ps, 0, lane, s_fa); tp_scal(ps, 1, lane, s_a1); tp_scal(ps, 2, lane, s_a2); tp_scal(ps, 3, lane, s_a4);
    tp_scal(ps, 4, lane, s_a5); tp_scal(ps, 5, lane, s_b2);
    if (MODE == 1) tp_scal(ps, 6, lane, s_sa1);
}

// Backward of temporal_query_2(PReLU3(temporal_query_1(t / scale_t))) (module.py:329) from d query [TQ_ROWS][75] (the RB_DQ blocks,
// reduced behind the gradient blob): one workgroup, fixed summation order.
__global__ __launch_bounds__(256) void k_tq_bwd(const float* __restrict__ raw, const float* __restrict__ dq, const float* __restrict__ t_query, int T,
                                                float scale_t, int o_q1w, int o_q1b, int o_q2w, int o_q2b, int o_a3, float* __restrict__ blob) {
    __shared__ float z1[RO_TMAX][32], dz1[RO_TMAX][32], red[RO_TMAX][32];
    const float act3 = raw[o_a3];
    for (int i = threadIdx.x; i < T * 32; i += blockDim.x) {
        const int t = i >> 5, k = i & 31;
        z1[t][k] = k < 30 ? raw[o_q1w + k] * (t_query[t] / scale_t) + raw[o_q1b + k] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 75 * 30; i += blockDim.x) {          // d temporal_query_2.weight[ch][k]
        const int ch = i / 30, k = i - ch * 30;
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += dq[t * 75 + ch] * prelu1(z1[t][k], act3);
        blob[o_q2w + i] += s;
    }
    for (int ch = threadIdx.x; ch < 75; ch += blockDim.x) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += dq[t * 75 + ch];
        blob[o_q2b + ch] += s;
    }
    for (int i = threadIdx.x; i < T * 32; i += blockDim.x) {
        const int t = i >> 5, k = i & 31;
        float dh = 0.f;
        if (k < 30)
            for (int ch = 0; ch < 75; ++ch) dh += raw[o_q2w + ch * 30 + k] * dq[t * 75 + ch];
        red[t][k] = dh * fminf(z1[t][k], 0.f);
        dz1[t][k] = dh * (z1[t][k] > 0.f ? 1.f : act3);
    }
    __syncthreads();
    if (threadIdx.x < 30) {
        const int k = threadIdx.x;
        float sw = 0.f, sb = 0.f;
        for (int t = 0; t < T; ++t) { sw += dz1[t][k] * (t_query[t] / scale_t); sb += dz1[t][k]; }
        blob[o_q1w + k] += sw;
        blob[o_q1b + k] += sb;
    }
    if (threadIdx.x == 32) {
        float s = 0.f;
        for (int t = 0; t < T; ++t)
            for (int k = 0; k < 30; ++k) s += red[t][k];
        blob[o_a3] += s;
    }
}

// Grid-node side of SpatialAttention's backward: node j sums, over the attention edges that END in it (reverse kNN lists, edge
// ids i * 10 + k ascending), the gradients of its context / value rows cv[j] = f_*.weight[:, 0:30] x_j + bias:
//   d ctxrow_h += d(pre) / sqrt(L) . q_h(e),   d valrow_h += 0.2 alpha_h . d xm_i,
// then d x_spatial[j] and the gradients of the x_j columns of f_context / f_values and of their biases.
struct SnArgs {
    int G, nq;
    const float* x_spatial; const float* x_grid; const float* x_query;
    const int32_t* r_rowptr; const int32_t* r_edge;      // reverse kNN: rowptr [G + 1], edge ids
    const float* eb; const float* dxm;
    const float* raw; int o_sq_w, o_sq_b;
    float scale_rel;
    const float* timg;          // PL_TSN
    float* dxs;                 // [G][32]
    float* part; int n_acc, n_vec;
};
__global__ __launch_bounds__(256, 1) void k_sat_node_bwd(SnArgs a) {
    __shared__ __attribute__((aligned(16))) float sm[GTN_GROUPS * 256 + 16 + 4 * 80 + 4 * 16 * 17];
    for (int i = threadIdx.x; i < (GTN_GROUPS * 256 + 16) / 4; i += blockDim.x) ((f32x4*)sm)[i] = ((const f32x4*)a.timg)[i];
    const f32x4* tw = (const f32x4*)sm;
    float* et = sm + GTN_GROUPS * 256 + 16;             // [4][80]: f_queries columns 0..2, bias
    for (int i = threadIdx.x; i < 4 * 80; i += blockDim.x) {
        const int m = i / 80, rem = i - m * 80, h = rem >> 4, l = rem & 15, ch = 15 * h + l;
        et[i] = l < 15 ? (m < 3 ? a.raw[a.o_sq_w + ch * 3 + m] : a.raw[a.o_sq_b + ch]) : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    float* trs = et + 4 * 80 + wave * 16 * 17;
    const TpSlot ps = tp_open(a.part, a.n_acc, a.n_vec, blockIdx.x * 4 + wave, lane);
    const int ntiles = (a.G + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int g = tile * 16 + j;
        const bool ok = g < a.G;
        const int gc = ok ? g : a.G - 1;
        const float* row = a.x_spatial + (long long)gc * 30;
        const f32x4 xb[2] = {tl_load30(row, 0, q), tl_load30(row, 1, q)};
        const float g0 = a.x_grid[gc * 3 + 0], g1 = a.x_grid[gc * 3 + 1], g2 = a.x_grid[gc * 3 + 2];
        f32x4 dC[5], dV[5];
#pragma unroll
        for (int h = 0; h < 5; ++h) { dC[h] = tl_zero(); dV[h] = tl_zero(); }
        const int eb_ = a.r_rowptr[gc], ee = ok ? a.r_rowptr[gc + 1] : eb_;
        for (int e = eb_; e < ee; ++e) {
            const int eid = a.r_edge[e], i = eid / RO_K;
            const float e0 = (a.x_query[i * 3 + 0] - g0) / a.scale_rel, e1 = (a.x_query[i * 3 + 1] - g1) / a.scale_rel,
                        e2 = (a.x_query[i * 3 + 2] - g2) / a.scale_rel;
            const float* ep = a.eb + (long long)eid * 12;
            const f32x4 dx4 = *(const f32x4*)(a.dxm + (long long)i * 16 + 4 * q) * 0.2f;
#pragma unroll
            for (int h = 0; h < 5; ++h) {
                const float* eh = et + h * 16 + 4 * q;
                f32x4 q4 = *(const f32x4*)(eh + 3 * 80);
                q4 += *(const f32x4*)(eh) * e0;
                q4 += *(const f32x4*)(eh + 80) * e1;
                q4 += *(const f32x4*)(eh + 160) * e2;
                dC[h] += q4 * ep[5 + h];
                dV[h] += dx4 * ep[h];
            }
        }
        const f32x4 xt[2] = {tr16(xb[0], trs, j, q), tr16(xb[1], trs, j, q)};
        f32x4 dx[2] = {tl_zero(), tl_zero()};
#pragma unroll
        for (int h = 0; h < 5; ++h) {
            const f32x4 dct = tr16(dC[h], trs, j, q), dvt = tr16(dV[h], trs, j, q);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                tp_acc(ps, GTN(0, h, k), lane, outer16(tl_zero(), dct, xt[k]));
                tp_acc(ps, GTN(1, h, k), lane, outer16(tl_zero(), dvt, xt[k]));
                dx[k] = mma_block(dx[k], tw[GTN(0, h, k) * 64 + lane], dC[h]);
                dx[k] = mma_block(dx[k], tw[GTN(1, h, k) * 64 + lane], dV[h]);
            }
            tp_vec(ps, h, j, q, dC[h]);
            tp_vec(ps, 5 + h, j, q, dV[h]);
        }
        if (ok) {
            *(f32x4*)(a.dxs + (long long)g * 32 + 4 * q) = dx[0];
            *(f32x4*)(a.dxs + (long long)g * 32 + 16 + 4 * q) = dx[1];
        }
    }
}

// ---- SpatialAggregation (module.py:243-249) -------------------------------------------------------------------------------
// forward of one layer: pj[j] = fc1[:, 0:C] x_j; c = sum_j outdeg(j) PReLU3(fglobal x_j) / E; base = b1 + fc1[:, C+3:C+8] c;
// m_ji = PReLU1(pj[j] + base + fc1[:, C:C+3] (pos_i - pos_j) / scale); a_i = mean_j m_ji; out_i = PReLU2(fc2 [x_i || a_i] + b2).
// Pass A (per TARGET node i): pre2, d pre2, the x_i part of d x, d a_i / deg_i (`dan`), fc2 / PReLU2 gradients, and -- second sweep
// over the in-edges -- the sums over ALL edges of d m (= d base) and d m (x) (pos_i - pos_j) / scale (fc1's position columns).
// Pass B (per SOURCE node j): d pj[j] = sum over the out-edges (j -> i) of dan[i] PReLU1'(m_ji) (reversed source graph), the
// global term's chain d base -> d c -> d fglobal, d x_j, fc1[:, 0:C] / fglobal / PReLU3 gradients; workgroup 0 also emits the
// gradient of fc1's global columns (d base (x) c).
struct SbArgs {
    int G, C;
    long long E;
    const float* x_in; const float* pos;
    const int32_t* rowptr; const int32_t* col; const int32_t* outdeg;      // in-edges by target (forward graph)
    const int32_t* r_rowptr; const int32_t* r_col;                          // out-edges by source (reversed graph)
    const float* raw; int fc1_w;
    float scale_rel;
    const float* pj; const float* gpart; int n_gpart;                       // forward pre-pass of this layer (k_sa_pre_m)
    const float* img; const float* timg;
    const float* dout_a; const float* dout_b; const float* dout_x30;        // upstream gradient = sum of the non-null ones
    float* dxd;                  // [G][32] x_i part of d x (pass A) -> pass B adds the rest
    float* dan;                  // [G][32] d a_i / deg_i
    float* dx;                   // [G][32] pass B: gradient w.r.t. this layer's input
    float* part; int n_acc, n_vec;
    const float* part_a; int n_waves_a, n_acc_a, n_vec_a;                   // pass B: pass A's partial slots (d base = its vectors 2, 3)
    float* blob;                 // pass B: gradient blob (workgroup 0 writes the global columns of fc1 directly)
};
constexpr int SBA_NACC = 8, SBA_NVEC = 10, SBB_NACC = 6, SBB_NVEC = 1;

__device__ __forceinline__ void sa_gsum(const float* gpart, int n_gpart, float* gred, float* gsum) {
    const int m = threadIdx.x & 7, chunk = threadIdx.x >> 3;
    float sgl = 0.f;
    for (int bk = chunk; bk < n_gpart; bk += 32) sgl += gpart[bk * 8 + m];
    gred[chunk * 8 + m] = sgl;
    __syncthreads();
    if (threadIdx.x < 8) {
        float t = 0.f;
        for (int k = 0; k < 32; ++k) t += gred[k * 8 + threadIdx.x];
        gsum[threadIdx.x] = t;
    }
    __syncthreads();
}

template <int C>
__global__ __launch_bounds__(256, 1) void k_sa_bwd_a(SbArgs a) {
    __shared__ __attribute__((aligned(16))) float sm[GS_IMG_FLOATS + GTS_GROUPS * 256 + 16 + 8 * 32 + 8 + 32 * 8 + 4 * (16 * 36 + 16 * 17)];
    const TlImg im = tl_stage_image(sm, a.img, GS_GROUPS, GS_BIAS);
    float* tw_ = sm + GS_IMG_FLOATS;
    for (int i = threadIdx.x; i < (GTS_GROUPS * 256 + 16) / 4; i += blockDim.x) ((f32x4*)tw_)[i] = ((const f32x4*)a.timg)[i];
    const f32x4* tw = (const f32x4*)tw_;
    float* w1p = tw_ + GTS_GROUPS * 256 + 16;
    float* gsum = w1p + 8 * 32;
    float* gred = gsum + 8;
    float* wscr = gred + 32 * 8;
    for (int i = threadIdx.x; i < 8 * 32; i += blockDim.x) {
        const int k = i >> 5, cc = i & 31;
        w1p[i] = cc < 30 ? a.raw[a.fc1_w + cc * (C + 8) + C + k] : 0.f;
    }
    sa_gsum(a.gpart, a.n_gpart, gred, gsum);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const int jl = lane >> 2, ql = lane & 3;
    float* ts = wscr + wave * (16 * 36 + 16 * 17);
    float* trs = ts + 16 * 36;
    const float act1 = im.scal[0], act2 = im.scal[1];
    f32x4 base[2], wp[3][2];
    {
        const float invE = 1.f / (float)(a.E > 0 ? a.E : 1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            base[t] = tl_bias(im, 2 + t, ql);
#pragma unroll
            for (int m = 0; m < 5; ++m) base[t] += *(const f32x4*)(w1p + (3 + m) * 32 + 16 * t + 4 * ql) * (gsum[m] * invE);
#pragma unroll
            for (int d = 0; d < 3; ++d) wp[d][t] = *(const f32x4*)(w1p + d * 32 + 16 * t + 4 * ql);
        }
    }
    const TpSlot ps = tp_open(a.part, a.n_acc, a.n_vec, blockIdx.x * 4 + wave, lane);
    float s_a1 = 0.f, s_a2 = 0.f;
    f32x4 dbase[2] = {tl_zero(), tl_zero()}, dwp[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d) dwp[d][0] = dwp[d][1] = tl_zero();
    const int ntiles = (a.G + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int i = tile * 16 + j;
        const bool ok = i < a.G;
        const int ic = ok ? i : a.G - 1;
        const float* row = a.x_in + (long long)ic * C;
        f32x4 xb[2];
        if (C == 15) { xb[0] = tl_load15(row, q); xb[1] = tl_zero(); }
        else { xb[0] = tl_load30(row, 0, q); xb[1] = tl_load30(row, 1, q); }
        const int il = tile * 16 + jl;
        const bool okl = il < a.G;
        const int icl = okl ? il : a.G - 1;
        const float pi0 = a.pos[icl * 3 + 0] / a.scale_rel, pi1 = a.pos[icl * 3 + 1] / a.scale_rel, pi2 = a.pos[icl * 3 + 2] / a.scale_rel;
        const int eb = a.rowptr[icl], ee = okl ? a.rowptr[icl + 1] : eb;
        f32x4 as[2] = {tl_zero(), tl_zero()};
        for (int e = eb; e < ee; ++e) {                                         // edge means, as k_sa_layer_m (edge order)
            const int jn = a.col[e];
            const float d0 = pi0 - a.pos[jn * 3 + 0] / a.scale_rel, d1 = pi1 - a.pos[jn * 3 + 1] / a.scale_rel,
                        d2 = pi2 - a.pos[jn * 3 + 2] / a.scale_rel;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 m = *(const f32x4*)(a.pj + (long long)jn * 32 + 16 * t + 4 * ql) + base[t];
                m += wp[0][t] * d0;
                m += wp[1][t] * d1;
                m += wp[2][t] * d2;
                as[t] += prelu4(m, act1);
            }
        }
        const float deg = (float)max(ee - eb, 1);
        *(f32x4*)(ts + jl * 36 + 4 * ql) = as[0] / deg;
        *(f32x4*)(ts + jl * 36 + 16 + 4 * ql) = as[1] / deg;
        GSYNC();
        const f32x4 av[2] = {*(const f32x4*)(ts + j * 36 + 4 * q), *(const f32x4*)(ts + j * 36 + 16 + 4 * q)};
        GSYNC();
        f32x4 dp2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 v = mma_block(tl_bias(im, t, q), TLW(im, GS_FC2(t, 0)), xb[0]);
            if (C == 30) v = mma_block(v, TLW(im, GS_FC2(t, 1)), xb[1]);
            v = mma_block(v, TLW(im, GS_FC2(t, 2)), av[0]);
            v = mma_block(v, TLW(im, GS_FC2(t, 3)), av[1]);
            f32x4 d = tl_zero();
            if (ok) {
                if (a.dout_a) d += *(const f32x4*)(a.dout_a + (long long)i * 32 + 16 * t + 4 * q);
                if (a.dout_b) d += *(const f32x4*)(a.dout_b + (long long)i * 32 + 16 * t + 4 * q);
                if (a.dout_x30) d += tl_load30(a.dout_x30 + (long long)i * 30, t, q);
            }
            s_a2 += negsum4(d, v);
            dp2[t] = d * dprelu4(v, act2);
            tp_vec(ps, t, j, q, dp2[t]);
        }
        {
            const f32x4 d0t = tr16(dp2[0], trs, j, q), d1t = tr16(dp2[1], trs, j, q);
            const f32x4 x0t = tr16(xb[0], trs, j, q);
            tp_acc(ps, 0, lane, outer16(tl_zero(), d0t, x0t));
            tp_acc(ps, 4, lane, outer16(tl_zero(), d1t, x0t));
            if (C == 30) {
                const f32x4 x1t = tr16(xb[1], trs, j, q);
                tp_acc(ps, 1, lane, outer16(tl_zero(), d0t, x1t));
                tp_acc(ps, 5, lane, outer16(tl_zero(), d1t, x1t));
            }
            const f32x4 a0t = tr16(av[0], trs, j, q), a1t = tr16(av[1], trs, j, q);
            tp_acc(ps, 2, lane, outer16(tl_zero(), d0t, a0t));
            tp_acc(ps, 3, lane, outer16(tl_zero(), d0t, a1t));
            tp_acc(ps, 6, lane, outer16(tl_zero(), d1t, a0t));
            tp_acc(ps, 7, lane, outer16(tl_zero(), d1t, a1t));
        }
        f32x4 dxd[2] = {tl_zero(), tl_zero()}, da[2];
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            if (C == 30 || bb == 0) {
                dxd[bb] = mma_block(mma_block(tl_zero(), tw[GTS_X(bb, 0) * 64 + lane], dp2[0]), tw[GTS_X(bb, 1) * 64 + lane], dp2[1]);
            }
            da[bb] = mma_block(mma_block(tl_zero(), tw[GTS_A(bb, 0) * 64 + lane], dp2[0]), tw[GTS_A(bb, 1) * 64 + lane], dp2[1]);
        }
        *(f32x4*)(ts + j * 36 + 4 * q) = da[0];                                  // MFMA layout -> row layout
        *(f32x4*)(ts + j * 36 + 16 + 4 * q) = da[1];
        GSYNC();
        const f32x4 dan[2] = {*(const f32x4*)(ts + jl * 36 + 4 * ql) / deg, *(const f32x4*)(ts + jl * 36 + 16 + 4 * ql) / deg};
        GSYNC();
        if (ok) {
            *(f32x4*)(a.dxd + (long long)i * 32 + 4 * q) = dxd[0];
            *(f32x4*)(a.dxd + (long long)i * 32 + 16 + 4 * q) = dxd[1];
        }
        if (okl) {
            *(f32x4*)(a.dan + (long long)il * 32 + 4 * ql) = dan[0];
            *(f32x4*)(a.dan + (long long)il * 32 + 16 + 4 * ql) = dan[1];
        }
        for (int e = eb; e < ee; ++e) {                                         // second sweep: d m summed over the node's in-edges
            const int jn = a.col[e];
            const float d0 = pi0 - a.pos[jn * 3 + 0] / a.scale_rel, d1 = pi1 - a.pos[jn * 3 + 1] / a.scale_rel,
                        d2 = pi2 - a.pos[jn * 3 + 2] / a.scale_rel;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 m = *(const f32x4*)(a.pj + (long long)jn * 32 + 16 * t + 4 * ql) + base[t];
                m += wp[0][t] * d0;
                m += wp[1][t] * d1;
                m += wp[2][t] * d2;
                s_a1 += negsum4(dan[t], m);
                const f32x4 dm = dan[t] * dprelu4(m, act1);
                dbase[t] += dm;
                dwp[0][t] += dm * d0;
                dwp[1][t] += dm * d1;
                dwp[2][t] += dm * d2;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        tp_vec_row(ps, 2 + t, lane, dbase[t]);
#pragma unroll
        for (int d = 0; d < 3; ++d) tp_vec_row(ps, 4 + d * 2 + t, lane, dwp[d][t]);
    }
    tp_scal(ps, 0, lane, s_a2);
    tp_scal(ps, 1, lane, s_a1);
}

template <int C>
__global__ __launch_bounds__(256, 1) void k_sa_bwd_b(SbArgs a) {
    __shared__ __attribute__((aligned(16))) float sm[GS_IMG_FLOATS + GTS_GROUPS * 256 + 16 + 8 * 32 + 8 + 32 * 8 + 32 + 8 * 32 + 8 + 4 * (16 * 36 + 16 * 17)];
    const TlImg im = tl_stage_image(sm, a.img, GS_GROUPS, GS_BIAS);
    float* tw_ = sm + GS_IMG_FLOATS;
    for (int i = threadIdx.x; i < (GTS_GROUPS * 256 + 16) / 4; i += blockDim.x) ((f32x4*)tw_)[i] = ((const f32x4*)a.timg)[i];
    const f32x4* tw = (const f32x4*)tw_;
    float* w1p = tw_ + GTS_GROUPS * 256 + 16;
    float* gsum = w1p + 8 * 32;
    float* gred = gsum + 8;
    float* dbs = gred + 32 * 8;             // [32] d base, summed over pass A's waves
    float* dred = dbs + 32;                 // [8][32]
    float* dcs = dred + 8 * 32;             // [8] d c (global term)
    float* wscr = dcs + 8;
    for (int i = threadIdx.x; i < 8 * 32; i += blockDim.x) {
        const int k = i >> 5, cc = i & 31;
        w1p[i] = cc < 30 ? a.raw[a.fc1_w + cc * (C + 8) + C + k] : 0.f;
    }
    sa_gsum(a.gpart, a.n_gpart, gred, gsum);
    {   // d base = sum over pass A's waves of its vectors 2, 3, in a fixed two-level order
        const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
        const size_t stride = (size_t)a.n_acc_a * 256 + (size_t)a.n_vec_a * 16 + 16;
        float s = 0.f;
        for (int wv = grp; wv < a.n_waves_a; wv += 8) s += a.part_a[(size_t)wv * stride + (size_t)a.n_acc_a * 256 + 2 * 16 + o];
        dred[grp * 32 + o] = s;
        __syncthreads();
        if (threadIdx.x < 32) {
            float t = 0.f;
            for (int k = 0; k < 8; ++k) t += dred[k * 32 + threadIdx.x];
            dbs[threadIdx.x] = t;
        }
        __syncthreads();
        if (threadIdx.x < 8) {
            float t = 0.f;
            if (threadIdx.x < 5)
                for (int cc = 0; cc < 30; ++cc) t += w1p[(3 + threadIdx.x) * 32 + cc] * dbs[cc];
            dcs[threadIdx.x] = t;
        }
        __syncthreads();
    }
    const float invE = 1.f / (float)(a.E > 0 ? a.E : 1);
    if (blockIdx.x == 0 && threadIdx.x < 30 * 5) {       // gradient of fc1's global columns: d base (x) c
        const int cc = threadIdx.x / 5, m = threadIdx.x - cc * 5;
        a.blob[a.fc1_w + cc * (C + 8) + C + 3 + m] += dbs[cc] * (gsum[m] * invE);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const int jl = lane >> 2, ql = lane & 3;
    float* ts = wscr + wave * (16 * 36 + 16 * 17);
    float* trs = ts + 16 * 36;
    const float act1 = im.scal[0], act3 = im.scal[3];
    f32x4 base[2], wp[3][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        base[t] = tl_bias(im, 2 + t, ql);
#pragma unroll
        for (int m = 0; m < 5; ++m) base[t] += *(const f32x4*)(w1p + (3 + m) * 32 + 16 * t + 4 * ql) * (gsum[m] * invE);
#pragma unroll
        for (int d = 0; d < 3; ++d) wp[d][t] = *(const f32x4*)(w1p + d * 32 + 16 * t + 4 * ql);
    }
    f32x4 dcs4 = tl_zero();                                // d c in the MFMA layout of the fglobal tile (rows m = 4 q + r < 5)
    if (q == 0) dcs4 = f32x4{dcs[0], dcs[1], dcs[2], dcs[3]};
    if (q == 1) dcs4 = f32x4{dcs[4], 0.f, 0.f, 0.f};
    const TpSlot ps = tp_open(a.part, a.n_acc, a.n_vec, blockIdx.x * 4 + wave, lane);
    float s_a3 = 0.f;
    const int ntiles = (a.G + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int g = tile * 16 + j;
        const bool ok = g < a.G;
        const int gc = ok ? g : a.G - 1;
        const float* row = a.x_in + (long long)gc * C;
        f32x4 xb[2];
        if (C == 15) { xb[0] = tl_load15(row, q); xb[1] = tl_zero(); }
        else { xb[0] = tl_load30(row, 0, q); xb[1] = tl_load30(row, 1, q); }
        // ---- d pj[j]: out-edges of node jl (row layout)
        const int gl = tile * 16 + jl;
        const bool okl = gl < a.G;
        const int gcl = okl ? gl : a.G - 1;
        const float pj0 = a.pos[gcl * 3 + 0] / a.scale_rel, pj1 = a.pos[gcl * 3 + 1] / a.scale_rel, pj2 = a.pos[gcl * 3 + 2] / a.scale_rel;
        const f32x4 pjr[2] = {*(const f32x4*)(a.pj + (long long)gcl * 32 + 4 * ql) + base[0],
                              *(const f32x4*)(a.pj + (long long)gcl * 32 + 16 + 4 * ql) + base[1]};
        f32x4 dpj[2] = {tl_zero(), tl_zero()};
        const int eb = a.r_rowptr[gcl], ee = okl ? a.r_rowptr[gcl + 1] : eb;
        for (int e = eb; e < ee; ++e) {
            const int it = a.r_col[e];
            const float d0 = a.pos[it * 3 + 0] / a.scale_rel - pj0, d1 = a.pos[it * 3 + 1] / a.scale_rel - pj1,
                        d2 = a.pos[it * 3 + 2] / a.scale_rel - pj2;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 m = pjr[t];
                m += wp[0][t] * d0;
                m += wp[1][t] * d1;
                m += wp[2][t] * d2;
                dpj[t] += *(const f32x4*)(a.dan + (long long)it * 32 + 16 * t + 4 * ql) * dprelu4(m, act1);
            }
        }
        *(f32x4*)(ts + jl * 36 + 4 * ql) = dpj[0];
        *(f32x4*)(ts + jl * 36 + 16 + 4 * ql) = dpj[1];
        GSYNC();
        dpj[0] = *(const f32x4*)(ts + j * 36 + 4 * q);
        dpj[1] = *(const f32x4*)(ts + j * 36 + 16 + 4 * q);
        GSYNC();
        // ---- global term
        f32x4 gl4 = mma_block(tl_bias(im, 5, q), TLW(im, GS_FG(0)), xb[0]);
        if (C == 30) gl4 = mma_block(gl4, TLW(im, GS_FG(1)), xb[1]);
        const f32x4 dgl = mask4(dcs4 * ((float)a.outdeg[gc] * invE), ok);
        s_a3 += negsum4(dgl, gl4);
        const f32x4 dp3 = dgl * dprelu4(gl4, act3);
        tp_vec(ps, 0, j, q, dp3);
        // ---- d x_j
        f32x4 dx[2] = {tl_zero(), tl_zero()};
        if (ok) { dx[0] = *(const f32x4*)(a.dxd + (long long)g * 32 + 4 * q); dx[1] = *(const f32x4*)(a.dxd + (long long)g * 32 + 16 + 4 * q); }
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            if (C == 30 || bb == 0) {
                dx[bb] = mma_block(dx[bb], tw[GTS_PJ(bb, 0) * 64 + lane], dpj[0]);
                dx[bb] = mma_block(dx[bb], tw[GTS_PJ(bb, 1) * 64 + lane], dpj[1]);
                dx[bb] = mma_block(dx[bb], tw[GTS_FG(bb) * 64 + lane], dp3);
            }
        }
        if (ok) {
            *(f32x4*)(a.dx + (long long)g * 32 + 4 * q) = dx[0];
            *(f32x4*)(a.dx + (long long)g * 32 + 16 + 4 * q) = dx[1];
        }
        {
            const f32x4 x0t = tr16(xb[0], trs, j, q);
            const f32x4 p0t = tr16(dpj[0], trs, j, q), p1t = tr16(dpj[1], trs, j, q), g3t = tr16(dp3, trs, j, q);
            tp_acc(ps, 0, lane, outer16(tl_zero(), p0t, x0t));
            tp_acc(ps, 2, lane, outer16(tl_zero(), p1t, x0t));
            tp_acc(ps, 4, lane, outer16(tl_zero(), g3t, x0t));
            if (C == 30) {
                const f32x4 x1t = tr16(xb[1], trs, j, q);
                tp_acc(ps, 1, lane, outer16(tl_zero(), p0t, x1t));
                tp_acc(ps, 3, lane, outer16(tl_zero(), p1t, x1t));
                tp_acc(ps, 5, lane, outer16(tl_zero(), g3t, x1t));
            }
        }
    }
    tp_scal(ps, 0, lane, s_a3);
}

// Bipartite_ReadIn.fc2 / PReLU_b2 (module.py:229): r [G][32] -> bip; d bip [G][32] -> d r [G][32]
struct BbArgs {
    int G;
    const float* r; const float* dbip;
    const float* img; const float* timg;
    float* dr;
    float* part; int n_acc, n_vec;
};
__global__ __launch_bounds__(256, 1) void k_bip_bwd(BbArgs a) {
    __shared__ __attribute__((aligned(16))) float sm[GB2_IMG_FLOATS + GTB_GROUPS * 256 + 16 + 4 * 16 * 17];
    const TlImg im = tl_stage_image(sm, a.img, GB_GROUPS2, GB_BIAS2);
    float* tw_ = sm + GB2_IMG_FLOATS;
    for (int i = threadIdx.x; i < (GTB_GROUPS * 256 + 16) / 4; i += blockDim.x) ((f32x4*)tw_)[i] = ((const f32x4*)a.timg)[i];
    const f32x4* tw = (const f32x4*)tw_;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    float* trs = tw_ + GTB_GROUPS * 256 + 16 + wave * 16 * 17;
    const float act = im.scal[0];
    const TpSlot ps = tp_open(a.part, a.n_acc, a.n_vec, blockIdx.x * 4 + wave, lane);
    float s_a = 0.f;
    const int ntiles = (a.G + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int g = tile * 16 + j;
        const bool ok = g < a.G;
        const int gc = ok ? g : a.G - 1;
        const f32x4 r0 = *(const f32x4*)(a.r + (long long)gc * 32 + 4 * q), r1 = *(const f32x4*)(a.r + (long long)gc * 32 + 16 + 4 * q);
        const f32x4 pre = mma_block(mma_block(tl_bias(im, 0, q), TLW(im, 0), r0), TLW(im, 1), r1);
        const f32x4 d = ok ? *(const f32x4*)(a.dbip + (long long)g * 32 + 4 * q) : tl_zero();
        s_a += negsum4(d, pre);
        const f32x4 dp = d * dprelu4(pre, act);
        tp_vec(ps, 0, j, q, dp);
        const f32x4 dpt = tr16(dp, trs, j, q);
        tp_acc(ps, 0, lane, outer16(tl_zero(), dpt, tr16(r0, trs, j, q)));
        tp_acc(ps, 1, lane, outer16(tl_zero(), dpt, tr16(r1, trs, j, q)));
        // (MFMAs stay outside divergent branches: an inactive lane would drop its row of the A operand)
        const f32x4 dr0 = mma_block(tl_zero(), tw[GTB(0) * 64 + lane], dp), dr1 = mma_block(tl_zero(), tw[GTB(1) * 64 + lane], dp);
        if (ok) {
            *(f32x4*)(a.dr + (long long)g * 32 + 4 * q) = dr0;
            *(f32x4*)(a.dr + (long long)g * 32 + 16 + 4 * q) = dr1;
        }
    }
    tp_scal(ps, 0, lane, s_a);
}

// r[g] = sum over the tiles' partial rows in tile order, as [G][32] rows (the training tail's copy of the station sums)
__global__ void k_part_sum32(const float* __restrict__ part, int G, int T, float* __restrict__ r_out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= G * 32) return;
    const int g = idx >> 5, c = idx & 31;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += part[((size_t)g * T + t) * 32 + c];
    r_out[idx] = c < 30 ? s : 0.f;
}
