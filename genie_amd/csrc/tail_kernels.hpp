// Part of genie_hip.hip (one translation unit, included inside its anonymous namespace): the G- / Q-sized tail (Bipartite read-out, SpatialAggregation x3, read-out heads) and the pick-sized heads (k_lslc, k_arrivals).

// The scalar form of the G-sized tail (32 lanes per node, both matvec operands from LDS: 3.5 LDS cycles per wave-FMA,
// tools/lds_matvec.hip) was replaced by the fp32-MFMA tile kernels below in round 2 (103.5 -> 45.4 us per window, DESIGN.md section
// 4e); its last user, the Bipartite read-out of irregular product graphs, went in round 4 (k_seg_sum32 + k_bip_out_m).

// out-degree of every source node (number of edges whose message source is j)
__global__ void k_outdeg(const int32_t* __restrict__ col, long long E, int32_t* __restrict__ deg) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < E) atomicAdd(&deg[col[i]], 1);
}

struct SaArgs {
    int G, C;
    long long E;
    const float* x_in; const float* pos;
    const int32_t* rowptr; const int32_t* col; const int32_t* outdeg;
    const float* raw;
    int fc1_w, fc1_b, fc2_w, fc2_b, fg_w, fg_b, act1, act2, act3;     // this layer
    int nx_fc1_w, nx_fg_w, nx_fg_b, nx_act3;                           // next layer (k_sa_layer<.., NEXT = true>)
    float scale_rel;
    const float* pj_in;    // [G,32] x-part of this layer's messages: fc1.weight[:, 0:C] x_j
    const float* gpart_in; // [n_gpart_in][8] per-block partials of sum_j outdeg(j) PReLU3(fglobal x_j)
    int n_gpart_in;
    float* pj_out;         // [G,32] same for the next layer (NEXT) / for this layer (k_sa_pre)
    float* gpart_out;      // [vg][8]
    int vg;                // virtual blocks of the global-term sum: tile t belongs to virtual block (t / 4) % vg whatever the launch's
                           // grid is, so the partials -- and with them every output -- do not depend on the number of workgroups
    float* out;            // [G,30]
    const float* img;      // k_sa_pre_m / k_sa_layer_m: the layer's k_pack_all image (plan PL_SA1 + layer - 1)
    // batched tail: blockIdx.y = window; the window's copy of each buffer sits this many floats further on
    long long ws_x_in, ws_slot, ws_out;
};
__device__ __forceinline__ void sa_select_window(SaArgs& a) {
    const long long w = blockIdx.y;
    a.x_in += w * a.ws_x_in;
    if (a.pj_in) a.pj_in += w * a.ws_slot;
    if (a.gpart_in) a.gpart_in += w * a.ws_slot;
    if (a.pj_out) a.pj_out += w * a.ws_slot;
    if (a.gpart_out) a.gpart_out += w * a.ws_slot;
    if (a.out) a.out += w * a.ws_out;
}




// read-out heads (module.py:251-331): arguments shared by k_ro_pre_m / k_readout_m and their backward passes

struct RoArgs {
    int N, G, T;                 // nodes handled (G for MODE 0, Q for MODE 1), grid size, number of time queries (<= 16)
    int Nw;                      // batched tail: N = nwin * Nw node ids, window w = n / Nw reads x_spatial / cv of window w
    long long cv_ws;             // floats between the cv buffers of consecutive windows
    const float* x_spatial;      // [G,30]
    const float* x_grid;         // [G,3]   (MODE 1)
    const float* x_query;        // [Q,3]   (MODE 1)
    const int32_t* knn;          // [Q,10]  (MODE 1)
    const float* cv;             // [G,160] per-grid-node parts of f_context / f_values (MODE 1)
    const float* img;            // pre-transposed weight image of this MODE (k_pack_t)
    const float* t_query;        // [T]
    const float* raw;
    float scale_rel, scale_t;
    float* out;                  // [N,T]
    float* cv_out;               // MODE 0, optional: also write the per-grid-node table cv of MODE 1 (the work of k_ro_pre_m: one
    const float* pimg;           // pass over x_spatial and one launch less), with the PL_ROP image `pimg`; cv_ws as for `cv`
    float* lat_out;              // optional [N,30]: the head's latent input of TemporalAttention (MODE 0: SpatialDirect(x_spatial) = y_latent,
                                 // MODE 1: SpatialAttention(x_spatial, x_query)); k_readout_m only
    int o_sd_w, o_sd_b, o_sd_a;
    int o_q1w, o_q1b, o_q2w, o_q2b, o_c1w, o_c1b, o_c2w, o_c2b, o_v1w, o_v1b, o_v2w, o_v2b, o_p1w, o_p1b, o_p2w, o_p2b;
    int o_a1, o_a2, o_a3, o_a4, o_a5;
    int o_sq_w, o_sq_b, o_sc_w, o_sc_b, o_sv_w, o_sv_b, o_sp_w, o_sp_b, o_sa1, o_sa2;
};

// Per-grid-node part of SpatialAttention's edge Linears (module.py:290-291): f_context / f_values act on
// [x_j || edge_attr]; the x_j part  C_j = f_context.weight[:, 0:30] x_j,  V_j = f_values.weight[:, 0:30] x_j  is the same
// for every query that has j as a neighbour, so it is computed once per grid node: cv[j] = [C_j (75, pad 80) | V_j].
constexpr int CVP = 160;

constexpr int RO_K = 10;   // SpatialAttention neighbours (module.py:280 default k, asserted 10 elsewhere in the reference)
constexpr int RO_TMAX = 10;   // time queries per call (the reference uses 9, process_continuous_days.py:359)


// ------------------------------------------------------------------------------------------------
// The G- / Q-sized tail on fp32 MFMA tiles (plans PL_RO0 .. PL_BIP above): Bipartite read-out (module.py:229), SpatialAggregation
// (:243-249), SpatialDirect / SpatialAttention / TemporalAttention (:251-331) with the per-node Linears as MFMA chains over 16
// nodes per wave.
// ------------------------------------------------------------------------------------------------
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));

struct TlImg {                 // LDS copy of a k_pack_all image
    const f32x4* w; const float* bias; const float* scal;
};
__device__ __forceinline__ TlImg tl_stage_image(float* sm, const float* __restrict__ img, int n_groups, int n_bias) {
    const int n4 = (n_groups * 256 + n_bias * 16 + 16) / 4;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) ((f32x4*)sm)[i] = ((const f32x4*)img)[i];
    TlImg im;
    im.w = (const f32x4*)sm; im.bias = sm + n_groups * 256; im.scal = im.bias + n_bias * 16;
    return im;
}
#define TLW(im, g) ((im).w[(g) * 64 + lane])
__device__ __forceinline__ f32x4 tl_bias(const TlImg& im, int tile, int q) { return *(const f32x4*)(im.bias + tile * 16 + 4 * q); }
// channel block t (channels 16t + 4q + {0..3}) of a 30-float row; the row is only 4-byte aligned and ends at channel 29
__device__ __forceinline__ f32x4 tl_load30(const float* __restrict__ row, int t, int q) {
    if (t == 1 && q == 3) { const f32x2u v = *(const f32x2u*)(row + 28); return f32x4{v.x, v.y, 0.f, 0.f}; }
    const f32x4u v = *(const f32x4u*)(row + 16 * t + 4 * q);
    return f32x4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void tl_store30(float* __restrict__ row, int t, int q, f32x4 v) {
    if (t == 1 && q == 3) { *(f32x2u*)(row + 28) = f32x2u{v.x, v.y}; return; }
    *(f32x4u*)(row + 16 * t + 4 * q) = f32x4u{v.x, v.y, v.z, v.w};
}
// the single channel block of a 15-float row
__device__ __forceinline__ f32x4 tl_load15(const float* __restrict__ row, int q) {
    if (q == 3) return f32x4{row[12], row[13], row[14], 0.f};
    const f32x4u v = *(const f32x4u*)(row + 4 * q);
    return f32x4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ f32x4 tl_zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// ------------------------------------------------------------------------------------------------
// WIDE tail (round 4): the same kernels with every Linear of the G-sized tail as a chain of v_mfma_f64_16x16x4_f64 (same lane
// layout as the fp32 form: lane (j, q) holds rows 4 q + r of column j), fp64 accumulators, PReLUs and per-node sums; inputs and
// weights are the fp32 values (exact in fp64), results are rounded to fp32 once per kernel. Why: with fp32 chains the error of
// (y, x) against the reference's fp64 run is made almost entirely HERE (oracle-level attribution on the o1_20x500 fixture,
// profiles/r04_c_error_attribution.txt: grid read-out 1.15e-6 rms, SpatialAggregation 0.5-0.66e-6 per layer, Bipartite 0.34e-6,
// DataAggregation 0.29e-6, of 1.50e-6 in total), and the reference's own fp32 run sits at 7.6e-6 of the 1e-5 bound on that
// fixture. The G-sized tail is ~0.5 GFLOP per window: at the fp64 matrix rate (half the fp32 one) it stays latency-bound.
// The training forward keeps the fp32 form (its backward recomputes pre-activations with fp32 chains).
// ------------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <bool W> struct TlV { typedef f32x4 v; };
template <> struct TlV<true> { typedef f64x4 v; };
__device__ __forceinline__ f64x4 mma_block(f64x4 acc, const f32x4 w, const f64x4 x) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)w.x, x.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)w.y, x.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)w.z, x.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)w.w, x.w, acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ f64x4 prelu4(f64x4 x, float a) {
    const double ad = (double)a;
    return f64x4{x.x >= 0. ? x.x : ad * x.x, x.y >= 0. ? x.y : ad * x.y, x.z >= 0. ? x.z : ad * x.z, x.w >= 0. ? x.w : ad * x.w};
}
// v_mfma_f64_16x16x4_f64 leaves row 4 r + q of D in register r of lane (j, q) (tools/mfma_f64_layout.hip), the fp32 form row
// 4 q + r -- the order the packed weight fragments, the row loads and the stores of these kernels assume. A chain therefore runs
// on accumulators in the instruction's own order and crosses over once at each end: tl_tr4 transposes the 4 x 4 (q, r) block of a
// column with v_permlane32_swap / v_permlane16_swap (gfx950), eight swaps per four doubles; it is its own inverse.
__device__ __forceinline__ void tl_swap32(double& x, double& y) {      // x of lanes 32..63 <-> y of lanes 0..31
    const unsigned long long xb = __builtin_bit_cast(unsigned long long, x), yb = __builtin_bit_cast(unsigned long long, y);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)xb, (unsigned)yb, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(xb >> 32), (unsigned)(yb >> 32), false, false);
    x = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]);
    y = __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
}
__device__ __forceinline__ void tl_swap16(double& x, double& y) {      // x of the odd rows of 16 lanes <-> y of the even rows
    const unsigned long long xb = __builtin_bit_cast(unsigned long long, x), yb = __builtin_bit_cast(unsigned long long, y);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)xb, (unsigned)yb, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(xb >> 32), (unsigned)(yb >> 32), false, false);
    x = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]);
    y = __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
}
__device__ __forceinline__ f64x4 tl_tr4(f64x4 v) {
    double a = v.x, b = v.y, c = v.z, d = v.w;
    tl_swap32(a, c); tl_swap32(b, d);        // (q bit 1) <-> (r bit 1)
    tl_swap16(a, b); tl_swap16(c, d);        // (q bit 0) <-> (r bit 0)
    return f64x4{a, b, c, d};
}
template <bool W> __device__ __forceinline__ typename TlV<W>::v tl_wide(f32x4 v);
template <> __device__ __forceinline__ f32x4 tl_wide<false>(f32x4 v) { return v; }
template <> __device__ __forceinline__ f64x4 tl_wide<true>(f32x4 v) { return f64x4{(double)v.x, (double)v.y, (double)v.z, (double)v.w}; }
// start / end of a chain: the bias (usual order) as an accumulator in the instruction's order, the result back in the usual order
template <bool W> __device__ __forceinline__ typename TlV<W>::v tl_cin(f32x4 v);
template <> __device__ __forceinline__ f32x4 tl_cin<false>(f32x4 v) { return v; }
template <> __device__ __forceinline__ f64x4 tl_cin<true>(f32x4 v) { return tl_tr4(f64x4{(double)v.x, (double)v.y, (double)v.z, (double)v.w}); }
__device__ __forceinline__ f32x4 tl_cout(f32x4 v) { return v; }
__device__ __forceinline__ f64x4 tl_cout(f64x4 v) { return tl_tr4(v); }
__device__ __forceinline__ f32x4 tl_f32(f32x4 v) { return v; }
__device__ __forceinline__ f32x4 tl_f32(f64x4 v) { return f32x4{(float)v.x, (float)v.y, (float)v.z, (float)v.w}; }

// r_g = the per-tile partial rows of a source node summed in tile order (row layout: four consecutive lanes read 64 contiguous
// bytes), then across the wave's LDS scratch into the MFMA layout. WIDE: summed and carried in fp64.
template <bool W> struct TlRowSum { typename TlV<W>::v r0, r1; };
template <bool WIDE>
__device__ __forceinline__ TlRowSum<WIDE> tl_sum_part_rows(const float* __restrict__ pg, int T, float* ts, int jl, int ql, int j, int q) {
    typedef typename TlV<WIDE>::v V;
    V r0 = tl_wide<WIDE>(tl_zero()), r1 = r0;
    int tb = 0;
    for (; tb + 4 <= T; tb += 4) {            // four rows in flight, added in tile order
        f32x4 v0[4], v1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { v0[k] = *(const f32x4*)(pg + (tb + k) * 32); v1[k] = *(const f32x4*)(pg + (tb + k) * 32 + 16); }
#pragma unroll
        for (int k = 0; k < 4; ++k) { r0 += tl_wide<WIDE>(v0[k]); r1 += tl_wide<WIDE>(v1[k]); }
    }
    for (; tb < T; ++tb) { r0 += tl_wide<WIDE>(*(const f32x4*)(pg + tb * 32)); r1 += tl_wide<WIDE>(*(const f32x4*)(pg + tb * 32 + 16)); }
    TlRowSum<WIDE> o;
    if (WIDE) {      // scratch rows of 36 floats = 18 doubles: the two halves of a node's row go through in two rounds
        double* td = (double*)ts;
        *(V*)(td + jl * 18 + 4 * ql) = r0;
        GSYNC();
        o.r0 = *(const V*)(td + j * 18 + 4 * q);
        GSYNC();
        *(V*)(td + jl * 18 + 4 * ql) = r1;
        GSYNC();
        o.r1 = *(const V*)(td + j * 18 + 4 * q);
        GSYNC();
    } else {
        *(V*)(ts + jl * 36 + 4 * ql) = r0;    // row layout -> MFMA layout
        *(V*)(ts + jl * 36 + 16 + 4 * ql) = r1;
        GSYNC();
        o.r0 = *(const V*)(ts + j * 36 + 4 * q);
        o.r1 = *(const V*)(ts + j * 36 + 16 + 4 * q);
        GSYNC();
    }
    return o;
}

// Bipartite read-out (module.py:229): r_g = sum over the tiles' partial rows in tile order, out = PReLU_b2(fc2 r_g).
template <bool WIDE>
__global__ __launch_bounds__(256) void k_bip_out_m(const float* __restrict__ part, int G, int T, const float* __restrict__ img,
                                                  float* __restrict__ out, long long part_ws, long long out_ws) {
    typedef typename TlV<WIDE>::v V;
    __shared__ __attribute__((aligned(16))) float sm[GB2_IMG_FLOATS + 4 * 16 * 36];
    const TlImg im = tl_stage_image(sm, img, GB_GROUPS2, GB_BIAS2);
    __syncthreads();
    part += blockIdx.y * part_ws;
    out += blockIdx.y * out_ws;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const int jl = lane >> 2, ql = lane & 3;          // row layout of the partial-row loads: four consecutive lanes read 64 contiguous bytes
    float* ts = sm + GB2_IMG_FLOATS + wave * 16 * 36;
    const float act = im.scal[0];
    const int ntiles = (G + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int g = tile * 16 + j;
        const bool ok = g < G;
        const int gl = tile * 16 + jl;
        const float* pg = part + (long long)(gl < G ? gl : G - 1) * T * 32 + 4 * ql;
        const TlRowSum<WIDE> rs = tl_sum_part_rows<WIDE>(pg, T, ts, jl, ql, j, q);
        V o = tl_cin<WIDE>(tl_bias(im, 0, q));
        o = mma_block(o, TLW(im, 0), rs.r0);
        o = tl_cout(mma_block(o, TLW(im, 1), rs.r1));
        const f32x4 of = tl_f32(prelu4(o, act));
        if (ok) {
            float* og = out + (long long)g * 15 + 4 * q;
            og[0] = of.x; og[1] = of.y; og[2] = of.z;
            if (q < 3) og[3] = of.w;
        }
    }
}

// fixed-order reduction of the per-lane global-term partials (rows m = 4q + r < 5 of the fglobal tile) -> gpart_out[m] (the row of one virtual block)
__device__ __forceinline__ void tl_store_gpart(f64x4 accw, int lane, int wave, float* red, float* __restrict__ gpart_out);
__device__ __forceinline__ void tl_store_gpart(f32x4 acc, int lane, int wave, float* red, float* __restrict__ gpart_out) {
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        acc.x += __shfl_xor(acc.x, d); acc.y += __shfl_xor(acc.y, d); acc.z += __shfl_xor(acc.z, d); acc.w += __shfl_xor(acc.w, d);
    }
    const int j = lane & 15, q = lane >> 4;
    __syncthreads();                          // (`red` of the previous virtual block has been read)
    if (j == 0 && q < 2) *(f32x4*)(red + wave * 8 + 4 * q) = acc;
    __syncthreads();
    if (threadIdx.x < 8) {
        float s = 0.f;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k * 8 + threadIdx.x];
        gpart_out[threadIdx.x] = threadIdx.x < 5 ? s : 0.f;
    }
}
// WIDE: the per-lane partials were accumulated in fp64; the cross-lane / cross-wave sums keep the fp32 buffers' fixed order
__device__ __forceinline__ void tl_store_gpart(f64x4 accw, int lane, int wave, float* red, float* __restrict__ gpart_out) {
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        accw.x += __shfl_xor(accw.x, d); accw.y += __shfl_xor(accw.y, d); accw.z += __shfl_xor(accw.z, d); accw.w += __shfl_xor(accw.w, d);
    }
    const int j = lane & 15, q = lane >> 4;
    __syncthreads();                          // (`red` of the previous virtual block has been read)
    if (j == 0 && q < 2) *(f32x4*)(red + wave * 8 + 4 * q) = tl_f32(accw);
    __syncthreads();
    if (threadIdx.x < 8) {
        float s = 0.f;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k * 8 + threadIdx.x];
        gpart_out[threadIdx.x] = threadIdx.x < 5 ? s : 0.f;
    }
}

// Pre-pass of a SpatialAggregation layer (see k_sa_pre): pj[j] = fc1.weight[:, 0:C] x_j and the block partial of
// sum_j outdeg(j) PReLU3(fglobal x_j).
template <int C, bool WIDE>
__global__ __launch_bounds__(256) void k_sa_pre_m(SaArgs a) {
    typedef typename TlV<WIDE>::v V;
    sa_select_window(a);
    __shared__ __attribute__((aligned(16))) float sm[GS_IMG_FLOATS + 32];
    const TlImg im = tl_stage_image(sm, a.img, GS_GROUPS, GS_BIAS);
    float* red = sm + GS_IMG_FLOATS;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const float act3 = im.scal[3];
    const int ntiles = (a.G + 15) / 16;
    for (int vb = blockIdx.x; vb < a.vg; vb += gridDim.x) {
    V acc = tl_wide<WIDE>(tl_zero());
    for (int tile = vb * 4 + wave; tile < ntiles; tile += a.vg * 4) {
        const int g = tile * 16 + j;
        const bool ok = g < a.G;
        const float* row = a.x_in + (long long)(ok ? g : a.G - 1) * C;
        V xb[2];
        if (C == 15) { xb[0] = tl_wide<WIDE>(tl_load15(row, q)); xb[1] = tl_wide<WIDE>(tl_zero()); }
        else { xb[0] = tl_wide<WIDE>(tl_load30(row, 0, q)); xb[1] = tl_wide<WIDE>(tl_load30(row, 1, q)); }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            V pj = mma_block(tl_wide<WIDE>(tl_zero()), TLW(im, GS_PJ(t, 0)), xb[0]);
            if (C == 30) pj = mma_block(pj, TLW(im, GS_PJ(t, 1)), xb[1]);
            if (ok) *(f32x4*)(a.pj_out + (long long)g * 32 + 16 * t + 4 * q) = tl_f32(tl_cout(pj));
        }
        V gl = mma_block(tl_cin<WIDE>(tl_bias(im, 5, q)), TLW(im, GS_FG(0)), xb[0]);
        if (C == 30) gl = mma_block(gl, TLW(im, GS_FG(1)), xb[1]);
        if (ok) acc += prelu4(tl_cout(gl), act3) * (float)a.outdeg[g];
    }
    tl_store_gpart(acc, lane, wave, red, a.gpart_out + vb * 8);
    }
}

// k_bip_out_m + k_sa_pre_m<15> in one launch (the batched tail): the Bipartite output of a node is the input of
// SpatialAggregation1's pre-pass of the same node. Same MFMA chains as the two kernels (bitwise equal results).
template <bool WIDE>
__global__ __launch_bounds__(256) void k_bip_pre_m(const float* __restrict__ part, int T, const float* __restrict__ img_bip, long long part_ws,
                                                  SaArgs a) {
    typedef typename TlV<WIDE>::v V;
    sa_select_window(a);
    __shared__ __attribute__((aligned(16))) float sm[GB2_IMG_FLOATS + 4 * 16 * 36 + GS_IMG_FLOATS + 32];
    const TlImg im = tl_stage_image(sm, img_bip, GB_GROUPS2, GB_BIAS2);
    float* tsc = sm + GB2_IMG_FLOATS;
    const TlImg is = tl_stage_image(tsc + 4 * 16 * 36, a.img, GS_GROUPS, GS_BIAS);
    float* red = tsc + 4 * 16 * 36 + GS_IMG_FLOATS;
    __syncthreads();
    part += blockIdx.y * part_ws;
    const int G = a.G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const int jl = lane >> 2, ql = lane & 3;
    float* ts = tsc + wave * 16 * 36;
    const float act = im.scal[0], act3 = is.scal[3];
    const int ntiles = (G + 15) / 16;
    for (int vb = blockIdx.x; vb < a.vg; vb += gridDim.x) {
    V acc = tl_wide<WIDE>(tl_zero());
    for (int tile = vb * 4 + wave; tile < ntiles; tile += a.vg * 4) {
        const int g = tile * 16 + j;
        const bool ok = g < G;
        const int gl = tile * 16 + jl;
        const float* pg = part + (long long)(gl < G ? gl : G - 1) * T * 32 + 4 * ql;
        const TlRowSum<WIDE> rs = tl_sum_part_rows<WIDE>(pg, T, ts, jl, ql, j, q);
        V ow = tl_cin<WIDE>(tl_bias(im, 0, q));
        ow = mma_block(ow, TLW(im, 0), rs.r0);
        ow = tl_cout(mma_block(ow, TLW(im, 1), rs.r1));
        ow = prelu4(ow, act);
        // the stored row (fp32) is what every later consumer reads: SpatialAggregation1's pre-pass continues from the SAME values
        f32x4 of = tl_f32(ow);
        if (q == 3) of.w = 0.f;                   // channel 15 does not exist (tl_load15 of the stored row reads it as zero)
        if (ok) {
            float* og = a.out + (long long)g * 15 + 4 * q;
            og[0] = of.x; og[1] = of.y; og[2] = of.z;
            if (q < 3) og[3] = of.w;
        }
        const V o = tl_wide<WIDE>(of);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const V pj = mma_block(tl_wide<WIDE>(tl_zero()), TLW(is, GS_PJ(t, 0)), o);
            if (ok) *(f32x4*)(a.pj_out + (long long)g * 32 + 16 * t + 4 * q) = tl_f32(tl_cout(pj));
        }
        const V glb = mma_block(tl_cin<WIDE>(tl_bias(is, 5, q)), TLW(is, GS_FG(0)), o);
        if (ok) acc += prelu4(tl_cout(glb), act3) * (float)a.outdeg[g];
    }
    tl_store_gpart(acc, lane, wave, red, a.gpart_out + vb * 8);
    }
}

// One SpatialAggregation layer (see k_sa_layer): per-edge messages on the VALU (8 channels per lane), fc2 and the next layer's
// pre-pass as MFMA chains on the 16 nodes of the wave.
template <int C, bool NEXT, bool WIDE>
__global__ __launch_bounds__(256) void k_sa_layer_m(SaArgs a) {
    typedef typename TlV<WIDE>::v V;
    sa_select_window(a);
    __shared__ __attribute__((aligned(16))) float sm[GS_IMG_FLOATS + 8 * 32 + 8 + 32 * 8 + 32 + 4 * 16 * 36];
    const TlImg im = tl_stage_image(sm, a.img, GS_GROUPS, GS_BIAS);
    float* w1p = sm + GS_IMG_FLOATS;         // fc1 columns C..C+7 (3 position + 5 global), [k][32]
    float* gsum = w1p + 8 * 32;
    float* gred = gsum + 8;                  // [32][8]
    float* red = gred + 32 * 8;
    float* tsc = red + 32;                   // per wave [16][36]: edge means, row layout -> MFMA layout
    for (int i = threadIdx.x; i < 8 * 32; i += blockDim.x) {
        const int k = i >> 5, cc = i & 31;
        w1p[i] = cc < 30 ? a.raw[a.fc1_w + cc * (C + 8) + C + k] : 0.f;
    }
    {   // global term: the producer's per-block partials in the same fixed two-level order as k_sa_layer
        const int m = threadIdx.x & 7, chunk = threadIdx.x >> 3;
        float sgl = 0.f;
        for (int b = chunk; b < a.n_gpart_in; b += 32) sgl += a.gpart_in[b * 8 + m];
        gred[chunk * 8 + m] = sgl;
        __syncthreads();
        if (threadIdx.x < 8) {
            float t = 0.f;
            for (int k = 0; k < 32; ++k) t += gred[k * 8 + threadIdx.x];
            gsum[threadIdx.x] = t;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    // The per-edge phase is elementwise per (node, channel) and runs in the ROW layout lane = 4 r + cq (node r, chunk cq): four
    // consecutive lanes read one 64-B half of a gathered pj row (in the MFMA layout they read 16-B chunks of four different
    // rows: a quarter of the texture path's rate); the edge means cross a per-wave LDS scratch into the MFMA layout for fc2.
    const int jl = lane >> 2, ql = lane & 3;
    float* ts = tsc + wave * 16 * 36;
    const float act1 = im.scal[0], act2 = im.scal[1], act3n = im.scal[2];
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
    const int ntiles = (a.G + 15) / 16;
    for (int vb = blockIdx.x; vb < a.vg; vb += gridDim.x) {
    V acc = tl_wide<WIDE>(tl_zero());
    for (int tile = vb * 4 + wave; tile < ntiles; tile += a.vg * 4) {
        const int i = tile * 16 + j;
        const bool ok = i < a.G;
        const int ic = ok ? i : a.G - 1;
        const float* row = a.x_in + (long long)ic * C;
        V xb[2];
        if (C == 15) { xb[0] = tl_wide<WIDE>(tl_load15(row, q)); xb[1] = tl_wide<WIDE>(tl_zero()); }
        else { xb[0] = tl_wide<WIDE>(tl_load30(row, 0, q)); xb[1] = tl_wide<WIDE>(tl_load30(row, 1, q)); }
        // ---- edges of node il (row layout; fp32 in both forms: a message is a five-term sum, the mean runs over <= 15 of them)
        const int il = tile * 16 + jl;
        const bool okl = il < a.G;
        const int icl = okl ? il : a.G - 1;
        const float pi0 = a.pos[icl * 3 + 0] / a.scale_rel, pi1 = a.pos[icl * 3 + 1] / a.scale_rel, pi2 = a.pos[icl * 3 + 2] / a.scale_rel;
        const int eb = a.rowptr[icl], ee = okl ? a.rowptr[icl + 1] : eb;
        f32x4 as[2] = {tl_zero(), tl_zero()};
        // edges in chunks of 4: ids, the gathered rows / positions in flight, then the arithmetic in edge order (no cross-lane
        // operation inside: the nodes of a wave may differ in trip count). (Round 5: the ids of 16 edges in one request and rows 8
        // edges at a time -- 3 round trips per node instead of 8 -- changed nothing, 16.4 / 17.9 / 15.0 us against 15.3 / 16.9 /
        // 15.2 for the three layers of a single window: these launches are not bound by this chain. A launch that returns before its
        // tile loop takes 4-5 us: image staging, the global term and the launch itself; the one tile a wave owns is the other ~11 us.)
        for (int e0 = eb; e0 < ee; e0 += 4) {
            int jn[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) jn[k] = a.col[min(e0 + k, ee - 1)];
            f32x4 pjv[4][2];
            float q0[4], q1[4], q2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pjv[k][0] = *(const f32x4*)(a.pj_in + (long long)jn[k] * 32 + 4 * ql);
                pjv[k][1] = *(const f32x4*)(a.pj_in + (long long)jn[k] * 32 + 16 + 4 * ql);
                q0[k] = a.pos[jn[k] * 3 + 0]; q1[k] = a.pos[jn[k] * 3 + 1]; q2[k] = a.pos[jn[k] * 3 + 2];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d0 = pi0 - q0[k] / a.scale_rel, d1 = pi1 - q1[k] / a.scale_rel, d2 = pi2 - q2[k] / a.scale_rel;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 m = pjv[k][t] + base[t];
                    m += wp[0][t] * d0;
                    m += wp[1][t] * d1;
                    m += wp[2][t] * d2;
                    if (e0 + k < ee) as[t] += prelu4(m, act1);
                }
            }
        }
        const float deg = (float)max(ee - eb, 1);
        *(f32x4*)(ts + jl * 36 + 4 * ql) = as[0] / deg;
        *(f32x4*)(ts + jl * 36 + 16 + 4 * ql) = as[1] / deg;
        GSYNC();
        const V av[2] = {tl_wide<WIDE>(*(const f32x4*)(ts + j * 36 + 4 * q)), tl_wide<WIDE>(*(const f32x4*)(ts + j * 36 + 16 + 4 * q))};
        GSYNC();
        V o[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            V v = mma_block(tl_cin<WIDE>(tl_bias(im, t, q)), TLW(im, GS_FC2(t, 0)), xb[0]);
            if (C == 30) v = mma_block(v, TLW(im, GS_FC2(t, 1)), xb[1]);
            v = mma_block(v, TLW(im, GS_FC2(t, 2)), av[0]);
            v = tl_cout(mma_block(v, TLW(im, GS_FC2(t, 3)), av[1]));
            const f32x4 of = tl_f32(prelu4(v, act2));
            if (ok) tl_store30(a.out + (long long)i * 30, t, q, of);
            o[t] = tl_wide<WIDE>(of);            // the next layer's pre-pass continues from the stored (fp32) row
        }
        if (NEXT) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                V pj = mma_block(tl_wide<WIDE>(tl_zero()), TLW(im, GS_PJN(t, 0)), o[0]);
                pj = mma_block(pj, TLW(im, GS_PJN(t, 1)), o[1]);
                if (ok) *(f32x4*)(a.pj_out + (long long)i * 32 + 16 * t + 4 * q) = tl_f32(tl_cout(pj));
            }
            V gl = mma_block(tl_cin<WIDE>(tl_bias(im, 4, q)), TLW(im, GS_FGN(0)), o[0]);
            gl = mma_block(gl, TLW(im, GS_FGN(1)), o[1]);
            if (ok) acc += prelu4(tl_cout(gl), act3n) * (float)a.outdeg[i];
        }
    }
    if (NEXT) tl_store_gpart(acc, lane, wave, red, a.gpart_out + vb * 8);
    }
}

// Per-grid-node part of SpatialAttention's edge Linears (see k_ro_pre), biases included, in a head-padded layout:
// cv[j] = [f_context: head h at 16h + l (l < 15, slot 15 zero) | f_values: 80 + 16h + l], CVP floats per node.
template <bool WIDE>
__global__ __launch_bounds__(256) void k_ro_pre_m(const float* __restrict__ x_spatial, int G, const float* __restrict__ img,
                                                 float* __restrict__ cv, int Gw, long long cv_ws) {
    typedef typename TlV<WIDE>::v V;
    __shared__ __attribute__((aligned(16))) float sm[GP_IMG_FLOATS];
    const TlImg im = tl_stage_image(sm, img, GP_GROUPS, GP_BIAS);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const int ntiles = (G + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int g = tile * 16 + j;
        const bool ok = g < G;
        const int gc = ok ? g : G - 1;
        const float* row = x_spatial + (long long)gc * 30;
        const V xb0 = tl_wide<WIDE>(tl_load30(row, 0, q)), xb1 = tl_wide<WIDE>(tl_load30(row, 1, q));
        const int w = gc / Gw;
        float* o = cv + w * cv_ws + (long long)(gc - w * Gw) * CVP + 4 * q;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int h = 0; h < 5; ++h) {
                V v = mma_block(tl_cin<WIDE>(tl_bias(im, m * 5 + h, q)), TLW(im, GP(m, h, 0)), xb0);
                v = mma_block(v, TLW(im, GP(m, h, 1)), xb1);
                if (ok) *(f32x4*)(o + m * 80 + h * 16) = tl_f32(tl_cout(v));
            }
    }
}

// Read-out heads (see k_readout). MODE 0: y = TemporalAttention(SpatialDirect(x_spatial)) per grid node; MODE 1:
// x = TemporalAttention(SpatialAttention(x_spatial, x_query, x_grid)) per query. SpatialAttention's per-edge arithmetic runs on
// the VALU head by head (a lane holds 4 of a head's 16 slots for its query; the head dot product is a 4-lane butterfly); the
// attention scores of TemporalAttention are an MFMA against the time-query fragments (score[t] = Q_h[t, :] . ctx_h), and the
// score x value products, per node, go through a per-wave LDS scratch (a lane needs all T x 5 scores of its node).
constexpr int RO_SCS = 68;      // floats per node in the score scratch: [5 heads][12 time slots] + pad
constexpr int ROM_LDS_FLOATS = GR_IMG_FLOATS + 5 * 256 + 10 * 80 + 4 * 16 * RO_SCS;
template <int MODE, bool WIDE = false, bool PF = false>      // PF (MODE 1): the gathered rows of head h + 1 requested before head h is worked on
__global__ __launch_bounds__(256) void k_readout_m(RoArgs a) {
    static_assert(!(WIDE && MODE != 0), "the fp64 form exists for the grid read-out (MODE 0)");
    typedef typename TlV<WIDE>::v V;
    typedef typename std::conditional<WIDE, double, float>::type R;
    constexpr int SCW = WIDE ? 2 : 1;        // the score scratch holds R values
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const TlImg im = tl_stage_image(sm, a.img, GR_GROUPS, GR_BIAS);
    float* qf = sm + GR_IMG_FLOATS;          // [5][64][4]: A fragments of the temporal queries, head h
    float* et = qf + 5 * 256;                // [10][80] (MODE 1): f_queries columns 0..2, f_context / f_values edge columns, f_queries bias
    float* scr = et + 10 * 80;
    TlImg imp = im;                          // MODE 0 with cv_out: the PL_ROP image behind the score scratch
    if (MODE == 0 && a.cv_out != nullptr) imp = tl_stage_image(scr + SCW * 4 * 16 * RO_SCS, a.pimg, GP_GROUPS, GP_BIAS);
    {   // qf[h][lane][r] = query[t = lane & 15][head h][l = 4 (lane >> 4) + r], query = temporal_query_2(PReLU3(temporal_query_1(t / scale_t)))  :329
        const float act3 = a.raw[a.o_a3];
        for (int i = threadIdx.x; i < 5 * 256; i += blockDim.x) {
            const int h = i >> 8, ln = (i & 255) >> 2, r = i & 3, t = ln & 15, l = 4 * (ln >> 4) + r;
            float v = 0.f;
            if (t < a.T && l < 15) {
                const int ch = 15 * h + l;
                const float tq = a.t_query[t] / a.scale_t;
                v = a.raw[a.o_q2b + ch];
                for (int k = 0; k < 30; ++k) {
                    const float hq = prelu1(a.raw[a.o_q1w + k] * tq + a.raw[a.o_q1b + k], act3);
                    v += a.raw[a.o_q2w + ch * 30 + k] * hq;
                }
            }
            qf[i] = v;
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const float fa = im.scal[0], sa1 = im.scal[1], act1 = im.scal[2], act2 = im.scal[3], act4 = im.scal[4], act5 = im.scal[5];
    const float b_p2 = im.scal[6];
    const R inv_sqrt_l = WIDE ? (R)(1.0 / sqrt(15.0)) : (R)(1.f / sqrtf(15.f));
    R* ws = (R*)scr + (wave * 16 + j) * RO_SCS;
    const int ntiles = (a.N + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int n = tile * 16 + j;
        const bool ok = n < a.N;
        const int nc = ok ? n : a.N - 1;
        V xin[2];
        if (MODE == 0) {
            const float* row = a.x_spatial + (long long)nc * 30;
            const V xb0 = tl_wide<WIDE>(tl_load30(row, 0, q)), xb1 = tl_wide<WIDE>(tl_load30(row, 1, q));
            if (a.cv_out != nullptr) {                                                  // k_ro_pre_m's work for this node (same MFMA chains)
                const int w = nc / a.Nw;
                float* o = a.cv_out + w * a.cv_ws + (long long)(nc - w * a.Nw) * CVP + 4 * q;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int h = 0; h < 5; ++h) {
                        V v = mma_block(tl_cin<WIDE>(tl_bias(imp, m * 5 + h, q)), TLW(imp, GP(m, h, 0)), xb0);
                        v = mma_block(v, TLW(imp, GP(m, h, 1)), xb1);
                        if (ok) *(f32x4*)(o + m * 80 + h * 16) = tl_f32(tl_cout(v));
                    }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {                                               // SpatialDirect  :258-260
                V y = mma_block(tl_cin<WIDE>(tl_bias(im, t, q)), TLW(im, GR_FRONT(t, 0)), xb0);
                y = mma_block(y, TLW(im, GR_FRONT(t, 1)), xb1);
                xin[t] = prelu4(tl_cout(y), fa);
            }
        } else {
            // SpatialAttention in the ROW layout lane = 4 r + cq (query r = lane >> 2, chunk cq = lane & 3): four consecutive lanes
            // read one 64-B head block of a gathered cv row (the MFMA layout reads 16-B chunks of four rows per quad: a quarter of
            // the texture path's rate on the 6.4 KB a query gathers), the head sums are butterflies inside a quad; the aggregated
            // vector crosses the wave's LDS scratch into the MFMA layout for proj.
            const int jl = lane >> 2, ql = lane & 3;
            const int n_l = tile * 16 + jl;
            const int ncl = n_l < a.N ? n_l : a.N - 1;
            const int wq = ncl / a.Nw, nl = ncl - wq * a.Nw;
            const float* cvw = a.cv + wq * a.cv_ws + 4 * ql;
            const float xq0 = a.x_query[nl * 3 + 0], xq1 = a.x_query[nl * 3 + 1], xq2 = a.x_query[nl * 3 + 2];
            int jn[RO_K];
            float e[RO_K][3];
#pragma unroll
            for (int k = 0; k < RO_K; ++k) jn[k] = a.knn[(long long)nl * RO_K + k];
#pragma unroll
            for (int k = 0; k < RO_K; ++k) {                                            // edge_attr  :283
                e[k][0] = (xq0 - a.x_grid[jn[k] * 3 + 0]) / a.scale_rel;
                e[k][1] = (xq1 - a.x_grid[jn[k] * 3 + 1]) / a.scale_rel;
                e[k][2] = (xq2 - a.x_grid[jn[k] * 3 + 2]) / a.scale_rel;
            }
            f32x4 xm = tl_zero();
            // PF: the context and value rows of head h + 1 are requested before head h is worked on (two register sets, the head loop
            // is unrolled). A call on ONE window gives every wave one tile and a CU four such waves, so each round trip a head waits
            // for is exposed: ten in a row (context, then values, per head); 32 -> 28 us at 10 000 queries. With the tiles of a batch
            // of windows in flight the waves hide each other's round trips and the extra requests only queue (+2.5 %): PF is for
            // single-window calls.
            f32x4 cck[2][RO_K], cvv[2][RO_K];
            unsigned jo[RO_K];
#pragma unroll
            for (int k = 0; k < RO_K; ++k) jo[k] = (unsigned)jn[k] * (unsigned)CVP;
            if (PF) {
#pragma unroll
                for (int k = 0; k < RO_K; ++k) { cck[0][k] = *(const f32x4*)(cvw + jo[k]); cvv[0][k] = *(const f32x4*)(cvw + jo[k] + 80); }
            }
#pragma unroll
            for (int h = 0; h < 5; ++h) {
                const int hl = PF ? h + 1 : h;                  // the head whose rows are requested now
                if (hl < 5) {
#pragma unroll
                    for (int k = 0; k < RO_K; ++k) cck[hl & 1][k] = *(const f32x4*)(cvw + jo[k] + hl * 16);
                }
                const float* eh = et + h * 16 + 4 * ql;
                const f32x4 bq = *(const f32x4*)(eh + 9 * 80);
                f32x4 wq_[3], wc_[3], wv_[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    wq_[d] = *(const f32x4*)(eh + d * 80); wc_[d] = *(const f32x4*)(eh + (3 + d) * 80); wv_[d] = *(const f32x4*)(eh + (6 + d) * 80);
                }
                f32x4 cvk[RO_K];
                float al[RO_K];
#pragma unroll
                for (int k = 0; k < RO_K; ++k) {                                        // alpha = PReLU1(sum_l q c / sqrt(L))  :293
                    f32x4 q4 = bq, c4 = cck[h & 1][k];
#pragma unroll
                    for (int d = 0; d < 3; ++d) { q4 += wq_[d] * e[k][d]; c4 += wc_[d] * e[k][d]; }
                    const f32x4 pr = q4 * c4;
                    al[k] = ((pr.x + pr.y) + pr.z) + pr.w;
                }
                if (hl < 5) {                                   // (not PF: this head's value rows, in flight under the scores)
#pragma unroll
                    for (int k = 0; k < RO_K; ++k) cvv[hl & 1][k] = *(const f32x4*)(cvw + jo[k] + 80 + hl * 16);
                }
#pragma unroll
                for (int k = 0; k < RO_K; ++k) cvk[k] = cvv[h & 1][k];
#pragma unroll
                for (int k = 0; k < RO_K; ++k) {
                    al[k] += __shfl_xor(al[k], 1);
                    al[k] += __shfl_xor(al[k], 2);
                    al[k] = prelu1(al[k] * inv_sqrt_l, sa1);
                }
                float mx = al[0];                                                       // segment softmax over the K edges  :295
#pragma unroll
                for (int k = 1; k < RO_K; ++k) mx = fmaxf(mx, al[k]);
                float ssum = 0.f;
#pragma unroll
                for (int k = 0; k < RO_K; ++k) { al[k] = expf(al[k] - mx); ssum += al[k]; }
                const float den = ssum + 1e-16f;
                f32x4 gh = tl_zero();
#pragma unroll
                for (int k = 0; k < RO_K; ++k) {                                        // 'add' aggregation of alpha * v  :264,297
                    f32x4 v4 = cvk[k];
#pragma unroll
                    for (int d = 0; d < 3; ++d) v4 += wv_[d] * e[k][d];
                    gh += v4 * (al[k] / den);
                }
                xm += gh;
            }
            xm *= 0.2f;                                                                 // mean over heads  :285
            float* wsb = scr + wave * 16 * RO_SCS;
            *(f32x4*)(wsb + jl * RO_SCS + 4 * ql) = xm;                                 // row layout -> MFMA layout
            GSYNC();
            xm = *(const f32x4*)(wsb + j * RO_SCS + 4 * q);
            GSYNC();
#pragma unroll
            for (int t = 0; t < 2; ++t)                                                 // PReLU2(proj(.))  :285
                xin[t] = tl_wide<WIDE>(prelu4(mma_block(tl_bias(im, t, q), TLW(im, GR_FRONT(t, 0)), xm), fa));
        }
        if (a.lat_out != nullptr && ok) {
            tl_store30(a.lat_out + (long long)n * 30, 0, q, tl_f32(xin[0]));
            tl_store30(a.lat_out + (long long)n * 30, 1, q, tl_f32(xin[1]));
        }
        // ------------------------------------------------------------------ TemporalAttention on xin  :325-331
        V h1[2], h2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            V c1 = mma_block(tl_cin<WIDE>(tl_bias(im, 2 + t, q)), TLW(im, GR_C1(t, 0)), xin[0]);
            c1 = mma_block(c1, TLW(im, GR_C1(t, 1)), xin[1]);
            h1[t] = prelu4(tl_cout(c1), act1);
            V v1 = mma_block(tl_cin<WIDE>(tl_bias(im, 4 + t, q)), TLW(im, GR_V1(t, 0)), xin[0]);
            v1 = mma_block(v1, TLW(im, GR_V1(t, 1)), xin[1]);
            h2[t] = prelu4(tl_cout(v1), act2);
        }
        V val[5];
#pragma unroll
        for (int h = 0; h < 5; ++h) {
            V cx = mma_block(tl_cin<WIDE>(tl_bias(im, 6 + h, q)), TLW(im, GR_C2(h, 0)), h1[0]);
            cx = tl_cout(mma_block(cx, TLW(im, GR_C2(h, 1)), h1[1]));
            // score[t, h] = ctx[h, :] . query[t, h, :] / sqrt(L): rows t = 4q + r of the result
            const V sc = tl_cout(mma_block(tl_wide<WIDE>(tl_zero()), ((const f32x4*)qf)[h * 64 + lane], cx)) * inv_sqrt_l;
            if (q < 3) *(V*)(ws + h * 12 + 4 * q) = sc;
            V vx = mma_block(tl_cin<WIDE>(tl_bias(im, 11 + h, q)), TLW(im, GR_V2(h, 0)), h2[0]);
            val[h] = tl_cout(mma_block(vx, TLW(im, GR_V2(h, 1)), h2[1]));
        }
        GSYNC();
        const f32x4 w2a = tl_bias(im, 18, q), w2b = tl_bias(im, 19, q);
#pragma unroll 2
        for (int t = 0; t < a.T; ++t) {
            V z = tl_wide<WIDE>(tl_zero());                                             // z[t, l] = mean_h score[t, h] val[h, l]
#pragma unroll
            for (int h = 0; h < 5; ++h) z += val[h] * ws[h * 12 + t];
            z = prelu4(z * (WIDE ? (R)0.2 : (R)0.2f), act4);
            V pa = prelu4(tl_cout(mma_block(tl_cin<WIDE>(tl_bias(im, 16, q)), TLW(im, GR_P1(0)), z)), act5);        // proj_2(PReLU5(proj_1(.)))
            V pb = prelu4(tl_cout(mma_block(tl_cin<WIDE>(tl_bias(im, 17, q)), TLW(im, GR_P1(1)), z)), act5);
            R o = (R)w2a.x * pa.x;
            o += (R)w2a.y * pa.y; o += (R)w2a.z * pa.z; o += (R)w2a.w * pa.w;
            o += (R)w2b.x * pb.x; o += (R)w2b.y * pb.y; o += (R)w2b.z * pb.z; o += (R)w2b.w * pb.w;
            o += __shfl_xor(o, 16);
            o += __shfl_xor(o, 32);
            if (ok && q == 0) a.out[(long long)n * a.T + t] = (float)(o + (R)b_p2);
        }
        GSYNC();
    }
}

// LocalSliceLgCollapse (module.py:610-659), pick-sized: per pick a the K = 10 product nodes of its station whose theoretical
// arrival is nearest the pick time (time-pointer table A_edges[(ipick * l_dt + t_index) * K + k], :635-640), those within
// 2 eps of the pick time kept (:642-647), message PReLU1(fc1[s[e] || (tpick - tlatent[e]) / eps || phase]) (:657-659), 'mean'
// over the kept edges (:612), PReLU2(fc2 .) (:651). A wave owns 16 picks (fp32-MFMA tile layout of the tail kernels).
constexpr int LS_K = 10;
struct LsArgs {
    int n_picks, l_dt;
    long long n_edges;            // entries of the time-pointer table (indices are clamped into it)
    float t0, dt, eps;
    const float* dtp;             // device copy of dt_partition (or null): t0 = dtp[0], dt = dtp[1] - dtp[0] replace the two above
    const float* s;               // [P, 30] association embedding (genie_assoc_fwd)
    const int32_t* A_edges;       // [n_sta * l_dt * K] product-node ids
    const float* tlatent; int tl_stride, tl_col;     // theoretical arrival of product node e: tlatent[e * tl_stride + tl_col]
    const float* tpick; const int32_t* ipick; const float* phase;
    const float* img;
    float* out;                   // [n_picks, 15]
    unsigned* flag;               // host-mapped word: bit 0 set when a pick indexes outside the time-pointer table (genie_index_flags)
};
__global__ __launch_bounds__(256) void k_lslc(LsArgs a) {
    __shared__ __attribute__((aligned(16))) float sm[GL_IMG_FLOATS];
    const TlImg im = tl_stage_image(sm, a.img, GL_GROUPS, GL_BIAS);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const float act1 = im.scal[0], act2 = im.scal[1];
    const float t0_ = a.dtp ? a.dtp[0] : a.t0, dt_ = a.dtp ? a.dtp[1] - a.dtp[0] : a.dt;
    const int ntiles = (a.n_picks + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int p = tile * 16 + j;
        const bool ok = p < a.n_picks;
        const int pc = ok ? p : a.n_picks - 1;
        const float tp = a.tpick[pc], ph = a.phase[pc];
        const int ti = (int)floorf((tp - t0_) / dt_);                                     // :635
        const int ipk = a.ipick[pc];
        long long base = ((long long)ipk * a.l_dt + ti) * LS_K;
        // the reference indexes the table with these and fails on an index outside it (module.py:635-640: a device-side assertion,
        // reported at the next synchronisation); here the index is clamped and the call that follows on the host raises
        if (ok && (ti < 0 || ti >= a.l_dt || ipk < 0 || base + LS_K > a.n_edges) && a.flag != nullptr)
            __hip_atomic_fetch_or(a.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        base = base < 0 ? 0 : (base > a.n_edges - LS_K ? a.n_edges - LS_K : base);
        f32x4 acc[2] = {tl_zero(), tl_zero()};
        float cnt = 0.f;
#pragma unroll 2
        for (int k = 0; k < LS_K; ++k) {
            const int e = a.A_edges[base + k];
            const float rt = tp - a.tlatent[(long long)e * a.tl_stride + a.tl_col];
            const bool keep = ok && fabsf(rt) < 2.0f * a.eps;                             // :642-645
            const float* row = a.s + (long long)e * 30;
            const f32x4 xb0 = tl_load30(row, 0, q), xb1 = tl_load30(row, 1, q);
            const float xs = q == 0 ? rt / a.eps : (q == 1 ? ph : 0.f);                   // columns 30, 31 of fc1
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 m = mma_block(tl_bias(im, t, q), TLW(im, GL_FC1(t, 0)), xb0);
                m = mma_block(m, TLW(im, GL_FC1(t, 1)), xb1);
                m = MFMA16(TLW(im, GL_FC1(t, 2)).x, xs, m);
                m = prelu4(m, act1);
                if (keep) acc[t] += m;
            }
            cnt += keep ? 1.f : 0.f;
        }
        const float den = fmaxf(cnt, 1.f);
        f32x4 o = mma_block(tl_bias(im, 2, q), TLW(im, GL_FC2(0)), acc[0] / den);
        o = mma_block(o, TLW(im, GL_FC2(1)), acc[1] / den);
        o = prelu4(o, act2);
        if (ok) {
            float* og = a.out + (long long)p * 15 + 4 * q;
            og[0] = o.x; og[1] = o.y; og[2] = o.z;
            if (q < 3) og[3] = o.w;
        }
    }
}

// StationSourceAttentionMergedPhases (module.py:662-775, use_sparse = True, use_neighbor_assoc_edges = False), pick-sized. For
// every source i and pick a the reference attends over the picks b of a's station plus a null pick (:703-718), keeps the edges
// whose observed - theoretical arrival time (P or S) is inside 2 eps (:722-729) -- a property of (b, i) alone --, and runs three
// edge MLPs; queries and the relative-time features depend on (b, i) only, the context on (i, self_link, null_link), the values
// on (b, i, self_link, null_link). One workgroup per (source i, station u with picks):
//   A1  keep flags of the station's picks + the null pick, compacted in order into an LDS list;
//   A2  per kept b (16 per wave, fp32-MFMA tiles): query -> the three head scores against the context of a plain edge and of a
//       self edge (the null pick: of a null edge), values of a plain edge and of a self edge (null pick: of a null edge) -> LDS;
//   B   per target a of the station (16 per wave): segment softmax over the kept list (the entry b == a takes its self variant),
//       'add' aggregation, mean over heads, proj_2(PReLU4(proj_1(.))).
// k_arr_ctx prepares the three context vectors of every source. Needs at least one source with |stime| < 2 eps (then the
// reference's `edge_index[0].max()` (:762-763) is the null pick, as assumed here); the host checks it and otherwise keeps the
// PyTorch restatement. The softmax runs in its streaming form over chunks of AR_CAP picks (running maximum, denominator and
// weighted value sum per target; the sum is divided by (denominator + 1e-16) at the end instead of every weight being divided
// first: rounding-order difference only), so a station may hold any number of picks.
constexpr int AR_CAP = 192;       // picks of a station (null included) per LDS chunk
constexpr int AR_ENT = 104;       // floats per kept entry: scores plain [3], self [3], pad 2, values plain [3][16], self [3][16]
constexpr int AT_STAT = 64;       // training forward, per (source, pick): normalised head aggregates [3][16], running max [3], denominator [3]
struct ArArgs {
    int n_src, n_sta, n_arv, n_useg;
    float eps;
    const float* stime;           // [n_src]
    const float* trv_src;         // [n_src, n_sta, 2]
    const float* ctx;             // [n_src][4][48] (k_arr_ctx): plain, self, null, self + null edge; head h at 16h
    const float* arv_p; const float* arv_s;          // [n_arv, 15]
    const float* tpick; const float* phase;          // [n_arv]
    const int32_t* order;         // picks sorted by station (stable)
    const int32_t* seg_sta; const int32_t* seg_start; const int32_t* seg_len;     // [n_useg] stations with picks
    const float* img;
    float* out;                   // [n_src, n_arv, 2]
    int* e0max;                   // [1] `edge_index[0].max()` over the kept edges (module.py:762-763), by k_arr_e0max: the pick the
                                  // reference treats as "the null pick"; = n_arv (the real null pick) whenever some source has
                                  // |stime| < 2 eps, i.e. always in practice
    float* save;                  // training forward: [n_src * n_arv][AT_STAT] (else null)
};

// e0max = max over the kept (pick b, source i) pairs of b, the null pick (index n_arv) included (module.py:740-763). A pick's
// edges towards source i survive the 2-eps filter or not as a whole (the test involves only b and i), and every pick has at
// least its self pair, so "b has a kept edge towards i" = "its test passes".
__global__ __launch_bounds__(256) void k_arr_e0max(ArArgs a) {
    const int i = blockIdx.x / a.n_useg, ug = blockIdx.x - i * a.n_useg;
    const int u = a.seg_sta[ug], r0 = a.seg_start[ug], L = a.seg_len[ug];
    const float eps = a.eps, st = a.stime[i];
    const float tp_src = a.trv_src[((long long)i * a.n_sta + u) * 2 + 0] + st, ts_src = a.trv_src[((long long)i * a.n_sta + u) * 2 + 1] + st;
    int best = -1;
    if (threadIdx.x == 0 && ug == 0 && fabsf(st) < 2.f * eps) best = a.n_arv;
    for (int r = threadIdx.x; r < L; r += blockDim.x) {
        const int b = a.order[r0 + r];
        const float tp = a.tpick[b];
        if (fabsf(tp - tp_src) < 2.f * eps || fabsf(tp - ts_src) < 2.f * eps) best = max(best, b);
    }
    if (best >= 0) atomicMax(a.e0max, best);
}

__global__ __launch_bounds__(128) void k_arr_ctx(const float* __restrict__ raw, int o_c1w, int o_c1b, int o_c2w, int o_c2b, int o_a1,
                                                const float* __restrict__ src_embed, const float* __restrict__ stime, int n_src,
                                                float* __restrict__ ctx) {
    __shared__ float hid[4][32];
    const int i = blockIdx.x;
    if (i >= n_src) return;
    const float act1 = raw[o_a1];
    for (int idx = threadIdx.x; idx < 4 * 30; idx += blockDim.x) {
        const int v = idx / 30, c = idx - v * 30;
        float t = raw[o_c1b + c];
        for (int k = 0; k < 30; ++k) t += raw[o_c1w + c * 33 + k] * src_embed[(long long)i * 30 + k];
        t += raw[o_c1w + c * 33 + 30] * stime[i];
        if (v & 1) t += raw[o_c1w + c * 33 + 31];      // self_link
        if (v & 2) t += raw[o_c1w + c * 33 + 32];      // null_link
        hid[v][c] = prelu1(t, act1);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 4 * 48; idx += blockDim.x) {
        const int v = idx / 48, r = idx - v * 48, h = r >> 4, l = r & 15;
        float t = 0.f;
        if (l < 15) {
            const int ch = 15 * h + l;
            t = raw[o_c2b + ch];
            for (int k = 0; k < 30; ++k) t += raw[o_c2w + ch * 30 + k] * hid[v][k];
        }
        ctx[((long long)i * 4 + v) * 48 + r] = t;
    }
}

__global__ __launch_bounds__(256) void k_arrivals(ArArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const TlImg im = tl_stage_image(sm, a.img, GA_GROUPS2, GA_BIAS2);
    float* ent = sm + GA2_IMG_FLOATS;                 // [AR_CAP][AR_ENT]
    int* kept = (int*)(ent + AR_CAP * AR_ENT);        // [AR_CAP] position r in the station's pick list (L = the null pick)
    int* kbi = kept + AR_CAP;                         // [AR_CAP] its pick index (n_arv = the null pick)
    float* cx = (float*)(kbi + AR_CAP);               // [4][48] context vectors of this source
    int* wcnt = (int*)(cx + 192);                     // [4] per-wave counts of the compaction, [4] = total
    const int i = blockIdx.x / a.n_useg, ug = blockIdx.x - i * a.n_useg;
    const int u = a.seg_sta[ug], r0 = a.seg_start[ug], L = a.seg_len[ug];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const float eps = a.eps, st = a.stime[i];
    const float tp_src = a.trv_src[((long long)i * a.n_sta + u) * 2 + 0] + st, ts_src = a.trv_src[((long long)i * a.n_sta + u) * 2 + 1] + st;
    const float rel_null = -eps - (-eps + st);        // null pick: atime = -eps, theoretical time = -eps (:722-725)
    for (int k = threadIdx.x; k < 192; k += blockDim.x) cx[k] = a.ctx[(long long)i * 192 + k];
    // the pick index the reference takes for the null pick (:762-765): n_arv unless NO source keeps the real null pick
    const int E = *a.e0max;
    __syncthreads();      // the weight image and cx are complete before any wave reads them (the slopes and proj_2 rows below!)
    const float act2 = im.scal[0], act3 = im.scal[1], act4 = im.scal[2];
    const float e2 = eps * eps, sq = sqrtf(15.f);
    const f32x4 w2[2][2] = {{tl_bias(im, 12, q), tl_bias(im, 13, q)}, {tl_bias(im, 14, q), tl_bias(im, 15, q)}};
    // Targets in blocks of 256 (4 tiles of 16 per wave, their softmax state in registers); the station's picks + the null pick
    // (r = 0 .. L) in chunks of AR_CAP: A1 / A2 fill the LDS list with the kept picks of the chunk, B folds them into the running
    // (max, denominator, weighted value sum) of every target (one chunk = the plain two-pass segment softmax).
    for (int tb0 = 0; tb0 < L; tb0 += 256) {
        float mx[4][3], den[4][3];
        f32x4 agg[4][3];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int h = 0; h < 3; ++h) { mx[tt][h] = -INFINITY; den[tt][h] = 0.f; agg[tt][h] = tl_zero(); }
        for (int cb = 0; cb <= L; cb += AR_CAP) {
            __syncthreads();                          // the previous chunk's list is no longer read
            // ---- A1: ordered compaction of the kept picks of the chunk
            {
                const int r = cb + (int)threadIdx.x;
                bool keep = false;
                if ((int)threadIdx.x < AR_CAP) {
                    if (r < L) {
                        const float tp = a.tpick[a.order[r0 + r]];
                        keep = fabsf(tp - tp_src) < 2.f * eps || fabsf(tp - ts_src) < 2.f * eps;
                    } else if (r == L) keep = fabsf(rel_null) < 2.f * eps;
                }
                const unsigned long long bal = __ballot(keep);
                if (lane == 0) wcnt[wave] = __popcll(bal);
                __syncthreads();
                int off = 0;
                for (int k = 0; k < wave; ++k) off += wcnt[k];
                if (keep) {
                    const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
                    kept[pos] = r;
                    kbi[pos] = r == L ? a.n_arv : a.order[r0 + r];
                }
                if (threadIdx.x == 0) wcnt[4] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
                __syncthreads();
            }
            const int K = wcnt[4];
            // ---- A2: queries / scores / values of the kept picks
            for (int tile = wave; tile * 16 < K; tile += 4) {
                const int kk = tile * 16 + j;
                const bool ok = kk < K;
                const int r = kept[ok ? kk : K - 1];
                const bool nul = r == L;
                const int b = nul ? 0 : a.order[r0 + r];
                const bool nl = (nul ? a.n_arv : b) == E;          // null_link of this pick's edges (:765)
                const float tp = nul ? 0.f : a.tpick[b];
                const float rp = nul ? rel_null : tp - tp_src, rs = nul ? rel_null : tp - ts_src;
                const float ph = nul ? -1.f : a.phase[b];
                const float f6[6] = {expf(-0.5f * (rp * rp) / e2), (rp > 0.f) - (rp < 0.f) + 0.f, ph,
                                     expf(-0.5f * (rs * rs) / e2), (rs > 0.f) - (rs < 0.f) + 0.f, ph};
                const float x0 = q == 0 ? f6[0] : (q == 1 ? f6[1] : (q == 2 ? f6[2] : f6[3]));     // columns 30 + q
                const float x1 = q == 0 ? f6[4] : (q == 1 ? f6[5] : 0.f);                          // columns 34 + q
                const f32x4 xp = nul ? tl_zero() : tl_load15(a.arv_p + (long long)b * 15, q);
                const f32x4 xs = nul ? tl_zero() : tl_load15(a.arv_s + (long long)b * 15, q);
                f32x4 hq[2], hv[2], hw[2];   // hidden layers: query, values of an edge without / with self_link (null_link = nl in both)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 z = mma_block(tl_bias(im, t, q), TLW(im, GA_Q1(t, 0)), xp);
                    z = mma_block(z, TLW(im, GA_Q1(t, 1)), xs);
                    z = MFMA16(TLW(im, GA_Q1(t, 2)).x, x0, z);
                    z = MFMA16(TLW(im, GA_Q1(t, 2)).y, x1, z);
                    hq[t] = prelu4(z, act2);
                    f32x4 v = mma_block(tl_bias(im, 2 + t, q), TLW(im, GA_V1(t, 0)), xp);
                    v = mma_block(v, TLW(im, GA_V1(t, 1)), xs);
                    v = MFMA16(TLW(im, GA_V1(t, 2)).x, x0, v);
                    v = MFMA16(TLW(im, GA_V1(t, 2)).y, x1, v);
                    const float lnk = (q == 1 && nl) ? 1.f : 0.f;          // k-step [self_link, null_link]: lane q = 0 / 1 supplies it
                    const f32x4 vb = MFMA16(TLW(im, GA_V1(t, 3)).x, lnk, v);
                    const f32x4 w = MFMA16(TLW(im, GA_V1(t, 3)).x, q == 0 ? 1.f : lnk, v);
                    hv[t] = prelu4(vb, act3);
                    hw[t] = prelu4(w, act3);
                }
                float* eo = ent + (long long)(ok ? kk : K - 1) * AR_ENT;
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    f32x4 qh = mma_block(tl_bias(im, 4 + h, q), TLW(im, GA_Q2(h, 0)), hq[0]);
                    qh = mma_block(qh, TLW(im, GA_Q2(h, 1)), hq[1]);
                    // context of an edge without / with self_link (variants: bit 0 self_link, bit 1 null_link)
                    const f32x4 c0 = *(const f32x4*)(cx + (nl ? 96 : 0) + h * 16 + 4 * q), c1 = *(const f32x4*)(cx + (nl ? 144 : 48) + h * 16 + 4 * q);
                    const f32x4 p0 = qh * c0, p1 = qh * c1;
                    float s0 = ((p0.x + p0.y) + p0.z) + p0.w, s1 = ((p1.x + p1.y) + p1.z) + p1.w;
                    s0 += __shfl_xor(s0, 16); s0 += __shfl_xor(s0, 32);
                    s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
                    f32x4 vh = mma_block(tl_bias(im, 7 + h, q), TLW(im, GA_V2(h, 0)), hv[0]);
                    vh = mma_block(vh, TLW(im, GA_V2(h, 1)), hv[1]);
                    f32x4 wh = mma_block(tl_bias(im, 7 + h, q), TLW(im, GA_V2(h, 0)), hw[0]);
                    wh = mma_block(wh, TLW(im, GA_V2(h, 1)), hw[1]);
                    if (ok) {
                        if (q == 0) { eo[h] = s0 / sq; eo[3 + h] = s1 / sq; }
                        *(f32x4*)(eo + 8 + h * 16 + 4 * q) = vh;
                        *(f32x4*)(eo + 56 + h * 16 + 4 * q) = wh;
                    }
                }
            }
            __syncthreads();
            // ---- B: fold the chunk into the targets' softmax state
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int r = tb0 + (tt * 4 + wave) * 16 + j;
                if (tb0 + (tt * 4 + wave) * 16 >= L || K == 0) continue;          // (uniform per wave)
                // self_link = (e0 == e1 mod e0max), e1 = a + i n_arv (:764): the target pick itself when e0max = n_arv
                const int tsel = (r < L && E > 0) ? (int)(((long long)a.order[r0 + r] + (long long)i * a.n_arv) % E) : -1;
                float cm[3] = {mx[tt][0], mx[tt][1], mx[tt][2]};
                for (int k = 0; k < K; ++k) {
                    const float* e = ent + k * AR_ENT + (kbi[k] == tsel ? 3 : 0);
#pragma unroll
                    for (int h = 0; h < 3; ++h) cm[h] = fmaxf(cm[h], e[h]);
                }
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    const float sc = mx[tt][h] == -INFINITY ? 0.f : expf(mx[tt][h] - cm[h]);
                    den[tt][h] *= sc; agg[tt][h] *= sc; mx[tt][h] = cm[h];
                }
                for (int k = 0; k < K; ++k) {
                    const bool self = kbi[k] == tsel;
                    const float* e = ent + k * AR_ENT;
#pragma unroll
                    for (int h = 0; h < 3; ++h) {
                        const float ex = expf(e[(self ? 3 : 0) + h] - cm[h]);
                        den[tt][h] += ex;
                        agg[tt][h] += *(const f32x4*)(e + (self ? 56 : 8) + h * 16 + 4 * q) * ex;
                    }
                }
            }
        }
        // ---- every pick of the block: normalise, mean over heads (:760), proj_2(PReLU4(proj_1(.)))
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int rb = tb0 + (tt * 4 + wave) * 16;
            if (rb >= L) continue;
            const int r = rb + j;
            const bool ok = r < L;
            const f32x4 z = ((agg[tt][0] / (den[tt][0] + 1e-16f) + agg[tt][1] / (den[tt][1] + 1e-16f)) + agg[tt][2] / (den[tt][2] + 1e-16f)) / 3.f;
            if (a.save && ok) {
                float* sv = a.save + ((long long)i * a.n_arv + a.order[r0 + r]) * AT_STAT;
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    *(f32x4*)(sv + 16 * h + 4 * q) = agg[tt][h] / (den[tt][h] + 1e-16f);
                    if (q == 0) { sv[48 + h] = mx[tt][h]; sv[51 + h] = den[tt][h]; }
                }
            }
            f32x4 pa = prelu4(mma_block(tl_bias(im, 10, q), TLW(im, GA_P1(0)), z), act4);
            f32x4 pb = prelu4(mma_block(tl_bias(im, 11, q), TLW(im, GA_P1(1)), z), act4);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float o = w2[m][0].x * pa.x;
                o += w2[m][0].y * pa.y; o += w2[m][0].z * pa.z; o += w2[m][0].w * pa.w;
                o += w2[m][1].x * pb.x; o += w2[m][1].y * pb.y; o += w2[m][1].z * pb.z; o += w2[m][1].w * pb.w;
                o += __shfl_xor(o, 16);
                o += __shfl_xor(o, 32);
                if (ok && q == 0) a.out[((long long)i * a.n_arv + a.order[r0 + r]) * 2 + m] = o + im.scal[3 + m];
            }
        }
    }
}
