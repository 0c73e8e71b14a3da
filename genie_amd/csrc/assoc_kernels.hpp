// Part of genie_hip.hip (one translation unit, included inside its anonymous namespace): the association heads' P-sized kernels (k_assoc_pre / _ps / _a / _b).

// ------------------------------------------------------------------------------------------------
// Association heads on the product graph: BipartiteGraphReadOutOperator (module.py:333-352) and
// DataAggregationAssociationPhase (:356-403) as three P-sized passes in the layout of the DataAggregation kernels
// (fp32 MFMA, tile = 16 stations of one source node, outputs of one Linear are the B operands of the next):
//   k_assoc_pre (G-sized): per-source-node terms pg[g] (the y_latent part of fc1, mask1 and the mask1 columns)
//   k_assoc_a: s = PReLU2(fc2(mask1 PReLU1(fc1[y_latent[g] || e_p])))                       :343-352
//              tr = PReLU(init_trns[s || x_latent || mask1 || Mask]), q1 = PReLU11(l1_t1_1 tr), q2 = PReLU12(l1_t2_1 tr)   :389-396
//   k_assoc_b: tr1 = PReLU1([l1_t1_2[tr || mean_sta q1 || mask] || l1_t2_2[tr || mean_src q2 || mask]])             :397-398
//              r1 = PReLU21(l2_t1_1 tr1), r2 = PReLU22(l2_t2_1 tr1); c / wu / wv exactly as stage 1 leaves them     :399-400
//   stage-2 kernel with `no_bip` (second pair of means + PReLU2 -> [P, 30])                                          :401
// Unlike DataAggregation the first-layer gather operand q is not a function of 8 raw floats, so q1 / q2 (32-float rows) are
// stored and gathered. tr / q1 / q2 and c / wu / wv live in (station) processing order like the stage-1 outputs.
// ------------------------------------------------------------------------------------------------
struct AsArgs {
    int S, G, T, seg, nxcd;
    const int32_t* order;
    const int32_t* sta_rowptr; const int32_t* sta_col; const int32_t* src_rowptr; const int32_t* src_col;
    const int32_t* sta_user;     // internal station -> caller's station (inputs are in the caller's order), or null
    const float* pg;             // [G][AS_PG]
    const float* ps;             // [S][AS_PS] static per-station terms of the two model variants (caller's station order), or null
    const float* x_latent; const float* mask; const float* edge_attr;   // caller's order: [P,30], [P,4], [P,3]
    float* tr; float* q1; float* q2;                                    // [P,32]
    float* c; float* wu; float* wv;
    const float* packed;
    float* save; long long Pn;   // training forward: pre-activations kept for the backward passes, [AV_*][Pn][16], or null
    const int32_t* ptile;        // PCSR: processing order of the 16-node tiles (PtileIter), or null
    const int32_t* src_of;       // PCSR (irregular product graph, `use_subgraph`): source node of every product node; the CSR arrays
                                 // above are then the PRODUCT-level ones (indexed by product node, columns = product-node ids)
};
// blocks of the association phase's saved pre-activations: BipartiteGraphReadOutOperator fc1 (before PReLU and the mask gate) and
// fc2, init_trns, l1_t1_1 / l1_t2_1 (w, tile), [10, 11] = the output layer (written by the stage-2 kernel as its SV_O), layer 1
// (half, tile), l2_t1_1 / l2_t2_1 (w, tile)
constexpr int AV_Z1 = 0, AV_SV = 2, AV_TR = 3, AV_Q = 5, AV_O = 10, AV_T = 12, AV_UV = 16, AV_BLOCKS = 20;
static_assert(AV_O == SV_O, "the stage-2 kernel stores the output layer's pre-activations at SV_O");

struct AsPreOffs { int ro_fc1_w, ro_fc1_b, as_init_w, as_l1t12_w, as_l1t22_w, as_l2t12_w, as_l2t22_w;
                   int as_init_abs, as_l1t12_p, as_l1t22_p, as_l2t12_p, as_l2t22_p; };

// The two model variants in the association phase: under use_updated_model_definition the mean edge feature of a node's
// in-neighbourhood (static: mpos_sta [S][4] / mpos_src [G][4], genie_set_edge_features) enters l1_t?_2 / l2_t?_2 (module.py:462-467,
// :472-480); under use_absolute_pos the station / source positions / (3 scale_rel) (abs_sta [S][4] / abs_src [G][4]) are appended to
// the head's input (module.py:987-988, 6 more columns of init_trns). Both are per-station / per-source-node ADDITIVE terms of a
// pre-activation: the source-node ones are folded into pg, the station ones are ps [S][AS_PS]: [0:30] init_trns, [32:62]
// l1_t1_2, [64:79] l2_t1_2. On an irregular product graph under use_updated_model_definition the mean edge feature of a node runs over its
// PRESENT neighbours: ps is then [n_prod][AS_PS], indexed by product node, and also carries the source-side terms ([80:110] l1_t2_2,
// [112:127] l2_t2_2) that pg holds per source node on Cartesian graphs.
constexpr int AS_PS = 128;
__device__ __forceinline__ float dot4w(const float* __restrict__ w, const float* __restrict__ m, int n) {
    float v = 0.f;
    for (int c = 0; c < n; ++c) v = fmaf(w[c], m[c], v);
    return v;
}

__global__ __launch_bounds__(256) void k_assoc_pre(const float* __restrict__ raw, AsPreOffs o, const float* __restrict__ y_latent,
                                                   const float* __restrict__ mask_src, int G, const float* __restrict__ mpos_src,
                                                   const float* __restrict__ abs_src, float* __restrict__ pg) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= G * AS_PG) return;
    const int g = idx / AS_PG, k = idx - g * AS_PG;
    const float m = mask_src[g];
    float v = 0.f;
    if (k < 30) {
        v = raw[o.ro_fc1_b + k];
        const float* w = raw + o.ro_fc1_w + k * 33;
        for (int c = 0; c < 30; ++c) v = fmaf(w[c], y_latent[g * 30 + c], v);
    } else if (k == 31) v = m;
    else if (k >= 32 && k < 62) {
        v = m * raw[o.as_init_w + (k - 32) * 50 + 45];
        if (abs_src) v += dot4w(raw + o.as_init_abs + (k - 32) * 6 + 3, abs_src + g * 4, 3);
    } else if (k >= 64 && k < 94) v = m * raw[o.as_l1t12_w + (k - 64) * 65 + 60];
    else if (k >= 96 && k < 126) {
        v = m * raw[o.as_l1t22_w + (k - 96) * 65 + 60];
        if (mpos_src) v += dot4w(raw + o.as_l1t22_p + (k - 96) * 4, mpos_src + g * 4, 4);
    } else if (k >= 128 && k < 143) v = m * raw[o.as_l2t12_w + (k - 128) * 95 + 90];
    else if (k >= 144 && k < 159) {
        v = m * raw[o.as_l2t22_w + (k - 144) * 95 + 90];
        if (mpos_src) v += dot4w(raw + o.as_l2t22_p + (k - 144) * 4, mpos_src + g * 4, 4);
    }
    pg[idx] = v;
}

__global__ void k_assoc_ps(const float* __restrict__ raw, AsPreOffs o, long long S, const float* __restrict__ mpos_sta,
                           const float* __restrict__ abs_sta, const float* __restrict__ mpos_src_p, const float* __restrict__ abs_src_p,
                           float* __restrict__ ps) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * AS_PS) return;
    const long long s = idx / AS_PS;
    const int k = (int)(idx - s * AS_PS);
    float v = 0.f;
    if (k < 30) {
        if (abs_sta) v = dot4w(raw + o.as_init_abs + k * 6, abs_sta + s * 4, 3);
        if (abs_src_p) v += dot4w(raw + o.as_init_abs + k * 6 + 3, abs_src_p + s * 4, 3);      // rows = product nodes: the source side too
    }
    else if (k >= 32 && k < 62) { if (mpos_sta) v = dot4w(raw + o.as_l1t12_p + (k - 32) * 4, mpos_sta + s * 4, 4); }
    else if (k >= 64 && k < 79) { if (mpos_sta) v = dot4w(raw + o.as_l2t12_p + (k - 64) * 4, mpos_sta + s * 4, 4); }
    else if (k >= 80 && k < 110) { if (mpos_src_p) v = dot4w(raw + o.as_l1t22_p + (k - 80) * 4, mpos_src_p + s * 4, 4); }      // rows = product nodes
    else if (k >= 112 && k < 127) { if (mpos_src_p) v = dot4w(raw + o.as_l2t22_p + (k - 112) * 4, mpos_src_p + s * 4, 4); }
    ps[idx] = v;
}

__device__ __forceinline__ f32x4 ld_row30(const float* row, int b, int q) {     // channels 16b + 4q .. +3 of a 30-float row (8-B aligned)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 lo = *(const f32x2*)(row + 16 * b + 4 * q);
    f32x2 hi = {0.f, 0.f};
    if (b == 0 || q < 3) hi = *(const f32x2*)(row + 16 * b + 4 * q + 2);
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}

// PCSR: irregular product graph: a tile is 16 consecutive product nodes, the source node (hence the pg row) is per lane
template <bool PCSR>
__global__ __launch_bounds__(256) void k_assoc_a(AsArgs a) {
    constexpr int NF4 = (GA_GROUPS * 256 + GA_BIAS * 16 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lbias = (const float*)(lw + GA_GROUPS * 64);
    const float* lscal = lbias + GA_BIAS * 16;
    const float r1 = lscal[0], r2 = lscal[1], a0 = lscal[2], a11 = lscal[3], a12 = lscal[4];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = a.S;
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave);
    PtileIter ptw(PCSR ? (a.Pn + 15) / 16 : 0, blockDim.x >> 6, wave);      // PCSR: positions in the processing order of the tiles
    const long long n_items = PCSR ? ptw.end : w.nitems;
    const long long it0 = PCSR ? ptw.i : w.it;
    const long long its = PCSR ? ptw.stride : w.stride;
    for (long long it = it0; it < n_items; it += its) {
        int gi = 0, tb = 0, g, su;
        bool valid;
        long long pi, pu;
        if (PCSR) {
            const long long pr = ptile_at(a.ptile, it) * 16 + j;
            valid = pr < a.Pn;
            pi = pu = valid ? pr : a.Pn - 1;
            g = a.src_of[pi];
            su = (int)pi;          // (ps, when present, is per product node on an irregular graph)
        } else {
            w.decode(it, gi, tb);
            g = __builtin_amdgcn_readfirstlane(a.order[gi]);
            const int s = tb * 16 + j;
            valid = s < S;
            const int sc = valid ? s : S - 1;
            su = a.sta_user != nullptr ? a.sta_user[sc] : sc;
            pi = (long long)g * S + sc; pu = (long long)g * S + su;
        }
        asm volatile("" : "+v"(lane));
        const float* pg = a.pg + (long long)g * AS_PG;
        const float eq = q < 3 ? a.edge_attr[pu * 3 + q] : 0.f;
        const float mq = a.mask[pu * 4 + q];
        const float m1 = pg[31];
        const f32x4 lat0 = ld_row30(a.x_latent + pu * 30, 0, q), lat1 = ld_row30(a.x_latent + pu * 30, 1, q);
        // BipartiteGraphReadOutOperator: one edge per product node, so aggr 'add' is the identity (module.py:343-352)
        f32x4 msg[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 z = *(const f32x4*)(pg + 16 * t + 4 * q);
            if (t == 1 && q == 3) z.w = 0.f;                                   // slot 31 carries mask1, not a channel
            z = MFMA16(lw[GA_FC1E(t) * 64 + lane].x, eq, z);
            if (a.save != nullptr && valid) *(f32x4*)(a.save + ((size_t)(AV_Z1 + t) * a.Pn + pi) * 16 + 4 * q) = z;
            msg[t] = prelu4u(z, r1) * m1;
        }
        f32x4 sv = *(const f32x4*)(lbias + 0 * 16 + 4 * q);
        sv = mma_block(sv, lw[GA_FC2(0) * 64 + lane], msg[0]);
        sv = mma_block(sv, lw[GA_FC2(1) * 64 + lane], msg[1]);
        if (a.save != nullptr && valid) *(f32x4*)(a.save + ((size_t)AV_SV * a.Pn + pi) * 16 + 4 * q) = sv;
        sv = prelu4u(sv, r2);
        // init_trns [s || x_latent || mask1 || Mask]
        f32x4 tr[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 acc = *(const f32x4*)(lbias + (1 + t) * 16 + 4 * q) + *(const f32x4*)(pg + 32 + 16 * t + 4 * q);
            if (a.ps != nullptr) acc += *(const f32x4*)(a.ps + (long long)su * AS_PS + 16 * t + 4 * q);
            acc = mma_block(acc, lw[GA_INIT(t, 0) * 64 + lane], sv);
            acc = mma_block(acc, lw[GA_INIT(t, 1) * 64 + lane], lat0);
            acc = mma_block(acc, lw[GA_INIT(t, 2) * 64 + lane], lat1);
            acc = MFMA16(lw[GA_INIT(t, 3) * 64 + lane].x, mq, acc);
            if (a.save != nullptr && valid) *(f32x4*)(a.save + ((size_t)(AV_TR + t) * a.Pn + pi) * 16 + 4 * q) = acc;
            tr[t] = prelu4u(acc, a0);
        }
        f32x4 qv[2][2];
#pragma unroll
        for (int wq = 0; wq < 2; ++wq)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 acc = *(const f32x4*)(lbias + (3 + 2 * wq + t) * 16 + 4 * q);
                acc = mma_block(acc, lw[GA_Q(wq, t, 0) * 64 + lane], tr[0]);
                acc = mma_block(acc, lw[GA_Q(wq, t, 1) * 64 + lane], tr[1]);
                if (a.save != nullptr && valid) *(f32x4*)(a.save + ((size_t)(AV_Q + 2 * wq + t) * a.Pn + pi) * 16 + 4 * q) = acc;
                qv[wq][t] = prelu4u(acc, wq == 0 ? a11 : a12);
            }
        if (valid) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                *(f32x4*)(a.tr + pi * 32 + 16 * t + 4 * q) = tr[t];
                *(f32x4*)(a.q1 + pi * 32 + 16 * t + 4 * q) = qv[0][t];
                *(f32x4*)(a.q2 + pi * 32 + 16 * t + 4 * q) = qv[1][t];
            }
        }
    }
}

template <bool PCSR, int WPB = 4>      // WPB: waves per workgroup (they share one copy of the weight image: more of them fit a CU)
__global__ __launch_bounds__(WPB * 64) void k_assoc_b(AsArgs a) {
    constexpr int NF4 = (GB_GROUPS * 256 + GB_BIAS * 16 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    __shared__ __attribute__((aligned(16))) float tsc[WPB * 16 * 68];
    for (int i = threadIdx.x; i < NF4; i += WPB * 64) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lbias = (const float*)(lw + GB_GROUPS * 64);
    const float* lscal = lbias + GB_BIAS * 16;
    const float a1 = lscal[0], a21 = lscal[1], a22 = lscal[2];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the 23 gathered 128-B rows of a node are read in the row layout lane = 4 r + cq (four consecutive lanes: one 64-B half of a
    // row); the four neighbour means then cross a per-wave LDS scratch into the MFMA layout (see k_stage2_ord, RL)
    const int jl = lane >> 2, ql = lane & 3;
    float* ts = tsc + wave * 16 * 68;
    const int S = a.S;
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave);
    PtileIter ptw(PCSR ? (a.Pn + 15) / 16 : 0, blockDim.x >> 6, wave);      // PCSR: positions in the processing order of the tiles
    const long long n_items = PCSR ? ptw.end : w.nitems;
    const long long it0 = PCSR ? ptw.i : w.it;
    const long long its = PCSR ? ptw.stride : w.stride;
    for (long long it = it0; it < n_items; it += its) {
        int gi = 0, tb = 0, g, su;
        bool valid;
        long long pi, pu, pl = 0;            // pl (PCSR): the product node whose rows this lane gathers
        if (PCSR) {
            const long long pr = ptile_at(a.ptile, it) * 16 + j;
            valid = pr < a.Pn;
            pi = pu = valid ? pr : a.Pn - 1;
            g = a.src_of[pi];
            su = (int)pi;          // (ps, when present, is per product node on an irregular graph)
            pl = min(ptile_at(a.ptile, it) * 16 + jl, a.Pn - 1);
        } else {
            w.decode(it, gi, tb);
            g = __builtin_amdgcn_readfirstlane(a.order[gi]);
            const int s = tb * 16 + j;
            valid = s < S;
            const int sc = valid ? s : S - 1;
            su = a.sta_user != nullptr ? a.sta_user[sc] : sc;
            pi = (long long)g * S + sc; pu = (long long)g * S + su;
        }
        asm volatile("" : "+v"(lane));
        const float* pg = a.pg + (long long)g * AS_PG;
        const float mq = a.mask[pu * 4 + q];
        const f32x4 x0 = *(const f32x4*)(a.tr + pi * 32 + 4 * q), x1 = *(const f32x4*)(a.tr + pi * 32 + 16 + 4 * q);
        // neighbour means of q1 (stations of the same source node) and q2 (same station, neighbouring source nodes), edge order
        f32x4 n1a = {0.f, 0.f, 0.f, 0.f}, n1b = n1a, n2a = n1a, n2b = n1a;
        const int s_l = tb * 16 + jl, scl = s_l < S ? s_l : S - 1;        // the node whose rows this lane gathers
        if (PCSR) {        // both neighbourhoods are lists of product-node ids, ragged: predicated chunks of four, edge order
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int32_t* rp = nb == 0 ? a.sta_rowptr : a.src_rowptr;
                const int32_t* cl = nb == 0 ? a.sta_col : a.src_col;
                const float* base = (nb == 0 ? a.q1 : a.q2) + 4 * ql;
                const int eb = rp[pl], ee = rp[pl + 1];
                f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = sa;
                for (int e = eb; __any(e < ee); e += 4) {
                    f32x4 ra[4], rb[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool ok = e + k < ee;
                        const float* r = base + (long long)cl[ok ? e + k : max(ee - 1, 0)] * 32;
                        ra[k] = *(const f32x4*)r; rb[k] = *(const f32x4*)(r + 16);
                        if (!ok) { ra[k] = f32x4{0.f, 0.f, 0.f, 0.f}; rb[k] = ra[k]; }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) { sa += ra[k]; sb += rb[k]; }
                }
                const float inv = 1.f / (float)max(ee - eb, 1);
                if (nb == 0) { n1a = sa * inv; n1b = sb * inv; } else { n2a = sa * inv; n2b = sb * inv; }
            }
        } else {
        {
            const int eb = a.sta_rowptr[scl], ee = a.sta_rowptr[scl + 1];
            const float* base = a.q1 + (long long)g * S * 32 + 4 * ql;
            for (int e = eb; __any(e < ee); e += 4) {
                f32x4 ra[4], rb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool ok = e + k < ee;
                    const float* r = base + (long long)a.sta_col[ok ? e + k : max(ee - 1, 0)] * 32;
                    ra[k] = *(const f32x4*)r; rb[k] = *(const f32x4*)(r + 16);
                    if (!ok) { ra[k] = f32x4{0.f, 0.f, 0.f, 0.f}; rb[k] = ra[k]; }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) { n1a += ra[k]; n1b += rb[k]; }
            }
            const float inv = 1.f / (float)max(ee - eb, 1);
            n1a *= inv; n1b *= inv;
        }
        {
            const int eb = __builtin_amdgcn_readfirstlane(a.src_rowptr[g]);
            const int ee = __builtin_amdgcn_readfirstlane(a.src_rowptr[g + 1]);
            const float* base = a.q2 + (long long)scl * 32 + 4 * ql;
            int e = eb;
            for (; e + 4 <= ee; e += 4) {
                f32x4 ra[4], rb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float* r = base + (long long)a.src_col[e + k] * S * 32;
                    ra[k] = *(const f32x4*)r; rb[k] = *(const f32x4*)(r + 16);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) { n2a += ra[k]; n2b += rb[k]; }
            }
            for (; e < ee; ++e) {
                const float* r = base + (long long)a.src_col[e] * S * 32;
                n2a += *(const f32x4*)r; n2b += *(const f32x4*)(r + 16);
            }
            const float inv = 1.f / (float)max(ee - eb, 1);
            n2a *= inv; n2b *= inv;
        }
        }
        *(f32x4*)(ts + jl * 68 + 4 * ql) = n1a; *(f32x4*)(ts + jl * 68 + 16 + 4 * ql) = n1b;
        *(f32x4*)(ts + jl * 68 + 32 + 4 * ql) = n2a; *(f32x4*)(ts + jl * 68 + 48 + 4 * ql) = n2b;
        GSYNC();
        n1a = *(const f32x4*)(ts + j * 68 + 4 * q); n1b = *(const f32x4*)(ts + j * 68 + 16 + 4 * q);
        n2a = *(const f32x4*)(ts + j * 68 + 32 + 4 * q); n2b = *(const f32x4*)(ts + j * 68 + 48 + 4 * q);
        GSYNC();
        // layer 1
        f32x4 acc[4], w4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            acc[k] = *(const f32x4*)(lbias + k * 16 + 4 * q) + *(const f32x4*)(pg + 64 + 32 * (k >> 1) + 16 * (k & 1) + 4 * q);
        if (a.ps != nullptr) {
            acc[0] += *(const f32x4*)(a.ps + (long long)su * AS_PS + 32 + 4 * q);
            acc[1] += *(const f32x4*)(a.ps + (long long)su * AS_PS + 48 + 4 * q);
            if (PCSR) {
                acc[2] += *(const f32x4*)(a.ps + (long long)su * AS_PS + 80 + 4 * q);
                acc[3] += *(const f32x4*)(a.ps + (long long)su * AS_PS + 96 + 4 * q);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) w4[k] = lw[GB_L1(k >> 1, k & 1, 0) * 64 + lane];
        mma_blocks<4>(acc, w4, x0);
#pragma unroll
        for (int k = 0; k < 4; ++k) w4[k] = lw[GB_L1(k >> 1, k & 1, 1) * 64 + lane];
        mma_blocks<4>(acc, w4, x1);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x4 na = b == 0 ? n1a : n1b, nb = b == 0 ? n2a : n2b;
            f32x4 wa[2], wb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                wa[t] = lw[GB_L1(0, t, 2 + b) * 64 + lane];
                wb[t] = lw[GB_L1(1, t, 2 + b) * 64 + lane];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[0] = MFMA16(wa[0][r], na[r], acc[0]);
                acc[2] = MFMA16(wb[0][r], nb[r], acc[2]);
                acc[1] = MFMA16(wa[1][r], na[r], acc[1]);
                acc[3] = MFMA16(wb[1][r], nb[r], acc[3]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = MFMA16(lw[GB_L1(k >> 1, k & 1, 4) * 64 + lane].x, mq, acc[k]);
        if (a.save != nullptr && valid) {
#pragma unroll
            for (int k = 0; k < 4; ++k) *(f32x4*)(a.save + ((size_t)(AV_T + k) * a.Pn + pi) * 16 + 4 * q) = acc[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = prelu4u(acc[k], a1);
        // r1 / r2 and the node-local layer-2 terms
        f32x4 o6[6], w6[6];
#pragma unroll
        for (int k = 0; k < 4; ++k) o6[k] = *(const f32x4*)(lbias + (4 + k) * 16 + 4 * q);
        o6[4] = *(const f32x4*)(lbias + 8 * 16 + 4 * q) + *(const f32x4*)(pg + 128 + 4 * q);
        o6[5] = *(const f32x4*)(lbias + 9 * 16 + 4 * q) + *(const f32x4*)(pg + 144 + 4 * q);
        if (a.ps != nullptr) {
            o6[4] += *(const f32x4*)(a.ps + (long long)su * AS_PS + 64 + 4 * q);
            if (PCSR) o6[5] += *(const f32x4*)(a.ps + (long long)su * AS_PS + 112 + 4 * q);
        }
#pragma unroll
        for (int hb = 0; hb < 4; ++hb) {
#pragma unroll
            for (int k = 0; k < 4; ++k) w6[k] = lw[GB_UV(k >> 1, k & 1, hb) * 64 + lane];
            w6[4] = lw[GB_C(0, hb) * 64 + lane];
            w6[5] = lw[GB_C(1, hb) * 64 + lane];
            mma_blocks<6>(o6, w6, acc[hb]);
        }
        o6[4] = MFMA16(lw[GB_C(0, 4) * 64 + lane].x, mq, o6[4]);
        o6[5] = MFMA16(lw[GB_C(1, 4) * 64 + lane].x, mq, o6[5]);
        if (a.save != nullptr && valid) {
#pragma unroll
            for (int k = 0; k < 4; ++k) *(f32x4*)(a.save + ((size_t)(AV_UV + k) * a.Pn + pi) * 16 + 4 * q) = o6[k];
        }
        o6[0] = prelu4u(o6[0], a21); o6[1] = prelu4u(o6[1], a21);
        o6[2] = prelu4u(o6[2], a22); o6[3] = prelu4u(o6[3], a22);
        f32x4 wuv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, w2[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            w2[0] = lw[GB_W(0, b) * 64 + lane];
            w2[1] = lw[GB_W(1, b) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wuv[0] = MFMA16(w2[0][r], o6[b][r], wuv[0]);
                wuv[1] = MFMA16(w2[1][r], o6[2 + b][r], wuv[1]);
            }
        }
        if (valid) {
            *(f32x4*)(a.c + pi * ROWC + 4 * q) = o6[4];
            *(f32x4*)(a.c + pi * ROWC + 16 + 4 * q) = o6[5];
            *(f32x4*)(a.wu + pi * ROWW + 4 * q) = wuv[0];
            *(f32x4*)(a.wv + pi * ROWW + 4 * q) = wuv[1];
        }
    }
}
