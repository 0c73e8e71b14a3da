// train_assoc_kernels.hpp -- backward of the P-sized association heads of the 4-output training step (train_GENIE_model.py:1786-1861):
// BipartiteGraphReadOutOperator (module.py:343-352) + DataAggregationAssociationPhase (:389-403). Included by genie_hip.hip.
//
// Forward of a training step = the inference kernels k_assoc_pre / k_assoc_a / k_assoc_b + the generic stage-2 kernel without its
// Bipartite half, in the caller's station order, with their pre-activations kept (AV_* blocks of 16 floats per product node).
// Backward = the passes of DataAggregation's backward with this head's shapes:
//   k_as_b3   d s (the head's output [P, 30]) -> do = d s PReLU2'(o)                                                    store do
//   k_train_b1<true>  transposed means of do1 / do2 -> d r1, d r2 -> d tr1 -> dt = d tr1 PReLU1'(t)                    store dt, dtr_local
//   k_as_b1   transposed means of dt1 / dt2 -> d q1, d q2 (through l1_t1_1 / l1_t2_1) -> d tr -> d(init_trns pre-activation)   store dtrp
//   k_as_b0   init_trns -> d s_in -> BipartiteGraphReadOutOperator fc2 / PReLU / mask gate / fc1: weight gradients, and the
//             per-tile station sums of d z1 (fc1's pre-activation gradient)
//   k_as_g    per source node: d y_latent[g] = fc1[:, 0:30]^T sum_s d z1, and fc1's y_latent columns (sum_s d z1 (x) y_latent[g])
// x_latent enters init_trns detached (module.py:990): no gradient flows into DataAggregation from here.

// group index maps of the transposed plans
//  k_as_b1 (PL_TAB1): l1_t?_2[:, 30:60]^T (half h, out block b of d q, in block k of the transposed mean of dt),
//                     l1_t?_1^T (w, out block b of d tr, in block k of d q-pre)
#define GA1_Q(h, b, k) ((h) * 4 + (b) * 2 + (k))
#define GA1_L(w, b, k) (8 + (w) * 4 + (b) * 2 + (k))
#define GA1_GROUPS 16
//  k_as_b0 (PL_TAB0): init_trns[:, 0:15]^T (in block t of dtrp), fc2^T (out block b of d msg)
#define GA0_I(t) (t)
#define GA0_F2(b) (2 + (b))
#define GA0_GROUPS 4
//  k_as_g (PL_TAG): fc1[:, 0:30]^T (out block b of d y_latent, in block t of the d z1 sums)
#define GAG(b, t) ((b) * 2 + (t))
#define GAG_GROUPS 4

constexpr int GR_DTRP = 0;      // k_as_b1 stores d(init_trns pre-activation) over the (consumed) do blocks

__device__ __forceinline__ f32x4 ld30_half(const float* __restrict__ row, int t, int q) {      // channels 15 t + 4 q + {0..3} (< 15) of a 30-float row
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const float* r = row + 15 * t + 4 * q;
    v.x = r[0]; v.y = r[1]; v.z = r[2];
    if (q < 3) v.w = r[3];
    return v;
}

// ---- d s -> do
template <bool PCSR, int WPB = 4>      // WPB: waves per workgroup (as k_train_b2: a streaming pass, more resident waves hide its latency)
__global__ __launch_bounds__(WPB * 64) void k_as_b3(TrArgs a, const float* __restrict__ ds, const float* __restrict__ slope) {
    const float a2 = *slope;
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = a.S;
    const long long P = a.P;
    float scal[1] = {0.f};
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave);
    PtileIter ptw(PCSR ? (P + 15) / 16 : 0, WPB, wave);      // PCSR: positions in the processing order of the tiles
    const long long n_it = PCSR ? ptw.end : w.nitems, it0 = PCSR ? ptw.i : w.it, its = PCSR ? ptw.stride : w.stride;
    for (long long it = it0; it < n_it; it += its) {
        bool valid;
        long long p;
        if (PCSR) {
            const long long pr = ptile_at(a.ptile, it) * 16 + j;
            valid = pr < P;
            p = valid ? pr : P - 1;
        } else {
            int gi, tb;
            w.decode(it, gi, tb);
            const int g = __builtin_amdgcn_readfirstlane(a.order[gi]);
            const int s = tb * 16 + j;
            valid = s < S;
            p = (long long)g * S + (valid ? s : S - 1);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 o = ldb(a.save, AV_O + t, P, p, q);
            f32x4 d = ld30_half(ds + p * 30, t, q);
            if (!valid) d = f32x4{0.f, 0.f, 0.f, 0.f};
            scal[0] += negsum4(d, o);
            if (valid) stb(a.gr, GR_DO + t, P, p, q, d * dprelu4(o, a2));
        }
    }
    write_partials(a, blockIdx.x * WPB + wave, nullptr, 0, nullptr, 0, scal, 1, lane, j, q);
}

// ---- layer 1, l1_t1_1 / l1_t2_1 and the activation of init_trns
// accumulators: l1_t1_2 {2 x (tr x2, Mask), adjoint 2 x 2} = 10, l1_t2_2 = 10, l1_t1_1 (2 x 2) = 4, l1_t2_1 = 4  -> 28
// vec: b(l1_t1_2) x2, b(l1_t2_2) x2, mask1 column of l1_t1_2 x2, of l1_t2_2 x2, b(l1_t1_1) x2, b(l1_t2_1) x2 = 12; scal: a, a11, a12
template <bool PCSR, bool O32 = false>      // O32: rows addressed by 32-bit offsets on scalar bases (ldo / sto, train_front_kernels.hpp; 20 x P x 64 B < 4 GiB)
__global__ __launch_bounds__(256, 1) void k_as_b1(TrArgs a) {
    static_assert(!(PCSR && O32), "32-bit row offsets: Cartesian product graphs only");
    constexpr int NF4 = (GA1_GROUPS * 256 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    __shared__ __attribute__((aligned(16))) float tsc[4][16 * 17];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lscal = (const float*)(lw + GA1_GROUPS * 64);
    const float a0 = lscal[0], a11 = lscal[1], a12 = lscal[2];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sc = tsc[wave];
    const int S = a.S;
    const long long P = a.P;
    f32x4 acc[28], vec[12];
    float scal[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 12; ++k) vec[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned q16 = 16u * (unsigned)q, P64 = (unsigned)P * 64u;
    const unsigned long long grb = sbase(a.gr), svb = sbase(a.save);
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave);
    PtileIter ptw(PCSR ? (P + 15) / 16 : 0, 4, wave);      // PCSR: positions in the processing order of the tiles
    const long long n_it = PCSR ? ptw.end : w.nitems, it0 = PCSR ? ptw.i : w.it, its = PCSR ? ptw.stride : w.stride;
    struct Tile { int g, scn; bool valid; long long p; };
    struct Own { f32x4 mb, zt[2], qp[2][2], dt[4], dh0[2]; float m1; };       // the node's own rows
    auto tile_of = [&](long long it) {
        Tile t;
        if (PCSR) {       // 16 consecutive product nodes of an irregular product graph; the source node is per lane
            const long long pr = ptile_at(a.ptile, it) * 16 + j;
            t.valid = pr < P;
            t.p = t.valid ? pr : P - 1;
            t.g = a.src_of[t.p];
            t.scn = 0;
        } else {
            int gi, tb;
            w.decode(it < n_it ? it : it0, gi, tb);              // past the end: a tile of this wave again (loads nobody uses)
            t.g = __builtin_amdgcn_readfirstlane(a.order[gi]);
            const int s_ = tb * 16 + j;
            t.valid = s_ < S;
            t.scn = t.valid ? s_ : S - 1;
            t.p = (long long)t.g * S + t.scn;
        }
        return t;
    };
    auto own_load = [&](const Tile& t, Own& o) {
        const long long p = t.p;
        o.m1 = a.pg[(long long)t.g * AS_PG + 31];
        o.mb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (q == 0) o.mb = *(const f32x4*)(a.mask + p * 4);
        const unsigned pofs = (unsigned)p * 64u + q16;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            o.zt[b] = O32 ? ldo(svb, (unsigned)(AV_TR + b) * P64 + pofs) : ldb(a.save, AV_TR + b, P, p, q);
            o.qp[0][b] = O32 ? ldo(svb, (unsigned)(AV_Q + b) * P64 + pofs) : ldb(a.save, AV_Q + b, P, p, q);
            o.qp[1][b] = O32 ? ldo(svb, (unsigned)(AV_Q + 2 + b) * P64 + pofs) : ldb(a.save, AV_Q + 2 + b, P, p, q);
            o.dh0[b] = O32 ? ldo(grb, (unsigned)(GR_DH0 + b) * P64 + pofs) : ldb(a.gr, GR_DH0 + b, P, p, q);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) o.dt[k] = O32 ? ldo(grb, (unsigned)(GR_DT + k) * P64 + pofs) : ldb(a.gr, GR_DT + k, P, p, q);
    };
    // everything of a tile after its transposed means
    auto compute = [&](const Tile& t, const Own& o, f32x4 (&tmd1)[2], f32x4 (&tmd2)[2]) {
        const bool valid = t.valid;
        const long long p = t.p;
        const float vm = valid ? 1.f : 0.f, m1 = o.m1;
        const unsigned pofs = (unsigned)p * 64u + q16;
        f32x4 zt[2], tr[2], qp[2][2], dt[4];
#pragma unroll
        for (int b = 0; b < 2; ++b) { zt[b] = o.zt[b]; tr[b] = prelu4u(zt[b], a0); qp[0][b] = o.qp[0][b]; qp[1][b] = o.qp[1][b]; }
#pragma unroll
        for (int b = 0; b < 2; ++b) { tmd1[b] *= vm; tmd2[b] *= vm; }
#pragma unroll
        for (int k = 0; k < 4; ++k) dt[k] = o.dt[k] * vm;
        // d q = l1_t?_2[:, 30:60]^T (transposed mean of dt), through PReLU11' / PReLU12' -> d(l1_t?_1 output)
        f32x4 dqp[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x4 dq1 = {0.f, 0.f, 0.f, 0.f}, dq2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                dq1 = mma_block(dq1, lw[GA1_Q(0, b, k) * 64 + lane], tmd1[k]);
                dq2 = mma_block(dq2, lw[GA1_Q(1, b, k) * 64 + lane], tmd2[k]);
            }
            scal[1] += negsum4(dq1, qp[0][b]);
            scal[2] += negsum4(dq2, qp[1][b]);
            dqp[0][b] = dq1 * dprelu4(qp[0][b], a11);
            dqp[1][b] = dq2 * dprelu4(qp[1][b], a12);
        }
        // d tr = node-local part (pass before) + l1_t1_1^T d q1-pre + l1_t2_1^T d q2-pre; through the activation of init_trns
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x4 d = o.dh0[b] * vm;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                d = mma_block(d, lw[GA1_L(0, b, k) * 64 + lane], dqp[0][k]);
                d = mma_block(d, lw[GA1_L(1, b, k) * 64 + lane], dqp[1][k]);
            }
            scal[0] += negsum4(d, zt[b]);
            const f32x4 dz = d * dprelu4(zt[b], a0);
            if (valid) { if (O32) sto(grb, (unsigned)(GR_DTRP + b) * P64 + pofs, dz); else stb(a.gr, GR_DTRP + b, P, p, q, dz); }
        }
        vec[0] += dt[0]; vec[1] += dt[1]; vec[2] += dt[2]; vec[3] += dt[3];
        vec[4] += dt[0] * m1; vec[5] += dt[1] * m1; vec[6] += dt[2] * m1; vec[7] += dt[3] * m1;
        vec[8] += dqp[0][0]; vec[9] += dqp[0][1]; vec[10] += dqp[1][0]; vec[11] += dqp[1][1];
        // weight gradients
        const f32x4 mt = tr16(o.mb, sc, j, q);
        const f32x4 trt[2] = {tr16(tr[0], sc, j, q), tr16(tr[1], sc, j, q)};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int base = 10 * h;
            const f32x4 qt[2] = {tr16(prelu4u(qp[h][0], h == 0 ? a11 : a12), sc, j, q), tr16(prelu4u(qp[h][1], h == 0 ? a11 : a12), sc, j, q)};
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const f32x4 dtt = tr16(dt[2 * h + b], sc, j, q);
                acc[base + b * 3 + 0] = outer16(acc[base + b * 3 + 0], dtt, trt[0]);
                acc[base + b * 3 + 1] = outer16(acc[base + b * 3 + 1], dtt, trt[1]);
                acc[base + b * 3 + 2] = outer16(acc[base + b * 3 + 2], dtt, mt);
                const f32x4 tmt = tr16(h == 0 ? tmd1[b] : tmd2[b], sc, j, q);
                acc[base + 6 + b * 2 + 0] = outer16(acc[base + 6 + b * 2 + 0], tmt, qt[0]);
                acc[base + 6 + b * 2 + 1] = outer16(acc[base + 6 + b * 2 + 1], tmt, qt[1]);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const f32x4 dqt = tr16(dqp[h][b], sc, j, q);
                acc[20 + 4 * h + b * 2 + 0] = outer16(acc[20 + 4 * h + b * 2 + 0], dqt, trt[0]);
                acc[20 + 4 * h + b * 2 + 1] = outer16(acc[20 + 4 * h + b * 2 + 1], dqt, trt[1]);
            }
        }
    };
    if (O32) {
        // Software pipeline over the tiles of a wave (one wave per SIMD, as k_train_b1): while tile i computes, the row pointers of tile
        // i + 2 and the (column, weight) pairs and own rows of tile i + 1 are in flight; a tile waits for ONE round trip (its gathered
        // rows: 48 of them, too many registers to hold a tile ahead) where it waited for three in a row.
        constexpr int EBS = 8, EBG = 16;
        struct Rp { int s0, s1, g0, g1; };
        auto rp_of = [&](const Tile& t) {
            return Rp{a.r_sta_rowptr[t.scn], a.r_sta_rowptr[t.scn + 1], __builtin_amdgcn_readfirstlane(a.r_src_rowptr[t.g]),
                      __builtin_amdgcn_readfirstlane(a.r_src_rowptr[t.g + 1])};
        };
        const unsigned S64 = (unsigned)S * 64u;
        Tile cur = {0, 0, false, 0}, nxt = cur;
        Rp rp_n = {0, 0, 0, 0};
        Own own;
        NbrIdx<EBS> xs;
        NbrIdx<EBG> xg;
        if (it0 < n_it) {
            cur = tile_of(it0);
            const Rp rp = rp_of(cur);
            nbr_idx_load<EBS>(a.r_sta_cw, rp.s0, rp.s1, xs);
            nbr_idx_load<EBG>(a.r_src_cw, rp.g0, rp.g1, xg);
            own_load(cur, own);
            nxt = tile_of(it0 + its);
            rp_n = rp_of(nxt);
        }
        for (long long it = it0; it < n_it; it += its) {
            asm volatile("" : "+v"(lane));
            const Tile nn = tile_of(it + 2 * its);
            const Rp rp_nn = rp_of(nn);
            NbrIdx<EBS> xs_n;
            NbrIdx<EBG> xg_n;
            nbr_idx_load<EBS>(a.r_sta_cw, rp_n.s0, rp_n.s1, xs_n);
            nbr_idx_load<EBG>(a.r_src_cw, rp_n.g0, rp_n.g1, xg_n);
            Own own_n;
            own_load(nxt, own_n);
            // this tile's transposed means: rows of its first 8 / 16 out-edges in batches of four, then the rest of a long list; edge order kept
            const unsigned gs64 = (unsigned)(cur.g * S) * 64u;
            const unsigned vs0 = (unsigned)(GR_DT + 0) * P64 + gs64 + q16, vs1 = (unsigned)(GR_DT + 1) * P64 + gs64 + q16;
            const unsigned vg0 = (unsigned)(GR_DT + 2) * P64 + (unsigned)cur.scn * 64u + q16, vg1 = (unsigned)(GR_DT + 3) * P64 + (unsigned)cur.scn * 64u + q16;
            auto rows_s = [&](int b, int c) { return ldo(grb, (b == 0 ? vs0 : vs1) + ((unsigned)c << 6)); };
            auto rows_g = [&](int b, int c) { return ldo(grb, __umul24((unsigned)c, S64) + (b == 0 ? vg0 : vg1)); };
            f32x4 tmd1[2], tmd2[2];
            tmean_idx<2, EBS, 4>(xs, false, rows_s, tmd1);
            tmean_rest<2, 4>(a.r_sta_cw, xs.e_next, xs.e_end, false, rows_s, tmd1);
            tmean_idx<2, EBG, 4>(xg, true, rows_g, tmd2);
            tmean_rest<2, 4>(a.r_src_cw, xg.e_next, xg.e_end, true, rows_g, tmd2);
            compute(cur, own, tmd1, tmd2);
            cur = nxt; nxt = nn; rp_n = rp_nn; own = own_n; xs = xs_n; xg = xg_n;
        }
    } else {
        for (long long it = it0; it < n_it; it += its) {
            const Tile t = tile_of(it);
            asm volatile("" : "+v"(lane));
            Own own;
            own_load(t, own);                       // (own rows requested before the gathers)
            const float* gr = a.gr;
            const int g = t.g, scn = t.scn;
            const long long p = t.p;
            f32x4 tmd1[2], tmd2[2];
            if (PCSR) {      // reversed PRODUCT-level graphs, rows by product-node id
                tmean_pre<2, 8, 4>(a.r_sta_rowptr, a.r_sta_cw, (int)p, false, [&](int b, int c) { return ldb(gr, GR_DT + b, P, c, q); }, tmd1);
                tmean_pre<2, 16, 4>(a.r_src_rowptr, a.r_src_cw, (int)p, false, [&](int b, int c) { return ldb(gr, GR_DT + 2 + b, P, c, q); }, tmd2);
            } else {
                tmean_pre<2, 8, 4>(a.r_sta_rowptr, a.r_sta_cw, scn, false,
                                   [&](int b, int c) { return ldb(gr, GR_DT + b, P, (long long)g * S + c, q); }, tmd1);
                tmean_pre<2, 16, 4>(a.r_src_rowptr, a.r_src_cw, g, true,
                                    [&](int b, int c) { return ldb(gr, GR_DT + 2 + b, P, (long long)c * S + scn, q); }, tmd2);
            }
            compute(t, own, tmd1, tmd2);
        }
    }
    write_partials(a, blockIdx.x * 4 + wave, acc, 28, vec, 12, scal, 3, threadIdx.x & 63, j, q);
}

// ---- init_trns and BipartiteGraphReadOutOperator
// accumulators: init_trns (tile t) x {s, x_latent 0:16, x_latent 16:30, Mask} = 8; fc2 (msg blocks) = 2; fc1 edge_attr columns (t) = 2 -> 12
// vec: b(init_trns) x2, its mask1 column x2, b(fc2), b(fc1) x2 = 7; scal: activate1, activate2 of the read-out operator
template <bool PCSR, int WPB = 4>
__global__ __launch_bounds__(WPB * 64, 1) void k_as_b0(TrArgs a) {
    constexpr int NF4 = (GA0_GROUPS * 256 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    __shared__ __attribute__((aligned(16))) float tsc[WPB][16 * 17];
    for (int i = threadIdx.x; i < NF4; i += WPB * 64) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lscal = (const float*)(lw + GA0_GROUPS * 64);
    const float r1 = lscal[0], r2 = lscal[1];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sc = tsc[wave];
    const int S = a.S;
    const long long P = a.P;
    f32x4 acc[12], vec[7];
    float scal[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 7; ++k) vec[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave);
    PtileIter ptw(PCSR ? (P + 15) / 16 : 0, WPB, wave);      // PCSR: positions in the processing order of the tiles
    const long long n_it = PCSR ? ptw.end : w.nitems, it0 = PCSR ? ptw.i : w.it, its = PCSR ? ptw.stride : w.stride;
    for (long long it = it0; it < n_it; it += its) {
        int g, scn = 0, tb = 0;
        bool valid;
        long long p;
        if (PCSR) {       // 16 consecutive product nodes of an irregular product graph; the source node is per lane
            const long long pr = ptile_at(a.ptile, it) * 16 + j;
            valid = pr < P;
            p = valid ? pr : P - 1;
            g = a.src_of[p];
        } else {
            int gi;
            w.decode(it, gi, tb);
            g = __builtin_amdgcn_readfirstlane(a.order[gi]);
            const int s = tb * 16 + j;
            valid = s < S;
            scn = valid ? s : S - 1;
            p = (long long)g * S + scn;
        }
        asm volatile("" : "+v"(lane));
        const float vm = valid ? 1.f : 0.f;
        const float m1 = a.pg[(long long)g * AS_PG + 31];
        f32x4 mb = {0.f, 0.f, 0.f, 0.f}, eb = {0.f, 0.f, 0.f, 0.f};
        if (q == 0) {
            mb = *(const f32x4*)(a.mask + p * 4);
            eb.x = a.edge_attr[p * 3]; eb.y = a.edge_attr[p * 3 + 1]; eb.z = a.edge_attr[p * 3 + 2];
        }
        const f32x4 xl0 = ld_row30(a.x_latent + p * 30, 0, q), xl1 = ld_row30(a.x_latent + p * 30, 1, q);
        const f32x4 dz[2] = {ldb(a.gr, GR_DTRP + 0, P, p, q) * vm, ldb(a.gr, GR_DTRP + 1, P, p, q) * vm};
        const f32x4 svp = ldb(a.save, AV_SV, P, p, q), sv = prelu4u(svp, r2);
        const f32x4 z1[2] = {ldb(a.save, AV_Z1 + 0, P, p, q), ldb(a.save, AV_Z1 + 1, P, p, q)};
        // d s_in = init_trns[:, 0:15]^T dz, through PReLU_r2'
        f32x4 dsv = mma_block(f32x4{0.f, 0.f, 0.f, 0.f}, lw[GA0_I(0) * 64 + lane], dz[0]);
        dsv = mma_block(dsv, lw[GA0_I(1) * 64 + lane], dz[1]);
        scal[1] += negsum4(dsv, svp);
        const f32x4 dsp = dsv * dprelu4(svp, r2);
        // d msg = fc2^T d s-pre; msg = mask1 PReLU_r1(z1)
        f32x4 dz1[2], msg[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x4 dm = mma_block(f32x4{0.f, 0.f, 0.f, 0.f}, lw[GA0_F2(b) * 64 + lane], dsp) * m1;
            scal[0] += negsum4(dm, z1[b]);
            dz1[b] = dm * dprelu4(z1[b], r1);
            msg[b] = prelu4u(z1[b], r1) * m1;
        }
        // station sums of d z1 of this tile (-> d y_latent[g] and fc1's y_latent columns, k_as_g)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (PCSR) {      // a tile may straddle source nodes: the rows go to the (spent) dt blocks and are summed per source node after the pass
                if (valid) stb(a.gr, GR_DT + b, P, p, q, dz1[b]);
                continue;
            }
            f32x4 v = dz1[b];
            v.x = row_sum16(v.x); v.y = row_sum16(v.y); v.z = row_sum16(v.z); v.w = row_sum16(v.w);
            if (j == 0) *(f32x4*)(a.zsum + ((long long)g * a.T + tb) * 32 + 16 * b + 4 * q) = v;
        }
        vec[0] += dz[0]; vec[1] += dz[1];
        vec[2] += dz[0] * m1; vec[3] += dz[1] * m1;
        vec[4] += dsp;
        vec[5] += dz1[0]; vec[6] += dz1[1];
        const f32x4 svt = tr16(sv, sc, j, q), x0t = tr16(xl0, sc, j, q), x1t = tr16(xl1, sc, j, q), mt = tr16(mb, sc, j, q), et = tr16(eb, sc, j, q);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 dzt = tr16(dz[t], sc, j, q);
            acc[t * 4 + 0] = outer16(acc[t * 4 + 0], dzt, svt);
            acc[t * 4 + 1] = outer16(acc[t * 4 + 1], dzt, x0t);
            acc[t * 4 + 2] = outer16(acc[t * 4 + 2], dzt, x1t);
            acc[t * 4 + 3] = outer16(acc[t * 4 + 3], dzt, mt);
            acc[10 + t] = outer16(acc[10 + t], tr16(dz1[t], sc, j, q), et);
        }
        const f32x4 dspt = tr16(dsp, sc, j, q);
        acc[8] = outer16(acc[8], dspt, tr16(msg[0], sc, j, q));
        acc[9] = outer16(acc[9], dspt, tr16(msg[1], sc, j, q));
    }
    write_partials(a, blockIdx.x * WPB + wave, acc, 12, vec, 7, scal, 2, threadIdx.x & 63, j, q);
}

// irregular product graph: out[g][16 b + c] = sum over the product nodes of source node g of block b's rows ([2][P][16], row order)
__global__ void k_seg_sum_blocks(const float* __restrict__ blk, long long P, const int32_t* __restrict__ seg, int G, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= G * 32) return;
    const int g = idx >> 5, c = idx & 31, b = c >> 4;
    float s = 0.f;
    for (long long pr = seg[g]; pr < seg[g + 1]; ++pr) s += blk[((size_t)b * P + pr) * 16 + (c & 15)];
    out[idx] = s;
}

// ---- per source node: d y_latent[g] = fc1[:, 0:30]^T sum_s d z1[g, s]; fc1[:, 0:30] += (sum_s d z1) (x) y_latent[g]
struct AgArgs {
    int G, T;
    const float* zsum;           // [G * T][32]
    const float* y_latent;       // [G][30]
    const float* timg;           // PL_TAG
    float* d_ylat;               // [G][30]
    float* part; int n_acc, n_vec;
};
__global__ __launch_bounds__(256, 1) void k_as_g(AgArgs a) {
    __shared__ __attribute__((aligned(16))) float sm[GAG_GROUPS * 256 + 16 + 4 * 16 * 17];
    for (int i = threadIdx.x; i < (GAG_GROUPS * 256 + 16) / 4; i += blockDim.x) ((f32x4*)sm)[i] = ((const f32x4*)a.timg)[i];
    const f32x4* tw = (const f32x4*)sm;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    float* trs = sm + GAG_GROUPS * 256 + 16 + wave * 16 * 17;
    const TpSlot ps = tp_open(a.part, a.n_acc, a.n_vec, blockIdx.x * 4 + wave, lane);
    const int ntiles = (a.G + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int g = tile * 16 + j;
        const bool ok = g < a.G;
        const int gc = ok ? g : a.G - 1;
        f32x4 z[2] = {tl_zero(), tl_zero()};
        const float* zp = a.zsum + (long long)gc * a.T * 32 + 4 * q;
        for (int tb = 0; tb < a.T; ++tb) { z[0] += *(const f32x4*)(zp + tb * 32); z[1] += *(const f32x4*)(zp + tb * 32 + 16); }
        z[0] = mask4(z[0], ok); z[1] = mask4(z[1], ok);
        const float* row = a.y_latent + (long long)gc * 30;
        const f32x4 yb[2] = {tl_load30(row, 0, q), tl_load30(row, 1, q)};
        const f32x4 yt[2] = {tr16(yb[0], trs, j, q), tr16(yb[1], trs, j, q)};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 zt = tr16(z[t], trs, j, q);
            tp_acc(ps, t * 2 + 0, lane, outer16(tl_zero(), zt, yt[0]));
            tp_acc(ps, t * 2 + 1, lane, outer16(tl_zero(), zt, yt[1]));
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x4 d = mma_block(tl_zero(), tw[GAG(b, 0) * 64 + lane], z[0]);
            d = mma_block(d, tw[GAG(b, 1) * 64 + lane], z[1]);
            if (ok) tl_store30(a.d_ylat + (long long)g * 30, b, q, d);
        }
    }
}

// ---- LocalSliceLgCollapse (module.py:610-659), backward ---------------------------------------------------------------------
// Forward recomputed per 16-pick tile exactly as k_lslc does it (the s rows of the 10 time-pointer nodes as MFMA B blocks); the
// gradient w.r.t. every gathered s row goes to a per-edge row buffer (zero for an edge the 2-eps filter dropped) and is summed
// into d s [P, 30] by k_seg_rows over the edges sorted by product node: several picks may point at the same node, and the sum must
// not depend on scheduling.
//  transposed plan (PL_TLSP / PL_TLSS): fc2^T (out block b of the mean message), fc1[:, 0:30]^T (out block b of the row, in tile t)
#define GLT_F2(b) (b)
#define GLT_F1(b, t) (2 + (b) * 2 + (t))
#define GLT_GROUPS 6
struct LbArgs {
    LsArgs f;                    // the forward's arguments (img = the head's forward image)
    const float* timg;
    const float* d_out;          // [n_picks, 15]
    float* erow;                 // [n_picks * 10][32] gradient w.r.t. the s row of every edge
    int32_t* etgt;               // [n_picks * 10] its product node
    float* part; int n_acc, n_vec;
};
__global__ __launch_bounds__(256, 1) void k_lslc_bwd(LbArgs b) {
    __shared__ __attribute__((aligned(16))) float sm[GL_IMG_FLOATS + GLT_GROUPS * 256 + 16 + 4 * 16 * 17];
    const LsArgs& a = b.f;
    const TlImg im = tl_stage_image(sm, a.img, GL_GROUPS, GL_BIAS);
    float* tw_ = sm + GL_IMG_FLOATS;
    for (int i = threadIdx.x; i < (GLT_GROUPS * 256 + 16) / 4; i += blockDim.x) ((f32x4*)tw_)[i] = ((const f32x4*)b.timg)[i];
    const f32x4* tw = (const f32x4*)tw_;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    float* trs = tw_ + GLT_GROUPS * 256 + 16 + wave * 16 * 17;
    const float act1 = im.scal[0], act2 = im.scal[1];
    const TpSlot ps = tp_open(b.part, b.n_acc, b.n_vec, blockIdx.x * 4 + wave, lane);
    float s_a1 = 0.f, s_a2 = 0.f;
    const int ntiles = (a.n_picks + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int p = tile * 16 + j;
        const bool ok = p < a.n_picks;
        const int pc = ok ? p : a.n_picks - 1;
        const float tp = a.tpick[pc], ph = a.phase[pc];
        const int ti = (int)floorf((tp - (a.dtp ? a.dtp[0] : a.t0)) / (a.dtp ? a.dtp[1] - a.dtp[0] : a.dt));
        long long base = ((long long)a.ipick[pc] * a.l_dt + ti) * LS_K;
        base = base < 0 ? 0 : (base > a.n_edges - LS_K ? a.n_edges - LS_K : base);
        // ---- forward: mean message, output pre-activation
        f32x4 acc[2] = {tl_zero(), tl_zero()};
        float cnt = 0.f;
#pragma unroll 2
        for (int k = 0; k < LS_K; ++k) {
            const int e = a.A_edges[base + k];
            const float rt = tp - a.tlatent[(long long)e * a.tl_stride + a.tl_col];
            const bool keep = ok && fabsf(rt) < 2.0f * a.eps;
            const float* row = a.s + (long long)e * 30;
            const f32x4 xb0 = tl_load30(row, 0, q), xb1 = tl_load30(row, 1, q);
            const float xs = q == 0 ? rt / a.eps : (q == 1 ? ph : 0.f);
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
        const f32x4 ag[2] = {acc[0] / den, acc[1] / den};
        f32x4 o = mma_block(tl_bias(im, 2, q), TLW(im, GL_FC2(0)), ag[0]);
        o = mma_block(o, TLW(im, GL_FC2(1)), ag[1]);
        // ---- backward of fc2 / PReLU2
        f32x4 d = tl_zero();
        if (ok) d = tl_load15(b.d_out + (long long)p * 15, q);
        s_a2 += negsum4(d, o);
        const f32x4 dp2 = d * dprelu4(o, act2);
        tp_vec(ps, 2, j, q, dp2);
        {
            const f32x4 dt_ = tr16(dp2, trs, j, q);
            tp_acc(ps, 6, lane, outer16(tl_zero(), dt_, tr16(ag[0], trs, j, q)));
            tp_acc(ps, 7, lane, outer16(tl_zero(), dt_, tr16(ag[1], trs, j, q)));
        }
        const f32x4 dag[2] = {mma_block(tl_zero(), tw[GLT_F2(0) * 64 + lane], dp2) / den, mma_block(tl_zero(), tw[GLT_F2(1) * 64 + lane], dp2) / den};
        // ---- edges again: d message -> fc1 gradients and the gradient of every gathered row
        f32x4 wacc[6], bacc[2] = {tl_zero(), tl_zero()};
#pragma unroll
        for (int k = 0; k < 6; ++k) wacc[k] = tl_zero();
#pragma unroll 1
        for (int k = 0; k < LS_K; ++k) {
            const int e = a.A_edges[base + k];
            const float rt = tp - a.tlatent[(long long)e * a.tl_stride + a.tl_col];
            const bool keep = ok && fabsf(rt) < 2.0f * a.eps;
            const float* row = a.s + (long long)e * 30;
            const f32x4 xb0 = tl_load30(row, 0, q), xb1 = tl_load30(row, 1, q);
            const float xs = q == 0 ? rt / a.eps : (q == 1 ? ph : 0.f);
            const f32x4 xsb = q == 0 ? f32x4{rt / a.eps, ph, 0.f, 0.f} : tl_zero();      // the two scalar columns as channels 0, 1 of a block
            f32x4 dp1[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 m = mma_block(tl_bias(im, t, q), TLW(im, GL_FC1(t, 0)), xb0);
                m = mma_block(m, TLW(im, GL_FC1(t, 1)), xb1);
                m = MFMA16(TLW(im, GL_FC1(t, 2)).x, xs, m);
                const f32x4 dm = mask4(dag[t], keep);
                s_a1 += negsum4(dm, m);
                dp1[t] = dm * dprelu4(m, act1);
                bacc[t] += dp1[t];
            }
            const f32x4 x0t = tr16(xb0, trs, j, q), x1t = tr16(xb1, trs, j, q), xst = tr16(xsb, trs, j, q);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x4 dt_ = tr16(dp1[t], trs, j, q);
                wacc[t * 3 + 0] = outer16(wacc[t * 3 + 0], dt_, x0t);
                wacc[t * 3 + 1] = outer16(wacc[t * 3 + 1], dt_, x1t);
                wacc[t * 3 + 2] = outer16(wacc[t * 3 + 2], dt_, xst);
            }
            f32x4 dr[2];
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
                dr[bb] = mma_block(mma_block(tl_zero(), tw[GLT_F1(bb, 0) * 64 + lane], dp1[0]), tw[GLT_F1(bb, 1) * 64 + lane], dp1[1]);
            if (ok) {
                float* er = b.erow + ((long long)p * LS_K + k) * 32;
                *(f32x4*)(er + 4 * q) = dr[0];
                *(f32x4*)(er + 16 + 4 * q) = dr[1];
                if (q == 0) b.etgt[(long long)p * LS_K + k] = keep ? e : -1;
            }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) tp_acc(ps, k, lane, wacc[k]);
        tp_vec(ps, 0, j, q, bacc[0]);
        tp_vec(ps, 1, j, q, bacc[1]);
    }
    tp_scal(ps, 0, lane, s_a1);
    tp_scal(ps, 1, lane, s_a2);
}

// d s [P, 30] += per-edge rows, summed per product node in the order of `order` (edges sorted by target, stable): one group of 8
// lanes per segment start; targets < 0 (dropped edges) are skipped. `ds` must be zero where no edge points.
__global__ __launch_bounds__(256) void k_seg_rows(const float* __restrict__ erow, const int32_t* __restrict__ etgt, const int32_t* __restrict__ order,
                                                 long long n_edges, float* __restrict__ ds) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int c = threadIdx.x & 7;
    if (i >= n_edges) return;
    const int tgt = etgt[order[i]];
    if (tgt < 0 || (i > 0 && etgt[order[i - 1]] == tgt)) return;            // not a segment start
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (long long k = i; k < n_edges && etgt[order[k]] == tgt; ++k) s += *(const f32x4*)(erow + (long long)order[k] * 32 + 4 * c);
    float* o = ds + (long long)tgt * 30 + 4 * c;
    if (c < 7) { o[0] += s.x; o[1] += s.y; o[2] += s.z; o[3] += s.w; }
    else { o[0] += s.x; o[1] += s.y; }
}
