// Part of genie_hip.hip (one translation unit, included inside its anonymous namespace): the P-sized stage kernels of DataAggregation + Bipartite_ReadIn: generic fp32-MFMA stage 1 / stage 2 (any graph), the f16x2 stage 1 (k_stage1_h2) with its split / pack kernels, the straight-line stage 2 (k_stage2_ord).

// ------------------------------------------------------------------------------------------------
// stage 1: everything of DataAggregation that does not need the SECOND pair of neighbour means
//   h0 = PReLU(init_trns [X || M])                                        module.py:87-88 (own node + every neighbour)
//   h1 = PReLU1([l1_t1_2 [h0 || mean_sta PReLU11(h0) || M] || l1_t2_2 [h0 || mean_src PReLU12(h0) || M]])   :90-92
//   u = PReLU21(l2_t1_1 h1), v = PReLU22(l2_t2_1 h1)                      :94-95
//   wu = l2_t1_2[:, 60:90] u,  wv = l2_t2_2[:, 60:90] v                  (operands of the second pair of means)
//   c  = [l2_t1_2[:, 0:60] h1 + l2_t1_2[:, 90:94] M + b || l2_t2_2[...]]  (node-local part of :94-95)
// Reads 32 B per product node (+ its neighbours' 32-B rows from L2), writes 256 B; h0 / h1 / u / v stay in VGPRs.
// ------------------------------------------------------------------------------------------------
// dense tail of stage 1 for one tile: layer 1 from (x0,x1 = own h0; n1*, n2* = neighbour means), then u / v, the
// projected operands wu / wv and the node-local layer-2 terms c; stores c, wu, wv (and h0 / h1 for parity runs)
// N independent accumulators x one 16-channel input block, k-step OUTER and accumulator INNER: consecutive MFMAs never
// target the same accumulator, so the 40-cycle dependent-issue latency of v_mfma_f32_16x16x4_f32 (32-cycle issue) is
// always covered (hipcc otherwise keeps the 4 dependent k-steps of one accumulator back to back).
template <int N>
__device__ __forceinline__ void mma_blocks(f32x4 (&acc)[N], const f32x4 (&w)[N], const f32x4 x) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = MFMA16(w[k][r], x[r], acc[k]);
    }
}

// dense tail of stage 1 for one tile: layer 1 from (x0,x1 = own h0; n1*, n2* = neighbour means), then u / v, the
// projected operands wu / wv and the node-local layer-2 terms c; stores c, wu, wv (and h0 / h1 for parity runs)
__device__ __forceinline__ void stage1_dense(const DaArgs& a, const f32x4* lw, const float* lbias, int lane, int q,
                                             bool valid, long long p, int g, int sc, float mq, f32x4 x0, f32x4 x1, f32x4 n1a,
                                             f32x4 n1b, f32x4 n2a, f32x4 n2b, float a1, float a21, float a22) {
    if (a.dbg_h0 != nullptr && valid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a.dbg_h0[p * 30 + 4 * q + r] = x0[r];
            if (16 + 4 * q + r < 30) a.dbg_h0[p * 30 + 16 + 4 * q + r] = x1[r];
        }
    }
    // layer 1: tr1 = l1_t1_2 [h0 || n1 || M], tr2 = l1_t2_2 [h0 || n2 || M]; acc[k]: k = (half, tile)
    f32x4 acc[4], w4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = *(const f32x4*)(lbias + (2 + k) * 16 + 4 * q);
    if (a.eb_sta != nullptr) {   // DataAggregationEdges: the mean edge feature of a node is static, its Linear a per-node bias
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            acc[t] += *(const f32x4*)(a.eb_sta + (long long)sc * 48 + 16 * t + 4 * q);
            acc[2 + t] += *(const f32x4*)(a.eb_src + (long long)g * 48 + 16 * t + 4 * q);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) w4[k] = lw[G1_L1(k >> 1, k & 1, 0) * 64 + lane];
    mma_blocks<4>(acc, w4, x0);
#pragma unroll
    for (int k = 0; k < 4; ++k) w4[k] = lw[G1_L1(k >> 1, k & 1, 1) * 64 + lane];
    mma_blocks<4>(acc, w4, x1);
    {   // the neighbour-mean blocks feed only their own half: two accumulators per operand, interleave the two operands
        f32x4 wa[2], wb[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x4 na = b == 0 ? n1a : n1b, nb = b == 0 ? n2a : n2b;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                wa[t] = lw[G1_L1(0, t, 2 + b) * 64 + lane];
                wb[t] = lw[G1_L1(1, t, 2 + b) * 64 + lane];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[0] = MFMA16(wa[0][r], na[r], acc[0]);
                acc[2] = MFMA16(wb[0][r], nb[r], acc[2]);
                acc[1] = MFMA16(wa[1][r], na[r], acc[1]);
                acc[3] = MFMA16(wb[1][r], nb[r], acc[3]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = MFMA16(lw[G1_L1(k >> 1, k & 1, 4) * 64 + lane].x, mq, acc[k]);
    if (a.save != nullptr && valid) {
#pragma unroll
        for (int k = 0; k < 4; ++k) *(f32x4*)(a.save + ((size_t)(SV_T + k) * a.Pn + p) * 16 + 4 * q) = acc[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = prelu4u(acc[k], a1);                      // h1 block k = (half, tile)
    if (a.dbg_h1 != nullptr && valid) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * (k & 1) + 4 * q + r < 30) a.dbg_h1[p * 60 + 30 * (k >> 1) + 16 * (k & 1) + 4 * q + r] = acc[k][r];
    }
    // u = PReLU21(l2_t1_1 h1), v = PReLU22(l2_t2_1 h1); c_w = node-local layer-2 terms: 6 independent accumulators
    f32x4 o6[6], w6[6];
#pragma unroll
    for (int k = 0; k < 4; ++k) o6[k] = *(const f32x4*)(lbias + (6 + k) * 16 + 4 * q);
    o6[4] = *(const f32x4*)(lbias + 10 * 16 + 4 * q);
    o6[5] = *(const f32x4*)(lbias + 11 * 16 + 4 * q);
    if (a.eb_sta != nullptr) {
        o6[4] += *(const f32x4*)(a.eb_sta + (long long)sc * 48 + 32 + 4 * q);
        o6[5] += *(const f32x4*)(a.eb_src + (long long)g * 48 + 32 + 4 * q);
    }
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
#pragma unroll
        for (int k = 0; k < 4; ++k) w6[k] = lw[G1_UV(k >> 1, k & 1, hb) * 64 + lane];
        w6[4] = lw[G1_C(0, hb) * 64 + lane];
        w6[5] = lw[G1_C(1, hb) * 64 + lane];
        mma_blocks<6>(o6, w6, acc[hb]);
    }
    o6[4] = MFMA16(lw[G1_C(0, 4) * 64 + lane].x, mq, o6[4]);
    o6[5] = MFMA16(lw[G1_C(1, 4) * 64 + lane].x, mq, o6[5]);
    if (a.save != nullptr && valid) {
#pragma unroll
        for (int k = 0; k < 4; ++k) *(f32x4*)(a.save + ((size_t)(SV_UP + k) * a.Pn + p) * 16 + 4 * q) = o6[k];
    }
    o6[0] = prelu4u(o6[0], a21); o6[1] = prelu4u(o6[1], a21);
    o6[2] = prelu4u(o6[2], a22); o6[3] = prelu4u(o6[3], a22);
    // wu = l2_t1_2[:, 60:90] u, wv = l2_t2_2[:, 60:90] v: two accumulators, interleaved
    f32x4 wuv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, w2[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        w2[0] = lw[G1_W(0, b) * 64 + lane];
        w2[1] = lw[G1_W(1, b) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            wuv[0] = MFMA16(w2[0][r], o6[b][r], wuv[0]);
            wuv[1] = MFMA16(w2[1][r], o6[2 + b][r], wuv[1]);
        }
    }
    if (valid && !ABL(a, 3)) {
        *(f32x4*)(a.c + p * ROWC + 4 * q) = o6[4];
        *(f32x4*)(a.c + p * ROWC + 16 + 4 * q) = o6[5];
        *(f32x4*)(a.wu + p * ROWW + 4 * q) = wuv[0];
        *(f32x4*)(a.wv + p * ROWW + 4 * q) = wuv[1];
    }
}

// generic stage 1: any CSR graphs (ragged degrees, empty neighbourhoods)
__global__ __launch_bounds__(256) void k_stage1(DaArgs a) {
    constexpr int NF4 = (G1_GROUPS * 256 + G1_BIAS * 16 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lbias = (const float*)(lw + G1_GROUPS * 64);
    const float* lscal = lbias + G1_BIAS * 16;
    const float a0 = lscal[0], a1 = lscal[3], a21 = lscal[4], a22 = lscal[5];
    const float s11 = compose_slopes(a0, lscal[1]), s12 = compose_slopes(a0, lscal[2]);
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = a.S;
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave, a.gi0);
    for (; w.it < w.nitems; w.it += w.stride) {
        int gi, tb;
        w.decode(w.it, gi, tb);
        const int g = __builtin_amdgcn_readfirstlane(a.order[gi]);
        asm volatile("" : "+v"(lane));  // A fragments are re-read from LDS per tile, not held in VGPRs across tiles
        const int s = tb * 16 + j;
        const bool valid = s < S;
        const int sc = valid ? s : S - 1;
        const long long p = (long long)g * S + sc;
        const float xs = a.slice[p * 4 + q];
        const float mq = a.mask[p * 4 + q];
        const f32x4 wi0 = lw[G1_INIT(0) * 64 + lane], wi1 = lw[G1_INIT(1) * 64 + lane];
        const f32x4 bi0 = *(const f32x4*)(lbias + 0 * 16 + 4 * q), bi1 = *(const f32x4*)(lbias + 1 * 16 + 4 * q);
        // own hidden state
        f32x4 x0 = MFMA16(wi0.x, xs, bi0), x1 = MFMA16(wi1.x, xs, bi1);
        x0 = MFMA16(wi0.y, mq, x0);
        x1 = MFMA16(wi1.y, mq, x1);
        float lq = 0.f, gq = 0.f;                 // use_absolute_pos: this node's station / source position channel q
        if (a.abs_sta != nullptr) {
            lq = a.abs_sta[sc * 4 + q];
            gq = a.abs_src[g * 4 + q];
            x0 = MFMA16(wi0.z, lq, x0); x1 = MFMA16(wi1.z, lq, x1);
            x0 = MFMA16(wi0.w, gq, x0); x1 = MFMA16(wi1.w, gq, x1);
        }
        if (a.save != nullptr && valid) {
            *(f32x4*)(a.save + ((size_t)(SV_Z0 + 0) * a.Pn + p) * 16 + 4 * q) = x0;
            *(f32x4*)(a.save + ((size_t)(SV_Z0 + 1) * a.Pn + p) * 16 + 4 * q) = x1;
        }
        x0 = prelu4u(x0, a0);
        x1 = prelu4u(x1, a0);
        // station-neighbour mean of PReLU11(h0): rows of the same source node
        f32x4 n1a = {0.f, 0.f, 0.f, 0.f}, n1b = {0.f, 0.f, 0.f, 0.f};
        {
            const int eb = a.sta_rowptr[sc], ee = a.sta_rowptr[sc + 1];
            if (!ABL(a, 0)) {
                if (s11 <= 1.f)
                    gather_recompute<false, true>(a.slice, a.mask, (long long)g * S, 1, q, a.sta_col, eb, ee, wi0, wi1, bi0,
                                                  bi1, s11, n1a, n1b, AbsNbr{a.abs_sta, true, gq});
                else
                    gather_recompute<false, false>(a.slice, a.mask, (long long)g * S, 1, q, a.sta_col, eb, ee, wi0, wi1, bi0,
                                                   bi1, s11, n1a, n1b, AbsNbr{a.abs_sta, true, gq});
            }
            const float inv = 1.f / (float)max(ee - eb, 1);
            n1a *= inv; n1b *= inv;
        }
        // source-neighbour mean of PReLU12(h0): same station, neighbouring source nodes (wave-uniform list)
        f32x4 n2a = {0.f, 0.f, 0.f, 0.f}, n2b = {0.f, 0.f, 0.f, 0.f};
        {
            const int eb = __builtin_amdgcn_readfirstlane(a.src_rowptr[g]);
            const int ee = __builtin_amdgcn_readfirstlane(a.src_rowptr[g + 1]);
            if (!ABL(a, 1)) {
                if (s12 <= 1.f)
                    gather_recompute<true, true>(a.slice, a.mask, (long long)sc, (long long)S, q, a.src_col, eb, ee, wi0, wi1,
                                                 bi0, bi1, s12, n2a, n2b, AbsNbr{a.abs_src, false, lq});
                else
                    gather_recompute<true, false>(a.slice, a.mask, (long long)sc, (long long)S, q, a.src_col, eb, ee, wi0, wi1,
                                                  bi0, bi1, s12, n2a, n2b, AbsNbr{a.abs_src, false, lq});
            }
            const float inv = 1.f / (float)max(ee - eb, 1);
            n2a *= inv; n2b *= inv;
        }
        stage1_dense(a, lw, lbias, lane, q, valid, p, g, sc, mq, x0, x1, n1a, n1b, n2a, n2b, a1, a21, a22);
    }
}

// Stage 1 on an IRREGULAR product graph (`use_subgraph: True`, process_utils.py:744-849): the product nodes are an arbitrary
// list of (station, source) pairs and both edge sets are CSR lists over PRODUCT-node ids (a.sta_rowptr/col, a.src_rowptr/col
// are indexed by product node here). A tile is 16 consecutive product nodes; same arithmetic as k_stage1.
__global__ __launch_bounds__(256) void k_stage1_pcsr(DaArgs a) {
    constexpr int NF4 = (G1_GROUPS * 256 + G1_BIAS * 16 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lbias = (const float*)(lw + G1_GROUPS * 64);
    const float* lscal = lbias + G1_BIAS * 16;
    const float a0 = lscal[0], a1 = lscal[3], a21 = lscal[4], a22 = lscal[5];
    const float s11 = compose_slopes(a0, lscal[1]), s12 = compose_slopes(a0, lscal[2]);
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const long long ntiles = (a.Pn + 15) / 16;
    PtileIter pt(ntiles, blockDim.x >> 6, threadIdx.x >> 6);
    for (; pt.i < pt.end; pt.i += pt.stride) {
        const long long tile = ptile_at(a.ptile, pt.i);
        asm volatile("" : "+v"(lane));
        const long long pr = tile * 16 + j;
        const bool valid = pr < a.Pn;
        const long long p = valid ? pr : a.Pn - 1;
        const float xs = a.slice[p * 4 + q];
        const float mq = a.mask[p * 4 + q];
        const f32x4 wi0 = lw[G1_INIT(0) * 64 + lane], wi1 = lw[G1_INIT(1) * 64 + lane];
        const f32x4 bi0 = *(const f32x4*)(lbias + 0 * 16 + 4 * q), bi1 = *(const f32x4*)(lbias + 1 * 16 + 4 * q);
        f32x4 x0 = MFMA16(wi0.x, xs, bi0), x1 = MFMA16(wi1.x, xs, bi1);
        x0 = MFMA16(wi0.y, mq, x0);
        x1 = MFMA16(wi1.y, mq, x1);
        float lq = 0.f, gq = 0.f;     // use_absolute_pos: the position tables are per PRODUCT node here (genie_set_absolute_pos on a subgraph context)
        if (a.abs_sta != nullptr) {
            lq = a.abs_sta[p * 4 + q];
            gq = a.abs_src[p * 4 + q];
            x0 = MFMA16(wi0.z, lq, x0); x1 = MFMA16(wi1.z, lq, x1);
            x0 = MFMA16(wi0.w, gq, x0); x1 = MFMA16(wi1.w, gq, x1);
        }
        if (a.save != nullptr && valid) {      // training forward: the pre-activation of h0
            *(f32x4*)(a.save + ((size_t)(SV_Z0 + 0) * a.Pn + p) * 16 + 4 * q) = x0;
            *(f32x4*)(a.save + ((size_t)(SV_Z0 + 1) * a.Pn + p) * 16 + 4 * q) = x1;
        }
        x0 = prelu4u(x0, a0);
        x1 = prelu4u(x1, a0);
        f32x4 n1a = {0.f, 0.f, 0.f, 0.f}, n1b = n1a, n2a = n1a, n2b = n1a;
        {       // a station-type neighbour shares the source node: its source position is this node's; its station position is its own row
            const int eb = a.sta_rowptr[p], ee = a.sta_rowptr[p + 1];
            if (s11 <= 1.f) gather_recompute<false, true>(a.slice, a.mask, 0, 1, q, a.sta_col, eb, ee, wi0, wi1, bi0, bi1, s11, n1a, n1b, AbsNbr{a.abs_sta, true, gq});
            else gather_recompute<false, false>(a.slice, a.mask, 0, 1, q, a.sta_col, eb, ee, wi0, wi1, bi0, bi1, s11, n1a, n1b, AbsNbr{a.abs_sta, true, gq});
            const float inv = 1.f / (float)max(ee - eb, 1);
            n1a *= inv; n1b *= inv;
        }
        {
            const int eb = a.src_rowptr[p], ee = a.src_rowptr[p + 1];
            if (s12 <= 1.f) gather_recompute<false, true>(a.slice, a.mask, 0, 1, q, a.src_col, eb, ee, wi0, wi1, bi0, bi1, s12, n2a, n2b, AbsNbr{a.abs_src, false, lq});
            else gather_recompute<false, false>(a.slice, a.mask, 0, 1, q, a.src_col, eb, ee, wi0, wi1, bi0, bi1, s12, n2a, n2b, AbsNbr{a.abs_src, false, lq});
            const float inv = 1.f / (float)max(ee - eb, 1);
            n2a *= inv; n2b *= inv;
        }
        // (DataAggregationEdges on an irregular graph: the static-term tables are per product node, indexed by p on both sides)
        stage1_dense(a, lw, lbias, lane, q, valid, p, (int)p, (int)p, mq, x0, x1, n1a, n1b, n2a, n2b, a1, a21, a22);
    }
}

// the KS station-neighbour ids of one station as wide loads: a dword load whose lanes hit 16 different 32-B segments costs the
// texture path about as much as two and a half full 1-KB row loads (tools/: skeleton ablations of k_stage2_fast), and a tile
// issued KS of them
template <int KS>
__device__ __forceinline__ void load_sta_ids(const int32_t* __restrict__ sta_col, int sc, int (&sta)[KS]) {
    static_assert(KS % 4 == 0, "station-neighbour rows are read as 16-byte chunks");
    typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int c = 0; c < KS / 4; ++c) {
        const i32x4 v = *(const i32x4*)(sta_col + sc * KS + 4 * c);
        sta[4 * c] = v.x; sta[4 * c + 1] = v.y; sta[4 * c + 2] = v.z; sta[4 * c + 3] = v.w;
    }
}

// ------------------------------------------------------------------------------------------------
// Stage 1 on the 16-bit matrix pipe with fp32-class operands ("f16x2").
//
// Measured on MI355X (tools/mfma_peak*.hip, tools/valu_rate.hip, tools/mfma_overlap.hip): v_mfma_f32_16x16x4_f32 runs at the
// fp32 VECTOR rate and does not overlap with VALU work, so the fp32-MFMA kernel above is bound by the sum of both; a 16-bit
// 32x32x16 MFMA does 16x the FLOPs in the same 32 cycles (and hides ~10 of them behind vector work).
//
//  * v_mfma_f32_32x32x16_f16: D[ch, node] for 32 channels x 32 nodes. A wave owns TWO 16-station tiles (lanes
//    0-15/32-47 and 16-31/48-63). Lane (j = lane&31, h = lane>>5) holds D channels 8*(r>>2) + 4h + (r&3), r = 0..15,
//    of node j; K-step ks of the next layer consumes registers 8ks..8ks+7 of both lanes of a node (16 channels), so an
//    accumulator block becomes B operands without any cross-lane movement. All our channel groups are 30 wide: one
//    32-block each; the two padding slots of a block (channels 30, 31: lane h = 1, registers 14, 15) carry the Mask
//    inputs of `cat(h, n, Mask)`.
//  * raw inputs arrive as 32-B rows [x0 | x1] of 8 fp16 each (x = Slice || Mask), written by k_split_rows, stored PLANAR
//    (piece q of row p at q * rows * 16 + p * 16: a half-wave reads one piece of 32 consecutive rows as 512 contiguous bytes).
//    A neighbour's hidden state is 2 MFMAs (K = 16 = the two 8-wide pieces).
//  * mean_k PReLU_s(z_k) = sum_k (al z_k + be |z_k|): two fused multiply-adds per neighbour value into one accumulator.
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int XROW = 32;                 // bytes per split input row: two 16-B pieces
constexpr int XPC = 16;                  // bytes per piece
#ifndef GENIE_H2_THREADS
#define GENIE_H2_THREADS 512             // waves x 64 of one k_stage1_h2 workgroup (tuning builds: 1024 = 16 waves sharing the weight image)
#endif
constexpr int H2_THREADS = GENIE_H2_THREADS;

// ---- f16x2: x ~ x0 + x1 with x0 = rn16(x), x1 = rn16(x - x0): 11 + 1 + 11 significant bits, i.e. within one fp32 ulp of x
// (exact when the residual needs <= 11 bits) while x1 stays a normal fp16 number (|x| >= 2^-2), within 2^-25 absolute below
// that (fp16 subnormals: the MFMA keeps them, tools/h2_probe.hip). Overflow needs |x| > 65504.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MFMA32H(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (a)), __builtin_bit_cast(f16x8, (b)), (c), 0, 0, 0)
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {            // round to nearest even, both halves
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float sub_f16_lo(float x, unsigned p) {              // x - float(p.lo), one exact fp32 operation
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(x));
    return r;
}
__device__ __forceinline__ float sub_f16_hi(float x, unsigned p) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(x));
    return r;
}
constexpr unsigned H2_SIXTEENTH = 0x2c002c00u;        // (1/16, 1/16) as an fp16 pair
__device__ __forceinline__ unsigned pk_mul_f16(unsigned p, unsigned c) {
    unsigned r;
    asm("v_pk_mul_f16 %0, %1, %2" : "=v"(r) : "v"(p), "v"(c));
    return r;
}
// one fp16 piece (low 16 bits) of v; codes as in build_h2_table
__device__ __forceinline__ unsigned f16_piece(float v, int piece) {
    if (piece >= 2) { v *= 16.f; piece -= 2; }
    const unsigned p0 = cvt_pk_f16(v, 0.f);
    if (piece == 0) return p0 & 0xffffu;
    const float r = sub_f16_lo(v, p0);
    return cvt_pk_f16(r, 0.f) & 0xffffu;
}
__device__ __forceinline__ unsigned f16_piece_w(float v, int piece) {            // weight pieces: code 1 = rn16(16 (W - W0))
    if (piece != 1) return f16_piece(v, piece);
    const unsigned p0 = cvt_pk_f16(v, 0.f);
    return cvt_pk_f16(16.f * sub_f16_lo(v, p0), 0.f) & 0xffffu;
}
// the split rows of one [Slice || Mask] row: two fp16 planes
__device__ __forceinline__ void store_split_row(unsigned* __restrict__ out, long long rows, long long p, const float (&v)[8]) {
    u32x4 o0, o1;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        o0[d] = cvt_pk_f16(v[2 * d], v[2 * d + 1]);
        o1[d] = cvt_pk_f16(sub_f16_lo(v[2 * d], o0[d]), sub_f16_hi(v[2 * d + 1], o0[d]));
    }
    *(u32x4*)(out + p * 4) = o0;
    *(u32x4*)(out + (rows + p) * 4) = o1;
}

__global__ void k_pack_h2(const float* __restrict__ raw, const int32_t* __restrict__ tbl, float* __restrict__ out, int nfrag,
                          int ntail) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < nfrag * 64) {
        u32x4 o;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            unsigned u[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int32_t ent = tbl[idx * 8 + 2 * d + k];
                u[k] = ent < 0 ? 0u : f16_piece_w(raw[ent & 0x0fffffff], (ent >> 28) & 3);
            }
            o[d] = u[0] | (u[1] << 16);
        }
        ((u32x4*)out)[idx] = o;
    } else if (idx < nfrag * 64 + ntail) {     // fp32 tail: bias blocks and PReLU slopes
        const int k = idx - nfrag * 64;
        const int32_t ent = tbl[nfrag * 512 + k];
        out[nfrag * 256 + k] = ent < 0 ? 0.f : raw[ent];
    }
}

// Input-range check of the f16x2 path (round 5). The fp16 range guard (k_h2_range) bounds the hidden states for inputs in [-1, 1],
// what the reference's embedding produces; the bound is linear in the input magnitude, so the kernels are valid up to
// |input| <= x_ok = 60000 / bound (~900 with default-initialised weights). A row beyond that -- not producible by the reference's
// pipeline, but accepted by its fp32 arithmetic -- would overflow to inf without a trace: the split pass, which touches every input
// anyway, reports it through a word of host-mapped memory (one system-scope atomic max, executed by offending rows only), which the
// host reads without synchronising at its next call (genie_input_range): the call fails loudly and the context falls back to the
// fp32 kernels.
__device__ __forceinline__ void flag_input_range(const float (&v)[8], float x_ok, unsigned* __restrict__ flag) {
    const float m = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))),
                          fmaxf(fmaxf(fabsf(v[4]), fabsf(v[5])), fmaxf(fabsf(v[6]), fabsf(v[7]))));
    if (!(m <= x_ok) && flag != nullptr)          // (also true for NaN)
        __hip_atomic_fetch_max(flag, __float_as_uint(m == m ? m : __builtin_inff()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// [Slice || Mask] rows (8 fp32) -> 32-B rows of two fp16x8 pieces
// sta_user (internal station -> caller's station, or null): the rows of a source node are written in the station processing order
__global__ void k_split_rows(const float* __restrict__ slice, const float* __restrict__ mask, long long rows,
                             unsigned* __restrict__ out, const int32_t* __restrict__ sta_user, int S, float* __restrict__ mm,
                             float x_ok = __builtin_inff(), unsigned* __restrict__ flag = nullptr) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= rows) return;
    long long pu = p;
    if (sta_user != nullptr) {
        const long long g = p / S;
        pu = g * S + sta_user[(int)(p - g * S)];
    }
    const f32x4 s = *(const f32x4*)(slice + pu * 4), m = *(const f32x4*)(mask + pu * 4);
    if (mm != nullptr) mm[p] = fmaxf(fmaxf(m.x, m.y), fmaxf(m.z, m.w));            // the message mask of stage 2 (module.py:226)
    const float v[8] = {s.x, s.y, s.z, s.w, m.x, m.y, m.z, m.w};
    flag_input_range(v, x_ok, flag);
    store_split_row(out, rows, p, v);
}

// Same with a station processing order, one workgroup per source node: the node's S rows are read in the caller's order
// (coalesced), staged in LDS, and written in processing order (coalesced); S <= SPLIT_G_MAXS rows fit the 64-KB staging buffer.
constexpr int SPLIT_G_MAXS = 2048;
__global__ __launch_bounds__(256) void k_split_rows_g(const float* __restrict__ slice, const float* __restrict__ mask, int S,
                                                      unsigned* __restrict__ out, const int32_t* __restrict__ sta_user,
                                                      float* __restrict__ mm, long long rows, float x_ok = __builtin_inff(),
                                                      unsigned* __restrict__ flag = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float stg[];        // [S][8]: Slice row | Mask row
    const long long base = (long long)blockIdx.x * S;
    for (int r = threadIdx.x; r < S; r += blockDim.x) {
        *(f32x4*)(stg + r * 8) = *(const f32x4*)(slice + (base + r) * 4);
        *(f32x4*)(stg + r * 8 + 4) = *(const f32x4*)(mask + (base + r) * 4);
    }
    __syncthreads();
    for (int r = threadIdx.x; r < S; r += blockDim.x) {
        const int u = sta_user[r];
        const f32x4 s = *(const f32x4*)(stg + u * 8), m = *(const f32x4*)(stg + u * 8 + 4);
        mm[base + r] = fmaxf(fmaxf(m.x, m.y), fmaxf(m.z, m.w));
        const float v[8] = {s.x, s.y, s.z, s.w, m.x, m.y, m.z, m.w};
        flag_input_range(v, x_ok, flag);
        store_split_row(out, rows, base + r, v);
    }
}

// exact PReLU in two VALU ops for any slope: max(x, s*x) when s <= 1, min(x, s*x) otherwise, as med3(x, s*x, +-inf)
__device__ __forceinline__ f32x16 prelu16(f32x16 x, float s, float sel) {
    f32x16 y;
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = __builtin_amdgcn_fmed3f(x[r], x[r] * s, sel);
    return y;
}
__device__ __forceinline__ f32x16 bias16(const float* lbias, int blk, int h) {
    f32x16 y;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const f32x4 t = *(const f32x4*)(lbias + blk * 32 + 8 * b + 4 * h);
        y[4 * b] = t.x; y[4 * b + 1] = t.y; y[4 * b + 2] = t.z; y[4 * b + 3] = t.w;
    }
    return y;
}
// training forward: a 32-channel accumulator (register r = channel 8 (r >> 2) + 4 h + (r & 3)) as two 16-float blocks of the
// block-planar save buffer [blk][P][16] the backward passes read (channels 30, 31 are padding there: zero)
__device__ __forceinline__ void h2_save32(float* __restrict__ save, long long Pn, int blk0, long long p, int h, const f32x16& v) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f32x4 o = {v[4 * m], v[4 * m + 1], v[4 * m + 2], v[4 * m + 3]};
        if (m == 3 && h == 1) { o.z = 0.f; o.w = 0.f; }
        *(f32x4*)(save + ((size_t)(blk0 + (m >> 1)) * Pn + p) * 16 + 8 * (m & 1) + 4 * h) = o;
    }
}
// ------------------------------------------------------------------------------------------------
// STAGE 1, f16x2 form. An activation x is split into x0 = rn16(x), x1 = rn16(x - x0) (one v_cvt_pk_f16_f32 per pair and
// piece, one v_fma_mix_f32 per value) and a weight into W0 = rn16(W), W1' = rn16(16 (W - W0)); a K-step is THREE products,
// smallest first: W0 x1 + W1' (x0 / 16) + W0 x0 (x0 / 16: one v_pk_mul_f16 per pair). The scaled pair keeps the weight's second
// piece a normal fp16 number; without it that piece falls into fp16's subnormal range (absolute floor 2^-25) and the hidden
// states lose ~2x in accuracy (oracle-level emulation of the arithmetic on the o1_20x500 fixture: x_latent rms error vs fp64
// 1.30e-7 unscaled, 0.81e-7 scaled; three exact bf16 pieces with six products, the round-1..3 form: 0.68e-7; the reference's
// own fp32: 1.13e-7). Dropped: W1 x1 (2^-24 of a product) and the last-bit rounding of x1: the result is fp32-CLASS, not
// bit-for-bit fp32. The input layer (K = 8: [x0 ; x1] fill one K = 16 step) is computed 16 x too large as a whole,
// [P|P][x0;x1] + [Q|Q][x0;x1] with P + Q = 16 W and C = 16 b: the neighbour sums absorb the factor in their constants, the
// node's own h0 in its PReLU. Per wave-tile (32 nodes): 120 MFMAs (48 neighbour recompute, 24 layer 1, 36 u / v / c,
// 12 wu / wv; the bf16x3 form needed 216) and ~1500 vector instructions (2050).
// ------------------------------------------------------------------------------------------------
// lane k of every row of 16 lanes, broadcast to the row (DPP row_newbcast, gfx90a+)
template <int K_>
__device__ __forceinline__ int row_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + K_, 0xf, 0xf, false); }
__device__ __forceinline__ int row_bcast_dyn(int v, int k) {     // k is a compile-time constant after unrolling
    switch (k) {
        case 0: return row_bcast<0>(v); case 1: return row_bcast<1>(v); case 2: return row_bcast<2>(v); case 3: return row_bcast<3>(v);
        case 4: return row_bcast<4>(v); case 5: return row_bcast<5>(v); case 6: return row_bcast<6>(v); case 7: return row_bcast<7>(v);
        case 8: return row_bcast<8>(v); case 9: return row_bcast<9>(v); case 10: return row_bcast<10>(v); case 11: return row_bcast<11>(v);
        case 12: return row_bcast<12>(v); case 13: return row_bcast<13>(v); case 14: return row_bcast<14>(v); default: return row_bcast<15>(v);
    }
}
template <int KS_>
__device__ __forceinline__ void split8h(const f32x16& v, u32x4 (&p)[3], unsigned sixteenth) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const float a = v[8 * KS_ + 2 * d], b = v[8 * KS_ + 2 * d + 1];
        const unsigned p0 = cvt_pk_f16(a, b);
        p[0][d] = p0;
        p[1][d] = cvt_pk_f16(sub_f16_lo(a, p0), sub_f16_hi(b, p0));
        p[2][d] = pk_mul_f16(p0, sixteenth);
    }
}
// the three partial products of one K-step for N independent accumulators sharing the B pieces {x0, x1, x0 / 16}
template <int N>
__device__ __forceinline__ void mma3(f32x16 (&acc)[N], const f32x4* lw, const int (&f0)[N], int lane, const u32x4 (&b)[3]) {
    f32x4 w[N][2];
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
        for (int p = 0; p < 2; ++p) w[k][p] = lw[(f0[k] + p) * 64 + lane];
    constexpr int WP[3] = {0, 1, 0}, BP[3] = {1, 2, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = MFMA32H(w[k][WP[t]], b[BP[t]], acc[k]);
}

template <int KS, int KP, bool EDGES, bool BIG, bool ABS = false, bool PCSR = false>
__global__ __launch_bounds__(H2_THREADS) void k_stage1_h2(DaArgs a) {
    // PCSR: irregular product graph (use_subgraph). A wave item is 32 consecutive product nodes; the neighbours of a node are
    // product-node ids from the product-level CSRs (at most KS / KP of them: a missing one is the node itself with weight 0,
    // the mean of an empty neighbourhood is 0); everything after the neighbour phase is the same code.
    // (irregular product graphs: the EDGES terms and the ABS position pieces are per PRODUCT node there, indexed by product-node id)
    typedef typename std::conditional<BIG, unsigned long long, unsigned>::type off_t_;
    constexpr int NF4 = H2_IMG_FLOATS / 4;
    __shared__ f32x4 lw[NF4];
    for (int i = threadIdx.x; i < NF4; i += H2_THREADS) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lbias = (const float*)(lw + H2_FRAGS * 64);
    const float* lscal = lbias + H2_NBIAS * 32;
    const float a0 = lscal[0], a1 = lscal[3], a21 = lscal[4], a22 = lscal[5];
    const float s11 = compose_slopes(a0, lscal[1]), s12 = compose_slopes(a0, lscal[2]);
    const float inf = __builtin_inff();
    const float sel0 = a0 <= 1.f ? inf : -inf, sel1 = a1 <= 1.f ? inf : -inf;
    const float sel21 = a21 <= 1.f ? inf : -inf, sel22 = a22 <= 1.f ? inf : -inf;
    // mean_k PReLU_s(z_k) = al * sum z_k + be * sum |z_k|; the z_k arrive 16 x too large
    const float al1 = (1.f + s11) / (32.f * KS), be1 = (1.f - s11) / (32.f * KS);
    const float al2 = (1.f + s12) / (32.f * KP), be2 = (1.f - s12) / (32.f * KP);
    const unsigned c16 = __builtin_amdgcn_readfirstlane(H2_SIXTEENTH);

    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, half = (lane >> 4) & 1, jj = lane & 15;
    const bool hi = h != 0;
    const int S = a.S;
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave, a.gi0);
    const char* xs = (const char*)a.xs;
    const off_t_ la = hi ? (off_t_)a.xs_plane : (off_t_)0;          // lane h = 0 loads x0, lane h = 1 loads x1: [x0 ; x1] is one K = 16 step
    const off_t_ gstride = (off_t_)((unsigned)S * (unsigned)XPC);

    const f32x4 fa0 = lw[(H2_FA + 0) * 64 + lane], fa1 = lw[(H2_FA + 1) * 64 + lane];
    // use_absolute_pos: the six position columns of init_trns are one more K = 16 step per unit, B = {station piece, source piece}
    f32x4 fp0, fp1;
    if (ABS) { fp0 = lw[(H2_FABS + 0) * 64 + lane]; fp1 = lw[(H2_FABS + 1) * 64 + lane]; }
    const unsigned tp_h = (unsigned)h * 8u;           // piece tables: [node][piece] x 8 B
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    f32x16 biasA = bias16(lbias, 0, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) biasA[r] *= 16.f;

    int jt = jj;
    auto fetch_ids = [&](long long pit_, int& idv_, int& sc_, bool& valid_, int (&sta_)[KS]) {
        int gi0, tb0, gi1, tb1;
        w.decode(2 * pit_, gi0, tb0);
        const bool second = 2 * pit_ + 1 < w.nitems;
        w.decode(second ? 2 * pit_ + 1 : 2 * pit_, gi1, tb1);
        idv_ = a.src_tab[(half ? gi1 : gi0) * 16 + jt];
        const int s = (half ? tb1 : tb0) * 16 + jt;
        valid_ = s < S && (second || !half);
        sc_ = s < S ? s : S - 1;
        load_sta_ids<KS>(a.sta_col, sc_, sta_);
    };
    constexpr int KPP = PCSR ? KP : 1;
    auto fetch_pcsr = [&](long long pit_, long long& p_, bool& valid_, int (&sta_)[KS], int (&src_)[KPP], int& ds_, int& dp_) {
        const long long pr = ptile_at(a.ptile, pit_) * 32 + (lane & 31);
        valid_ = pr < a.Pn;
        p_ = valid_ ? pr : a.Pn - 1;
        const int eb = a.sta_rowptr[p_], fb = a.src_rowptr[p_];
        ds_ = a.sta_rowptr[p_ + 1] - eb;
        dp_ = a.src_rowptr[p_ + 1] - fb;
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const int v = a.sta_col[max(eb + min(q, ds_ - 1), 0)];
            sta_[q] = q < ds_ ? v : (int)p_;
        }
#pragma unroll
        for (int q = 0; q < KPP; ++q) {
            const int v = a.src_col[max(fb + min(q, dp_ - 1), 0)];
            src_[q] = q < dp_ ? v : (int)p_;
        }
    };
    int idv = 0, sc = 0, sta_id[KS], src_id[KPP], dgs = 0, dgp = 0;
    long long pcur = 0;
    bool valid = false;
    // wave items: Cartesian = pairs of (source node, station tile) items of the XCD-aware sweep; PCSR = 32 consecutive product nodes
    PtileIter ptw(PCSR ? (a.Pn + 31) / 32 : 0, H2_THREADS / 64, wave);       // PCSR: positions in the processing order of the 32-node items
    const long long pit0 = PCSR ? ptw.i : w.it;
    const long long pstride = PCSR ? ptw.stride : w.stride;
    const long long pend = PCSR ? ptw.end : (w.nitems + 1) / 2;
    if (pit0 < pend) {
        if constexpr (PCSR) fetch_pcsr(pit0, pcur, valid, sta_id, src_id, dgs, dgp);
        else fetch_ids(pit0, idv, sc, valid, sta_id);
    }
    for (long long pit = pit0, pnext = 0; pit < pend; pit = pnext) {
        asm volatile("" : "+v"(lane));    // keeps the LDS fragment reads inside the loop (LICM would park them all in VGPRs)
        // idv: every row of 16 lanes holds {source node, its KP neighbours} of its own tile: one DPP row broadcast per id
        const int g = PCSR ? 0 : row_bcast<0>(idv);
        const long long p = PCSR ? pcur : (long long)g * S + sc;
        const off_t_ gbase0 = PCSR ? (off_t_)(unsigned long long)p * (off_t_)XPC : (off_t_)(unsigned)g * gstride;
        const unsigned sbase0 = PCSR ? 0u : (unsigned)sc * (unsigned)XPC;
        off_t_ gbase = gbase0 + la;                  // + this lane's plane: one multiply-add per neighbour row address
        off_t_ sbase = (off_t_)sbase0 + la;
        const int srcv = idv;
        // PCSR: per-lane weights of a present neighbour (16 x scaling and 1 / degree folded in)
        float alS = 0.f, beS = 0.f, alP = 0.f, beP = 0.f;
        if constexpr (PCSR) {
            const float is = 1.f / (float)max(dgs, 1), ip = 1.f / (float)max(dgp, 1);
            alS = (1.f + s11) * 0.03125f * is; beS = (1.f - s11) * 0.03125f * is;
            alP = (1.f + s12) * 0.03125f * ip; beP = (1.f - s12) * 0.03125f * ip;
        }

        // unit u: 0 = the node itself, 1..KS = station neighbours, KS+1..KS+KP = source neighbours
        constexpr int NU = 1 + KS + KP;
        static_assert(NU % 2 == 0, "units are processed in pairs");
        constexpr int DEPTH = ABS ? 4 : GENIE_H2_DEPTH;
        u32x4 buf[NU];
        u32x2 tp[NU], tso, tgo;          // ABS: the unit's own position piece; this tile's station / source piece
        if (ABS) {
            tso = *(const u32x2*)((const char*)a.abs_ts + (tp_h + (unsigned)(PCSR ? (int)p : sc) * 16u));
            tgo = *(const u32x2*)((const char*)a.abs_tg + (tp_h + (unsigned)(PCSR ? (int)p : g) * 16u));
        }
        auto issue = [&](int u) {
            off_t_ off;
            if (u == 0) off = gbase + sbase0;
            else if (PCSR) off = (off_t_)(unsigned)(u <= KS ? sta_id[u - 1] : src_id[(u - KS - 1) % KPP]) * (off_t_)XPC + la;
            else if (u <= KS) off = gbase + (unsigned)sta_id[u - 1] * (unsigned)XPC;
            else {
                const unsigned nb = (unsigned)row_bcast_dyn(srcv, u - KS);
                off = (BIG ? (off_t_)nb * gstride : (off_t_)__umul24(nb, (unsigned)gstride)) + sbase;
            }
            if (ABS && u > 0) {
                if (u <= KS) tp[u] = *(const u32x2*)((const char*)a.abs_ts + (tp_h + (unsigned)sta_id[u - 1] * 16u));
                else if (PCSR) tp[u] = *(const u32x2*)((const char*)a.abs_tg + (tp_h + (unsigned)src_id[(u - KS - 1) % KPP] * 16u));
                else tp[u] = *(const u32x2*)((const char*)a.abs_tg + (tp_h + (unsigned)row_bcast_dyn(srcv, u - KS) * 16u));
            }
            if (ABL(a, 12) && u > 0) { buf[u] = buf[0]; return; }     // tuning: no neighbour-row loads
            if (ABL(a, 13) && u >= 1 && u <= KS) { buf[u] = buf[0]; return; }     // tuning bit 13: the station-neighbour units cost nothing (upper bound of ANY caching of them)
            buf[u] = *(const u32x4*)(xs + off);
        };
        const u32x4 own0 = *(const u32x4*)(xs + (gbase0 + sbase0));         // x0 of the own row (lanes h = 1: Mask pads)
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) issue(u);

        f32x16 sn, h0;          // sn: running mean_k PReLU_s(z_k) = sum_k (al z_k + be |z_k|), two fused multiply-adds per value
        u32x4 h0p[2][3], n1p[2][3], n2p[2][3];
        unsigned m01[3], m23[3];          // Mask pieces {x0, x1, x0 / 16} (lanes h = 1): fp16 pairs (M0,M1) and (M2,M3)
#pragma unroll
        for (int u = 0; u < NU; u += 2) {
#pragma unroll
            for (int d = 0; d < 2; ++d)
                if (u + DEPTH + d < NU) issue(u + DEPTH + d);
            asm volatile("" : "+v"(buf[u]), "+v"(buf[u + 1]));
            f32x16 z0, z1;
            if (ABS) {
                auto posb = [&](int uu) {
                    return uu == 0 ? u32x4{tso.x, tso.y, tgo.x, tgo.y}
                                   : uu <= KS ? u32x4{tp[uu].x, tp[uu].y, tgo.x, tgo.y} : u32x4{tso.x, tso.y, tp[uu].x, tp[uu].y};
                };
                const u32x4 p0 = posb(u), p1 = posb(u + 1);
                z0 = MFMA32H(fp1, p0, biasA); z1 = MFMA32H(fp1, p1, biasA);
                z0 = MFMA32H(fa1, buf[u], z0); z1 = MFMA32H(fa1, buf[u + 1], z1);
                z0 = MFMA32H(fp0, p0, z0); z1 = MFMA32H(fp0, p1, z1);
            } else {
                const bool skip0 = ABL(a, 13) && u >= 1 && u <= KS, skip1 = ABL(a, 13) && u + 1 >= 1 && u + 1 <= KS;
                z0 = biasA; z1 = biasA;
                if (!skip0) z0 = MFMA32H(fa1, buf[u], biasA);
                if (!skip1) z1 = MFMA32H(fa1, buf[u + 1], biasA);
                if (!skip0) z0 = MFMA32H(fa0, buf[u], z0);
                if (!skip1) z1 = MFMA32H(fa0, buf[u + 1], z1);
            }
            if (ABS) {
            z0 = MFMA32H(fa0, buf[u], z0);
            z1 = MFMA32H(fa0, buf[u + 1], z1);
            }
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                f32x16 z = d == 0 ? z0 : z1;
                const int uu = u + d;
                if (uu == 0) {
                    if (a.save != nullptr && valid) {
                        f32x16 zu;
#pragma unroll
                        for (int r = 0; r < 16; ++r) zu[r] = z[r] * 0.0625f;
                        h2_save32(a.save, a.Pn, SV_Z0, p, h, zu);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) h0[r] = __builtin_amdgcn_fmed3f(z[r] * 0.0625f, z[r] * (0.0625f * a0), sel0);
                    m01[0] = own0.z;   m23[0] = own0.w;
                    m01[1] = buf[0].z; m23[1] = buf[0].w;      // lane h = 1: buf = x1
                    m01[2] = pk_mul_f16(own0.z, c16); m23[2] = pk_mul_f16(own0.w, c16);
                } else {
                    float al = uu <= KS ? al1 : al2, be = uu <= KS ? be1 : be2;
                    if constexpr (PCSR) {
                        const bool present = uu <= KS ? uu - 1 < dgs : uu - KS - 1 < dgp;
                        al = present ? (uu <= KS ? alS : alP) : 0.f;
                        be = present ? (uu <= KS ? beS : beP) : 0.f;
                    }
                    if (ABL(a, 13) && uu <= KS) {
                        // tuning bit 14 (with 13): what a cache of the station neighbours' rows in LDS would cost instead: address, four
                        // ds_read_b128 of a 136-B-pitch row picked by the neighbour's station id, sixteen adds (stand-in data: the weight image)
                        if (uu == 1) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) sn[r] = 0.f;
                        }
                        if (ABL(a, 14)) {
                            const unsigned x = ((unsigned)sta_id[uu - 1] * 136u + (unsigned)h * 64u) % 49152u;
                            const char* lb = (const char*)lw + (x & ~15u);
#pragma unroll
                            for (int k4 = 0; k4 < 4; ++k4) {
                                const f32x4 t = *(const f32x4*)(lb + 16 * k4);
                                sn[4 * k4] += t.x; sn[4 * k4 + 1] += t.y; sn[4 * k4 + 2] += t.z; sn[4 * k4 + 3] += t.w;
                            }
                        }
                    } else if (uu == 1 || uu == KS + 1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) sn[r] = fmaf(be, __builtin_fabsf(z[r]), al * z[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) sn[r] = fmaf(be, __builtin_fabsf(z[r]), fmaf(al, z[r], sn[r]));
                    }
                }
                if (uu == KS) { split8h<0>(sn, n1p[0], c16); split8h<1>(sn, n1p[1], c16); }
                if (uu == NU - 1) { split8h<0>(sn, n2p[0], c16); split8h<1>(sn, n2p[1], c16); }
            }
            asm volatile("" : "+v"(sn), "+v"(gbase), "+v"(sbase), "+v"(jt));
        }
        int idv_n = 0, sc_n = 0, sta_n[KS], src_n[KPP], dgs_n = 0, dgp_n = 0;
        long long p_n = 0;
        bool valid_n = false;
#pragma unroll
        for (int k = 0; k < KS; ++k) sta_n[k] = 0;
#pragma unroll
        for (int k = 0; k < KPP; ++k) src_n[k] = 0;
        pnext = pit + pstride;
        const bool has_next = pnext < pend;
        if (has_next) {
            if constexpr (PCSR) fetch_pcsr(pnext, p_n, valid_n, sta_n, src_n, dgs_n, dgp_n);
            else fetch_ids(pnext, idv_n, sc_n, valid_n, sta_n);
        }
        if (a.dbg_h0 != nullptr && valid) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = 8 * (r >> 2) + 4 * h + (r & 3);
                if (ch < 30) a.dbg_h0[p * 30 + ch] = h0[r];
            }
        }
        split8h<0>(h0, h0p[0], c16);
        split8h<1>(h0, h0p[1], c16);
#pragma unroll
        for (int q = 0; q < 3; ++q) {     // padding slots (channels 30, 31) carry the Mask: [h0 | M0 M1], [n | M2 M3]
            h0p[1][q].w = hi ? m01[q] : h0p[1][q].w;
            n1p[1][q].w = hi ? m23[q] : n1p[1][q].w;
            n2p[1][q].w = hi ? m23[q] : n2p[1][q].w;
        }
        // ---- layer 1: tr_t = l1_t{1,2}_2 [h0 || n_t || Mask], both halves at once
        f32x16 acc[2] = {bias16(lbias, 1, h), bias16(lbias, 2, h)};
        if (EDGES) {   // DataAggregationEdges: static per-station / per-source-node terms of layer 1
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const f32x4 es = *(const f32x4*)(a.eb_sta + (PCSR ? p : (long long)sc) * 48 + 8 * b + 4 * h);
                const f32x4 eg = *(const f32x4*)(a.eb_src + (PCSR ? p : (long long)g) * 48 + 8 * b + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[0][4 * b + e] += es[e]; acc[1][4 * b + e] += eg[e]; }
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int f0[2] = {H2_FL1 + (0 * 4 + ks) * 2, H2_FL1 + (1 * 4 + ks) * 2};
            mma3<2>(acc, lw, f0, lane, h0p[ks]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {     // the neighbour-mean blocks differ per half: interleave by hand
            const int fa_ = H2_FL1 + (0 * 4 + 2 + ks) * 2, fb_ = H2_FL1 + (1 * 4 + 2 + ks) * 2;
            f32x4 wa[2], wb[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) { wa[q] = lw[(fa_ + q) * 64 + lane]; wb[q] = lw[(fb_ + q) * 64 + lane]; }
            constexpr int WP[3] = {0, 1, 0}, BP[3] = {1, 2, 0};
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                acc[0] = MFMA32H(wa[WP[t]], n1p[ks][BP[t]], acc[0]);
                acc[1] = MFMA32H(wb[WP[t]], n2p[ks][BP[t]], acc[1]);
            }
        }
        if (a.save != nullptr && valid) { h2_save32(a.save, a.Pn, SV_T, p, h, acc[0]); h2_save32(a.save, a.Pn, SV_T + 2, p, h, acc[1]); }
        acc[0] = prelu16(acc[0], a1, sel1);
        acc[1] = prelu16(acc[1], a1, sel1);
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));
        if (a.dbg_h1 != nullptr && valid) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = 8 * (r >> 2) + 4 * h + (r & 3);
                    if (ch < 30) a.dbg_h1[p * 60 + 30 * t + ch] = acc[t][r];
                }
        }
        // ---- u, v and the node-local layer-2 terms c from h1 = [h1a (30) | M0 M1 | h1b (30) | M2 M3]
        f32x16 o3[3] = {bias16(lbias, 3, h), bias16(lbias, 4, h), bias16(lbias, 5, h)};
        if (EDGES) {   // ... and of the node-local layer-2 block c = [o1 (15), 0 | o2 (15), 0]
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const f32x4 es = *(const f32x4*)(a.eb_sta + (PCSR ? p : (long long)sc) * 48 + 32 + 8 * b + 4 * h);
                const f32x4 eg = *(const f32x4*)(a.eb_src + (PCSR ? p : (long long)g) * 48 + 32 + 8 * b + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) { o3[2][4 * b + e] += es[e]; o3[2][8 + 4 * b + e] += eg[e]; }
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            u32x4 hp[2][3];
            split8h<0>(acc[t], hp[0], c16);
            split8h<1>(acc[t], hp[1], c16);
#pragma unroll
            for (int q = 0; q < 3; ++q) hp[1][q].w = hi ? (t == 0 ? m01[q] : m23[q]) : hp[1][q].w;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int ks = 2 * t + kb;
                const int f0[3] = {H2_FUVC + (0 * 4 + ks) * 2, H2_FUVC + (1 * 4 + ks) * 2, H2_FUVC + (2 * 4 + ks) * 2};
                mma3<3>(o3, lw, f0, lane, hp[kb]);
            }
        }
        // node-planar rows (a.np, read by k_stage2_h2): chunk q = 2 b + h of station sc at [g][q][sc] x 16 B inside the node's block
        const long long npb = PCSR ? 0 : (long long)g * S;
        const long long npl = (long long)S * 4;
        if (valid) {
            float* cr = a.np ? a.c + npb * ROWC + (long long)h * npl + (long long)sc * 4 : a.c + p * ROWC + 4 * h;
            const long long cs = a.np ? 2 * npl : 8;
#pragma unroll
            for (int b = 0; b < 4; ++b)
                *(f32x4*)(cr + b * cs) = f32x4{o3[2][4 * b], o3[2][4 * b + 1], o3[2][4 * b + 2], o3[2][4 * b + 3]};
        }
        if (a.save != nullptr && valid) { h2_save32(a.save, a.Pn, SV_UP, p, h, o3[0]); h2_save32(a.save, a.Pn, SV_VP, p, h, o3[1]); }
        o3[0] = prelu16(o3[0], a21, sel21);
        o3[1] = prelu16(o3[1], a22, sel22);
        asm volatile("" : "+v"(o3[0]), "+v"(o3[1]));
        // ---- projected gather operands [wu | wv] = [l2_t1_2[:, 60:90] u | l2_t2_2[:, 60:90] v]
        f32x16 ow[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { ow[0][r] = 0.f; ow[1][r] = 0.f; }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            u32x4 up[3], vp[3];
            if (kb == 0) { split8h<0>(o3[0], up, c16); split8h<0>(o3[1], vp, c16); }
            else { split8h<1>(o3[0], up, c16); split8h<1>(o3[1], vp, c16); }
            f32x4 wa[2], wb[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                wa[q] = lw[(H2_FW + (0 + kb) * 2 + q) * 64 + lane];
                wb[q] = lw[(H2_FW + (2 + kb) * 2 + q) * 64 + lane];
            }
            constexpr int WP[3] = {0, 1, 0}, BP[3] = {1, 2, 0};
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                ow[0] = MFMA32H(wa[WP[t]], up[BP[t]], ow[0]);
                ow[1] = MFMA32H(wb[WP[t]], vp[BP[t]], ow[1]);
            }
        }
        if (valid) {
            float* vr = a.np ? a.wv + npb * ROWW + (long long)h * npl + (long long)sc * 4 : a.wv + p * ROWW + 4 * h;
            const long long vs = a.np ? 2 * npl : 8;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                *(f32x4*)(a.wu + p * ROWW + 8 * b + 4 * h) = f32x4{ow[0][4 * b], ow[0][4 * b + 1], ow[0][4 * b + 2], ow[0][4 * b + 3]};
                *(f32x4*)(vr + b * vs) = f32x4{ow[1][8 + 4 * b], ow[1][8 + 4 * b + 1], ow[1][8 + 4 * b + 2], ow[1][8 + 4 * b + 3]};
            }
        }
        idv = idv_n; sc = sc_n; valid = valid_n; pcur = p_n; dgs = dgs_n; dgp = dgp_n;
#pragma unroll
        for (int k = 0; k < KS; ++k) sta_id[k] = sta_n[k];
#pragma unroll
        for (int k = 0; k < KPP; ++k) src_id[k] = src_n[k];
    }
}

// ------------------------------------------------------------------------------------------------
// stage 2: second pair of neighbour means (of the projected operands), PReLU2 -> x_latent; Bipartite fc1 + PReLU,
// mask gate, and the per-tile station sum.                      module.py:94-96, :229
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stage2(DaArgs a) {
    constexpr int NF4 = (G2_GROUPS * 256 + G2_BIAS * 16 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lbias = (const float*)(lw + G2_GROUPS * 64);
    const float* lscal = lbias + G2_BIAS * 16;
    const float a2 = a.slope2 != nullptr ? *a.slope2 : lscal[0], ab1 = lscal[1];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = a.S;
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave, a.gi0);
    for (; w.it < w.nitems; w.it += w.stride) {
        int gi, tb;
        w.decode(w.it, gi, tb);
        const int g = __builtin_amdgcn_readfirstlane(a.order[gi]);
        asm volatile("" : "+v"(lane));
        const int s = tb * 16 + j;
        const bool valid = s < S;
        const int sc = valid ? s : S - 1;
        const long long p = (long long)g * S + sc;
        f32x4 o[2];
        o[0] = *(const f32x4*)(a.c + p * ROWC + 4 * q);
        o[1] = *(const f32x4*)(a.c + p * ROWC + 16 + 4 * q);
        const float mq = a.mask[p * 4 + q];
        const float eq = q < 3 ? a.edge_attr[p * 3 + q] : 0.f;
        // neighbour means of the projected operands (16-float rows): they ARE the accumulator contributions
        f32x4 n1 = {0.f, 0.f, 0.f, 0.f}, n2 = {0.f, 0.f, 0.f, 0.f};
        {
            const int eb = a.sta_rowptr[sc], ee = a.sta_rowptr[sc + 1];
            const float* base = a.wu + (long long)g * S * ROWW + 4 * q;
            if (!ABL(a, 0)) gather_sum16<false>(base, ROWW, a.sta_col, eb, ee, n1);
            o[0] = fma4(n1, 1.f / (float)max(ee - eb, 1), o[0]);
        }
        {
            const int eb = __builtin_amdgcn_readfirstlane(a.src_rowptr[g]);
            const int ee = __builtin_amdgcn_readfirstlane(a.src_rowptr[g + 1]);
            const float* base = a.wv + (long long)sc * ROWW + 4 * q;
            if (!ABL(a, 1)) gather_sum16<true>(base, (long long)S * ROWW, a.src_col, eb, ee, n2);
            o[1] = fma4(n2, 1.f / (float)max(ee - eb, 1), o[1]);
        }
        if (a.save != nullptr && valid) {
            *(f32x4*)(a.save + ((size_t)(SV_O + 0) * a.Pn + p) * 16 + 4 * q) = o[0];
            *(f32x4*)(a.save + ((size_t)(SV_O + 1) * a.Pn + p) * 16 + 4 * q) = o[1];
        }
        o[0] = prelu4u(o[0], a2);   // x_latent[0:15]  (lane (j,q) holds channels 4q..4q+3, channel 15 is zero)
        o[1] = prelu4u(o[1], a2);   // x_latent[15:30]
        if (a.x_latent != nullptr && valid) {
            float* xl = a.x_latent + p * 30;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (4 * q + r < 15) {
                    xl[4 * q + r] = o[0][r];
                    xl[15 + 4 * q + r] = o[1][r];
                }
            }
        }
        if (a.no_bip) continue;
        // Bipartite message: m_p * PReLU_b1(fc1 [x_latent || edge_attr])
        f32x4 bp[2];
        bp[0] = *(const f32x4*)(lbias + 0 * 16 + 4 * q);
        bp[1] = *(const f32x4*)(lbias + 1 * 16 + 4 * q);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bp[t] = mma_block(bp[t], lw[G2_BP(t, 0) * 64 + lane], o[0]);
            bp[t] = mma_block(bp[t], lw[G2_BP(t, 1) * 64 + lane], o[1]);
            bp[t] = MFMA16(lw[G2_BP(t, 2) * 64 + lane].x, eq, bp[t]);
            if (a.save != nullptr && valid) *(f32x4*)(a.save + ((size_t)(SV_ZB + t) * a.Pn + p) * 16 + 4 * q) = bp[t];
            bp[t] = prelu4u(bp[t], ab1);
        }
        float mm = fmaxf(mq, __shfl_xor(mq, 16));
        mm = fmaxf(mm, __shfl_xor(mm, 32));
        if (!valid) mm = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 v = bp[t] * mm;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                v.x += __shfl_xor(v.x, d);
                v.y += __shfl_xor(v.y, d);
                v.z += __shfl_xor(v.z, d);
                v.w += __shfl_xor(v.w, d);
            }
            if (j == 0) *(f32x4*)(a.part + ((long long)g * a.T + tb) * 32 + 16 * t + 4 * q) = v;
        }
    }
}

// Stage 2 on an irregular product graph (see k_stage1_pcsr). The Bipartite messages of a source node are not the rows of
// whole tiles here, so every node's gated message row is written in place of its c row and k_seg_sum32 sums the row range
// of each source node (product nodes are grouped by source node, process_utils.py:790-794) in row order into the window's `part` slot.
__global__ __launch_bounds__(256) void k_stage2_pcsr(DaArgs a) {
    constexpr int NF4 = (G2_GROUPS * 256 + G2_BIAS * 16 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lbias = (const float*)(lw + G2_GROUPS * 64);
    const float* lscal = lbias + G2_BIAS * 16;
    const float a2 = a.slope2 != nullptr ? *a.slope2 : lscal[0], ab1 = lscal[1];
    __shared__ __attribute__((aligned(16))) float tsc[4 * 16 * 36];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int jl = lane >> 2, ql = lane & 3;      // (node, 16-B chunk) this lane LOADS: four consecutive lanes read one 64-B row
    float* ts = tsc + (threadIdx.x >> 6) * 16 * 36;
    const long long ntiles = (a.Pn + 15) / 16;
    PtileIter pt(ntiles, blockDim.x >> 6, threadIdx.x >> 6);
    for (; pt.i < pt.end; pt.i += pt.stride) {
        const long long tile = ptile_at(a.ptile, pt.i);
        asm volatile("" : "+v"(lane));
        const long long pr = tile * 16 + j;
        const bool valid = pr < a.Pn;
        const long long p = valid ? pr : a.Pn - 1;
        // rows are loaded, summed and activated in the ROW layout (lane = 4 node + chunk: the four lanes of a node read one 64-B
        // row, as in k_stage2_ord), then cross the wave's LDS scratch into the MFMA layout for fc1
        const long long prl = tile * 16 + jl;
        const bool valid_l = prl < a.Pn;
        const long long pl = valid_l ? prl : a.Pn - 1;
        f32x4 o[2];
        o[0] = *(const f32x4*)(a.c + pl * ROWC + 4 * ql);
        o[1] = *(const f32x4*)(a.c + pl * ROWC + 16 + 4 * ql);
        const float mq = a.mask[p * 4 + q];
        const float eq = q < 3 ? a.edge_attr[p * 3 + q] : 0.f;
        f32x4 n1 = {0.f, 0.f, 0.f, 0.f}, n2 = {0.f, 0.f, 0.f, 0.f};
        {
            const int eb = a.sta_rowptr[pl], ee = a.sta_rowptr[pl + 1];
            gather_sum16<false>(a.wu + 4 * ql, ROWW, a.sta_col, eb, ee, n1);
            o[0] = fma4(n1, 1.f / (float)max(ee - eb, 1), o[0]);
        }
        {
            const int eb = a.src_rowptr[pl], ee = a.src_rowptr[pl + 1];
            gather_sum16<false>(a.wv + 4 * ql, ROWW, a.src_col, eb, ee, n2);
            o[1] = fma4(n2, 1.f / (float)max(ee - eb, 1), o[1]);
        }
        if (a.save != nullptr && valid_l) {      // training forward: pre-activations of x_latent (chunk ql of node jl: the blocks' own layout)
            *(f32x4*)(a.save + ((size_t)(SV_O + 0) * a.Pn + pl) * 16 + 4 * ql) = o[0];
            *(f32x4*)(a.save + ((size_t)(SV_O + 1) * a.Pn + pl) * 16 + 4 * ql) = o[1];
        }
        o[0] = prelu4u(o[0], a2);
        o[1] = prelu4u(o[1], a2);
        if (a.x_latent != nullptr && valid_l) {
            float* xl = a.x_latent + pl * 30;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (4 * ql + r < 15) {
                    xl[4 * ql + r] = o[0][r];
                    xl[15 + 4 * ql + r] = o[1][r];
                }
            }
        }
        if (a.no_bip) continue;         // last pass of the association heads: x_latent only (wave-uniform)
        *(f32x4*)(ts + jl * 36 + 4 * ql) = o[0];
        *(f32x4*)(ts + jl * 36 + 16 + 4 * ql) = o[1];
        GSYNC();
        o[0] = *(const f32x4*)(ts + j * 36 + 4 * q);
        o[1] = *(const f32x4*)(ts + j * 36 + 16 + 4 * q);
        GSYNC();        // the scratch is rewritten by the next tile of this wave
        f32x4 bp[2];
        bp[0] = *(const f32x4*)(lbias + 0 * 16 + 4 * q);
        bp[1] = *(const f32x4*)(lbias + 1 * 16 + 4 * q);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bp[t] = mma_block(bp[t], lw[G2_BP(t, 0) * 64 + lane], o[0]);
            bp[t] = mma_block(bp[t], lw[G2_BP(t, 1) * 64 + lane], o[1]);
            bp[t] = MFMA16(lw[G2_BP(t, 2) * 64 + lane].x, eq, bp[t]);
            if (a.save != nullptr && valid) *(f32x4*)(a.save + ((size_t)(SV_ZB + t) * a.Pn + p) * 16 + 4 * q) = bp[t];
            bp[t] = prelu4u(bp[t], ab1);
        }
        float mm = fmaxf(mq, __shfl_xor(mq, 16));
        mm = fmaxf(mm, __shfl_xor(mm, 32));
        if (valid) {      // the message row replaces the c row of the node (read above by these same lanes)
            *(f32x4*)(a.c + p * ROWC + 4 * q) = bp[0] * mm;
            *(f32x4*)(a.c + p * ROWC + 16 + 4 * q) = bp[1] * mm;
        }
    }
}

// Stage 2 on an irregular product graph with the station sum FOLDED IN (round 6): a wave owns whole source nodes. The product nodes
// of a source node are one contiguous row range (process_utils.py:790-794: `seg_rowptr`), so the wave walks that range 16 nodes at a
// time -- the arithmetic of k_stage2_pcsr per tile -- and keeps the gated Bipartite messages of its node columns in two accumulators
// across the tiles; one butterfly over the 16 columns at the end gives the source node's station sum, written straight to the
// window's `part` row. No message row is written back over c (128 B per product node) and read again by k_seg_sum32 (one launch
// less). Source nodes are taken in processing order (space-filling curve), chunked by XCD as the tiles of k_stage2_pcsr are, so the
// wv rows the waves of a CU gather overlap. Inference only (no kept pre-activations, no x_latent output): the other calls keep
// k_stage2_pcsr + k_seg_sum32. Station sum order: per node column over the tiles (rows j, j + 16, ...), then the butterfly.
// Measured and dropped (profiles/EXPERIMENTS.md, round 6): the range's own `wu` rows staged in the wave's LDS for the 8 station
// gathers of every node (48 KB per workgroup, a bounds check per neighbour): 0.310 -> 0.345 ms per window, back to the unfused time.
__global__ __launch_bounds__(256) void k_stage2_pseg(DaArgs a, const int32_t* __restrict__ seg_rowptr, int n_src) {
    constexpr int NF4 = (G2_GROUPS * 256 + G2_BIAS * 16 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lbias = (const float*)(lw + G2_GROUPS * 64);
    const float* lscal = lbias + G2_BIAS * 16;
    const float a2 = a.slope2 != nullptr ? *a.slope2 : lscal[0], ab1 = lscal[1];
    __shared__ __attribute__((aligned(16))) float tsc[4 * 16 * 36];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int jl = lane >> 2, ql = lane & 3;
    float* ts = tsc + (threadIdx.x >> 6) * 16 * 36;
    PtileIter pt(n_src, blockDim.x >> 6, threadIdx.x >> 6);
    for (; pt.i < pt.end; pt.i += pt.stride) {
        const int g = __builtin_amdgcn_readfirstlane(a.order != nullptr ? a.order[pt.i] : (int)pt.i);
        const long long r0 = __builtin_amdgcn_readfirstlane(seg_rowptr[g]), r1 = __builtin_amdgcn_readfirstlane(seg_rowptr[g + 1]);
        f32x4 sum[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        for (long long base = r0; base < r1; base += 16) {
            asm volatile("" : "+v"(lane));
            const long long pr = base + j;
            const bool valid = pr < r1;
            const long long p = valid ? pr : r1 - 1;
            const long long prl = base + jl;
            const long long pl = prl < r1 ? prl : r1 - 1;
            f32x4 o[2];
            o[0] = *(const f32x4*)(a.c + pl * ROWC + 4 * ql);
            o[1] = *(const f32x4*)(a.c + pl * ROWC + 16 + 4 * ql);
            const float mq = a.mask[p * 4 + q];
            const float eq = q < 3 ? a.edge_attr[p * 3 + q] : 0.f;
            f32x4 n1 = {0.f, 0.f, 0.f, 0.f}, n2 = {0.f, 0.f, 0.f, 0.f};
            {
                const int eb = a.sta_rowptr[pl], ee = a.sta_rowptr[pl + 1];
                gather_sum16<false>(a.wu + 4 * ql, ROWW, a.sta_col, eb, ee, n1);
                o[0] = fma4(n1, 1.f / (float)max(ee - eb, 1), o[0]);
            }
            {
                const int eb = a.src_rowptr[pl], ee = a.src_rowptr[pl + 1];
                gather_sum16<false>(a.wv + 4 * ql, ROWW, a.src_col, eb, ee, n2);
                o[1] = fma4(n2, 1.f / (float)max(ee - eb, 1), o[1]);
            }
            o[0] = prelu4u(o[0], a2);
            o[1] = prelu4u(o[1], a2);
            *(f32x4*)(ts + jl * 36 + 4 * ql) = o[0];
            *(f32x4*)(ts + jl * 36 + 16 + 4 * ql) = o[1];
            GSYNC();
            o[0] = *(const f32x4*)(ts + j * 36 + 4 * q);
            o[1] = *(const f32x4*)(ts + j * 36 + 16 + 4 * q);
            GSYNC();
            float mm = fmaxf(mq, __shfl_xor(mq, 16));
            mm = fmaxf(mm, __shfl_xor(mm, 32));
            mm = valid ? mm : 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 bp = *(const f32x4*)(lbias + t * 16 + 4 * q);
                bp = mma_block(bp, lw[G2_BP(t, 0) * 64 + lane], o[0]);
                bp = mma_block(bp, lw[G2_BP(t, 1) * 64 + lane], o[1]);
                bp = MFMA16(lw[G2_BP(t, 2) * 64 + lane].x, eq, bp);
                sum[t] += prelu4u(bp, ab1) * mm;
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 v = sum[t];
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                v.x += __shfl_xor(v.x, d); v.y += __shfl_xor(v.y, d); v.z += __shfl_xor(v.z, d); v.w += __shfl_xor(v.w, d);
            }
            if (j == 0) *(f32x4*)(a.part + (long long)g * 32 + 16 * t + 4 * q) = v;
        }
    }
}

// k_stage2_fast for the production configuration (uniform-degree graphs, station processing order with a registered static
// edge_attr, static item stream, Bipartite half on), straight-line: the ISA of k_stage2_fast spends a fifth of its vector
// instructions on register copies at the joins of its option branches (the 15 source rows were copied out and back every tile),
// 34 ds_bpermute per tile on the station sum and a dozen uniform branches. Here
//  * the item after the last one is clamped to the last one, so every load of the software pipeline is unconditional and no
//    value has two definitions at a join;
//  * a tile's ids are its (wave-uniform) item number, one src_tab row and the 8 station-neighbour ids, which are loaded into the
//    registers the previous tile's ids have just left: nothing rotates but one register;
//  * the station sum over the 16 nodes of a tile is a DPP row reduction (row_shl:1, 2, 4, 8): lane 0 of every row adds the same
//    operands in the same tree as the xor butterfly of k_stage2 (bitwise identical), without the LDS round trips;
//  * the message mask is read by all four lanes of a node (one address) instead of max-reduced across them.
// Same arithmetic and summation order as k_stage2 / k_stage2_fast (bitwise identical results; tests).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sum16_tree(float v) {      // lane 0 of every row of 16: the butterfly's sum tree
    v = dpp_add<0x101>(v);      // row_shl:1
    v = dpp_add<0x102>(v);
    v = dpp_add<0x104>(v);
    v = dpp_add<0x108>(v);
    return v;
}
// Row-layout loads: in the MFMA layout lane (j = lane & 15, q = lane >> 4) reads the 16-B chunk q of node j's row, so four
// CONSECUTIVE lanes touch four different rows and the texture path works on 16 useful bytes per 64-B request (measured: 24 B per
// clock and CU where contiguous row gathers reach 57). Every row is therefore loaded in the layout lane = 4 r + cq (node r = lane >> 2,
// chunk cq = lane & 3): four consecutive lanes read one 64-B row, sixteen consecutive source rows one contiguous KB. Everything up
// to x_latent is elementwise per (node, channel) and runs in that layout; x_latent, edge_attr and the gated mask then go through a
// 2.3-KB per-wave LDS scratch (rows of 36 floats) into the MFMA layout for fc1.
// XL: also store x_latent [P, 30] (caller's station order). NB: stop after x_latent (no Bipartite message / station sum): the
// last pass of the association heads (genie_assoc_fwd).
// Measured and dropped: other positions of the three load bursts (0.2627 / 0.2650 / 0.2649 ms), waves of a
// workgroup phased half an iteration apart by barriers (0.262 -> 0.290), streamed rows two tiles ahead (246 VGPRs, 0.226 -> 0.240),
// MFMA-layout loads (0.262 vs 0.226), station rows staged in LDS behind a barrier (0.282 -> 0.299 after a cold stage 1).
template <int KS, int KP, bool XL, bool NB = false, bool SAVE = false>      // SAVE: training forward (pre-activations kept)
__global__ __launch_bounds__(256, GENIE_S2_WAVES) void k_stage2_ord(DaArgs a) {
    constexpr int NF4 = (G2_GROUPS * 256 + G2_BIAS * 16 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    __shared__ __attribute__((aligned(16))) float tsc[4 * 16 * 36];
    for (int i = threadIdx.x; i < NF4; i += blockDim.x) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lbias = (const float*)(lw + G2_GROUPS * 64);
    const float* lscal = lbias + G2_BIAS * 16;
    const float a2 = a.slope2 != nullptr ? *a.slope2 : lscal[0], ab1 = lscal[1];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int jl = lane >> 2, ql = lane & 3;      // (node, chunk) this lane LOADS
    float* ts = tsc + wave * 16 * 36;
    const int S = a.S;
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave, a.gi0);
    // a.wgmap: a workgroup takes BLOCKS of 4 consecutive source nodes of its XCD's chunk, wave k sweeps the tiles of the k-th
    // node of the block: the four waves of a CU then read the source-neighbour rows of four adjacent source nodes (half of
    // them shared) for the same station tile at about the same time, and every station row wu[g] is gathered on one CU only
    if (a.wgmap) {
        const int nx = (a.nxcd > 1 && gridDim.x >= (unsigned)a.nxcd && (gridDim.x % a.nxcd) == 0) ? a.nxcd : 1;
        const int lb = blockIdx.x / nx, nbx = gridDim.x / nx, n = w.gend - w.gbeg, n_blk = (n + 3) / 4;
        int n_my = lb < n_blk ? (n_blk - lb + nbx - 1) / nbx : 0;
        if (n_my > 0 && 4 * (lb + (n_my - 1) * nbx) + wave >= n) --n_my;
        w.it = 0; w.stride = 1; w.nitems = (long long)n_my * a.T;
        w.lead_ = lb; w.chunk_ = nbx;                     // (reused as: first block, block stride)
    }
    if (w.it >= w.nitems) return;
    const char* wub = (const char*)a.wu;
    const char* wvb = (const char*)a.wv;
    const unsigned q16 = 16u * (unsigned)ql;
    const size_t gpitch = (size_t)S * 64u;                 // bytes of one source node's rows in wu / wv
    const unsigned m_T = ItemIter::recip((unsigned)a.T);

    // item -> wave-uniform (processing position gi, station tile tb); every XCD's chunk is swept BACKWARDS: the c / wu / wv rows
    // stage 1 wrote last (still in the Infinity Cache) are read first (0.278 -> 0.276 ms)
    auto item_of = [&](long long it, int& gi, int& tb) {
        const long long itr = w.nitems - 1 - it;
        if (a.wgmap) {
            unsigned rem;
            const unsigned kb = a.T <= 1 ? (rem = 0u, (unsigned)itr) : ItemIter::fdiv((unsigned)itr, (unsigned)a.T, m_T, rem);
            tb = (int)rem;
            gi = w.gbeg + 4 * (w.lead_ + (int)kb * w.chunk_) + wave;
        } else {
            w.decode(itr, gi, tb);
        }
        gi = __builtin_amdgcn_readfirstlane(gi);
        tb = __builtin_amdgcn_readfirstlane(tb);
    };
    struct Stream { f32x4 o[2]; float mq, eq; };                  // streamed rows of a tile: c, message mask, edge_attr
    struct Rows { f32x4 ru[KS], rv[KP]; } rows;                    // gathered rows
    Stream sA;
    int sta[KS];
    auto load_ids = [&](int gi, int tb, int& idv) {
        idv = a.src_tab[gi * 16 + j];
        const int s = tb * 16 + jl;
        load_sta_ids<KS>(a.sta_col, s < S ? s : S - 1, sta);
    };
    auto issue0 = [&](Stream& st, int idv, int tb) {
        const int g = __builtin_amdgcn_readlane(idv, 0);
        const int s = tb * 16 + jl, sc = s < S ? s : S - 1;
        long long p = (long long)g * S + sc;
        if (ABL(a, 9)) p &= 4095;          // tuning: streamed rows from a cache-resident region
        st.o[0] = *(const f32x4*)(a.c + p * ROWC + 4 * ql);
        st.o[1] = *(const f32x4*)(a.c + p * ROWC + 16 + 4 * ql);
        st.mq = NB ? 0.f : a.mm_int[p];
        st.eq = (!NB && ql < 3) ? a.ea_int[p * 3 + ql] : 0.f;
        const char* wug = wub + (ABL(a, 11) ? (size_t)0 : (size_t)g * gpitch);     // tuning bit 11: gathers hit one resident block
#pragma unroll
        for (int k = 0; k < KS; ++k) rows.ru[k] = ABL(a, 0) ? st.o[0] : *(const f32x4*)(wug + ((unsigned)sta[k] * 64u + q16));
    };
    auto issue_v = [&](int idv, int tb, int k0, int k1) {
        const int s = tb * 16 + jl, sc = s < S ? s : S - 1;
        const unsigned so = (unsigned)sc * 64u + q16;
#pragma unroll
        for (int k = 0; k < KP; ++k)
            if (k >= k0 && k < k1) {
                // the row base of a source neighbour is wave-uniform: kept opaque in an SGPR pair, so that the load is
                // `global_load v, voffset, s[base]` (left to itself hipcc hoists wvb + so into a VGPR pair and adds the
                // scalar part with a 64-bit vector multiply-add per neighbour: 3 vector instructions each)
                unsigned long long wvk = (unsigned long long)wvb + (ABL(a, 11) ? (size_t)k : (size_t)__builtin_amdgcn_readlane(idv, 1 + k)) * gpitch;
                asm volatile("" : "+s"(wvk));
                typedef const __attribute__((address_space(1))) char* gbytes;
                typedef const __attribute__((address_space(1))) f32x4* grow;
                rows.rv[k] = ABL(a, 1) ? rows.ru[0] : *(grow)((gbytes)wvk + so);
            }
    };
    constexpr int KH = (KP + 1) / 2;

    long long it = w.it;
    int gi_c, tb_c, gi_n, tb_n, idv_c, idv_n;
    item_of(it, gi_c, tb_c);
    load_ids(gi_c, tb_c, idv_c);
    issue0(sA, idv_c, tb_c);
    issue_v(idv_c, tb_c, 0, KP);
    {
        const long long itn = it + w.stride < w.nitems ? it + w.stride : it;
        item_of(itn, gi_n, tb_n);
        load_ids(gi_n, tb_n, idv_n);
    }
    for (;;) {
        asm volatile("" : "+v"(lane));
        const int g_c = __builtin_amdgcn_readlane(idv_c, 0);
        const bool has_next = it + w.stride < w.nitems;
        const long long it2 = it + 2 * w.stride < w.nitems ? it + 2 * w.stride : (has_next ? it + w.stride : it);
        int gi_2, tb_2, idv_2;
        item_of(it2, gi_2, tb_2);
        // (1) consume the rows of this tile: neighbour means of the projected operands in edge order, PReLU2 -> x_latent
        f32x4 n1 = {0.f, 0.f, 0.f, 0.f}, n2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KS; ++k) n1 += rows.ru[k];
#pragma unroll
        for (int k = 0; k < KP; ++k) n2 += rows.rv[k];
        f32x4 o[2];
        o[0] = fma4(n1, 1.f / (float)KS, sA.o[0]);
        o[1] = fma4(n2, 1.f / (float)KP, sA.o[1]);
        float mq = sA.mq, eq = sA.eq;
        const int s_l = tb_c * 16 + jl;              // the node this lane loaded (not the node it holds in the MFMA layout)
        const bool valid_l = s_l < S;
        if (SAVE && valid_l) {       // at the caller's product-node index, as 16-float blocks [o1 | o2]
            const size_t pu = (size_t)g_c * S + a.sta_user[s_l];
            *(f32x4*)(a.save + ((size_t)(SV_O + 0) * a.Pn + pu) * 16 + 4 * ql) = o[0];
            *(f32x4*)(a.save + ((size_t)(SV_O + 1) * a.Pn + pu) * 16 + 4 * ql) = o[1];
        }
        o[0] = prelu4u(o[0], a2);
        o[1] = prelu4u(o[1], a2);
        const f32x4 ol0 = o[0], ol1 = o[1];
        if (!NB) {      // row layout -> MFMA layout through the wave's LDS scratch: node r's row = [o1 (16) | o2 (16) | edge_attr (3) | gated mask]
            *(f32x4*)(ts + jl * 36 + 4 * ql) = o[0];
            *(f32x4*)(ts + jl * 36 + 16 + 4 * ql) = o[1];
            ts[jl * 36 + 32 + ql] = ql < 3 ? eq : (valid_l ? mq : 0.f);
            GSYNC();
            o[0] = *(const f32x4*)(ts + j * 36 + 4 * q);
            o[1] = *(const f32x4*)(ts + j * 36 + 16 + 4 * q);
            eq = q < 3 ? ts[j * 36 + 32 + q] : 0.f;
            mq = ts[j * 36 + 35];
        }
        asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(idv_n));
        // (2) first burst of the next tile's rows (the station-neighbour ids are dead after it)
        issue0(sA, idv_n, tb_n);
        if (NB) issue_v(idv_n, tb_n, 0, KP);
        if (XL && valid_l) {
            const int su = a.sta_user[s_l];
            float* xl = a.x_latent + ((long long)g_c * S + su) * 30;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * ql + r < 15) { xl[4 * ql + r] = ol0[r]; xl[15 + 4 * ql + r] = ol1[r]; }
        }
        f32x4 bp[2];
        bp[0] = *(const f32x4*)(lbias + 0 * 16 + 4 * q);
        bp[1] = *(const f32x4*)(lbias + 1 * 16 + 4 * q);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (NB) break;
            if (!ABL(a, 6)) {      // (tuning bit 6: no fc1 MFMAs)
                bp[t] = mma_block(bp[t], lw[G2_BP(t, 0) * 64 + lane], o[0]);
                bp[t] = mma_block(bp[t], lw[G2_BP(t, 1) * 64 + lane], o[1]);
                bp[t] = MFMA16(lw[G2_BP(t, 2) * 64 + lane].x, eq, bp[t]);
            } else bp[t] += o[0] + o[1] + eq;
            if (SAVE && tb_c * 16 + j < S)
                *(f32x4*)(a.save + ((size_t)(SV_ZB + t) * a.Pn + (size_t)g_c * S + a.sta_user[tb_c * 16 + j]) * 16 + 4 * q) = bp[t];
            bp[t] = prelu4u(bp[t], ab1);
            // (3) second / third burst, behind the first / second output tile of fc1
            asm volatile("" : "+v"(bp[t]), "+v"(idv_n));
            if (t == 0) issue_v(idv_n, tb_n, 0, KH); else issue_v(idv_n, tb_n, KH, KP);
        }
        // (4) ids of the tile after next (the item after the last one repeats the last one: its loads are never consumed)
        load_ids(gi_2, tb_2, idv_2);
        // (5) mask gate and station sum of this tile
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (NB) break;
            f32x4 v = bp[t] * mq;
            v.x = row_sum16_tree(v.x); v.y = row_sum16_tree(v.y); v.z = row_sum16_tree(v.z); v.w = row_sum16_tree(v.w);
            if (j == 0) *(f32x4*)(a.part + ((long long)g_c * a.T + tb_c) * 32 + 16 * t + 4 * q) = v;
        }
        if (!has_next) break;
        it += w.stride;
        idv_c = idv_n; tb_c = tb_n;
        idv_n = idv_2; tb_n = tb_2;
    }
}

// ------------------------------------------------------------------------------------------------
// STAGE 2 on the 16-bit matrix pipe (k_stage2_h2, round 4): the production stage 2 of the reference's kNN graphs.
//
// What k_stage2_ord spent its time on (profiles/r03_zz_pmc_stage_kernels.txt, r03_y_s2_ablations.txt): 18 fp32 MFMAs of 32 cycles
// per 16 nodes that do not overlap with vector work, a round trip through LDS per tile (row layout -> MFMA layout) in the middle
// of the dependency chain, 64-bit scalar address arithmetic per gathered row (as many scalar as vector instructions), and 47 % of
// all wave cycles waiting. Here
//  * Bipartite fc1 (33 -> 30) runs as v_mfma_f32_16x16x32_f16 with fp32 operands as two fp16 pieces (the f16x2 form of stage 1:
//    W0 x1 + W1' (x0 / 16) + W0 x0): D[channel, node] for 16 channels x 16 nodes, K = 32 = [o1 chunk | o2 chunk] x 4 lane groups:
//    ONE K-step for all of x_latent, so a tile takes 2 x 3 MFMAs of 16 cycles + 2 for edge_attr instead of 18 x 32 cycles;
//  * lane (m = lane & 15, kg = lane >> 4) holds channels 4 kg .. 4 kg + 3 of BOTH halves of node m's x_latent, which is the
//    MFMA's B operand as it stands: no LDS transpose. For that the rows stage 1 writes are NODE-PLANAR: inside the block of a
//    source node, chunk q (16 B) of all S stations is contiguous (c: [g][8][S] x 16 B, wv: [g][4][S] x 16 B; DaArgs.np), so the 16
//    lanes of a lane group read 256 contiguous bytes and the block of a source node stays contiguous (the halo exchange of the
//    sharded path moves whole blocks, genie_amd/dist.py);
//  * the station-neighbour rows (wu, row layout [p][16]) are still gathered four lanes to a 64-B row (the texture path's fast
//    pattern), summed there, and the SUM crosses into the operand layout with four ds_bpermute_b32;
//  * the static edge_attr arrives as a ready-made B fragment (k_ea_frag, written once per registered edge_attr): its K-step is one
//    MFMA per channel block;
//  * row bases are 32-bit scalar products on top of a 64-bit pointer (BIG: 64-bit products, config 4 on one GPU).
// Same arithmetic up to x_latent as k_stage2_ord (bitwise equal x_latent); the Bipartite message is fp32-class like stage 1
// (products exact in the fp32 accumulator, operands within one fp32 ulp), tests compare it with the oracle and the fp32 kernels.
// ------------------------------------------------------------------------------------------------
#define MFMA16H(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, (a)), __builtin_bit_cast(f16x8, (b)), (c), 0, 0, 0)

// edge_attr [P, 3] (caller's station order) -> B fragments of the edge_attr K-step, node-planar [g][2][S] x 16 B in station
// processing order: lane group 0 = {e0, e1, e2, 0 (first pieces) | e0, e1, e2, 0 (second pieces)}, group 1 = {e / 16 (3), 0 | 0}
__global__ void k_ea_frag(const float* __restrict__ ea, long long rows, int S, const int32_t* __restrict__ sta_user,
                          unsigned* __restrict__ out) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= rows) return;
    const long long g = p / S;
    const int s = (int)(p - g * S);
    const long long pu = g * S + (sta_user != nullptr ? sta_user[s] : s);
    const float e0 = ea[pu * 3], e1 = ea[pu * 3 + 1], e2 = ea[pu * 3 + 2];
    const unsigned a0 = cvt_pk_f16(e0, e1), b0 = cvt_pk_f16(e2, 0.f);
    const unsigned a1 = cvt_pk_f16(sub_f16_lo(e0, a0), sub_f16_hi(e1, a0)), b1 = cvt_pk_f16(sub_f16_lo(e2, b0), 0.f);
    *(u32x4*)(out + ((g * 2) * S + s) * 4) = u32x4{a0, b0, a1, b1};
    *(u32x4*)(out + ((g * 2 + 1) * S + s) * 4) = u32x4{pk_mul_f16(a0, H2_SIXTEENTH), pk_mul_f16(b0, H2_SIXTEENTH), 0u, 0u};
}

template <bool BIG>
__device__ __forceinline__ unsigned long long s2h_base(const void* b, int id, unsigned pitch) {
    const unsigned long long off = BIG ? (unsigned long long)(unsigned)id * (unsigned long long)pitch
                                       : (unsigned long long)((unsigned)id * pitch);
    unsigned long long r = (unsigned long long)b + off;
    asm volatile("" : "+s"(r));      // stays an SGPR pair: the load is `global_load v, voffset, s[base]`
    return r;
}

template <bool XL, bool BIG>
__global__ __launch_bounds__(256, GENIE_S2H_WAVES) void k_stage2_h2(DaArgs a) {
    constexpr int KS = 8, KP = 15;
    constexpr int NF4 = S2H_IMG_FLOATS / 4;
    __shared__ f32x4 lw[NF4];
    for (int i = threadIdx.x; i < NF4; i += blockDim.x) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lbias = (const float*)(lw + S2H_FRAGS * 64);
    const float a2 = a.slope2 != nullptr ? *a.slope2 : lbias[32], ab1 = lbias[33];
    int lane = threadIdx.x & 63;
    const int m = lane & 15, kg = lane >> 4;      // operand layout: node m of the tile, K-slot group kg
    const int jl = lane >> 2, ql = lane & 3;      // row layout of the station-neighbour gathers: node jl, 16-B chunk ql
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = a.S;
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave, a.gi0);
    if (a.wgmap) {     // blocks of 4 adjacent source nodes per workgroup, one node per wave (see k_stage2_ord)
        const int nx = (a.nxcd > 1 && gridDim.x >= (unsigned)a.nxcd && (gridDim.x % a.nxcd) == 0) ? a.nxcd : 1;
        const int lb = blockIdx.x / nx, nbx = gridDim.x / nx, n = w.gend - w.gbeg, n_blk = (n + 3) / 4;
        int n_my = lb < n_blk ? (n_blk - lb + nbx - 1) / nbx : 0;
        if (n_my > 0 && 4 * (lb + (n_my - 1) * nbx) + wave >= n) --n_my;
        w.it = 0; w.stride = 1; w.nitems = (long long)n_my * a.T;
        w.lead_ = lb; w.chunk_ = nbx;
    }
    if (w.it >= w.nitems) return;
    const unsigned m_T = ItemIter::recip((unsigned)a.T);
    auto item_of = [&](long long it, int& gi, int& tb) {      // every XCD's chunk is swept backwards (the rows stage 1 wrote last first)
        const long long itr = w.nitems - 1 - it;
        if (a.wgmap) {
            unsigned rem;
            const unsigned kb = a.T <= 1 ? (rem = 0u, (unsigned)itr) : ItemIter::fdiv((unsigned)itr, (unsigned)a.T, m_T, rem);
            tb = (int)rem;
            gi = w.gbeg + 4 * (w.lead_ + (int)kb * w.chunk_) + wave;
        } else {
            w.decode(itr, gi, tb);
        }
        gi = __builtin_amdgcn_readfirstlane(gi);
        tb = __builtin_amdgcn_readfirstlane(tb);
    };
    typedef const __attribute__((address_space(1))) char* gbytes;
    typedef const __attribute__((address_space(1))) f32x4* grow;
    typedef const __attribute__((address_space(1))) u32x4* gfrag;
    typedef const __attribute__((address_space(1))) float* gflt;
    const unsigned plane = (unsigned)S * 16u;          // bytes of one chunk plane inside a source node's block
    const unsigned pc = (unsigned)S * 128u, pw = (unsigned)S * 64u, pe = (unsigned)S * 32u, pm = (unsigned)S * 4u;
    const unsigned kgp = (unsigned)kg * plane, kge = (unsigned)min(kg, 1) * plane;
    const unsigned q16 = 16u * (unsigned)ql;
    const int bperm = (4 * m + kg) * 4;                // this lane's operand = the row-layout lane of (node m, chunk kg)

    struct Tile { f32x4 ru[KS], rv[KP], c1, c2; u32x4 ea; float mq; } R;
    int sta[KS];
    auto load_ids = [&](int gi, int tb, int& idv) {
        idv = a.src_tab[gi * 16 + m];
        const int s = tb * 16 + jl;
        load_sta_ids<KS>(a.sta_col, s < S ? s : S - 1, sta);
    };
    auto issue = [&](int idv, int tb) {
        const int g = __builtin_amdgcn_readlane(idv, 0);
        const int s = tb * 16 + m, sc = s < S ? s : S - 1;
        const unsigned so = (unsigned)sc * 16u;
        const unsigned lo = kgp + so;
        const int gs = ABL(a, 9) ? (g & 7) : g;        // tuning bit 9: streamed rows (c, mask, edge_attr) from a cache-resident region
        const unsigned long long cb = s2h_base<BIG>(a.c, gs, pc);
        R.c1 = *(grow)((gbytes)cb + lo);
        R.c2 = *(grow)((gbytes)cb + (lo + 4u * plane));
        const unsigned long long mb = s2h_base<BIG>(a.mm_int, gs, pm);
        R.mq = *(gflt)((gbytes)mb + (unsigned)sc * 4u);
        if (s >= S) R.mq = 0.f;
        const unsigned long long eb = s2h_base<BIG>(a.ea_frag, gs, pe);
        R.ea = *(gfrag)((gbytes)eb + (kge + so));
        const unsigned long long ub = s2h_base<BIG>(a.wu, ABL(a, 11) ? (g & 7) : g, pw);
#pragma unroll
        for (int k = 0; k < KS; ++k) R.ru[k] = ABL(a, 0) ? R.c1 : *(grow)((gbytes)ub + ((unsigned)sta[k] * 64u + q16));
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            // tuning bit 11: every gather from a cache-resident block (same instruction stream); bit 12: the 15 source rows of a tile
            // are ONE row (the L1 misses of the source gathers collapse to those of one row); bit 1: no source gathers
            const int idk = ABL(a, 11) ? k : __builtin_amdgcn_readlane(idv, ABL(a, 12) ? 1 : 1 + k);
            const unsigned long long vb = s2h_base<BIG>(a.wv, idk, pw);
            R.rv[k] = ABL(a, 1) ? R.c2 : *(grow)((gbytes)vb + lo);
        }
    };

    long long it = w.it;
    int gi_c, tb_c, gi_n, tb_n, idv_c, idv_n;
    item_of(it, gi_c, tb_c);
    load_ids(gi_c, tb_c, idv_c);
    issue(idv_c, tb_c);
    {
        const long long itn = it + w.stride < w.nitems ? it + w.stride : it;
        item_of(itn, gi_n, tb_n);
        load_ids(gi_n, tb_n, idv_n);
    }
    for (;;) {
        asm volatile("" : "+v"(lane));
        const int g_c = __builtin_amdgcn_readlane(idv_c, 0);
        const bool has_next = it + w.stride < w.nitems;
        const long long it2 = it + 2 * w.stride < w.nitems ? it + 2 * w.stride : (has_next ? it + w.stride : it);
        int gi_2, tb_2, idv_2;
        item_of(it2, gi_2, tb_2);
        // (1) neighbour means of the projected operands in edge order (as k_stage2_ord), PReLU2 -> x_latent
        f32x4 n1 = {0.f, 0.f, 0.f, 0.f}, n2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KS; ++k) n1 += R.ru[k];
#pragma unroll
        for (int k = 0; k < KP; ++k) n2 += R.rv[k];
        f32x4 n1t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {     // (through a scalar temporary: __builtin_bit_cast applied to a vector ELEMENT reads element 0, hipcc 7.0)
            const float v = n1[r];
            n1t[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(bperm, __float_as_int(v)));
        }
        f32x4 o1 = fma4(n1t, 1.f / (float)KS, R.c1), o2 = fma4(n2, 1.f / (float)KP, R.c2);
        const float mq = R.mq;
        const u32x4 eab = R.ea;
        o1 = prelu4u(o1, a2);
        o2 = prelu4u(o2, a2);
        asm volatile("" : "+v"(o1), "+v"(o2), "+v"(idv_n));
        // (2) every row of the next tile (the item after the last repeats the last one: its rows are never consumed)
        issue(idv_n, tb_n);
        if (XL && tb_c * 16 + m < S) {
            const int su = a.sta_user[tb_c * 16 + m];
            float* xl = a.x_latent + ((long long)g_c * S + su) * 30;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * kg + r < 15) { xl[4 * kg + r] = o1[r]; xl[15 + 4 * kg + r] = o2[r]; }
        }
        // (3) B operand: K slots (kg, e) = o1 channels 4 kg + e (e < 4), o2 channels 4 kg + e - 4; pieces {x0, x1, x0 / 16}
        u32x4 p0, p1, p2;
        p0[0] = cvt_pk_f16(o1[0], o1[1]); p0[1] = cvt_pk_f16(o1[2], o1[3]);
        p0[2] = cvt_pk_f16(o2[0], o2[1]); p0[3] = cvt_pk_f16(o2[2], o2[3]);
        p1[0] = cvt_pk_f16(sub_f16_lo(o1[0], p0[0]), sub_f16_hi(o1[1], p0[0]));
        p1[1] = cvt_pk_f16(sub_f16_lo(o1[2], p0[1]), sub_f16_hi(o1[3], p0[1]));
        p1[2] = cvt_pk_f16(sub_f16_lo(o2[0], p0[2]), sub_f16_hi(o2[1], p0[2]));
        p1[3] = cvt_pk_f16(sub_f16_lo(o2[2], p0[3]), sub_f16_hi(o2[3], p0[3]));
#pragma unroll
        for (int d = 0; d < 4; ++d) p2[d] = pk_mul_f16(p0[d], H2_SIXTEENTH);
        // (4) Bipartite fc1: D[channel 16 t + 4 kg + r, node m], smallest products first
        f32x4 bp[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bp[t] = *(const f32x4*)(lbias + 16 * t + 4 * kg);
            const f32x4 w0 = lw[(S2H_FW + 2 * t) * 64 + lane], w1 = lw[(S2H_FW + 2 * t + 1) * 64 + lane];
            const f32x4 we = lw[(S2H_FE + t) * 64 + lane];
            bp[t] = MFMA16H(w0, p1, bp[t]);
            bp[t] = MFMA16H(w1, p2, bp[t]);
            bp[t] = MFMA16H(we, eab, bp[t]);
            bp[t] = MFMA16H(w0, p0, bp[t]);
        }
        // (5) ids of the tile after next
        load_ids(gi_2, tb_2, idv_2);
        // (6) PReLU, mask gate, station sum of this tile (DPP row reduction in the butterfly's order), one partial row per tile
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 v = prelu4u(bp[t], ab1) * mq;
            v.x = row_sum16_tree(v.x); v.y = row_sum16_tree(v.y); v.z = row_sum16_tree(v.z); v.w = row_sum16_tree(v.w);
            if (m == 0) *(f32x4*)(a.part + ((long long)g_c * a.T + tb_c) * 32 + 16 * t + 4 * kg) = v;
        }
        if (!has_next) break;
        it += w.stride;
        idv_c = idv_n; tb_c = tb_n;
        idv_n = idv_2; tb_n = tb_2;
    }
}

// ------------------------------------------------------------------------------------------------
// fp16 range guard of the f16x2 kernels (k_stage1_h2, k_stage2_h2). They split every hidden state into fp16 pieces, so a value
// above 65 504 would turn into inf where the reference's fp32 arithmetic is still fine. This kernel computes, from the weights
// alone, a RIGOROUS bound of every such value (interval arithmetic per channel: |W x + b| <= sum_j |W_ij| X_j + |b_i|,
// |PReLU_a(z)| <= max(1, |a|) |z|, |mean| <= max) for inputs Slice, Mask in [-1, 1] (process_utils.py:262-275: exp(-r^2 / 2 s^2) in
// [0, 1], or +-1 with the sign input; Mask in {0, 1}) and the actual maxima of the static tables (absolute positions, edge-term
// biases). out = {largest bound, largest weight magnitude in its fp16 form (init_trns enters 16 x), ok flag, 0}. The context
// selects the fp32-MFMA kernels when ok == 0 (genie_ctx::range_ok; genie_set_stage_precision overrides).
// ------------------------------------------------------------------------------------------------
enum { RG_INIT_W, RG_INIT_B, RG_INIT_ABS, RG_L1T12_W, RG_L1T12_B, RG_L1T22_W, RG_L1T22_B, RG_L2T11_W, RG_L2T11_B, RG_L2T21_W,
       RG_L2T21_B, RG_L2T12_W, RG_L2T12_B, RG_L2T22_W, RG_L2T22_B, RG_ACT, RG_ACT11, RG_ACT12, RG_ACT1, RG_ACT21, RG_ACT22, RG_ACT2,
       RG_FC1_W, RG_N };
struct RangeArgs {
    const float* raw;
    int off[RG_N];
    const float *abs_sta, *abs_src, *eb_sta, *eb_src;
    long long n_abs_sta, n_abs_src, n_eb_sta, n_eb_src;
    float* out;
};
constexpr float H2_RANGE_LIMIT = 60000.f;

__device__ __forceinline__ float blk_max256(float v, float* red) {
    const int t = threadIdx.x;
    red[t] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) red[t] = fmaxf(red[t], red[t + s]);     // (NaN-propagating where it matters: the flag test below is !(x < limit))
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ float tab_max256(const float* p, long long n, float* red) {
    float v = 0.f;
    bool bad = false;
    for (long long i = threadIdx.x; i < n; i += 256) { const float x = fabsf(p[i]); bad |= !(x <= 3.0e38f); v = fmaxf(v, x); }
    return blk_max256(bad ? __builtin_inff() : v, red);
}

// |.|-maximum of a long table in RG_PART partial maxima (one workgroup each; a non-finite entry gives +inf): k_h2_range is ONE workgroup and
// would walk the [n_grid, 48] edge-feature table alone (0.5 ms per weight update at 10 000 source nodes)
constexpr int RG_PART = 64;
__global__ __launch_bounds__(256) void k_tab_absmax(const float* __restrict__ p, long long n, float* __restrict__ out) {
    __shared__ float red[256];
    float v = 0.f;
    bool bad = false;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float x = fabsf(p[i]);
        bad |= !(x <= 3.0e38f);
        v = fmaxf(v, x);
    }
    const float r = blk_max256(bad ? __builtin_inff() : v, red);
    if (threadIdx.x == 0) out[blockIdx.x] = r;
}

__global__ __launch_bounds__(256) void k_h2_range(RangeArgs a) {
    __shared__ float red[256];
    __shared__ float H0[32], N1[32], N2[32], H1a[32], H1b[32], U[32], V[32];
    const int t = threadIdx.x;
    const float* R = a.raw;
    auto W = [&](int id) { return R + a.off[id]; };
    auto g1 = [](float s) { return fmaxf(1.f, fabsf(s)); };
    const float pos_sta = a.abs_sta ? tab_max256(a.abs_sta, a.n_abs_sta, red) : 0.f;
    const float pos_src = a.abs_src ? tab_max256(a.abs_src, a.n_abs_src, red) : 0.f;
    const float ebs = a.eb_sta ? tab_max256(a.eb_sta, a.n_eb_sta, red) : 0.f;
    const float ebg = a.eb_src ? tab_max256(a.eb_src, a.n_eb_src, red) : 0.f;
    const float a0 = *W(RG_ACT), a1 = *W(RG_ACT1), a21 = *W(RG_ACT21), a22 = *W(RG_ACT22), a2 = *W(RG_ACT2);
    const float s11 = a0 >= 0.f ? a0 * *W(RG_ACT11) : a0, s12 = a0 >= 0.f ? a0 * *W(RG_ACT12) : a0;      // compose_slopes
    float wmax = 0.f, amax = 0.f;
    bool bad = false;
    // largest weight magnitude in the form that is rounded to fp16 (the input layer enters as 16 W)
    {
        const int ids[8] = {RG_INIT_W, RG_INIT_ABS, RG_L1T12_W, RG_L1T22_W, RG_L2T11_W, RG_L2T21_W, RG_L2T12_W, RG_L2T22_W};
        const int nel[8] = {240, 180, 1920, 1920, 1800, 1800, 1410, 1410};
        for (int k = 0; k < 8; ++k)
            for (int i = t; i < nel[k]; i += 256) {
                const float x = fabsf(W(ids[k])[i]) * (k < 2 ? 16.f : 1.f);
                bad |= !(x <= 3.0e38f);
                wmax = fmaxf(wmax, x);
            }
        for (int i = t; i < 990; i += 256) { const float x = fabsf(W(RG_FC1_W)[i]); bad |= !(x <= 3.0e38f); wmax = fmaxf(wmax, x); }
    }
    // h0 and the two neighbour means of its re-activated forms
    if (t < 30) {
        float z = fabsf(W(RG_INIT_B)[t]);
        for (int j = 0; j < 8; ++j) z += fabsf(W(RG_INIT_W)[t * 8 + j]);
        for (int j = 0; j < 3; ++j) z += fabsf(W(RG_INIT_ABS)[t * 6 + j]) * pos_sta + fabsf(W(RG_INIT_ABS)[t * 6 + 3 + j]) * pos_src;
        H0[t] = g1(a0) * z; N1[t] = g1(s11) * z; N2[t] = g1(s12) * z;
        amax = fmaxf(fmaxf(amax, fmaxf(pos_sta, pos_src)), fmaxf(H0[t], fmaxf(N1[t], N2[t])));   // (the positions are fp16 operands too)
    }
    __syncthreads();
    // h1 = [PReLU1(l1_t1_2 [h0 | n1 | M]) | PReLU1(l1_t2_2 [h0 | n2 | M])]
    if (t < 60) {
        const int i = t % 30, hf = t / 30;
        const float* w = W(hf ? RG_L1T22_W : RG_L1T12_W) + i * 64;
        const float* n = hf ? N2 : N1;
        float z = fabsf(W(hf ? RG_L1T22_B : RG_L1T12_B)[i]) + (hf ? ebg : ebs);
        for (int j = 0; j < 30; ++j) z += fabsf(w[j]) * H0[j] + fabsf(w[30 + j]) * n[j];
        for (int j = 0; j < 4; ++j) z += fabsf(w[60 + j]);
        (hf ? H1b : H1a)[i] = g1(a1) * z;
        amax = fmaxf(amax, g1(a1) * z);
    }
    __syncthreads();
    if (t < 60) {       // u, v
        const int i = t % 30, hf = t / 30;
        const float* w = W(hf ? RG_L2T21_W : RG_L2T11_W) + i * 60;
        float z = fabsf(W(hf ? RG_L2T21_B : RG_L2T11_B)[i]);
        for (int j = 0; j < 30; ++j) z += fabsf(w[j]) * H1a[j] + fabsf(w[30 + j]) * H1b[j];
        const float s = hf ? a22 : a21;
        (hf ? V : U)[i] = g1(s) * z;
        amax = fmaxf(amax, g1(s) * z);
    }
    __syncthreads();
    if (t < 30) {       // x_latent = PReLU2(c + mean of the projected u / v), the B operand of k_stage2_h2
        const int i = t % 15, hf = t / 15;
        const float* w = W(hf ? RG_L2T22_W : RG_L2T12_W) + i * 94;
        const float* uv = hf ? V : U;
        float z = fabsf(W(hf ? RG_L2T22_B : RG_L2T12_B)[i]) + (hf ? ebg : ebs);
        for (int j = 0; j < 30; ++j) z += fabsf(w[j]) * H1a[j] + fabsf(w[30 + j]) * H1b[j] + fabsf(w[60 + j]) * uv[j];
        for (int j = 0; j < 4; ++j) z += fabsf(w[90 + j]);
        amax = fmaxf(amax, g1(a2) * z);
    }
    bad |= !(amax <= 3.0e38f);
    const float am = blk_max256(bad ? __builtin_inff() : amax, red);
    const float wm = blk_max256(bad ? __builtin_inff() : wmax, red);
    if (t == 0) {
        a.out[0] = am; a.out[1] = wm;
        a.out[2] = (am < H2_RANGE_LIMIT && wm < H2_RANGE_LIMIT) ? 1.f : 0.f;
        a.out[3] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// k_stage2_h2u: k_stage2_h2 with the source-neighbour rows of BLOCKS of adjacent source nodes staged once in LDS (round 4).
//
// What bounds k_stage2_h2 is the vector-memory path: 27 KB per tile through the texture addresser and
// 170 L1 misses per tile, 120 of them the 15 source-neighbour rows, which are never L1 hits: the ~15 users of a row (g', tile) are
// the tiles (g, same station tile) of the source nodes g that list g', and they run on other CUs. Adjacent source nodes of the
// space-filling-curve order share about half of their neighbours, so a workgroup takes a BLOCK of up to 8 consecutive source nodes
// for ONE station tile: the union of their neighbour rows (<= S2U_UCAP, ~45 of 120) is copied into LDS once (each wave fetches a
// quarter of the rows into registers while the previous block-tile computes), and every node of the block sums its 15 rows from
// LDS (lane-contiguous 16-B reads, no bank conflicts). Texture traffic of the source rows 15 -> ~5.6 KB per tile, L1 misses 120 -> ~45.
// Blocks and their union lists are static (genie_ctx_create: build_union_blocks); two workgroup barriers per block-tile (8 tiles).
// Everything per node (streamed rows, station-neighbour gathers, f16x2 fc1, station sum) is k_stage2_h2's code: same results bit for
// bit (the row sums keep the edge order).
// ------------------------------------------------------------------------------------------------
constexpr int S2U_WPB = 4;           // waves per workgroup (round 5: 8 waves = blocks of 16 source nodes on the same 64-row union, twice the resident
                                     // waves: 0.264 against 0.198 ms -- the 64-row cap cuts such blocks at 11-13 nodes and leaves a quarter of the slots empty)
constexpr int S2U_NB = 2 * S2U_WPB;  // source nodes per block (two per wave)
#define GENIE_S2U_BPC 2              // workgroups per CU
#define GENIE_S2U_UCAP 64
constexpr int S2U_UCAP = GENIE_S2U_UCAP;   // distinct neighbour rows of a block (a block is cut short where the union would exceed it), <= 64.
                                           // (Round 5: 32 rows at 4 workgroups per CU, 48 at 3 -- more resident waves, less sharing and blocks cut short
                                           // with empty node slots -- 0.711 and 0.374 ms against 0.202: the staging is what this kernel lives on.)
constexpr int S2U_NSTG = S2U_UCAP / S2U_WPB;
struct S2uBlock {                    // one block of the processing order
    int32_t gi0, n, U, pad;          // first position, source nodes (1 .. 8), union size
    int32_t ids[64];                 // source node of union row u (padded with row 0); one per lane of the wave that stages them
    int32_t idx[S2U_NB][16];         // node b: [0] = its source node id (-1: the block has no node b), [1 + k] = union row of its k-th neighbour
};

template <bool XL, bool BIG, bool SAVE = false>      // SAVE: training forward (pre-activations of x_latent and of the Bipartite message kept)
__global__ __launch_bounds__(S2U_WPB * 64, GENIE_S2U_BPC) void k_stage2_h2u(DaArgs a, const S2uBlock* __restrict__ blocks, const int32_t* __restrict__ xcd_blk0) {
    constexpr int KS = 8, KP = 15;
    constexpr int NF4 = S2H_IMG_FLOATS / 4;
    extern __shared__ __attribute__((aligned(16))) f32x4 s2u_smem[];
    f32x4* lw = s2u_smem;
    char* lrows = (char*)(s2u_smem + NF4);          // union rows of the current block-tile: row u at u * 1024, lane l's chunk at l * 16
    for (int i = threadIdx.x; i < NF4; i += blockDim.x) lw[i] = ((const f32x4*)a.packed)[i];
    const float* lbias = (const float*)(lw + S2H_FRAGS * 64);
    int lane = threadIdx.x & 63;
    const int m = lane & 15, kg = lane >> 4;
    const int jl = lane >> 2, ql = lane & 3;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = a.S, T = a.T;
    const int nx = (a.nxcd > 1 && gridDim.x >= (unsigned)a.nxcd && (gridDim.x % a.nxcd) == 0) ? a.nxcd : 1;
    const int xcd = blockIdx.x % nx, lb = blockIdx.x / nx, nbx = gridDim.x / nx;
    const int blk0 = nx == 1 ? xcd_blk0[0] : xcd_blk0[xcd], blk1 = nx == 1 ? xcd_blk0[a.nxcd] : xcd_blk0[xcd + 1];
    const long long n_items = (long long)(blk1 - blk0) * T;          // item = (block, station tile), swept backwards
    __syncthreads();
    if (lb >= n_items) return;
    const float a2 = a.slope2 != nullptr ? *a.slope2 : lbias[32], ab1 = lbias[33];
    typedef const __attribute__((address_space(1))) char* gbytes;
    typedef const __attribute__((address_space(1))) f32x4* grow;
    typedef const __attribute__((address_space(1))) u32x4* gfrag;
    typedef const __attribute__((address_space(1))) float* gflt;
    const unsigned plane = (unsigned)S * 16u;
    const unsigned pc = (unsigned)S * 128u, pw = (unsigned)S * 64u, pe = (unsigned)S * 32u, pm = (unsigned)S * 4u;
    const unsigned kgp = (unsigned)kg * plane, kge = (unsigned)min(kg, 1) * plane;
    const unsigned q16 = 16u * (unsigned)ql;
    const int bperm = (4 * m + kg) * 4;
    const unsigned m_T = ItemIter::recip((unsigned)T);

    // item number (clamped to this workgroup's last one) -> (block, station tile)
    long long it_last = lb;
    while (it_last + nbx < n_items) it_last += nbx;
    auto item_of = [&](long long it, int& blk, int& tb) {
        const long long itr = n_items - 1 - (it <= it_last ? it : it_last);
        unsigned rem;
        const unsigned kb = T <= 1 ? (rem = 0u, (unsigned)itr) : ItemIter::fdiv((unsigned)itr, (unsigned)T, m_T, rem);
        blk = __builtin_amdgcn_readfirstlane(blk0 + (int)kb);
        tb = __builtin_amdgcn_readfirstlane((int)rem);
    };
    // ---- union rows of a block-tile: global -> registers (issued one block-tile ahead) -> LDS. `idl`: lane u holds the source node
    // of union row u, lane 63's copy of U rides in `Uv` (both loaded one MORE block-tile ahead: nothing here waits on a fresh load)
    f32x4 stg[S2U_NSTG];
    auto stage_issue = [&](int idl, int U, int tb) {
        const int s = tb * 16 + m, sc = s < S ? s : S - 1;
        const unsigned lo = kgp + (unsigned)sc * 16u;
#pragma unroll
        for (int i = 0; i < S2U_NSTG; ++i) {
            const int u = wave + S2U_WPB * i;
            if (u < U) {
                const unsigned long long vb = s2h_base<BIG>(a.wv, __builtin_amdgcn_readlane(idl, u), pw);
                stg[i] = *(grow)((gbytes)vb + lo);
            }
        }
    };
    auto stage_write = [&](int U) {
#pragma unroll
        for (int i = 0; i < S2U_NSTG; ++i) {
            const int u = wave + S2U_WPB * i;
            if (u < U) *(f32x4*)(lrows + (unsigned)u * 1024u + (unsigned)lane * 16u) = stg[i];
        }
    };
    // ---- per-node rows: streamed rows + station-neighbour gathers (one node-tile ahead), as in k_stage2_h2
    struct Node { f32x4 ru[KS], c1, c2; u32x4 ea; float mq; } R;
    int sta[KS];
    auto load_sta = [&](int tb) {
        const int s = tb * 16 + jl;
        load_sta_ids<KS>(a.sta_col, s < S ? s : S - 1, sta);
    };
    auto issue = [&](int idv, int tb) {
        const int g = max(__builtin_amdgcn_readlane(idv, 0), 0);         // (-1: an empty slot of a short block: rows of node 0, result dropped)
        const int s = tb * 16 + m, sc = s < S ? s : S - 1;
        const unsigned so = (unsigned)sc * 16u;
        const unsigned lo = kgp + so;
        const unsigned long long cb = s2h_base<BIG>(a.c, g, pc);
        R.c1 = *(grow)((gbytes)cb + lo);
        R.c2 = *(grow)((gbytes)cb + (lo + 4u * plane));
        const unsigned long long mb = s2h_base<BIG>(a.mm_int, g, pm);
        R.mq = *(gflt)((gbytes)mb + (unsigned)sc * 4u);
        if (s >= S) R.mq = 0.f;
        const unsigned long long eb = s2h_base<BIG>(a.ea_frag, g, pe);
        R.ea = *(gfrag)((gbytes)eb + (kge + so));
        const unsigned long long ub = s2h_base<BIG>(a.wu, g, pw);
#pragma unroll
        for (int k = 0; k < KS; ++k) R.ru[k] = *(grow)((gbytes)ub + ((unsigned)sta[k] * 64u + q16));
    };
    // Every wave takes the two slots b = wave, wave + 4 of EVERY item of its workgroup (a short block leaves slots empty: their
    // index row starts with -1): the slot sequence is regular, so no control decision waits on a table load.
    // slot number j -> (item j / 2, half j % 2); index rows are loaded two slots ahead, station ids and per-node rows one ahead.
    auto slot = [&](long long j, int& blk, int& tb, int& b) {
        item_of(lb + (j >> 1) * nbx, blk, tb);
        b = wave + S2U_WPB * (int)(j & 1);
    };
    auto idv_at = [&](long long j) {
        int blk, tb, b;
        slot(j, blk, tb, b);
        return blocks[blk].idx[b][m];
    };
    const long long n_my_items = (n_items - lb + nbx - 1) / nbx;
    int blk_c, tb_c, blk_n, tb_n, blk_2, tb_2, dummy;
    item_of(lb, blk_c, tb_c);
    item_of(lb + nbx, blk_n, tb_n);
    int idl_c = blocks[blk_c].ids[lane], U_c = blocks[blk_c].U;
    int idl_n = blocks[blk_n].ids[lane], U_n = blocks[blk_n].U;
    stage_issue(idl_c, __builtin_amdgcn_readfirstlane(U_c), tb_c);
    int idv_c = idv_at(0), idv_n = idv_at(1);
    load_sta(tb_c);
    issue(idv_c, tb_c);
    for (long long ii = 0; ii < n_my_items; ++ii) {
        const bool has_next_item = ii + 1 < n_my_items;
        item_of(lb + (ii + 2) * nbx, blk_2, tb_2);
        const int idl_2 = blocks[blk_2].ids[lane], U_2 = blocks[blk_2].U;       // two items ahead (clamped), consumed next iteration
        stage_write(__builtin_amdgcn_readfirstlane(U_c));
        __syncthreads();
        if (has_next_item) stage_issue(idl_n, __builtin_amdgcn_readfirstlane(U_n), tb_n);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            asm volatile("" : "+v"(lane));
            const long long j = 2 * ii + h;
            const int idv_2 = idv_at(j + 2 < 2 * n_my_items ? j + 2 : j);
            const int g_raw = __builtin_amdgcn_readlane(idv_c, 0);
            const bool live = g_raw >= 0;
            const int g_c = max(g_raw, 0);
            const int tbc = tb_c;
            const int tb_next = h == 0 ? tb_c : tb_n;          // station tile of the next slot
            (void)dummy;
            // (1) neighbour sums in edge order: station rows from their registers (row layout -> operand layout), source rows from LDS
            f32x4 n1 = {0.f, 0.f, 0.f, 0.f}, n2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KS; ++k) n1 += R.ru[k];
#pragma unroll
            for (int k0 = 0; k0 < KP; k0 += 5) {
                f32x4 rv[5];
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    rv[k] = *(const f32x4*)(lrows + (unsigned)__builtin_amdgcn_readlane(idv_c, 1 + k0 + k) * 1024u + (unsigned)lane * 16u);
#pragma unroll
                for (int k = 0; k < 5; ++k) n2 += rv[k];
            }
            f32x4 n1t;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = n1[r];
                n1t[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(bperm, __float_as_int(v)));
            }
            f32x4 o1 = fma4(n1t, 1.f / (float)KS, R.c1), o2 = fma4(n2, 1.f / (float)KP, R.c2);
            const float mq = R.mq;
            const u32x4 eab = R.ea;
            const long long p_sv = SAVE ? (long long)g_c * S + a.sta_user[min(tbc * 16 + m, S - 1)] : 0;
            const bool sv_ok = SAVE && live && tbc * 16 + m < S;
            if (sv_ok) {
                *(f32x4*)(a.save + ((size_t)(SV_O + 0) * a.Pn + p_sv) * 16 + 4 * kg) = o1;
                *(f32x4*)(a.save + ((size_t)(SV_O + 1) * a.Pn + p_sv) * 16 + 4 * kg) = o2;
            }
            o1 = prelu4u(o1, a2);
            o2 = prelu4u(o2, a2);
            asm volatile("" : "+v"(o1), "+v"(o2), "+v"(idv_n));
            // (2) per-node rows of the next slot (it may belong to the next item: only LDS contents are per item). The two slots of an
            // item share their station tile, so the station-neighbour ids change once per item and are loaded a slot ahead of their use
            issue(idv_n, tb_next);
            if (h == 0) load_sta(tb_n);
            if (XL && live && tbc * 16 + m < S) {
                const int su = a.sta_user[tbc * 16 + m];
                float* xl = a.x_latent + ((long long)g_c * S + su) * 30;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * kg + r < 15) { xl[4 * kg + r] = o1[r]; xl[15 + 4 * kg + r] = o2[r]; }
            }
            u32x4 p0, p1, p2;
            p0[0] = cvt_pk_f16(o1[0], o1[1]); p0[1] = cvt_pk_f16(o1[2], o1[3]);
            p0[2] = cvt_pk_f16(o2[0], o2[1]); p0[3] = cvt_pk_f16(o2[2], o2[3]);
            p1[0] = cvt_pk_f16(sub_f16_lo(o1[0], p0[0]), sub_f16_hi(o1[1], p0[0]));
            p1[1] = cvt_pk_f16(sub_f16_lo(o1[2], p0[1]), sub_f16_hi(o1[3], p0[1]));
            p1[2] = cvt_pk_f16(sub_f16_lo(o2[0], p0[2]), sub_f16_hi(o2[1], p0[2]));
            p1[3] = cvt_pk_f16(sub_f16_lo(o2[2], p0[3]), sub_f16_hi(o2[3], p0[3]));
#pragma unroll
            for (int d = 0; d < 4; ++d) p2[d] = pk_mul_f16(p0[d], H2_SIXTEENTH);
            f32x4 bp[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bp[t] = *(const f32x4*)(lbias + 16 * t + 4 * kg);
                const f32x4 w0 = lw[(S2H_FW + 2 * t) * 64 + lane], w1 = lw[(S2H_FW + 2 * t + 1) * 64 + lane];
                const f32x4 we = lw[(S2H_FE + t) * 64 + lane];
                bp[t] = MFMA16H(w0, p1, bp[t]);
                bp[t] = MFMA16H(w1, p2, bp[t]);
                bp[t] = MFMA16H(we, eab, bp[t]);
                bp[t] = MFMA16H(w0, p0, bp[t]);
                if (sv_ok) *(f32x4*)(a.save + ((size_t)(SV_ZB + t) * a.Pn + p_sv) * 16 + 4 * kg) = bp[t];
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 v = prelu4u(bp[t], ab1) * mq;
                v.x = row_sum16_tree(v.x); v.y = row_sum16_tree(v.y); v.z = row_sum16_tree(v.z); v.w = row_sum16_tree(v.w);
                if (m == 0 && live) *(f32x4*)(a.part + ((long long)g_c * T + tbc) * 32 + 16 * t + 4 * kg) = v;
            }
            idv_c = idv_n; idv_n = idv_2;
        }
        if (!has_next_item) break;
        __syncthreads();              // every wave has finished reading this block-tile's rows
        blk_c = blk_n; tb_c = tb_n; blk_n = blk_2; tb_n = tb_2;
        idl_c = idl_n; U_c = U_n; idl_n = idl_2; U_n = U_2;
    }
}
