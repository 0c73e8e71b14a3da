// train_arrival_kernels.hpp -- backward of the arrival-association head of a training step (reference
// StationSourceAttentionMergedPhases, module.py:662-775, inside train_GENIE_model.py:1786-1861). Included by genie_hip.hip after
// train_assoc_kernels.hpp (one translation unit: MFMA tile helpers, weight images, partial-slot conventions).
//
// The training forward IS the inference kernel (k_arr_ctx, k_arr_e0max, k_arrivals); with ArArgs::save set it also keeps, per
// (source i, pick a), the three normalised head aggregates and the softmax statistics (AT_STAT floats). The backward never builds
// the reference's edge list either: queries / values are functions of (pick b, source i) plus the self / null link bits, so
//   k_arrt_tgt_bwd  per target (i, a), pointwise:  d out -> proj_2, PReLU4, proj_1 (weight gradients) -> d agg = d z / 3 and
//                   c_h = d agg . agg_h (the softmax's "sum over the segment" term, closed form: sum_k alpha_k v_k = agg_h);
//   k_arrt_ent_bwd  per (source i, station u) workgroup, 16 entries (picks b of u, then the null pick) per wave: forward of the
//                   entry recomputed, then one sweep over the station's targets (rows of k_arrt_tgt_bwd staged through LDS) gives
//                   d score / d value of the entry's plain and self variants; then the edge MLPs backwards (weight gradients as
//                   node-contracting MFMAs into the wave's partial slot), d arrival_p / d arrival_s of pick b, and the entry's
//                   share of d context (per source and link variant, summed over the workgroup in a fixed order);
//   k_arrt_ctx_bwd  per source: d context [4 variants][3 heads] -> f_src_context_2 / PReLU1 / f_src_context_1 gradients, d src_embed;
//   k_arrt_ctx_red / k_arrt_pick_sum: fixed-order sums over the sources.
// No atomics: every output element has one writer, all reductions run in a fixed order (bitwise reproducible).

#define GTA_P1(t) (t)                          // proj_1^T: d z from d pre (hidden block t)
#define GTA_V2(b, h) (2 + (b) * 3 + (h))       // f_values_2^T: d hidden block b from d value head h
#define GTA_Q2(b, h) (8 + (b) * 3 + (h))       // f_arrival_query_2^T
#define GTA_V1(s, t) (14 + (s) * 2 + (t))      // f_values_1^T: d arrival_p (s = 0) / arrival_s (s = 1) from d pre block t
#define GTA_Q1(s, t) (18 + (s) * 2 + (t))      // f_arrival_query_1^T
#define GTA_GROUPS 22
constexpr int GTA_IMG_FLOATS = GTA_GROUPS * 256 + 16;
constexpr int AT_TG = 32;                      // floats per target between the two passes: d agg [16], c [3], max [3], den [3]
constexpr int AE_TCH = 64;                     // targets staged per LDS chunk
constexpr int AE_NACC = 24, AE_NVEC = 10;      // k_arrt_ent_bwd: Q1 (6), V1 (6), Q2 (6), V2 (6) blocks; Q1_B 2, V1_B 2, Q2_B 3, V2_B 3
constexpr int AT_NACC = 2, AT_NVEC = 6;        // k_arrt_tgt_bwd: proj_1 (2); P1_B 2, proj_2 rows (m, t) 4
constexpr int AC_PARAMS = 30 * 33 + 30 + 45 * 30 + 45 + 1;     // f_src_context_1 (w, b), f_src_context_2 (w, b), activate1
constexpr int AC_STRIDE = 2432;

struct AtArgs {
    int n_tgt;                   // n_src * n_arv
    const float* img; const float* timg;
    const float* tstat;          // [n_tgt][AT_STAT] of the training forward
    const float* d_out;          // [n_tgt][2]
    float* tg;                   // [n_tgt][AT_TG]
    float* part; int n_acc, n_vec;
};

__global__ __launch_bounds__(256) void k_arrt_tgt_bwd(AtArgs b) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const TlImg im = tl_stage_image(sm, b.img, GA_GROUPS2, GA_BIAS2);
    float* tw_ = sm + GA2_IMG_FLOATS;
    for (int i = threadIdx.x; i < GTA_IMG_FLOATS / 4; i += blockDim.x) ((f32x4*)tw_)[i] = ((const f32x4*)b.timg)[i];
    const f32x4* tw = (const f32x4*)tw_;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    float* trs = tw_ + GTA_IMG_FLOATS + wave * 16 * 17;
    const float act4 = im.scal[2];
    const f32x4 w2[2][2] = {{tl_bias(im, 12, q), tl_bias(im, 13, q)}, {tl_bias(im, 14, q), tl_bias(im, 15, q)}};
    const TpSlot ps = tp_open(b.part, b.n_acc, b.n_vec, blockIdx.x * 4 + wave, lane);
    float s_a4 = 0.f, s_b0 = 0.f, s_b1 = 0.f;
    const int ntiles = (b.n_tgt + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int t = tile * 16 + j;
        const bool ok = t < b.n_tgt;
        const float* sv = b.tstat + (long long)(ok ? t : b.n_tgt - 1) * AT_STAT;
        f32x4 ag[3];
#pragma unroll
        for (int h = 0; h < 3; ++h) ag[h] = *(const f32x4*)(sv + 16 * h + 4 * q);
        const f32x4 z = ((ag[0] + ag[1]) + ag[2]) / 3.f;
        const float d0 = ok ? b.d_out[(long long)t * 2] : 0.f, d1 = ok ? b.d_out[(long long)t * 2 + 1] : 0.f;
        f32x4 dpre[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const f32x4 pre = mma_block(tl_bias(im, 10 + k, q), TLW(im, GA_P1(k)), z);
            const f32x4 pa = prelu4(pre, act4);
            const f32x4 dp = w2[0][k] * d0 + w2[1][k] * d1;
            tp_vec(ps, 2 + k, j, q, pa * d0);
            tp_vec(ps, 4 + k, j, q, pa * d1);
            s_a4 += negsum4(dp, pre);
            dpre[k] = dp * dprelu4(pre, act4);
            tp_vec(ps, k, j, q, dpre[k]);
        }
        if (q == 0) { s_b0 += d0; s_b1 += d1; }
        {
            const f32x4 zt = tr16(z, trs, j, q);
            tp_acc(ps, 0, lane, outer16(tl_zero(), tr16(dpre[0], trs, j, q), zt));
            tp_acc(ps, 1, lane, outer16(tl_zero(), tr16(dpre[1], trs, j, q), zt));
        }
        const f32x4 dg = mma_block(mma_block(tl_zero(), tw[GTA_P1(0) * 64 + lane], dpre[0]), tw[GTA_P1(1) * 64 + lane], dpre[1]) / 3.f;
        float ch[3];
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const f32x4 p = dg * ag[h];
            float s = ((p.x + p.y) + p.z) + p.w;
            s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
            ch[h] = s;
        }
        if (ok) {
            float* o = b.tg + (long long)t * AT_TG;
            *(f32x4*)(o + 4 * q) = dg;
            if (q == 0) {
#pragma unroll
                for (int h = 0; h < 3; ++h) { o[16 + h] = ch[h]; o[19 + h] = sv[48 + h]; o[22 + h] = sv[51 + h]; }
            }
        }
    }
    tp_scal(ps, 0, lane, s_a4);
    tp_scal(ps, 1, lane, s_b0);
    tp_scal(ps, 2, lane, s_b1);
}

struct AeArgs {
    ArArgs f;                    // the forward's arguments (ctx: k_arr_ctx's output, e0max: k_arr_e0max's)
    const float* timg;
    const float* tg;             // [n_src * n_arv][AT_TG] (k_arrt_tgt_bwd)
    float* darv;                 // [n_src][n_arv][32]: d arrival_p at 0..14, d arrival_s at 16..30 of pick b through source i
    float* cpair;                // [n_src * n_useg][192]: the (source, station) pair's share of d context [variant][head][16]
    float* part; int n_acc, n_vec;
};

__global__ __launch_bounds__(256) void k_arrt_ent_bwd(AeArgs b) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const ArArgs& a = b.f;
    const TlImg im = tl_stage_image(sm, a.img, GA_GROUPS2, GA_BIAS2);
    float* tw_ = sm + GA2_IMG_FLOATS;
    for (int i = threadIdx.x; i < GTA_IMG_FLOATS / 4; i += blockDim.x) ((f32x4*)tw_)[i] = ((const f32x4*)b.timg)[i];
    const f32x4* tw = (const f32x4*)tw_;
    float* trs_all = tw_ + GTA_IMG_FLOATS;
    float* cx = trs_all + 4 * 16 * 17;                // [4][48] context vectors of the source
    float* tgs = cx + 192;                            // [AE_TCH][AT_TG]
    int* tsl = (int*)(tgs + AE_TCH * AT_TG);          // [AE_TCH] pick index whose entry is the target's self edge
    float* cacc = (float*)(tsl + AE_TCH);             // [4 waves][3][256] d context tiles of the pair
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    float* trs = trs_all + wave * 16 * 17;
    const int E = *a.e0max;
    __syncthreads();
    const float act2 = im.scal[0], act3 = im.scal[1];
    const float eps = a.eps, e2 = eps * eps, sq = sqrtf(15.f);
    const TpSlot ps = tp_open(b.part, b.n_acc, b.n_vec, blockIdx.x * 4 + wave, lane);
    float s_a2 = 0.f, s_a3 = 0.f;
    const int npairs = a.n_src * a.n_useg;
    for (int pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
        const int i = pair / a.n_useg, ug = pair - i * a.n_useg;
        const int u = a.seg_sta[ug], r0 = a.seg_start[ug], L = a.seg_len[ug];
        const float st = a.stime[i];
        const float tp_src = a.trv_src[((long long)i * a.n_sta + u) * 2 + 0] + st, ts_src = a.trv_src[((long long)i * a.n_sta + u) * 2 + 1] + st;
        const float rel_null = -eps - (-eps + st);
        __syncthreads();                              // the previous pair's cx / cacc are no longer read
        for (int k = threadIdx.x; k < 192; k += blockDim.x) cx[k] = a.ctx[(long long)i * 192 + k];
        f32x4 accc[3] = {tl_zero(), tl_zero(), tl_zero()};
        const int ntile = (L + 1 + 15) / 16;
        for (int round = 0; round * 4 < ntile; ++round) {
            const int tile = round * 4 + wave;
            const bool active = tile < ntile;         // uniform per wave
            const int r = tile * 16 + j;
            const bool ok = active && r <= L;
            const bool nul = !(active && r < L);      // the null pick, and the idle lanes of a partial tile (zero inputs, never kept)
            const int bp = nul ? 0 : a.order[r0 + r];
            const int kb = ok ? (nul ? a.n_arv : bp) : -2;
            const bool nl = kb == E;
            const float tp = nul ? 0.f : a.tpick[bp];
            const float rp = nul ? rel_null : tp - tp_src, rs = nul ? rel_null : tp - ts_src;
            const bool keep = ok && (nul ? fabsf(rel_null) < 2.f * eps : (fabsf(rp) < 2.f * eps || fabsf(rs) < 2.f * eps));
            const float ph = nul ? -1.f : a.phase[bp];
            const float f6[6] = {expf(-0.5f * (rp * rp) / e2), (rp > 0.f) - (rp < 0.f) + 0.f, ph,
                                 expf(-0.5f * (rs * rs) / e2), (rs > 0.f) - (rs < 0.f) + 0.f, ph};
            const float x0 = q == 0 ? f6[0] : (q == 1 ? f6[1] : (q == 2 ? f6[2] : f6[3]));
            const float x1 = q == 0 ? f6[4] : (q == 1 ? f6[5] : 0.f);
            const float lnk = (q == 1 && nl) ? 1.f : 0.f;
            const f32x4 xp = nul ? tl_zero() : tl_load15(a.arv_p + (long long)bp * 15, q);
            const f32x4 xs = nul ? tl_zero() : tl_load15(a.arv_s + (long long)bp * 15, q);
            f32x4 zq[2], zv[2], zw[2], hq[2], hv[2], hw[2], qh[3], vh[3], wh[3], c0[3], c1[3];
            float s0[3], s1[3];
            __syncthreads();                          // cx of this pair is complete (first round); the previous round's tgs are done
            if (active) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 z = mma_block(tl_bias(im, t, q), TLW(im, GA_Q1(t, 0)), xp);
                    z = mma_block(z, TLW(im, GA_Q1(t, 1)), xs);
                    z = MFMA16(TLW(im, GA_Q1(t, 2)).x, x0, z);
                    z = MFMA16(TLW(im, GA_Q1(t, 2)).y, x1, z);
                    zq[t] = z; hq[t] = prelu4(z, act2);
                    f32x4 v = mma_block(tl_bias(im, 2 + t, q), TLW(im, GA_V1(t, 0)), xp);
                    v = mma_block(v, TLW(im, GA_V1(t, 1)), xs);
                    v = MFMA16(TLW(im, GA_V1(t, 2)).x, x0, v);
                    v = MFMA16(TLW(im, GA_V1(t, 2)).y, x1, v);
                    zv[t] = MFMA16(TLW(im, GA_V1(t, 3)).x, lnk, v);
                    zw[t] = MFMA16(TLW(im, GA_V1(t, 3)).x, q == 0 ? 1.f : lnk, v);
                    hv[t] = prelu4(zv[t], act3);
                    hw[t] = prelu4(zw[t], act3);
                }
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    qh[h] = mma_block(mma_block(tl_bias(im, 4 + h, q), TLW(im, GA_Q2(h, 0)), hq[0]), TLW(im, GA_Q2(h, 1)), hq[1]);
                    c0[h] = *(const f32x4*)(cx + (nl ? 96 : 0) + h * 16 + 4 * q);
                    c1[h] = *(const f32x4*)(cx + (nl ? 144 : 48) + h * 16 + 4 * q);
                    const f32x4 p0 = qh[h] * c0[h], p1 = qh[h] * c1[h];
                    float t0 = ((p0.x + p0.y) + p0.z) + p0.w, t1 = ((p1.x + p1.y) + p1.z) + p1.w;
                    t0 += __shfl_xor(t0, 16); t0 += __shfl_xor(t0, 32);
                    t1 += __shfl_xor(t1, 16); t1 += __shfl_xor(t1, 32);
                    s0[h] = t0 / sq; s1[h] = t1 / sq;
                    vh[h] = mma_block(mma_block(tl_bias(im, 7 + h, q), TLW(im, GA_V2(h, 0)), hv[0]), TLW(im, GA_V2(h, 1)), hv[1]);
                    wh[h] = mma_block(mma_block(tl_bias(im, 7 + h, q), TLW(im, GA_V2(h, 0)), hw[0]), TLW(im, GA_V2(h, 1)), hw[1]);
                }
            }
            // ---- one sweep over the station's targets: d score, d value of the entry's plain / self variants
            float dS0[3] = {0.f, 0.f, 0.f}, dS1[3] = {0.f, 0.f, 0.f};
            f32x4 dV0[3] = {tl_zero(), tl_zero(), tl_zero()}, dV1[3] = {tl_zero(), tl_zero(), tl_zero()};
            for (int tc = 0; tc < L; tc += AE_TCH) {
                if (tc) __syncthreads();
                const int nt = min(AE_TCH, L - tc);
                for (int idx = threadIdx.x; idx < nt * 8; idx += blockDim.x) {
                    const int row = idx >> 3, c4 = idx & 7;
                    const long long tgt = (long long)i * a.n_arv + a.order[r0 + tc + row];
                    ((f32x4*)tgs)[row * 8 + c4] = ((const f32x4*)(b.tg + tgt * AT_TG))[c4];
                }
                for (int idx = threadIdx.x; idx < nt; idx += blockDim.x)
                    tsl[idx] = E > 0 ? (int)(((long long)a.order[r0 + tc + idx] + (long long)i * a.n_arv) % E) : -1;
                __syncthreads();
                if (active) {
                    for (int k = 0; k < nt; ++k) {
                        const float* tr = tgs + k * AT_TG;
                        const bool self = kb == tsl[k];
                        const f32x4 dg = *(const f32x4*)(tr + 4 * q);
#pragma unroll
                        for (int h = 0; h < 3; ++h) {
                            const float sc = self ? s1[h] : s0[h];
                            const f32x4 v = self ? wh[h] : vh[h];
                            const float al = keep ? expf(sc - tr[19 + h]) / (tr[22 + h] + 1e-16f) : 0.f;
                            const f32x4 p = dg * v;
                            float t = ((p.x + p.y) + p.z) + p.w;
                            t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
                            const float ds = al * (t - tr[16 + h]);
                            dS1[h] += self ? ds : 0.f;
                            dS0[h] += self ? 0.f : ds;
                            dV1[h] += dg * (self ? al : 0.f);
                            dV0[h] += dg * (self ? 0.f : al);
                        }
                    }
                }
            }
            if (!active) continue;                    // (no barrier below this point inside the round)
            // ---- the value MLP backwards
            f32x4 dzv[2], dzw[2], dzq[2];
            {
                const f32x4 hvt[2] = {tr16(hv[0], trs, j, q), tr16(hv[1], trs, j, q)};
                const f32x4 hwt[2] = {tr16(hw[0], trs, j, q), tr16(hw[1], trs, j, q)};
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    const f32x4 d0t = tr16(dV0[h], trs, j, q), d1t = tr16(dV1[h], trs, j, q);
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb)
                        tp_acc(ps, 18 + h * 2 + bb, lane, outer16(outer16(tl_zero(), d0t, hvt[bb]), d1t, hwt[bb]));
                    tp_vec(ps, 7 + h, j, q, dV0[h] + dV1[h]);
                }
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    f32x4 dh = tl_zero(), dw = tl_zero();
#pragma unroll
                    for (int h = 0; h < 3; ++h) {
                        dh = mma_block(dh, tw[GTA_V2(bb, h) * 64 + lane], dV0[h]);
                        dw = mma_block(dw, tw[GTA_V2(bb, h) * 64 + lane], dV1[h]);
                    }
                    s_a3 += negsum4(dh, zv[bb]) + negsum4(dw, zw[bb]);
                    dzv[bb] = dh * dprelu4(zv[bb], act3);
                    dzw[bb] = dw * dprelu4(zw[bb], act3);
                    tp_vec(ps, 2 + bb, j, q, dzv[bb] + dzw[bb]);
                }
            }
            // ---- the query MLP backwards
            {
                f32x4 dq[3];
                const f32x4 hqt[2] = {tr16(hq[0], trs, j, q), tr16(hq[1], trs, j, q)};
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    dq[h] = (c0[h] * dS0[h] + c1[h] * dS1[h]) / sq;
                    const f32x4 dqt = tr16(dq[h], trs, j, q);
                    tp_acc(ps, 12 + h * 2 + 0, lane, outer16(tl_zero(), dqt, hqt[0]));
                    tp_acc(ps, 12 + h * 2 + 1, lane, outer16(tl_zero(), dqt, hqt[1]));
                    tp_vec(ps, 4 + h, j, q, dq[h]);
                }
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    f32x4 dh = tl_zero();
#pragma unroll
                    for (int h = 0; h < 3; ++h) dh = mma_block(dh, tw[GTA_Q2(bb, h) * 64 + lane], dq[h]);
                    s_a2 += negsum4(dh, zq[bb]);
                    dzq[bb] = dh * dprelu4(zq[bb], act2);
                    tp_vec(ps, bb, j, q, dzq[bb]);
                }
            }
            // ---- first layers: weight gradients (inputs arrival_p, arrival_s, the six relative-time features, the two link bits)
            {
                const f32x4 xpt = tr16(xp, trs, j, q), xst = tr16(xs, trs, j, q);
                const float nlf = nl ? 1.f : 0.f;
                const f32x4 fp = q == 0 ? f32x4{f6[0], f6[1], f6[2], f6[3]} : (q == 1 ? f32x4{f6[4], f6[5], 0.f, nlf} : tl_zero());
                const f32x4 fs = q == 0 ? f32x4{f6[0], f6[1], f6[2], f6[3]} : (q == 1 ? f32x4{f6[4], f6[5], 1.f, nlf} : tl_zero());
                const f32x4 fpt = tr16(fp, trs, j, q), fst = tr16(fs, trs, j, q);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x4 dqt = tr16(dzq[t], trs, j, q), dvt = tr16(dzv[t], trs, j, q), dwt = tr16(dzw[t], trs, j, q);
                    const f32x4 dst = dvt + dwt;
                    tp_acc(ps, t * 3 + 0, lane, outer16(tl_zero(), dqt, xpt));
                    tp_acc(ps, t * 3 + 1, lane, outer16(tl_zero(), dqt, xst));
                    tp_acc(ps, t * 3 + 2, lane, outer16(tl_zero(), dqt, fpt));
                    tp_acc(ps, 6 + t * 3 + 0, lane, outer16(tl_zero(), dst, xpt));
                    tp_acc(ps, 6 + t * 3 + 1, lane, outer16(tl_zero(), dst, xst));
                    tp_acc(ps, 6 + t * 3 + 2, lane, outer16(outer16(tl_zero(), dvt, fpt), dwt, fst));
                }
            }
            // ---- d arrival_p / d arrival_s of the pick (through this source)
            {
                f32x4 dx[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f32x4 d = mma_block(tl_zero(), tw[GTA_V1(s, 0) * 64 + lane], dzv[0] + dzw[0]);
                    d = mma_block(d, tw[GTA_V1(s, 1) * 64 + lane], dzv[1] + dzw[1]);
                    d = mma_block(d, tw[GTA_Q1(s, 0) * 64 + lane], dzq[0]);
                    dx[s] = mma_block(d, tw[GTA_Q1(s, 1) * 64 + lane], dzq[1]);
                }
                if (ok && !nul) {
                    float* o = b.darv + ((long long)i * a.n_arv + bp) * 32;
                    *(f32x4*)(o + 4 * q) = dx[0];
                    *(f32x4*)(o + 16 + 4 * q) = dx[1];
                }
            }
            // ---- d context: D[channel][variant] += sum over the entries q_h[channel] * d score[variant] / sqrt(15)
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                const f32x4 xc = q == 0 ? f32x4{nl ? 0.f : dS0[h], nl ? 0.f : dS1[h], nl ? dS0[h] : 0.f, nl ? dS1[h] : 0.f} / sq : tl_zero();
                accc[h] = outer16(accc[h], tr16(qh[h], trs, j, q), tr16(xc, trs, j, q));
            }
        }
        // ---- the pair's d context: the four waves' tiles summed in wave order
#pragma unroll
        for (int h = 0; h < 3; ++h) *(f32x4*)(cacc + (wave * 3 + h) * 256 + lane * 4) = accc[h];
        __syncthreads();
        if (threadIdx.x < 192) {
            const int v = threadIdx.x / 48, rr = threadIdx.x - v * 48, h = rr >> 4, l = rr & 15;
            const int el = (v + 16 * (l >> 2)) * 4 + (l & 3);         // tile element (row l = channel, column v = variant)
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) s += cacc[(w * 3 + h) * 256 + el];
            b.cpair[(long long)pair * 192 + threadIdx.x] = s;
        }
    }
    tp_scal(ps, 0, lane, s_a2);
    tp_scal(ps, 1, lane, s_a3);
}

// per source: d context -> f_src_context_2, PReLU1, f_src_context_1 (module.py:734-737); parameter gradients of the source into
// cpart[i] (layout: W1 [30][33], b1 [30], W2 [45][30], b2 [45], activate1), d src_embed[i]
__global__ __launch_bounds__(128) void k_arrt_ctx_bwd(const float* __restrict__ raw, int o_c1w, int o_c1b, int o_c2w, int o_c2b, int o_a1,
                                                     const float* __restrict__ src_embed, const float* __restrict__ stime, int n_src,
                                                     int n_useg, const float* __restrict__ cpair, float* __restrict__ cpart,
                                                     float* __restrict__ d_src_embed) {
    __shared__ float dctx[4][48], pre[4][32], hid[4][32], dpre[4][32], inp[33], red[128];
    const int i = blockIdx.x;
    if (i >= n_src) return;
    const float act1 = raw[o_a1];
    for (int idx = threadIdx.x; idx < 192; idx += blockDim.x) {
        float s = 0.f;
        for (int ug = 0; ug < n_useg; ++ug) s += cpair[((long long)i * n_useg + ug) * 192 + idx];
        dctx[idx / 48][idx % 48] = s;
    }
    for (int k = threadIdx.x; k < 33; k += blockDim.x) inp[k] = k < 30 ? src_embed[(long long)i * 30 + k] : (k == 30 ? stime[i] : 0.f);
    __syncthreads();
    for (int idx = threadIdx.x; idx < 4 * 30; idx += blockDim.x) {
        const int v = idx / 30, c = idx - v * 30;
        float t = raw[o_c1b + c];
        for (int k = 0; k < 30; ++k) t += raw[o_c1w + c * 33 + k] * inp[k];
        t += raw[o_c1w + c * 33 + 30] * inp[30];
        if (v & 1) t += raw[o_c1w + c * 33 + 31];
        if (v & 2) t += raw[o_c1w + c * 33 + 32];
        pre[v][c] = t;
        hid[v][c] = prelu1(t, act1);
    }
    __syncthreads();
    float sa = 0.f;
    for (int idx = threadIdx.x; idx < 4 * 30; idx += blockDim.x) {
        const int v = idx / 30, k = idx - v * 30;
        float d = 0.f;
        for (int ch = 0; ch < 45; ++ch) d += raw[o_c2w + ch * 30 + k] * dctx[v][(ch / 15) * 16 + ch % 15];
        sa += d * fminf(pre[v][k], 0.f);
        dpre[v][k] = d * (pre[v][k] > 0.f ? 1.f : act1);
    }
    red[threadIdx.x] = sa;
    __syncthreads();
    float* out = cpart + (long long)i * AC_STRIDE;
    for (int idx = threadIdx.x; idx < 30 * 33; idx += blockDim.x) {
        const int c = idx / 33, k = idx - c * 33;
        float s = 0.f;
        for (int v = 0; v < 4; ++v) {
            const float x = k < 31 ? inp[k] : (k == 31 ? (float)(v & 1) : (float)(v >> 1));
            s += dpre[v][c] * x;
        }
        out[idx] = s;
    }
    for (int c = threadIdx.x; c < 30; c += blockDim.x) out[990 + c] = ((dpre[0][c] + dpre[1][c]) + dpre[2][c]) + dpre[3][c];
    for (int idx = threadIdx.x; idx < 45 * 30; idx += blockDim.x) {
        const int ch = idx / 30, k = idx - ch * 30, cc = (ch / 15) * 16 + ch % 15;
        float s = 0.f;
        for (int v = 0; v < 4; ++v) s += dctx[v][cc] * hid[v][k];
        out[1020 + idx] = s;
    }
    for (int ch = threadIdx.x; ch < 45; ch += blockDim.x) {
        const int cc = (ch / 15) * 16 + ch % 15;
        out[2370 + ch] = ((dctx[0][cc] + dctx[1][cc]) + dctx[2][cc]) + dctx[3][cc];
    }
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int k = 0; k < 120 && k < (int)blockDim.x; ++k) s += red[k];
        out[2415] = s;
    }
    for (int k = threadIdx.x; k < 30; k += blockDim.x) {
        float s = 0.f;
        for (int v = 0; v < 4; ++v)
            for (int c = 0; c < 30; ++c) s += raw[o_c1w + c * 33 + k] * dpre[v][c];
        d_src_embed[(long long)i * 30 + k] = s;
    }
}

// sum over the sources of the context MLP's parameter gradients -> their entries of the gradient blob
__global__ void k_arrt_ctx_red(const float* __restrict__ cpart, int n_src, int o_c1w, int o_c1b, int o_c2w, int o_c2b, int o_a1,
                               float* __restrict__ blob) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= AC_PARAMS) return;
    float s = 0.f;
    for (int i = 0; i < n_src; ++i) s += cpart[(long long)i * AC_STRIDE + idx];
    const int dst = idx < 990 ? o_c1w + idx : (idx < 1020 ? o_c1b + idx - 990 : (idx < 2370 ? o_c2w + idx - 1020 : (idx < 2415 ? o_c2b + idx - 2370 : o_a1)));
    blob[dst] = s;
}

// d arrival_p / d arrival_s of every pick: sum over the sources, in source order
__global__ void k_arrt_pick_sum(const float* __restrict__ darv, int n_src, int n_arv, float* __restrict__ d_p, float* __restrict__ d_s) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_arv * 30) return;
    const int bp = idx / 30, c = idx - bp * 30, s = c >= 15, cc = c - 15 * s;
    float t = 0.f;
    for (int i = 0; i < n_src; ++i) t += darv[((long long)i * n_arv + bp) * 32 + 16 * s + cc];
    (s ? d_s : d_p)[(long long)bp * 15 + cc] = t;
}
