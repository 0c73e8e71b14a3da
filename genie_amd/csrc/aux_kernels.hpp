// Part of genie_hip.hip (one translation unit, included inside its anonymous namespace): kernels around the path: pick embedding, position / edge-feature tables, neighbour means and the small backward helpers, exact kNN, row selection, irregular-graph CSR builder, exports.

// ------------------------------------------------------------------------------------------------
// Pick -> Slice/Mask embedding on device: `extract_input_from_data`,
// /root/reference/Code/process_utils.py:460-642 (use_sign_input False or True). Step 1: per-station Gaussian-kernel time
// series of the P- and S-labelled picks by scatter-max (:499-569; max is order independent -> deterministic atomics).
// Step 2: every product node reads the series of its station at the theoretical P / S arrival index (:599-629).
// ------------------------------------------------------------------------------------------------
struct EmbArgs {
    const double* pick_t; const int32_t* pick_sta; const int32_t* pick_phase;
    int n_picks, n_time, n_extra, S;
    double t0, tref0, dt, sigma;
    float* emb;            // [2][S][n_time]: P-labelled series, then S-labelled
    const float* trv;      // [rows, 2] theoretical P / S travel time of every product node
    long long rows;
    float* slice; float* mask;
    unsigned* xs;          // optional: the split rows of k_stage1_h2, written together with Slice / Mask
    const int32_t* sta_inv; // station processing order of the split rows (caller's station -> internal), or null
    float* mm;              // with sta_inv: max of the Mask row, in processing order
    int sign_input;         // use_sign_input: True (config.yaml:93, process_utils.py:610-614): every feature times the sign of the negative
                            // forward difference of the series it is read from, at the index it is read at
    const int32_t* sta_of;  // irregular product graph: station of every product node (else p % S)
    int no_phase;           // use_phase_types: False (config.yaml:91): the phase-informed columns 2, 3 of Slice / Mask are zero
                            // (process_continuous_days.py:783-786; the caller passes every pick with phase 0, :562-563)
};

__global__ void k_embed_scatter(EmbArgs a) {
    const int per = 2 * a.n_extra + 1;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)a.n_picks * per) return;
    const int pk = (int)(i / per), off = (int)(i - (long long)pk * per) - a.n_extra;
    const int sta = a.pick_sta[pk];
    const int ph = a.pick_phase[pk];
    if (sta < 0 || sta >= a.S || (ph != 0 && ph != 1)) return;
    const double t = a.pick_t[pk];
    const int idx = (int)((t - a.tref0) / a.dt) + off;                 // :514-515, :534
    if (idx < 0 || idx >= a.n_time) return;                            // :537
    const double tv = t - (a.tref0 + (double)idx * a.dt);              // abs_time_ref[idx] = arange(...)[idx]
    const float val = (float)exp(-0.5 * tv * tv / (a.sigma * a.sigma));   // :545, cast at torch.Tensor(vals) :563
    atomicMax((int*)(a.emb + ((long long)ph * a.S + sta) * a.n_time + idx), __float_as_int(val));   // val >= 0
}

__global__ void k_embed_edges(EmbArgs a) {   // overflow guard: first / last sample of every series is zero (:565-568)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * a.S) return;
    a.emb[(long long)i * a.n_time] = 0.f;
    a.emb[(long long)i * a.n_time + a.n_time - 1] = 0.f;
}

__global__ void k_embed_gather(EmbArgs a) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.rows) return;
    const int sta = a.sta_of != nullptr ? a.sta_of[p] : (int)(p % a.S);
    const float2 tt = *(const float2*)(a.trv + p * 2);
    int ip = (int)((((double)tt.x + a.t0) - a.tref0) / a.dt);           // :605 (float64 arithmetic, truncation)
    int is = (int)((((double)tt.y + a.t0) - a.tref0) / a.dt);
    ip = min(max(ip, 0), a.n_time - 1);
    is = min(max(is, 0), a.n_time - 1);
    const float* ep = a.emb + (long long)sta * a.n_time;
    const float* es = a.emb + ((long long)a.S + sta) * a.n_time;
    f32x4 sl;
    sl.x = fmaxf(ep[ip], es[ip]);                                        // any-phase series = max(P, S)  :569, :612
    sl.y = fmaxf(ep[is], es[is]);                                        // :613
    sl.z = ep[ip];                                                       // :614
    sl.w = es[is];                                                       // :615
    if (a.sign_input) {      // :610-614. (The sample after a series' last one is the next series' first: both are zero, so is the value read there.)
        auto sg = [](float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); };
        const int ip1 = min(ip + 1, a.n_time - 1), is1 = min(is + 1, a.n_time - 1);
        sl.x *= sg(sl.x - fmaxf(ep[ip1], es[ip1]));
        sl.y *= sg(sl.y - fmaxf(ep[is1], es[is1]));
        sl.z *= sg(sl.z - ep[ip1]);
        sl.w *= sg(sl.w - es[is1]);
    }
    if (a.no_phase) { sl.z = 0.f; sl.w = 0.f; }
    f32x4 mk;
    mk.x = fabsf(sl.x) > 0.01f ? 1.f : 0.f; mk.y = fabsf(sl.y) > 0.01f ? 1.f : 0.f;      // :629
    mk.z = fabsf(sl.z) > 0.01f ? 1.f : 0.f; mk.w = fabsf(sl.w) > 0.01f ? 1.f : 0.f;
    *(f32x4*)(a.slice + p * 4) = sl;
    *(f32x4*)(a.mask + p * 4) = mk;
    if (a.xs != nullptr) {            // same rows as k_split_rows would produce from (sl, mk)
        const float v[8] = {sl.x, sl.y, sl.z, sl.w, mk.x, mk.y, mk.z, mk.w};
        const long long px = a.sta_inv != nullptr ? p - sta + a.sta_inv[sta] : p;
        if (a.sta_inv != nullptr) a.mm[px] = fmaxf(fmaxf(mk.x, mk.y), fmaxf(mk.z, mk.w));
        store_split_row(a.xs, a.rows, px, v);
    }
}

// de-pad rows of a workspace tensor for parity tests
// use_absolute_pos (config.yaml:92; module.py:1007): Slice gets locs[sta] / (3 scale_rel) and x_grid[src] / (3 scale_rel) appended
__global__ void k_abs_table(const float* __restrict__ pos, int n, float inv, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 4) return;
    const int r = i >> 2, k = i & 3;
    out[i] = k < 3 ? pos[r * 3 + k] * inv : 0.f;
}

// the two fp16 pieces of every row of an [n][4] scaled-position table, [n][2] x 8 B; `perm` (or null): row i = table row perm[i]
__global__ void k_abs_pieces(const float* __restrict__ tab, const int32_t* __restrict__ perm, int n, unsigned* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f32x4 v = *(const f32x4*)(tab + (size_t)(perm ? perm[i] : i) * 4);
    const unsigned a0 = cvt_pk_f16(v.x, v.y), b0 = cvt_pk_f16(v.z, 0.f);
    const unsigned a1 = cvt_pk_f16(sub_f16_lo(v.x, a0), sub_f16_hi(v.y, a0)), b1 = cvt_pk_f16(sub_f16_lo(v.z, b0), 0.f);
    *(u32x4*)(out + (size_t)i * 4) = u32x4{a0, b0, a1, b1};
}

// DataAggregationEdges (module.py:102-174, forward :1059-1072): every message carries phi(pos_j - pos_i) (3) and phi(|pos_j - pos_i|),
// phi(d) = sign(d) exp(-d^2 / (2 scale_rel^2)); after mean aggregation that is a STATIC 4-vector per node of a base graph.
__global__ void k_edge_feat(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int n,
                            const float* __restrict__ pos, float scale_rel, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int eb = rowptr[i], ee = rowptr[i + 1];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float inv = 1.f / (scale_rel * scale_rel);
    for (int e = eb; e < ee; ++e) {
        const int j = col[e];
        float d[4];
        d[0] = pos[j * 3] - pos[i * 3]; d[1] = pos[j * 3 + 1] - pos[i * 3 + 1]; d[2] = pos[j * 3 + 2] - pos[i * 3 + 2];
        d[3] = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sg = d[k] > 0.f ? 1.f : (d[k] < 0.f ? -1.f : 0.f);
            acc[k] += sg * expf(-0.5f * d[k] * d[k] * inv);
        }
    }
    const float w = ee > eb ? 1.f / (float)(ee - eb) : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) out[i * 4 + k] = acc[k] * w;
}
// ... and its Linear a per-node additive term: row n = {W1pos (30x4) m_n, 0, 0, W2pos (15x4) m_n, 0}
__global__ void k_edge_bias(const float* __restrict__ raw, int off1, int off2, const float* __restrict__ mpos, int n,
                            float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * 48) return;
    const int i = idx / 48, ch = idx - i * 48;
    const float* m = mpos + i * 4;
    float v = 0.f;
    if (ch < 30) { const float* wr = raw + off1 + ch * 4; v = wr[0] * m[0] + wr[1] * m[1] + wr[2] * m[2] + wr[3] * m[3]; }
    else if (ch >= 32 && ch < 47) { const float* wr = raw + off2 + (ch - 32) * 4; v = wr[0] * m[0] + wr[1] * m[1] + wr[2] * m[2] + wr[3] * m[3]; }
    out[idx] = v;
}

// Neighbour means on the implicit product graph for arbitrary row widths (association heads, module.py:389-403):
//   out_sta[(g,s)] = mean_k x_sta[(g, sta_nbr_k(s))],   out_src[(g,s)] = mean_k x_src[(src_nbr_k(g), s)]
// rows of CL * VW floats; CL lanes per node, every lane keeps up to 8 row chunks in flight; sums in edge order.
// With per-edge weights (w_sta / w_src non-null) the same kernel is the ADJOINT of the mean on the reversed graphs:
//   dx[j] = sum_{i : j in N(i)} g[i] / deg(i)   (genie_nbr_mean_bwd; edge lists = out-edges of j, weights 1 / in-degree of i)
template <int CL, int VW>       // CL lanes per row, VW floats per lane: rows of CL * VW floats (16 / 32 padded, or 30 unpadded)
__global__ __launch_bounds__(256) void k_nbr_mean(int S, int G, const int32_t* __restrict__ sta_rowptr, const int32_t* __restrict__ sta_col,
                                                  const int32_t* __restrict__ src_rowptr, const int32_t* __restrict__ src_col,
                                                  const float* __restrict__ x_sta, const float* __restrict__ x_src,
                                                  float* __restrict__ out_sta, float* __restrict__ out_src,
                                                  const float* __restrict__ w_sta = nullptr, const float* __restrict__ w_src = nullptr) {
    typedef float V __attribute__((ext_vector_type(VW)));
    constexpr int NPB_ = 256 / CL, RF = CL * VW;
    if ((int)threadIdx.x >= NPB_ * CL) return;
    const int cl = threadIdx.x % CL;
    const long long P = (long long)S * G;
    for (long long p = (long long)blockIdx.x * NPB_ + threadIdx.x / CL; p < P; p += (long long)gridDim.x * NPB_) {
        const int g = (int)(p / S), s = (int)(p - (long long)g * S);
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const float* x = which == 0 ? x_sta : x_src;
            float* out = which == 0 ? out_sta : out_src;
            if (x == nullptr) continue;
            const int32_t* col = which == 0 ? sta_col : src_col;
            const float* ew = which == 0 ? w_sta : w_src;
            const int eb = which == 0 ? sta_rowptr[s] : src_rowptr[g], ee = which == 0 ? sta_rowptr[s + 1] : src_rowptr[g + 1];
            V acc = 0.f;
            for (int e0 = eb; e0 < ee; e0 += 8) {
                V v[8];
                float wk[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = min(e0 + k, ee - 1);
                    const int j = col[e];
                    const long long row = which == 0 ? (long long)g * S + j : (long long)j * S + s;
                    v[k] = *(const V*)(x + row * RF + VW * cl);
                    wk[k] = ew ? ew[e] : 1.f;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (e0 + k < ee) acc += v[k] * wk[k];
            }
            const float w = ew ? 1.f : (ee > eb ? 1.f / (float)(ee - eb) : 0.f);
            *(V*)(out + p * RF + VW * cl) = acc * w;
        }
    }
}

// PReLU backward for the training path (one slope per call): dx = dy * (x >= 0 ? 1 : a), da = sum_{x < 0} dy * x. PyTorch's
// own backward materialises a full-size slope gradient and reduces it in a second pass (0.9 ms per [2M, 30] tensor); this
// is one pass plus a fixed-order two-level sum (deterministic).
constexpr int PRELU_BLOCKS = 2048;
__global__ __launch_bounds__(256) void k_prelu_bwd(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ slope,
                                                   long long n, float* __restrict__ dx, float* __restrict__ partial) {
    const float a = slope[0];
    float acc = 0.f;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 xv = ((const f32x4*)x)[i], gv = ((const f32x4*)dy)[i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool neg = xv[k] < 0.f;
            o[k] = neg ? gv[k] * a : gv[k];
            acc += neg ? gv[k] * xv[k] : 0.f;
        }
        ((f32x4*)dx)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {      // tail elements
        const long long i = (n4 << 2) + threadIdx.x;
        const bool neg = x[i] < 0.f;
        dx[i] = neg ? dy[i] * a : dy[i];
        acc += neg ? dy[i] * x[i] : 0.f;
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void k_prelu_bwd_sum(const float* __restrict__ partial, int nb, float* __restrict__ da) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) da[0] = red[0];
}

// Weight and bias gradients of a per-node Linear over N rows (training path): dW[m][k] = sum_n dy[n][m] x[n][k],
// db[m] = sum_n dy[n][m], M <= 32, K <= 64 KC. The library GEMM for this shape (a [M, N] x [N, K] product with N = 2M rows) runs
// at 1-3 ms plus a separate 0.25 ms bias reduction; this reads x and dy once. Wave w owns outputs m in [8w, 8w+8), lane l the
// columns k = l + 64 c; dy rows are staged through LDS and read back as wave-uniform broadcasts. Partials per workgroup are
// summed by k_linear_bwd_sum in a fixed order.
constexpr int LBW_ROWS = 32, LBW_BLOCKS = 1024;
template <int KC>
__global__ __launch_bounds__(256) void k_linear_bwd_w(const float* __restrict__ x, const float* __restrict__ dy, long long N, int K, int M,
                                                      int ldy, float* __restrict__ partial) {
    __shared__ float sdy[LBW_ROWS][32];
    __shared__ float sb[8][32];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sm = threadIdx.x & 31, sr = threadIdx.x >> 5;          // staging role: column sm of rows sr + 8 q
    const int smc = min(sm, M - 1);
    int kc[KC];                                                       // lanes beyond K re-read column K-1; their sums are dropped
#pragma unroll
    for (int c = 0; c < KC; ++c) kc[c] = min(lane + 64 * c, K - 1);
    float acc[KC][8];
    float accb = 0.f;
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[c][j] = 0.f;
    const long long ntile = (N + LBW_ROWS - 1) / LBW_ROWS;
    for (long long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long long n0 = t * LBW_ROWS;
        const int nr = (int)min((long long)LBW_ROWS, N - n0);        // rows beyond N: dy staged as 0, x re-reads the last row
        const float* __restrict__ xt = x + n0 * K;
        const float* __restrict__ dt = dy + n0 * ldy;            // M <= 32 columns of rows that are ldy floats apart
        float st[LBW_ROWS / 8];
#pragma unroll
        for (int q = 0; q < LBW_ROWS / 8; ++q) {
            const int r = sr + 8 * q;
            const float v = dt[min(r, nr - 1) * ldy + smc];
            st[q] = (sm < M && r < nr) ? v : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < LBW_ROWS / 8; ++q) {
            sdy[sr + 8 * q][sm] = st[q];
            accb += st[q];
        }
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < LBW_ROWS; rb += 8) {
            float xv[8][KC];
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < KC; ++c) xv[r][c] = xt[min(rb + r, nr - 1) * K + kc[c]];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const f32x4 d0 = *(const f32x4*)&sdy[rb + r][8 * w], d1 = *(const f32x4*)&sdy[rb + r][8 * w + 4];
#pragma unroll
                for (int c = 0; c < KC; ++c) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[c][j] += d0[j] * xv[r][c];
                        acc[c][4 + j] += d1[j] * xv[r][c];
                    }
                }
            }
        }
    }
    float* out = partial + (size_t)blockIdx.x * (32 * 64 * KC + 32);
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) out[(8 * w + j) * (64 * KC) + 64 * c + lane] = acc[c][j];
    sb[sr][sm] = accb;
    __syncthreads();
    if (threadIdx.x < 32) {
        float v = sb[0][sm];
#pragma unroll
        for (int q = 1; q < 8; ++q) v += sb[q][sm];
        out[32 * 64 * KC + sm] = v;
    }
}
__global__ __launch_bounds__(256) void k_linear_bwd_sum(const float* __restrict__ partial, int nb, int KC, int K, int M,
                                                        float* __restrict__ dW, float* __restrict__ db) {
    // 32 outputs x 8 slices of the workgroup partials per block; slices combined in a fixed order
    __shared__ float red[8][32];
    const int per = 32 * 64 * KC + 32;
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < per) {
        const int per_sl = (nb + 7) / 8, b0 = sl * per_sl, b1 = min(nb, b0 + per_sl);
        int b = b0;
        for (; b + 4 <= b1; b += 4) {
            a0 += partial[(size_t)b * per + e];
            a1 += partial[(size_t)(b + 1) * per + e];
            a2 += partial[(size_t)(b + 2) * per + e];
            a3 += partial[(size_t)(b + 3) * per + e];
        }
        for (; b < b1; ++b) a0 += partial[(size_t)b * per + e];
    }
    red[sl][el] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl != 0 || e >= per) return;
    float v = red[0][el];
#pragma unroll
    for (int q = 1; q < 8; ++q) v += red[q][el];
    if (e < 32 * 64 * KC) {
        const int m = e / (64 * KC), k = e % (64 * KC);
        if (m < M && k < K) dW[m * K + k] = v;
    } else if (db && e - 32 * 64 * KC < M) db[e - 32 * 64 * KC] = v;
}

// XCC (XCD) id and CU id of the CU a workgroup runs on (genie_where_am_i): workgroup b of a launch lands on XCD b % 8
__global__ void k_where_am_i(int* __restrict__ out) {
    int xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(20, 0, 4)" : "=s"(xcc));       // HW_REG_XCC_ID
    asm volatile("s_getreg_b32 %0, hwreg(4, 0, 32)" : "=s"(hwid));      // HW_REG_HW_ID
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}

// ------------------------------------------------------------------------------------------------
// Exact k-nearest-neighbour search on the device: the `knn(x_context / 1000, x_query / 1000, k)` calls of the
// reference -- SpatialAttention's query edges (module.py:282; a NEW 112 000-point query set per candidate in the refine pass,
// process_continuous_days.py:926-980) and the base graphs of the product graph (`knn(x/1000, x/1000, k + 1)` +
// `remove_self_loops`, process_utils.py:718-719). Brute force in fp64 on the fp32 coordinates themselves (the
// common 1 / 1000 scale does not change the order; coordinate differences of fp32 values are exact in fp64) (3-D, n_context
// is 10^4..10^5: 10^9 pair distances = a millisecond): one wave per query, lane l scans candidates l, l + 64, ... keeping its K
// best in registers (sorted, ties by smaller index), then K rounds of a wave-wide lexicographic (distance, index) minimum pop
// the global K best in order. exclude_self: skip candidate == query id (query set = context set).
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void k_knn(const float* __restrict__ xc, int nc, const float* __restrict__ xq, int nq, int k,
                                             int exclude_self, int32_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const double q0 = (double)xq[qi * 3 + 0], q1 = (double)xq[qi * 3 + 1], q2 = (double)xq[qi * 3 + 2];
    double bd[K];
    int bi[K];
#pragma unroll
    for (int t = 0; t < K; ++t) { bd[t] = __builtin_inf(); bi[t] = 0x7fffffff; }
    // candidates four at a time: their twelve coordinate loads are in flight together (112 000 queries x 10 000 points 4.3 -> 3.7 ms,
    // 50 000 x 50 000 at k = 15 14.7 -> 10.4 ms, the 8 source queries of forward_fixed 210 -> 173 us: tools/knn_time.py, ff_time.py)
    constexpr int UB = 4;
    for (int c0 = lane; c0 < nc; c0 += 64 * UB) {
        float cx[UB][3];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int c = min(c0 + 64 * u, nc - 1);
            cx[u][0] = xc[c * 3 + 0]; cx[u][1] = xc[c * 3 + 1]; cx[u][2] = xc[c * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int c = c0 + 64 * u;
            if (c >= nc || (exclude_self && c == qi)) continue;
            const double d0 = q0 - (double)cx[u][0], d1 = q1 - (double)cx[u][1], d2 = q2 - (double)cx[u][2];   // exact
            double d = __dadd_rn(__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), __dmul_rn(d2, d2));
            int id = c;
            if (d < bd[K - 1]) {          // candidates arrive in increasing index order: a tie never displaces an earlier entry
#pragma unroll
                for (int t = 0; t < K; ++t) {
                    if (d < bd[t] || (d == bd[t] && id < bi[t])) {       // lexicographic (distance, index): a displaced entry that ties with
                                                                          // the next slot goes in front of it (it has the smaller index)
                        const double td = bd[t]; const int ti = bi[t];
                        bd[t] = d; bi[t] = id; d = td; id = ti;
                    }
                }
            }
        }
    }
    for (int r = 0; r < k; ++r) {
        double md = bd[0];
        int mi = bi[0];
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const double od = __shfl_xor(md, s);
            const int oi = __shfl_xor(mi, s);
            if (od < md || (od == md && oi < mi)) { md = od; mi = oi; }
        }
        if (bi[0] == mi && bd[0] == md) {        // the owner pops its head (indices are unique across lanes)
#pragma unroll
            for (int t = 0; t + 1 < K; ++t) { bd[t] = bd[t + 1]; bi[t] = bi[t + 1]; }
            bd[K - 1] = __builtin_inf(); bi[K - 1] = 0x7fffffff;
        }
        if (lane == 0) out[(long long)qi * k + r] = mi == 0x7fffffff ? -1 : mi;
    }
}

// A handful of queries (the one spatial query and the one candidate source of the association pass, process_continuous_days.py:1052;
// the 4-8 source queries of a training sample): with a wave per query ONE wave walked the whole context -- 134-170 us per call at 10 000
// grid nodes, twice per forward_fixed -- so a query gets a workgroup of 16 waves instead: every lane scans a 1024-th of the context
// (same exact arithmetic and per-lane lists as k_knn), every wave pops its K best into LDS, wave 0 merges the 16 lists. Same table.
template <int K>
__global__ __launch_bounds__(1024) void k_knn_b(const float* __restrict__ xc, int nc, const float* __restrict__ xq, int nq, int k,
                                                int exclude_self, int32_t* __restrict__ out) {
    __shared__ double sd[16 * K];
    __shared__ int si[16 * K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qi = blockIdx.x;
    const double q0 = (double)xq[qi * 3 + 0], q1 = (double)xq[qi * 3 + 1], q2 = (double)xq[qi * 3 + 2];
    double bd[K];
    int bi[K];
#pragma unroll
    for (int t = 0; t < K; ++t) { bd[t] = __builtin_inf(); bi[t] = 0x7fffffff; }
    auto insert = [&](double d, int id) {
        if (d < bd[K - 1] || (d == bd[K - 1] && id < bi[K - 1])) {
#pragma unroll
            for (int t = 0; t < K; ++t) {
                if (d < bd[t] || (d == bd[t] && id < bi[t])) {
                    const double td = bd[t]; const int ti = bi[t];
                    bd[t] = d; bi[t] = id; d = td; id = ti;
                }
            }
        }
    };
    auto pop = [&](double& md, int& mi) {          // the wave's smallest (distance, index); its owner drops it
        md = bd[0]; mi = bi[0];
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const double od = __shfl_xor(md, s);
            const int oi = __shfl_xor(mi, s);
            if (od < md || (od == md && oi < mi)) { md = od; mi = oi; }
        }
        if (bi[0] == mi && bd[0] == md && mi != 0x7fffffff) {
#pragma unroll
            for (int t = 0; t + 1 < K; ++t) { bd[t] = bd[t + 1]; bi[t] = bi[t + 1]; }
            bd[K - 1] = __builtin_inf(); bi[K - 1] = 0x7fffffff;
        }
    };
    for (int c = threadIdx.x; c < nc; c += 1024) {
        if (exclude_self && c == qi) continue;
        const double d0 = q0 - (double)xc[c * 3 + 0], d1 = q1 - (double)xc[c * 3 + 1], d2 = q2 - (double)xc[c * 3 + 2];   // exact
        insert(__dadd_rn(__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), __dmul_rn(d2, d2)), c);
    }
    for (int r = 0; r < K; ++r) {
        double md; int mi;
        pop(md, mi);
        if (lane == 0) { sd[wave * K + r] = md; si[wave * K + r] = mi; }
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int t = 0; t < K; ++t) { bd[t] = __builtin_inf(); bi[t] = 0x7fffffff; }
    for (int e = lane; e < 16 * K; e += 64)
        if (si[e] != 0x7fffffff) insert(sd[e], si[e]);
    for (int r = 0; r < k; ++r) {
        double md; int mi;
        pop(md, mi);
        if (lane == 0) out[(long long)qi * k + r] = mi == 0x7fffffff ? -1 : mi;
    }
}

// The same search with ONE LANE per query, for large query sets (the refine pass's 112 000-point clouds, process_continuous_days.py:929:
// with a wave per query every candidate step paid a lane's insertion -- 156 candidates per lane, a quarter of them enter its list -- and
// the 64 partial lists were merged afterwards: 4.3 ms of the pass's 4.9 ms per source). Here a lane scans every candidate (tiles of the
// context staged in LDS, read by all lanes at the same address: a broadcast), its list only ever holds the K best so far, and an
// insertion happens ~K ln(n / K) times per query: after the first few hundred candidates a wave rarely leaves the distance-and-compare
// path. That path is fp32 with a guard band (a candidate is dropped only if its fp32 distance exceeds the list's worst by more than
// 1e-5 relative: fp32 rounding is < 1e-6); everything that passes is decided on the EXACT distance (fp64 of the fp32 coordinates,
// as k_knn): same neighbours in the same order (distance, then index), bit for bit.
constexpr int KNN_TILE = 2048;
template <int K>
__global__ __launch_bounds__(256) void k_knn_t(const float* __restrict__ xc, int nc, const float* __restrict__ xq, int nq, int k,
                                               int exclude_self, int32_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float tile[KNN_TILE * 4];
    const int qi = blockIdx.x * 256 + threadIdx.x;
    const int qc = qi < nq ? qi : nq - 1;
    const float f0 = xq[qc * 3 + 0], f1 = xq[qc * 3 + 1], f2 = xq[qc * 3 + 2];
    const double q0 = (double)f0, q1 = (double)f1, q2 = (double)f2;
    // The K best so far UNSORTED, with the slot of the worst one (largest (distance, index)) tracked: an accepted candidate overwrites
    // that slot and the worst is found again (K compares) -- about half the instructions of keeping the list sorted by a K-step
    // bubble, and in this kernel the accept path is what a wave mostly pays for (some lane accepts in ~a quarter of the steps). The
    // list is sorted once per query at the end.
    double bd[K];
    int bi[K];
#pragma unroll
    for (int t = 0; t < K; ++t) { bd[t] = __builtin_inf(); bi[t] = 0x7fffffff - t; }      // (placeholders with distinct indices; slot 0 is the worst)
    double wd = __builtin_inf();
    int wi = 0x7fffffff;                             // the worst entry: candidates arrive in increasing index order, so a tie never displaces it
    float thr = __builtin_inff();                    // fp32 guard of wd
    // tiles of the context staged in LDS, read by all lanes at the same address (a broadcast). (Reading the wave-uniform candidate
    // through the scalar cache instead -- three 16-byte scalar loads per four candidates, no staging, no barrier -- measured the same:
    // 2.44 against 2.38 ms at 112 000 x 10 000.)
    for (int c0 = 0; c0 < nc; c0 += KNN_TILE) {
        const int n = min(KNN_TILE, nc - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) {
            const float* p = xc + (long long)(c0 + i) * 3;
            *(f32x4*)(tile + i * 4) = f32x4{p[0], p[1], p[2], 0.f};
        }
        __syncthreads();
        for (int i = 0; i < n; ++i) {
            const f32x4 cv = *(const f32x4*)(tile + i * 4);
            const float e0 = f0 - cv.x, e1 = f1 - cv.y, e2 = f2 - cv.z;
            const float d32 = e0 * e0 + e1 * e1 + e2 * e2;
            if (!(d32 <= thr)) continue;
            const int c = c0 + i;
            if (exclude_self && c == qi) continue;
            const double d0 = q0 - (double)cv.x, d1 = q1 - (double)cv.y, d2 = q2 - (double)cv.z;                       // exact
            const double d = __dadd_rn(__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), __dmul_rn(d2, d2));
            if (!(d < wd)) continue;
            double nd = -1.0;                            // replace the worst, find the new worst (lexicographic maximum)
            int ni = 0;
#pragma unroll
            for (int t = 0; t < K; ++t) {
                if (bi[t] == wi) { bd[t] = d; bi[t] = c; }
                if (bd[t] > nd || (bd[t] == nd && bi[t] > ni)) { nd = bd[t]; ni = bi[t]; }
            }
            wd = nd; wi = ni;
            thr = (float)wd;
            thr = thr + thr * 1e-5f;                    // inf stays inf
        }
    }
    // sort by (distance, index): K is small, once per query
#pragma unroll
    for (int a = 0; a < K - 1; ++a) {
#pragma unroll
        for (int t = 0; t < K - 1 - a; ++t) {
            if (bd[t + 1] < bd[t] || (bd[t + 1] == bd[t] && bi[t + 1] < bi[t])) {
                const double td = bd[t]; const int ti = bi[t];
                bd[t] = bd[t + 1]; bi[t] = bi[t + 1]; bd[t + 1] = td; bi[t + 1] = ti;
            }
        }
    }
    if (qi < nq)
        for (int r = 0; r < k; ++r) out[(long long)qi * k + r] = bi[r] >= 0x7fffffff - K ? -1 : bi[r];
}

// ------------------------------------------------------------------------------------------------
// Downstream reduction of the apply loop on the device: the stacked query output Out_2 [rows = queries,
// cols = time steps] never leaves the GPU whole. MODE 0: entries above a threshold, `np.where(Out_2 > 0.01)`
// (process_continuous_days.py:812-813). MODE 1: the local maxima of every row that reach `height`, i.e. the first two steps
// of `scipy.signal.find_peaks(Out[i, :], height = thresh, ...)` (:846; scipy's `_local_maxima_1d`: a sample or the midpoint
// of a flat run that is strictly higher than both neighbours, never the first or last sample; then `x >= height`).
// One workgroup per row, chunks of 256 columns, selected entries written in column order at `offsets[row]` + rank
// (two passes: COUNT fills counts[row], the caller scans them; the second pass fills) -> row-major order like np.where.
// ------------------------------------------------------------------------------------------------
template <int MODE, bool COUNT>
__global__ __launch_bounds__(256) void k_row_select(const float* __restrict__ x, long long cols, float thr, int32_t* __restrict__ counts,
                                                    const long long* __restrict__ offsets, int32_t* __restrict__ out_row,
                                                    int32_t* __restrict__ out_col, float* __restrict__ out_val) {
    __shared__ int wsum[4];
    const long long row = blockIdx.x;
    const float* xr = x + row * cols;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long base = COUNT ? 0 : offsets[row];
    int total = 0;
    for (long long c0 = 0; c0 < cols; c0 += 256) {
        const long long i = c0 + threadIdx.x;
        bool sel = false;
        long long col = i;
        float v = 0.f;
        if (i < cols) {
            v = xr[i];
            if (MODE == 0) {
                sel = v > thr;
            } else if (i >= 1 && i + 1 < cols && v >= thr && xr[i - 1] < v) {      // rising edge into a candidate (flat) top
                long long e = i + 1;
                while (e < cols - 1 && xr[e] == v) ++e;
                if (xr[e] < v) { sel = true; col = (i + e - 1) / 2; }
            }
        }
        const unsigned long long b = __ballot(sel);
        const int rank = __popcll(b & ((1ull << lane) - 1ull)), wtot = __popcll(b);
        if (lane == 0) wsum[wave] = wtot;
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (k < wave) before += wsum[k]; all += wsum[k]; }
        if (!COUNT && sel) {
            const long long o = base + total + before + rank;
            out_row[o] = (int32_t)row; out_col[o] = (int32_t)col; out_val[o] = v;
        }
        total += all;
        __syncthreads();
    }
    if (COUNT && threadIdx.x == 0) counts[row] = total;
}

// Product-level CSRs of the IRREGULAR product graph of `use_subgraph` (process_utils.py:744-849) on the device. The product
// nodes are the (station, source) pairs sorted by (source, station): node n = (pair_sta[n], pair_src[n]), source node g owns
// nodes [seg[g], seg[g + 1]). In-edges of node n (the reference's `subgraph(...)` calls, :824-839):
//   station graph: m -> n for every base edge j -> pair_sta[n] whose pair (j, pair_src[n]) exists, in base edge order;
//   source graph:  m -> n for every base edge g' -> pair_src[n] whose pair (pair_sta[n], g') exists, in base edge order.
// One thread per product node; a pair is looked up by binary search in the station list of its source node. FILL = false
// counts the in-edges, FILL = true writes them behind the node's row pointer.
template <bool FILL>
__global__ __launch_bounds__(256) void k_subgraph_csr(const int32_t* __restrict__ pair_sta, const int32_t* __restrict__ pair_src,
                                                     long long N, const int32_t* __restrict__ seg,
                                                     const int32_t* __restrict__ sta_rowptr, const int32_t* __restrict__ sta_col,
                                                     const int32_t* __restrict__ src_rowptr, const int32_t* __restrict__ src_col,
                                                     int32_t* __restrict__ cnt_sta, int32_t* __restrict__ cnt_src,
                                                     const int32_t* __restrict__ p_sta_rowptr, const int32_t* __restrict__ p_src_rowptr,
                                                     int32_t* __restrict__ p_sta_col, int32_t* __restrict__ p_src_col) {
    const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int s = pair_sta[n], g = pair_src[n];
    auto find = [&](int sta, int src) -> int {          // product node of the pair (sta, src), or -1
        int lo = seg[src], hi = seg[src + 1];
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (pair_sta[mid] < sta) lo = mid + 1; else hi = mid;
        }
        return (lo < seg[src + 1] && pair_sta[lo] == sta) ? lo : -1;
    };
    int c1 = 0, c2 = 0;
    int32_t* o1 = FILL ? p_sta_col + p_sta_rowptr[n] : nullptr;
    int32_t* o2 = FILL ? p_src_col + p_src_rowptr[n] : nullptr;
    for (int e = sta_rowptr[s]; e < sta_rowptr[s + 1]; ++e) {
        const int m = find(sta_col[e], g);
        if (m >= 0) { if (FILL) o1[c1] = m; ++c1; }
    }
    for (int e = src_rowptr[g]; e < src_rowptr[g + 1]; ++e) {
        const int m = find(s, src_col[e]);
        if (m >= 0) { if (FILL) o2[c2] = m; ++c2; }
    }
    if (!FILL) { cnt_sta[n] = c1; cnt_src[n] = c2; }
}

// owner[p] = g for every product node p of the row range [seg_rowptr[g], seg_rowptr[g + 1]) (irregular product graphs)
__global__ void k_seg_owner(const int32_t* __restrict__ seg_rowptr, int G, int32_t* __restrict__ owner) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    for (int p = seg_rowptr[g]; p < seg_rowptr[g + 1]; ++p) owner[p] = g;
}

__global__ void k_export(const float* __restrict__ src, long long rows, int pitch, int ncol, float* __restrict__ dst,
                         const int32_t* __restrict__ sta_user, int S, int np) {
    // padded rows are [15 valid, 1 pad] blocks (c has two of them, wu / wv one); rows in station processing order -> caller's order
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * ncol) return;
    const long long r = idx / ncol;
    const int cc = (int)(idx % ncol);
    const int off = ncol == 15 ? cc : (cc / 15) * 16 + cc % 15;   // c = [c1 15,0 | c2 15,0]
    long long ru = r;
    if (sta_user != nullptr) {
        const long long g = r / S;
        ru = g * S + sta_user[(int)(r - g * S)];
    }
    if (np) {      // node-planar rows (DaArgs.np): chunk off / 4 of station s at [g][chunk][s] x 4 floats inside the node's block
        const long long g = r / S;
        const long long s = r - g * S;
        dst[ru * ncol + cc] = src[g * S * pitch + ((long long)(off >> 2) * S + s) * 4 + (off & 3)];
        return;
    }
    dst[ru * ncol + cc] = src[r * pitch + off];
}
// rows [G][S][width] from station processing order to the caller's order (debug outputs) or, with `inv`, the other way (tables)
__global__ void k_permute_sta_rows(const float* __restrict__ src, long long rows, int width, const int32_t* __restrict__ map, int S,
                                   float* __restrict__ dst) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * width) return;
    const long long r = idx / width;
    const int cc = (int)(idx - r * width);
    const long long g = r / S;
    dst[(g * S + map[(int)(r - g * S)]) * width + cc] = src[idx];
}


// ---- Cartesian-structure check of the reference's product edge lists (process_utils.py:720-721) -----------------------------------------
// A_in_sta [2][G * e_sta] must be `A_sta_sta.repeat(1, G) + S * arange(G).repeat_interleave(e_sta)` and A_in_src [2][S * e_src]
// `S * A_src_src.repeat(1, S) + arange(S).repeat_interleave(e_src)`: every entry is compared with the one predicted from the list's own
// first block (the base graph), which is range-checked as well. flags[0] |= 1: A_in_sta is not Cartesian, |= 2: A_in_src is not. One pass
// over the 46 M int64 pairs of config 3 (0.74 GB: HBM-bound) where three materialised int64 copies + torch.equal took 4.4 ms.
__global__ void __launch_bounds__(256) k_product_check(const long long* __restrict__ A1, long long E1, int e_sta, const long long* __restrict__ A2,
                                                        long long E2, int e_src, int S, int G, int* __restrict__ flags) {
    const long long n = E1 + E2;
    int bad = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (i < E1) {
            const long long g = i / e_sta, k = i - g * e_sta;
            const long long b0 = A1[k], b1 = A1[E1 + k];
            const bool ok = b0 >= 0 && b0 < S && b1 >= 0 && b1 < S && A1[i] == b0 + g * S && A1[E1 + i] == b1 + g * S;
            bad |= ok ? 0 : 1;
        } else {
            const long long e = i - E1, s = e / e_src, k = e - s * e_src;
            const long long b0 = A2[k], b1 = A2[E2 + k];
            const bool ok = b0 >= 0 && b1 >= 0 && b0 % S == 0 && b1 % S == 0 && b0 / S < G && b1 / S < G && A2[e] == b0 + s && A2[E2 + e] == b1 + s;
            bad |= ok ? 0 : 2;
        }
    }
    if (bad) atomicOr(flags, bad);
}

// Range check of an index list without a host round trip: bit `bit` of the host-mapped flag word is set when some idx[i] lies outside
// [lo, hi) (genie_index_check -> genie_index_flags). The callers clamp such indices for their own use; the host raises at its next call.
__global__ void __launch_bounds__(256) k_index_check(const long long* __restrict__ idx, long long n, long long lo, long long hi, unsigned bit,
                                                      unsigned* __restrict__ flag) {
    bool bad = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long v = idx[i];
        bad |= v < lo || v >= hi;
    }
    if (bad) __hip_atomic_fetch_or(flag, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

