// Part of genie_hip.hip (one translation unit, included inside its anonymous namespace): the backward passes of the P-sized front (k_train_b2 / b1 / b0, k_train_reduce).

// ------------------------------------------------------------------------------------------------
// Backward of DataAggregation + Bipartite_ReadIn (training step, train_GENIE_model.py:1843-1861).
// The training forward is the generic stage kernels with `save` set (pre-activations of h0, h1, u, v, x_latent and the
// Bipartite message: 14 blocks of 16 floats per product node); the backward mirrors the stages in reverse order:
//   k_train_b2: d(station sum)[g] -> dz (message) -> dx_latent = fc1[:, 0:30]^T dz -> do = dx_latent PReLU2'(o)      store do
//   k_train_b1: transposed means of do1 / do2 (reversed base graphs) -> du, dv -> dh1 -> dt = dh1 PReLU1'(t)           store dt, dh0_local
//   k_train_b0: transposed means of dt1 / dt2 -> dh0 -> dz0
// Weight gradients are accumulated INSIDE the passes: dW[out, in] = sum over nodes dY[out] X[in] is an MFMA whose contraction
// runs over the 16 nodes of a tile (operands transposed through a per-wave LDS scratch), into register accumulators that live
// across all tiles of a wave; the terms that multiply a neighbour mean use the adjoint identity
//   sum_p dY[p] (x) mean_{k in N(p)} x[k] = sum_k (transposed mean of dY)[k] (x) x[k],
// so no mean is stored. Bias and PReLU-slope gradients are per-lane running sums. Every wave writes its partials once; a
// fixed-order reduction over the waves (k_train_reduce) makes the result run-to-run deterministic.
// ------------------------------------------------------------------------------------------------
constexpr int GR_DO = 0, GR_DT = 2, GR_DH0 = 6, GR_BLOCKS = 8;      // gradient rows kept between the passes: [GR_*][P][16]

struct AccDesc { int32_t mat_off, ld, row0, nrows, col0, ncols, n0, pad; };   // dW block: D[i][n] -> W[row0 + i][col0 + n], n0 <= n < ncols
struct VecDesc { int32_t off, row0, nrows, stride; };                // bias block: sum dY[i] -> b[(row0 + i) * stride]

struct TrArgs {
    int S, G, T, seg, nxcd;
    long long P;
    const int32_t* order;
    const int32_t* r_sta_rowptr; const int32_t* r_sta_col; const float* r_sta_w;    // reversed base graphs (out-edges, 1 / in-degree)
    const int32_t* r_src_rowptr; const int32_t* r_src_col; const float* r_src_w;
    const int2* r_sta_cw; const int2* r_src_cw;    // the same edges as (column, weight bits) pairs
    const int32_t* ptile;        // irregular product graph: processing order of the 16-node tiles (PtileIter), or null
    const int32_t* src_of;       // irregular product graph (PCSR kernels): source node of every product node; the r_* arrays are then
                                 // the reversed PRODUCT-level graphs (indexed by product node, columns = product-node ids)
    const float* slice; const float* mask; const float* edge_attr;
    const float* save; float* gr;
    const float* dr;             // [G][32] gradient of the per-source-node station sum (Bipartite, before fc2)
    const float* packed;
    float* part;                 // per-wave partials: [wave][n_acc * 256 + n_vec * 16 + 16]
    int n_acc, n_vec;
    int sv_t, sv_up, sv_vp;      // k_train_b1: blocks of `save` holding the pre-activations of h1 / u / v (SV_T, SV_UP, SV_VP, or the
                                 // association phase's AV_T, AV_UV, AV_UV + 2)
    const float* pg;             // association phase (k_train_b1<true>, k_as_*): [G][AS_PG] per-source-node terms, pg[31] = mask1[g]
    const float* x_latent;       // association phase: [P, 30] DataAggregation output (an input of init_trns there)
    int abl;                     // GENIE_TUNING builds: ablation bits of k_train_b1 (tools/train_abl.sh)
    int store_dz0;               // k_train_b0: keep dz0 in the GR_DH0 blocks (read by the static-term gradients of use_absolute_pos)
    float* zsum;                 // k_as_b0: [G * T][32] per-tile station sums of d z1 (-> d y_latent, fc1's y_latent columns)
};

__device__ __forceinline__ f32x4 ldb(const float* buf, int blk, long long P, long long p, int q) {
    return *(const f32x4*)(buf + ((size_t)blk * P + p) * 16 + 4 * q);
}
__device__ __forceinline__ void stb(float* buf, int blk, long long P, long long p, int q, f32x4 v) {
    *(f32x4*)(buf + ((size_t)blk * P + p) * 16 + 4 * q) = v;
}
// The same rows addressed as `global_load v, voffset, s[base]`: the buffer's base stays a scalar register pair and the byte offset
// (block x P x 64 + row x 64 + 16 q) is ONE 32-bit vector value, where `ldb` made the compiler build a 64-bit address per load (a shift, a
// 64-bit multiply-add and a 64-bit add: a third of the backward passes' vector instructions were address arithmetic, and these passes are
// bound by their vector-instruction stream: DESIGN.md section 5, round 5). Valid while every block offset fits 32 bits (O32 launches:
// 14 x P x 64 B < 4 GiB, i.e. P < 4.79 M product nodes; config 3 has 2 M).
typedef const __attribute__((address_space(1))) char* gbyte_p;
typedef const __attribute__((address_space(1))) f32x4* grow_p;
typedef __attribute__((address_space(1))) f32x4* groww_p;
__device__ __forceinline__ unsigned long long sbase(const void* b) {
    unsigned long long r = (unsigned long long)b;
    asm volatile("" : "+s"(r));
    return r;
}
__device__ __forceinline__ f32x4 ldo(unsigned long long base, unsigned off) { return *(grow_p)((gbyte_p)base + off); }
__device__ __forceinline__ void sto(unsigned long long base, unsigned off, f32x4 v) {
    *(groww_p)((__attribute__((address_space(1))) char*)base + off) = v;
}
__device__ __forceinline__ f32x4 dprelu4(f32x4 x, float s) {     // PReLU'(x): 1 for x > 0, the slope otherwise
    return f32x4{x.x > 0.f ? 1.f : s, x.y > 0.f ? 1.f : s, x.z > 0.f ? 1.f : s, x.w > 0.f ? 1.f : s};
}
__device__ __forceinline__ float negsum4(f32x4 g, f32x4 x) {    // sum of g * min(x, 0): the slope gradient of PReLU
    return g.x * fminf(x.x, 0.f) + g.y * fminf(x.y, 0.f) + g.z * fminf(x.z, 0.f) + g.w * fminf(x.w, 0.f);
}
// V[ch 4q + r][node j] held by lane (j, q) -> vt[s] = V[ch j][node 4s + q]: the operand form of a node-contracting MFMA.
// Through a per-wave LDS scratch of 256 floats, [node][16 channels]: one 16-byte write per lane (the 64 lanes cover the scratch
// contiguously), four 4-byte reads (row 4s + q, column j: the four q groups hit disjoint banks). The LDS executes the DS instructions of ONE
// wave in issue order, so neither the write -> read dependence between lanes nor the reuse of the scratch by the next transposition
// needs a counter drain (three `s_waitcnt lgkmcnt(0)` per transposition until round 4: seventeen transpositions per tile in
// k_train_b1 were ~17 % of its time); the two compiler barriers only keep the compiler from moving the accesses across each other,
// and the reads are waited for where their values are first used.
__device__ __forceinline__ f32x4 tr16(f32x4 v, float* sc, int j, int q) {
    asm volatile("" ::: "memory");
    *(f32x4*)(sc + j * 16 + 4 * q) = v;
    asm volatile("" ::: "memory");
    f32x4 t;
#pragma unroll
    for (int s = 0; s < 4; ++s) t[s] = sc[(4 * s + q) * 16 + j];
    asm volatile("" ::: "memory");
    return t;
}
__device__ __forceinline__ f32x4 outer16(f32x4 acc, f32x4 at, f32x4 bt) {       // acc[out][in] += sum over the tile's nodes
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = MFMA16(at[s], bt[s], acc);
    return acc;
}
// transposed mean: sum over the out-edges e of `node` of w_e * rows[col_e], rows of 16 floats addressed by `rowof(block, col)`,
// for NB row blocks at once. The edges are taken EB at a time (indices, weights and the EB x NB rows of a batch are all in
// flight together; a lane past its last edge re-reads edge 0 with weight zero): with one index -> row load chain per edge the
// backward passes spent ~80 % of their time waiting on these gathers (one wave per SIMD, nothing to switch to). The sum keeps
// the edge order.
template <int NB, int EB = 4, typename F>
__device__ __forceinline__ void tmean_n(const int32_t* __restrict__ rp, const int32_t* __restrict__ col, const float* __restrict__ w,
                                        int node, bool uniform, F rowof, f32x4 (&out)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) out[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    int eb = rp[node], ee = rp[node + 1];
    if (uniform) { eb = __builtin_amdgcn_readfirstlane(eb); ee = __builtin_amdgcn_readfirstlane(ee); }
    for (int e = eb; uniform ? (e < ee) : (bool)__any(e < ee); e += EB) {
        int c[EB];
        float ww[EB];
#pragma unroll
        for (int k = 0; k < EB; ++k) {
            const bool ok = e + k < ee;
            const int ei = ok ? e + k : 0;
            c[k] = col[ei];
            ww[k] = ok ? w[ei] : 0.f;
        }
        f32x4 r[NB][EB];
#pragma unroll
        for (int k = 0; k < EB; ++k)
#pragma unroll
            for (int b = 0; b < NB; ++b) r[b][k] = rowof(b, c[k]);
#pragma unroll
        for (int k = 0; k < EB; ++k)
#pragma unroll
            for (int b = 0; b < NB; ++b) out[b] += r[b][k] * ww[k];
    }
}
template <typename F>
__device__ __forceinline__ f32x4 tmean(const int32_t* rp, const int32_t* col, const float* w, int node, bool uniform, F rowof) {
    f32x4 o[1];
    tmean_n<1, 8>(rp, col, w, node, uniform, [&](int, int c) { return rowof(c); }, o);
    return o[0];
}
// ---- the gathers of a tile, software-pipelined (round 4). With the means taken edge batch by edge batch inside the tile (tmean_n),
// a tile paid the row pointers, then the columns / weights, then the rows, once per batch and side: ~8 dependent memory round trips
// that nothing overlapped (per-phase clocks: 38 % of k_train_b1; more resident waves did not help). Now the (column, weight) pairs
// of a tile's first edges are loaded while the PREVIOUS tile computes (NbrIdx, one 8-byte load per edge), so at the top of a tile
// all its rows are requested at once: one round trip. Out-edges past the prefetched ones (a station / source node that many others
// list as a neighbour) go through the batch loop as before. The sums keep the edge order.
template <int EB> struct NbrIdx { int c[EB]; float w[EB]; int e_next, e_end; };
template <int EB>
__device__ __forceinline__ void nbr_idx_load(const int2* __restrict__ cw, int eb, int ee, NbrIdx<EB>& o) {
#pragma unroll
    for (int k = 0; k < EB; ++k) {
        const bool ok = eb + k < ee;
        const int2 v = cw[ok ? eb + k : 0];
        o.c[k] = v.x;
        o.w[k] = ok ? __int_as_float(v.y) : 0.f;
    }
    o.e_next = eb + EB; o.e_end = ee;
}
// the edges [e, ee) of a node beyond the prefetched ones, EB at a time (as tmean_n)
template <int NB, int EB, typename F>
__device__ __forceinline__ void tmean_rest(const int2* __restrict__ cw, int e, int ee, bool uniform, F rowof, f32x4 (&out)[NB]) {
    for (; uniform ? (e < ee) : (bool)__any(e < ee); e += EB) {
        int c[EB];
        float ww[EB];
#pragma unroll
        for (int k = 0; k < EB; ++k) {
            const bool ok = e + k < ee;
            const int2 v = cw[ok ? e + k : 0];
            c[k] = v.x;
            ww[k] = ok ? __int_as_float(v.y) : 0.f;
        }
        f32x4 r[NB][EB];
#pragma unroll
        for (int k = 0; k < EB; ++k)
#pragma unroll
            for (int b = 0; b < NB; ++b) r[b][k] = rowof(b, c[k]);
#pragma unroll
        for (int k = 0; k < EB; ++k)
#pragma unroll
            for (int b = 0; b < NB; ++b) out[b] += r[b][k] * ww[k];
    }
}
// transposed mean with every (column, weight) pair of the node's first EBI out-edges requested at once, then their rows EB edges at a
// time (no index load between the row batches: the next batch's rows are in flight while this one is summed), then the rest of a
// long edge list. Edge order kept.
template <int NB, int EBI, int EB, typename F>
__device__ __forceinline__ void tmean_pre(const int32_t* __restrict__ rp, const int2* __restrict__ cw, int node, bool uniform, F rowof,
                                          f32x4 (&out)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) out[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    int eb = rp[node], ee = rp[node + 1];
    if (uniform) { eb = __builtin_amdgcn_readfirstlane(eb); ee = __builtin_amdgcn_readfirstlane(ee); }
    NbrIdx<EBI> x;
    nbr_idx_load<EBI>(cw, eb, ee, x);
    static_assert(EBI % EB == 0, "whole batches");
#pragma unroll
    for (int k0 = 0; k0 < EBI; k0 += EB) {
        if (k0 > 0 && (uniform ? !(eb + k0 < ee) : !(bool)__any(eb + k0 < ee))) break;
        f32x4 r[NB][EB];
#pragma unroll
        for (int k = 0; k < EB; ++k)
#pragma unroll
            for (int b = 0; b < NB; ++b) r[b][k] = rowof(b, x.c[k0 + k]);
#pragma unroll
        for (int k = 0; k < EB; ++k)
#pragma unroll
            for (int b = 0; b < NB; ++b) out[b] += r[b][k] * x.w[k0 + k];
    }
    tmean_rest<NB, EB>(cw, x.e_next, x.e_end, uniform, rowof, out);
}
// transposed mean over the out-edges whose (column, weight) pairs were loaded earlier (NbrIdx: by the previous tile of a software
// pipeline): their rows EB edges at a time, edge order kept. The edges past the prefetched ones go through tmean_rest.
template <int NB, int EBI, int EB, typename F>
__device__ __forceinline__ void tmean_idx(const NbrIdx<EBI>& x, bool uniform, F rowof, f32x4 (&out)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) out[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    static_assert(EBI % EB == 0, "whole batches");
    const int eb = x.e_next - EBI, ee = x.e_end;
#pragma unroll
    for (int k0 = 0; k0 < EBI; k0 += EB) {
        if (k0 > 0 && (uniform ? !(eb + k0 < ee) : !(bool)__any(eb + k0 < ee))) break;
        f32x4 r[NB][EB];
#pragma unroll
        for (int k = 0; k < EB; ++k)
#pragma unroll
            for (int b = 0; b < NB; ++b) r[b][k] = rowof(b, x.c[k0 + k]);
#pragma unroll
        for (int k = 0; k < EB; ++k)
#pragma unroll
            for (int b = 0; b < NB; ++b) out[b] += r[b][k] * x.w[k0 + k];
    }
}
__device__ __forceinline__ void write_partials(const TrArgs& a, int wid, const f32x4* acc, int n_acc, const f32x4* vec, int n_vec,
                                               const float* scal, int n_scal, int lane, int j, int q) {
    float* out = a.part + (size_t)wid * ((size_t)a.n_acc * 256 + (size_t)a.n_vec * 16 + 16);
    for (int k = 0; k < n_acc; ++k) *(f32x4*)(out + (size_t)k * 256 + lane * 4) = acc[k];
    out += (size_t)a.n_acc * 256;
    for (int k = 0; k < n_vec; ++k) {
        f32x4 v = vec[k];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            v.x += __shfl_xor(v.x, d); v.y += __shfl_xor(v.y, d); v.z += __shfl_xor(v.z, d); v.w += __shfl_xor(v.w, d);
        }
        if (j == 0) *(f32x4*)(out + k * 16 + 4 * q) = v;
    }
    out += (size_t)a.n_vec * 16;
    for (int k = 0; k < n_scal; ++k) {
        float v = scal[k];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
        if (lane == 0) out[k] = v;
    }
}

// ---- pass 2': Bipartite message + PReLU2.  accumulators: fc1 (t, {x_latent 0:15, x_latent 15:30, edge_attr}) = 6; vec: fc1 bias (2)
template <bool PCSR, int WPB = 4>      // WPB: waves per workgroup (60 registers, 6 accumulator tiles: this pass is bound by the latency of its
                                       // streamed rows, and more resident waves are what hides it)
__global__ __launch_bounds__(WPB * 64, 1) void k_train_b2(TrArgs a) {
    constexpr int NF4 = (GT2_GROUPS * 256 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    __shared__ __attribute__((aligned(16))) float tsc[WPB][16 * 17];
    for (int i = threadIdx.x; i < NF4; i += WPB * 64) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lscal = (const float*)(lw + GT2_GROUPS * 64);
    const float a2 = lscal[0], ab1 = lscal[1];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sc = tsc[wave];
    const int S = a.S;
    const long long P = a.P;
    f32x4 acc[6], vec[2];
    float scal[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    vec[0] = vec[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave);
    PtileIter ptw(PCSR ? (P + 15) / 16 : 0, WPB, wave);      // PCSR: positions in the processing order of the tiles
    const long long n_it = PCSR ? ptw.end : w.nitems, it0 = PCSR ? ptw.i : w.it, its = PCSR ? ptw.stride : w.stride;
    for (long long it = it0; it < n_it; it += its) {
        int g;
        bool valid;
        long long p;
        if (PCSR) {       // a tile is 16 consecutive product nodes, the source node is per lane
            const long long pr = ptile_at(a.ptile, it) * 16 + j;
            valid = pr < P;
            p = valid ? pr : P - 1;
            g = a.src_of[p];
        } else {
            int gi, tb;
            w.decode(it, gi, tb);
            g = __builtin_amdgcn_readfirstlane(a.order[gi]);
            const int s = tb * 16 + j;
            valid = s < S;
            p = (long long)g * S + (valid ? s : S - 1);
        }
        asm volatile("" : "+v"(lane));
        float mq = a.mask[p * 4 + q];
        float mm = fmaxf(mq, __shfl_xor(mq, 16));
        mm = fmaxf(mm, __shfl_xor(mm, 32));
        if (!valid) mm = 0.f;
        f32x4 eb = {0.f, 0.f, 0.f, 0.f};
        if (q == 0) { eb.x = a.edge_attr[p * 3]; eb.y = a.edge_attr[p * 3 + 1]; eb.z = a.edge_attr[p * 3 + 2]; }
        f32x4 dz[2], o[2], xl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 zb = ldb(a.save, SV_ZB + t, P, p, q);
            const f32x4 d = *(const f32x4*)(a.dr + (long long)g * 32 + 16 * t + 4 * q) * mm;      // through the mask gate
            scal[1] += negsum4(d, zb);
            dz[t] = d * dprelu4(zb, ab1);
            vec[t] += dz[t];
            o[t] = ldb(a.save, SV_O + t, P, p, q);
            xl[t] = prelu4u(o[t], a2);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x4 dx = {0.f, 0.f, 0.f, 0.f};
            dx = mma_block(dx, lw[GT2(b, 0) * 64 + lane], dz[0]);
            dx = mma_block(dx, lw[GT2(b, 1) * 64 + lane], dz[1]);
            if (!valid) dx = f32x4{0.f, 0.f, 0.f, 0.f};
            scal[0] += negsum4(dx, o[b]);
            const f32x4 dob = dx * dprelu4(o[b], a2);
            if (valid) stb(a.gr, GR_DO + b, P, p, q, dob);
        }
        const f32x4 x0t = tr16(xl[0], sc, j, q), x1t = tr16(xl[1], sc, j, q), et = tr16(eb, sc, j, q);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 dt_ = tr16(dz[t], sc, j, q);
            acc[t * 3 + 0] = outer16(acc[t * 3 + 0], dt_, x0t);
            acc[t * 3 + 1] = outer16(acc[t * 3 + 1], dt_, x1t);
            acc[t * 3 + 2] = outer16(acc[t * 3 + 2], dt_, et);
        }
    }
    write_partials(a, blockIdx.x * WPB + wave, acc, 6, vec, 2, scal, 2, threadIdx.x & 63, j, q);
}

// ---- pass 1': layer 2 and the activation of layer 1.
// accumulators: l2_t1_2 {h1 x4, Mask, u x2 (adjoint)} = 7, l2_t2_2 = 7, l2_t1_1 (2 x h1 x4) = 8, l2_t2_1 = 8  -> 30
// vec: b(l2_t1_2), b(l2_t2_2), b(l2_t1_1) x2, b(l2_t2_1) x2 = 6; scal: a1, a21, a22
// AS: the same pass for DataAggregationAssociationPhase (module.py:397-401; 95-wide l2_t?_2 with mask width 5): the column of mask1
// (one value per source node) gets its gradient as two extra vectors.
#if GENIE_TUNING
#define TPH_DECL long long tph[6] = {0, 0, 0, 0, 0, 0}; long long tph_t0 = 0
#define TPH_START() do { if (a.abl & 16) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tph_t0 = __builtin_amdgcn_s_memtime(); } } while (0)
#define TPH_MARK(k) do { if (a.abl & 16) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long t_ = __builtin_amdgcn_s_memtime(); tph[k] += t_ - tph_t0; tph_t0 = t_; } } while (0)
#else
#define TPH_DECL
#define TPH_START()
#define TPH_MARK(k)
#endif
template <bool AS, bool O32 = false>      // O32: rows addressed by 32-bit offsets on scalar bases (ldo / sto; P < 4.79 M)
__global__ __launch_bounds__(256, 1) void k_train_b1(TrArgs a) {
    constexpr int NF4 = (GT1_GROUPS * 256 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    __shared__ __attribute__((aligned(16))) float tsc[4][16 * 17];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lscal = (const float*)(lw + GT1_GROUPS * 64);
    const float a1 = lscal[0], a21 = lscal[1], a22 = lscal[2];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sc = tsc[wave];
    const int S = a.S;
    const long long P = a.P;
    constexpr int NV = AS ? 8 : 6;
    f32x4 acc[30], vec[NV];
    float scal[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 30; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NV; ++k) vec[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int SVT = a.sv_t, SVU = a.sv_up, SVV = a.sv_vp;
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave);
    TPH_DECL;
    constexpr int EBS = 8, EBG = 16;            // prefetched out-edges per station (per lane) / per source node (uniform)
    struct Tile { int g, scn; bool valid; };
    struct Rp { int s0, s1, g0, g1; };
    struct Own { f32x4 mb, do1, do2, t[4], up[2], vp[2]; };       // the node's own rows: Mask, do, the kept pre-activations
    const float* gr = a.gr;
    const unsigned q16 = 16u * (unsigned)q, P64 = (unsigned)P * 64u, S64 = (unsigned)S * 64u;
    const unsigned long long grb = sbase(a.gr), svb = sbase(a.save);
    const unsigned oDO0 = (unsigned)(GR_DO + 0) * P64, oDO1 = (unsigned)(GR_DO + 1) * P64;
    auto tile_of = [&](long long it) {
        int gi, tb;
        w.decode(it < w.nitems ? it : w.it, gi, tb);               // past the end: the current tile again (loads nobody uses)
        Tile t;
        t.g = __builtin_amdgcn_readfirstlane(a.order[gi]);
        const int s = tb * 16 + j;
        t.valid = s < S;
        t.scn = t.valid ? s : S - 1;
        return t;
    };
    auto rp_of = [&](const Tile& t) {
        return Rp{a.r_sta_rowptr[t.scn], a.r_sta_rowptr[t.scn + 1], __builtin_amdgcn_readfirstlane(a.r_src_rowptr[t.g]),
                  __builtin_amdgcn_readfirstlane(a.r_src_rowptr[t.g + 1])};
    };
    auto own_load = [&](const Tile& t, Own& o) {
        const long long p_ = (long long)t.g * S + t.scn;
        o.mb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (q == 0) o.mb = *(const f32x4*)(a.mask + p_ * 4);
        if (O32) {
            const unsigned po = (unsigned)p_ * 64u + q16;
            o.do1 = ldo(grb, oDO0 + po); o.do2 = ldo(grb, oDO1 + po);
#pragma unroll
            for (int k = 0; k < 4; ++k) o.t[k] = ldo(svb, (unsigned)(SVT + k) * P64 + po);
#pragma unroll
            for (int b = 0; b < 2; ++b) { o.up[b] = ldo(svb, (unsigned)(SVU + b) * P64 + po); o.vp[b] = ldo(svb, (unsigned)(SVV + b) * P64 + po); }
            return;
        }
        o.do1 = ldb(a.gr, GR_DO + 0, P, p_, q); o.do2 = ldb(a.gr, GR_DO + 1, P, p_, q);
#pragma unroll
        for (int k = 0; k < 4; ++k) o.t[k] = ldb(a.save, SVT + k, P, p_, q);
#pragma unroll
        for (int b = 0; b < 2; ++b) { o.up[b] = ldb(a.save, SVU + b, P, p_, q); o.vp[b] = ldb(a.save, SVV + b, P, p_, q); }
    };
    auto rows_issue = [&](const Tile& t, const NbrIdx<EBS>& xs, const NbrIdx<EBG>& xg, f32x4 (&rs)[EBS], f32x4 (&rg)[EBG]) {
        if (O32) {      // station rows of this source node: v_lshl_add_u32 per row; this station's row of another source node: scalar row base + one add
            const unsigned vs = oDO0 + (unsigned)(t.g * S) * 64u + q16, vg = oDO1 + (unsigned)t.scn * 64u + q16;
#pragma unroll
            for (int k = 0; k < EBS; ++k) rs[k] = ldo(grb, vs + ((unsigned)xs.c[k] << 6));
#pragma unroll
            for (int k = 0; k < EBG; ++k) rg[k] = ldo(grb, (unsigned)xg.c[k] * S64 + vg);
            return;
        }
#pragma unroll
        for (int k = 0; k < EBS; ++k) rs[k] = ldb(gr, GR_DO + 0, P, (long long)t.g * S + xs.c[k], q);
#pragma unroll
        for (int k = 0; k < EBG; ++k) rg[k] = ldb(gr, GR_DO + 1, P, (long long)xg.c[k] * S + t.scn, q);
    };
    auto rows_sum = [&](const Tile& t, const NbrIdx<EBS>& xs, const NbrIdx<EBG>& xg, const f32x4 (&rs)[EBS], const f32x4 (&rg)[EBG], f32x4& tm1,
                        f32x4& tm2) {
        f32x4 tms[1] = {f32x4{0.f, 0.f, 0.f, 0.f}}, tmg[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int k = 0; k < EBS; ++k) tms[0] += rs[k] * xs.w[k];
#pragma unroll
        for (int k = 0; k < EBG; ++k) tmg[0] += rg[k] * xg.w[k];
        const int g_ = t.g, scn_ = t.scn;
        if (O32) {
            const unsigned vs = oDO0 + (unsigned)(g_ * S) * 64u + q16, vg = oDO1 + (unsigned)scn_ * 64u + q16;
            tmean_rest<1, 8>(a.r_sta_cw, xs.e_next, xs.e_end, false, [&](int, int c) { return ldo(grb, vs + ((unsigned)c << 6)); }, tms);
            tmean_rest<1, 8>(a.r_src_cw, xg.e_next, xg.e_end, true, [&](int, int c) { return ldo(grb, (unsigned)c * S64 + vg); }, tmg);
        } else {
            tmean_rest<1, 8>(a.r_sta_cw, xs.e_next, xs.e_end, false, [&](int, int c) { return ldb(gr, GR_DO + 0, P, (long long)g_ * S + c, q); }, tms);
            tmean_rest<1, 8>(a.r_src_cw, xg.e_next, xg.e_end, true, [&](int, int c) { return ldb(gr, GR_DO + 1, P, (long long)c * S + scn_, q); }, tmg);
        }
        const float vm_ = t.valid ? 1.f : 0.f;
        tm1 = tms[0] * vm_; tm2 = tmg[0] * vm_;
    };
    // Software pipeline over the tiles of a wave (one wave per SIMD: the 30 accumulator tiles leave no room for a second one, so
    // nothing else hides a round trip to memory): while tile i computes, the row pointers of tile i + 2, the edge pairs and the own
    // rows of tile i + 1 and -- from the middle of the tile -- the gathered rows of tile i + 1 are in flight; its transposed
    // means are summed at the end of tile i.
    Tile cur = {0, 0, false}, nxt = cur;
    Rp rp_n = {0, 0, 0, 0};
    Own own;
    f32x4 tm1 = {0.f, 0.f, 0.f, 0.f}, tm2 = tm1;
    if (w.it < w.nitems) {
        cur = tile_of(w.it);
        const Rp rp = rp_of(cur);
        NbrIdx<EBS> xs;
        NbrIdx<EBG> xg;
        nbr_idx_load<EBS>(a.r_sta_cw, rp.s0, rp.s1, xs);
        nbr_idx_load<EBG>(a.r_src_cw, rp.g0, rp.g1, xg);
        own_load(cur, own);
        f32x4 rs[EBS], rg[EBG];
        rows_issue(cur, xs, xg, rs, rg);
        rows_sum(cur, xs, xg, rs, rg, tm1, tm2);
        nxt = tile_of(w.it + w.stride);
        rp_n = rp_of(nxt);
    }
    for (; w.it < w.nitems; w.it += w.stride) {
        TPH_START();
        asm volatile("" : "+v"(lane));
        const int g = cur.g, scn = cur.scn;
        const bool valid = cur.valid;
        const long long p = (long long)g * S + scn;
        const float vm = valid ? 1.f : 0.f;
        const Tile nn = tile_of(w.it + 2 * w.stride);
        const Rp rp_nn = rp_of(nn);
        NbrIdx<EBS> xs_n;
        NbrIdx<EBG> xg_n;
        nbr_idx_load<EBS>(a.r_sta_cw, rp_n.s0, rp_n.s1, xs_n);
        nbr_idx_load<EBG>(a.r_src_cw, rp_n.g0, rp_n.g1, xg_n);
        Own own_n;
        own_load(nxt, own_n);
        const f32x4 mb = own.mb;
        const f32x4 do1 = own.do1 * vm, do2 = own.do2 * vm;
        TPH_MARK(0);
        if (AS) {
            const float m1 = a.pg[(long long)g * AS_PG + 31];
            vec[6] += do1 * m1; vec[7] += do2 * m1;
        }
        TPH_MARK(1);
        f32x4 t[4], h1[4], up[2], vp[2], u[2], v[2];
#pragma unroll
        for (int k = 0; k < 4; ++k) { t[k] = own.t[k]; h1[k] = prelu4u(t[k], a1); }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            up[b] = own.up[b]; u[b] = prelu4u(up[b], a21);
            vp[b] = own.vp[b]; v[b] = prelu4u(vp[b], a22);
        }
        // du = l2_t1_2[:, 60:90]^T tm1 through PReLU21', dv likewise
        f32x4 du[2], dv[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 gu = mma_block(z, lw[GT_U(b) * 64 + lane], tm1), gv = mma_block(z, lw[GT_V(b) * 64 + lane], tm2);
            scal[1] += negsum4(gu, up[b]);
            scal[2] += negsum4(gv, vp[b]);
            du[b] = gu * dprelu4(up[b], a21);
            dv[b] = gv * dprelu4(vp[b], a22);
        }
        TPH_MARK(2);
        // dh1 and dt
        f32x4 dt[4];
#pragma unroll
        for (int hb = 0; hb < 4; ++hb) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            d = mma_block(d, lw[GT_H(hb, 0) * 64 + lane], du[0]);
            d = mma_block(d, lw[GT_H(hb, 1) * 64 + lane], du[1]);
            d = mma_block(d, lw[GT_H(hb, 2) * 64 + lane], dv[0]);
            d = mma_block(d, lw[GT_H(hb, 3) * 64 + lane], dv[1]);
            d = mma_block(d, lw[GT_H(hb, 4) * 64 + lane], do1);
            d = mma_block(d, lw[GT_H(hb, 5) * 64 + lane], do2);
            scal[0] += negsum4(d, t[hb]);
            dt[hb] = d * dprelu4(t[hb], a1);
#if GENIE_TUNING
            if (a.abl & 8) { if (dt[hb].x == 1.2345f) stb(a.gr, GR_DT + hb, P, p, q, dt[hb]); continue; }
#endif
            if (valid) { if (O32) sto(grb, (unsigned)(GR_DT + hb) * P64 + (unsigned)p * 64u + q16, dt[hb]); else stb(a.gr, GR_DT + hb, P, p, q, dt[hb]); }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) d = mma_block(d, lw[GT_D(b, k) * 64 + lane], dt[k]);
            if (valid) { if (O32) sto(grb, (unsigned)(GR_DH0 + b) * P64 + (unsigned)p * 64u + q16, d); else stb(a.gr, GR_DH0 + b, P, p, q, d); }
        }
        vec[0] += do1; vec[1] += do2;
        vec[2] += du[0]; vec[3] += du[1]; vec[4] += dv[0]; vec[5] += dv[1];
        TPH_MARK(3);
        // the next tile's gathered rows: in flight under the weight gradients
        f32x4 rs[EBS], rg[EBG];
        rows_issue(nxt, xs_n, xg_n, rs, rg);
        // weight gradients
        f32x4 h1t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) h1t[k] = tr16(h1[k], sc, j, q);
        const f32x4 mt = tr16(mb, sc, j, q);
        {
            const f32x4 d1t = tr16(do1, sc, j, q), d2t = tr16(do2, sc, j, q);
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[k] = outer16(acc[k], d1t, h1t[k]); acc[7 + k] = outer16(acc[7 + k], d2t, h1t[k]); }
            acc[4] = outer16(acc[4], d1t, mt);
            acc[11] = outer16(acc[11], d2t, mt);
            const f32x4 m1t = tr16(tm1, sc, j, q), m2t = tr16(tm2, sc, j, q);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc[5 + b] = outer16(acc[5 + b], m1t, tr16(u[b], sc, j, q));
                acc[12 + b] = outer16(acc[12 + b], m2t, tr16(v[b], sc, j, q));
            }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x4 dut = tr16(du[b], sc, j, q), dvt = tr16(dv[b], sc, j, q);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[14 + b * 4 + k] = outer16(acc[14 + b * 4 + k], dut, h1t[k]);
                acc[22 + b * 4 + k] = outer16(acc[22 + b * 4 + k], dvt, h1t[k]);
            }
        }
        TPH_MARK(4);
        rows_sum(nxt, xs_n, xg_n, rs, rg, tm1, tm2);
        cur = nxt; nxt = nn; rp_n = rp_nn; own = own_n;
    }
#if GENIE_TUNING
    if ((a.abl & 16) && (threadIdx.x & 63) == 0 && wave == 1 && (blockIdx.x == 3 || blockIdx.x == 200))
        printf("b1 blk %d: own loads %lld, gathers %lld, save loads + du/dv %lld, dh1/dt/dh0 + stores %lld, weight grads %lld\n", blockIdx.x,
               tph[0], tph[1], tph[2], tph[3], tph[4]);
#endif
    write_partials(a, blockIdx.x * 4 + wave, acc, 30, vec, NV, scal, 3, threadIdx.x & 63, j, q);
}

// ---- pass 1' with a tile shared by TWO waves (round 6). k_train_b1 holds 30 accumulator tiles, the gathered rows of both reversed
// graphs and the next tile's prefetch in ONE wave: 434-452 unified registers = one wave per SIMD, and its counters say what that costs
// (profiles/r05_d_pmc_train_b1_f16x2.txt, quad-cycle units: SQ_WAIT_ANY 3.4e8 of SQ_WAVE_CYCLES 7.2e8 = 48 % of the wave's life parked
// on a counter, 22 % issuing vector instructions, the matrix pipe busy 22 %): nothing else is resident to run meanwhile. Here the two
// branches of the layer (`l?_t1_*` = station side, `l?_t2_*` = source side, module.py:90-96) go to the two waves of a PAIR:
//   role 0: transposed mean of do1 over the reversed station graph, du, dh1 / dt blocks 0-1, dh0 block 0, the 15 weight-gradient tiles
//           of l2_t1_2 and l2_t1_1;   role 1: the same with do2 / the reversed source graph / dv / blocks 2-3 / block 1 / l2_t2_*.
// dh1 needs both branches' du and dv, dh0 all four dt blocks: two exchanges of two rows per lane through LDS with a workgroup barrier
// each (a workgroup = two pairs with the same trip count). Every sum keeps the order it has in k_train_b1, so dt / dh0 and the
// per-tile products are the same bits; only the grouping of the per-wave partials changes. <= 256 registers: two workgroups = two
// waves per SIMD. Cartesian graphs with 32-bit row offsets (the O32 form of k_train_b1) only.
template <bool AS, int ROLE>
__device__ __forceinline__ void train_b1s_role(const TrArgs& a, const f32x4* lw, float* sc, f32x4* ex1, f32x4* ex2, float* sx, int pair) {
    const float* lscal = (const float*)(lw + GT1_GROUPS * 64);
    const float a1 = lscal[0], ax = lscal[1 + ROLE];            // a21 (u) / a22 (v)
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int S = a.S;
    const long long P = a.P;
    f32x4 acc[15], vec[4];
    float scal0 = 0.f, scalx = 0.f;
#pragma unroll
    for (int k = 0; k < 15; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) vec[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int SVT = a.sv_t, SVX = ROLE == 0 ? a.sv_up : a.sv_vp;
    ItemIter w(a.G, a.T, a.seg, a.nxcd, 0);
    // a workgroup takes its chunk's items two at a time (one per pair): both pairs run the same number of tiles (and barriers)
    const int nx = (a.nxcd > 1 && (int)gridDim.x >= a.nxcd && ((int)gridDim.x % a.nxcd) == 0) ? a.nxcd : 1;
    const long long lb = blockIdx.x / nx, nbx = gridDim.x / nx;
    constexpr int EB = ROLE == 0 ? 8 : 16;          // prefetched out-edges: per station (per lane) / per source node (uniform)
    struct Tile { int g, scn; bool valid; };
    const unsigned q16 = 16u * (unsigned)q, P64 = (unsigned)P * 64u, S64 = (unsigned)S * 64u;
    const unsigned long long grb = sbase(a.gr), svb = sbase(a.save);
    const unsigned oDO0 = (unsigned)(GR_DO + 0) * P64, oDO1 = (unsigned)(GR_DO + 1) * P64;
    const int2* cw = ROLE == 0 ? a.r_sta_cw : a.r_src_cw;
    auto tile_of = [&](long long base) {         // base: the workgroup's item pair; past the end or an odd last item: pair 0's tile, all lanes invalid
        const bool live = base < w.nitems;
        const long long b_ = live ? base : lb * 2;
        const bool mine = live && b_ + pair < w.nitems;
        int gi, tb;
        w.decode(mine ? b_ + pair : b_, gi, tb);
        Tile t;
        t.g = __builtin_amdgcn_readfirstlane(a.order[gi]);
        const int s_ = tb * 16 + j;
        t.valid = mine && s_ < S;
        t.scn = s_ < S ? s_ : S - 1;
        return t;
    };
    auto idx_of = [&](const Tile& t, NbrIdx<EB>& x) {
        if (ROLE == 0) nbr_idx_load<EB>(cw, a.r_sta_rowptr[t.scn], a.r_sta_rowptr[t.scn + 1], x);
        else nbr_idx_load<EB>(cw, __builtin_amdgcn_readfirstlane(a.r_src_rowptr[t.g]), __builtin_amdgcn_readfirstlane(a.r_src_rowptr[t.g + 1]), x);
    };
    auto row_at = [&](const Tile& t, int c) {      // the row of out-neighbour c: this source node's station c / source node c's row of this station
        if (ROLE == 0) return ldo(grb, oDO0 + (unsigned)(t.g * S) * 64u + q16 + ((unsigned)c << 6));
        return ldo(grb, (unsigned)c * S64 + oDO1 + (unsigned)t.scn * 64u + q16);
    };
    auto rows_sum = [&](const Tile& t, const NbrIdx<EB>& x, const f32x4 (&r)[EB]) {
        f32x4 tm[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int k = 0; k < EB; ++k) tm[0] += r[k] * x.w[k];
        tmean_rest<1, 8>(cw, x.e_next, x.e_end, ROLE == 1, [&](int, int c) { return row_at(t, c); }, tm);
        return tm[0] * (t.valid ? 1.f : 0.f);
    };
    struct Own { f32x4 mb, do1, do2, t[4], xp[2]; };
    auto own_load = [&](const Tile& t, Own& o) {
        const long long p_ = (long long)t.g * S + t.scn;
        o.mb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (q == 0) o.mb = *(const f32x4*)(a.mask + p_ * 4);
        const unsigned po = (unsigned)p_ * 64u + q16;
        o.do1 = ldo(grb, oDO0 + po); o.do2 = ldo(grb, oDO1 + po);
#pragma unroll
        for (int k = 0; k < 4; ++k) o.t[k] = ldo(svb, (unsigned)(SVT + k) * P64 + po);
#pragma unroll
        for (int b = 0; b < 2; ++b) o.xp[b] = ldo(svb, (unsigned)(SVX + b) * P64 + po);
    };
    long long base = lb * 2;
    Tile cur = {0, 0, false}, nxt = cur;
    f32x4 tm = {0.f, 0.f, 0.f, 0.f};
    if (base < w.nitems) {
        cur = tile_of(base);
        NbrIdx<EB> x;
        idx_of(cur, x);
        f32x4 r[EB];
#pragma unroll
        for (int k = 0; k < EB; ++k) r[k] = row_at(cur, x.c[k]);
        tm = rows_sum(cur, x, r);
        nxt = tile_of(base + nbx * 2);
    }
    for (; base < w.nitems; base += nbx * 2) {
        asm volatile("" : "+v"(lane));
        const int g = cur.g, scn = cur.scn;
        const bool valid = cur.valid;
        const unsigned po = (unsigned)((long long)g * S + scn) * 64u + q16;
        const float vm = valid ? 1.f : 0.f;
        Own own;
        own_load(cur, own);
        NbrIdx<EB> xn;
        idx_of(nxt, xn);                      // the next tile's (column, weight) pairs: in flight under this tile
        const f32x4 do1 = own.do1 * vm, do2 = own.do2 * vm;
        const f32x4 dob = ROLE == 0 ? do1 : do2;
        if (AS) vec[3] += dob * a.pg[(long long)g * AS_PG + 31];
        // this branch's du (dv) = l2_t?_2[:, 60:90]^T tm through PReLU2?'
        f32x4 dx[2], xv[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 gx = mma_block(z, lw[(ROLE == 0 ? GT_U(b) : GT_V(b)) * 64 + lane], tm);
            scalx += negsum4(gx, own.xp[b]);
            dx[b] = gx * dprelu4(own.xp[b], ax);
            xv[b] = prelu4u(own.xp[b], ax);
            ex1[b * 64 + lane] = dx[b];
        }
        __syncthreads();
        const f32x4 dy0 = ex1[(ROLE == 0 ? 128 : -128) + lane], dy1 = ex1[(ROLE == 0 ? 128 : -128) + 64 + lane];     // the partner's rows
        const f32x4 du0 = ROLE == 0 ? dx[0] : dy0, du1 = ROLE == 0 ? dx[1] : dy1;
        const f32x4 dv0 = ROLE == 0 ? dy0 : dx[0], dv1 = ROLE == 0 ? dy1 : dx[1];
        // dh1 / dt of this wave's two blocks: the six-term chain of k_train_b1 per block, same order inside a chain. The two chains are
        // written INTERLEAVED, MFMA by MFMA, with the next pair of weight fragments read from LDS one step ahead: left to itself the
        // compiler ran chain 0 then chain 1, 48 dependent MFMAs in a row (40-cycle dependent latency on a 32-cycle instruction) with an
        // LDS read and a full lgkmcnt(0) wait in front of every fourth one (round-6 disassembly)
        f32x4 dt[4];
        {
            const f32x4 xs6[6] = {du0, du1, dv0, dv1, do1, do2};
            constexpr int hb0 = 2 * ROLE, hb1 = 2 * ROLE + 1;
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
            f32x4 wa = lw[GT_H(hb0, 0) * 64 + lane], wb = lw[GT_H(hb1, 0) * 64 + lane];
#pragma unroll
            for (int sidx = 0; sidx < 6; ++sidx) {
                f32x4 na = wa, nb = wb;
                if (sidx < 5) { na = lw[GT_H(hb0, sidx + 1) * 64 + lane]; nb = lw[GT_H(hb1, sidx + 1) * 64 + lane]; }
                const f32x4 x = xs6[sidx];
                d0 = MFMA16(wa.x, x.x, d0); d1 = MFMA16(wb.x, x.x, d1);
                d0 = MFMA16(wa.y, x.y, d0); d1 = MFMA16(wb.y, x.y, d1);
                d0 = MFMA16(wa.z, x.z, d0); d1 = MFMA16(wb.z, x.z, d1);
                d0 = MFMA16(wa.w, x.w, d0); d1 = MFMA16(wb.w, x.w, d1);
                wa = na; wb = nb;
            }
            scal0 += negsum4(d0, own.t[hb0]);
            scal0 += negsum4(d1, own.t[hb1]);
            dt[hb0] = d0 * dprelu4(own.t[hb0], a1);
            dt[hb1] = d1 * dprelu4(own.t[hb1], a1);
            if (valid) {
                sto(grb, (unsigned)(GR_DT + hb0) * P64 + po, dt[hb0]);
                sto(grb, (unsigned)(GR_DT + hb1) * P64 + po, dt[hb1]);
            }
            ex2[lane] = dt[hb0];
            ex2[64 + lane] = dt[hb1];
        }
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) dt[2 * (1 - ROLE) + hh] = ex2[(ROLE == 0 ? 128 : -128) + hh * 64 + lane];
        {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) d = mma_block(d, lw[GT_D(ROLE, k) * 64 + lane], dt[k]);
            if (valid) sto(grb, (unsigned)(GR_DH0 + ROLE) * P64 + po, d);
        }
        vec[0] += dob; vec[1] += dx[0]; vec[2] += dx[1];
        // the next tile's gathered rows: in flight under the weight gradients
        f32x4 r[EB];
#pragma unroll
        for (int k = 0; k < EB; ++k) r[k] = row_at(nxt, xn.c[k]);
        // weight gradients of this branch: l2_t?_2 {h1 x4, Mask, x x2 (adjoint)} = 7 tiles, l2_t?_1 (2 x h1 x4) = 8 tiles
        f32x4 h1t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) h1t[k] = tr16(prelu4u(own.t[k], a1), sc, j, q);
        {
            const f32x4 dbt = tr16(dob, sc, j, q);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = outer16(acc[k], dbt, h1t[k]);
            acc[4] = outer16(acc[4], dbt, tr16(own.mb, sc, j, q));
            const f32x4 mt = tr16(tm, sc, j, q);
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[5 + b] = outer16(acc[5 + b], mt, tr16(xv[b], sc, j, q));
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x4 dxt = tr16(dx[b], sc, j, q);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[7 + b * 4 + k] = outer16(acc[7 + b * 4 + k], dxt, h1t[k]);
        }
        tm = rows_sum(nxt, xn, r);
        cur = nxt;
        nxt = tile_of(base + nbx * 4);
    }
    // partials of the PAIR in the layout of k_train_b1's wave (k_train_reduce): accumulator tiles 0-6 / 14-21 (role 0), 7-13 / 22-29
    // (role 1); vectors b(l2_t1_2) 0, b(l2_t2_2) 1, b(l2_t1_1) 2-3, b(l2_t2_1) 4-5, AS: the mask1 column 6 / 7; slopes a1 (both), a21, a22
    float* out = a.part + (size_t)(blockIdx.x * 2 + pair) * ((size_t)a.n_acc * 256 + (size_t)a.n_vec * 16 + 16);
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        const int gk = ROLE == 0 ? (k < 7 ? k : k + 7) : (k < 7 ? k + 7 : k + 15);
        *(f32x4*)(out + (size_t)gk * 256 + lane * 4) = acc[k];
    }
    out += (size_t)a.n_acc * 256;
    constexpr int NVW = AS ? 4 : 3;
#pragma unroll
    for (int k = 0; k < NVW; ++k) {
        f32x4 v = vec[k];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            v.x += __shfl_xor(v.x, d); v.y += __shfl_xor(v.y, d); v.z += __shfl_xor(v.z, d); v.w += __shfl_xor(v.w, d);
        }
        const int gk = k == 0 ? ROLE : (k == 3 ? 6 + ROLE : 2 + 2 * ROLE + (k - 1));
        if (j == 0) *(f32x4*)(out + gk * 16 + 4 * q) = v;
    }
    out += (size_t)a.n_vec * 16;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { scal0 += __shfl_xor(scal0, d); scalx += __shfl_xor(scalx, d); }
    if (ROLE == 1 && lane == 0) sx[pair] = scal0;
    __syncthreads();
    if (lane == 0) {
        if (ROLE == 0) out[0] = scal0 + sx[pair];
        out[1 + ROLE] = scalx;
    }
}
template <bool AS>
__global__ __launch_bounds__(256, 2) void k_train_b1s(TrArgs a) {
    constexpr int NF4 = (GT1_GROUPS * 256 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    __shared__ __attribute__((aligned(16))) float tsc[4][16 * 17];
    __shared__ f32x4 ex1[4 * 2 * 64], ex2[4 * 2 * 64];      // [wave][row][lane]: the rows a wave hands to its partner (du | dv, then dt)
    __shared__ float sx[2];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((wave & 1) == 0) train_b1s_role<AS, 0>(a, lw, tsc[wave], ex1 + wave * 128, ex2 + wave * 128, sx, wave >> 1);
    else train_b1s_role<AS, 1>(a, lw, tsc[wave], ex1 + wave * 128, ex2 + wave * 128, sx, wave >> 1);
}

// ---- pass 1' on an irregular product graph (`use_subgraph`): the structure k_train_b1 had before its software pipeline. A tile is 16
// consecutive product nodes; the transposed means run over the reversed PRODUCT-level graphs (out-edges of a product node, per lane
// on both sides), gradient rows are addressed by product-node id.
template <bool AS>
__global__ __launch_bounds__(256, 1) void k_train_b1p(TrArgs a) {
    constexpr int NF4 = (GT1_GROUPS * 256 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    __shared__ __attribute__((aligned(16))) float tsc[4][16 * 17];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lscal = (const float*)(lw + GT1_GROUPS * 64);
    const float a1 = lscal[0], a21 = lscal[1], a22 = lscal[2];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sc = tsc[wave];
    const int S = a.S;
    const long long P = a.P;
    constexpr int NV = AS ? 8 : 6;
    f32x4 acc[30], vec[NV];
    float scal[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 30; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NV; ++k) vec[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int SVT = a.sv_t, SVU = a.sv_up, SVV = a.sv_vp;
    (void)S;
    const long long ntiles = (P + 15) / 16;
    PtileIter ptw(ntiles, 4, wave);
    for (; ptw.i < ptw.end; ptw.i += ptw.stride) {
        asm volatile("" : "+v"(lane));
        const long long pr = ptile_at(a.ptile, ptw.i) * 16 + j;
        const bool valid = pr < P;
        const long long p = valid ? pr : P - 1;
        const int g = a.src_of[p];               // per lane: a tile may straddle source nodes
        const float vm = valid ? 1.f : 0.f;
        f32x4 mb = {0.f, 0.f, 0.f, 0.f};
        if (q == 0) mb = *(const f32x4*)(a.mask + p * 4);
        const f32x4 do1 = ldb(a.gr, GR_DO + 0, P, p, q) * vm, do2 = ldb(a.gr, GR_DO + 1, P, p, q) * vm;
        if (AS) {
            const float m1 = a.pg[(long long)g * AS_PG + 31];
            vec[6] += do1 * m1; vec[7] += do2 * m1;
        }
        const float* gr = a.gr;
        f32x4 tms[1], tmg[1];                    // reversed PRODUCT-level graphs: out-edges of product node p, rows by product-node id
        tmean_pre<1, 8, 8>(a.r_sta_rowptr, a.r_sta_cw, (int)p, false, [&](int, int c) { return ldb(gr, GR_DO + 0, P, c, q); }, tms);
        tmean_pre<1, 16, 8>(a.r_src_rowptr, a.r_src_cw, (int)p, false, [&](int, int c) { return ldb(gr, GR_DO + 1, P, c, q); }, tmg);
        const f32x4 tm1 = tms[0] * vm, tm2 = tmg[0] * vm;
        f32x4 t[4], h1[4], up[2], vp[2], u[2], v[2];
#pragma unroll
        for (int k = 0; k < 4; ++k) { t[k] = ldb(a.save, SVT + k, P, p, q); h1[k] = prelu4u(t[k], a1); }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            up[b] = ldb(a.save, SVU + b, P, p, q); u[b] = prelu4u(up[b], a21);
            vp[b] = ldb(a.save, SVV + b, P, p, q); v[b] = prelu4u(vp[b], a22);
        }
        // du = l2_t1_2[:, 60:90]^T tm1 through PReLU21', dv likewise
        f32x4 du[2], dv[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 gu = mma_block(z, lw[GT_U(b) * 64 + lane], tm1), gv = mma_block(z, lw[GT_V(b) * 64 + lane], tm2);
            scal[1] += negsum4(gu, up[b]);
            scal[2] += negsum4(gv, vp[b]);
            du[b] = gu * dprelu4(up[b], a21);
            dv[b] = gv * dprelu4(vp[b], a22);
        }
        // dh1 and dt
        f32x4 dt[4];
#pragma unroll
        for (int hb = 0; hb < 4; ++hb) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            d = mma_block(d, lw[GT_H(hb, 0) * 64 + lane], du[0]);
            d = mma_block(d, lw[GT_H(hb, 1) * 64 + lane], du[1]);
            d = mma_block(d, lw[GT_H(hb, 2) * 64 + lane], dv[0]);
            d = mma_block(d, lw[GT_H(hb, 3) * 64 + lane], dv[1]);
            d = mma_block(d, lw[GT_H(hb, 4) * 64 + lane], do1);
            d = mma_block(d, lw[GT_H(hb, 5) * 64 + lane], do2);
            scal[0] += negsum4(d, t[hb]);
            dt[hb] = d * dprelu4(t[hb], a1);
            if (valid) stb(a.gr, GR_DT + hb, P, p, q, dt[hb]);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) d = mma_block(d, lw[GT_D(b, k) * 64 + lane], dt[k]);
            if (valid) stb(a.gr, GR_DH0 + b, P, p, q, d);
        }
        vec[0] += do1; vec[1] += do2;
        vec[2] += du[0]; vec[3] += du[1]; vec[4] += dv[0]; vec[5] += dv[1];
        // weight gradients
        f32x4 h1t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) h1t[k] = tr16(h1[k], sc, j, q);
        const f32x4 mt = tr16(mb, sc, j, q);
        {
            const f32x4 d1t = tr16(do1, sc, j, q), d2t = tr16(do2, sc, j, q);
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[k] = outer16(acc[k], d1t, h1t[k]); acc[7 + k] = outer16(acc[7 + k], d2t, h1t[k]); }
            acc[4] = outer16(acc[4], d1t, mt);
            acc[11] = outer16(acc[11], d2t, mt);
            const f32x4 m1t = tr16(tm1, sc, j, q), m2t = tr16(tm2, sc, j, q);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc[5 + b] = outer16(acc[5 + b], m1t, tr16(u[b], sc, j, q));
                acc[12 + b] = outer16(acc[12 + b], m2t, tr16(v[b], sc, j, q));
            }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x4 dut = tr16(du[b], sc, j, q), dvt = tr16(dv[b], sc, j, q);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[14 + b * 4 + k] = outer16(acc[14 + b * 4 + k], dut, h1t[k]);
                acc[22 + b * 4 + k] = outer16(acc[22 + b * 4 + k], dvt, h1t[k]);
            }
        }
    }
    write_partials(a, blockIdx.x * 4 + wave, acc, 30, vec, NV, scal, 3, threadIdx.x & 63, j, q);
}

// ---- pass 0': layer 1 and init_trns.
// accumulators: init_trns (2 x [Slice || Mask]) = 2; l1_t1_2 {2 x (h0 x2, Mask), adjoint 2 x 2} = 10; l1_t2_2 = 10  -> 22
// vec: b(init_trns) x2, b(l1_t1_2) x2, b(l1_t2_2) x2 = 6; scal: a, a11, a12
// O32 (Cartesian graphs, P < 4.79 M): rows addressed by 32-bit offsets on scalar bases (ldo / sto)
template <bool PCSR, bool O32 = false>
__global__ __launch_bounds__(256, 1) void k_train_b0(TrArgs a) {
    static_assert(!(PCSR && O32), "32-bit row offsets: Cartesian product graphs only");
    constexpr int NF4 = (GT0_GROUPS * 256 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    __shared__ __attribute__((aligned(16))) float tsc[4][16 * 17];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lscal = (const float*)(lw + GT0_GROUPS * 64);
    const float a0 = lscal[0], a11 = lscal[1], a12 = lscal[2];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sc = tsc[wave];
    const int S = a.S;
    const long long P = a.P;
    f32x4 acc[22], vec[6];
    float scal[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 22; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 6; ++k) vec[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned q16 = 16u * (unsigned)q, P64 = (unsigned)P * 64u;
    const unsigned long long grb = sbase(a.gr), svb = sbase(a.save);
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave);
    PtileIter ptw(PCSR ? (P + 15) / 16 : 0, 4, wave);      // PCSR: positions in the processing order of the tiles
    const long long n_it = PCSR ? ptw.end : w.nitems, it0 = PCSR ? ptw.i : w.it, its = PCSR ? ptw.stride : w.stride;
    for (long long it = it0; it < n_it; it += its) {
        int g = 0, scn = 0;
        bool valid;
        long long p;
        if (PCSR) {       // 16 consecutive product nodes; the transposed means run over the reversed PRODUCT-level graphs
            const long long pr = ptile_at(a.ptile, it) * 16 + j;
            valid = pr < P;
            p = valid ? pr : P - 1;
        } else {
            int gi, tb;
            w.decode(it, gi, tb);
            g = __builtin_amdgcn_readfirstlane(a.order[gi]);
            const int s = tb * 16 + j;
            valid = s < S;
            scn = valid ? s : S - 1;
            p = (long long)g * S + scn;
        }
        asm volatile("" : "+v"(lane));
        const float vm = valid ? 1.f : 0.f;
        f32x4 xm = {0.f, 0.f, 0.f, 0.f}, mb = {0.f, 0.f, 0.f, 0.f};
        if (q == 0) { xm = *(const f32x4*)(a.slice + p * 4); mb = *(const f32x4*)(a.mask + p * 4); }
        if (q == 1) xm = *(const f32x4*)(a.mask + p * 4);
        const float* gr = a.gr;
        f32x4 z0[2], h0[2], dt[4], tmd1[2], tmd2[2];
        const unsigned pofs = (unsigned)p * 64u + q16;
#pragma unroll
        for (int b = 0; b < 2; ++b) z0[b] = O32 ? ldo(svb, (unsigned)(SV_Z0 + b) * P64 + pofs) : ldb(a.save, SV_Z0 + b, P, p, q);
#pragma unroll
        for (int k = 0; k < 4; ++k) dt[k] = O32 ? ldo(grb, (unsigned)(GR_DT + k) * P64 + pofs) : ldb(a.gr, GR_DT + k, P, p, q);      // (own rows requested before the gathers, not after)
        if (O32) {
            // one vector instruction per gathered row: v_lshl_add_u32 (station rows of this source node) / v_mad_u32_u24 (this station's row
            // of another source node) on per-tile offsets that already hold the block, the tile's own coordinate and the lane's 16 q
            const unsigned gs64 = (unsigned)(g * S) * 64u;
            const unsigned vs0 = (unsigned)(GR_DT + 0) * P64 + gs64 + q16, vs1 = (unsigned)(GR_DT + 1) * P64 + gs64 + q16;
            const unsigned vg0 = (unsigned)(GR_DT + 2) * P64 + (unsigned)scn * 64u + q16, vg1 = (unsigned)(GR_DT + 3) * P64 + (unsigned)scn * 64u + q16;
            const unsigned S64 = (unsigned)S * 64u;
            tmean_pre<2, 8, 4>(a.r_sta_rowptr, a.r_sta_cw, scn, false,
                               [&](int b, int c) { return ldo(grb, (b == 0 ? vs0 : vs1) + ((unsigned)c << 6)); }, tmd1);
            tmean_pre<2, 16, 4>(a.r_src_rowptr, a.r_src_cw, g, true,
                                [&](int b, int c) { return ldo(grb, __umul24((unsigned)c, S64) + (b == 0 ? vg0 : vg1)); }, tmd2);
        } else if (PCSR) {
            tmean_pre<2, 8, 4>(a.r_sta_rowptr, a.r_sta_cw, (int)p, false, [&](int b, int c) { return ldb(gr, GR_DT + b, P, c, q); }, tmd1);
            tmean_pre<2, 16, 4>(a.r_src_rowptr, a.r_src_cw, (int)p, false, [&](int b, int c) { return ldb(gr, GR_DT + 2 + b, P, c, q); }, tmd2);
        } else {
            tmean_pre<2, 8, 4>(a.r_sta_rowptr, a.r_sta_cw, scn, false,
                               [&](int b, int c) { return ldb(gr, GR_DT + b, P, (long long)g * S + c, q); }, tmd1);
            tmean_pre<2, 16, 4>(a.r_src_rowptr, a.r_src_cw, g, true,
                                [&](int b, int c) { return ldb(gr, GR_DT + 2 + b, P, (long long)c * S + scn, q); }, tmd2);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) { tmd1[b] *= vm; tmd2[b] *= vm; h0[b] = prelu4u(z0[b], a0); }
#pragma unroll
        for (int k = 0; k < 4; ++k) dt[k] *= vm;
        f32x4 dz0[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x4 dq1 = {0.f, 0.f, 0.f, 0.f}, dq2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                dq1 = mma_block(dq1, lw[GT_Q(0, b, k) * 64 + lane], tmd1[k]);
                dq2 = mma_block(dq2, lw[GT_Q(1, b, k) * 64 + lane], tmd2[k]);
            }
            scal[1] += negsum4(dq1, h0[b]);
            scal[2] += negsum4(dq2, h0[b]);
            const f32x4 dh0l = O32 ? ldo(grb, (unsigned)(GR_DH0 + b) * P64 + pofs) : ldb(a.gr, GR_DH0 + b, P, p, q);
            const f32x4 dh0 = dh0l * vm + dq1 * dprelu4(h0[b], a11) + dq2 * dprelu4(h0[b], a12);
            scal[0] += negsum4(dh0, z0[b]);
            dz0[b] = dh0 * dprelu4(z0[b], a0);
            if (a.store_dz0 && valid) stb(a.gr, GR_DH0 + b, P, p, q, dz0[b]);
        }
        vec[0] += dz0[0]; vec[1] += dz0[1];
        vec[2] += dt[0]; vec[3] += dt[1]; vec[4] += dt[2]; vec[5] += dt[3];
        const f32x4 xmt = tr16(xm, sc, j, q), mt = tr16(mb, sc, j, q);
        const f32x4 h0t[2] = {tr16(h0[0], sc, j, q), tr16(h0[1], sc, j, q)};
        const f32x4 q1t[2] = {tr16(prelu4u(h0[0], a11), sc, j, q), tr16(prelu4u(h0[1], a11), sc, j, q)};
        const f32x4 q2t[2] = {tr16(prelu4u(h0[0], a12), sc, j, q), tr16(prelu4u(h0[1], a12), sc, j, q)};
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[b] = outer16(acc[b], tr16(dz0[b], sc, j, q), xmt);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int base = 2 + 10 * h;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const f32x4 dtt = tr16(dt[2 * h + b], sc, j, q);
                acc[base + b * 3 + 0] = outer16(acc[base + b * 3 + 0], dtt, h0t[0]);
                acc[base + b * 3 + 1] = outer16(acc[base + b * 3 + 1], dtt, h0t[1]);
                acc[base + b * 3 + 2] = outer16(acc[base + b * 3 + 2], dtt, mt);
                const f32x4 tmt = tr16(h == 0 ? tmd1[b] : tmd2[b], sc, j, q);
                acc[base + 6 + b * 2 + 0] = outer16(acc[base + 6 + b * 2 + 0], tmt, h == 0 ? q1t[0] : q2t[0]);
                acc[base + 6 + b * 2 + 1] = outer16(acc[base + 6 + b * 2 + 1], tmt, h == 0 ? q1t[1] : q2t[1]);
            }
        }
    }
    write_partials(a, blockIdx.x * 4 + wave, acc, 22, vec, 6, scal, 3, threadIdx.x & 63, j, q);
}

// fixed-order reduction of the per-wave partials into the gradient blob (registry layout of the weight mirror): a workgroup
// owns 32 entries; its 8 groups of 32 threads sum the waves w = group, group + 8, ... and the 8 sums are added in group order
__global__ __launch_bounds__(256) void k_train_reduce(const float* __restrict__ part, int n_waves, int n_acc, int n_vec, int n_scal,
                                                      const AccDesc* __restrict__ ad, const VecDesc* __restrict__ vd,
                                                      const int32_t* __restrict__ sd, float* __restrict__ blob, int accumulate) {
    __shared__ float ps[8][32];
    const int stride = n_acc * 256 + n_vec * 16 + 16;
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + o;
    float s = 0.f;
    if (idx < stride) {      // four independent partial sums (loads in flight), combined in a fixed order
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int wv = grp;
        for (; wv + 24 < n_waves; wv += 32) {
            s0 += part[(size_t)wv * stride + idx];
            s1 += part[(size_t)(wv + 8) * stride + idx];
            s2 += part[(size_t)(wv + 16) * stride + idx];
            s3 += part[(size_t)(wv + 24) * stride + idx];
        }
        for (; wv < n_waves; wv += 8) s0 += part[(size_t)wv * stride + idx];
        s = (s0 + s1) + (s2 + s3);
    }
    ps[grp][o] = s;
    __syncthreads();
    if (grp != 0 || idx >= stride) return;
    int dst = -1;
    if (idx < n_acc * 256) {
        const int k = idx >> 8, lane = (idx & 255) >> 2, r = idx & 3;
        const int i = 4 * (lane >> 4) + r, n = lane & 15;
        const AccDesc d = ad[k];
        if (i < d.nrows && n < d.ncols && n >= d.n0) dst = d.mat_off + (d.row0 + i) * d.ld + d.col0 + n;
    } else if (idx < n_acc * 256 + n_vec * 16) {
        const int k = (idx - n_acc * 256) >> 4, i = (idx - n_acc * 256) & 15;
        const VecDesc d = vd[k];
        if (i < d.nrows) dst = d.off + (d.row0 + i) * d.stride;
    } else {
        const int k = idx - n_acc * 256 - n_vec * 16;
        if (k < n_scal) dst = sd[k];
    }
    if (dst < 0) return;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += ps[k][o];
    blob[dst] = accumulate ? blob[dst] + t : t;      // accumulate: parameters that several passes contribute to (each pass in stream order)
}

// ---- weight gradients of the STATIC terms of the two other model definitions (DataAggregationEdges: mean edge features per
// station / per source node, module.py:102-174; use_absolute_pos: the scaled positions of a product node's station and source node,
// module.py:56-57). Such a term is W_f f[n] with f a fixed table over the nodes n of ONE base graph, so its weight gradient is
//   dW_f[out][e] = sum_p dY[p][out] f[n(p)][e] = sum_n (sum over the product nodes p of n of dY[p][out]) f[n][e]:
// the gradient rows the passes keep in `gr` (do, dt; dz0 when TrArgs.store_dz0) are summed per station / per source node
// (k_gr_sum_sta + k_gr_sum_parts / k_gr_sum_src, fixed order), then contracted with the table (k_static_dw, k_static_dw_sum: fixed order too).
constexpr int SG_CHUNKS = 256, SG_SLICES = 64;
__global__ __launch_bounds__(256) void k_gr_sum_src(const float* __restrict__ blk, int S, int G, float* __restrict__ out) {   // out[g][16]
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int st = j; st < S; st += 16) s += *(const f32x4*)(blk + ((size_t)g * S + st) * 16 + 4 * q);
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) { s.x += __shfl_xor(s.x, d); s.y += __shfl_xor(s.y, d); s.z += __shfl_xor(s.z, d); s.w += __shfl_xor(s.w, d); }
    if (j == 0) *(f32x4*)(out + (size_t)g * 16 + 4 * q) = s;
}
// per-station sums, two steps: chunk c of the source nodes -> part[c][s][16] (four independent running sums per thread, combined in
// a fixed order), then the chunks in order -> out[s][16]
__global__ __launch_bounds__(256) void k_gr_sum_sta(const float* __restrict__ blk, int S, int G, float* __restrict__ part) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // (s, q)
    if (idx >= S * 4) return;
    const int ch = blockIdx.y;
    const int g0 = (int)((long long)G * ch / SG_CHUNKS), g1 = (int)((long long)G * (ch + 1) / SG_CHUNKS);
    const float* b = blk + (size_t)idx * 4;
    const size_t gs = (size_t)S * 16;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int g = g0;
    for (; g + 3 < g1; g += 4) {
        s0 += *(const f32x4*)(b + (size_t)g * gs); s1 += *(const f32x4*)(b + (size_t)(g + 1) * gs);
        s2 += *(const f32x4*)(b + (size_t)(g + 2) * gs); s3 += *(const f32x4*)(b + (size_t)(g + 3) * gs);
    }
    for (; g < g1; ++g) s0 += *(const f32x4*)(b + (size_t)g * gs);
    *(f32x4*)(part + ((size_t)ch * S * 4 + idx) * 4) = (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(64) void k_gr_sum_parts(const float* __restrict__ part, int nparts, int n, float* __restrict__ out) {
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= n) return;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};        // eight loads in flight per thread, combined in a fixed order
    int k = 0;
    for (; k + 7 < nparts; k += 8)
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += part[(size_t)(k + u) * n + idx];
    for (; k < nparts; ++k) s[0] += part[(size_t)k * n + idx];
    out[idx] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}
// R[n][16] contracted with f[n][4]: slice x of the nodes -> part[x][16 out][4 e]; the slice's nodes are taken by four groups of
// 64 threads (node i of the slice by group i % 4) whose sums are added in group order
__global__ __launch_bounds__(256) void k_static_dw(const float* __restrict__ R, const float* __restrict__ f, int n, float* __restrict__ part) {
    __shared__ float ps[4][64];
    const int t = threadIdx.x & 63, sub = threadIdx.x >> 6, o = t & 15, e = t >> 4, x = blockIdx.x;
    const int n0 = (int)((long long)n * x / SG_SLICES), n1 = (int)((long long)n * (x + 1) / SG_SLICES);
    float s = 0.f;
    for (int i = n0 + sub; i < n1; i += 4) s += R[(size_t)i * 16 + o] * f[(size_t)i * 4 + e];
    ps[sub][t] = s;
    __syncthreads();
    if (sub == 0) part[x * 64 + t] = (ps[0][t] + ps[1][t]) + (ps[2][t] + ps[3][t]);
}
__global__ __launch_bounds__(64) void k_static_dw_sum(const float* __restrict__ part, int rows, int nf, float* __restrict__ dW, int ld, int row0,
                                                       int col0) {
    const int o = threadIdx.x & 15, e = threadIdx.x >> 4;
    if (o >= rows || e >= nf) return;
    float s = 0.f;
    for (int x = 0; x < SG_SLICES; ++x) s += part[x * 64 + threadIdx.x];
    dW[(row0 + o) * ld + col0 + e] = s;
}

// irregular product graph: r[g][32] = sum of the 32-float rows of the product nodes of source node g (seg[g] .. seg[g + 1]), in row order
__global__ void k_seg_sum32(const float* __restrict__ rows, const int32_t* __restrict__ seg, int G, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= G * 32) return;
    const int g = idx >> 5, c = idx & 31;
    float s = 0.f;
    for (long long pr = seg[g]; pr < seg[g + 1]; ++pr) s += rows[pr * 32 + c];
    out[idx] = s;
}

__global__ void k_part_sum(const float* __restrict__ part, int G, int T, float* __restrict__ r_out) {   // r[g] = sum over tiles
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= G * 30) return;
    const int g = idx / 30, c = idx - g * 30;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += part[((size_t)g * T + t) * 32 + (c < 16 ? c : c)];
    r_out[idx] = s;
}

// sum over the 16 lanes of a DPP row (all lanes end with the total)
__device__ __forceinline__ float row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}
