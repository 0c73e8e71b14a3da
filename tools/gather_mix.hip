// Micro-benchmark 7: L1/L2-resident row gathers (27 x dwordx4 per "tile", as stage 2) mixed with what else a stage-2 tile
// executes: MODE bit 0: 18 fp32 MFMAs 16x16x4; bit 1: 34 ds_bpermute; bit 2: ~300 VALU ops; bit 3: bf16 MFMAs instead.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const char* __restrict__ buf, unsigned rows_mask, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    const unsigned j = lane >> 2, q = lane & 3;
    unsigned s = (blockIdx.x * 256 + threadIdx.x) / 64 * 2654435761u + 12345u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, m0 = acc, m1 = acc;
    for (int it = 0; it < iters; ++it) {
        f32x4 v[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            s = s * 1664525u + 1013904223u;
            const unsigned row = (((s >> 8) & rows_mask & ~15u) + j);
            v[k] = *(const f32x4*)(buf + (size_t)row * 64 + q * 16);
        }
        f32x4 t = v[0];
#pragma unroll
        for (int k = 1; k < 27; ++k) t += v[k];
        if (MODE & 4) {
#pragma unroll
            for (int r = 0; r < 72; ++r) { t.x = fmaf(t.x, 1.0001f, t.y); t.y = fmaf(t.y, 0.9999f, t.z); t.z += t.w; t.w *= 1.00001f; }
        }
        if (MODE & 1) {
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(t.x, t.y, m0, 0, 0, 0);
                m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(t.z, t.w, m1, 0, 0, 0);
            }
            t += m0 + m1;
        }
        if (MODE & 8) {
            bf16x8 a = __builtin_bit_cast(bf16x8, t), b = a;
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                m0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, m0, 0, 0, 0);
                m1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, m1, 0, 0, 0);
            }
            t += m0 + m1;
        }
        if (MODE & 2) {
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                t.x += __shfl_xor(t.x, d); t.y += __shfl_xor(t.y, d); t.z += __shfl_xor(t.z, d); t.w += __shfl_xor(t.w, d);
            }
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                t.x += __shfl_xor(t.x, d); t.y += __shfl_xor(t.y, d); t.z += __shfl_xor(t.z, d); t.w += __shfl_xor(t.w, d);
            }
        }
        acc += t;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE>
static void run(const char* name, const char* buf, size_t ws_bytes, float* out, int blocks_per_cu) {
    const int iters = 300, blocks = 256 * blocks_per_cu;
    const unsigned rows_mask = (unsigned)(ws_bytes / 64 - 1);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(buf, rows_mask, out, 20);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(buf, rows_mask, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double tiles_per_cu = (double)blocks_per_cu * 4 * iters;
    printf("%-34s wg/CU=%d: %.3f ms, %.0f cycles per tile per CU @2.4GHz\n", name, blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / tiles_per_cu);
}

int main() {
    char* buf; float* out;
    (void)hipMalloc(&buf, 64 << 20); (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    (void)hipMemset(buf, 0, 64 << 20);
    const size_t ws = 2 << 20;
    for (int bpc = 2; bpc <= 3; ++bpc) {
        run<0>("gathers only", buf, ws, out, bpc);
        run<1>("+ 18 fp32 MFMA", buf, ws, out, bpc);
        run<2>("+ 32 bpermute", buf, ws, out, bpc);
        run<4>("+ 288 VALU", buf, ws, out, bpc);
        run<7>("+ fp32 MFMA + bpermute + VALU", buf, ws, out, bpc);
        run<14>("+ bf16 MFMA + bpermute + VALU", buf, ws, out, bpc);
    }
    return 0;
}
