// Same question as tools/mfma_overlap.hip for the 16x16x32 fp16 MFMA (half the work of a 32x32x16, 4 accumulator registers):
// would a 16-node-per-wave stage 1 at 3-4 waves per SIMD issue the same work faster than the 32-node form at 2 waves?
// Work unit = one 32x32x16 MFMA + NF v_fma_f32, or two 16x16x32 MFMAs + NF v_fma_f32.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_overlap2 tools/mfma_overlap2.hip && /tmp/mfma_overlap2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SMALL, int NF, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters, float seed) {
    f32x16 acc = {};
    f32x4 a0 = {}, a1 = {};
    u32x4 A = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, B = A;
    float f[16];
    for (int r = 0; r < 16; ++r) f[r] = seed * (r + threadIdx.x);
    const float c = seed;
    for (int it = 0; it < iters; ++it) {
        if (SMALL) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a0) : "v"(A), "v"(B));
#pragma unroll
            for (int r = 0; r < NF / 2; ++r) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[r % 16]) : "v"(c));
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a1) : "v"(A), "v"(B));
#pragma unroll
            for (int r = NF / 2; r < NF; ++r) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[r % 16]) : "v"(c));
        } else {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B));
#pragma unroll
            for (int r = 0; r < NF; ++r) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[r % 16]) : "v"(c));
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += f[r] + acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + a0[0] + a1[1];
}

template <int SMALL, int NF, int WAVES>
void run(float* d) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<SMALL, NF, WAVES><<<256, 64 * WAVES>>>(d, 200, 1e-3f);
    (void)hipEventRecord(e0);
    k<SMALL, NF, WAVES><<<256, 64 * WAVES>>>(d, iters, 1e-3f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // a wave of the small form covers half the nodes: per unit of NODE work (32 nodes x one K step) a SIMD spends
    const double ns_unit = ms * 1e6 / iters / (WAVES / 4.0) * (SMALL ? 2.0 : 1.0) / (SMALL ? 2.0 : 1.0);
    printf("%s, %2d fma per unit, %d waves/SIMD: %.1f ns per unit and SIMD-wave-slot -> %.1f ns per unit of 32-node work at full occupancy\n",
           SMALL ? "2 x 16x16x32" : "1 x 32x32x16", NF, WAVES / 4, ns_unit, ms * 1e6 / iters / (WAVES / 4.0) * (SMALL ? 1.0 : 1.0));
}

int main() {
    float* d; (void)hipMalloc(&d, 1 << 22);
    // the 32-node form: one unit per wave-iteration, 2 waves per SIMD; the 16-node form does the same node work with two waves' worth
    run<0, 12, 8>(d);
    run<1, 12, 8>(d); run<1, 12, 12>(d); run<1, 12, 16>(d);
    run<0, 8, 8>(d);
    run<1, 8, 8>(d); run<1, 8, 12>(d); run<1, 8, 16>(d);
    run<0, 16, 8>(d);
    run<1, 16, 12>(d); run<1, 16, 16>(d);
    return 0;
}
