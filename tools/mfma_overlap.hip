// Does vector work overlap a 16-bit MFMA on one SIMD, and does it depend on where the accumulator lives (VGPR or AGPR) and
// on whether the vector instructions touch the MFMA's result? Two waves per SIMD, NF independent v_fma_f32 per MFMA.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_overlap tools/mfma_overlap.hip && /tmp/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int ACC, int NF, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters, float seed) {
    f32x16 acc0 = {}, acc1 = {};
    u32x4 A = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, B = A;
    float f[16];
    for (int r = 0; r < 16; ++r) f[r] = seed * (r + threadIdx.x);
    const float c = seed;
    for (int it = 0; it < iters; ++it) {
        if (ACC == 0) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(A), "v"(B));
        } else {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc0) : "v"(A), "v"(B));
        }
#pragma unroll
        for (int r = 0; r < NF; ++r) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[r % 16]) : "v"(c));
        if (ACC == 0) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(A), "v"(B));
        } else {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc1) : "v"(A), "v"(B));
        }
#pragma unroll
        for (int r = 0; r < NF; ++r) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[r % 16]) : "v"(c));
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += f[r] + acc0[r] + acc1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ACC, int NF, int WAVES>
void run(float* d) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<ACC, NF, WAVES><<<256, 64 * WAVES>>>(d, 200, 1e-3f);
    (void)hipEventRecord(e0);
    k<ACC, NF, WAVES><<<256, 64 * WAVES>>>(d, iters, 1e-3f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: WAVES / 4 waves, each 2 MFMAs + 2 NF fillers per iteration
    const double ns_per_mfma = ms * 1e6 / iters / 2.0 / (WAVES / 4.0);
    printf("acc in %s, %2d fma per MFMA, %d waves/SIMD: %.1f ns per MFMA and SIMD (%.1f cycles at 2.4 GHz)\n", ACC ? "AGPR" : "VGPR", NF,
           WAVES / 4, ns_per_mfma, ns_per_mfma * 2.4);
}

int main() {
    float* d; (void)hipMalloc(&d, 1 << 22);
    run<0, 0, 8>(d); run<1, 0, 8>(d);
    run<0, 4, 8>(d); run<1, 4, 8>(d);
    run<0, 8, 8>(d); run<1, 8, 8>(d);
    run<0, 12, 8>(d); run<1, 12, 8>(d);
    run<0, 16, 8>(d); run<1, 16, 8>(d);
    run<0, 8, 4>(d); run<1, 8, 4>(d);
    run<0, 16, 4>(d); run<1, 16, 4>(d);
    run<0, 12, 16>(d); run<1, 12, 16>(d);
    return 0;
}
