// Micro-benchmark 4: v_mfma_f32_32x32x16_bf16 beside VALU work. DEP = 0: fillers touch registers no MFMA writes;
// DEP = d > 0: fillers read the accumulator written d MFMAs earlier (8 accumulators in rotation).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, int DEP, int FP32>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters) {
    f32x16 acc[8];
    for (int k = 0; k < 8; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    f32x4 xa = {in[threadIdx.x], in[threadIdx.x + 1], in[threadIdx.x + 2], in[threadIdx.x + 3]};
    bf16x8 a = __builtin_bit_cast(bf16x8, xa), b = a;
    float s[16], w[16];
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; w[r] = in[threadIdx.x + r]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (FP32) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[0], xa[1], acc[k], 0, 0, 0);
            else acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                if (DEP == 0) asm volatile("v_add_f32 %0, %0, |%1|" : "+v"(s[r & 15]) : "v"(w[r & 15]));
                else asm volatile("v_add_f32 %0, %0, |%1|" : "+v"(s[r & 15]) : "v"(acc[(k + 8 - DEP) & 7][r & 15]));
            }
        }
    }
    float t = 0.f;
    for (int r = 0; r < 16; ++r) t += s[r];
    for (int k = 0; k < 8; ++k) for (int r = 0; r < 16; ++r) t += acc[k][r];
    out[blockIdx.x * 512 + threadIdx.x] = t;
}

template <int NV, int DEP, int FP32>
static void run(float* out, float* in) {
    const int iters = 10000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, DEP, FP32><<<blocks, 512>>>(out, in, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NV, DEP, FP32><<<blocks, 512>>>(out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s valu/mfma=%2d dep=%d: %.3f ms, %.1f cyc per (MFMA + fillers) per SIMD @2.4GHz\n", FP32 ? "f32 32x32x2  " : "bf16 32x32x16", NV, DEP, ms,
           ms * 1e-3 * 2.4e9 / (2.0 * iters * 8));
}

int main() {
    float *out, *in;
    (void)hipMalloc(&out, 1024 * 512 * 4); (void)hipMalloc(&in, 8192 * 4);
    (void)hipMemset(in, 0, 8192 * 4);
    run<0, 0, 0>(out, in); run<4, 0, 0>(out, in); run<8, 0, 0>(out, in); run<12, 0, 0>(out, in); run<16, 0, 0>(out, in);
    run<4, 1, 0>(out, in); run<4, 2, 0>(out, in); run<4, 4, 0>(out, in); run<4, 6, 0>(out, in);
    run<8, 2, 0>(out, in); run<8, 4, 0>(out, in); run<16, 4, 0>(out, in);
    run<0, 0, 1>(out, in); run<8, 0, 1>(out, in); run<16, 0, 1>(out, in); run<8, 4, 1>(out, in);
    return 0;
}
