// Micro-benchmark 2: does v_mfma_f32_32x32x16_bf16 overlap with fp32 VALU work (it does not for fp32 MFMAs, see mfma_peak.hip),
// and how many VALU ops per MFMA are free? Also: LDS A-fragment fetch (one ds_read_b128 per MFMA).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, int LDS>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters) {
    __shared__ f32x4 lw[4608];   // 72 KB
    for (int i = threadIdx.x; i < 4608; i += 512) lw[i] = f32x4{in[i & 1023], 1.f, 2.f, 3.f};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    f32x4 xa = {in[threadIdx.x], in[threadIdx.x + 1], in[threadIdx.x + 2], in[threadIdx.x + 3]};
    bf16x8 a = __builtin_bit_cast(bf16x8, xa), b = a;
    float s[16];
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    int wi = lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (LDS) { a = __builtin_bit_cast(bf16x8, lw[wi]); wi += 64; if (wi >= 4608) wi = lane; }
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < NV; ++r) s[r & 15] += __builtin_fabsf(acc[(k + 2) & 3][r & 15]);
            asm volatile("" : "+v"(s[0]), "+v"(s[1]));
        }
    }
    float t = 0.f;
    for (int r = 0; r < 16; ++r) t += s[r];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) t += acc[k][r];
    out[blockIdx.x * 512 + threadIdx.x] = t;
}

template <int NV, int LDS>
static void run(float* out, float* in) {
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, LDS><<<blocks, 512>>>(out, in, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NV, LDS><<<blocks, 512>>>(out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * 8 * iters * 4;
    const double cyc = ms * 1e-3 * 2.4e9 / (2.0 * iters * 4);   // cycles per MFMA per SIMD at 2.4 GHz (2 waves / SIMD)
    printf("valu/mfma=%2d lds=%d: %.3f ms, %.0f TFLOP/s bf16, %.1f cyc/MFMA/SIMD @2.4GHz\n", NV, LDS, ms, mfma * 32768 / (ms * 1e-3) / 1e12, cyc);
}

int main() {
    float *out, *in;
    (void)hipMalloc(&out, 1024 * 512 * 4); (void)hipMalloc(&in, 8192 * 4);
    (void)hipMemset(in, 0, 8192 * 4);
    run<0, 0>(out, in); run<4, 0>(out, in); run<8, 0>(out, in); run<12, 0>(out, in); run<16, 0>(out, in); run<24, 0>(out, in);
    run<0, 1>(out, in); run<8, 1>(out, in); run<16, 1>(out, in);
    return 0;
}
