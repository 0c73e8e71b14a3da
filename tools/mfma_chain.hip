// Micro-benchmark 5: v_mfma_f32_32x32x16_bf16 issued round-robin over NA accumulators (dependent-chain latency).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NA>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters) {
    f32x16 acc[NA];
    for (int k = 0; k < NA; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    f32x4 xa = {in[threadIdx.x], in[threadIdx.x + 1], in[threadIdx.x + 2], in[threadIdx.x + 3]};
    bf16x8 a = __builtin_bit_cast(bf16x8, xa), b = a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12 / NA; ++u)
#pragma unroll
            for (int k = 0; k < NA; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
    }
    float t = 0.f;
    for (int k = 0; k < NA; ++k) for (int r = 0; r < 16; ++r) t += acc[k][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int NA>
static void run(float* out, float* in, int threads) {
    const int iters = 10000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NA><<<blocks, threads>>>(out, in, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NA><<<blocks, threads>>>(out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const int wps = threads / 256;
    printf("accumulators=%d waves/SIMD=%d: %.1f cyc per MFMA per SIMD @2.4GHz\n", NA, wps, ms * 1e-3 * 2.4e9 / (12.0 * iters * wps));
}

int main() {
    float *out, *in;
    (void)hipMalloc(&out, 1024 * 512 * 4); (void)hipMalloc(&in, 8192 * 4);
    (void)hipMemset(in, 0, 8192 * 4);
    for (int th = 256; th <= 512; th *= 2) { run<1>(out, in, th); run<2>(out, in, th); run<3>(out, in, th); run<4>(out, in, th); run<6>(out, in, th); }
    return 0;
}
