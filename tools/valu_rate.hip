// Micro-benchmark 3: issue rate of the VALU ops stage 1 is made of (cycles per wave64 instruction per SIMD at 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters) {
    float v[16], w[16];
    for (int r = 0; r < 16; ++r) { v[r] = in[threadIdx.x + r]; w[r] = in[threadIdx.x + 64 + r]; }
    const float c = in[5];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[r]) : "v"(w[r]), "v"(c));
                if (OP == 1) asm volatile("v_add_f32 %0, %0, |%1|" : "+v"(v[r]) : "v"(w[r]));
                if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(w[r]));
                if (OP == 3) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[r]) : "v"(w[r]));
                if (OP == 4) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(w[r]), "v"(c));
                if (OP == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[r]) : "v"(w[r]));
                if (OP == 6) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[r]) : "v"(w[r]));
                if (OP == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[r]) : "v"(w[r]));
                if (OP == 8 && (r & 1) == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(f32x2*)&v[r]) : "v"(*(f32x2*)&w[r]), "v"(*(f32x2*)&w[(r + 2) & 15]));
                if (OP == 9) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[r]) : "v"(w[r]));
                if (OP == 10) asm volatile("v_lshlrev_b32 %0, 16, %1" : "+v"(v[r]) : "v"(w[r]));
            }
        }
    }
    float t = 0.f;
    for (int r = 0; r < 16; ++r) t += v[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int OP>
static void run(const char* name, float* out, float* in, int threads) {
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(out, in, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, threads>>>(out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (OP == 8 ? 32.0 : 64.0) * iters * (threads / 256);   // instructions per SIMD
    printf("%-22s waves/SIMD=%d: %.3f ms, %.2f cyc per instruction per SIMD @2.4GHz\n", name, threads / 256, ms, ms * 1e-3 * 2.4e9 / n);
}

int main() {
    float *out, *in;
    (void)hipMalloc(&out, 1024 * 1024 * 4); (void)hipMalloc(&in, 8192 * 4);
    (void)hipMemset(in, 0, 8192 * 4);
    for (int th = 256; th <= 1024; th *= 2) {
        run<0>("v_fma_f32", out, in, th);
        run<1>("v_add_f32 |abs| (vop3)", out, in, th);
        run<2>("v_add_f32", out, in, th);
        run<3>("v_and_b32", out, in, th);
        run<4>("v_perm_b32", out, in, th);
        run<5>("v_cvt_pk_bf16_f32", out, in, th);
        run<6>("v_max_f32", out, in, th);
        run<7>("v_mul_f32", out, in, th);
        run<8>("v_pk_fma_f32", out, in, th);
        run<9>("v_sub_f32", out, in, th);
        run<10>("v_lshlrev_b32", out, in, th);
    }
    return 0;
}
