// Probe for the fp16 two-piece form of stage 1: does v_mfma_f32_32x32x16_f16 keep fp16 subnormal A / B inputs, does
// v_cvt_pk_f16_f32 round to nearest and produce subnormals, and what do the split sequences cost beside MFMAs?
// hipcc --offload-arch=gfx950 -O3 -o /tmp/h2_probe tools/h2_probe.hip && /tmp/h2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    unsigned r;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float sub_lo(float x, unsigned p) {     // x - float(p.lo)
    float r;
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(x));
    return r;
}
__device__ __forceinline__ float sub_hi(float x, unsigned p) {     // x - float(p.hi)
    float r;
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(x));
    return r;
}
__device__ __forceinline__ unsigned pk_scale(unsigned p, unsigned c) {
    unsigned r;
    asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(r) : "v"(p), "v"(c));
    return r;
}

__global__ void k_num(const float* a, const float* b, float* out, unsigned* pieces) {
    const int lane = threadIdx.x;
    // A[i][k] = a[k], B[k][j] = b[k] for every i, j: D = sum_k a[k] b[k] in every slot
    h8 av, bv;
    for (int e = 0; e < 8; ++e) { av[e] = (_Float16)a[8 * (lane >> 5) + e]; bv[e] = (_Float16)b[8 * (lane >> 5) + e]; }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
    out[lane] = c[0];
    // split of a[0..15]: pieces
    if (lane < 8) {
        const float x = a[16 + 2 * lane], y = a[17 + 2 * lane];
        const unsigned p0 = cvt_pk(x, y);
        const float rx = sub_lo(x, p0), ry = sub_hi(y, p0);
        const unsigned p1 = cvt_pk(rx, ry);
        const unsigned ps = pk_scale(p0, 0x2c002c00u);     // 0x2c00 = 1/16 in fp16
        pieces[3 * lane] = p0; pieces[3 * lane + 1] = p1; pieces[3 * lane + 2] = ps;
    }
}

// timing: NM MFMAs + split of NV values per iteration, two waves per SIMD
template <int FMT, int NM, int NV>
__global__ __launch_bounds__(512) void k_time(float* out, int iters, float seed) {
    f32x16 acc[2] = {};
    float v[16];
    for (int r = 0; r < 16; ++r) v[r] = seed * (r + 1 + threadIdx.x);
    u32x4 A = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, B[3] = {A, A, A};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (FMT == 0) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, A), __builtin_bit_cast(b8, B[m % 3]), acc[m & 1], 0, 0, 0);
            else acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A), __builtin_bit_cast(h8, B[m % 3]), acc[m & 1], 0, 0, 0);
        }
#pragma unroll
        for (int d = 0; d < NV / 2; ++d) {
            const float x = v[2 * d] + acc[0][2 * d], y = v[2 * d + 1] + acc[1][2 * d + 1];
            if (FMT == 0) {
                const unsigned xa = __float_as_uint(x) & 0xffff0000u, ya = __float_as_uint(y) & 0xffff0000u;
                B[0][d & 3] = __builtin_amdgcn_perm(ya, xa, 0x07060302u);
                const float rx = x - __uint_as_float(xa), ry = y - __uint_as_float(ya);
                const unsigned xb = __float_as_uint(rx) & 0xffff0000u, yb = __float_as_uint(ry) & 0xffff0000u;
                B[1][d & 3] = __builtin_amdgcn_perm(yb, xb, 0x07060302u);
                const float sx = rx - __uint_as_float(xb), sy = ry - __uint_as_float(yb);
                B[2][d & 3] = __builtin_amdgcn_perm(__float_as_uint(sy), __float_as_uint(sx), 0x07060302u);
            } else {
                const unsigned p0 = cvt_pk(x, y);
                B[0][d & 3] = p0;
                B[1][d & 3] = cvt_pk(sub_lo(x, p0), sub_hi(y, p0));
                B[2][d & 3] = pk_scale(p0, 0x2c002c00u);
            }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + __uint_as_float(B[0][0] ^ B[1][1] ^ B[2][2]);
}

template <int FMT, int NM, int NV>
void time_one(const char* name, float* dout) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_time<FMT, NM, NV><<<256, 512>>>(dout, 100, 1e-3f);
    hipEventRecord(e0);
    k_time<FMT, NM, NV><<<256, 512>>>(dout, iters, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 2 waves per SIMD: cycles per iteration per SIMD at 2.4 GHz nominal
    printf("%-40s %.3f ms  %.1f ns/iter/wave-pair\n", name, ms, ms * 1e6 / iters);
}

int main() {
    std::vector<float> a(64), b(64);
    float *da, *db, *dout; unsigned* dp;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dout, 1 << 22); hipMalloc(&dp, 256);
    struct { const char* name; float av, bv; } cases[] = {
        {"normal x normal (1 x 0.5)", 1.f, 0.5f},
        {"normal A x subnormal B (1 x 2^-20)", 1.f, ldexpf(1.f, -20)},
        {"subnormal A (2^-20) x normal B (1024)", ldexpf(1.f, -20), 1024.f},
        {"subnormal x subnormal (2^-16 x 2^-16)", ldexpf(1.f, -16), ldexpf(1.f, -16)},
        {"min subnormal B (2^-24) x 1", 1.f, ldexpf(1.f, -24)},
    };
    for (auto& c : cases) {
        for (int k = 0; k < 16; ++k) { a[k] = c.av; b[k] = c.bv; }
        const float xs[16] = {1.2345678f, -0.33333334f, 3.0517578e-5f, 1e-3f, 7.7e-6f, 123.456f, 0.1f, -2.5e-4f,
                              65504.f, 1.0009766f, 1.00048828125f, 5.9e-8f, 0.7853982f, -9.999e-2f, 2.9802322e-8f, 40000.f};
        for (int k = 0; k < 16; ++k) a[16 + k] = xs[k];
        hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
        k_num<<<1, 64>>>(da, db, dout, dp);
        float o[64]; unsigned p[24];
        hipMemcpy(o, dout, 256, hipMemcpyDeviceToHost); hipMemcpy(p, dp, 96, hipMemcpyDeviceToHost);
        printf("%-44s D = %.9g (expected %.9g)\n", c.name, o[0], 16.0 * (double)c.av * (double)c.bv);
        if (&c == &cases[0]) {
            for (int k = 0; k < 16; ++k) {
                const unsigned p0 = (p[3 * (k / 2)] >> (16 * (k & 1))) & 0xffff, p1 = (p[3 * (k / 2) + 1] >> (16 * (k & 1))) & 0xffff;
                const unsigned ps = (p[3 * (k / 2) + 2] >> (16 * (k & 1))) & 0xffff;
                _Float16 h0, h1, hs;
                __builtin_memcpy(&h0, &p0, 2); __builtin_memcpy(&h1, &p1, 2); __builtin_memcpy(&hs, &ps, 2);
                const _Float16 e0 = (_Float16)xs[k];
                const _Float16 e1 = (_Float16)(xs[k] - (float)e0);
                printf("  x = %-14.9g x0 = %-12.7g (host RN %-12.7g) x1 = %-14.7g (host %-14.7g) x0/16 = %-12.7g resid %.3g\n", xs[k], (float)h0,
                       (float)e0, (float)h1, (float)e1, (float)hs, (double)xs[k] - (double)(float)h0 - (double)(float)h1);
            }
        }
    }
    time_one<0, 6, 0>("bf16: 6 MFMA", dout);
    time_one<1, 6, 0>("f16:  6 MFMA", dout);
    time_one<0, 6, 8>("bf16: 6 MFMA + 3-way trunc split of 8", dout);
    time_one<1, 3, 8>("f16:  3 MFMA + 2-way RN split of 8 + x0/16", dout);
    time_one<1, 6, 8>("f16:  6 MFMA + 2-way RN split of 8 + x0/16", dout);
    time_one<0, 12, 16>("bf16: 12 MFMA + 3-way trunc split of 16", dout);
    time_one<1, 6, 16>("f16:  6 MFMA + 2-way RN split of 16 + x0/16", dout);
    time_one<0, 0, 16>("bf16: 3-way trunc split of 16 alone", dout);
    time_one<1, 0, 16>("f16:  2-way RN split of 16 + x0/16 alone", dout);
    return 0;
}
