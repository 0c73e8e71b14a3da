// Micro-benchmark: what fp32 MFMA rate does an MI355X actually sustain for the instruction mix of k_stage1?
//   A: independent v_mfma_f32_16x16x4_f32 only            B: + 6 VALU ops per MFMA (phase A of stage 1)
//   C: + one ds_read_b128 of a weight fragment per 4 MFMAs (dense phase)      D: B and C together
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters) {
    __shared__ f32x4 lw[3456];   // 54 KB like the stage-1 weight image
    for (int i = threadIdx.x; i < 3456; i += 512) lw[i] = f32x4{in[i & 1023], 1.f, 2.f, 3.f};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float x = in[threadIdx.x], y = in[threadIdx.x + 512];
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int wi = lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; k += 4) {
            f32x4 w = {x, y, x, y};
            if (MODE & 2) { w = lw[wi]; wi += 64; if (wi >= 3456) wi = lane; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[k + r] = MFMA16(w[r], x, acc[k + r]);
                if (MODE & 1) {
                    const float t0 = acc[(k + r + 4) & 7][0] * y, t1 = acc[(k + r + 4) & 7][1] * y;
                    s[0] += fmaxf(acc[(k + r + 4) & 7][0], t0);
                    s[1] += fmaxf(acc[(k + r + 4) & 7][1], t1);
                    asm volatile("" : "+v"(s), "+v"(x));
                }
            }
        }
    }
    f32x4 t = s;
    for (int k = 0; k < 8; ++k) t += acc[k];
    out[blockIdx.x * 512 + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
}

template <int MODE>
static void run(const char* name, float* out, float* in, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 512>>>(out, in, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 512>>>(out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * 8 * iters * 8;
    const double tf = mfma * 2048 / (ms * 1e-3) / 1e12;
    // 32 cycles per MFMA per SIMD: implied clock if the pipe were 100 % busy
    const double ghz = mfma / (256.0 * 4) * 32 / (ms * 1e-3) / 1e9 * (256.0 / blocks);
    printf("%s: %.3f ms, %.1f TFLOP/s fp32 MFMA, pipe-busy-equivalent clock %.2f GHz (blocks=%d)\n", name, ms, tf, ghz, blocks);
}

int main() {
    float *out, *in;
    hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&in, 8192 * 4);
    hipMemset(in, 0, 8192 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("A mfma only        ", out, in, 256);
        run<1>("B mfma + valu      ", out, in, 256);
        run<2>("C mfma + lds       ", out, in, 256);
        run<3>("D mfma + valu + lds", out, in, 256);
    }
    return 0;
}
