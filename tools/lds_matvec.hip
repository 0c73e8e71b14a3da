// Micro-benchmark: LDS cost of the 32-lanes-per-node matvec o[c] = sum_k W[k][c] x[k] used by the G-sized tail kernels.
//   A: w ds_read_b32 + x ds_read_b32 (broadcast)      B: w b32 + x as one broadcast ds_read_b128 per 4 k
//   C: w b32 + x via __shfl (ds_bpermute)              D: w b32 only (x in registers: lower bound)
//   E: w b32 + x of the wave's two nodes via v_readlane x 2 + select (no LDS for x)
// hipcc --offload-arch=gfx950 -O3 tools/lds_matvec.hip -o tools/lds_matvec.bin && ./tools/lds_matvec.bin   (*.bin is git-ignored)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 32, IT = 2000;
template <int MODE>
__global__ __launch_bounds__(256) void kern(const float* __restrict__ wsrc, float* __restrict__ out) {
    __shared__ float w[K * 32];
    __shared__ __attribute__((aligned(16))) float xin[8][K];
    for (int i = threadIdx.x; i < K * 32; i += 256) w[i] = wsrc[i];
    const int c = threadIdx.x & 31, grp = threadIdx.x >> 5;
    xin[grp][c] = wsrc[c] + grp;
    __syncthreads();
    float acc = 0.f, x = xin[grp][c];
    for (int it = 0; it < IT; ++it) {
        float o = 0.f;
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) o += w[k * 32 + c] * xin[grp][k];
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < K; k += 4) {
                const f32x4 xv = *(const f32x4*)&xin[grp][k];
                o += w[k * 32 + c] * xv.x; o += w[(k + 1) * 32 + c] * xv.y; o += w[(k + 2) * 32 + c] * xv.z; o += w[(k + 3) * 32 + c] * xv.w;
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int k = 0; k < K; ++k) o += w[k * 32 + c] * __shfl(x, k, 32);
        } else if (MODE == 4) {
            const bool hi = threadIdx.x & 32;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float sa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), k));
                const float sb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 32 + k));
                o += w[k * 32 + c] * (hi ? sb : sa);
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) o += w[k * 32 + c] * (x + k);
        }
        acc += o;
        x = o * 1e-3f + x;
        if (MODE != 2 && MODE != 3 && MODE != 4) { xin[grp][c] = x; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE>
void run(const char* name, const float* w, float* out, int wgs_per_cu) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * wgs_per_cu;
    kern<MODE><<<grid, 256>>>(w, out);
    hipEventRecord(a); kern<MODE><<<grid, 256>>>(w, out); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per CU: wgs_per_cu WGs x 4 waves x IT x K wave-FMAs
    const double fma = (double)wgs_per_cu * 4 * IT * K;
    printf("%-34s %d WG/CU: %.3f ms, %.2f cycles per wave-FMA per CU (2.4 GHz)\n", name, wgs_per_cu, ms, ms * 1e-3 * 2.4e9 / fma);
}
int main() {
    float *w, *out; hipMalloc(&w, K * 32 * 4); hipMalloc(&out, 256 * 8 * 256 * 4); hipMemset(w, 0, K * 32 * 4);
    for (int occ : {1, 2, 4}) {
        run<0>("A w b32 + x b32 broadcast", w, out, occ);
        run<1>("B w b32 + x b128 broadcast / 4k", w, out, occ);
        run<2>("C w b32 + x ds_bpermute", w, out, occ);
        run<3>("D w b32 only", w, out, occ);
        run<4>("E w b32 + x 2 readlanes + select", w, out, occ);
    }
    return 0;
}
