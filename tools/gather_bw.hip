// Micro-benchmark 6: wave-level row gathers as in stage 2: every load instruction fetches 16 rows x 64 B (4 lanes x 16 B per
// row) at pseudo-random row positions inside a working set of `ws_bytes`; 15 loads in flight per wave iteration.
// Reports bytes / clock / CU for working sets that fit L1 (16 KB), L2 (2 MB per XCD), Infinity Cache (128 MB) and HBM (2 GB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CONTIG>
__global__ __launch_bounds__(256) void k(const char* __restrict__ buf, unsigned rows_mask, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    const unsigned j = lane >> 2, q = lane & 3;
    unsigned s = (blockIdx.x * 256 + threadIdx.x) / 64 * 2654435761u + 12345u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        f32x4 v[15];
#pragma unroll
        for (int k = 0; k < 15; ++k) {
            s = s * 1664525u + 1013904223u;
            // CONTIG: the 16 rows of one load are consecutive (1 KB block); else 16 independent rows
            const unsigned row = CONTIG ? (((s >> 8) & rows_mask & ~15u) + j) : (((s >> 8) + j * 2654435761u) & rows_mask);
            v[k] = *(const f32x4*)(buf + (size_t)row * 64 + q * 16);
        }
#pragma unroll
        for (int k = 0; k < 15; ++k) acc += v[k];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int CONTIG>
static void run(const char* name, const char* buf, size_t ws_bytes, float* out, int blocks_per_cu) {
    const int iters = 400, blocks = 256 * blocks_per_cu;
    const unsigned rows_mask = (unsigned)(ws_bytes / 64 - 1);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<CONTIG><<<blocks, 256>>>(buf, rows_mask, out, 20);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<CONTIG><<<blocks, 256>>>(buf, rows_mask, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 4 * iters * 15 * 1024;
    printf("%-28s %s wg/CU=%d: %.3f ms, %.2f TB/s, %.1f B/clk/CU @2.4GHz\n", name, CONTIG ? "1KB-blocks" : "64B-rows  ", blocks_per_cu, ms,
           bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
    char* buf; float* out;
    const size_t big = 2ull << 30;
    (void)hipMalloc(&buf, big); (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    (void)hipMemset(buf, 0, big);
    for (int bpc = 3; bpc <= 6; bpc += 3) {
        run<1>("L1-resident (16 KB)", buf, 16 << 10, out, bpc);
        run<0>("L1-resident (16 KB)", buf, 16 << 10, out, bpc);
        run<1>("L2-resident (2 MB)", buf, 2 << 20, out, bpc);
        run<0>("L2-resident (2 MB)", buf, 2 << 20, out, bpc);
        run<1>("MALL-resident (128 MB)", buf, 128 << 20, out, bpc);
        run<0>("MALL-resident (128 MB)", buf, 128 << 20, out, bpc);
        run<1>("HBM (2 GB)", buf, big, out, bpc);
        run<0>("HBM (2 GB)", buf, big, out, bpc);
    }
    return 0;
}
