// Which row of D does register r of lane l hold after v_mfma_f64_16x16x4_f64?  hipcc --offload-arch=gfx950 tools/mfma_f64_layout.hip -o /tmp/f64l && /tmp/f64l
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out, int* sw) {
    const int l = threadIdx.x, i = l & 15, kq = l >> 4;
    const double a = kq == 0 ? (double)(i + 1) : 0.0;      // A[i][k]: lane i + 16 k
    const double b = kq == 0 ? 1.0 : 0.0;                  // B[k][j]: lane j + 16 k
    f64x4 c = {0., 0., 0., 0.};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
    // permlane swaps: what do they exchange?
    unsigned x = 1000 + l, y = 2000 + l;
    auto r32 = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    auto r16 = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    sw[l * 4 + 0] = r32[0]; sw[l * 4 + 1] = r32[1]; sw[l * 4 + 2] = r16[0]; sw[l * 4 + 3] = r16[1];
}
int main() {
    double* d; int* s;
    hipMalloc(&d, 64 * 4 * 8); hipMalloc(&s, 64 * 4 * 4);
    k<<<1, 64>>>(d, s);
    double h[256]; int hs[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hs, s, sizeof(hs), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 5) printf("lane %2d (j=%2d q=%d): rows %g %g %g %g | p32 %d %d  p16 %d %d\n", l, l & 15, l >> 4, h[l*4]-1, h[l*4+1]-1, h[l*4+2]-1, h[l*4+3]-1,
                                           hs[l*4], hs[l*4+1], hs[l*4+2], hs[l*4+3]);
    return 0;
}
