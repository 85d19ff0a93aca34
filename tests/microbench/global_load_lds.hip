// global_load_lds_dwordx4 on gfx950: semantics check (round 5).  Each lane names its own global address; the LDS destination is
// M0 base + instruction offset + 16 * lane — contiguous per wave whatever the global addresses are.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tests/microbench/global_load_lds.hip -o /tmp/gll && /tmp/gll
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* g, double* out, int stride) {
    __shared__ double buf[4 * 128];
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    // lane l of wave w loads the pair at row (l >> 3), column pair (l & 7) of an 8 x 16 block whose rows are `stride` doubles apart
    const double* src = g + (size_t)w * 8 * stride + (size_t)(l >> 3) * stride + 2 * (l & 7);
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)(buf + w * 128), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    out[2 * t] = buf[2 * t]; out[2 * t + 1] = buf[2 * t + 1];
}
int main() {
    const int stride = 220, n = 32 * stride;
    std::vector<double> h(n); for (int i = 0; i < n; i++) h[i] = i;
    double *g, *o; hipMalloc(&g, n * 8); hipMalloc(&o, 512 * 8);
    hipMemcpy(g, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, g, o, stride);
    std::vector<double> r(512); hipMemcpy(r.data(), o, 512 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; t++) { int w = t >> 6, l = t & 63; double e = (double)(w * 8 * stride + (l >> 3) * stride + 2 * (l & 7)); if (r[2 * t] != e || r[2 * t + 1] != e + 1) bad++; }
    printf("global_load_lds_dwordx4: %s (%d mismatches); lane 9 of wave 1 got %.0f %.0f\n", bad ? "UNEXPECTED" : "LDS destination = base + 16 * lane, per-lane global addresses", bad, r[2 * (64 + 9)], r[2 * (64 + 9) + 1]);
    return bad != 0;
}
