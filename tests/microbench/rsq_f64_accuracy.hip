// Accuracy of v_rsq_f64 and of 1/2/3 Newton steps on it (max relative error over a log-uniform sample).
//   hipcc --offload-arch=gfx950 -O3 tests/microbench/rsq_f64_accuracy.hip -o /tmp/rsq && /tmp/rsq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* out, int n) {
#pragma clang fp contract(off)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i], y = __builtin_amdgcn_rsq(v), h = 0.5 * v;
    out[i] = y;
    y = y * __builtin_fma(-(h * y), y, 1.5); out[n + i] = y;
    y = y * __builtin_fma(-(h * y), y, 1.5); out[2 * n + i] = y;
    y = y * __builtin_fma(-(h * y), y, 1.5); out[3 * n + i] = y;
}
int main() {
    int n = 1 << 22;
    std::vector<double> x(n), o(4 * n);
    unsigned long long s = 88172645463325252ULL;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); x[i] = std::exp((u - 0.5) * 80.0) * (1.0 + u); }
    double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, 4 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(dx, dout, n);
    hipMemcpy(o.data(), dout, 4 * n * 8, hipMemcpyDeviceToHost);
    for (int st = 0; st < 4; st++) {
        double worst = 0;
        for (int i = 0; i < n; i++) { long double ref = 1.0L / sqrtl((long double)x[i]); double e = (double)fabsl(((long double)o[st * n + i] - ref) / ref); if (e > worst) worst = e; }
        printf("v_rsq_f64 + %d Newton steps: max relative error %.3e (%.2f ulp of 2^-53)\n", st, worst, worst / 1.1102230246251565e-16);
    }
    return 0;
}
