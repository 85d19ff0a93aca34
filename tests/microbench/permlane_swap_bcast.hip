#include <hip/hip_runtime.h>
__global__ void k(unsigned* out, int cr) {
    unsigned x = threadIdx.x;
    auto p = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    unsigned y = (cr < 2) ? p[0] : p[1];
    auto q = __builtin_amdgcn_permlane16_swap(y, y, false, false);
    out[threadIdx.x] = (cr & 1) ? q[1] : q[0];
}
int main() {
    unsigned* d; hipMalloc(&d, 256);
    for (int cr = 0; cr < 4; cr++) {
        k<<<1, 64>>>(d, cr); unsigned h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
        printf("cr %d:", cr); for (int i = 0; i < 64; i += 5) printf(" %u", h[i]); printf("\n");
    }
}
