// xcd_local_barrier.hip — what a grid-wide hand-off costs on this 8-XCD part when the workgroups that take part sit on ONE XCD (round 6).
// Question behind it ("after round 6"): the one-window path is five launches of ~5 us floor each; one persistent grid per
// solve needs a barrier between its phases whose data stays in an L2.  Device-scope release / acquire is an L2 write-back / invalidate on
// this part (rounds 4 - 5 measured what that costs); workgroups of one XCD share an L2, so stores that have reached it and loads that
// bypass the CU's vector L1 (scope bits: the relaxed agent-scope atomics of HIP) are coherent WITHOUT either.
//   1. where do the workgroups of a launch land?  (XCC_ID per workgroup: is it blockIdx % 8?)
//   2. a barrier among NB workgroups of one XCD (every eighth workgroup of a launch; the others leave at once): one atomic add on a
//      counter + polling it, payload written with plain stores and read back with relaxed agent-scope loads, no fence: time per
//      round, and how many payload values came back stale;
//   3. the same among NB workgroups spread over all XCDs with agent-scope release / acquire fences (the correct protocol there);
//   4. the same spread over all XCDs WITHOUT fences (how many stale values: the control that shows 2. is not luck).
// Bounded spins everywhere (a round that does not complete is counted, never waited for).
//   hipcc --offload-arch=gfx950 -O3 tests/microbench/xcd_local_barrier.hip -o /tmp/xlb && /tmp/xlb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_where(int* xcc) {
    if (threadIdx.x == 0) xcc[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15;      // HW_REG_XCC_ID, bits 3:0
}

// MODE 0: participants = every eighth workgroup (one XCD if the dispatcher deals round-robin), no fences, payload read with relaxed
//         agent-scope loads; MODE 1: participants = the first NB workgroups (all XCDs), release / acquire fences; MODE 2: as 1 without fences
template <int MODE>
__global__ void k_rounds(unsigned* counter, double* payload, int nb, int rounds, unsigned long long* out /* [0] stale, [1] timeouts, [2] cycles of block 0 */) {
    const int b = MODE == 0 ? ((blockIdx.x & 7) == 0 ? (int)(blockIdx.x >> 3) : -1) : ((int)blockIdx.x < nb ? (int)blockIdx.x : -1);
    if (b < 0 || b >= nb) return;
    const int t = threadIdx.x;
    unsigned long long stale = 0, timeouts = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; r++) {
        // payload of this round: 64 doubles per workgroup, double-buffered by round parity
        double* mine = payload + ((size_t)(r & 1) * nb + b) * 64;
        if (t < 64) mine[t] = (double)(r * 1000 + b);
        if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();                                   // the workgroup's stores are issued (and, for MODE 1, released)
        if (t == 0) {
            __builtin_amdgcn_s_waitcnt(0);                 // ... and have left the CU
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * (unsigned)nb;
            int spin = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && spin < (1 << 20)) { spin++; __builtin_amdgcn_s_sleep(1); }
            if (spin >= (1 << 20)) timeouts++;
        }
        __syncthreads();
        if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // read the neighbour's payload
        const int nbr = (b + 1) % nb;
        const double* theirs = payload + ((size_t)(r & 1) * nb + nbr) * 64;
        if (t < 64) {
            const double v = MODE == 1 ? theirs[t] : __hip_atomic_load(theirs + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (double)(r * 1000 + nbr)) stale++;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (stale) atomicAdd(&out[0], stale);
    if (timeouts) atomicAdd(&out[1], timeouts);
    if (b == 0 && t == 0) out[2] = t1 - t0;
}

template <int MODE>
static int run(const char* what, int nb, int rounds) {
    unsigned* counter; double* payload; unsigned long long* out;
    CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&payload, (size_t)2 * nb * 64 * 8)); CHECK(hipMalloc(&out, 24));
    CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(payload, 0, (size_t)2 * nb * 64 * 8)); CHECK(hipMemset(out, 0, 24));
    const int grid = MODE == 0 ? 8 * nb : nb;
    CHECK(hipDeviceSynchronize());
    const auto w0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_rounds<MODE>, dim3(grid), dim3(256), 0, 0, counter, payload, nb, rounds, out);
    CHECK(hipDeviceSynchronize());
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
    unsigned long long h[3]; CHECK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
    printf("%-78s %2d workgroups, %d rounds: %.2f us per round (wall), stale payload values %llu, timeouts %llu\n", what, nb, rounds, us / rounds, h[0], h[1]);
    (void)hipFree(counter); (void)hipFree(payload); (void)hipFree(out);
    return 0;
}

int main() {
    const int NBLK = 64;
    int* d; CHECK(hipMalloc(&d, NBLK * 4));
    hipLaunchKernelGGL(k_where, dim3(NBLK), dim3(64), 0, 0, d);
    std::vector<int> h(NBLK); CHECK(hipMemcpy(h.data(), d, NBLK * 4, hipMemcpyDeviceToHost));
    int rr = 1; for (int i = 0; i < NBLK; i++) rr = rr && (h[i] == h[i & 7]);
    int distinct = 1; for (int i = 1; i < 8; i++) for (int j = 0; j < i; j++) if (h[i] == h[j]) distinct = 0;
    printf("XCC_ID of workgroups 0..15:"); for (int i = 0; i < 16; i++) printf(" %d", h[i]);
    printf("   -> workgroup i sits on the XCD of workgroup i mod 8: %s; the first eight on eight different XCDs: %s\n", rr ? "yes" : "NO", distinct ? "yes" : "NO");
    for (int nb : { 8, 20, 32 }) {
        if (run<0>("one XCD (every eighth workgroup), plain stores + relaxed agent-scope loads, no fence", nb, 2000)) return 1;
        if (run<1>("all XCDs (first workgroups), agent-scope release / acquire fences", nb, 2000)) return 1;
        if (run<2>("all XCDs (first workgroups), NO fences (control)", nb, 2000)) return 1;
    }
    return 0;
}
