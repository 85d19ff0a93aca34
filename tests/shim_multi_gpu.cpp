// One host thread, every GPU of the node: the C-ABI's in-process multi-device entry (include/swf_solver.h, "several GPUs of one
// node from ONE process"; SURVEY.md 8b "swf_solve_batch(handles[], n, device_mask)", 8e "one host thread + one HIP stream per GPU").
// Independent windows are dealt to the devices in contiguous blocks; there is no data-path collective.  The windows here are the
// smallest the engine accepts — a few scalar blocks anchored by InitialBlackFactor residuals r = w x (R/factor/initial_factor.cpp:81-87)
// and tied in pairs by FixedIntegerFactor residuals (R/factor/gnss_factor.cpp:85-96) — so that the example needs no generator; the
// estimator's real windows go through exactly the same three calls.
//   g++ -std=c++17 -Iinclude tests/shim_multi_gpu.cpp -L<libdir> -lswf_hip ... ; ./shim_multi_gpu [device_mask]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "swf_solver.h"

struct TinyWindow {
    std::vector<double> sc; std::vector<uint8_t> is_const; std::vector<int32_t> order_block, order_group, sp_idx, fix_idx;
    std::vector<double> sp_w, fix_dat;
    swf_flat_window w;
    explicit TinyWindow(int seed) {
        const int n = 6;
        for (int i = 0; i < n; i++) { sc.push_back(0.5 + 0.1 * i + 0.01 * seed); is_const.push_back(0); order_block.push_back(i); order_group.push_back(i + 1); }
        for (int i = 0; i < n; i++) { sp_idx.push_back(i); sp_w.push_back(2.0 + i); }                 // r = w x: the minimiser is x = 0 ...
        for (int i = 0; i + 1 < n; i += 2) { fix_idx.push_back(i); fix_idx.push_back(i + 1); fix_dat.push_back(0.0); fix_dat.push_back(3.0); }   // ... and x_b - x_a = 0 agrees
        std::memset(&w, 0, sizeof w);
        w.n_sc = n; w.sc = sc.data(); w.is_const = is_const.data();
        w.n_order = n; w.order_block = order_block.data(); w.order_group = order_group.data(); w.n_tail = 0;
        w.n_sp = n; w.sp_idx = sp_idx.data(); w.sp_w = sp_w.data();
        w.n_fix = (int32_t)fix_idx.size() / 2; w.fix_idx = fix_idx.data(); w.fix_dat = fix_dat.data();
        w.proj_sqrt_info = 1.0;
    }
};

int main(int argc, char** argv) {
    const uint32_t mask = argc > 1 ? (uint32_t)std::strtoul(argv[1], nullptr, 0) : 0u;         // 0 = every visible device
    int32_t ndev = 0;
    if (swf_device_count(&ndev) != SWF_OK || ndev <= 0) { std::printf("no HIP device: %s\n", swf_last_error()); return 1; }
    const int n_win = 37;                                                                       // not a multiple of anything
    std::vector<TinyWindow*> wins; std::vector<const swf_flat_window*> ptrs;
    for (int i = 0; i < n_win; i++) { wins.push_back(new TinyWindow(i)); ptrs.push_back(&wins.back()->w); }
    std::vector<swf_batch*> batches((size_t)ndev, nullptr); std::vector<int32_t> first((size_t)ndev), count((size_t)ndev);
    int32_t nb = 0;
    if (swf_batch_create_sharded(ptrs.data(), n_win, mask, batches.data(), first.data(), count.data(), &nb) != SWF_OK) { std::printf("create: %s\n", swf_last_error()); return 1; }
    swf_options opt; swf_default_options(&opt);
    if (swf_solve_batches(batches.data(), nb, &opt) != SWF_OK) { std::printf("solve: %s\n", swf_last_error()); return 1; }
    int covered = 0; double worst = 0;
    for (int k = 0; k < nb; k++) {
        int32_t dev = -1; swf_batch_device(batches[k], &dev);
        if (swf_batch_download_state(batches[k]) != SWF_OK) { std::printf("download: %s\n", swf_last_error()); return 1; }
        std::vector<swf_summary> sm((size_t)count[k]);
        swf_batch_summaries(batches[k], sm.data());
        for (int i = 0; i < count[k]; i++) for (double x : wins[(size_t)(first[k] + i)]->sc) worst = std::fmax(worst, std::fabs(x));
        std::printf("device %d: windows [%d, %d), final cost of the first %.3e\n", dev, first[k], first[k] + count[k], sm[0].final_cost);
        covered += count[k];
        swf_batch_destroy(batches[k]);
    }
    std::printf("multi-gpu: %d batches on %d visible devices, %d of %d windows solved, max |x| = %.2e\n", nb, ndev, covered, n_win, worst);
    for (auto* w : wins) delete w;
    return (covered == n_win && worst < 1e-8) ? 0 : 2;
}
