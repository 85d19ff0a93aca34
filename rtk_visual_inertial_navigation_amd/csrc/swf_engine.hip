// swf_engine.hip — host side of the batch engine + the swf_batch_* C-ABI (include/swf_solver.h).
//
// Symbolic phase: flat windows -> index arrays (DevBatch), once per structure.
// Numeric phase: a fixed launch sequence per solve, no host synchronisation inside.
// gfx950 only; there is no CPU path: without a HIP device every entry point fails loudly.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <memory>
#include <thread>
#include <vector>
#include "../../include/swf_solver.h"
#include "swf_kernels2.h"
#include "swf_kernels3.h"
#include "swf_kernels4.h"

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(SWF_E_NODEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

extern "C" const char* swf_last_error(void) { return g_err.c_str(); }
void swf_internal_set_error(const std::string& m) { g_err = m; }
extern "C" int swf_version(void) { return 106; }
extern "C" int swf_abi_sizes(int32_t out[5]) {
    if (!out) return fail(SWF_E_INVALID, "swf_abi_sizes: null");
    out[0] = (int32_t)sizeof(swf_options); out[1] = (int32_t)sizeof(swf_summary); out[2] = (int32_t)sizeof(swf_timing);
    out[3] = (int32_t)sizeof(swf_flat_window); out[4] = (int32_t)sizeof(swf_iteration);
    return SWF_OK;
}
extern "C" int swf_device_count(int32_t* n) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return fail(SWF_E_NODEVICE, hipGetErrorString(e)); }
    *n = c;
    return SWF_OK;
}
extern "C" int swf_set_device(int32_t d) { HIPCHK(hipSetDevice(d)); return SWF_OK; }
extern "C" void swf_default_options(swf_options* o) {
    if (!o) return;
    *o = swf_options{};
    o->max_num_iterations = 8; o->step_mode = SWF_OPTIMIZE; o->num_threads = 1; o->trust_region_strategy = SWF_DOGLEG; o->jacobi_scaling = 0;
    o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3; o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
    o->min_mu = 1e-8; o->max_mu = 1.0; o->mu_increase_factor = 10.0; o->min_diagonal = 1e-6; o->max_diagonal = 1e32;
}

// ------------------------------------------------------------------ device buffer helper
// Every buffer of a batch is carved out of a few slabs (bump allocation, 256-byte aligned): buffers with initial data share
// "data" slabs that are mirrored in one host staging area and reach the device in ONE copy per slab (flush); zero-initialised
// buffers share "zero" slabs that get one memset each.  Released slabs go to a process-wide cache instead of hipFree: the
// ceres::Problem surface rebuilds its batch at every structure change (the estimator adds and removes landmarks every frame,
// R/swf/swf_image.cpp:65-114), and ~100 hipMalloc + hipMemcpy + hipMemset calls per rebuild were most of that path's cost.
// After flush() the pool is sealed: later allocations (the marginalisation consumer's outputs) are initialised on the spot.
#include <mutex>
namespace {
struct SlabCache {
    std::mutex mu; std::vector<std::pair<void*, size_t>> free_; size_t held = 0;
    void* acquire(size_t& bytes) {
        {
            std::lock_guard<std::mutex> g(mu);
            int best = -1;
            for (int i = 0; i < (int)free_.size(); i++)
                if (free_[i].second >= bytes && free_[i].second <= 4 * bytes && (best < 0 || free_[i].second < free_[best].second)) best = i;
            if (best >= 0) { void* p = free_[best].first; bytes = free_[best].second; held -= bytes; free_.erase(free_.begin() + best); return p; }
        }
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            trim(0);                                       // give the cache back and try once more
            if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        }
        return p;
    }
    void give(void* p, size_t bytes) {
        std::lock_guard<std::mutex> g(mu);
        if (held + bytes > (size_t)2 << 30 || free_.size() >= 32) { (void)hipFree(p); return; }
        free_.push_back({ p, bytes }); held += bytes;
    }
    void trim(size_t keep) {
        std::lock_guard<std::mutex> g(mu);
        while (held > keep && !free_.empty()) { held -= free_.back().second; (void)hipFree(free_.back().first); free_.pop_back(); }
    }
};
// never destroyed: a ceres::Problem with static storage duration may release its batch after this file's statics are gone.
// One cache per device (a slab, a stream or an event belongs to the device it was created on): the CURRENT device's; every
// swf_batch_* entry point runs under the batch's device (DeviceGuard).
constexpr int SWF_MAX_DEVICES = 32;
int current_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= SWF_MAX_DEVICES) d = 0; return d; }
SlabCache& slab_cache() { static SlabCache* c = new SlabCache[SWF_MAX_DEVICES]; return c[current_device()]; }
}  // namespace

// streams and events are as expensive to create and destroy as device memory (milliseconds for a non-blocking stream): the
// auxiliary stream + fork / join events of the latency path, and the timing events, are recycled the same way
namespace {
struct HandleCache {
    std::mutex mu; std::vector<hipStream_t> streams; std::vector<hipEvent_t> sync_events, timing_events;
    hipStream_t stream() {
        { std::lock_guard<std::mutex> g(mu); if (!streams.empty()) { hipStream_t s_ = streams.back(); streams.pop_back(); return s_; } }
        hipStream_t s_ = nullptr;
        return hipStreamCreateWithFlags(&s_, hipStreamNonBlocking) == hipSuccess ? s_ : nullptr;
    }
    hipEvent_t event(bool timing) {
        {
            std::lock_guard<std::mutex> g(mu);
            auto& v = timing ? timing_events : sync_events;
            if (!v.empty()) { hipEvent_t e = v.back(); v.pop_back(); return e; }
        }
        hipEvent_t e = nullptr;
        hipError_t rc = timing ? hipEventCreate(&e) : hipEventCreateWithFlags(&e, hipEventDisableTiming);
        return rc == hipSuccess ? e : nullptr;
    }
    void give(hipStream_t s_) { if (!s_) return; std::lock_guard<std::mutex> g(mu); if (streams.size() < 16) streams.push_back(s_); else (void)hipStreamDestroy(s_); }
    void give(hipEvent_t e, bool timing) {
        if (!e) return;
        std::lock_guard<std::mutex> g(mu);
        auto& v = timing ? timing_events : sync_events;
        if (v.size() < 4096) v.push_back(e); else (void)hipEventDestroy(e);
    }
};
HandleCache& handle_cache() { static HandleCache* c = new HandleCache[SWF_MAX_DEVICES]; return c[current_device()]; }
// makes a batch's device the current one for the duration of an entry point (several batches on several GPUs may be driven from
// one host thread: swf_solve_batches)
struct DeviceGuard {
    int prev = -1; bool switched = false;
    explicit DeviceGuard(int dev) {
        if (dev < 0) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
}  // namespace

struct DevPool {
    struct Slab { char* dev = nullptr; size_t cap = 0, used = 0, flushed = 0; std::vector<char> host; };
    std::vector<Slab> data, zero;
    bool sealed = false;
    void* bump(std::vector<Slab>& v, size_t bytes, bool mirrored, size_t first) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (v.empty() || v.back().used + bytes > v.back().cap) {
            size_t want = std::max(bytes, v.empty() ? first : std::min<size_t>(2 * v.back().cap, (size_t)256 << 20));
            Slab sl;
            sl.dev = (char*)slab_cache().acquire(want);
            if (!sl.dev) return nullptr;
            sl.cap = want;
            if (mirrored) sl.host.assign(want, 0);
            v.push_back(std::move(sl));
        }
        Slab& sl = v.back();
        void* p = sl.dev + sl.used;
        sl.used += bytes;
        return p;
    }
    template <class T> int put(const std::vector<T>& h, const T** out, size_t min_elems = 1) {
        size_t n = std::max(h.size(), min_elems), bytes = n * sizeof(T);
        void* p = bump(data, bytes, true, (size_t)4 << 20);
        if (!p) return -1;
        Slab& sl = data.back();
        char* m = sl.host.data() + ((char*)p - sl.dev);
        if (!h.empty()) memcpy(m, h.data(), h.size() * sizeof(T));
        if (sealed) {                                      // late allocation: initialise now
            if (hipMemcpy(p, m, (bytes + 255) & ~(size_t)255, hipMemcpyHostToDevice) != hipSuccess) return -1;
            sl.flushed = sl.used;
        }
        *out = (const T*)p;
        return 0;
    }
    template <class T> int zeros(size_t n, T** out) {
        n = std::max<size_t>(n, 1);
        void* p = bump(zero, n * sizeof(T), false, (size_t)16 << 20);
        if (!p) return -1;
        if (sealed) { if (hipMemset(p, 0, n * sizeof(T)) != hipSuccess) return -1; zero.back().flushed = zero.back().used; }
        *out = (T*)p;
        return 0;
    }
    // one host-to-device copy per data slab, one memset per zero slab
    int flush() {
        for (Slab& sl : data) if (sl.used > sl.flushed) { if (hipMemcpy(sl.dev + sl.flushed, sl.host.data() + sl.flushed, sl.used - sl.flushed, hipMemcpyHostToDevice) != hipSuccess) return -1; sl.flushed = sl.used; }
        for (Slab& sl : zero) if (sl.used > sl.flushed) { if (hipMemset(sl.dev + sl.flushed, 0, sl.used - sl.flushed) != hipSuccess) return -1; sl.flushed = sl.used; }
        sealed = true;
        return 0;
    }
    void release() {
        for (Slab& sl : data) slab_cache().give(sl.dev, sl.cap);
        for (Slab& sl : zero) slab_cache().give(sl.dev, sl.cap);
        data.clear(); zero.clear(); sealed = false;
    }
};

struct HostWin {       // what the host keeps per window for state transfer / export
    double *pose, *sb, *lm, *sc;
    int n_pose, n_sb, n_lm, n_sc;
    int tail_dim;
    double *comp_pose = nullptr, *comp_sb = nullptr;    // hidden epochs of the window's composite factors (caller memory)
    int comp_e0 = 0, comp_ne = 0;                       // their range in the batch-wide hidden-epoch arrays
    std::vector<int> p_orig;                            // device observation (proj0 + q) -> the caller's projection factor index
    int n_proj_all = 0;                                 // the caller's projection factors, fast path + generic path (GF_PROJX)
};

struct swf_batch {
    int device = 0;                    // the HIP device the batch lives on (current at swf_batch_create)
    DevBatch D{};
    DevPool pool;
    hipStream_t stream = nullptr;
    std::vector<WinRec> win;
    std::vector<HostWin> hw;
    int max_tiles = 0, max_prior_dim = 0, max_red = 0, min_red = 1 << 30, n_cu = 256;
    bool clc_imu[5] = { false, false, false, false, false };
    // composite IMU-GNSS factors of the batch (swf_kernels4.h): operator arguments, solver-side bookkeeping, initial hidden epochs
    CompArgs CA{}; CompMeta CM{}; int n_comp = 0, comp_nmax = 0, comp_nmin = 1 << 30; long long comp_ne = 0;
    bool comp_eigen_root = false;                      // swf_options::composite_root == SWF_ROOT_EIGEN in the current solve: the composite factors expose the reference's eigen square root
    void* h_sum = nullptr; size_t h_sum_bytes = 0;       // page-locked staging of the per-window states and traces (swf_batch_summaries)
    double* h_x = nullptr;                // page-locked staging of the parameter blocks (state upload / download: one DMA instead of a pageable copy)
    double* co_pose0 = nullptr; double* co_sb0 = nullptr;    // clique class holds IMU factors (its elimination must follow k_eval_imu)
    // marginalisation consumer outputs (allocated at the first swf_batch_marginalize)
    int* mg_tail = nullptr; double* mg_A = nullptr; double* mg_b = nullptr; double* mg_J = nullptr; double* mg_r0 = nullptr; double* mg_w = nullptr; int* mg_rank = nullptr; double* mg_M = nullptr;
    bool mg_valid = false; int mg_ld = 0;
    // host mirror of the consumer's outputs, filled by the first swf_batch_get_prior after a swf_batch_marginalize of a many-window batch
    // (one copy per array instead of six small ones per window: 576 GNSS-epoch priors took 46 ms of hipMemcpy latency)
    bool mg_host = false; std::vector<double> h_mgA, h_mgJ, h_mgb, h_mgr0, h_mgw; std::vector<int> h_mgrank;
    double* mg_resM = nullptr; double* mg_resb = nullptr; int* mg_resok = nullptr;      // k_marg_rescue outputs (rank-deficient tails)
    int* mg_rot = nullptr; int* mg_bjok = nullptr; unsigned long long* mg_crit = nullptr;                                       // k_marg_bj: rotations per sweep, windows taking part
    // ambiguity covariance hand-off outputs (allocated at the first swf_batch_tail_covariance)
    int* tc_tail = nullptr; double* tc_A = nullptr; double* tc_Q = nullptr; double* tc_X = nullptr; int* tc_rank = nullptr; bool tc_valid = false; int tc_ld = 0;
    // latency path (small batches): an auxiliary stream runs the IMU / clique branch of a linearisation next to the
    // projection / landmark branch; three reusable events carry the dependencies
    hipStream_t aux = nullptr; hipEvent_t ev_fork[3] = { nullptr, nullptr, nullptr };
    ~swf_batch() {
        if (aux) { (void)hipStreamSynchronize(aux); handle_cache().give(aux); }
        for (auto& e : ev_fork) handle_cache().give(e, false);
    }
    WinState* ws_primary = nullptr; WinState* ws_alt = nullptr; WinState* ws_alt2 = nullptr;      // the per-window solver states and the two further buffers the latency path's fused kernels rotate through (k_step_eval, k_decide_lm_clique)
    bool no_decide_fuse = false;          // SWF_NO_DECIDE_FUSE=1: k_decide as its own launch on the latency path too (parity: bit-identical)
    bool no_step_fuse = false;            // SWF_NO_STEP_FUSE=1: k_dogleg and the candidate's evaluation as two launches on the latency path too (parity: bit-identical)
    int n_pch_split = 0;                  // row chunks of priors evaluated by several workgroups (dimension > PRIOR_SPLIT_DIM) with a static clique: k_prior_graw is launched
    bool no_spec = false;                 // SWF_NO_SPEC_EVAL=1: the dogleg loop with a cost pass at the candidate and a Jacobian pass behind k_decide (parity: bit-identical to the speculative flow)
    bool no_comp_fuse = false;            // SWF_NO_COMP_FUSE=1: the composite chain and the visual branch as launches of their own on the latency path too (A/B, parity)
    bool lat_fuse = false;                // latency path: fused grids on one stream (see swf_batch_create)
    bool rr4_has15 = false, rr4_has16 = false;      // some window has 224 < n_red <= 240 / 240 < n_red <= 256: the 15- / 16-column instance of k_chol_rr4 is launched as well
    int rr_nmax = 256;                    // largest reduced system of the register-resident Cholesky (k_chol_rr4)
    bool L_full = false;                  // the L buffer holds the whole factor of the last linear solve
    int asm_programs = 0;                 // distinct assembly programs of the batch (windows of identical structure share one)
    int ls_qpb = 1, ls_var = 0, ls_kms = 8; bool ls_folded = false, s_direct = false;     // k_lm_schur launch shape, fixed at creation (the pair lists depend on it)
    int timing = 0;                       // bitmask of SWF_K_* brackets
    swf_timing last{};
    std::vector<hipEvent_t> ev;           // event pool (pairs)
    std::vector<int> ev_kind;             // kernel id per recorded pair
    int ev_used = 0;
    int64_t jac_bytes = 0, proj_bytes = 0, chol_flops = 0, lm_schur_flops = 0, lm_schur_flops_sym = 0, lm_schur_mfma = 0;
    int last_mode = -1;
};

// ------------------------------------------------------------------ symbolic phase
namespace {
struct Build {
    // concatenated host arrays
    std::vector<WinRec> win;
    std::vector<int> blk_xoff, blk_loc, blk_gs, loc2x;
    std::vector<unsigned char> x_var;
    std::vector<int> p_win, p_xpose, p_xex, p_xlm, p_lpose, p_llm, p_fr, p_lm;
    std::vector<double> p_uv;
    std::vector<int> lm_win, lm_obs0, lm_loc, lm_col;
    std::vector<unsigned long long> lm_fmask;
    std::vector<int> fsb_win, fsb_obs0, fsb_perm, fsb_foff, fsb_foff0, fsb_out0;
    long long fs_tot = 0;
    std::vector<int> fr_obs0, fr_obs, fr_red;
    std::vector<GFac> gf;
    std::vector<int> s_x, s_loc, s_ls, s_joff, s_ccol;
    std::vector<double> imu_pre, cp_dat, pr_dat, dop_dat, sp_w, gx_dat;
    std::vector<int> imu_gf, sc_gf, prior_gf, idp_gf;
    std::vector<int> prior_dim, prior_roff, prior_x0off;
    std::vector<long long> prior_Joff;
    std::vector<double> prior_J, prior_r0, prior_x0;
    std::vector<Clique> cl;
    std::vector<int> cl_fac, cl_frow, cm_loc, cm_ls, cm_col;
    std::vector<double> C_init, dgraw_init;      // static parts (prior cliques)
    // composite factors (concatenated over the batch)
    std::vector<int> co_M, co_N, co_gf, co_win, co_xo, co_xo_off{ 0 };
    std::vector<double> co_pose, co_sb, co_pose_lin, co_sb_lin, co_Hpp, co_HpN, co_rhs_p, co_HNN, co_rhsN, co_pre, co_pbgw, co_H12;
    std::vector<int> co_mid;
    std::vector<Pair> pair;
    std::vector<long long> pc_coff;
    std::vector<int> pc_cld, pc_voff;
    long long n_x = 0, n_loc = 0, S_tot = 0, Lt_tot = 0, P_tot = 0, C_tot = 0;
    int v_tot = 0, e_tot = 0, r_tot = 0, j_tot = 0, n_fr = 0;
    int max_tiles = 0, max_prior_dim = 0;
    int64_t jac_bytes = 0;
};

int build_window(Build& B, const swf_flat_window* w, int wi, HostWin& hw) {
    WinRec R{};
    const int nP = w->n_pose, nS = w->n_sb, nL = w->n_lm, nC = w->n_sc;
    const int nb = nP + nS + nL + nC;
    if (nb <= 0) return fail(SWF_E_INVALID, "empty window");
    hw = HostWin{ w->pose, w->sb, w->lm, w->sc, nP, nS, nL, nC, 0 };
    R.x_base = (int)B.n_x; R.blk_base = (int)B.blk_xoff.size(); R.n_blk = nb;
    R.loc_base = (int)B.n_loc;
    std::vector<int> gs(nb), ls(nb), xo(nb), loc(nb, -1), grp(nb, -1);
    int xoff = 0;
    for (int b = 0; b < nb; b++) {
        int g = b < nP ? 7 : b < nP + nS ? 9 : b < nP + nS + nL ? 3 : 1;
        gs[b] = g; ls[b] = g == 7 ? 6 : g; xo[b] = xoff; xoff += g;
    }
    R.x_n = xoff;
    int lo = 0, ne = 0, prevg = 0;
    for (int i = 0; i < w->n_order; i++) {
        int b = w->order_block[i], g = w->order_group[i];
        if (b < 0 || b >= nb) return fail(SWF_E_INVALID, "ordering: block id out of range");
        if (w->is_const[b]) return fail(SWF_E_INVALID, "ordering: constant block in ordering");
        if (loc[b] >= 0) return fail(SWF_E_INVALID, "ordering: block listed twice");
        if (g < prevg) return fail(SWF_E_INVALID, "ordering: groups must ascend");
        prevg = g;
        loc[b] = lo; grp[b] = g; lo += ls[b];
        if (g == 0) ne += ls[b];
    }
    for (int b = 0; b < nb; b++) if (!w->is_const[b] && loc[b] < 0) return fail(SWF_E_INVALID, "ordering: variable block missing from ordering");
    R.n_loc = lo; R.n_e = ne; R.n_red = lo - ne;
    if (R.n_red + 1 > 1024) return fail(SWF_E_UNSUPPORTED, "reduced system larger than 1023");
    R.S_base = B.S_tot; B.S_tot += (long long)(R.n_red + 1) * R.n_red;      // n x n (lower used) + the reduced rhs as row n
    R.Lt_base = B.Lt_tot; B.Lt_tot += (long long)(R.n_red + 1) * (R.n_red + 1);
    {
        int td = 0;
        for (int i = w->n_order - w->n_tail; i < w->n_order; i++) if (i >= 0) td += ls[w->order_block[i]];
        hw.tail_dim = td; R.tail_dim = td;
    }
    if (nP > CTL_NT) return fail(SWF_E_UNSUPPORTED, "more than 256 pose blocks in a window");      // k_dogleg: a thread per pose block
    R.n_pose_blk = nP;
    B.loc2x.resize((size_t)R.loc_base + (size_t)R.n_loc, -1); B.x_var.resize((size_t)R.x_base + (size_t)R.x_n, 0);
    for (int b = 0; b < nb; b++) {
        B.blk_xoff.push_back(R.x_base + xo[b]);
        B.blk_loc.push_back(loc[b] >= 0 ? R.loc_base + loc[b] : -1);
        B.blk_gs.push_back(gs[b]);
        if (loc[b] < 0) continue;
        for (int k = 0; k < gs[b]; k++) B.x_var[(size_t)R.x_base + xo[b] + k] = 1;
        if (gs[b] != 7) for (int k = 0; k < gs[b]; k++) B.loc2x[(size_t)R.loc_base + loc[b] + k] = R.x_base + xo[b] + k;
    }
    auto bidP = [&](int i) { return i; };
    auto bidS = [&](int i) { return nP + i; };
    auto bidL = [&](int i) { return nP + nS + i; };
    auto bidC = [&](int i) { return nP + nS + nL + i; };
    auto is_e = [&](int b) { return grp[b] == 0; };
    auto gloc = [&](int b) { return loc[b] >= 0 ? R.loc_base + loc[b] : -1; };
    auto gx = [&](int b) { return R.x_base + xo[b]; };

    // ---- which landmarks leave the fast path (k_lm_schur: world point in group 0, constant extrinsic, one factor per frame)
    // for the generic one (GF_PROJX factors in cliques): a variable extrinsic on any of its factors — the reference's
    // marginalisation solves un-freeze para_ex_Pose (R/swf/swf_image.cpp:384-389) — or a variable landmark outside group 0
    std::vector<char> lm_generic(nL, 0);
    for (int i = 0; i < w->n_proj; i++) {
        int p = w->proj_idx[i * 3], ex = w->proj_idx[i * 3 + 1], l = w->proj_idx[i * 3 + 2];
        if (p < 0 || p >= nP || ex < 0 || ex >= nP || l < 0 || l >= nL) return fail(SWF_E_INVALID, "projection factor: index out of range");
        if (loc[bidP(ex)] >= 0 || (loc[bidL(l)] >= 0 && !is_e(bidL(l)))) lm_generic[l] = 1;
    }
    hw.n_proj_all = w->n_proj;
    // ---- projection observations of the fast path sorted by (landmark, pose)
    std::vector<int> ord;
    for (int i = 0; i < w->n_proj; i++) if (!lm_generic[w->proj_idx[i * 3 + 2]]) ord.push_back(i);
    const int n_fast = (int)ord.size();
    // (sorted below, once the landmark records have their order)
    // frames: variable, non-eliminated poses that carry observations, in pose order
    std::vector<int> frame_of(nP, -1);
    {
        std::vector<char> seen(nP, 0);
        for (int i : ord) seen[w->proj_idx[i * 3]] = 1;
        int nf = 0;
        for (int p = 0; p < nP; p++) if (seen[p] && loc[bidP(p)] >= 0) {
            if (is_e(bidP(p))) return fail(SWF_E_UNSUPPORTED, "pose block in elimination group 0");
            frame_of[p] = nf++;
            B.fr_red.push_back(loc[bidP(p)] - ne);
        }
        R.nF = nf; R.fr_base = B.n_fr;
    }
    // Landmark records — and with them the observations — are laid out in the order k_lm_schur packs them into wave tasks: by the
    // footprint of the track in the 16-row tiles of the reduced camera matrix (last tile, first tile), ties in the caller's order.
    // A wave task's four landmarks then read four adjacent runs of every Jacobian array.  (Internal order only: the elimination
    // order is that of the blocks, and hw.p_orig maps the observations back to the caller's factors.)
    std::vector<int> lm_rank(nL), lm_perm(nL);
    {
        std::vector<unsigned> trs(nL, 0u);
        for (int i : ord) {
            int f = frame_of[w->proj_idx[i * 3]];
            if (f >= 0 && f < 64) { trs[w->proj_idx[i * 3 + 2]] |= 1u << ((6 * f) / 16); trs[w->proj_idx[i * 3 + 2]] |= 1u << ((6 * f + 5) / 16); }
        }
        auto key = [&](int l) { unsigned t = trs[l]; return t ? (31 - __builtin_clz(t)) * 64 + __builtin_ctz(t) : (lm_generic[l] ? 1 << 20 : 0); };
        for (int l = 0; l < nL; l++) lm_perm[l] = l;
        std::stable_sort(lm_perm.begin(), lm_perm.end(), [&](int a, int b) { return key(a) < key(b); });
        for (int r = 0; r < nL; r++) lm_rank[lm_perm[r]] = r;
    }
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) {
        int la = lm_rank[w->proj_idx[a * 3 + 2]], lb = lm_rank[w->proj_idx[b * 3 + 2]];
        if (la != lb) return la < lb;
        return w->proj_idx[a * 3] < w->proj_idx[b * 3];
    });
    R.proj0 = (int)B.p_win.size();
    R.lm0 = (int)B.lm_win.size();
    hw.p_orig = ord;
    {
        std::vector<std::vector<int>> fobs(R.nF);
        std::vector<int> lm_first(nL + 1, 0);
        for (int q = 0; q < n_fast; q++) {
            int i = ord[q];
            int p = w->proj_idx[i * 3], ex = w->proj_idx[i * 3 + 1], l = w->proj_idx[i * 3 + 2];
            if (q > 0 && w->proj_idx[ord[q - 1] * 3 + 2] == l && w->proj_idx[ord[q - 1] * 3] == p && frame_of[p] >= 0)
                return fail(SWF_E_UNSUPPORTED, "two projection factors of one landmark in the same frame");
            int gi = (int)B.p_win.size();
            B.p_win.push_back(wi);
            B.p_xpose.push_back(gx(bidP(p))); B.p_xex.push_back(gx(bidP(ex))); B.p_xlm.push_back(gx(bidL(l)));
            B.p_lpose.push_back(gloc(bidP(p))); B.p_llm.push_back(gloc(bidL(l)));
            B.p_fr.push_back(frame_of[p]); B.p_lm.push_back(R.lm0 + lm_rank[l]);
            B.p_uv.push_back(w->proj_uv[i * 2]); B.p_uv.push_back(w->proj_uv[i * 2 + 1]);
            if (frame_of[p] >= 0) fobs[frame_of[p]].push_back(gi);
            lm_first[lm_rank[l] + 1]++;
        }
        for (int l = 0; l < nL; l++) lm_first[l + 1] += lm_first[l];
        for (int rk = 0; rk < nL; rk++) {
            const int l = lm_perm[rk];
            int b = bidL(l);
            B.lm_win.push_back(wi);
            B.lm_obs0.push_back(R.proj0 + lm_first[rk]);
            B.lm_loc.push_back(lm_generic[l] ? -1 : gloc(b));        // a generic-path landmark has no observations here: an inactive record
            B.lm_col.push_back(3 * l);
            B.lm_fmask.push_back(0ULL);
        }
        for (int q = R.proj0; q < (int)B.p_win.size(); q++) {
            int f = B.p_fr[q];
            if (f >= 0) B.lm_fmask[B.p_lm[q]] |= (f < 64) ? (1ULL << f) : ~0ULL;
        }
        if (R.nF > 64) for (int l = 0; l < nL; l++) B.lm_fmask[R.lm0 + l] = ~0ULL;   // no skipping beyond 64 frames
        for (int f = 0; f < R.nF; f++) {
            B.fr_obs0.push_back((int)B.fr_obs.size());
            for (int o : fobs[f]) B.fr_obs.push_back(o);
        }
        B.n_fr += R.nF;
    }
    R.proj1 = (int)B.p_win.size();
    R.lm1 = (int)B.lm_win.size();
    // frame-sum blocks: <= FS_BLK consecutive observations, frame-sorted permutation per block
    R.fsb0 = (int)B.fsb_win.size();
    for (int o0 = R.proj0; o0 < R.proj1; o0 += FS_BLK) {
        int cnt = std::min(FS_BLK, R.proj1 - o0);
        B.fsb_win.push_back(wi); B.fsb_obs0.push_back(o0);
        B.fsb_foff0.push_back((int)B.fsb_foff.size());
        B.fsb_out0.push_back((int)B.fs_tot); B.fs_tot += R.nF;
        std::vector<std::vector<int>> byf(R.nF);
        for (int t = 0; t < cnt; t++) { int f = B.p_fr[o0 + t]; if (f >= 0) byf[f].push_back(t); }
        // fsb_perm[o] = rank of observation o in the block's frame-sorted order (observations of constant poses go last)
        int pos = 0;
        B.fsb_perm.resize((size_t)o0 + cnt, -1);
        for (int f = 0; f < R.nF; f++) {
            B.fsb_foff.push_back(pos);
            for (int t : byf[f]) B.fsb_perm[(size_t)o0 + t] = pos++;
        }
        B.fsb_foff.push_back(pos);
        for (int t = 0; t < cnt; t++) if (B.fsb_perm[(size_t)o0 + t] < 0) B.fsb_perm[(size_t)o0 + t] = pos++;
    }
    R.fsb1 = (int)B.fsb_win.size();
    R.P_base = B.P_tot; B.P_tot += (long long)36 * R.nF * R.nF;       // x GEMM_SPLIT partial products at allocation
    {
        int m = 6 * R.nF, nt = (m + 15) / 16;
        B.max_tiles = std::max(B.max_tiles, nt * (nt + 1) / 2);
        if (R.nF > LS_MAXF) return fail(SWF_E_UNSUPPORTED, "more than 64 observing frames in one window");
    }

    // ---- generic factors
    R.gf0 = (int)B.gf.size();
    struct TmpF { std::vector<int> blk; };
    std::vector<TmpF> tf;
    auto add_gf = [&](int type, int nres, int data, const std::vector<int>& blks) {
        GFac G{};
        G.type = type; G.win = wi; G.nres = nres; G.nslot = (int)blks.size();
        G.slot0 = (int)B.s_x.size(); G.roff = -1; G.data = data; G.clique = -1;
        for (int b : blks) {
            B.s_x.push_back(gx(b)); B.s_loc.push_back(gloc(b)); B.s_ls.push_back(ls[b]);
            // Jacobian block placement (s_joff, s_jld) and the residual offset are assigned with the cliques below
            B.s_joff.push_back((loc[b] >= 0 && type != GF_PRIOR) ? 0 : -1);
            B.s_ccol.push_back(-1);
        }
        B.gf.push_back(G);
        tf.push_back(TmpF{ blks });
        return (int)B.gf.size() - 1;
    };
#define CHK(i, n, what) if ((i) < 0 || (i) >= (n)) return fail(SWF_E_INVALID, what ": index out of range");
    for (int i = 0; i < w->n_imu; i++) {
        const int* ix = w->imu_idx + i * 4;
        CHK(ix[0], nP, "imu") CHK(ix[1], nS, "imu") CHK(ix[2], nP, "imu") CHK(ix[3], nS, "imu")
        int data = (int)(B.imu_pre.size() / SWF_PRE_DOUBLES);
        B.imu_pre.insert(B.imu_pre.end(), w->imu_pre + (size_t)i * SWF_PRE_DOUBLES, w->imu_pre + (size_t)(i + 1) * SWF_PRE_DOUBLES);
        B.imu_gf.push_back(add_gf(GF_IMU, 15, data, { bidP(ix[0]), bidS(ix[1]), bidP(ix[2]), bidS(ix[3]) }));
    }
    for (int i = 0; i < w->n_cp; i++) {
        const int* ix = w->cp_idx + i * 3;
        CHK(ix[0], nP, "carrier phase") CHK(ix[1], nC, "carrier phase") CHK(ix[2], nC, "carrier phase")
        int data = (int)(B.cp_dat.size() / SWF_CP_DOUBLES);
        B.cp_dat.insert(B.cp_dat.end(), w->cp_dat + i * SWF_CP_DOUBLES, w->cp_dat + (i + 1) * SWF_CP_DOUBLES);
        B.sc_gf.push_back(add_gf(GF_CP, 1, data, { bidP(ix[0]), bidC(ix[1]), bidC(ix[2]) }));
    }
    for (int i = 0; i < w->n_pr; i++) {
        const int* ix = w->pr_idx + i * 2;
        CHK(ix[0], nP, "pseudorange") CHK(ix[1], nC, "pseudorange")
        int data = (int)(B.pr_dat.size() / SWF_PR_DOUBLES);
        B.pr_dat.insert(B.pr_dat.end(), w->pr_dat + i * SWF_PR_DOUBLES, w->pr_dat + (i + 1) * SWF_PR_DOUBLES);
        B.sc_gf.push_back(add_gf(GF_PR, 1, data, { bidP(ix[0]), bidC(ix[1]) }));
    }
    for (int i = 0; i < w->n_dop; i++) {
        const int* ix = w->dop_idx + i * 3;
        CHK(ix[0], nS, "doppler") CHK(ix[1], nC, "doppler") CHK(ix[2], nP, "doppler")
        int data = (int)(B.dop_dat.size() / SWF_DOP_DOUBLES);
        B.dop_dat.insert(B.dop_dat.end(), w->dop_dat + i * SWF_DOP_DOUBLES, w->dop_dat + (i + 1) * SWF_DOP_DOUBLES);
        B.sc_gf.push_back(add_gf(GF_DOP, 1, data, { bidS(ix[0]), bidC(ix[1]), bidP(ix[2]) }));
    }
    for (int i = 0; i < w->n_sp; i++) {
        CHK(w->sp_idx[i], nC, "scalar prior")
        int data = (int)B.sp_w.size();
        B.sp_w.push_back(w->sp_w[i]);
        B.sc_gf.push_back(add_gf(GF_SP, 1, data, { bidC(w->sp_idx[i]) }));
    }
    // rover-only pseudorange / carrier phase and fixed-integer factors share one record pool (GFac.data = offset in doubles)
    for (int i = 0; i < w->n_spr; i++) {
        const int* ix = w->spr_idx + i * 2;
        CHK(ix[0], nP, "spp pseudorange") CHK(ix[1], nC, "spp pseudorange")
        int data = (int)B.gx_dat.size();
        B.gx_dat.insert(B.gx_dat.end(), w->spr_dat + i * SWF_SPR_DOUBLES, w->spr_dat + (i + 1) * SWF_SPR_DOUBLES);
        B.sc_gf.push_back(add_gf(GF_SPR, 1, data, { bidP(ix[0]), bidC(ix[1]) }));
    }
    for (int i = 0; i < w->n_scp; i++) {
        const int* ix = w->scp_idx + i * 3;
        CHK(ix[0], nP, "spp carrier phase") CHK(ix[1], nC, "spp carrier phase") CHK(ix[2], nC, "spp carrier phase")
        int data = (int)B.gx_dat.size();
        B.gx_dat.insert(B.gx_dat.end(), w->scp_dat + i * SWF_SCP_DOUBLES, w->scp_dat + (i + 1) * SWF_SCP_DOUBLES);
        B.sc_gf.push_back(add_gf(GF_SCP, 1, data, { bidP(ix[0]), bidC(ix[1]), bidC(ix[2]) }));
    }
    for (int i = 0; i < w->n_fix; i++) {
        const int* ix = w->fix_idx + i * 2;
        CHK(ix[0], nC, "fixed integer") CHK(ix[1], nC, "fixed integer")
        if (ix[0] == ix[1]) return fail(SWF_E_INVALID, "fixed integer: both blocks are the same scalar");
        int data = (int)B.gx_dat.size();
        B.gx_dat.insert(B.gx_dat.end(), w->fix_dat + i * SWF_FIX_DOUBLES, w->fix_dat + (i + 1) * SWF_FIX_DOUBLES);
        B.sc_gf.push_back(add_gf(GF_FIX, 1, data, { bidC(ix[0]), bidC(ix[1]) }));
    }
    // inverse-depth projection factors: two residual rows, evaluated one lane each with the scalar factors; record = kind | pts (6)
    for (int i = 0; i < w->n_idp; i++) {
        const int* ix = w->idp_idx + i * 5; const int kd = w->idp_kind[i];
        if (kd < 0 || kd > 2) return fail(SWF_E_INVALID, "inverse-depth projection: kind must be 0, 1 or 2");
        std::vector<int> blks;
        if (kd != 2) { CHK(ix[0], nP, "inverse-depth projection") CHK(ix[1], nP, "inverse-depth projection") blks.push_back(bidP(ix[0])); blks.push_back(bidP(ix[1])); }
        CHK(ix[2], nP, "inverse-depth projection") blks.push_back(bidP(ix[2]));
        if (kd != 0) { CHK(ix[3], nP, "inverse-depth projection") blks.push_back(bidP(ix[3])); }
        CHK(ix[4], nC, "inverse-depth projection") blks.push_back(bidC(ix[4]));
        for (size_t a = 0; a < blks.size(); a++) for (size_t c2 = 0; c2 < a; c2++)
            if (blks[a] == blks[c2]) return fail(SWF_E_INVALID, "inverse-depth projection: repeated parameter block");
        int data = (int)B.gx_dat.size();
        B.gx_dat.push_back((double)kd);
        B.gx_dat.insert(B.gx_dat.end(), w->idp_pts + (size_t)i * 6, w->idp_pts + (size_t)(i + 1) * 6);
        { int g = add_gf(GF_IDP, 2, data, blks); B.sc_gf.push_back(g); B.idp_gf.push_back(g); }
    }
    // world-point projection factors of the generic path (GF_PROJX): record = uv; GFac.pad = the caller's factor index
    for (int i = 0; i < w->n_proj; i++) {
        const int* ix = w->proj_idx + i * 3;
        if (!lm_generic[ix[2]]) continue;
        int data = (int)B.gx_dat.size();
        B.gx_dat.push_back(w->proj_uv[i * 2]); B.gx_dat.push_back(w->proj_uv[i * 2 + 1]);
        int g = add_gf(GF_PROJX, 2, data, { bidP(ix[0]), bidP(ix[1]), bidL(ix[2]) });
        B.gf[g].pad = i;
        B.sc_gf.push_back(g); B.idp_gf.push_back(g);
    }
    std::vector<int> prior_first_gf;
    {
        int bo = 0; long long jo = 0; int ro = 0, x0o = 0;
        for (int k = 0; k < w->n_prior; k++) {
            int nbk = w->prior_nblk[k], dim = w->prior_dim[k];
            std::vector<int> blks(w->prior_blk + bo, w->prior_blk + bo + nbk);
            int dsum = 0, gsum = 0;
            for (int b : blks) { CHK(b, nb, "prior") dsum += ls[b]; gsum += gs[b]; }
            if (dsum != dim) return fail(SWF_E_INVALID, "prior: dim != sum of local block sizes");
            int data = (int)B.prior_dim.size();
            B.prior_dim.push_back(dim);
            B.prior_Joff.push_back((long long)B.prior_J.size());
            B.prior_roff.push_back((int)B.prior_r0.size());
            B.prior_x0off.push_back((int)B.prior_x0.size());
            B.prior_J.insert(B.prior_J.end(), w->prior_J + jo, w->prior_J + jo + (long long)dim * dim);
            B.prior_r0.insert(B.prior_r0.end(), w->prior_r0 + ro, w->prior_r0 + ro + dim);
            B.prior_x0.insert(B.prior_x0.end(), w->prior_x0 + x0o, w->prior_x0 + x0o + gsum);
            int g = add_gf(GF_PRIOR, dim, data, blks);
            B.prior_gf.push_back(g);
            prior_first_gf.push_back(g);
            B.max_prior_dim = std::max(B.max_prior_dim, dim);
            bo += nbk; jo += (long long)dim * dim; ro += dim; x0o += gsum;
        }
    }
    // composite IMU-GNSS factors: carried as prior-type factors whose record k_comp_scatter rewrites at every linearisation
    hw.comp_pose = w->comp_pose; hw.comp_sb = w->comp_sb; hw.comp_e0 = (int)(B.co_pose.size() / 7);
    {
        int io = 0; long long pn = 0, nn = 0; int no = 0, e0 = 0;
        for (int k = 0; k < w->n_comp; k++) {
            const int M = w->comp_M[k], N = w->comp_N[k], G = 30 + N;
            if (M < 1) return fail(SWF_E_INVALID, "composite factor without hidden epochs");
            if (N < 0 || N > CO_MAXN) return fail(SWF_E_UNSUPPORTED, "composite factor with more than 64 ambiguities");
            const int* ix = w->comp_idx + io;
            CHK(ix[0], nP, "composite") CHK(ix[1], nS, "composite") CHK(ix[2], nP, "composite") CHK(ix[3], nS, "composite")
            std::vector<int> blks = { bidP(ix[0]), bidS(ix[1]), bidP(ix[2]), bidS(ix[3]) };
            for (int q = 0; q < N; q++) { CHK(ix[4 + q], nC, "composite") blks.push_back(bidC(ix[4 + q])); }
            for (size_t a = 0; a < blks.size(); a++) {
                if (loc[blks[a]] < 0) return fail(SWF_E_UNSUPPORTED, "composite factor on a constant parameter block");
                for (size_t c2 = 0; c2 < a; c2++) if (blks[c2] == blks[a]) return fail(SWF_E_INVALID, "composite factor: repeated parameter block");
            }
            int data = (int)B.prior_dim.size();
            B.prior_dim.push_back(G);
            B.prior_Joff.push_back((long long)B.prior_J.size()); B.prior_roff.push_back((int)B.prior_r0.size()); B.prior_x0off.push_back((int)B.prior_x0.size());
            B.prior_J.resize(B.prior_J.size() + (size_t)G * G, 0.0); B.prior_r0.resize(B.prior_r0.size() + G, 0.0);
            {   // a valid linearisation point until the first k_comp_scatter: the blocks' current values
                const double* src[4] = { w->pose + 7 * ix[0], w->sb + 9 * ix[1], w->pose + 7 * ix[2], w->sb + 9 * ix[3] };
                const int gsz[4] = { 7, 9, 7, 9 };
                for (int a = 0; a < 4; a++) B.prior_x0.insert(B.prior_x0.end(), src[a], src[a] + gsz[a]);
                for (int q = 0; q < N; q++) B.prior_x0.push_back(w->sc[ix[4 + q]]);
            }
            int g = add_gf(GF_PRIOR, G, data, blks);
            B.prior_gf.push_back(g);
            B.max_prior_dim = std::max(B.max_prior_dim, G);
            B.co_M.push_back(M); B.co_N.push_back(N); B.co_gf.push_back(g); B.co_win.push_back(wi);
            for (int b : blks) B.co_xo.push_back(gx(b));
            B.co_xo_off.push_back((int)B.co_xo.size());
            B.co_pose.insert(B.co_pose.end(), w->comp_pose + (size_t)e0 * 7, w->comp_pose + (size_t)(e0 + M) * 7);
            B.co_sb.insert(B.co_sb.end(), w->comp_sb + (size_t)e0 * 9, w->comp_sb + (size_t)(e0 + M) * 9);
            B.co_pose_lin.insert(B.co_pose_lin.end(), w->comp_pose_lin + (size_t)e0 * 7, w->comp_pose_lin + (size_t)(e0 + M) * 7);
            B.co_sb_lin.insert(B.co_sb_lin.end(), w->comp_sb_lin + (size_t)e0 * 9, w->comp_sb_lin + (size_t)(e0 + M) * 9);
            B.co_Hpp.insert(B.co_Hpp.end(), w->comp_Hpp + (size_t)e0 * 225, w->comp_Hpp + (size_t)(e0 + M) * 225);
            B.co_HpN.insert(B.co_HpN.end(), w->comp_HpN + pn, w->comp_HpN + pn + 15LL * M * N);
            B.co_rhs_p.insert(B.co_rhs_p.end(), w->comp_rhs_p + (size_t)e0 * 15, w->comp_rhs_p + (size_t)(e0 + M) * 15);
            B.co_HNN.insert(B.co_HNN.end(), w->comp_HNN + nn, w->comp_HNN + nn + (long long)N * N);
            B.co_rhsN.insert(B.co_rhsN.end(), w->comp_rhsN + no, w->comp_rhsN + no + N);
            B.co_pre.insert(B.co_pre.end(), w->comp_pre + (size_t)(e0 + k) * SWF_PRE_DOUBLES, w->comp_pre + (size_t)(e0 + k + M + 1) * SWF_PRE_DOUBLES);
            {   // middle-marginalisation link (AddMidMargInfo): optional
                int mid = w->comp_mid ? w->comp_mid[k] : 0;
                if (mid != 0 && (mid < 1 || mid > M - 1 || !w->comp_H12)) return fail(SWF_E_INVALID, "composite factor: comp_mid must be 0 or a link between two hidden epochs (1..M-1), with comp_H12 given");
                B.co_mid.push_back(mid);
                if (mid) B.co_H12.insert(B.co_H12.end(), w->comp_H12 + (size_t)k * 225, w->comp_H12 + (size_t)(k + 1) * 225);
                else B.co_H12.resize(B.co_H12.size() + 225, 0.0);
            }
            for (int q = 0; q < 3; q++) B.co_pbgw.push_back(w->pbg[q]);
            for (int q = 0; q < 3; q++) B.co_pbgw.push_back(w->gw[q]);
            io += 4 + N; pn += 15LL * M * N; nn += (long long)N * N; no += N; e0 += M;
        }
        hw.comp_ne = e0;
    }
#undef CHK
    R.gf1 = (int)B.gf.size();

    // ---- cliques
    R.cl0 = (int)B.cl.size();
    std::map<int, int> e_clique;            // window block id -> clique
    std::map<int, int> free_clique;         // first variable block -> clique (free factors)
    struct TmpC { int e; std::vector<int> facs; std::vector<int> mem; bool is_static; };
    std::vector<TmpC> tc;
    // group-0 non-landmark blocks, in ordering order, always get a clique
    for (int i = 0; i < w->n_order; i++) {
        int b = w->order_block[i];
        if (w->order_group[i] != 0) break;
        if (b >= nP + nS && b < nP + nS + nL && !lm_generic[b - nP - nS]) continue;      // fast-path landmarks: k_lm_schur
        e_clique[b] = (int)tc.size();
        tc.push_back(TmpC{ b, {}, {}, false });
    }
    for (int f = R.gf0; f < R.gf1; f++) {
        const TmpF& t = tf[f - R.gf0];
        int e = -1, first_var = -1;
        for (int b : t.blk) {
            if (loc[b] < 0) continue;
            if (first_var < 0) first_var = b;
            if (is_e(b)) {
                if (b >= nP + nS && b < nP + nS + nL && !lm_generic[b - nP - nS]) return fail(SWF_E_UNSUPPORTED, "non-projection factor on a landmark");
                if (e >= 0 && e != b) return fail(SWF_E_INVALID, "elimination group 0 is not an independent set");
                e = b;
            }
        }
        int c;
        if (e >= 0) c = e_clique[e];
        else if (B.gf[f].type == GF_PRIOR) { c = (int)tc.size(); tc.push_back(TmpC{ -1, {}, {}, true }); }
        else if (first_var < 0) continue;    // all-constant factor: contributes only to the cost
        else {
            auto it = free_clique.find(first_var);
            if (it == free_clique.end()) { c = (int)tc.size(); free_clique[first_var] = c; tc.push_back(TmpC{ -1, {}, {}, false }); }
            else c = it->second;
        }
        tc[c].facs.push_back(f);
        for (int b : t.blk) {
            if (loc[b] < 0 || b == e) continue;
            if (std::find(tc[c].mem.begin(), tc[c].mem.end(), b) == tc[c].mem.end()) tc[c].mem.push_back(b);
        }
    }
    // reduced offsets
    auto red = [&](int b) { return loc[b] - ne; };
    std::map<std::pair<int, int>, std::vector<std::array<long long, 3>>> pmap;   // (a,b) -> (coff, cld, voff)
    for (size_t ci = 0; ci < tc.size(); ci++) {
        TmpC& t = tc[ci];
        Clique C{};
        C.win = wi;
        C.d_e = t.e >= 0 ? ls[t.e] : 0;
        C.e_loc = t.e >= 0 ? gloc(t.e) : -1;
        C.fac0 = (int)B.cl_fac.size();
        int nrows = 0;
        for (int f : t.facs) { B.cl_fac.push_back(f); B.cl_frow.push_back(nrows); nrows += B.gf[f].nres; B.gf[f].clique = (int)B.cl.size(); }
        C.fac1 = (int)B.cl_fac.size();
        C.n_rows = nrows;

        C.mem0 = (int)B.cm_loc.size();
        int df = 0;
        std::map<int, int> colof;
        for (int b : t.mem) {
            B.cm_loc.push_back(gloc(b)); B.cm_ls.push_back(ls[b]); B.cm_col.push_back(df);
            colof[b] = df; df += ls[b];
        }
        C.mem1 = (int)B.cm_loc.size();
        C.d_f = df;
        if (!t.is_static && C.d_e + df > CB_MAXD) return fail(SWF_E_UNSUPPORTED, "clique with more than 768 columns");
        if (!t.is_static && C.d_e > 9) return fail(SWF_E_UNSUPPORTED, "group-0 block larger than 9 dimensions");
        // (k_clique_big keeps the e-rows of M and T = Einv M_ef in LDS: only the cliques that take it — beyond 64 x 64 / 96 x 64 — are bound by that)
        if (!t.is_static && (nrows > CLQ_TALLR || C.d_e + df > 64) && (long long)C.d_e * (C.d_e + df) > CB_MAXED)
            return fail(SWF_E_UNSUPPORTED, "clique beyond one wavefront with d_e (d_e + d_f) > 1536 (swf_solver.h: limits of a group-0 clique)");
        C.C_off = B.C_tot; B.C_tot += (long long)df * df;
        C.v_off = B.v_tot; B.v_tot += df;
        C.e_off = B.e_tot; B.e_tot += C.d_e * C.d_e + C.d_e * df + C.d_e;
        C.is_static = t.is_static ? 1 : 0;
        // slot -> clique column
        for (int f : t.facs) {
            const TmpF& tff = tf[f - R.gf0];
            for (size_t sl = 0; sl < tff.blk.size(); sl++) {
                int b = tff.blk[sl];
                int cc = -1;
                if (loc[b] >= 0) cc = (b == t.e) ? 0 : C.d_e + colof[b];
                B.s_ccol[B.gf[f].slot0 + sl] = cc;
            }
        }
        // storage: a non-static clique owns a dense column-major Jacobian [d][n_rows] in g_J (each factor's blocks sit at
        // their (row, column) position, column stride n_rows) and contiguous residual rows in g_r; factors of static cliques
        // only need residual rows
        C.r_off = B.r_tot; B.r_tot += nrows;
        C.j_off = B.j_tot;
        {
            int dcl = C.d_e + df, frow = 0;
            for (int f : t.facs) {
                GFac& G = B.gf[f];
                G.roff = C.r_off + frow; G.jld = nrows;
                for (int sl = 0; sl < G.nslot; sl++) {
                    int cc = B.s_ccol[G.slot0 + sl];
                    // a prior-type record inside the clique of a group-0 block (a composite factor on an eliminated speed-bias block, as
                    // MyOrdering produces them, R/swf/swf_gnss.cpp:683-691): its rows join the clique's dense Jacobian like any factor's —
                    // the prior evaluation copies the record's columns there at every linearisation
                    if (G.type == GF_PRIOR && !t.is_static && cc >= 0) B.s_joff[G.slot0 + sl] = 0;
                    if (B.s_joff[G.slot0 + sl] < 0) continue;
                    if (t.is_static || cc < 0) { B.s_joff[G.slot0 + sl] = -1; continue; }
                    B.s_joff[G.slot0 + sl] = C.j_off + cc * nrows + frow;
                }
                frow += G.nres;
            }
            if (!t.is_static) B.j_tot += nrows * dcl;
        }
        // static prior clique: C = J^T J over member columns, dgraw = diag
        B.C_init.resize((size_t)B.C_tot, 0.0);
        B.dgraw_init.resize((size_t)B.v_tot, 0.0);
        if (t.is_static) {
            const GFac& G = B.gf[t.facs[0]];
            int dim = G.nres;
            const double* J = B.prior_J.data() + B.prior_Joff[G.data];
            // prior column -> member column (or -1)
            std::vector<int> pcol(dim, -1);
            {
                int col = 0;
                const TmpF& tff = tf[t.facs[0] - R.gf0];
                for (int b : tff.blk) { if (loc[b] >= 0) for (int j = 0; j < ls[b]; j++) pcol[col + j] = colof[b] + j; col += ls[b]; }
            }
            double* Cm = B.C_init.data() + C.C_off;
            for (int a = 0; a < dim; a++) {
                if (pcol[a] < 0) continue;
                for (int b2 = 0; b2 < dim; b2++) {
                    if (pcol[b2] < 0) continue;
                    double sacc = 0;
                    for (int r = 0; r < dim; r++) sacc += J[(size_t)r * dim + a] * J[(size_t)r * dim + b2];
                    Cm[(size_t)pcol[a] * df + pcol[b2]] = sacc;
                }
                B.dgraw_init[C.v_off + pcol[a]] = Cm[(size_t)pcol[a] * df + pcol[a]];
            }
        }
        // pair contributions
        for (int a : t.mem) for (int b2 : t.mem) {
            if (red(a) < red(b2)) continue;
            pmap[{ a, b2 }].push_back({ C.C_off + (long long)colof[a] * df + colof[b2], df, C.v_off + colof[a] });
        }
        B.cl.push_back(C);
    }
    R.cl1 = (int)B.cl.size();
    // factors outside every clique (all blocks constant) still own residual rows (cost only)
    for (int f = R.gf0; f < R.gf1; f++) if (B.gf[f].roff < 0) { B.gf[f].roff = B.r_tot; B.r_tot += B.gf[f].nres; }

    // ---- pairs: clique pairs, all frame pairs, a diagonal pair for every reduced block
    for (int p = 0; p < nP; p++) if (frame_of[p] >= 0)
        for (int q = 0; q < nP; q++) if (frame_of[q] >= 0 && red(bidP(p)) >= red(bidP(q))) pmap[{ bidP(p), bidP(q) }];
    for (int i = 0; i < w->n_order; i++) { int b = w->order_block[i]; if (!is_e(b)) pmap[{ b, b }]; }
    R.pair0 = (int)B.pair.size();
    for (auto& kv : pmap) {
        int a = kv.first.first, b2 = kv.first.second;
        Pair P{};
        P.win = wi; P.ra = red(a); P.rb = red(b2); P.la = ls[a]; P.lb = ls[b2];
        P.fa = a < nP ? frame_of[a] : -1; P.fb = b2 < nP ? frame_of[b2] : -1;
        P.c0 = (int)B.pc_coff.size();
        for (auto& c : kv.second) { B.pc_coff.push_back(c[0]); B.pc_cld.push_back((int)c[1]); B.pc_voff.push_back((int)c[2]); }
        P.c1 = (int)B.pc_coff.size();
        P.is_diag = (a == b2) ? 1 : 0;
        P.loc_a = gloc(a);
        B.pair.push_back(P);
    }
    R.pair1 = (int)B.pair.size();

    R.proj_sqrt_info = w->proj_sqrt_info; R.proj_loss_a = w->proj_loss_a;
    for (int k = 0; k < 3; k++) { R.pbg[k] = w->pbg[k]; R.gw[k] = w->gw[k]; R.base[k] = w->base[k]; }
    B.n_x += R.x_n; B.n_loc += R.n_loc;
    // algorithmic Jacobian bytes of one evaluation (SURVEY.md §8d formula)
    {
        int64_t pb = 0;
        for (int k = 0; k < w->n_prior; k++) { int64_t n = w->prior_dim[k]; pb += 8 * (n * n + 4 * n); }
        B.jac_bytes += (int64_t)312 * w->n_proj + (int64_t)5480 * w->n_imu + (int64_t)176 * w->n_cp + (int64_t)152 * w->n_pr + (int64_t)208 * w->n_dop
                     + (int64_t)136 * w->n_spr + (int64_t)160 * w->n_scp + (int64_t)56 * w->n_fix + (int64_t)584 * w->n_idp + pb;
    }
    B.win.push_back(R);
    return SWF_OK;
}
}  // namespace

// ------------------------------------------------------------------ batch API
extern "C" int swf_batch_create(const swf_flat_window* const* windows, int32_t n, void* stream, swf_batch** out) {
    if (!windows || n <= 0 || !out) return fail(SWF_E_INVALID, "swf_batch_create: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(SWF_E_NODEVICE, "no HIP device: this library has no CPU fallback");
    Build B;
    std::vector<HostWin> hw(n);
    for (int i = 0; i < n; i++) {
        int rc = build_window(B, windows[i], i, hw[i]);
        if (rc != SWF_OK) return rc;
    }
    if (B.n_x > 0x7fffffffLL || B.n_loc > 0x7fffffffLL) return fail(SWF_E_UNSUPPORTED, "batch too large for 32-bit offsets");
    swf_batch* b = new swf_batch();
    b->device = current_device();
    b->stream = (hipStream_t)stream;
    { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) b->n_cu = pr.multiProcessorCount; }
    b->D.rr_nmax = b->rr_nmax;
    // (the launch-shape knobs that remain are test aids, each exercised by the GPU tier, and are read once, here: no getenv on the latency
    // path or inside the marginalisation's sweep loop, and none that could race a setenv from the per-device enqueue threads of
    // swf_solve_batches.  Nothing that changes RESULTS is an environment variable: those are fields of swf_options.)
    b->no_comp_fuse = getenv("SWF_NO_COMP_FUSE") != nullptr;
    b->no_spec = getenv("SWF_NO_SPEC_EVAL") != nullptr;
    b->no_step_fuse = getenv("SWF_NO_STEP_FUSE") != nullptr;
    b->no_decide_fuse = getenv("SWF_NO_DECIDE_FUSE") != nullptr;
    // auxiliary stream: the latency path (<= n_CU / 16 windows), and batches of half a chip to a chip of windows, where the IMU / clique branch
    // fills what one-block-per-window kernels leave idle (measured: 256 windows 5.52 -> 5.22 ms, 128 windows 3.73 -> 3.49 ms per solve; 64 and
    // 512 windows: no gain)
    // latency path (up to n_CU / 8 windows): independent kernels of an iteration ride in ONE grid (the IMU factors with the projection /
    // scalar factors, every clique size class in one launch) on ONE stream.  Round 3 ran the IMU / clique branch of such batches on the
    // auxiliary stream instead; the kernel trace shows what that buys: every cross-queue edge (event record -> stream wait) costs 6-13 us
    // of dependency resolution, as much as the overlap saves (one window: 1.432 ms with the auxiliary stream, 1.443 without).
    b->lat_fuse = n * 8 <= b->n_cu && !getenv("SWF_NO_LAT_FUSE");
    if ((n * 16 <= b->n_cu && !b->lat_fuse) || (2 * n >= b->n_cu && n <= b->n_cu)) {            // fork / join inside a linearisation
        bool ok = (b->aux = handle_cache().stream()) != nullptr;
        for (int i = 0; i < 3 && ok; i++) ok = (b->ev_fork[i] = handle_cache().event(false)) != nullptr;
        if (!ok) { handle_cache().give(b->aux); b->aux = nullptr; }
    }
    b->win = B.win; b->hw = hw; b->max_tiles = B.max_tiles; b->max_prior_dim = B.max_prior_dim; b->jac_bytes = B.jac_bytes;
    b->proj_bytes = (int64_t)312 * (int64_t)B.p_win.size();
    for (size_t l = 0; l + 1 < B.lm_obs0.size() + 1 && l < B.lm_win.size(); l++) {
        int64_t k = (l + 1 < B.lm_obs0.size() ? B.lm_obs0[l + 1] : (int)B.p_win.size()) - B.lm_obs0[l];
        b->lm_schur_flops += 216 * k * k + 108 * k;
        b->lm_schur_flops_sym += 108 * k * (k - 1) + 162 * k;
    }
    for (auto& W : B.win) { b->chol_flops += (int64_t)W.n_red * W.n_red * W.n_red / 3; b->max_red = std::max(b->max_red, W.n_red); b->min_red = std::min(b->min_red, W.n_red); if (W.n_red > 224 && W.n_red <= 240) b->rr4_has15 = true; if (W.n_red > 240 && W.n_red <= 256) b->rr4_has16 = true; }
    DevBatch& D = b->D;
    DevPool& P = b->pool;
    int rc = 0;
    D.n_win = n; D.n_x = (int)B.n_x; D.n_loc_total = (int)B.n_loc; D.max_iter_trace = SWF_MAX_TRACE;
    std::vector<int> lmb_lr;
    {
        // landmark back-substitution blocks: consecutive landmarks of one window with at most 256 observations together
        // (built before the window records are uploaded: a window knows its block range, WinRec::lmb0 / lmb1)
        std::vector<int>& lr = lmb_lr;
        std::vector<int> obs0 = B.lm_obs0; obs0.push_back((int)B.p_win.size());
        for (WinRec& Wr : B.win) {
            Wr.lmb0 = (int)(lr.size() / 4);
            int l = Wr.lm0;
            while (l < Wr.lm1) {
                const int o0 = obs0[(size_t)l]; int l1 = l, cnt = 0;
                while (l1 < Wr.lm1 && l1 - l < 256 && cnt + (obs0[(size_t)l1 + 1] - obs0[(size_t)l1]) <= 256) { cnt += obs0[(size_t)l1 + 1] - obs0[(size_t)l1]; l1++; }
                if (l1 == l) { P.release(); delete b; return fail(SWF_E_UNSUPPORTED, "landmark with more than 256 observations"); }
                lr.push_back(o0); lr.push_back(cnt); lr.push_back(l); lr.push_back(l1 - l);
                l = l1;
            }
            Wr.lmb1 = (int)(lr.size() / 4);
        }
        D.n_lmb = (int)(lr.size() / 4);
        if (lr.empty()) lr.resize(4, 0);
    }
    std::vector<int> pch_q, pch_r0, prior_nch;
    {
        // row chunks of the priors (swf_dev.h): a prior beyond PRIOR_SPLIT_DIM rows is evaluated by one workgroup per PRIOR_CHUNK rows;
        // a window's priors (the composite factors' records among them) are contiguous in prior_gf, and so are their chunks
        for (WinRec& Wr : B.win) { Wr.pch0 = 0; Wr.pch1 = 0; }
        int cur_w = -1;
        for (size_t q = 0; q < B.prior_gf.size(); q++) {
            const GFac& G = B.gf[(size_t)B.prior_gf[q]];
            const int n = G.nres, nch = n > PRIOR_SPLIT_DIM ? (n + PRIOR_CHUNK - 1) / PRIOR_CHUNK : 1;
            if (G.win != cur_w) { cur_w = G.win; B.win[(size_t)cur_w].pch0 = (int)pch_q.size(); }
            prior_nch.push_back(nch);
            for (int c = 0; c < nch; c++) { pch_q.push_back((int)q); pch_r0.push_back(nch > 1 ? c * PRIOR_CHUNK : 0); }
            B.win[(size_t)cur_w].pch1 = (int)pch_q.size();
            if (nch > 1 && G.clique >= 0 && B.cl[(size_t)G.clique].is_static) b->n_pch_split += nch;
        }
        D.n_pch = (int)pch_q.size();
        if (pch_q.empty()) { pch_q.push_back(0); pch_r0.push_back(0); }
        if (prior_nch.empty()) prior_nch.push_back(1);
        b->win = B.win;                                   // (the host's copy of the records, with the block ranges)
    }
#define PUT(field, vec) rc |= P.put(vec, &D.field)
    PUT(win, B.win);
    PUT(lmb_rec, lmb_lr); PUT(pch_q, pch_q); PUT(pch_r0, pch_r0); PUT(prior_nch, prior_nch);
    PUT(blk_xoff, B.blk_xoff); PUT(blk_loc, B.blk_loc); PUT(blk_gs, B.blk_gs);
    if (B.loc2x.empty()) B.loc2x.push_back(-1);
    PUT(loc2x, B.loc2x); PUT(x_var, B.x_var);
    D.n_proj = (int)B.p_win.size();
    PUT(p_win, B.p_win); PUT(p_xpose, B.p_xpose); PUT(p_xex, B.p_xex); PUT(p_xlm, B.p_xlm);
    PUT(p_lpose, B.p_lpose); PUT(p_llm, B.p_llm); PUT(p_fr, B.p_fr); PUT(p_lm, B.p_lm); PUT(p_uv, B.p_uv);
    D.n_lm = (int)B.lm_win.size();
    B.lm_obs0.push_back(D.n_proj);
    PUT(lm_win, B.lm_win); PUT(lm_obs0, B.lm_obs0); PUT(lm_loc, B.lm_loc); PUT(lm_col, B.lm_col); PUT(lm_fmask, B.lm_fmask);
    {
        // k_lm_schur's launch shape: the row class of the panel by the batch's largest window (<= 10 / 21 / 42 / 64 observing frames) and
        // the landmark parts per block — as many as still leave >= 2 blocks per CU.  SWF_LS_VARIANT / SWF_LS_QPB: test / debugging aids.
        // A block that covers all parts folds them in registers (ls_folded) and, in that case, writes -P straight into S (s_direct); the
        // off-diagonal frame pairs without any other contribution then leave the assembly's list.
        const int force = getenv("SWF_LS_VARIANT") ? atoi(getenv("SWF_LS_VARIANT")) : 0;
        const int force_qpb = getenv("SWF_LS_QPB") ? atoi(getenv("SWF_LS_QPB")) : 0;
        int qpb = 1;
        while (qpb < GEMM_SPLIT && (long long)n * GEMM_SPLIT / (2 * qpb) >= 2LL * b->n_cu) qpb *= 2;
        // from half a chip of windows on, one block per window: the folded product and the direct-to-S write-out are worth more than the
        // second round of blocks
        if (2LL * n >= b->n_cu) qpb = GEMM_SPLIT;
        if (force_qpb >= 1 && force_qpb <= GEMM_SPLIT && (force_qpb & (force_qpb - 1)) == 0) qpb = force_qpb;
        b->ls_qpb = qpb;
        b->ls_var = (b->max_tiles <= 10 && force < 1) ? 0 : (b->max_tiles <= 36 && force < 2) ? 1 : (b->max_tiles <= 136 && force < 3) ? 2 : 3;
        b->ls_folded = qpb == GEMM_SPLIT && b->ls_var <= 1;          // must mirror CAN_FOLD in k_lm_schur
        b->s_direct = b->ls_folded;
    }
    {
        // k_lm_schur task table.  A wave task = the four 16-lane groups of one producer wave = four landmarks, one group and three of the
        // task's twelve panel columns each (a track of more than 16 observations takes further rounds of its group's lanes).  Record of
        // (task, group): L (-1 = empty), loc, first / end observation of the landmark, first column within the task (0, 3, 6, 9).  The window's landmarks
        // enter in the order of their tile footprint (last, first 16-row tile of the reduced camera matrix they touch), so the landmarks of a
        // task mostly share theirs; bit g of the tile mask of (chunk, tile) — some landmark of the chunk's wave task g is seen from the tile's
        // row frames and from its column frames, per k-step of the task since round 4 (three bits per task: a tile skips the k-steps none of
        // whose landmarks touch it) — is what the consumer waves walk (a chunk = TW tasks, by size class; the packing into tasks
        // and the parts, which end on even task numbers, are the same in every class: so is every sum).  Tile list of a window: the nt
        // diagonal tiles, then (tr > tc) row by row.
        const int TW = b->ls_var <= 1 ? 2 : 1;
        const int NCW = b->ls_var <= 1 ? 8 : 12, TPW = b->ls_var == 0 ? 2 : b->ls_var == 1 ? 5 : 6;
        const int n_launch = std::max(1, (b->max_tiles + NCW * TPW - 1) / (NCW * TPW));
        b->ls_kms = n_launch * NCW;                                  // mask words per chunk
        std::vector<int> c0, rec, km;
        std::vector<std::vector<unsigned>> task_tiles;               // per task of the current window: tile-list entries it touches
        for (auto& W : B.win) {
            const int m = 6 * W.nF, nt = (m + 15) / 16, ntl = nt * (nt + 1) / 2;
            auto tile_rows = [&](unsigned long long fm) {          // 16-row tiles the frames of fm touch
                unsigned t = 0;
                for (int f = 0; f < W.nF && f < 64; f++) if ((fm >> f) & 1ULL) { t |= 1u << ((6 * f) / 16); t |= 1u << ((6 * f + 5) / 16); }
                return t;
            };
            std::vector<int> ord; std::vector<unsigned> trs((size_t)(W.lm1 - W.lm0), 0u);
            for (int l = W.lm0; l < W.lm1; l++) {
                if (B.lm_loc[l] < 0) continue;                     // constant landmark: nothing to eliminate
                int k = B.lm_obs0[l + 1] - B.lm_obs0[l];
                if (k > 64) { P.release(); delete b; return fail(SWF_E_UNSUPPORTED, "landmark with more than 64 observations"); }
                trs[(size_t)(l - W.lm0)] = tile_rows(B.lm_fmask[l]);
                ord.push_back(l);
            }
            auto key = [&](int l) { unsigned t = trs[(size_t)(l - W.lm0)]; int lo = t ? __builtin_ctz(t) : 0, hi = t ? 31 - __builtin_clz(t) : 0; return hi * 64 + lo; };
            std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return key(x) < key(y); });
            const int first_task = (int)(rec.size() / 32);
            task_tiles.clear();
            size_t at = 0; int lw = 4;                             // force a new task at the first landmark
            auto new_task = [&]() { at = rec.size(); rec.resize(at + 32, 0); for (int g = 0; g < 4; g++) rec[at + g * 8] = -1; task_tiles.emplace_back((size_t)ntl, 0u); lw = 0; };
            for (int l : ord) {
                int o0 = B.lm_obs0[l], k = B.lm_obs0[l + 1] - o0;
                if (lw >= 4) new_task();
                int* r = &rec[at + (size_t)lw * 8];                 // group lw of the task = this landmark, whatever its track length
                r[0] = l; r[1] = B.lm_loc[l]; r[2] = o0; r[3] = o0 + k; r[4] = 3 * lw; r[5] = 0; r[6] = 0; r[7] = 0;
                // tile-list entries this landmark touches, as the k-steps of the task that carry its three columns (columns 3 lw .. 3 lw + 2 of
                // the task's twelve; a k-step = four columns): group 0 -> k-step 0, group 1 -> 0 and 1, group 2 -> 1 and 2, group 3 -> 2
                unsigned t = trs[(size_t)(l - W.lm0)];
                const unsigned ks = lw == 0 ? 1u : lw == 1 ? 3u : lw == 2 ? 6u : 4u;
                std::vector<unsigned>& tt = task_tiles.back();
                for (int tr = 0; tr < nt; tr++) {
                    if (!((t >> tr) & 1u)) continue;
                    tt[(size_t)tr] |= ks;
                    for (int tc = 0; tc < tr; tc++) if ((t >> tc) & 1u) tt[(size_t)(nt + tr * (tr - 1) / 2 + tc)] |= ks;
                }
                lw++;
            }
            if (task_tiles.size() & 1) new_task();                 // an even number of tasks per window
            const int ntask = (int)task_tiles.size();
            // masks: chunk c of this window = tasks [c TW, (c + 1) TW)
            for (int c = 0; c < ntask / TW; c++) {
                size_t kw = km.size(); km.resize(kw + (size_t)b->ls_kms, 0);
                for (int g = 0; g < TW; g++) {
                    const std::vector<unsigned>& tt = task_tiles[(size_t)(c * TW + g)];
                    for (int e = 0; e < ntl; e++) if (tt[(size_t)e]) {
                        b->lm_schur_mfma += __builtin_popcount(tt[(size_t)e]);
                        int lp = e / (NCW * TPW), r = e % (NCW * TPW), sl = r / NCW, cw = r % NCW;
                        km[kw + (size_t)(lp * NCW + cw)] |= (tt[(size_t)e] << (3 * g)) << (3 * TW * sl);
                    }
                }
            }
            // part sp = task pairs [sp np / 16, (sp + 1) np / 16): the parts differ by at most one pair, so any grouping of consecutive parts
            // into workgroups (qpb = 1 .. 16, by batch size) is balanced
            const int np = ntask / 2;
            for (int sp = 0; sp < GEMM_SPLIT; sp++) c0.push_back(first_task + 2 * (int)((long long)sp * np / GEMM_SPLIT));
        }
        c0.push_back((int)(rec.size() / 32));
        rec.resize(rec.size() + (size_t)32 * 4 * LS_NB, 0);           // slack behind the table
        km.resize(km.size() + (size_t)b->ls_kms * 2, 0);
        PUT(sch_c0, c0); PUT(sch_rec, rec); PUT(sch_km, km);
    }
    D.n_fr = B.n_fr;
    B.fr_obs0.push_back((int)B.fr_obs.size());
    PUT(fr_obs0, B.fr_obs0); PUT(fr_obs, B.fr_obs); PUT(fr_red, B.fr_red);
    D.n_fsb = (int)B.fsb_win.size();
    B.fsb_obs0.push_back(D.n_proj); B.fsb_perm.resize((size_t)D.n_proj + 1, 0);
    PUT(fsb_win, B.fsb_win); PUT(fsb_obs0, B.fsb_obs0); PUT(fsb_perm, B.fsb_perm); PUT(fsb_foff, B.fsb_foff); PUT(fsb_foff0, B.fsb_foff0); PUT(fsb_out0, B.fsb_out0);
    rc |= P.zeros((size_t)B.fs_tot * FS_VAL, &D.fs_part);
    rc |= P.zeros((size_t)std::max<long long>(B.n_loc, 1), &D.jsc);
    D.n_gf = (int)B.gf.size();
    PUT(gf, B.gf);
    PUT(s_x, B.s_x); PUT(s_loc, B.s_loc); PUT(s_ls, B.s_ls); PUT(s_joff, B.s_joff); PUT(s_ccol, B.s_ccol);
    PUT(imu_pre, B.imu_pre); PUT(cp_dat, B.cp_dat); PUT(pr_dat, B.pr_dat); PUT(dop_dat, B.dop_dat); PUT(sp_w, B.sp_w); PUT(gx_dat, B.gx_dat);
    D.n_imu = (int)B.imu_gf.size(); D.n_sc = (int)B.sc_gf.size(); D.n_prior = (int)B.prior_gf.size();
    PUT(imu_gf, B.imu_gf); PUT(sc_gf, B.sc_gf); PUT(prior_gf, B.prior_gf); D.n_idp = (int)B.idp_gf.size(); PUT(idp_gf, B.idp_gf);
    PUT(prior_dim, B.prior_dim); PUT(prior_Joff, B.prior_Joff); PUT(prior_roff, B.prior_roff); PUT(prior_x0off, B.prior_x0off);
    {   // transposed copies of the prior records for the J v products
        std::vector<double> Jt(B.prior_J.size());
        for (size_t k = 0; k < B.prior_dim.size(); k++) {
            const size_t n = (size_t)B.prior_dim[k]; const double* J = B.prior_J.data() + B.prior_Joff[k]; double* T = Jt.data() + B.prior_Joff[k];
            for (size_t r = 0; r < n; r++) for (size_t c = 0; c < n; c++) T[c * n + r] = J[r * n + c];
        }
        PUT(prior_Jt, Jt);
        // column -> local index of every prior record (the J v products walk the columns flat, eight at a time)
        std::vector<int> cl(B.prior_r0.size(), -1), cc(B.prior_r0.size(), -1), pcol(B.s_ls.size(), 0), pxo(B.s_ls.size(), 0);
        for (const GFac& G : B.gf) {
            if (G.type != GF_PRIOR) continue;
            int col = 0, xo = 0;
            for (int t = 0; t < G.nslot; t++) {
                int l = B.s_ls[G.slot0 + t], lo = B.s_loc[G.slot0 + t], mc = B.s_ccol[G.slot0 + t];
                pcol[G.slot0 + t] = col; pxo[G.slot0 + t] = xo;
                for (int q = 0; q < l; q++) {
                    cl[(size_t)B.prior_roff[G.data] + col + q] = lo >= 0 ? lo + q : -1;
                    cc[(size_t)B.prior_roff[G.data] + col + q] = mc >= 0 ? mc + q : -1;
                }
                col += l; xo += (l == 6 ? 7 : l);
            }
        }
        PUT(prior_colloc, cl); PUT(prior_colcc, cc); PUT(s_pcol, pcol); PUT(s_pxo, pxo);
    }
    PUT(prior_J, B.prior_J); PUT(prior_r0, B.prior_r0); PUT(prior_x0, B.prior_x0);
    D.n_cl = (int)B.cl.size();
    PUT(cl, B.cl); PUT(cl_fac, B.cl_fac); PUT(cl_frow, B.cl_frow); PUT(cm_loc, B.cm_loc); PUT(cm_ls, B.cm_ls); PUT(cm_col, B.cm_col);
    {
        std::vector<int> cvl((size_t)std::max(B.v_tot, 1), 0);
        for (const Clique& c : B.cl)
            for (int m = c.mem0; m < c.mem1; m++)
                for (int q = 0; q < B.cm_ls[(size_t)m]; q++) cvl[(size_t)c.v_off + (size_t)B.cm_col[(size_t)m] + (size_t)q] = B.cm_loc[(size_t)m] + q;
        PUT(cv_loc, cvl);
    }
    D.n_pair = (int)B.pair.size();
    PUT(pc_coff, B.pc_coff); PUT(pc_cld, B.pc_cld); PUT(pc_voff, B.pc_voff);
    {
        std::vector<int> pd, po, clc[5], cle;
        for (size_t i = 0; i < B.pair.size(); i++) {
            Pair& Pq = B.pair[i]; const WinRec& Rw = B.win[Pq.win];      // self-contained records (see Pair)
            Pq.fsb0 = Rw.fsb0; Pq.fsb1 = Rw.fsb1; Pq.n = Rw.n_red; Pq.m = 6 * Rw.nF; Pq.S_base = Rw.S_base; Pq.P_base = Rw.P_base; Pq.q_base = (long long)6 * Rw.fr_base * GEMM_SPLIT;
            if (b->s_direct && !Pq.is_diag && Pq.fa >= 0 && Pq.fb >= 0 && Pq.c0 == Pq.c1) continue;       // -P is already in S, nothing to add
            (Pq.is_diag ? pd : po).push_back((int)i);
        }
        // off-diagonal pairs by descending entry rounds (16 entries per round): the round-2 pair-walking assembly's waves (four pairs each) become
        // homogeneous and skip the rounds none of their pairs has; every pair is still written once, by the same arithmetic
        std::stable_sort(po.begin(), po.end(), [&](int a, int c) {
            return (B.pair[a].la * B.pair[a].lb + 15) / 16 > (B.pair[c].la * B.pair[c].lb + 15) / 16; });
        for (size_t i = 0; i < B.cl.size(); i++) {
            const Clique& c = B.cl[i];
            if (c.d_e > 0) cle.push_back((int)i);
            if (c.is_static) continue;
            int d = c.d_e + c.d_f;
            // one wavefront per clique up to 64 x 64 (three size classes); anything larger takes the workgroup kernel (class 3)
            int cls = (c.d_e <= 1 && d <= 32 && c.n_rows <= 48) ? 0 : (c.n_rows <= 32 && d <= 48) ? 1 : (c.n_rows <= CLQ_MAXR && d <= CLQ_MAXD) ? 2 : (c.n_rows <= CLQ_TALLR && d <= 64) ? 4 : 3;
            // latency path: every one-wavefront clique in ONE launch (the 64 x 64 instantiation; the classes differ in loop bounds and zero
            // padding only, the sums and their order are the same: bit-identical results)
            if (b->lat_fuse && cls < 2) cls = 2;
            clc[cls].push_back((int)i);
            for (int q = c.fac0; q < c.fac1; q++) if (B.gf[B.cl_fac[q]].type == GF_IMU) b->clc_imu[cls] = true;
        }
        D.n_pd = (int)pd.size(); D.n_po = (int)po.size(); D.n_cle = (int)cle.size();
        {
            // ---- assembly programs (k_assemble_flat): every pair of the two lists above, flattened into per-entry source lists with
            // window-relative offsets; windows whose programs come out identical share one copy.  Entry order inside a window: the
            // window's pairs in pair order, (i, j) row-major — any order would do, every entry is written by exactly one thread.
            const int n_part = b->s_direct ? 0 : b->ls_folded ? 1 : GEMM_SPLIT, n_qpart = b->ls_folded ? 1 : GEMM_SPLIT;
            struct Prog { std::vector<int> dst, src0, aux, src, vloc, vred, vsrc0, vi, vsrc; std::vector<unsigned> cnt, vcnt; };
            std::vector<AsmWin> asw((size_t)n);
            std::vector<int> t_dst, t_src0, t_aux, t_src, tv_loc, tv_red, tv_src0, tv_i, tv_src; std::vector<unsigned> t_cnt, tv_cnt;
            std::map<std::vector<int>, std::array<int, 4>> seen;           // serialised program -> (se0, ne, ve0, nv)
            std::vector<std::vector<int>> wpairs((size_t)n);
            for (int i : pd) wpairs[(size_t)B.pair[i].win].push_back(i);
            for (int i : po) wpairs[(size_t)B.pair[i].win].push_back(i);
            bool overflow = false;
            // the 16 x 16 tiles of S some block pair reaches, strictly below the tile diagonal (the diagonal tiles are always loaded): from
            // EVERY pair of the window — the frame pairs whose only contribution is the -P that k_lm_schur writes straight into S have
            // left the assembly's lists above, but their tiles are not zero
            std::vector<unsigned> tnz((size_t)n * 4, 0u);
            for (const Pair& Pq : B.pair) {
                if (B.win[(size_t)Pq.win].n_red > 256) continue;
                for (int I = Pq.ra / 16; I <= (Pq.ra + Pq.la - 1) / 16; I++)
                    for (int J = Pq.rb / 16; J <= (Pq.rb + Pq.lb - 1) / 16 && J < I; J++) { const int t = I * (I - 1) / 2 + J; tnz[(size_t)Pq.win * 4 + (t >> 5)] |= 1u << (t & 31); }
            }
            for (int w = 0; w < n; w++) {
                const WinRec& Rw = B.win[w];
                AsmWin& A = asw[(size_t)w];
                A.win = w; A.n_red = Rw.n_red; A.m = 6 * Rw.nF; A.loc_base = Rw.loc_base; A.S_base = Rw.S_base;
                A.P_base = Rw.P_base * GEMM_SPLIT; A.q_base = (long long)6 * Rw.fr_base * GEMM_SPLIT;
                A.fs_base = Rw.fsb1 > Rw.fsb0 ? B.fsb_out0[(size_t)Rw.fsb0] : 0;
                long long Cb = -1; int vb = -1;
                for (int c = Rw.cl0; c < Rw.cl1; c++) { if (Cb < 0 || B.cl[c].C_off < Cb) Cb = B.cl[c].C_off; if (vb < 0 || B.cl[c].v_off < vb) vb = B.cl[c].v_off; }
                A.C_base = Cb < 0 ? 0 : Cb; A.v_base = vb < 0 ? 0 : vb;
                Prog Pg;
                const int nfsb = Rw.fsb1 - Rw.fsb0, mm = A.m;
                for (int pi_ : wpairs[(size_t)w]) {
                    const Pair& Pq = B.pair[(size_t)pi_];
                    const bool frame_pair = Pq.fa >= 0 && Pq.fb >= 0, obs = Pq.is_diag && Pq.fa >= 0;
                    const int ncon = Pq.c1 - Pq.c0;
                    for (int i = 0; i < Pq.la; i++) for (int j = 0; j < Pq.lb; j++) {
                        if (Pq.is_diag && j > i) continue;                  // lower half only; the mirror is the host's job at export
                        const bool dg = Pq.is_diag && i == j;
                        const int nP = frame_pair ? n_part : 0, nH = obs ? nfsb : 0;
                        if (ncon > 4095 || nH > 4095) overflow = true;
                        Pg.dst.push_back((Pq.ra + i) * Pq.n + Pq.rb + j);
                        Pg.cnt.push_back((unsigned)ncon | ((unsigned)nP << 12) | ((unsigned)nH << 17) | ((dg ? 1u : 0u) << 29) | ((frame_pair && b->s_direct ? 1u : 0u) << 30));
                        Pg.src0.push_back((int)Pg.src.size());
                        Pg.aux.push_back(dg ? (Pq.loc_a - Rw.loc_base) + i : 0);
                        for (int c = Pq.c0; c < Pq.c1; c++) Pg.src.push_back((int)(B.pc_coff[(size_t)c] + (long long)i * B.pc_cld[(size_t)c] + j - A.C_base));
                        if (nP) {
                            const int pr = 6 * Pq.fa + i, pc = 6 * Pq.fb + j;
                            const long long pel = pr >= pc ? (long long)pr * mm + pc : (long long)pc * mm + pr;
                            for (int q = 0; q < nP; q++) Pg.src.push_back((int)((long long)q * mm * mm + pel));
                        }
                        if (nH) {
                            const int hi = i > j ? i : j, lo = i > j ? j : i;
                            for (int k2 = Rw.fsb0; k2 < Rw.fsb1; k2++) Pg.src.push_back((B.fsb_out0[(size_t)k2] - A.fs_base + Pq.fa) * FS_VAL + hi * (hi + 1) / 2 + lo);
                        }
                        if (dg) for (int c = Pq.c0; c < Pq.c1; c++) Pg.src.push_back(B.pc_voff[(size_t)c] + i - A.v_base);
                    }
                    if (!Pq.is_diag) continue;
                    for (int i = 0; i < Pq.la; i++) {
                        const int nH = obs ? nfsb : 0, nQ = obs ? n_qpart : 0;
                        Pg.vloc.push_back(Pq.loc_a - Rw.loc_base + i); Pg.vred.push_back(Pq.ra + i); Pg.vi.push_back(obs ? i : 0);
                        Pg.vcnt.push_back((unsigned)ncon | ((unsigned)nQ << 12) | ((unsigned)nH << 17));
                        Pg.vsrc0.push_back((int)Pg.vsrc.size());
                        if (nH) for (int k2 = Rw.fsb0; k2 < Rw.fsb1; k2++) Pg.vsrc.push_back((B.fsb_out0[(size_t)k2] - A.fs_base + Pq.fa) * FS_VAL);
                        for (int c = Pq.c0; c < Pq.c1; c++) Pg.vsrc.push_back(B.pc_voff[(size_t)c] + i - A.v_base);
                        for (int q = 0; q < nQ; q++) Pg.vsrc.push_back(q * mm + 6 * Pq.fa + i);
                    }
                }
                // serialise and look up
                std::vector<int> key;
                key.reserve(Pg.dst.size() * 4 + Pg.src.size() + Pg.vloc.size() * 5 + Pg.vsrc.size() + 8);
                key.push_back((int)Pg.dst.size()); key.push_back((int)Pg.vloc.size());
                key.insert(key.end(), Pg.dst.begin(), Pg.dst.end()); for (unsigned c : Pg.cnt) key.push_back((int)c);
                key.insert(key.end(), Pg.src0.begin(), Pg.src0.end()); key.insert(key.end(), Pg.aux.begin(), Pg.aux.end()); key.insert(key.end(), Pg.src.begin(), Pg.src.end());
                key.insert(key.end(), Pg.vloc.begin(), Pg.vloc.end()); key.insert(key.end(), Pg.vred.begin(), Pg.vred.end()); for (unsigned c : Pg.vcnt) key.push_back((int)c);
                key.insert(key.end(), Pg.vsrc0.begin(), Pg.vsrc0.end()); key.insert(key.end(), Pg.vi.begin(), Pg.vi.end()); key.insert(key.end(), Pg.vsrc.begin(), Pg.vsrc.end());
                auto it = seen.find(key);
                if (it == seen.end()) {
                    const int se0 = (int)t_dst.size(), ve0 = (int)tv_loc.size(), so = (int)t_src.size(), vo = (int)tv_src.size();
                    t_dst.insert(t_dst.end(), Pg.dst.begin(), Pg.dst.end()); t_cnt.insert(t_cnt.end(), Pg.cnt.begin(), Pg.cnt.end()); t_aux.insert(t_aux.end(), Pg.aux.begin(), Pg.aux.end());
                    for (int x : Pg.src0) t_src0.push_back(x + so);
                    t_src.insert(t_src.end(), Pg.src.begin(), Pg.src.end());
                    tv_loc.insert(tv_loc.end(), Pg.vloc.begin(), Pg.vloc.end()); tv_red.insert(tv_red.end(), Pg.vred.begin(), Pg.vred.end()); tv_cnt.insert(tv_cnt.end(), Pg.vcnt.begin(), Pg.vcnt.end());
                    tv_i.insert(tv_i.end(), Pg.vi.begin(), Pg.vi.end());
                    for (int x : Pg.vsrc0) tv_src0.push_back(x + vo);
                    tv_src.insert(tv_src.end(), Pg.vsrc.begin(), Pg.vsrc.end());
                    it = seen.emplace(std::move(key), std::array<int, 4>{ se0, (int)Pg.dst.size(), ve0, (int)Pg.vloc.size() }).first;
                }
                A.se0 = it->second[0]; A.ne = it->second[1]; A.ve0 = it->second[2]; A.nv = it->second[3];
                D.as_max_ne = std::max(D.as_max_ne, A.ne); D.as_max_nv = std::max(D.as_max_nv, A.nv);
            }
            if (overflow) { P.release(); delete b; return fail(SWF_E_UNSUPPORTED, "assembly program: more than 4095 contributions to one entry of the reduced system"); }
            b->asm_programs = (int)seen.size();
            auto nonempty_i = [](std::vector<int>& v) { if (v.empty()) v.push_back(0); };
            auto nonempty_u = [](std::vector<unsigned>& v) { if (v.empty()) v.push_back(0u); };
            nonempty_i(t_dst); nonempty_u(t_cnt); nonempty_i(t_src0); nonempty_i(t_aux); nonempty_i(t_src);
            nonempty_i(tv_loc); nonempty_i(tv_red); nonempty_u(tv_cnt); nonempty_i(tv_src0); nonempty_i(tv_i); nonempty_i(tv_src);
            PUT(asw, asw); PUT(s_tnz, tnz);
            D.rr_nmax = b->rr_nmax;
            PUT(as_dst, t_dst); PUT(as_cnt, t_cnt); PUT(as_src0, t_src0); PUT(as_aux, t_aux); PUT(as_src, t_src);
            PUT(av_loc, tv_loc); PUT(av_red, tv_red); PUT(av_cnt, tv_cnt); PUT(av_src0, tv_src0); PUT(av_i, tv_i); PUT(av_src, tv_src);
        }
        {
            std::vector<Pair> vd, vo;
            for (int i : pd) vd.push_back(B.pair[i]);
            for (int i : po) vo.push_back(B.pair[i]);
            PUT(pair_d, vd); PUT(pair_o, vo);
        }
        {
            std::vector<Clique> v;
            for (int i : cle) v.push_back(B.cl[i]);
            PUT(cle_rec, v);
            for (int k = 0; k < 5; k++) {
                v.clear();
                for (int i : clc[k]) v.push_back(B.cl[i]);
                D.n_clc[k] = (int)clc[k].size(); rc |= P.put(v, &D.clc_rec[k]);
            }
        }
    }
#undef PUT
    // mutable buffers
    rc |= P.zeros(B.n_x, &D.x); rc |= P.zeros(B.n_x, &D.xc); rc |= P.zeros(B.n_x, &D.x0);
    rc |= P.zeros(B.n_loc, &D.g); rc |= P.zeros(B.n_loc, &D.diag); rc |= P.zeros(B.n_loc, &D.rhs); rc |= P.zeros(B.n_loc, &D.vc);
    rc |= P.zeros(B.n_loc, &D.y); rc |= P.zeros(B.n_loc, &D.step);
    rc |= P.zeros(B.S_tot, &D.S); rc |= P.zeros(B.Lt_tot, &D.L);
    if (b->max_red > b->rr_nmax && b->max_red <= CB_NMAX) rc |= P.zeros((size_t)n * (CB_MAXT - 1) * 256, &D.Linv);
    // few windows, one of them on the streamed Cholesky: the factorisation is spread over the chip, two tile columns per launch (k_chol_col)
    if (b->max_red > b->rr_nmax && b->max_red <= CC_NMAX && n * 4 <= b->n_cu && !getenv("SWF_NO_CHOL_COL")) rc |= P.zeros(B.Lt_tot, &D.Wk);      // >= 4 workgroups per window
    rc |= P.zeros((size_t)n, &D.ws); rc |= P.zeros((size_t)n, &b->ws_alt); rc |= P.zeros((size_t)n, &b->ws_alt2); rc |= P.zeros((size_t)n * SWF_MAX_TRACE, &D.trace);
    b->ws_primary = D.ws;
    size_t np = (size_t)D.n_proj;
    rc |= P.zeros(2 * np, &D.p_r); rc |= P.zeros(12 * np, &D.p_Jp); rc |= P.zeros(6 * np, &D.p_Jl);
    rc |= P.zeros((size_t)std::max(1, D.n_fsb), &D.p_cpart); rc |= P.zeros((size_t)std::max(1, D.n_lmb), &D.p_apart);
    rc |= P.zeros((size_t)std::max(1, D.n_pch), &D.pr_cpart); rc |= P.zeros((size_t)std::max(1, D.n_pch), &D.pr_apart);
    rc |= P.zeros(6 * (size_t)D.n_lm, &D.lm_Einv); rc |= P.zeros(3 * (size_t)D.n_lm, &D.lm_g);
    rc |= P.zeros(B.P_tot * GEMM_SPLIT, &D.P);
    rc |= P.zeros((size_t)std::max(1, 6 * B.n_fr) * GEMM_SPLIT, &D.lmq);
    rc |= P.zeros((size_t)B.r_tot, &D.g_r); rc |= P.zeros((size_t)B.j_tot, &D.g_J);
    rc |= P.zeros((size_t)D.n_gf, &D.g_cost); rc |= P.zeros((size_t)D.n_gf, &D.g_aux);
    {
        const double* ci = nullptr; const double* di = nullptr;
        B.C_init.resize((size_t)B.C_tot, 0.0); B.dgraw_init.resize((size_t)B.v_tot, 0.0);
        rc |= P.put(B.C_init, &ci); rc |= P.put(B.dgraw_init, &di);
        D.C = (double*)ci; D.cv_dgraw = (double*)di;
    }
    rc |= P.zeros((size_t)B.v_tot, &D.cv_graw); rc |= P.zeros((size_t)B.v_tot, &D.cv_cs);
    rc |= P.zeros((size_t)B.e_tot, &D.cE);
    // composite IMU-GNSS factors: operator arguments over the whole batch + where each factor's prior record and clique live
    b->n_comp = (int)B.co_M.size();
    if (b->n_comp) {
        const int nc = b->n_comp;
        std::vector<int> eo(nc + 1, 0), no(nc + 1, 0), roff(nc), x0off(nc), voff(nc);
        std::vector<long long> pno(nc + 1, 0), nno(nc + 1, 0), go(nc + 1, 0), g2o(nc + 1, 0), Joff(nc), Coff(nc);
        for (int f = 0; f < nc; f++) {
            const int M = B.co_M[f], N = B.co_N[f], G = 30 + N;
            b->comp_nmax = std::max(b->comp_nmax, N); b->comp_nmin = std::min(b->comp_nmin, N);
            eo[f + 1] = eo[f] + M; no[f + 1] = no[f] + N; pno[f + 1] = pno[f] + 15LL * M * N; nno[f + 1] = nno[f] + (long long)N * N;
            go[f + 1] = go[f] + G; g2o[f + 1] = g2o[f] + (long long)G * G;
            const GFac& Gf = B.gf[B.co_gf[f]];
            const Clique& Cq = B.cl[Gf.clique];
            // either its own static clique (every block outside group 0: k_comp_scatter writes C = H and the diagonal there), or a member of
            // the clique of the ONE group-0 block it touches (its rows reach the elimination through the clique's Jacobian: Coff = -1)
            if (Cq.is_static ? (Cq.d_f != G || Cq.d_e != 0) : (Cq.d_e <= 0)) { P.release(); delete b; return fail(SWF_E_UNSUPPORTED, "composite factor: its blocks must all be variable"); }
            Joff[f] = B.prior_Joff[Gf.data]; roff[f] = B.prior_roff[Gf.data]; x0off[f] = B.prior_x0off[Gf.data];
            Coff[f] = Cq.is_static ? Cq.C_off : -1; voff[f] = Cq.is_static ? Cq.v_off : -1;
        }
        b->comp_ne = eo[nc];
        CompArgs& A = b->CA; CompMeta& Mt = b->CM;
        A.n = nc; A.want_jac = 1;
        rc |= P.put(B.co_M, &A.M); rc |= P.put(B.co_N, &A.N); rc |= P.put(eo, &A.e_off); rc |= P.put(no, &A.n_off);
        rc |= P.put(pno, &A.pn_off); rc |= P.put(nno, &A.nn_off); rc |= P.put(go, &A.g_off); rc |= P.put(g2o, &A.g2_off);
        { const double* t1 = nullptr; const double* t2 = nullptr; rc |= P.put(B.co_pose, &t1); rc |= P.put(B.co_sb, &t2); A.pose = (double*)t1; A.sb = (double*)t2; }
        { const double* t1 = nullptr; const double* t2 = nullptr; rc |= P.put(B.co_pose, &t1); rc |= P.put(B.co_sb, &t2); b->co_pose0 = (double*)t1; b->co_sb0 = (double*)t2; }
        rc |= P.put(B.co_pose_lin, &A.pose_lin); rc |= P.put(B.co_sb_lin, &A.sb_lin); rc |= P.put(B.co_Hpp, &A.Hpp); rc |= P.put(B.co_HpN, &A.HpN);
        rc |= P.put(B.co_rhs_p, &A.rhs_p); rc |= P.put(B.co_HNN, &A.HNN); rc |= P.put(B.co_rhsN, &A.rhsN); rc |= P.put(B.co_pre, &A.pre); rc |= P.put(B.co_pbgw, &A.pbgw);
        rc |= P.put(B.co_mid, &A.mid); rc |= P.put(B.co_H12, &A.H12);
        rc |= P.zeros((size_t)eo[nc] * 225, &A.hmn_inv); rc |= P.zeros((size_t)eo[nc] * 225, &A.hmn_2); rc |= P.zeros((size_t)eo[nc] * 225, &A.hmn_0);
        rc |= P.zeros((size_t)pno[nc], &A.hmn_N); rc |= P.zeros((size_t)eo[nc] * 15, &A.rhsmn);
        rc |= P.zeros((size_t)g2o[nc], &A.Hd); rc |= P.zeros((size_t)go[nc], &A.rd); rc |= P.zeros((size_t)g2o[nc], &A.Ld); rc |= P.zeros((size_t)go[nc], &A.r0);
        rc |= P.zeros((size_t)nc * 32, &A.old); rc |= P.zeros((size_t)no[nc], &A.N_old); rc |= P.zeros((size_t)nc, &A.history); rc |= P.zeros((size_t)nc, &A.status);
        rc |= P.zeros((size_t)nc * 32, &Mt.outer); rc |= P.zeros((size_t)no[nc], &Mt.Nv); rc |= P.zeros((size_t)nc, &Mt.active);
        A.outer = Mt.outer; A.Nv = Mt.Nv; A.active = Mt.active;
        rc |= P.zeros((size_t)go[nc], &A.res_out); rc |= P.zeros((size_t)g2o[nc], &A.jac_out);
        rc |= P.zeros((size_t)(eo[nc] + nc) * 450, &A.Jw); rc |= P.zeros((size_t)(eo[nc] + nc) * 16, &A.rw);
        {
            std::vector<int> qf, qk;
            for (int f = 0; f < nc; f++) for (int k = 0; k <= B.co_M[f]; k++) { qf.push_back(f); qk.push_back(k); }
            A.n_iq = (int)qf.size();
            rc |= P.put(qf, &A.iq_f); rc |= P.put(qk, &A.iq_k); rc |= P.zeros((size_t)nc, &A.todo);
        }
        rc |= P.put(B.co_win, &Mt.win); rc |= P.put(B.co_xo_off, &Mt.xo_off); rc |= P.put(B.co_xo, &Mt.xo);
        rc |= P.put(Joff, &Mt.Joff); rc |= P.put(roff, &Mt.roff); rc |= P.put(x0off, &Mt.x0off); rc |= P.put(Coff, &Mt.Coff); rc |= P.put(voff, &Mt.voff);
        Mt.prior_J = (double*)D.prior_J; Mt.prior_Jt = (double*)D.prior_Jt; Mt.prior_r0 = (double*)D.prior_r0; Mt.prior_x0 = (double*)D.prior_x0;
    }
    if (!rc) rc = P.flush();
    if (rc) { P.release(); delete b; return fail(SWF_E_NODEVICE, "device allocation / upload failed"); }
    *out = b;
    int urc = swf_batch_upload_state(b);
    if (urc != SWF_OK) { swf_batch_destroy(b); *out = nullptr; return urc; }
    return SWF_OK;
}

// the static block partition of SURVEY.md 8e (512 windows, 64 per GPU at 8): shard k of G takes `count` consecutive windows from `first`,
// the first n % G shards one more — the same rule as the harness's shard.partition (no device needed)
extern "C" int swf_shard_partition(int32_t n, int32_t G, int32_t k, int32_t* first, int32_t* count) {
    if (n < 0 || G <= 0 || k < 0 || k >= G || !first || !count) return fail(SWF_E_INVALID, "swf_shard_partition: bad arguments");
    const int q = n / G, r = n % G;
    *first = k * q + std::min(k, r); *count = q + (k < r ? 1 : 0);
    return SWF_OK;
}

// ---- several GPUs of one node from one process (windows are independent units, "one host thread + one HIP stream per
// GPU", no data-path collective).  swf_batch_solve only ENQUEUES work on its batch's stream, so a single host thread keeps all devices busy.
extern "C" int swf_batch_create_on(int32_t device, const swf_flat_window* const* windows, int32_t n, void* stream, swf_batch** out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SWF_E_NODEVICE, "no HIP device: this library has no CPU fallback");
    if (device < 0 || device >= ndev || device >= SWF_MAX_DEVICES) return fail(SWF_E_INVALID, "swf_batch_create_on: no such device");
    DeviceGuard dg_(device);
    return swf_batch_create(windows, n, stream, out);
}

// windows [0, n) dealt in contiguous, near-equal blocks to the devices of device_mask (bit d = use device d; 0 = every visible device):
// one batch per device that receives windows, out_batches[k] / out_first[k] / out_count[k] for k < *n_batches (capacity: the device count)
extern "C" int swf_batch_create_sharded(const swf_flat_window* const* windows, int32_t n, uint32_t device_mask,
                                        swf_batch** out_batches, int32_t* out_first, int32_t* out_count, int32_t* n_batches) {
    if (!windows || n <= 0 || !out_batches || !n_batches) return fail(SWF_E_INVALID, "swf_batch_create_sharded: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SWF_E_NODEVICE, "no HIP device: this library has no CPU fallback");
    std::vector<int> devs;
    for (int d = 0; d < ndev && d < SWF_MAX_DEVICES; d++) if (device_mask == 0 || ((device_mask >> d) & 1u)) devs.push_back(d);
    if (devs.empty()) return fail(SWF_E_INVALID, "swf_batch_create_sharded: device_mask selects no visible device");
    const int G = (int)std::min<size_t>(devs.size(), (size_t)n);
    *n_batches = 0;
    for (int k = 0; k < G; k++) {
        int32_t lo = 0, cnt = 0;
        swf_shard_partition(n, G, k, &lo, &cnt);
        const int hi = lo + cnt;
        swf_batch* bk = nullptr;
        int rc = swf_batch_create_on(devs[k], windows + lo, hi - lo, nullptr, &bk);
        if (rc != SWF_OK) { for (int q = 0; q < *n_batches; q++) swf_batch_destroy(out_batches[q]); *n_batches = 0; return rc; }
        out_batches[k] = bk;
        if (out_first) out_first[k] = lo;
        if (out_count) out_count[k] = hi - lo;
        *n_batches = k + 1;
    }
    return SWF_OK;
}

extern "C" int swf_batch_device(swf_batch* b, int32_t* device) { if (!b || !device) return fail(SWF_E_INVALID, "bad arguments"); *device = b->device; return SWF_OK; }

extern "C" int swf_batch_destroy(swf_batch* b) {
    if (!b) return SWF_OK;
    DeviceGuard dg_(b->device);
    (void)hipStreamSynchronize(b->stream);
    if (b->aux) (void)hipStreamSynchronize(b->aux);     // nothing of this batch may still run when its slabs go back to the cache
    for (auto& e : b->ev) handle_cache().give(e, true);
    if (b->h_x) (void)hipHostFree(b->h_x);
    if (b->h_sum) (void)hipHostFree(b->h_sum);
    b->pool.release();
    delete b;
    return SWF_OK;
}

extern "C" int swf_batch_upload_state(swf_batch* b) {
    DeviceGuard dg_(b ? b->device : -1);
    if (!b) return fail(SWF_E_INVALID, "null batch");
    // the parameter blocks are gathered into ONE page-locked buffer the batch keeps (round 5: a pageable source made the runtime stage the
    // copy through its own pinned chunks: ~0.8 ms for the 6.7 MB of a 512-window batch)
    const size_t nx = (size_t)b->D.n_x;
    if (!b->h_x && nx) HIPCHK(hipHostMalloc((void**)&b->h_x, nx * sizeof(double), hipHostMallocDefault));
    double* xh = b->h_x;
    for (size_t i = 0; i < b->win.size(); i++) {
        const HostWin& h = b->hw[i];
        double* p = xh + b->win[i].x_base;
        memcpy(p, h.pose, sizeof(double) * 7 * h.n_pose); p += 7 * h.n_pose;
        memcpy(p, h.sb, sizeof(double) * 9 * h.n_sb); p += 9 * h.n_sb;
        memcpy(p, h.lm, sizeof(double) * 3 * h.n_lm); p += 3 * h.n_lm;
        memcpy(p, h.sc, sizeof(double) * h.n_sc);
    }
    HIPCHK(hipMemcpyAsync(b->D.x, xh, nx * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->D.x0, b->D.x, nx * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->D.xc, b->D.x, nx * sizeof(double), hipMemcpyDeviceToDevice, b->stream));      // (the candidate's constant blocks: k_dogleg writes the variable ones only)
    std::vector<double> hp, hs;
    if (b->n_comp) {
        hp.resize((size_t)b->comp_ne * 7); hs.resize((size_t)b->comp_ne * 9);
        for (size_t i = 0; i < b->win.size(); i++) {
            const HostWin& h = b->hw[i];
            if (!h.comp_ne) continue;
            memcpy(hp.data() + (size_t)h.comp_e0 * 7, h.comp_pose, sizeof(double) * 7 * h.comp_ne);
            memcpy(hs.data() + (size_t)h.comp_e0 * 9, h.comp_sb, sizeof(double) * 9 * h.comp_ne);
        }
        HIPCHK(hipMemcpyAsync(b->co_pose0, hp.data(), hp.size() * sizeof(double), hipMemcpyHostToDevice, b->stream));
        HIPCHK(hipMemcpyAsync(b->co_sb0, hs.data(), hs.size() * sizeof(double), hipMemcpyHostToDevice, b->stream));
        HIPCHK(hipMemcpyAsync(b->CA.pose, b->co_pose0, hp.size() * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
        HIPCHK(hipMemcpyAsync(b->CA.sb, b->co_sb0, hs.size() * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
        HIPCHK(hipMemsetAsync(b->CA.history, 0, (size_t)b->n_comp * sizeof(int), b->stream));
    }
    HIPCHK(hipStreamSynchronize(b->stream));     // staging buffers have stack lifetime
    return SWF_OK;
}

extern "C" int swf_batch_reset_state(swf_batch* b) {
    DeviceGuard dg_(b ? b->device : -1);
    if (!b) return fail(SWF_E_INVALID, "null batch");
    int n = b->D.n_x;
    hipLaunchKernelGGL(k_copy, dim3((n + 255) / 256), dim3(256), 0, b->stream, b->D.x, (const double*)b->D.x0, n);
    HIPCHK(hipGetLastError());
    if (b->n_comp) {       // hidden epochs back to the uploaded values, composite factors forget their last linearisation
        HIPCHK(hipMemcpyAsync(b->CA.pose, b->co_pose0, (size_t)b->comp_ne * 7 * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
        HIPCHK(hipMemcpyAsync(b->CA.sb, b->co_sb0, (size_t)b->comp_ne * 9 * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
        HIPCHK(hipMemsetAsync(b->CA.history, 0, (size_t)b->n_comp * sizeof(int), b->stream));
    }
    return SWF_OK;
}

static DevOpt to_devopt(const swf_options* o) {
    DevOpt d{};
    d.max_iter = o->max_num_iterations; d.step_mode = o->step_mode; d.strategy = o->trust_region_strategy; d.jacobi = o->jacobi_scaling ? 1 : 0;
    d.r0 = o->initial_trust_region_radius; d.max_r = o->max_trust_region_radius; d.min_r = o->min_trust_region_radius;
    d.min_rel_dec = o->min_relative_decrease; d.ftol = o->function_tolerance; d.gtol = o->gradient_tolerance;
    d.ptol = o->parameter_tolerance; d.min_mu = o->min_mu; d.max_mu = o->max_mu; d.mu_inc = o->mu_increase_factor;
    d.min_diag = o->min_diagonal; d.max_diag = o->max_diagonal;
    return d;
}

#define GRID(n, per) dim3((unsigned)(((n) + (per) - 1) / (per)))
namespace {
struct Launcher {
    swf_batch* b; DevOpt O; hipStream_t st;
    bool lm_folded = false;     // k_lm_schur of the current linearisation wrote ONE folded product (else GEMM_SPLIT partials)
    bool export_full = false;   // k_chol_rr4 writes the whole factor (ASSEMBLE_ELIMINATE_ONLY: marginalisation, swf_batch_export_reduced), else the tail block only
    int lm_next = 0, lm_qpb = 1;                 // first tile of the ranges of k_lm_schur still to launch (with the clique kernels)
    int ls_tiles_per_launch() const { return b->ls_var == 0 ? 16 : b->ls_var == 1 ? 40 : 72; }
    // one launch of the landmark Schur kernel over the tile-list entries [tile_base, tile_base + tiles per launch), by row class
    WinState* ws_next(WinState* p) const { return p == b->ws_primary ? b->ws_alt : p == b->ws_alt ? b->ws_alt2 : b->ws_primary; }
    bool decide_fused = false;  // this iteration's k_decide rides at the head of the elimination grid (k_decide_lm_clique)
    void lm_launch(int tile_base, hipStream_t on, int clique_rows = 0) {
        DevBatch& D = b->D;
        dim3 grid(D.n_win, GEMM_SPLIT / b->ls_qpb);
        const int qpb = b->ls_qpb, sd = b->s_direct ? 1 : 0, lp = tile_base / ls_tiles_per_launch(), kms = b->ls_kms;
        if (clique_rows > 0) {
            const int np = (int)grid.y;
            grid.y += clique_rows;
            if (decide_fused) {
                WinState* out = ws_next(D.ws);                  // (its failure flags were cleared by k_step_eval's lead)
                if (b->ls_var == 0) hipLaunchKernelGGL((k_decide_lm_clique<8, 2, 2, 80>), grid, dim3(LS_NT(8, 2)), 0, on, D, O, qpb, lp, kms, sd, np, out);
                else hipLaunchKernelGGL((k_decide_lm_clique<8, 5, 2, 144>), grid, dim3(LS_NT(8, 2)), 0, on, D, O, qpb, lp, kms, sd, np, out);
                D.ws = out;
                return;
            }
            switch (b->ls_var) {
            case 0: hipLaunchKernelGGL((k_lm_clique<8, 2, 2, 80>), grid, dim3(LS_NT(8, 2)), 0, on, D, O, qpb, lp, kms, sd, np); break;
            case 1: hipLaunchKernelGGL((k_lm_clique<8, 5, 2, 144>), grid, dim3(LS_NT(8, 2)), 0, on, D, O, qpb, lp, kms, sd, np); break;
            default: hipLaunchKernelGGL((k_lm_clique<12, 6, 1, 272>), grid, dim3(LS_NT(12, 1)), 0, on, D, O, qpb, lp, kms, 0, np); break;
            }
            return;
        }
        switch (b->ls_var) {
        case 0: hipLaunchKernelGGL((k_lm_schur<8, 2, 2, 80, true>), grid, dim3(LS_NT(8, 2)), 0, on, D, O, qpb, lp, kms, sd); break;
        case 1: hipLaunchKernelGGL((k_lm_schur<8, 5, 2, 144, true>), grid, dim3(LS_NT(8, 2)), 0, on, D, O, qpb, lp, kms, sd); break;
        case 2: hipLaunchKernelGGL((k_lm_schur<12, 6, 1, 272, true>), grid, dim3(LS_NT(12, 1)), 0, on, D, O, qpb, lp, kms, 0); break;
        default: hipLaunchKernelGGL((k_lm_schur<12, 6, 1, 400, true>), grid, dim3(LS_NT(12, 1)), 0, on, D, O, qpb, lp, kms, 0); break;
        }
    }
    // optional event pair around one launch
    struct Bracket {
        Launcher& L; int slot; hipStream_t bst;
        Bracket(Launcher& l, int kind, hipStream_t on = nullptr) : L(l), slot(-1), bst(on ? on : l.st) {
            swf_batch* b = L.b;
            if (!(b->timing & (1 << kind))) return;
            if ((size_t)(b->ev_used + 1) * 2 > b->ev.size()) {
                size_t old = b->ev.size();
                b->ev.resize(old + 64);
                bool ok = true;
                for (size_t i = old; i < b->ev.size(); i++) { b->ev[i] = handle_cache().event(true); ok = ok && b->ev[i] != nullptr; }
                if (!ok) {                                 // no events to be had: this launch goes untimed
                    for (size_t i = old; i < b->ev.size(); i++) handle_cache().give(b->ev[i], true);
                    b->ev.resize(old); return;
                }
            }
            slot = b->ev_used++;
            b->ev_kind.push_back(kind);
            (void)hipEventRecord(b->ev[2 * slot], bst);
        }
        ~Bracket() { if (slot >= 0) (void)hipEventRecord(L.b->ev[2 * slot + 1], bst); }
    };
    static int nb(size_t n, int per) { return (int)((n + per - 1) / per); }
    // One linearisation.  Dependencies: the cliques need the IMU and the scalar-factor Jacobians (k_eval_imu, k_eval_ps);
    // k_lm_schur needs k_eval_ps; k_assemble_flat needs everything.  With an auxiliary
    // stream (small batches) the IMU / clique branch runs next to the projection / landmark branch.
    // the reference-topology latency path (swf_kernels4.h, k_lm_comp): the composite chain and the visual branch in shared grids
    bool comp_fused = false;
    bool comp_fuse_ok(int write_S) const {
        const DevBatch& D = b->D;
        if (!b->n_comp || !write_S || b->no_comp_fuse || !b->lat_fuse || b->aux || b->comp_eigen_root) return false;
        if (b->comp_nmax > CO_SMALLN || b->ls_var > 1 || !D.n_lm || D.n_imu || D.n_idp || b->max_prior_dim > PRIOR_LDS_DIM) return false;
        if (D.n_clc[0] || D.n_clc[1]) return false;                                     // (the cliques of such a window: classes 2, 4 and — from 19 ambiguities on — 3)
        const int crow = (b->n_comp + D.n_win - 1) / D.n_win;
        return (long long)D.n_win * (GEMM_SPLIT / b->ls_qpb + crow) <= b->n_cu;         // every workgroup of k_lm_comp resident at once
    }
    // spec: the Jacobian evaluation of the dogleg loop's speculative flow — at the candidate of every window with a proposed step
    // (swf_kernels.h: eval_gate / eval_src); the evaluation kernels get a copy of the batch record with the flag set
    void lin_eval(int write_S, bool spec = false) {
        DevBatch Dv = b->D; Dv.spec = spec ? 1 : 0;
        const DevBatch& D = Dv;
        comp_fused = !spec && comp_fuse_ok(write_S);
        if (comp_fused) {
            hipLaunchKernelGGL(k_comp_gather_prep, dim3(b->n_comp), dim3(256), 0, st, D, b->CA, b->CM);
            Segs S{}; S.e[0] = D.n_fsb; S.e[1] = S.e[0] + nb(D.n_sc, 256);
            {   // projection + scalar factors next to the chains' IMU factors
                Bracket t(*this, SWF_K_EVAL_PS);
                hipLaunchKernelGGL(k_eval_ps_comp_imu, dim3(S.e[1] + (b->CA.n_iq + 7) / 8), dim3(256), 0, st, D, S, b->CA);
            }
            {   // the landmark Schur product (first tile range) next to the re-elimination of the hidden epochs
                Bracket t(*this, SWF_K_LM_SCHUR);
                lm_qpb = b->ls_qpb; lm_folded = b->ls_folded;
                const int np = GEMM_SPLIT / b->ls_qpb, crow = (b->n_comp + D.n_win - 1) / D.n_win;
                const int qpb = b->ls_qpb, sd = b->s_direct ? 1 : 0, kms = b->ls_kms;
                dim3 grid(D.n_win, np + crow);
                if (b->ls_var == 0) hipLaunchKernelGGL((k_lm_comp<8, 2, 2, 80>), grid, dim3(LS_NT(8, 2)), 0, st, D, O, b->CA, b->CM, qpb, 0, kms, sd, np);
                else hipLaunchKernelGGL((k_lm_comp<8, 5, 2, 144>), grid, dim3(LS_NT(8, 2)), 0, st, D, O, b->CA, b->CM, qpb, 0, kms, sd, np);
                lm_next = ls_tiles_per_launch();
            }
            if (D.n_prior) {   // the prior records (the composite factors' among them, just rewritten): the prior segment of k_eval_ps alone
                Segs P{}; P.e[2] = D.n_pch;
                hipLaunchKernelGGL((k_eval_ps<true, true>), dim3(P.e[2]), dim3(256), 0, st, D, P);
            }
            return;
        }
        if (b->n_comp && !spec) {
            // composite IMU-GNSS factors of the windows that re-linearise: hidden epochs move, re-elimination, prior records rewritten
            // 1024 threads per factor while the chip holds every factor at once (two such workgroups per CU), 256 for larger batches
            const bool wide = b->n_comp <= 2 * b->n_cu;
            {
            hipLaunchKernelGGL(k_comp_gather_prep, dim3(b->n_comp), dim3(256), 0, st, D, b->CA, b->CM);
            hipLaunchKernelGGL(k_comp_imu, dim3((b->CA.n_iq + 7) / 8), dim3(256), 0, st, b->CA);
            // (an instantiation per class of factor, each passing over the other's: a factor's arithmetic does not depend on its batch)
            if (b->comp_nmin <= CO_SMALLN) {
                if (wide) hipLaunchKernelGGL((k_comp_elim<CO_SMALLN, 1024>), dim3(b->n_comp), dim3(1024), 0, st, b->CA, 0);
                else {
                    // batches: factors of up to 12 ambiguities in the instantiation that fits six workgroups to a CU
                    if (b->comp_nmin <= CO_TINYN) hipLaunchKernelGGL((k_comp_elim<CO_TINYN, 256>), dim3(b->n_comp), dim3(256), 0, st, b->CA, 0);
                    if (b->comp_nmax > CO_TINYN) hipLaunchKernelGGL((k_comp_elim<CO_SMALLN, 256>), dim3(b->n_comp), dim3(256), 0, st, b->CA, CO_TINYN + 1);
                }
            }
            if (b->comp_nmax > CO_SMALLN) {
                if (wide) hipLaunchKernelGGL((k_comp_elim<CO_MAXN, 1024>), dim3(b->n_comp), dim3(1024), 0, st, b->CA, CO_SMALLN + 1);
                else hipLaunchKernelGGL((k_comp_elim<CO_MAXN, 256>), dim3(b->n_comp), dim3(256), 0, st, b->CA, CO_SMALLN + 1);
            }
            if (b->comp_eigen_root) {
                if (b->comp_nmax <= CO_SMALLN) hipLaunchKernelGGL(k_comp_eigroot<CO_SMALLN>, dim3(b->n_comp), dim3(256), 0, st, b->CA);
                else hipLaunchKernelGGL(k_comp_eigroot<CO_MAXN>, dim3(b->n_comp), dim3(512), 0, st, b->CA);
            }
            hipLaunchKernelGGL(k_comp_scatter, dim3(b->n_comp), dim3(256), 0, st, D, b->CA, b->CM);
            }
        }
        // (speculative flow: k_decide, on the main stream, reads every family's candidate costs — the whole evaluation stays on the main
        // stream, and the fork event the clique branch waits for is recorded behind k_decide, see swf_batch_solve)
        const bool fork = b->aux && !spec;
        hipStream_t sa = fork ? b->aux : st;
        if (fork) { (void)hipEventRecord(b->ev_fork[0], st); (void)hipStreamWaitEvent(b->aux, b->ev_fork[0], 0); }
        bool imu_fused = false;
        if (D.n_proj + D.n_sc + D.n_prior) {
            Bracket t(*this, SWF_K_EVAL_PS);
            bool pf = b->max_prior_dim <= PRIOR_LDS_DIM;          // priors fused as a segment
            // (the projection segment: one workgroup per frame-sum block, the per-frame sums formed in the same kernel)
            Segs S{}; S.e[0] = D.n_fsb; S.e[1] = S.e[0] + nb(D.n_sc, 256); S.e[2] = S.e[1] + (pf ? D.n_pch : 0);
            imu_fused = b->lat_fuse && D.n_imu > 0;          // latency path: the IMU factors as a segment of this grid
            S.e[3] = S.e[2] + (imu_fused ? nb(D.n_imu, IMU_FPB) : 0);
            if (imu_fused) hipLaunchKernelGGL((k_eval_ps<true, true, true>), dim3(S.e[3]), dim3(256), 0, st, D, S);
            else hipLaunchKernelGGL((k_eval_ps<true, true>), dim3(S.e[2]), dim3(256), 0, st, D, S);
        }
        if (D.n_idp) hipLaunchKernelGGL(k_eval_idp<true>, GRID(D.n_idp, 128), dim3(128), 0, st, D);
        // (large priors before the fork event: a prior-type record inside a group-0 clique writes its rows into that clique's Jacobian, and a
        // clique class with IMU factors runs on the auxiliary stream behind ev_fork[1] alone)
        if (D.n_prior && b->max_prior_dim > PRIOR_LDS_DIM) { Bracket t(*this, SWF_K_EVAL_PRIOR); hipLaunchKernelGGL(k_eval_prior<true>, dim3(D.n_pch), dim3(256), (2 * b->max_prior_dim + 16) * sizeof(double), st, D); }
        if (fork) (void)hipEventRecord(b->ev_fork[1], st);
        if (D.n_imu && !imu_fused) { Bracket t(*this, SWF_K_EVAL_IMU, sa); hipLaunchKernelGGL(k_eval_imu<true>, GRID(D.n_imu, IMU_FPB), dim3(IMU_FPB * IMU_LPF), 0, sa, D); }
    }
    void lin_elim(int write_S) {
        DevBatch& D = b->D;
        bool clq_fused = false;
        if (comp_fused) {
            // (k_lm_comp of lin_eval took the first tile range of the landmark product)
            Bracket t(*this, SWF_K_CLIQUE_ELIM);
            if (D.n_clc[3]) {
                // class 3 next to class 2 in one grid; class 4 (other factors of the batch with fewer ambiguities) on its own
                hipLaunchKernelGGL(k_clique_big2, dim3(D.n_clc[3] + D.n_clc[2]), dim3(CB_NT), 0, st, D, O);
                if (D.n_clc[4]) hipLaunchKernelGGL(k_clique_tall, dim3(D.n_clc[4]), dim3(256), 0, st, D, O);
            } else if (D.n_clc[4] + D.n_clc[2]) hipLaunchKernelGGL(k_clique_tall2, dim3(D.n_clc[4] + D.n_clc[2]), dim3(256), 0, st, D, O);
        } else if (D.n_lm) {
            Bracket t(*this, write_S ? SWF_K_LM_SCHUR : SWF_K_LM_ELIM);
            lm_qpb = b->ls_qpb; lm_folded = b->ls_folded;
            if (!write_S) {
                // cost / gradient pass (the solve's final linearisation): the elimination alone — producer waves only, no panel
                dim3 grid(D.n_win, GEMM_SPLIT / lm_qpb);
                hipLaunchKernelGGL((k_lm_schur<8, 2, 2, 80, false>), grid, dim3(256), 0, st, D, O, lm_qpb, 0, 0, 0);
            } else {
                // latency path: the one-wavefront cliques ride in the same grid (k_lm_clique); the 64-frame class has no LDS to spare for them
                // (a workgroup of that grid fills a CU: only while all of them — landmark parts and cliques — are resident at once)
                const int crow = D.n_win > 0 ? (D.n_clc[2] + D.n_win - 1) / D.n_win : 0;
                clq_fused = clq_fuse_ok();
                lm_launch(0, st, clq_fused ? crow : 0);
                lm_next = ls_tiles_per_launch();        // further tile ranges: launched below, on the auxiliary stream when there is one
            }
        }
        {
            hipStream_t sa = b->aux ? b->aux : st;
            if (b->aux) (void)hipStreamWaitEvent(b->aux, b->ev_fork[1], 0);          // scalar-factor Jacobians (k_eval_ps)
            // latency path: a class without IMU factors needs only k_eval_ps and runs on the main stream, next to the IMU branch
            auto cstream = [&](int cls) { return (b->aux && !b->clc_imu[cls]) ? st : sa; };
            if (comp_fused) {
                for (; lm_next < b->max_tiles; lm_next += ls_tiles_per_launch()) { Bracket t(*this, SWF_K_LM_SCHUR, sa); lm_launch(lm_next, sa); }
            } else {
            if (D.n_clc[1]) { Bracket t(*this, SWF_K_CLIQUE_ELIM, cstream(1)); hipLaunchKernelGGL((k_clique_elim<32, 48, 9, 1>), dim3(D.n_clc[1]), dim3(64), 0, cstream(1), D, O); }
            if (D.n_clc[0]) { Bracket t(*this, SWF_K_CLIQUE_ELIM, cstream(0)); hipLaunchKernelGGL((k_clique_elim<48, 32, 1, 0>), dim3(D.n_clc[0]), dim3(64), 0, cstream(0), D, O); }
            if (D.n_clc[2] && !clq_fused) {
                Bracket t(*this, SWF_K_CLIQUE_ELIM, cstream(2));
                if (b->lat_fuse) hipLaunchKernelGGL(k_clique_elim4, dim3(D.n_clc[2]), dim3(256), 0, cstream(2), D, O);      // latency form: four waves per clique, same bits
                else hipLaunchKernelGGL((k_clique_elim<64, 64, 9, 2>), dim3(D.n_clc[2]), dim3(64), 0, cstream(2), D, O);
            }
            if (D.n_clc[3]) { Bracket t(*this, SWF_K_CLIQUE_ELIM, cstream(3)); hipLaunchKernelGGL(k_clique_big, dim3(D.n_clc[3]), dim3(CB_NT), 0, cstream(3), D, O); }
            if (D.n_clc[4]) { Bracket t(*this, SWF_K_CLIQUE_ELIM, cstream(4)); hipLaunchKernelGGL(k_clique_tall, dim3(D.n_clc[4]), dim3(256), 0, cstream(4), D, O); }
            if (write_S && D.n_lm) {
                // further tile ranges write nothing but their tiles of P (k_lm_schur: outs), so on the latency path they run behind the
                // IMU / clique branch, next to the first range
                for (; lm_next < b->max_tiles; lm_next += ls_tiles_per_launch()) { Bracket t(*this, SWF_K_LM_SCHUR, sa); lm_launch(lm_next, sa); }
            }
            }
            if (b->aux) (void)hipEventRecord(b->ev_fork[2], b->aux);
        }
        // chunked priors with a static clique: graw = J^T r over column chunks, from the rows the evaluation left in g_r
        if (b->n_pch_split) hipLaunchKernelGGL(k_prior_graw, dim3(D.n_pch), dim3(256), 0, st, D);
        if (b->aux) (void)hipStreamWaitEvent(st, b->ev_fork[2], 0);                  // join before the assembly
        if (D.n_pd) {
            Bracket t(*this, SWF_K_ASSEMBLE);
            const int nbS = write_S ? nb((size_t)D.as_max_ne, 256) : 0, nbV = nb((size_t)D.as_max_nv, 256);
            hipLaunchKernelGGL(k_assemble_flat, dim3(nbS + nbV, D.n_win), dim3(256), 0, st, D, O, write_S, nbS);
        }
    }
    void reduced() {
        DevBatch& D = b->D;
        Bracket t(*this, SWF_K_CHOL);
        if (b->max_red <= CB_NMAX) {
            // per-window choice (each kernel skips the other's windows): register-resident tiles up to 240 dimensions, streamed above
            if (b->min_red <= b->rr_nmax) {
                if (b->min_red <= 224) hipLaunchKernelGGL(k_chol_rr4<14>, dim3(D.n_win), dim3(R4_NT), 0, st, D, export_full ? 1 : 0);
                if (b->rr4_has15) hipLaunchKernelGGL(k_chol_rr4<15>, dim3(D.n_win), dim3(R4_NT), 0, st, D, export_full ? 1 : 0);
                if (b->rr4_has16) hipLaunchKernelGGL(k_chol_rr4<16>, dim3(D.n_win), dim3(R4_NT), 0, st, D, export_full ? 1 : 0);
            }
            if (b->max_red > b->rr_nmax && D.Wk) {
                const int Tc = (b->max_red + 15) / 16;
                const int nbw = std::max(1, std::min(CC_NB, b->n_cu / D.n_win));       // the chip divided by the windows
                for (int j = 0; j < Tc; j += 2) hipLaunchKernelGGL(k_chol_col, dim3(D.n_win, nbw), dim3(CC_NT), 0, st, D, j);
                hipLaunchKernelGGL(k_chol_big<true>, dim3(D.n_win), dim3(1024), 0, st, D);      // backward substitution
            } else if (b->max_red > b->rr_nmax) hipLaunchKernelGGL(k_chol_big<false>, dim3(D.n_win), dim3(1024), 0, st, D);
        }
        else if (b->max_red + 1 <= 256) hipLaunchKernelGGL(k_chol_solve<256>, dim3(D.n_win), dim3(256), 0, st, D);
        else hipLaunchKernelGGL(k_chol_solve<1024>, dim3(D.n_win), dim3(1024), 0, st, D);
    }
    void step_rest() {
        DevBatch& D = b->D;
        {
            Bracket t(*this, SWF_K_POST_CHOL);
            Segs S{};
            S.e[0] = D.n_lmb; S.e[1] = S.e[0] + nb((size_t)D.n_cle * 16, 256);
            S.e[2] = S.e[1]; S.e[3] = S.e[2] + nb(D.n_sc, 256);                                 // (J D^-2 g of the projections rides in segment 0)
            S.e[4] = S.e[3] + nb((size_t)D.n_imu * 16, 256); S.e[5] = S.e[4] + D.n_pch;        // one workgroup per prior row chunk
            if (D.n_win < b->n_cu) { if (S.e[5]) hipLaunchKernelGGL(k_post_chol<0>, dim3(S.e[5]), dim3(256), 0, st, D, O, S); }
            else {
                if (S.e[0]) hipLaunchKernelGGL(k_post_chol<1>, dim3(S.e[0]), dim3(256), 0, st, D, O, S);
                if (S.e[5] > S.e[0]) hipLaunchKernelGGL(k_post_chol<2>, dim3(S.e[5] - S.e[0]), dim3(256), 0, st, D, O, S);
            }
        }
        if (!step_fused) dogleg();
    }
    bool step_fused = false;    // this solve runs k_dogleg inside the candidate's evaluation (k_step_eval): one window on the latency path, speculative flow
    void dogleg() {
        DevBatch& D = b->D;
        Bracket t(*this, SWF_K_DOGLEG);
        if (b->lat_fuse) hipLaunchKernelGGL((k_dogleg<16, 4>), dim3(D.n_win), dim3(CTL_NT), 0, st, D, O);
        else hipLaunchKernelGGL((k_dogleg<4, 2>), dim3(D.n_win), dim3(CTL_NT), 0, st, D, O);
    }
    bool clq_fuse_ok() const {
        const DevBatch& D = b->D;
        const int crow = D.n_win > 0 ? (D.n_clc[2] + D.n_win - 1) / D.n_win : 0;
        return b->lat_fuse && b->ls_var <= 2 && D.n_lm > 0 && D.n_clc[2] > 0 && !D.n_clc[0] && !D.n_clc[1]
               && (long long)D.n_win * (GEMM_SPLIT / b->ls_qpb + crow) <= b->n_cu;
    }
    // k_decide at the head of the elimination grid: one window, the fused step kernel in front (three state buffers in rotation), the
    // cliques in the landmark product's grid, the two smaller size classes of that grid (the third has no LDS to spare)
    bool decide_fuse_ok() const { return step_fused && !b->no_decide_fuse && clq_fuse_ok() && b->ls_var <= 1 && b->max_tiles <= ls_tiles_per_launch(); }
    // (spec = false: the two-pass flows — Levenberg-Marquardt, windows with composite factors — whose cost-only candidate evaluation takes the step the same way)
    bool step_fuse_ok() const {
        const DevBatch& D = b->D;
        return b->lat_fuse && !b->no_step_fuse && !b->aux && D.n_win == 1 && b->win[0].x_n <= XCL_MAX && b->max_prior_dim <= PRIOR_LDS_DIM && !D.n_idp
               && D.n_fsb + nb(D.n_sc, 256) + D.n_pch + nb(D.n_imu, IMU_FPB) > 0;
    }
    // k_dogleg + the Jacobian evaluation at its candidate in one grid (k_step_eval); the window's state moves to the other buffer
    void step_eval(bool jac = true) {
        DevBatch& D = b->D;
        Bracket t(*this, jac ? SWF_K_EVAL_PS : SWF_K_POST_DOGLEG);
        Segs S{}; S.e[0] = D.n_fsb; S.e[1] = S.e[0] + nb(D.n_sc, 256); S.e[2] = S.e[1] + D.n_pch;
        const bool imu = D.n_imu > 0;
        S.e[3] = S.e[2] + (imu ? nb(D.n_imu, IMU_FPB) : 0);
        WinState* out = ws_next(D.ws);
        WinState* clr = decide_fused ? ws_next(out) : nullptr;      // the buffer k_decide_lm_clique will write: its failure flags go down here
        if (!jac) {
            if (imu) hipLaunchKernelGGL((k_step_eval<true, false>), dim3(S.e[3]), dim3(256), 0, st, D, O, S, out, b->win[0], clr);
            else hipLaunchKernelGGL((k_step_eval<false, false>), dim3(S.e[2]), dim3(256), 0, st, D, O, S, out, b->win[0], clr);
        } else if (imu) hipLaunchKernelGGL((k_step_eval<true>), dim3(S.e[3]), dim3(256), 0, st, D, O, S, out, b->win[0], clr);
        else hipLaunchKernelGGL((k_step_eval<false>), dim3(S.e[2]), dim3(256), 0, st, D, O, S, out, b->win[0], clr);
        D.ws = out;
    }
    void decide() {
        DevBatch& D = b->D;
        { Bracket t(*this, SWF_K_DECIDE); hipLaunchKernelGGL(k_decide, dim3(D.n_win), dim3(CTL_NT), 0, st, D, O); }
    }
    void cand_eval() {
        DevBatch& D = b->D;
        {
            Bracket t(*this, SWF_K_POST_DOGLEG);
            Segs S{};
            S.e[0] = D.n_fsb; S.e[1] = S.e[0] + nb(D.n_sc, 256);            // (the projection segment: one workgroup per frame-sum block, as in the Jacobian evaluation)
            S.e[2] = S.e[1] + (b->max_prior_dim <= PRIOR_LDS_DIM ? D.n_pch : 0);
            // small batches (latency path): the candidate IMU residuals ride along as a segment; large batches keep them in
            // their own launch (the segment's LDS would cost the memory-bound segments occupancy).  Same results either way.
            bool fuse_imu = D.n_win < b->n_cu;
            S.e[3] = S.e[2] + (fuse_imu ? nb(D.n_imu, IMU_FPB) : 0);
            if (S.e[3] && fuse_imu) hipLaunchKernelGGL((k_post_dogleg<true, 0>), dim3(S.e[3]), dim3(256), 0, st, D, O, S);
            else if (S.e[3]) {
                if (S.e[0]) hipLaunchKernelGGL((k_post_dogleg<false, 1>), dim3(S.e[0]), dim3(256), 0, st, D, O, S);
                if (S.e[3] > S.e[0]) hipLaunchKernelGGL((k_post_dogleg<false, 2>), dim3(S.e[3] - S.e[0]), dim3(256), 0, st, D, O, S);
            }
            if (!fuse_imu && D.n_imu) hipLaunchKernelGGL(k_eval_imu<false>, GRID(D.n_imu, IMU_FPB), dim3(IMU_FPB * IMU_LPF), 0, st, D);
            if (D.n_idp) hipLaunchKernelGGL(k_eval_idp<false>, GRID(D.n_idp, 128), dim3(128), 0, st, D);
        }
        {
            Bracket t(*this, SWF_K_CAND_EVAL);
            if (D.n_prior && b->max_prior_dim > PRIOR_LDS_DIM) hipLaunchKernelGGL(k_eval_prior<false>, dim3(D.n_pch), dim3(256), (2 * b->max_prior_dim + 16) * sizeof(double), st, D);
        }
        decide();
    }
};
}  // namespace

extern "C" int swf_batch_solve(swf_batch* b, const swf_options* opt) {
    DeviceGuard dg_(b ? b->device : -1);
    if (!b || !opt) return fail(SWF_E_INVALID, "swf_batch_solve: bad arguments");
    if (opt->max_num_iterations < 0 || opt->max_num_iterations >= SWF_MAX_TRACE) return fail(SWF_E_INVALID, "max_num_iterations out of range");
    if (opt->trust_region_strategy != SWF_DOGLEG && opt->trust_region_strategy != SWF_LEVENBERG_MARQUARDT) return fail(SWF_E_INVALID, "unknown trust_region_strategy");
    if (opt->composite_root != SWF_ROOT_PIVOTED_CHOLESKY && opt->composite_root != SWF_ROOT_EIGEN) return fail(SWF_E_INVALID, "unknown composite_root");
    b->comp_eigen_root = opt->composite_root == SWF_ROOT_EIGEN;
    if (opt->jacobi_scaling && opt->trust_region_strategy == SWF_DOGLEG && opt->step_mode == SWF_OPTIMIZE)
        return fail(SWF_E_UNSUPPORTED, "jacobi_scaling with the dogleg strategy (the reference sets jacobi_scaling = 0 wherever it selects DOGLEG: R/swf/swf.cpp:26-27)");
    DevBatch& D = b->D;
    Launcher L{ b, to_devopt(opt), b->stream };
    L.export_full = opt->step_mode == SWF_ASSEMBLE_ELIMINATE_ONLY;
    b->L_full = L.export_full || b->min_red > b->rr_nmax || b->max_red > CB_NMAX;
    hipStream_t st = b->stream;
    b->ev_used = 0; b->ev_kind.clear();
    int nlin = 0;
    {
        Launcher::Bracket total(L, SWF_K_TOTAL);
        hipLaunchKernelGGL(k_init, dim3(D.n_win), dim3(256), 0, st, D, L.O);
        auto LIN = [&](int write_S) { L.lin_eval(write_S); L.lin_elim(write_S); nlin++; };
        LIN(1);
        if (opt->step_mode == SWF_ASSEMBLE_ELIMINATE_ONLY) {
            L.reduced();
        } else {
            // Speculative flow (dogleg, no composite factors): the candidate of a proposed step is evaluated WITH its Jacobians — an accepted
            // candidate is the next linearisation point, a rejected dogleg step re-uses the reduced system it came from, an invalid one never
            // gets a candidate (k_dogleg) — so the cost pass at the candidate and the Jacobian pass at the same point that followed it
            // are one pass, and the elimination kernels behind k_decide find the new point's Jacobians in place.  Levenberg-Marquardt
            // re-linearises at the UNCHANGED point after a rejected step (new damping) and keeps the two passes.
            const bool spec = opt->trust_region_strategy == SWF_DOGLEG && !b->n_comp && !b->no_spec;
            L.step_fused = L.step_fuse_ok();
            for (int it = 1; it <= opt->max_num_iterations; it++) {
                L.reduced();
                L.step_rest();
                if (spec) {
                    L.decide_fused = it < opt->max_num_iterations && L.decide_fuse_ok();
                    if (L.step_fused) L.step_eval(); else L.lin_eval(1, true);
                    if (!L.decide_fused) L.decide();
                    if (b->aux) (void)hipEventRecord(b->ev_fork[1], st);          // the auxiliary stream's clique branch starts behind k_decide
                    L.lin_elim(it < opt->max_num_iterations ? 1 : 0); nlin++;
                }
                else {
                    if (L.step_fused) { L.decide_fused = false; L.step_eval(false); L.decide(); } else L.cand_eval();
                    LIN(it < opt->max_num_iterations ? 1 : 0);
                }
            }
        }
        hipLaunchKernelGGL(k_finalize, dim3(D.n_win), dim3(256), 0, st, D, L.O, b->ws_primary);
        D.ws = b->ws_primary;
    }
    HIPCHK(hipGetLastError());
    b->last = swf_timing{};
    b->last.jacobian_bytes = b->jac_bytes; b->last.proj_bytes = b->proj_bytes; b->last.chol_flops = b->chol_flops;
    b->last.lm_schur_flops = b->lm_schur_flops; b->last.n_obs = b->D.n_proj;
    b->last.lm_schur_flops_sym = b->lm_schur_flops_sym; b->last.lm_schur_mfma = b->lm_schur_mfma;
    b->last.n_linearizations = nlin;
    b->last_mode = opt->step_mode; b->mg_valid = false; b->tc_valid = false;      // consumer outputs belong to the previous solve
    return SWF_OK;
}

extern "C" int swf_batch_sync(swf_batch* b) {
    DeviceGuard dg_(b ? b->device : -1);
    if (!b) return fail(SWF_E_INVALID, "null batch");
    HIPCHK(hipStreamSynchronize(b->stream));
    for (int i = 0; i < b->ev_used; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, b->ev[2 * i], b->ev[2 * i + 1]) == hipSuccess) {
            int k = b->ev_kind[i];
            b->last.ms[k] += ms; b->last.calls[k]++;
        }
    }
    b->ev_used = 0; b->ev_kind.clear();
    return SWF_OK;
}

// one call for the whole node: every batch's solve is enqueued on its own device first, then all are awaited
extern "C" int swf_solve_batches(swf_batch* const* batches, int32_t n, const swf_options* opt) {
    if (!batches || n <= 0 || !opt) return fail(SWF_E_INVALID, "swf_solve_batches: bad arguments");
    for (int i = 0; i < n; i++) if (!batches[i]) return fail(SWF_E_INVALID, "swf_solve_batches: null batch");
    // Enqueue: one host thread per DEVICE (a solve is ~90 launches; at 64 windows per GPU one thread enqueueing eight devices in turn
    // would put that host time on the critical path).  Batches that share a device are enqueued by that device's thread, in order.
    std::vector<int> rcs((size_t)n, SWF_OK);
    std::vector<std::string> msgs((size_t)n);
    std::vector<int> devs;
    for (int i = 0; i < n; i++) if (std::find(devs.begin(), devs.end(), batches[i]->device) == devs.end()) devs.push_back(batches[i]->device);
    auto enqueue_device = [&](int dev) {
        for (int i = 0; i < n; i++) {
            if (batches[i]->device != dev) continue;
            rcs[(size_t)i] = swf_batch_solve(batches[i], opt);
            if (rcs[(size_t)i] != SWF_OK) msgs[(size_t)i] = g_err;          // (g_err is thread-local: carried back to the caller below)
        }
    };
    if (devs.size() <= 1) enqueue_device(devs.empty() ? 0 : devs[0]);
    else {
        std::vector<std::thread> th;
        for (size_t d = 1; d < devs.size(); d++) th.emplace_back(enqueue_device, devs[d]);
        enqueue_device(devs[0]);
        for (auto& t : th) t.join();
    }
    // Await EVERY batch that was enqueued, whatever happened to the others: a caller that reads states or destroys windows after an
    // error return must not race work still in flight.  The first error is the one reported.
    int rc_all = SWF_OK; std::string msg;
    for (int i = 0; i < n; i++) if (rcs[(size_t)i] != SWF_OK && rc_all == SWF_OK) { rc_all = rcs[(size_t)i]; msg = msgs[(size_t)i]; }
    for (int i = 0; i < n; i++) {
        // (also a batch whose enqueue FAILED: swf_batch_solve can fail at a launch check with dozens of kernels already on its stream)
        if (!batches[i]) continue;
        int rc = swf_batch_sync(batches[i]);
        if (rc != SWF_OK && rc_all == SWF_OK) { rc_all = rc; msg = g_err; }
    }
    if (rc_all != SWF_OK) g_err = msg;
    return rc_all;
}

extern "C" int swf_batch_enable_timing(swf_batch* b, int32_t mask) { if (!b) return fail(SWF_E_INVALID, "null batch"); b->timing = mask; return SWF_OK; }
extern "C" int swf_batch_timing(swf_batch* b, swf_timing* out) { if (!b || !out) return fail(SWF_E_INVALID, "bad arguments"); *out = b->last; return SWF_OK; }

extern "C" int swf_batch_download_state(swf_batch* b) {
    DeviceGuard dg_(b ? b->device : -1);
    if (!b) return fail(SWF_E_INVALID, "null batch");
    const size_t nx = (size_t)b->D.n_x;
    if (!b->h_x && nx) HIPCHK(hipHostMalloc((void**)&b->h_x, nx * sizeof(double), hipHostMallocDefault));
    HIPCHK(hipMemcpyAsync(b->h_x, b->D.x, nx * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    for (size_t i = 0; i < b->win.size(); i++) {
        const HostWin& h = b->hw[i];
        const double* p = b->h_x + b->win[i].x_base;
        memcpy(h.pose, p, sizeof(double) * 7 * h.n_pose); p += 7 * h.n_pose;
        memcpy(h.sb, p, sizeof(double) * 9 * h.n_sb); p += 9 * h.n_sb;
        memcpy(h.lm, p, sizeof(double) * 3 * h.n_lm); p += 3 * h.n_lm;
        memcpy(h.sc, p, sizeof(double) * h.n_sc);
    }
    if (b->n_comp) {       // the hidden GNSS epochs are parameter memory too
        std::vector<double> hp((size_t)b->comp_ne * 7), hs((size_t)b->comp_ne * 9);
        HIPCHK(hipMemcpy(hp.data(), b->CA.pose, hp.size() * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hs.data(), b->CA.sb, hs.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < b->win.size(); i++) {
            const HostWin& h = b->hw[i];
            if (!h.comp_ne) continue;
            memcpy(h.comp_pose, hp.data() + (size_t)h.comp_e0 * 7, sizeof(double) * 7 * h.comp_ne);
            memcpy(h.comp_sb, hs.data() + (size_t)h.comp_e0 * 9, sizeof(double) * 9 * h.comp_ne);
        }
    }
    return SWF_OK;
}

extern "C" int swf_batch_summaries(swf_batch* b, swf_summary* out) {
    DeviceGuard dg_(b ? b->device : -1);
    if (!b || !out) return fail(SWF_E_INVALID, "bad arguments");
    size_t n = b->win.size();
    // (page-locked staging kept by the batch: the traces are 2.4 MB for 512 windows)
    const size_t ws_b = (n * sizeof(WinState) + 63) & ~(size_t)63, tr_b = n * SWF_MAX_TRACE * sizeof(swf_iteration);
    if (b->h_sum_bytes < ws_b + tr_b) {
        if (b->h_sum) (void)hipHostFree(b->h_sum);
        b->h_sum = nullptr; b->h_sum_bytes = 0;
        HIPCHK(hipHostMalloc(&b->h_sum, ws_b + tr_b, hipHostMallocDefault));
        b->h_sum_bytes = ws_b + tr_b;
    }
    WinState* ws = (WinState*)b->h_sum;
    swf_iteration* tr = (swf_iteration*)((char*)b->h_sum + ws_b);
    HIPCHK(hipMemcpyAsync(ws, b->D.ws, n * sizeof(WinState), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(tr, b->D.trace, tr_b, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    for (size_t i = 0; i < n; i++) {
        swf_summary& s = out[i];
        memset(&s, 0, sizeof(s));
        s.initial_cost = ws[i].initial_cost; s.final_cost = ws[i].x_cost;
        s.minimizer_time_in_seconds = b->last.ms[SWF_K_TOTAL] * 1e-3;
        s.num_successful_steps = ws[i].nsucc; s.num_unsuccessful_steps = ws[i].nunsucc;
        s.num_iterations = ws[i].iter; s.termination = ws[i].status;
        s.reduced_dim = b->win[i].n_red; s.tail_dim = b->hw[i].tail_dim;
        memcpy(s.trace, tr + i * SWF_MAX_TRACE, sizeof(swf_iteration) * SWF_MAX_TRACE);
    }
    return SWF_OK;
}

extern "C" int swf_batch_dims(swf_batch* b, int32_t w, int32_t* n_loc, int32_t* n_e, int32_t* n_red) {
    if (!b || w < 0 || w >= (int)b->win.size()) return fail(SWF_E_INVALID, "bad window index");
    if (n_loc) *n_loc = b->win[w].n_loc; if (n_e) *n_e = b->win[w].n_e; if (n_red) *n_red = b->win[w].n_red;
    return SWF_OK;
}

extern "C" int swf_batch_export_reduced(swf_batch* b, int32_t w, double* S, double* rhs, double* L) {
    DeviceGuard dg_(b ? b->device : -1);
    if (!b || w < 0 || w >= (int)b->win.size()) return fail(SWF_E_INVALID, "bad window index");
    if (b->last_mode < 0) return fail(SWF_E_STATE, "export before any solve");
    const WinRec& W = b->win[w];
    size_t n = (size_t)W.n_red;
    HIPCHK(hipStreamSynchronize(b->stream));
    if (S) {
        HIPCHK(hipMemcpy(S, b->D.S + W.S_base, n * n * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t r = 0; r < n; r++) for (size_t c = r + 1; c < n; c++) S[r * n + c] = S[c * n + r];   // device keeps the lower triangle
    }
    if (rhs) HIPCHK(hipMemcpy(rhs, b->D.rhs + W.loc_base + W.n_e, n * sizeof(double), hipMemcpyDeviceToHost));
    if (L) {
        std::vector<double> Lt((n + 1) * (n + 1));
        HIPCHK(hipMemcpy(Lt.data(), b->D.L + W.Lt_base, Lt.size() * sizeof(double), hipMemcpyDeviceToHost));
        bool rr = b->max_red <= CB_NMAX;      // k_chol_rr4 / k_chol_big write row-major lower, ld = n
        // k_chol_rr4 on a solve path keeps the factor in registers and writes only the block its readers use: the parameter_head
        // tail (from the 16-aligned row / column at or before its start).  Everything else is returned as zero.
        size_t first = 0;
        if (!b->L_full && n <= (size_t)b->rr_nmax) { const size_t td = (size_t)b->hw[w].tail_dim; first = td ? ((n - td) >> 4) << 4 : n; }
        for (size_t r = 0; r < n; r++) for (size_t c = 0; c < n; c++)
            L[r * n + c] = (c <= r && c >= first) ? (rr ? Lt[r * n + c] : Lt[c * (n + 1) + r]) : 0.0;
    }
    return SWF_OK;
}

extern "C" int swf_batch_marginalize(swf_batch* b, double eps, int32_t form) {
    DeviceGuard dg_(b ? b->device : -1);
    if (!b || (form != SWF_PRIOR_EIGEN && form != SWF_PRIOR_CHOLESKY) || !(eps >= 0.0)) return fail(SWF_E_INVALID, "swf_batch_marginalize: bad arguments");
    if (b->last_mode != SWF_ASSEMBLE_ELIMINATE_ONLY) return fail(SWF_E_STATE, "swf_batch_marginalize needs a preceding solve with step_mode = SWF_ASSEMBLE_ELIMINATE_ONLY");
    if (b->max_red > CB_NMAX) return fail(SWF_E_UNSUPPORTED, "marginalisation needs the row-major Cholesky factor (n_red <= 640)");
    int nw = (int)b->win.size(), ldn = 1;
    for (int w = 0; w < nw; w++) ldn = std::max(ldn, b->hw[w].tail_dim);
    if (form == SWF_PRIOR_EIGEN && ldn > MG_BIGN) return fail(SWF_E_UNSUPPORTED, "eigen square root: parameter_head tail larger than 640 dimensions (use SWF_PRIOR_CHOLESKY)");
    b->mg_ld = ldn;
    if (!b->mg_A) {
        std::vector<int> td(nw);
        for (int w = 0; w < nw; w++) td[w] = b->hw[w].tail_dim;
        const int* tdp = nullptr;
        int rc = b->pool.put(td, &tdp);
        b->mg_tail = (int*)tdp;
        rc |= b->pool.zeros((size_t)nw * ldn * ldn, &b->mg_A); rc |= b->pool.zeros((size_t)nw * ldn * ldn, &b->mg_J);
        rc |= b->pool.zeros((size_t)nw * ldn, &b->mg_b); rc |= b->pool.zeros((size_t)nw * ldn, &b->mg_r0); rc |= b->pool.zeros((size_t)nw * ldn, &b->mg_w);
        rc |= b->pool.zeros((size_t)nw, &b->mg_rank);
        rc |= b->pool.zeros((size_t)nw * ldn * ldn, &b->mg_resM); rc |= b->pool.zeros((size_t)nw * ldn, &b->mg_resb); rc |= b->pool.zeros((size_t)nw, &b->mg_resok);
        if (rc) return fail(SWF_E_NODEVICE, "device allocation failed");
    }
    const bool big = form == SWF_PRIOR_EIGEN && ldn > MG_MAXN;
    if (big && !b->mg_M && b->pool.zeros((size_t)nw * ldn * ldn, &b->mg_M)) return fail(SWF_E_NODEVICE, "device allocation failed");
    // windows whose factorisation broke down in the tail (a marginal that is singular on the kept states): partial factorisation +
    // rank-revealing factor of A; every other window leaves this kernel at once
    // rows of V the pivoted Cholesky (d_pivoted_chol) takes per pass over its pool: what 150 KB of LDS leave next to 25 rows of the tail
    const int mg_rc = std::max(16, std::min(ldn, (int)((150 * 1024 / 8 - (RS_POOL + 1) * (long)ldn) / RS_POOL)));
    const int force = getenv("SWF_FORCE_MARG_RESCUE") ? 1 : 0;      // (read once per call, outside the sweep loop: the tests toggle it between two calls on one batch)      // testing aid: healthy windows through the rank-deficient path too
    if (form == SWF_PRIOR_EIGEN)
        {
            // dynamic LDS: the 16-column panel over (n_red + 1) rows, later the pivoted Cholesky's pool (24 rows of the tail), its diagonal
            // and the staging of 24 x rc entries of V^T (rc = rows of V per pass: all of them up to 400 dimensions)
            const size_t l1 = (size_t)(b->max_red + 1) * (RS_NB + 1), l2 = (size_t)(RS_POOL + 1) * (size_t)ldn + (size_t)RS_POOL * (size_t)mg_rc;
            const size_t lds = sizeof(double) * std::max(l1, l2);
            if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_marg_rescue, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      // (up to 150 KB at 640 dimensions)
            hipLaunchKernelGGL(k_marg_rescue, dim3(nw), dim3(1024), lds, b->stream, b->D, (const int*)b->mg_tail, ldn, b->mg_resM, b->mg_resb, b->mg_resok, force, eps, b->mg_J, mg_rc);
        }
    hipLaunchKernelGGL(k_marginalize<false>, dim3(nw), dim3(MG_NT), 0, b->stream, b->D, (const int*)b->mg_tail, eps, (int)form, ldn,
                       b->mg_A, b->mg_b, b->mg_J, b->mg_r0, b->mg_w, b->mg_rank, (double*)nullptr,
                       (const double*)b->mg_resM, (const double*)b->mg_resb, (const int*)b->mg_resok, force, 0, (int*)nullptr);
    if (form == SWF_PRIOR_CHOLESKY)      // A = J^T J, one thread per entry (full-length sums: the zeros of the triangular J add exact zeros)
        hipLaunchKernelGGL(k_marg_gram, dim3((ldn * ldn + 255) / 256, nw), dim3(256), 0, b->stream, (const int*)b->mg_tail, ldn, (const double*)b->mg_J, b->mg_A, (const int*)nullptr);
    if (big) {
        // large tails: set-up, the block-Jacobi sweeps over many workgroups (fixed launch schedule; converged windows return at once), write-out
        if (!b->mg_rot && (b->pool.zeros((size_t)nw * MG_SWEEPS, &b->mg_rot) || b->pool.zeros((size_t)nw, &b->mg_bjok) || b->pool.zeros((size_t)nw * 2, &b->mg_crit))) return fail(SWF_E_NODEVICE, "device allocation failed");
        HIPCHK(hipMemsetAsync(b->mg_rot, 0, (size_t)nw * MG_SWEEPS * sizeof(int), b->stream));
        HIPCHK(hipMemsetAsync(b->mg_crit, 0, (size_t)nw * 2 * sizeof(unsigned long long), b->stream));
        auto phase = [&](int ph) {
            hipLaunchKernelGGL(k_marginalize<true>, dim3(nw), dim3(MG_NT), 0, b->stream, b->D, (const int*)b->mg_tail, eps, (int)form, ldn,
                               b->mg_A, b->mg_b, b->mg_J, b->mg_r0, b->mg_w, b->mg_rank, b->mg_M,
                               (const double*)b->mg_resM, (const double*)b->mg_resb, (const int*)b->mg_resok, force, ph, b->mg_bjok);
        };
        {
            phase(1);
            hipLaunchKernelGGL(k_marg_gram, dim3((ldn * ldn + 255) / 256, nw), dim3(256), 0, b->stream, (const int*)b->mg_tail, ldn, (const double*)b->mg_M, b->mg_A, (const int*)b->mg_bjok);
            {
                // the Jacobi preconditioner: pivoted Cholesky of A into the G slab (9 sweeps instead of 16 at the 263-dimension tail)
                const size_t lds = sizeof(double) * ((size_t)(RS_POOL + 1) * (size_t)ldn + (size_t)RS_POOL * (size_t)mg_rc);
                if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_marg_pchol, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(k_marg_pchol, dim3(nw), dim3(1024), lds, b->stream, (const int*)b->mg_tail, ldn, (const double*)b->mg_A, b->mg_resM, b->mg_M, (const int*)b->mg_bjok, mg_rc, eps);
            }
            // block size by window: 8 columns up to 576 dimensions (17 workgroups per launch at 263 dimensions, 8 inner steps each; 16-column
            // blocks halve the launches but leave a step to 9 workgroups whose 16 waves share 4 SIMDs: 49 us per launch against 16), 4 above.
            // One launch schedule per class, sized by the class's largest tail; every window follows its own round-robin inside it.
            int ldA = 0, ldB = 0;
            for (int w = 0; w < nw; w++) { const int t = b->hw[w].tail_dim; if (t > MG_MAXN && t <= 576) ldA = std::max(ldA, t); else if (t > 576) ldB = std::max(ldB, t); }
            std::vector<int> hrot((size_t)nw * MG_SWEEPS);
            // from the sixth sweep on, one look at the rotation counts per sweep: 30 us of synchronisation against the 34 launches of a
            // 263-dimension sweep (0.55 ms) that a check every fourth sweep ran up to three times too often
            const int mg_first_check = 6;
            for (int sweep = 0; sweep < MG_SWEEPS; sweep++) {
                if (sweep >= mg_first_check) {
                    // every four sweeps: has every window reported a sweep without rotations?  (the launches of a converged window return
                    // at once, but a 263-dimension sweep is still 34 launches)
                    HIPCHK(hipMemcpyAsync(hrot.data(), b->mg_rot, hrot.size() * sizeof(int), hipMemcpyDeviceToHost, b->stream));
                    HIPCHK(hipStreamSynchronize(b->stream));
                    bool all = true;
                    for (int w = 0; w < nw && all; w++) { bool done = false; for (int k = 0; k < sweep; k++) done = done || hrot[(size_t)w * MG_SWEEPS + k] == 0; all = done; }
                    if (all) break;
                }
#define BJ_LAUNCH(BS_, LDM_, NR_) hipLaunchKernelGGL((k_marg_bj<BS_, LDM_, NR_>), grid, dim3(1024), 0, b->stream, (const int*)b->mg_tail, ldn, b->mg_M, b->mg_rot, b->mg_crit, (const int*)b->mg_bjok, sweep, st)
                if (ldA) {
                    const int bs = 8, nbe = ((ldA + bs - 1) / bs + 1) & ~1;
                    for (int st = -1; st < nbe - 1; st++) {
                        dim3 grid(nbe / 2, nw);
                        if (ldA <= 320) BJ_LAUNCH(8, 576, 5);
                        else BJ_LAUNCH(8, 576, 9);
                    }
                }
                if (ldB) {
                    const int nbe = ((ldB + 3) / 4 + 1) & ~1;
                    for (int st = -1; st < nbe - 1; st++) { dim3 grid(nbe / 2, nw); BJ_LAUNCH(4, 640, 10); }
                }
#undef BJ_LAUNCH
                hipLaunchKernelGGL(k_marg_bj_crit, dim3((nw + 63) / 64), dim3(64), 0, b->stream, (const int*)b->mg_tail, nw, b->mg_rot, b->mg_crit, (const int*)b->mg_bjok, sweep);
            }
            phase(2);
        }
    }
    HIPCHK(hipGetLastError());
    b->mg_valid = true; b->mg_host = false;
    return SWF_OK;
}

// ambiguity covariance hand-off: information and covariance of the parameter_head tail from the factor of the last linear solve
extern "C" int swf_batch_tail_covariance(swf_batch* b) {
    DeviceGuard dg_(b ? b->device : -1);
    if (!b) return fail(SWF_E_INVALID, "swf_batch_tail_covariance: null batch");
    if (b->last_mode < 0) return fail(SWF_E_STATE, "swf_batch_tail_covariance needs a preceding solve");
    if (b->max_red > CB_NMAX) return fail(SWF_E_UNSUPPORTED, "the tail covariance needs the row-major Cholesky factor (n_red <= 640)");
    int nw = (int)b->win.size(), ldn = 1;
    for (int w = 0; w < nw; w++) ldn = std::max(ldn, b->hw[w].tail_dim);
    if (!b->tc_A) {
        std::vector<int> td(nw);
        for (int w = 0; w < nw; w++) td[w] = b->hw[w].tail_dim;
        const int* tdp = nullptr;
        int rc = b->pool.put(td, &tdp);
        b->tc_tail = (int*)tdp;
        rc |= b->pool.zeros((size_t)nw * ldn * ldn, &b->tc_A); rc |= b->pool.zeros((size_t)nw * ldn * ldn, &b->tc_Q);
        rc |= b->pool.zeros((size_t)nw * ldn * ldn, &b->tc_X); rc |= b->pool.zeros((size_t)nw, &b->tc_rank);
        if (rc) return fail(SWF_E_NODEVICE, "device allocation failed");
    }
    b->tc_ld = ldn;
    hipLaunchKernelGGL(k_tail_cov, dim3(nw), dim3(256), 0, b->stream, b->D, (const int*)b->tc_tail, ldn, b->tc_A, b->tc_Q, b->tc_X, b->tc_rank);
    HIPCHK(hipGetLastError());
    b->tc_valid = true;
    return SWF_OK;
}

extern "C" int swf_batch_get_tail_covariance(swf_batch* b, int32_t w, double* A, double* Qy, int32_t* n_out) {
    DeviceGuard dg_(b ? b->device : -1);
    if (!b || w < 0 || w >= (int)b->win.size()) return fail(SWF_E_INVALID, "bad window index");
    if (!b->tc_valid) return fail(SWF_E_STATE, "swf_batch_get_tail_covariance before swf_batch_tail_covariance");
    HIPCHK(hipStreamSynchronize(b->stream));
    size_t n = (size_t)b->hw[w].tail_dim, o2 = (size_t)w * b->tc_ld * b->tc_ld;
    int32_t rk = 0;
    HIPCHK(hipMemcpy(&rk, b->tc_rank + w, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (n_out) *n_out = rk < 0 ? -1 : (int32_t)n;
    if (rk < 0) return fail(SWF_E_STATE, "tail covariance: the window has no valid factor (failed linear solve or empty tail)");
    if (A) HIPCHK(hipMemcpy(A, b->tc_A + o2, n * n * sizeof(double), hipMemcpyDeviceToHost));
    if (Qy) HIPCHK(hipMemcpy(Qy, b->tc_Q + o2, n * n * sizeof(double), hipMemcpyDeviceToHost));
    return SWF_OK;
}

extern "C" int swf_batch_get_prior(swf_batch* b, int32_t w, double* A, double* bv, double* J, double* r0, double* eig, int32_t* n_out, int32_t* rank) {
    DeviceGuard dg_(b ? b->device : -1);
    if (!b || w < 0 || w >= (int)b->win.size()) return fail(SWF_E_INVALID, "bad window index");
    if (!b->mg_valid) return fail(SWF_E_STATE, "swf_batch_get_prior before swf_batch_marginalize");
    HIPCHK(hipStreamSynchronize(b->stream));
    size_t n = (size_t)b->hw[w].tail_dim, o2 = (size_t)w * b->mg_ld * b->mg_ld, o1 = (size_t)w * b->mg_ld;
    if (n_out) *n_out = (int32_t)n;
    const size_t nw = b->win.size(), tot2 = nw * b->mg_ld * b->mg_ld, tot1 = nw * b->mg_ld;
    if (nw >= 8 && tot2 <= ((size_t)1 << 23)) {
        if (!b->mg_host) {
            b->h_mgA.resize(tot2); b->h_mgJ.resize(tot2); b->h_mgb.resize(tot1); b->h_mgr0.resize(tot1); b->h_mgw.resize(tot1); b->h_mgrank.resize(nw);
            HIPCHK(hipMemcpy(b->h_mgA.data(), b->mg_A, tot2 * sizeof(double), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(b->h_mgJ.data(), b->mg_J, tot2 * sizeof(double), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(b->h_mgb.data(), b->mg_b, tot1 * sizeof(double), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(b->h_mgr0.data(), b->mg_r0, tot1 * sizeof(double), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(b->h_mgw.data(), b->mg_w, tot1 * sizeof(double), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(b->h_mgrank.data(), b->mg_rank, nw * sizeof(int), hipMemcpyDeviceToHost));
            b->mg_host = true;
        }
        if (A) memcpy(A, b->h_mgA.data() + o2, n * n * sizeof(double));
        if (J) memcpy(J, b->h_mgJ.data() + o2, n * n * sizeof(double));
        if (bv) memcpy(bv, b->h_mgb.data() + o1, n * sizeof(double));
        if (r0) memcpy(r0, b->h_mgr0.data() + o1, n * sizeof(double));
        if (eig) memcpy(eig, b->h_mgw.data() + o1, n * sizeof(double));
        if (rank) *rank = b->h_mgrank[(size_t)w];
        return SWF_OK;
    }
    if (A) HIPCHK(hipMemcpy(A, b->mg_A + o2, n * n * sizeof(double), hipMemcpyDeviceToHost));
    if (J) HIPCHK(hipMemcpy(J, b->mg_J + o2, n * n * sizeof(double), hipMemcpyDeviceToHost));
    if (bv) HIPCHK(hipMemcpy(bv, b->mg_b + o1, n * sizeof(double), hipMemcpyDeviceToHost));
    if (r0) HIPCHK(hipMemcpy(r0, b->mg_r0 + o1, n * sizeof(double), hipMemcpyDeviceToHost));
    if (eig) HIPCHK(hipMemcpy(eig, b->mg_w + o1, n * sizeof(double), hipMemcpyDeviceToHost));
    if (rank) HIPCHK(hipMemcpy(rank, b->mg_rank + w, sizeof(int32_t), hipMemcpyDeviceToHost));
    return SWF_OK;
}

extern "C" int swf_batch_export_vectors(swf_batch* b, int32_t w, double* grad, double* diag, double* y) {
    DeviceGuard dg_(b ? b->device : -1);
    if (!b || w < 0 || w >= (int)b->win.size()) return fail(SWF_E_INVALID, "bad window index");
    const WinRec& W = b->win[w];
    size_t n = (size_t)W.n_loc;
    HIPCHK(hipStreamSynchronize(b->stream));
    if (grad) HIPCHK(hipMemcpy(grad, b->D.g + W.loc_base, n * sizeof(double), hipMemcpyDeviceToHost));
    if (diag) HIPCHK(hipMemcpy(diag, b->D.diag + W.loc_base, n * sizeof(double), hipMemcpyDeviceToHost));
    if (y) HIPCHK(hipMemcpy(y, b->D.y + W.loc_base, n * sizeof(double), hipMemcpyDeviceToHost));
    return SWF_OK;
}

// Debug / parity export: residual vector and dense Jacobian of window w as the device holds them after its last
// linearisation (see include/swf_solver.h).  Host-side gather of the device buffers; nothing here is on the solve path.
extern "C" int swf_batch_export_jacobian(swf_batch* b, int32_t w, double* r, double* J, int32_t* n_res_out, int32_t* n_loc_out) {
    DeviceGuard dg_(b ? b->device : -1);
    if (!b || w < 0 || w >= (int)b->win.size()) return fail(SWF_E_INVALID, "bad window index");
    const WinRec& W = b->win[w];
    const DevBatch& D = b->D;
    const int ngf = W.gf1 - W.gf0, nobs = W.proj1 - W.proj0;
    std::vector<GFac> gf((size_t)std::max(ngf, 0));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (ngf > 0) HIPCHK(hipMemcpy(gf.data(), D.gf + W.gf0, (size_t)ngf * sizeof(GFac), hipMemcpyDeviceToHost));
    const int nproj_all = b->hw[w].n_proj_all;
    int nres = 2 * nproj_all;
    for (const GFac& G : gf) if (G.type != GF_PROJX) nres += G.nres;
    if (n_res_out) *n_res_out = nres;
    if (n_loc_out) *n_loc_out = W.n_loc;
    if (!r && !J) return SWF_OK;
    if (b->last_mode < 0) return fail(SWF_E_STATE, "swf_batch_export_jacobian before any solve");
    const size_t nl = (size_t)W.n_loc, N = (size_t)D.n_proj;
    if (J) memset(J, 0, (size_t)nres * nl * sizeof(double));
    auto dl = [&](void* dst, const void* src, size_t bytes) { return bytes == 0 || hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess; };
    bool ok = true;
    if (nobs > 0) {
        std::vector<double> pr(2 * (size_t)nobs), Jp(12 * (size_t)nobs), Jl(6 * (size_t)nobs);
        std::vector<int> lp((size_t)nobs), ll((size_t)nobs);
        for (int k = 0; k < 2; k++) ok &= dl(pr.data() + (size_t)k * nobs, D.p_r + k * N + W.proj0, (size_t)nobs * 8);
        for (int k = 0; k < 12; k++) ok &= dl(Jp.data() + (size_t)k * nobs, D.p_Jp + k * N + W.proj0, (size_t)nobs * 8);
        for (int k = 0; k < 6; k++) ok &= dl(Jl.data() + (size_t)k * nobs, D.p_Jl + k * N + W.proj0, (size_t)nobs * 8);
        ok &= dl(lp.data(), D.p_lpose + W.proj0, (size_t)nobs * 4); ok &= dl(ll.data(), D.p_llm + W.proj0, (size_t)nobs * 4);
        if (!ok) return fail(SWF_E_NODEVICE, "download failed");
        for (int q = 0; q < nobs; q++) {
            const size_t row = 2 * (size_t)b->hw[w].p_orig[q];
            for (int a = 0; a < 2; a++) {
                if (r) r[row + a] = pr[(size_t)a * nobs + q];
                if (!J) continue;
                // (the translation half of Jp is not stored next to a variable landmark: it is -Jl)
                if (lp[q] >= 0) for (int c = 0; c < 6; c++) J[(row + a) * nl + (lp[q] - W.loc_base) + c] = (c < 3 && ll[q] >= 0) ? -Jl[(size_t)(a * 3 + c) * nobs + q] : Jp[(size_t)(a * 6 + c) * nobs + q];
                if (ll[q] >= 0) for (int c = 0; c < 3; c++) J[(row + a) * nl + (ll[q] - W.loc_base) + c] = Jl[(size_t)(a * 3 + c) * nobs + q];
            }
        }
    }
    size_t row_seq = 2 * (size_t)nproj_all;
    for (const GFac& G : gf) {
        // generic-path projection factors sit at the caller's projection index; every other family follows in factor order
        const size_t row = G.type == GF_PROJX ? 2 * (size_t)G.pad : row_seq;
        std::vector<int> sloc((size_t)G.nslot), sls((size_t)G.nslot), sj((size_t)G.nslot), spc((size_t)G.nslot);
        ok &= dl(sloc.data(), D.s_loc + G.slot0, (size_t)G.nslot * 4); ok &= dl(sls.data(), D.s_ls + G.slot0, (size_t)G.nslot * 4);
        ok &= dl(sj.data(), D.s_joff + G.slot0, (size_t)G.nslot * 4); ok &= dl(spc.data(), D.s_pcol + G.slot0, (size_t)G.nslot * 4);
        if (r) ok &= dl(r + row, D.g_r + G.roff, (size_t)G.nres * 8);
        if (J && G.type == GF_PRIOR) {
            // linearised prior (and composite factors, whose record k_comp_scatter rewrites): the constant row-major J of the record
            int dim = 0; long long jo = 0;
            ok &= dl(&dim, D.prior_dim + G.data, 4); ok &= dl(&jo, D.prior_Joff + G.data, 8);
            std::vector<double> PJ((size_t)dim * dim);
            ok &= dl(PJ.data(), D.prior_J + jo, PJ.size() * 8);
            for (int sl = 0; sl < G.nslot; sl++) {
                if (sloc[sl] < 0) continue;
                for (int k = 0; k < G.nres; k++) for (int c = 0; c < sls[sl]; c++)
                    J[(row + k) * nl + (sloc[sl] - W.loc_base) + c] = PJ[(size_t)k * dim + spc[sl] + c];
            }
        } else if (J) {
            for (int sl = 0; sl < G.nslot; sl++) {
                if (sloc[sl] < 0 || sj[sl] < 0) continue;
                std::vector<double> blk((size_t)sls[sl] * G.jld);
                ok &= dl(blk.data(), D.g_J + sj[sl], (((size_t)sls[sl] - 1) * G.jld + G.nres) * 8);
                for (int k = 0; k < G.nres; k++) for (int c = 0; c < sls[sl]; c++)
                    J[(row + k) * nl + (sloc[sl] - W.loc_base) + c] = blk[(size_t)c * G.jld + k];
            }
        }
        if (G.type != GF_PROJX) row_seq += G.nres;
    }
    if (!ok) return fail(SWF_E_NODEVICE, "download failed");
    return SWF_OK;
}

#ifdef SWF_PROFILE_CHOL
extern "C" int swf_debug_chol_stamps(unsigned long long* out) {
    if (hipDeviceSynchronize() != hipSuccess) return SWF_E_NODEVICE;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chol_stamps), 64 * sizeof(unsigned long long)) != hipSuccess) return SWF_E_NODEVICE;
    return SWF_OK;
}
#endif

#ifdef SWF_PROFILE_CHOLW
extern "C" int swf_debug_chol_wstep(int step) {
    if (hipDeviceSynchronize() != hipSuccess) return SWF_E_NODEVICE;
    unsigned long long z[16 * 8] = { 0 };
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_chol_wst), z, sizeof(z)) != hipSuccess) return SWF_E_NODEVICE;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_chol_wstep), &step, sizeof(int)) != hipSuccess) return SWF_E_NODEVICE;
    return SWF_OK;
}
extern "C" int swf_debug_chol_wstamps(unsigned long long* out) {
    if (hipDeviceSynchronize() != hipSuccess) return SWF_E_NODEVICE;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chol_wst), 16 * 8 * sizeof(unsigned long long)) != hipSuccess) return SWF_E_NODEVICE;
    return SWF_OK;
}
extern "C" int swf_debug_chol_pstamps(unsigned long long* out) {
    if (hipDeviceSynchronize() != hipSuccess) return SWF_E_NODEVICE;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chol_pst), 16 * sizeof(unsigned long long)) != hipSuccess) return SWF_E_NODEVICE;
    return SWF_OK;
}
extern "C" int swf_debug_chol_cstamps(unsigned long long* out) {
    if (hipDeviceSynchronize() != hipSuccess) return SWF_E_NODEVICE;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chol_cst), 32 * sizeof(unsigned long long)) != hipSuccess) return SWF_E_NODEVICE;
    return SWF_OK;
}
#endif

#ifdef SWF_PROFILE_DOG
extern "C" int swf_debug_dog_stamps(unsigned long long* out) {
    if (hipDeviceSynchronize() != hipSuccess) return SWF_E_NODEVICE;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dog_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return SWF_E_NODEVICE;
    return SWF_OK;
}
#endif

#ifdef SWF_PROFILE_CLQ
extern "C" int swf_debug_clq_stamps(unsigned long long* out) {
    if (hipDeviceSynchronize() != hipSuccess) return SWF_E_NODEVICE;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clq_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return SWF_E_NODEVICE;
    return SWF_OK;
}
#endif

#ifdef SWF_PROFILE_GEMM
extern "C" int swf_debug_gemm_stamps(unsigned long long* out) {
    if (hipDeviceSynchronize() != hipSuccess) return SWF_E_NODEVICE;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return SWF_E_NODEVICE;
    return SWF_OK;
}
#endif


// =====================================================================================================================
// Composite IMU-GNSS factors as a batched operator (include/swf_solver.h, swf_kernels4.h).  Stateful like the reference's
// IMUGNSSBase: the handle owns the hidden epochs, the saved elimination blocks and the last linearisation.
// =====================================================================================================================
struct swf_composite {
    CompArgs A{};
    std::vector<void*> bufs;
    int n = 0, nmax = 0, nmin = 0; long long sumM = 0, sumN = 0, sumG = 0, sumG2 = 0;
    std::vector<int> M;
    bool eigen_root = false;
    hipStream_t stream = nullptr;
    ~swf_composite() { for (void* p : bufs) (void)hipFree(p); }
};

extern "C" int swf_composite_destroy(swf_composite* c) { delete c; return SWF_OK; }

extern "C" int swf_composite_create(int32_t n, const int32_t* M, const int32_t* N, const double* pose, const double* sb,
                                    const double* pose_lin, const double* sb_lin, const double* Hpp, const double* HpN,
                                    const double* rhs_p, const double* HNN, const double* rhsN, const double* pre,
                                    const double pbg[3], const double gw[3], void* stream, swf_composite** out) {
    if (!out || n <= 0 || !M || !N || !pose || !sb || !pose_lin || !sb_lin || !Hpp || !HpN || !rhs_p || !HNN || !rhsN || !pre || !pbg || !gw)
        return fail(SWF_E_INVALID, "swf_composite_create: null argument");
    std::vector<int> eo(n + 1, 0), no(n + 1, 0);
    std::vector<long long> pno(n + 1, 0), nno(n + 1, 0), go(n + 1, 0), g2o(n + 1, 0);
    int nmax_ = 0, nmin_ = 1 << 30;
    for (int f = 0; f < n; f++) {
        if (M[f] < 1) return fail(SWF_E_INVALID, "composite factor without hidden epochs");
        if (N[f] < 0 || N[f] > CO_MAXN) return fail(SWF_E_UNSUPPORTED, "composite factor with more than 64 ambiguities");
        nmax_ = std::max(nmax_, (int)N[f]); nmin_ = std::min(nmin_, (int)N[f]);
        eo[f + 1] = eo[f] + M[f]; no[f + 1] = no[f] + N[f];
        pno[f + 1] = pno[f] + 15LL * M[f] * N[f]; nno[f + 1] = nno[f] + (long long)N[f] * N[f];
        go[f + 1] = go[f] + 30 + N[f]; g2o[f + 1] = g2o[f] + (long long)(30 + N[f]) * (30 + N[f]);
    }
    std::unique_ptr<swf_composite> c(new swf_composite());
    c->n = n; c->nmax = nmax_; c->nmin = nmin_; c->sumM = eo[n]; c->sumN = no[n]; c->sumG = go[n]; c->sumG2 = g2o[n]; c->stream = (hipStream_t)stream;
    bool bad = false;
    auto up = [&](const void* src, size_t bytes) -> void* {
        void* d = nullptr;
        if (hipMalloc(&d, std::max<size_t>(bytes, 8)) != hipSuccess) { bad = true; return nullptr; }
        c->bufs.push_back(d);
        if (src) { if (hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) bad = true; }
        else if (hipMemset(d, 0, std::max<size_t>(bytes, 8)) != hipSuccess) bad = true;
        return d;
    };
    CompArgs& A = c->A;
    const size_t D = sizeof(double);
    A.n = n;
    A.M = (const int*)up(M, n * sizeof(int)); A.N = (const int*)up(N, n * sizeof(int));
    A.e_off = (const int*)up(eo.data(), (n + 1) * sizeof(int)); A.n_off = (const int*)up(no.data(), (n + 1) * sizeof(int));
    A.pn_off = (const long long*)up(pno.data(), (n + 1) * sizeof(long long)); A.nn_off = (const long long*)up(nno.data(), (n + 1) * sizeof(long long));
    A.g_off = (const long long*)up(go.data(), (n + 1) * sizeof(long long)); A.g2_off = (const long long*)up(g2o.data(), (n + 1) * sizeof(long long));
    A.pose = (double*)up(pose, c->sumM * 7 * D); A.sb = (double*)up(sb, c->sumM * 9 * D);
    A.pose_lin = (const double*)up(pose_lin, c->sumM * 7 * D); A.sb_lin = (const double*)up(sb_lin, c->sumM * 9 * D);
    A.Hpp = (const double*)up(Hpp, c->sumM * 225 * D); A.HpN = (const double*)up(HpN, pno[n] * D); A.rhs_p = (const double*)up(rhs_p, c->sumM * 15 * D);
    A.HNN = (const double*)up(HNN, nno[n] * D); A.rhsN = (const double*)up(rhsN, c->sumN * D);
    A.pre = (const double*)up(pre, (size_t)(c->sumM + n) * SWF_PRE_DOUBLES * D);
    {
        std::vector<double> pg((size_t)n * 6);
        for (int f = 0; f < n; f++) for (int k = 0; k < 3; k++) { pg[(size_t)f * 6 + k] = pbg[k]; pg[(size_t)f * 6 + 3 + k] = gw[k]; }
        A.pbgw = (const double*)up(pg.data(), pg.size() * D);
    }
    A.active = nullptr;
    A.hmn_inv = (double*)up(nullptr, c->sumM * 225 * D); A.hmn_2 = (double*)up(nullptr, c->sumM * 225 * D); A.hmn_0 = (double*)up(nullptr, c->sumM * 225 * D);
    A.hmn_N = (double*)up(nullptr, pno[n] * D); A.rhsmn = (double*)up(nullptr, c->sumM * 15 * D);
    A.Hd = (double*)up(nullptr, c->sumG2 * D); A.rd = (double*)up(nullptr, c->sumG * D); A.Ld = (double*)up(nullptr, c->sumG2 * D); A.r0 = (double*)up(nullptr, c->sumG * D);
    A.old = (double*)up(nullptr, (size_t)n * 32 * D); A.N_old = (double*)up(nullptr, c->sumN * D);
    A.history = (int*)up(nullptr, n * sizeof(int)); A.status = (int*)up(nullptr, n * sizeof(int));
    A.outer = (const double*)up(nullptr, (size_t)n * 32 * D); A.Nv = (const double*)up(nullptr, c->sumN * D);
    A.res_out = (double*)up(nullptr, c->sumG * D); A.jac_out = (double*)up(nullptr, c->sumG2 * D);
    A.Jw = (double*)up(nullptr, (size_t)(c->sumM + n) * 450 * D); A.rw = (double*)up(nullptr, (size_t)(c->sumM + n) * 16 * D);
    {
        std::vector<int> qf, qk;
        for (int f = 0; f < n; f++) for (int k = 0; k <= M[f]; k++) { qf.push_back(f); qk.push_back(k); }
        A.n_iq = (int)qf.size();
        A.iq_f = (const int*)up(qf.data(), qf.size() * sizeof(int)); A.iq_k = (const int*)up(qk.data(), qk.size() * sizeof(int));
        A.todo = (int*)up(nullptr, n * sizeof(int));
    }
    A.mid = (const int*)up(nullptr, n * sizeof(int)); A.H12 = (const double*)up(nullptr, (size_t)n * 225 * D);
    c->M.assign(M, M + n);
    if (bad) return fail(SWF_E_NODEVICE, "swf_composite_create: device allocation / upload failed");
    *out = c.release();
    return SWF_OK;
}

extern "C" int swf_composite_evaluate(swf_composite* c, const double* outer, const double* Nv, int32_t want_jac,
                                      double* residual, double* jac, double* Hd, double* rd, int32_t* status) {
    if (!c || !outer || (!Nv && c->sumN) || !residual) return fail(SWF_E_INVALID, "swf_composite_evaluate: null argument");
    hipStream_t st = c->stream;
    HIPCHK(hipMemcpyAsync((void*)c->A.outer, outer, (size_t)c->n * 32 * sizeof(double), hipMemcpyHostToDevice, st));
    if (c->sumN) HIPCHK(hipMemcpyAsync((void*)c->A.Nv, Nv, c->sumN * sizeof(double), hipMemcpyHostToDevice, st));
    CompArgs A = c->A;
    A.want_jac = want_jac ? 1 : 0;
    hipLaunchKernelGGL(k_comp_prep, dim3(c->n), dim3(256), 0, st, A);
    hipLaunchKernelGGL(k_comp_imu, dim3((A.n_iq + 7) / 8), dim3(256), 0, st, A);
    if (c->nmin <= CO_SMALLN) hipLaunchKernelGGL((k_comp_elim<CO_SMALLN, 256>), dim3(c->n), dim3(256), 0, st, A, 0);
    if (c->nmax > CO_SMALLN) hipLaunchKernelGGL((k_comp_elim<CO_MAXN, 256>), dim3(c->n), dim3(256), 0, st, A, CO_SMALLN + 1);
    if (c->eigen_root) {
        if (c->nmax <= CO_SMALLN) hipLaunchKernelGGL(k_comp_eigroot<CO_SMALLN>, dim3(c->n), dim3(256), 0, st, A);
        else hipLaunchKernelGGL(k_comp_eigroot<CO_MAXN>, dim3(c->n), dim3(512), 0, st, A);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(residual, c->A.res_out, c->sumG * sizeof(double), hipMemcpyDeviceToHost, st));
    if (jac && want_jac) HIPCHK(hipMemcpyAsync(jac, c->A.jac_out, c->sumG2 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (Hd) HIPCHK(hipMemcpyAsync(Hd, c->A.Hd, c->sumG2 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (rd) HIPCHK(hipMemcpyAsync(rd, c->A.rd, c->sumG * sizeof(double), hipMemcpyDeviceToHost, st));
    if (status) HIPCHK(hipMemcpyAsync(status, c->A.status, c->n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return SWF_OK;
}

extern "C" int swf_composite_set_root(swf_composite* c, int32_t form) {
    if (!c || (form != SWF_ROOT_PIVOTED_CHOLESKY && form != SWF_ROOT_EIGEN)) return fail(SWF_E_INVALID, "swf_composite_set_root: bad arguments");
    c->eigen_root = form == SWF_ROOT_EIGEN;
    return SWF_OK;
}

extern "C" int swf_composite_set_mid_links(swf_composite* c, const int32_t* mid, const double* H12) {
    if (!c || !mid || !H12) return fail(SWF_E_INVALID, "swf_composite_set_mid_links: null argument");
    for (int f = 0; f < c->n; f++)
        if (mid[f] != 0 && (mid[f] < 1 || mid[f] > c->M[(size_t)f] - 1)) return fail(SWF_E_INVALID, "swf_composite_set_mid_links: a link must lie between two hidden epochs (1..M-1)");
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy((void*)c->A.mid, mid, (size_t)c->n * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy((void*)c->A.H12, H12, (size_t)c->n * 225 * sizeof(double), hipMemcpyHostToDevice));
    return SWF_OK;
}

extern "C" int swf_composite_hidden(swf_composite* c, double* pose, double* sb) {
    if (!c || !pose || !sb) return fail(SWF_E_INVALID, "swf_composite_hidden: null argument");
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(pose, c->A.pose, c->sumM * 7 * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(sb, c->A.sb, c->sumM * 9 * sizeof(double), hipMemcpyDeviceToHost));
    return SWF_OK;
}
