// swf_problem.cpp — the ceres::Problem / Solver::Solve shaped surface of include/swf_solver.h.
//
// Host bookkeeping only: pointer-keyed parameter blocks and typed factors are flattened into a
// swf_flat_window (include/swf_types.h) whenever the STRUCTURE changed, and handed to the batch
// engine (B = 1); per solve only parameter values travel.  Semantics follow what the reference
// relies on: blocks identified by address, AddParameterBlock on an existing
// block only re-attaches the manifold (R/swf/swf_core.cpp:53-56), RemoveParameterBlock cascades
// to its residual blocks (R/factor/gnss_imu_factor.cpp:110-113), blocks that no enabled residual
// block touches are left out of the solve (ceres removes unused blocks from the reduced program),
// failures surface through the return code AND summary.termination.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/swf_solver.h"

namespace {
struct PBlock { int size; bool constant; long seq; };
enum FType { FT_PROJ, FT_IMU, FT_CP, FT_PR, FT_DOP, FT_SP, FT_PRIOR, FT_SPR, FT_SCP, FT_FIX, FT_COMP, FT_IDP };
struct PFactor {
    FType type; bool alive, enabled;
    std::vector<double*> keys;
    std::vector<double> data;      // type-specific record
    double sqrt_info = 0, loss_a = 0;
    int dim = 0;                   // prior
    int M = 0, N = 0;              // composite IMU-GNSS factor: hidden epochs, ambiguities
    double* hid_pose = nullptr; double* hid_sb = nullptr;     // its hidden epochs: caller memory, [M][7] / [M][9]
    int mid = 0; std::vector<double> H12;                     // its middle-marginalisation link (swf_set_imu_gnss_mid_link)
};
}  // namespace

struct swf_problem {
    std::unordered_map<double*, PBlock> blocks;
    long seq = 0;
    std::vector<PFactor> factors;
    std::vector<swf_factor_id> free_ids;   // slots of removed factors, reused by the next additions: a long-lived problem (landmarks come and go every frame) stays bounded
    std::vector<double*> order_keys; std::vector<int> order_groups;
    std::vector<double*> tail_keys;
    double pbg[3] = {0, 0, 0}, gw[3] = {0, 0, 9.8}, base[3] = {0, 0, 0};
    bool dirty = true;             // structure changed since the batch was built
    // flattened form (valid while !dirty)
    swf_batch* batch = nullptr;
    swf_flat_window fw{};
    std::vector<double> pose, sb, lm, sc;
    std::vector<double*> kpose, ksb, klm, ksc;       // key per pool slot
    std::vector<uint8_t> is_const;
    std::vector<int32_t> order_block, order_group;
    std::vector<int32_t> proj_idx, imu_idx, cp_idx, pr_idx, dop_idx, sp_idx, spr_idx, scp_idx, fix_idx, prior_nblk, prior_dim, prior_blk;
    std::vector<double> proj_uv, imu_pre, cp_dat, pr_dat, dop_dat, sp_w, spr_dat, scp_dat, fix_dat, prior_J, prior_r0, prior_x0;
    // exports
    std::vector<double> S, rhs, L;
    std::vector<double> mgA, mgb, mgJ, mgr0;      // swf_problem_marginalize outputs
    std::vector<double> tcA, tcQ;                 // swf_problem_tail_covariance outputs
    // composite IMU-GNSS factors, flattened
    std::vector<int32_t> comp_M, comp_N, comp_idx;
    std::vector<double> comp_pose, comp_sb, comp_pose_lin, comp_sb_lin, comp_Hpp, comp_HpN, comp_rhs_p, comp_HNN, comp_rhsN, comp_pre;
    std::vector<const PFactor*> comp_fac;
    std::vector<int32_t> comp_mid; std::vector<double> comp_H12;
    std::vector<int32_t> idp_kind, idp_idx; std::vector<double> idp_pts;      // inverse-depth projection factors, flattened
    int hs_row = 0;
    bool solved = false;
};

void swf_internal_set_error(const std::string& m);
static int pfail(int code, const std::string& m) { swf_internal_set_error(m); return code; }

extern "C" {

int swf_problem_create(swf_problem** out) { if (!out) return SWF_E_INVALID; *out = new swf_problem(); return SWF_OK; }
int swf_problem_destroy(swf_problem* p) {
    if (!p) return SWF_OK;
    if (p->batch) swf_batch_destroy(p->batch);
    delete p;
    return SWF_OK;
}

int swf_add_parameter_block(swf_problem* p, double* key, int32_t size, int32_t manifold) {
    if (!p || !key) return SWF_E_INVALID;
    if (!(size == 7 || size == 9 || size == 3 || size == 1)) return pfail(SWF_E_UNSUPPORTED, "parameter block size must be 7, 9, 3 or 1");
    auto it = p->blocks.find(key);
    if (it != p->blocks.end()) {
        if (it->second.size != size) return pfail(SWF_E_INVALID, "parameter block re-added with a different size");
        return SWF_OK;          // re-attaching the parameterization (R/swf/swf_core.cpp:53-56) changes nothing: see below
    }
    // the kernels fix the manifold by size: a 7-block is [p, q] on PoseLocalParameterization, everything else is Euclidean
    if (manifold == SWF_MANIFOLD_POSE && size != 7) return pfail(SWF_E_UNSUPPORTED, "SWF_MANIFOLD_POSE on a block that is not 7-dimensional");
    p->blocks[key] = PBlock{ size, false, p->seq++ };
    p->dirty = true;
    return SWF_OK;
}
// a removed factor gives its slot (and its memory) back; the id is handed out again by a later swf_add_* (as ceres reuses the slots of removed
// residual blocks: a ResidualBlockId is dead after RemoveResidualBlock)
static void kill_factor(swf_problem* p, swf_factor_id id) {
    PFactor dead; dead.type = p->factors[id].type; dead.alive = false; dead.enabled = false;
    p->factors[id] = std::move(dead);
    p->free_ids.push_back(id);
}
int swf_has_parameter_block(swf_problem* p, const double* key) { return p && p->blocks.count((double*)key) ? 1 : 0; }
int swf_remove_parameter_block(swf_problem* p, double* key) {
    if (!p) return SWF_E_INVALID;
    auto it = p->blocks.find(key);
    if (it == p->blocks.end()) return SWF_E_NOTFOUND;
    for (size_t i = 0; i < p->factors.size(); i++) { PFactor& f = p->factors[i]; if (f.alive && std::find(f.keys.begin(), f.keys.end(), key) != f.keys.end()) kill_factor(p, (swf_factor_id)i); }
    p->blocks.erase(it);
    p->dirty = true;
    return SWF_OK;
}
static int set_const(swf_problem* p, double* key, bool c) {
    if (!p) return SWF_E_INVALID;
    auto it = p->blocks.find(key);
    if (it == p->blocks.end()) return SWF_E_NOTFOUND;
    if (it->second.constant != c) { it->second.constant = c; p->dirty = true; }
    return SWF_OK;
}
int swf_set_parameter_block_constant(swf_problem* p, double* key) { return set_const(p, key, true); }
int swf_set_parameter_block_variable(swf_problem* p, double* key) { return set_const(p, key, false); }
int swf_is_parameter_block_constant(swf_problem* p, const double* key) {
    if (!p) return 0;
    auto it = p->blocks.find((double*)key);
    return it != p->blocks.end() && it->second.constant ? 1 : 0;
}
int swf_parameter_block_size(swf_problem* p, const double* key) {
    if (!p) return SWF_E_INVALID;
    auto it = p->blocks.find((double*)key);
    return it == p->blocks.end() ? SWF_E_NOTFOUND : it->second.size;
}
int swf_num_parameter_blocks(swf_problem* p) { return p ? (int)p->blocks.size() : SWF_E_INVALID; }
int swf_num_residual_blocks(swf_problem* p) {
    if (!p) return SWF_E_INVALID;
    int n = 0;
    for (auto& f : p->factors) if (f.alive) n++;
    return n;
}

static swf_factor_id add_factor(swf_problem* p, FType t, std::vector<double*> keys, const std::vector<int>& sizes,
                                const double* data, size_t ndata) {
    if (!p) return SWF_E_INVALID;
    for (size_t i = 0; i < keys.size(); i++) {
        if (!keys[i]) return SWF_E_INVALID;
        auto it = p->blocks.find(keys[i]);
        if (it == p->blocks.end()) {
            int rc = swf_add_parameter_block(p, keys[i], sizes[i], sizes[i] == 7 ? SWF_MANIFOLD_POSE : SWF_MANIFOLD_NONE);
            if (rc != SWF_OK) return rc;
        } else if (it->second.size != sizes[i]) return pfail(SWF_E_INVALID, "factor: parameter block has the wrong size");
    }
    PFactor f; f.type = t; f.alive = true; f.enabled = true; f.keys = std::move(keys);
    if (data && ndata) f.data.assign(data, data + ndata);
    p->dirty = true;
    if (!p->free_ids.empty()) { const swf_factor_id id = p->free_ids.back(); p->free_ids.pop_back(); p->factors[id] = std::move(f); return id; }
    p->factors.push_back(std::move(f));
    return (swf_factor_id)p->factors.size() - 1;
}

swf_factor_id swf_add_projection(swf_problem* p, double* pose, double* ex, double* pt, const double uv[2], double sqrt_info, double loss_a) {
    swf_factor_id id = add_factor(p, FT_PROJ, { pose, ex, pt }, { 7, 7, 3 }, uv, 2);
    if (id >= 0) { p->factors[id].sqrt_info = sqrt_info; p->factors[id].loss_a = loss_a; }
    return id;
}
swf_factor_id swf_add_imu(swf_problem* p, double* pi, double* si, double* pj, double* sj, const double* pre) {
    return add_factor(p, FT_IMU, { pi, si, pj, sj }, { 7, 9, 7, 9 }, pre, SWF_PRE_DOUBLES);
}
swf_factor_id swf_add_rtk_carrier_phase(swf_problem* p, double* pose, double* amb, double* clk, const double* dat) {
    return add_factor(p, FT_CP, { pose, amb, clk }, { 7, 1, 1 }, dat, SWF_CP_DOUBLES);
}
swf_factor_id swf_add_rtk_pseudorange(swf_problem* p, double* pose, double* clk, const double* dat) {
    return add_factor(p, FT_PR, { pose, clk }, { 7, 1 }, dat, SWF_PR_DOUBLES);
}
swf_factor_id swf_add_doppler(swf_problem* p, double* sbp, double* drift, double* pose, const double* dat) {
    return add_factor(p, FT_DOP, { sbp, drift, pose }, { 9, 1, 7 }, dat, SWF_DOP_DOUBLES);
}
swf_factor_id swf_add_scalar_prior(swf_problem* p, double* scalar, double w) {
    return add_factor(p, FT_SP, { scalar }, { 1 }, &w, 1);
}
swf_factor_id swf_add_spp_pseudorange(swf_problem* p, double* pose, double* clk, const double* dat) {
    return add_factor(p, FT_SPR, { pose, clk }, { 7, 1 }, dat, SWF_SPR_DOUBLES);
}
swf_factor_id swf_add_spp_carrier_phase(swf_problem* p, double* pose, double* clk, double* amb, const double* dat) {
    return add_factor(p, FT_SCP, { pose, clk, amb }, { 7, 1, 1 }, dat, SWF_SCP_DOUBLES);
}
swf_factor_id swf_add_fixed_integer(swf_problem* p, double* na, double* nb, double N21, double istd) {
    if (na == nb) return pfail(SWF_E_INVALID, "fixed integer: both blocks are the same scalar");
    double dat[SWF_FIX_DOUBLES] = { N21, istd };
    return add_factor(p, FT_FIX, { na, nb }, { 1, 1 }, dat, SWF_FIX_DOUBLES);
}
swf_factor_id swf_add_projection_inverse_depth(swf_problem* p, int32_t kind, double* pose_i, double* pose_j, double* ex, double* ex2,
                                               double* inv_depth, const double pts_i[3], const double pts_j[3], double sqrt_info, double loss_a) {
    if (!p || kind < 0 || kind > 2 || !ex || !inv_depth || !pts_i || !pts_j || (kind != 2 && (!pose_i || !pose_j)) || (kind != 0 && !ex2))
        return pfail(SWF_E_INVALID, "swf_add_projection_inverse_depth: bad arguments");
    std::vector<double*> keys; std::vector<int> sizes;
    if (kind != 2) { keys.push_back(pose_i); keys.push_back(pose_j); sizes.push_back(7); sizes.push_back(7); }
    keys.push_back(ex); sizes.push_back(7);
    if (kind != 0) { keys.push_back(ex2); sizes.push_back(7); }
    keys.push_back(inv_depth); sizes.push_back(1);
    double rec[7] = { (double)kind, pts_i[0], pts_i[1], pts_i[2], pts_j[0], pts_j[1], pts_j[2] };
    swf_factor_id id = add_factor(p, FT_IDP, keys, sizes, rec, 7);
    if (id >= 0) { p->factors[id].sqrt_info = sqrt_info; p->factors[id].loss_a = loss_a; }
    return id;
}
swf_factor_id swf_add_imu_gnss(swf_problem* p, double* pose_i, double* sb_i, double* pose_j, double* sb_j, double* const* ambiguities,
                               int32_t N, int32_t M, double* hidden_pose, double* hidden_sb, const double* pose_lin, const double* sb_lin,
                               const double* Hpp, const double* HpN, const double* rhs_p, const double* HNN, const double* rhsN, const double* pre) {
    if (!p || M < 1 || N < 0 || (N && !ambiguities) || !hidden_pose || !hidden_sb || !pose_lin || !sb_lin || !Hpp || (N && (!HpN || !HNN || !rhsN)) || !rhs_p || !pre)
        return pfail(SWF_E_INVALID, "swf_add_imu_gnss: bad arguments");
    std::vector<double*> keys = { pose_i, sb_i, pose_j, sb_j };
    std::vector<int> sizes = { 7, 9, 7, 9 };
    for (int q = 0; q < N; q++) { keys.push_back(ambiguities[q]); sizes.push_back(1); }
    // record: pose_lin [M][7] | sb_lin [M][9] | Hpp [M][225] | HpN [M][15 N] | rhs_p [M][15] | HNN [N N] | rhsN [N] | pre [M+1][SWF_PRE_DOUBLES]
    std::vector<double> rec;
    auto app = [&](const double* src, size_t n) { if (n) rec.insert(rec.end(), src, src + n); };
    app(pose_lin, (size_t)M * 7); app(sb_lin, (size_t)M * 9); app(Hpp, (size_t)M * 225); app(HpN, (size_t)M * 15 * N); app(rhs_p, (size_t)M * 15);
    app(HNN, (size_t)N * N); app(rhsN, (size_t)N); app(pre, (size_t)(M + 1) * SWF_PRE_DOUBLES);
    swf_factor_id id = add_factor(p, FT_COMP, keys, sizes, rec.data(), rec.size());
    if (id >= 0) { PFactor& f = p->factors[id]; f.M = M; f.N = N; f.hid_pose = hidden_pose; f.hid_sb = hidden_sb; }
    return id;
}
int swf_set_imu_gnss_mid_link(swf_problem* p, swf_factor_id id, int32_t k, const double* H12) {
    if (!p || id < 0 || id >= (swf_factor_id)p->factors.size() || p->factors[id].type != FT_COMP || !p->factors[id].alive) return pfail(SWF_E_INVALID, "swf_set_imu_gnss_mid_link: not a composite factor of this problem");
    PFactor& f = p->factors[id];
    if (k != 0 && (k < 1 || k > f.M - 1 || !H12)) return pfail(SWF_E_INVALID, "swf_set_imu_gnss_mid_link: the link must lie between two hidden epochs (1..M-1)");
    f.mid = k;
    if (k) f.H12.assign(H12, H12 + 225); else f.H12.clear();
    p->dirty = true;
    return SWF_OK;
}
swf_factor_id swf_add_linear_prior(swf_problem* p, double* const* keys, int32_t n_keys, const double* J, const double* r0, const double* x0) {
    if (!p || !keys || n_keys <= 0 || !J || !r0 || !x0) return SWF_E_INVALID;
    std::vector<double*> k(keys, keys + n_keys);
    std::vector<int> sizes;
    int dim = 0, gsum = 0;
    for (double* q : k) {
        auto it = p->blocks.find(q);
        if (it == p->blocks.end()) return pfail(SWF_E_NOTFOUND, "linear prior: unknown parameter block (sizes come from the problem)");
        sizes.push_back(it->second.size);
        dim += it->second.size == 7 ? 6 : it->second.size; gsum += it->second.size;
    }
    std::vector<double> data;
    data.insert(data.end(), J, J + (size_t)dim * dim);
    data.insert(data.end(), r0, r0 + dim);
    data.insert(data.end(), x0, x0 + gsum);
    swf_factor_id id = add_factor(p, FT_PRIOR, k, sizes, data.data(), data.size());
    if (id >= 0) p->factors[id].dim = dim;
    return id;
}
int swf_remove_factor(swf_problem* p, swf_factor_id id) {
    if (!p || id < 0 || id >= (int)p->factors.size() || !p->factors[id].alive) return SWF_E_NOTFOUND;
    kill_factor(p, id); p->dirty = true;
    return SWF_OK;
}
int swf_factor_set_enabled(swf_problem* p, swf_factor_id id, int32_t on) {
    if (!p || id < 0 || id >= (int)p->factors.size() || !p->factors[id].alive) return SWF_E_NOTFOUND;
    bool b = on != 0;
    if (p->factors[id].enabled != b) { p->factors[id].enabled = b; p->dirty = true; }
    return SWF_OK;
}
int swf_factor_is_enabled(swf_problem* p, swf_factor_id id) {
    if (!p || id < 0 || id >= (int)p->factors.size() || !p->factors[id].alive) return SWF_E_NOTFOUND;
    return p->factors[id].enabled ? 1 : 0;
}
// ---- the query surface of ceres::Problem: every getter reports the full count in *n and fills at most cap
// entries, so a caller may size its buffer with a first call (cap = 0)
int swf_get_residual_blocks(swf_problem* p, swf_factor_id* ids, int32_t cap, int32_t* n) {
    if (!p || !n || (cap > 0 && !ids)) return SWF_E_INVALID;
    int c = 0;
    for (size_t i = 0; i < p->factors.size(); i++) if (p->factors[i].alive) { if (c < cap) ids[c] = (swf_factor_id)i; c++; }
    *n = c;
    return SWF_OK;
}
int swf_get_residual_blocks_for_parameter_block(swf_problem* p, const double* key, swf_factor_id* ids, int32_t cap, int32_t* n) {
    if (!p || !n || (cap > 0 && !ids)) return SWF_E_INVALID;
    if (!p->blocks.count((double*)key)) return pfail(SWF_E_NOTFOUND, "GetResidualBlocksForParameterBlock: unknown parameter block");
    int c = 0;
    for (size_t i = 0; i < p->factors.size(); i++) {
        const PFactor& f = p->factors[i];
        if (!f.alive || std::find(f.keys.begin(), f.keys.end(), (double*)key) == f.keys.end()) continue;
        if (c < cap) ids[c] = (swf_factor_id)i;
        c++;
    }
    *n = c;
    return SWF_OK;
}
int swf_get_parameter_blocks(swf_problem* p, double** keys, int32_t cap, int32_t* n) {
    if (!p || !n || (cap > 0 && !keys)) return SWF_E_INVALID;
    std::map<long, double*> by_seq;                   // insertion order, as ceres::Problem::GetParameterBlocks reports it
    for (auto& kv : p->blocks) by_seq[kv.second.seq] = kv.first;
    int c = 0;
    for (auto& kv : by_seq) { if (c < cap) keys[c] = kv.second; c++; }
    *n = c;
    return SWF_OK;
}
int swf_get_parameter_blocks_for_residual_block(swf_problem* p, swf_factor_id id, double** keys, int32_t cap, int32_t* n) {
    if (!p || !n || (cap > 0 && !keys)) return SWF_E_INVALID;
    if (id < 0 || id >= (int)p->factors.size() || !p->factors[id].alive) return pfail(SWF_E_NOTFOUND, "GetParameterBlocksForResidualBlock: unknown residual block");
    const PFactor& f = p->factors[id];
    for (int i = 0; i < (int)f.keys.size() && i < cap; i++) keys[i] = f.keys[i];
    *n = (int32_t)f.keys.size();
    return SWF_OK;
}
int swf_set_constants(swf_problem* p, const double pbg[3], const double gw[3], const double base[3]) {
    if (!p) return SWF_E_INVALID;
    for (int k = 0; k < 3; k++) {
        if (pbg && p->pbg[k] != pbg[k]) { p->pbg[k] = pbg[k]; p->dirty = true; }
        if (gw && p->gw[k] != gw[k]) { p->gw[k] = gw[k]; p->dirty = true; }
        if (base && p->base[k] != base[k]) { p->base[k] = base[k]; p->dirty = true; }
    }
    return SWF_OK;
}
int swf_set_ordering(swf_problem* p, double* const* keys, const int32_t* groups, int32_t n) {
    if (!p || (n > 0 && (!keys || !groups))) return SWF_E_INVALID;
    // ceres::Solve through the adapter sets the ordering before every solve: only a CHANGE is a structure change
    if ((int32_t)p->order_keys.size() == n && (n == 0 || (std::equal(keys, keys + n, p->order_keys.begin()) && std::equal(groups, groups + n, p->order_groups.begin()))))
        return SWF_OK;
    if (n > 0) { p->order_keys.assign(keys, keys + n); p->order_groups.assign(groups, groups + n); }
    else { p->order_keys.clear(); p->order_groups.clear(); }
    p->dirty = true;
    return SWF_OK;
}
int swf_set_export_tail(swf_problem* p, double* const* keys, int32_t n) {
    if (!p || (n > 0 && !keys)) return SWF_E_INVALID;
    if ((int32_t)p->tail_keys.size() == n && (n == 0 || std::equal(keys, keys + n, p->tail_keys.begin()))) return SWF_OK;
    if (n > 0) p->tail_keys.assign(keys, keys + n); else p->tail_keys.clear();
    p->dirty = true;
    return SWF_OK;
}

static int flatten(swf_problem* p) {
    // blocks touched by at least one live, enabled factor, in insertion order per pool
    std::map<long, double*> used;
    for (auto& f : p->factors) if (f.alive && f.enabled) for (double* k : f.keys) used[p->blocks[k].seq] = k;
    p->kpose.clear(); p->ksb.clear(); p->klm.clear(); p->ksc.clear();
    for (auto& kv : used) {
        int s = p->blocks[kv.second].size;
        (s == 7 ? p->kpose : s == 9 ? p->ksb : s == 3 ? p->klm : p->ksc).push_back(kv.second);
    }
    int nP = (int)p->kpose.size(), nS = (int)p->ksb.size(), nL = (int)p->klm.size(), nC = (int)p->ksc.size();
    std::unordered_map<double*, int> pool_idx, bid;
    for (int i = 0; i < nP; i++) { pool_idx[p->kpose[i]] = i; bid[p->kpose[i]] = i; }
    for (int i = 0; i < nS; i++) { pool_idx[p->ksb[i]] = i; bid[p->ksb[i]] = nP + i; }
    for (int i = 0; i < nL; i++) { pool_idx[p->klm[i]] = i; bid[p->klm[i]] = nP + nS + i; }
    for (int i = 0; i < nC; i++) { pool_idx[p->ksc[i]] = i; bid[p->ksc[i]] = nP + nS + nL + i; }
    p->pose.assign(7 * (size_t)nP, 0); p->sb.assign(9 * (size_t)nS, 0); p->lm.assign(3 * (size_t)nL, 0); p->sc.assign((size_t)nC, 0);
    p->is_const.assign((size_t)nP + nS + nL + nC, 0);
    for (auto& kv : bid) p->is_const[kv.second] = p->blocks[kv.first].constant ? 1 : 0;
    // ordering: keep the caller's (group, position) for used variable blocks; the tail is whatever
    // the caller listed in parameter_head, expected at the end of the ordering
    p->order_block.clear(); p->order_group.clear();
    std::vector<double*> auto_keys; std::vector<int> auto_groups;
    const bool automatic = p->order_keys.empty();
    if (automatic) {
        // options.linear_solver_ordering == nullptr (the default-constructed Solver::Options of GnssPreprocess / GnssProcess,
        // R/swf/swf_gnss.cpp:200-216, 562-572): ceres then picks an independent set itself.  Here: group 0 = every landmark and,
        // greedily in insertion order, every scalar that shares no factor with a block already in group 0 (and whose clique
        // stays within the one-wavefront limits); every other variable block gets a group of its own, parameter_head last.
        std::unordered_map<double*, std::vector<const PFactor*>> touch;
        for (auto& f : p->factors) if (f.alive && f.enabled) for (double* k : f.keys) touch[k].push_back(&f);
        std::unordered_map<double*, char> in0;
        auto variable = [&](double* k) { return !p->blocks[k].constant; };
        for (double* k : p->klm) if (variable(k)) {
            bool ok = true;
            for (const PFactor* f : touch[k]) if (f->type != FT_PROJ) ok = false;
            if (ok) { in0[k] = 1; auto_keys.push_back(k); auto_groups.push_back(0); }
        }
        for (double* k : p->ksc) if (variable(k)) {
            bool ok = true; int rows = 0, cols = 1;
            std::vector<double*> nb;
            for (const PFactor* f : touch[k]) {
                if (f->type == FT_PRIOR || f->type == FT_COMP) { ok = false; break; }
                rows += f->type == FT_IDP ? 2 : 1;
                for (double* q : f->keys) if (q != k && variable(q)) {
                    if (in0.count(q)) ok = false;
                    if (std::find(nb.begin(), nb.end(), q) == nb.end()) { nb.push_back(q); int sz = p->blocks[q].size; cols += sz == 7 ? 6 : sz; }
                }
            }
            if (ok && rows <= 48 && cols <= 32) { in0[k] = 1; auto_keys.push_back(k); auto_groups.push_back(0); }
        }
        int g = 1;
        auto is_tail = [&](double* k) { return std::find(p->tail_keys.begin(), p->tail_keys.end(), k) != p->tail_keys.end(); };
        for (int pass = 0; pass < 2; pass++)
            for (auto& kv : used) {
                double* k = kv.second;
                if (!variable(k) || in0.count(k) || (is_tail(k) != (pass == 1))) continue;
                auto_keys.push_back(k); auto_groups.push_back(g++);
            }
    }
    const std::vector<double*>& okeys = automatic ? auto_keys : p->order_keys;
    const std::vector<int>& ogroups = automatic ? auto_groups : p->order_groups;
    for (size_t i = 0; i < okeys.size(); i++) {
        auto it = bid.find(okeys[i]);
        if (it == bid.end() || p->is_const[it->second]) continue;
        p->order_block.push_back(it->second); p->order_group.push_back(ogroups[i]);
    }
    int n_tail = 0;
    for (int i = (int)p->order_block.size() - 1; i >= 0; i--) {
        bool in_tail = false;
        for (double* t : p->tail_keys) { auto it = bid.find(t); if (it != bid.end() && it->second == p->order_block[i]) in_tail = true; }
        if (!in_tail) break;
        n_tail++;
    }
    // factors
    p->proj_idx.clear(); p->proj_uv.clear(); p->imu_idx.clear(); p->imu_pre.clear(); p->cp_idx.clear(); p->cp_dat.clear();
    p->pr_idx.clear(); p->pr_dat.clear(); p->dop_idx.clear(); p->dop_dat.clear(); p->sp_idx.clear(); p->sp_w.clear();
    p->spr_idx.clear(); p->spr_dat.clear(); p->scp_idx.clear(); p->scp_dat.clear(); p->fix_idx.clear(); p->fix_dat.clear();
    p->idp_kind.clear(); p->idp_idx.clear(); p->idp_pts.clear();
    p->comp_M.clear(); p->comp_N.clear(); p->comp_idx.clear(); p->comp_pose.clear(); p->comp_sb.clear(); p->comp_pose_lin.clear(); p->comp_sb_lin.clear();
    p->comp_Hpp.clear(); p->comp_HpN.clear(); p->comp_rhs_p.clear(); p->comp_HNN.clear(); p->comp_rhsN.clear(); p->comp_pre.clear(); p->comp_fac.clear(); p->comp_mid.clear(); p->comp_H12.clear();
    p->prior_nblk.clear(); p->prior_dim.clear(); p->prior_blk.clear(); p->prior_J.clear(); p->prior_r0.clear(); p->prior_x0.clear();
    double sqrt_info = 0, loss_a = 0; bool have_proj = false;
    for (auto& f : p->factors) {
        if (!f.alive || !f.enabled) continue;
        switch (f.type) {
        case FT_PROJ:
            if (have_proj && (f.sqrt_info != sqrt_info || f.loss_a != loss_a))
                return pfail(SWF_E_UNSUPPORTED, "projection factors must share sqrt_info and loss (static member in the reference)");
            have_proj = true; sqrt_info = f.sqrt_info; loss_a = f.loss_a;
            for (double* k : f.keys) p->proj_idx.push_back(pool_idx[k]);
            p->proj_uv.insert(p->proj_uv.end(), f.data.begin(), f.data.end());
            break;
        case FT_IMU: for (double* k : f.keys) p->imu_idx.push_back(pool_idx[k]); p->imu_pre.insert(p->imu_pre.end(), f.data.begin(), f.data.end()); break;
        case FT_CP: for (double* k : f.keys) p->cp_idx.push_back(pool_idx[k]); p->cp_dat.insert(p->cp_dat.end(), f.data.begin(), f.data.end()); break;
        case FT_PR: for (double* k : f.keys) p->pr_idx.push_back(pool_idx[k]); p->pr_dat.insert(p->pr_dat.end(), f.data.begin(), f.data.end()); break;
        case FT_DOP: for (double* k : f.keys) p->dop_idx.push_back(pool_idx[k]); p->dop_dat.insert(p->dop_dat.end(), f.data.begin(), f.data.end()); break;
        case FT_SP: p->sp_idx.push_back(pool_idx[f.keys[0]]); p->sp_w.push_back(f.data[0]); break;
        case FT_SPR: for (double* k : f.keys) p->spr_idx.push_back(pool_idx[k]); p->spr_dat.insert(p->spr_dat.end(), f.data.begin(), f.data.end()); break;
        case FT_SCP: for (double* k : f.keys) p->scp_idx.push_back(pool_idx[k]); p->scp_dat.insert(p->scp_dat.end(), f.data.begin(), f.data.end()); break;
        case FT_FIX: for (double* k : f.keys) p->fix_idx.push_back(pool_idx[k]); p->fix_dat.insert(p->fix_dat.end(), f.data.begin(), f.data.end()); break;
        case FT_IDP: {
            if (have_proj && (f.sqrt_info != sqrt_info || f.loss_a != loss_a))
                return pfail(SWF_E_UNSUPPORTED, "projection factors must share sqrt_info and loss (static member in the reference)");
            have_proj = true; sqrt_info = f.sqrt_info; loss_a = f.loss_a;
            const int kd = (int)f.data[0];
            int ix[5] = { -1, -1, -1, -1, -1 }, q = 0;
            if (kd != 2) { ix[0] = pool_idx[f.keys[q++]]; ix[1] = pool_idx[f.keys[q++]]; }
            ix[2] = pool_idx[f.keys[q++]];
            if (kd != 0) ix[3] = pool_idx[f.keys[q++]];
            ix[4] = pool_idx[f.keys[q++]];
            p->idp_kind.push_back(kd); p->idp_idx.insert(p->idp_idx.end(), ix, ix + 5); p->idp_pts.insert(p->idp_pts.end(), f.data.begin() + 1, f.data.end());
            break;
        }
        case FT_COMP: {
            const int M = f.M, N = f.N;
            p->comp_M.push_back(M); p->comp_N.push_back(N); p->comp_fac.push_back(&f);
            for (double* k : f.keys) p->comp_idx.push_back(pool_idx[k]);
            p->comp_pose.insert(p->comp_pose.end(), f.hid_pose, f.hid_pose + (size_t)M * 7);
            p->comp_sb.insert(p->comp_sb.end(), f.hid_sb, f.hid_sb + (size_t)M * 9);
            const double* d = f.data.data();
            auto take = [&](std::vector<double>& dst, size_t n) { dst.insert(dst.end(), d, d + n); d += n; };
            take(p->comp_pose_lin, (size_t)M * 7); take(p->comp_sb_lin, (size_t)M * 9); take(p->comp_Hpp, (size_t)M * 225); take(p->comp_HpN, (size_t)M * 15 * N);
            take(p->comp_rhs_p, (size_t)M * 15); take(p->comp_HNN, (size_t)N * N); take(p->comp_rhsN, (size_t)N); take(p->comp_pre, (size_t)(M + 1) * SWF_PRE_DOUBLES);
            p->comp_mid.push_back(f.mid);
            if (f.mid) p->comp_H12.insert(p->comp_H12.end(), f.H12.begin(), f.H12.end()); else p->comp_H12.resize(p->comp_H12.size() + 225, 0.0);
            break;
        }
        case FT_PRIOR: {
            int dim = f.dim, gsum = 0;
            p->prior_nblk.push_back((int)f.keys.size()); p->prior_dim.push_back(dim);
            for (double* k : f.keys) { p->prior_blk.push_back(bid[k]); gsum += p->blocks[k].size; }
            p->prior_J.insert(p->prior_J.end(), f.data.begin(), f.data.begin() + (size_t)dim * dim);
            p->prior_r0.insert(p->prior_r0.end(), f.data.begin() + (size_t)dim * dim, f.data.begin() + (size_t)dim * dim + dim);
            p->prior_x0.insert(p->prior_x0.end(), f.data.begin() + (size_t)dim * dim + dim, f.data.begin() + (size_t)dim * dim + dim + gsum);
            break;
        }
        }
    }
    swf_flat_window& w = p->fw;
    memset(&w, 0, sizeof(w));
    w.n_pose = nP; w.pose = p->pose.data(); w.n_sb = nS; w.sb = p->sb.data(); w.n_lm = nL; w.lm = p->lm.data(); w.n_sc = nC; w.sc = p->sc.data();
    w.is_const = p->is_const.data();
    w.n_order = (int)p->order_block.size(); w.order_block = p->order_block.data(); w.order_group = p->order_group.data(); w.n_tail = n_tail;
    w.n_proj = (int)p->proj_idx.size() / 3; w.proj_idx = p->proj_idx.data(); w.proj_uv = p->proj_uv.data();
    w.proj_sqrt_info = sqrt_info; w.proj_loss_a = loss_a;
    w.n_imu = (int)p->imu_idx.size() / 4; w.imu_idx = p->imu_idx.data(); w.imu_pre = p->imu_pre.data();
    w.n_cp = (int)p->cp_idx.size() / 3; w.cp_idx = p->cp_idx.data(); w.cp_dat = p->cp_dat.data();
    w.n_pr = (int)p->pr_idx.size() / 2; w.pr_idx = p->pr_idx.data(); w.pr_dat = p->pr_dat.data();
    w.n_dop = (int)p->dop_idx.size() / 3; w.dop_idx = p->dop_idx.data(); w.dop_dat = p->dop_dat.data();
    w.n_sp = (int)p->sp_idx.size(); w.sp_idx = p->sp_idx.data(); w.sp_w = p->sp_w.data();
    w.n_spr = (int)p->spr_idx.size() / 2; w.spr_idx = p->spr_idx.data(); w.spr_dat = p->spr_dat.data();
    w.n_scp = (int)p->scp_idx.size() / 3; w.scp_idx = p->scp_idx.data(); w.scp_dat = p->scp_dat.data();
    w.n_fix = (int)p->fix_idx.size() / 2; w.fix_idx = p->fix_idx.data(); w.fix_dat = p->fix_dat.data();
    w.n_idp = (int)p->idp_kind.size(); w.idp_kind = p->idp_kind.data(); w.idp_idx = p->idp_idx.data(); w.idp_pts = p->idp_pts.data();
    w.n_comp = (int)p->comp_M.size(); w.comp_M = p->comp_M.data(); w.comp_N = p->comp_N.data(); w.comp_idx = p->comp_idx.data();
    w.comp_pose = p->comp_pose.data(); w.comp_sb = p->comp_sb.data(); w.comp_pose_lin = p->comp_pose_lin.data(); w.comp_sb_lin = p->comp_sb_lin.data();
    w.comp_Hpp = p->comp_Hpp.data(); w.comp_HpN = p->comp_HpN.data(); w.comp_rhs_p = p->comp_rhs_p.data(); w.comp_HNN = p->comp_HNN.data();
    w.comp_rhsN = p->comp_rhsN.data(); w.comp_pre = p->comp_pre.data(); w.comp_mid = p->comp_mid.data(); w.comp_H12 = p->comp_H12.data();
    w.n_prior = (int)p->prior_nblk.size(); w.prior_nblk = p->prior_nblk.data(); w.prior_dim = p->prior_dim.data();
    w.prior_blk = p->prior_blk.data(); w.prior_J = p->prior_J.data(); w.prior_r0 = p->prior_r0.data(); w.prior_x0 = p->prior_x0.data();
    for (int k = 0; k < 3; k++) { w.pbg[k] = p->pbg[k]; w.gw[k] = p->gw[k]; w.base[k] = p->base[k]; }
    return SWF_OK;
}

static void gather_values(swf_problem* p) {      // Vector2Double direction: caller memory -> staging pools
    for (size_t i = 0; i < p->kpose.size(); i++) memcpy(&p->pose[7 * i], p->kpose[i], 7 * sizeof(double));
    for (size_t i = 0; i < p->ksb.size(); i++) memcpy(&p->sb[9 * i], p->ksb[i], 9 * sizeof(double));
    for (size_t i = 0; i < p->klm.size(); i++) memcpy(&p->lm[3 * i], p->klm[i], 3 * sizeof(double));
    for (size_t i = 0; i < p->ksc.size(); i++) p->sc[i] = *p->ksc[i];
    size_t e = 0;                                  // hidden epochs of the composite factors (gnss_poses / gnss_speed_bias)
    for (const PFactor* f : p->comp_fac) { memcpy(&p->comp_pose[7 * e], f->hid_pose, (size_t)f->M * 7 * sizeof(double)); memcpy(&p->comp_sb[9 * e], f->hid_sb, (size_t)f->M * 9 * sizeof(double)); e += f->M; }
}
static void scatter_values(swf_problem* p) {     // Double2Vector direction; constant blocks are never written
    auto cst = [&](double* k) { return p->blocks[k].constant; };
    for (size_t i = 0; i < p->kpose.size(); i++) if (!cst(p->kpose[i])) memcpy(p->kpose[i], &p->pose[7 * i], 7 * sizeof(double));
    for (size_t i = 0; i < p->ksb.size(); i++) if (!cst(p->ksb[i])) memcpy(p->ksb[i], &p->sb[9 * i], 9 * sizeof(double));
    for (size_t i = 0; i < p->klm.size(); i++) if (!cst(p->klm[i])) memcpy(p->klm[i], &p->lm[3 * i], 3 * sizeof(double));
    for (size_t i = 0; i < p->ksc.size(); i++) if (!cst(p->ksc[i])) *p->ksc[i] = p->sc[i];
    size_t e = 0;
    for (const PFactor* f : p->comp_fac) { memcpy(f->hid_pose, &p->comp_pose[7 * e], (size_t)f->M * 7 * sizeof(double)); memcpy(f->hid_sb, &p->comp_sb[9 * e], (size_t)f->M * 9 * sizeof(double)); e += f->M; }
}

int swf_problem_solve(swf_problem* p, const swf_options* opt, swf_summary* summary) {
    if (!p || !opt || !summary) return SWF_E_INVALID;
    int rc;
    p->solved = false;                             // consumers (marginalize, tail covariance, get_reduced) answer only for a solve that went through
    const bool trace = getenv("SWF_TRACE_REBUILD") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    if (p->dirty || !p->batch) {
        if (p->batch) { swf_batch_destroy(p->batch); p->batch = nullptr; }
        double t1 = now();
        if ((rc = flatten(p)) != SWF_OK) return rc;
        gather_values(p);
        double t2 = now();
        const swf_flat_window* wp = &p->fw;
        if ((rc = swf_batch_create(&wp, 1, nullptr, &p->batch)) != SWF_OK) return rc;
        p->dirty = false;
        if (trace) fprintf(stderr, "[swf] rebuild: destroy %.3f ms, flatten %.3f ms, batch_create %.3f ms\n", t1 - t0, t2 - t1, now() - t2);
    } else {
        gather_values(p);
        if ((rc = swf_batch_upload_state(p->batch)) != SWF_OK) return rc;
    }
    swf_batch_enable_timing(p->batch, 1);      // bracket the whole solve: summary.minimizer_time_in_seconds
    if ((rc = swf_batch_solve(p->batch, opt)) != SWF_OK) return rc;
    if ((rc = swf_batch_sync(p->batch)) != SWF_OK) return rc;
    if ((rc = swf_batch_download_state(p->batch)) != SWF_OK) return rc;
    scatter_values(p);
    if ((rc = swf_batch_summaries(p->batch, summary)) != SWF_OK) return rc;
    p->hs_row = summary->reduced_dim;
    size_t n = (size_t)p->hs_row;
    p->S.assign(n * n, 0); p->rhs.assign(n, 0); p->L.assign(n * n, 0);
    if ((rc = swf_batch_export_reduced(p->batch, 0, p->S.data(), p->rhs.data(), p->L.data())) != SWF_OK) return rc;
    p->solved = true;
    return SWF_OK;
}

int swf_problem_marginalize(swf_problem* p, double eps, int32_t form, const double** J, const double** r0,
                            const double** A, const double** bv, int32_t* n_out, int32_t* rank) {
    if (!p) return SWF_E_INVALID;
    if (!p->solved || !p->batch) return SWF_E_STATE;
    int rc;
    if ((rc = swf_batch_marginalize(p->batch, eps, form)) != SWF_OK) return rc;
    int32_t n = 0, rk = 0;
    if ((rc = swf_batch_get_prior(p->batch, 0, nullptr, nullptr, nullptr, nullptr, nullptr, &n, &rk)) != SWF_OK) return rc;
    p->mgA.assign((size_t)n * n, 0); p->mgJ.assign((size_t)n * n, 0); p->mgb.assign((size_t)n, 0); p->mgr0.assign((size_t)n, 0);
    if ((rc = swf_batch_get_prior(p->batch, 0, p->mgA.data(), p->mgb.data(), p->mgJ.data(), p->mgr0.data(), nullptr, &n, &rk)) != SWF_OK) return rc;
    if (J) *J = p->mgJ.data(); if (r0) *r0 = p->mgr0.data(); if (A) *A = p->mgA.data(); if (bv) *bv = p->mgb.data();
    if (n_out) *n_out = n; if (rank) *rank = rk;
    return SWF_OK;
}

int swf_problem_tail_covariance(swf_problem* p, const double** A, const double** Qy, int32_t* n_out) {
    if (!p) return SWF_E_INVALID;
    if (!p->solved || !p->batch) return SWF_E_STATE;
    int rc;
    if ((rc = swf_batch_tail_covariance(p->batch)) != SWF_OK) return rc;
    int32_t n = 0;
    if ((rc = swf_batch_get_tail_covariance(p->batch, 0, nullptr, nullptr, &n)) != SWF_OK) return rc;
    p->tcA.assign((size_t)n * n, 0); p->tcQ.assign((size_t)n * n, 0);
    if ((rc = swf_batch_get_tail_covariance(p->batch, 0, p->tcA.data(), p->tcQ.data(), &n)) != SWF_OK) return rc;
    if (A) *A = p->tcA.data(); if (Qy) *Qy = p->tcQ.data(); if (n_out) *n_out = n;
    return SWF_OK;
}

int swf_get_reduced(swf_problem* p, const double** S, const double** rhs, const double** L, int32_t* hs_row) {
    if (!p) return SWF_E_INVALID;
    if (!p->solved) return SWF_E_STATE;
    if (S) *S = p->S.data(); if (rhs) *rhs = p->rhs.data(); if (L) *L = p->L.data(); if (hs_row) *hs_row = p->hs_row;
    return SWF_OK;
}

}  // extern "C"
