#!/usr/bin/env python
"""bench.py — Gauss-Newton iterations/sec of the sliding-window solver on MI355X.

One "step" = one full solve (<= 8 dogleg iterations, each = Jacobian evaluation of every block +
J^T J assembly + Schur + dense reduced solve + step + cost re-evaluation + accept/reject) of a
batch of independent BASELINE-cfg3 windows (20 keyframes / 300 features / 10 satellites) that is
already resident in HBM.  Windows shard across GPUs with no data-path collective (weak scaling:
--windows per GPU).  value = whole-job Gauss-Newton iterations per second.

Driver contract: python bench.py --gpus N --steps K --warmup W   (N > 1 via torch.distributed.run)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
FP64_MATRIX_PEAK_TFLOPS = 78.6  # MI355X datasheet FP64 matrix peak; the guide lists no fp64 figure
FP64_MATRIX_MEASURED_TFLOPS = 47.7  # tests/microbench/mfma_f64_peak.hip on the gpurun box (v_mfma_f64_16x16x4_f64, 4 waves/SIMD)


def _gen(args):
    cfg, seed = args
    from rtk_visual_inertial_navigation_amd import synth
    return synth.make_window(config_id=cfg, seed=seed)


def make_windows(cfg, seeds):
    import multiprocessing as mp
    n = min(len(seeds), max(1, min(32, (os.cpu_count() or 2) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))))
    if n <= 1 or len(seeds) < 8:
        return [_gen((cfg, s)) for s in seeds]
    with mp.get_context("fork").Pool(n) as pool:
        return pool.map(_gen, [(cfg, s) for s in seeds], chunksize=max(1, len(seeds) // (4 * n)))


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(windows, iters, budget_s=12.0, threads=4):
    """The oracle (a plain-C port of the reference path, oracle/swf_oracle.c) timed on the host's
    cores on a bounded sample of the SAME windows; `threads` windows are solved concurrently (the
    reference's num_threads = 4, R/swf/swf.cpp:29, parallelises inside one solve instead)."""
    import oracle_binding as ob
    from concurrent.futures import ThreadPoolExecutor
    from rtk_visual_inertial_navigation_amd.flat import default_options
    ob.lib()
    t0 = time.perf_counter()
    probe = windows[0].copy()
    sm, _ = ob.solve(probe, default_options(max_num_iterations=iters), export=False)
    t1 = time.perf_counter() - t0
    n = int(max(threads, min(len(windows), budget_s / max(t1, 1e-4) * threads * 0.8)))
    sample = [w.copy() for w in windows[:n]]

    def run(w):
        s, _ = ob.solve(w, default_options(max_num_iterations=iters), export=False)
        return s.num_iterations

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        its = list(ex.map(run, sample))
    dt = time.perf_counter() - t0
    return dict(value=sum(its) / dt, unit="gauss_newton_iterations/s", cores=threads, kind="port",
                sample="%d of the benchmark's cfg3 windows, %d dogleg iterations each, %d windows solved concurrently "
                       "(one thread per window), %.1f s of wall time" % (len(sample), iters, threads, dt),
                single_thread_us_per_iteration=1e6 * t1 / max(1, sm.num_iterations),
                host_cores=os.cpu_count(), host_cpu_model=_cpu_model())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--windows", type=int, default=512, help="windows per GPU (BASELINE cfg4 = 512 over the job)")
    ap.add_argument("--iters", type=int, default=8, help="max_num_iterations (yaml MAX_NUM_ITERATIONS = 8)")
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-window", action="store_true",
                    help="skip the single-window latency leg (its launches share kernel names with the batch and would dilute rocprofv3 per-kernel averages)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # synthetic windows first: the generator forks a process pool, which must happen before this
    # process touches HIP / RCCL
    from rtk_visual_inertial_navigation_amd import synth, shard
    B = a.windows
    t0 = time.perf_counter()
    windows = make_windows(a.config, shard.window_seeds(synth.BASE_SEED, a.config, B, rank))
    t_gen = time.perf_counter() - t0

    import torch
    import torch.distributed as dist
    # SWF_BENCH_SHARE_GPU=1 (testing the multi-rank path on a box with fewer GPUs than ranks): ranks share the
    # devices round-robin and the harness collectives run over gloo on host tensors; never set it for a measurement
    share = os.environ.get("SWF_BENCH_SHARE_GPU") == "1"
    cdev = "cpu" if share else "cuda"
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("gloo" if share else "nccl")
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product has no CPU path"
    if share:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    from rtk_visual_inertial_navigation_amd import solver
    from rtk_visual_inertial_navigation_amd.flat import default_options
    solver.set_device(local_rank)
    t0 = time.perf_counter()
    bs = solver.BatchSolver(windows)
    t_struct = time.perf_counter() - t0          # symbolic phase + one-off upload (reported, not timed)
    opt = default_options(max_num_iterations=a.iters)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up, with every kernel bracketed once to find the dominant one
    bs.reset_state(); bs.solve_async(opt); bs.sync()             # first touch of every buffer, code objects loaded
    bs.enable_timing(True)
    for _ in range(max(1, a.warmup)):
        bs.reset_state(); bs.solve_async(opt); bs.sync()
    calib = bs.timing()
    kern = {k: v for k, v in calib["kernels"].items() if k != "total"}
    dom = max(kern, key=lambda k: kern[k]["ms"])
    from rtk_visual_inertial_navigation_amd.solver import K_NAMES
    mask = 1 | (1 << K_NAMES.index(dom)) | (1 << K_NAMES.index("eval_ps"))
    bs.enable_timing(mask)

    barrier()
    t0 = time.perf_counter()
    acc = {}
    for _ in range(a.steps):
        bs.reset_state(); bs.solve_async(opt); bs.sync()
        t = bs.timing()
        for k, v in t["kernels"].items():
            d = acc.setdefault(k, dict(ms=0.0, calls=0)); d["ms"] += v["ms"]; d["calls"] += v["calls"]
    barrier()
    dt = time.perf_counter() - t0
    dt = shard.allreduce([dt], "max", device=cdev)[0]          # RCCL all-reduce (harness only)
    sms = bs.summaries()
    its_total = shard.allreduce([float(sum(s.num_iterations for s in sms))], "sum", device=cdev)[0]
    job = shard.gather_summaries(np.array([[s.final_cost, s.num_iterations, s.termination] for s in sms]), device=cdev)
    value = its_total * a.steps / dt

    if rank == 0:
        def avg_ms(k):
            return acc[k]["ms"] / max(1, acc[k]["calls"])

        # algorithmic work of ONE launch over this GPU's batch (DESIGN.md §3 states the per-unit figures)
        n_obs = calib["n_obs"]
        work = {
            "eval_ps": ("hbm", calib["proj_bytes"], "312 B per observation (152 read + 160 written)"),
            "lm_schur": ("mfma", calib["lm_schur_flops"], "sum over landmarks of 216 k^2 + 108 k flops (SURVEY.md 8d landmark Schur); "
                         "HBM side: 208 B per observation (Jp, Jl, r read = 160 B, Y g_l written = 48 B) + the P partials"),
            "chol_solve": ("mfma", calib["chol_flops"], "sum_w n_red^3 / 3 flops (n_red^3 / 6 multiply-adds)"),
            "frame_sums": ("hbm", 160 * n_obs, "160 B per observation (Jp, r, Y g_l read)"),
            "post_chol": ("hbm", 2 * 144 * n_obs, "Jp, Jl (144 B per observation) read by the back-substitution and by J D^-2 g"),
            "post_dogleg": ("hbm", (144 + 152 + 16) * n_obs, "Jp, Jl read for J*step; candidate residuals: 152 B read + 16 B written per observation"),
        }
        bound, units, what = work.get(dom, ("hbm", calib["jacobian_bytes"], "Jacobian bytes of the batch (SURVEY.md 8d formula)"))
        knames = {"eval_ps": "k_eval_ps<true>", "lm_schur": "k_lm_schur<8, 5>", "assemble": "k_assemble_all", "post_chol": "k_post_chol", "post_dogleg": "k_post_dogleg", "frame_sums": "k_frame_sums", "chol_solve": "k_chol_rr2<9>"}
        # HBM traffic from the committed PMC passes of the same workload (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
        # separate passes; gfx950: FETCH_SIZE counts half of wide coalesced reads -> x2), if available
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01", "batch512_pmc_fetch_write.json")))
            kn = knames.get(dom, "k_" + dom)
            fk = [k for k in pmc["FETCH_SIZE"] if kn.split("<")[0] in k]
            if fk and B == 512:
                traffic = (2 * pmc["FETCH_SIZE"][fk[0]]["avg_kb_per_launch"] + pmc["WRITE_SIZE"][fk[0]]["avg_kb_per_launch"]) * 1024.0
        except Exception:
            traffic = None
        if bound == "mfma":
            achieved = units / (avg_ms(dom) * 1e-3) / 1e12
            roof = dict(kernel=knames.get(dom, "k_" + dom), bound="mfma", achieved=achieved, peak=FP64_MATRIX_PEAK_TFLOPS,
                        unit="TFLOP/s", frac=achieved / FP64_MATRIX_PEAK_TFLOPS, traffic=traffic, algorithmic=what,
                        algorithmic_flops_per_launch=units, avg_launch_ms=avg_ms(dom),
                        measured_fp64_mfma_ceiling_tflops=FP64_MATRIX_MEASURED_TFLOPS)
            if dom == "lm_schur":      # the same kernel is also the elimination pass over the observations: report its HBM side too
                hb = 208 * n_obs
                roof["hbm_side"] = dict(algorithmic_bytes_per_launch=hb, achieved_GBs=hb / (avg_ms(dom) * 1e-3) / 1e9,
                                        frac_of_hbm_peak=hb / (avg_ms(dom) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        note="208 B per observation (Jp, Jl, r read = 160 B; Y g_l written = 48 B); the Y|W cells stay in LDS")
        else:
            achieved = units / (avg_ms(dom) * 1e-3) / 1e9
            roof = dict(kernel=knames.get(dom, "k_" + dom), bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBS, traffic=traffic, algorithmic=what, algorithmic_bytes_per_launch=units,
                        avg_launch_ms=avg_ms(dom))
        jac = dict(kernel="k_eval_ps<true>", bound="hbm",
                   achieved=calib["proj_bytes"] / (avg_ms("eval_ps") * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                   algorithmic_bytes_per_launch=calib["proj_bytes"], avg_launch_ms=avg_ms("eval_ps"))
        jac["frac"] = jac["achieved"] / HBM_PEAK_GBS
        # single-window latency path (rank 0, extra information)
        single = None
        if not a.no_single_window:
            one = solver.BatchSolver([windows[0].copy()])
            one.enable_timing(1)
            for _ in range(3):
                one.reset_state(); one.solve_async(opt); one.sync()
            lat = []
            for _ in range(20):
                one.reset_state(); one.solve_async(opt); one.sync()
                lat.append(one.timing()["total_ms"])
            it1 = one.summaries()[0].num_iterations
            one.close()
            single = dict(us_per_iteration=1e3 * float(np.median(lat)) / max(1, it1), iterations_per_s=max(1, it1) / (1e-3 * float(np.median(lat))),
                          solve_ms_median=float(np.median(lat)), solve_ms_p10=float(np.percentile(lat, 10)), solve_ms_p90=float(np.percentile(lat, 90)))
        out = {
            "metric": "gauss_newton_iterations_per_sec", "value": value, "unit": "iterations/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE cfg4: batch of independent cfg3 windows (20 keyframes, 300 features, 3000 observations, "
                                   "19 IMU factors, 10 satellites x 20 epochs carrier-phase+pseudorange, gauge prior), "
                                   "%d windows per GPU, %d dogleg iterations per solve" % (B, a.iters),
                       "windows_per_gpu": B, "max_num_iterations": a.iters, "sharding": "independent windows, no data-path collective"},
            "windows_per_sec": world * B * a.steps / dt,
            "iterations_per_window": its_total / (world * B),
            "job_final_cost_mean": float(job[:, 0].mean()), "job_windows": int(job.shape[0]),
            "roofline": roof,
            "roofline_jacobian": jac,
            "kernel_ms_per_solve_calibration": {k: v["ms"] for k, v in calib["kernels"].items()},
            "single_window": single,
            "setup": {"generate_s": t_gen, "structure_upload_s": t_struct},
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(windows, a.iters)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            if single:
                out["single_window"]["speedup_vs_cpu_single_thread"] = out["cpu_baseline"]["single_thread_us_per_iteration"] / single["us_per_iteration"]
        print(json.dumps(out))
    bs.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
