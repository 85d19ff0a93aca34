"""CPU-side tests of the host logic: MyOrdering policy, generator determinism, flat-window
packing, and that the C-ABI library loads and exports every symbol include/swf_solver.h declares
(no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest
from rtk_visual_inertial_navigation_amd.flat import default_options

from rtk_visual_inertial_navigation_amd import synth, solver, build
from rtk_visual_inertial_navigation_amd.ordering import my_ordering

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generator_is_deterministic_and_matches_baseline_sizes():
    a, b = synth.make_window(3), synth.make_window(3)
    for k in a.a:
        assert np.array_equal(a.a[k], b.a[k])
    c = a.counts()
    assert (c["n_proj"], c["n_imu"], c["n_cp"] + c["n_pr"], c["prior_dim"]) == (3000, 19, 400, [15])
    c2 = synth.make_window(2).counts()
    assert (c2["n_proj"], c2["n_imu"], c2["n_cp"]) == (500, 9, 0)


def test_my_ordering_policy():
    """App. B of SURVEY.md: group 0 = dummy, landmarks, every other ELIGIBLE speed-bias (prior
    blocks are not eligible), clocks; then speed-biases, poses, ambiguities, prior blocks last."""
    w = synth.make_window(3, K=6, F=10, S=3, seed=1)
    r = w.meta["roles"]
    ob, og = list(w.a["order_block"]), list(w.a["order_group"])
    g = dict(zip(ob, og))
    assert g[r["dummy"]] == 0 and all(g[b] == 0 for b in r["landmarks"]) and all(g[b] == 0 for b in r["clocks"])
    sb = r["speed_bias"]
    # sb0 is held by the prior -> not eligible; eligible = sb1..sb5, alternate ones (1,3,5) in group 0
    assert [g[b] == 0 for b in sb] == [False, True, False, True, False, True]
    # groups strictly ascend one block at a time after group 0
    tail = [x for x in og if x > 0]
    assert tail == list(range(1, len(tail) + 1))
    # order: remaining speed-biases (frame order), poses, ambiguities, then prior's kept blocks
    rest = [b for b, gg in zip(ob, og) if gg > 0]
    exp = [sb[2], sb[4]] + r["poses"][1:] + r["rtk_ambiguities"] + [r["poses"][0], sb[0]]
    assert rest == exp
    # every variable block exactly once
    nvar = int((w.a["is_const"] == 0).sum())
    assert len(ob) == len(set(ob)) == nvar
    # parameter_head goes last and is counted as the export tail
    r2 = dict(r); r2["parameter_head"] = list(r["rtk_ambiguities"])
    ob2, og2, nt = my_ordering(r2, w.a["is_const"])
    assert nt == 3 and list(ob2[-3:]) == r["rtk_ambiguities"]


def test_flat_window_struct_roundtrip():
    w = synth.make_window(3, K=4, F=6, S=2, seed=2)
    s = w.c_struct()
    assert s.n_pose == 5 and s.n_sb == 4 and s.n_lm == 6 and s.n_proj == w.a["proj_idx"].size // 3
    assert s.pose[7] == w.a["pose"].ravel()[7]
    assert s.n_order == w.a["order_block"].size
    assert abs(s.proj_sqrt_info - 1000.0 / 1.5) < 1e-12


def test_c_abi_library_exports_every_declared_symbol():
    build.build()
    lib = ctypes.CDLL(solver.LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "swf_solver.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(swf_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 40
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert set(solver.EXPORTED) <= declared
    assert lib.swf_version() >= 104
    # the ctypes mirrors of the ABI structs have the library's sizes (swf_abi_sizes: options, summary, timing, flat window, iteration)
    from rtk_visual_inertial_navigation_amd.flat import FlatWindowC, OptionsC, SummaryC
    sz = (ctypes.c_int32 * 5)()
    assert lib.swf_abi_sizes(sz) == 0
    assert (sz[0], sz[1], sz[2], sz[3]) == (ctypes.sizeof(OptionsC), ctypes.sizeof(SummaryC), ctypes.sizeof(solver.TimingC), ctypes.sizeof(FlatWindowC))


def test_no_cpu_fallback_without_gpu():
    """The product path must fail loudly when no HIP device is present."""
    if solver.device_count() > 0:
        pytest.skip("a GPU is present")
    w = synth.make_window(2, K=3, F=5, S=0, seed=3)
    with pytest.raises(solver.SwfError):
        solver.BatchSolver([w])


def _compile_shim_example(tmp_path, name="shim_example"):
    import subprocess
    build.build()
    exe = os.path.join(str(tmp_path), name)
    libdir = os.path.dirname(solver.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", name + ".cpp"),
                           "-o", exe, "-L" + libdir, "-lswf_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_ceres_shaped_cpp_header_compiles_and_fails_loudly_without_gpu(tmp_path):
    """Estimator code in the reference's style (ceres::Problem / AddResidualBlock / ceres::Solve)
    compiles against include/swf_ceres.hpp; without a GPU the solve reports final_cost > 1e10
    (the failure convention the reference checks, R/swf/swf_image.cpp:220-223)."""
    import subprocess
    exe = _compile_shim_example(tmp_path)
    if solver.device_count() > 0:
        pytest.skip("a GPU is present (the run is covered by the gpu test)")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "Final cost: 1.000000e+300" in r.stdout


def test_reference_constructor_signatures_of_the_adapter(tmp_path):
    """IMUFactor(IntegrationBase*) (R/factor/imu_factor.h:11) and MarginalizationFactor(MarginalizationInfo*)
    (R/factor/marginalization_factor.h:106): the adapter takes the reference's own objects (stand-ins with its member names and Eigen's
    accessors here) and files their contents in the C-ABI's record layouts — the prior's columns re-ordered to the order of its blocks."""
    import subprocess
    exe = _compile_shim_example(tmp_path, "shim_reference_ctors")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "reference constructors: ok" in r.stdout, r.stdout + r.stderr


def test_globalmarge_sequence_compiles_against_the_adapter(tmp_path):
    """SWFOptimization::GlobalMarge (R/swf/swf_image.cpp:343-433) statement by statement — GetParameterBlocks, GetResidualBlocks,
    ->is_use, GetResidualBlocksForParameterBlock, GetParameterBlocksForResidualBlock, parameter_head, is_optimize — compiles
    against include/swf_ceres.hpp (it runs on the GPU tier)."""
    _compile_shim_example(tmp_path, "shim_globalmarge")


def test_reference_constructor_solve_compiles_as_cxx14(tmp_path):
    """tests/shim_reference_solve.cpp — IMUFactor(IntegrationBase*), MarginalizationFactor(MarginalizationInfo*), IMUGNSSFactor(IMUGNSSBase*)
    (R/factor/gnss_imu_factor.h:145-151) and the raw globals ceres::internal::lhs_out / rhs_out / lhs_out2 / hs_row read as
    UpdateSchur reads them (R/swf/swf_gnss.cpp:25-94) — compiles with the reference's -std=c++14 (CMakeLists.txt:5); without a GPU
    the solve reports the failure convention."""
    import subprocess
    exe = _compile_shim_example(tmp_path, "shim_reference_solve")
    if solver.device_count() == 0:
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 1 and "Final cost: 1.000000e+300" in r.stdout


@pytest.mark.gpu
def test_solve_built_through_the_reference_constructors_runs_on_gpu(tmp_path):
    """The same window built from the C-ABI's records and through the reference's three pointer-taking constructors (stand-ins with
    the reference's member names; every hidden GNSS epoch in its own allocation) solves to the same bits, the hidden epochs are
    written back into their own memory, and UpdateSchur's / UpdateSchurHessianOnly's reads of the raw globals see the reduced system."""
    import subprocess
    exe = _compile_shim_example(tmp_path, "shim_reference_solve")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bit for bit" in r.stdout and "reference solve: ok" in r.stdout


@pytest.mark.gpu
def test_globalmarge_sequence_runs_on_gpu(tmp_path):
    import subprocess
    exe = _compile_shim_example(tmp_path, "shim_globalmarge")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GlobalMarge: 4 kept blocks, prior n = 24" in r.stdout and "position error after the slide" in r.stdout


@pytest.mark.gpu
def test_ceres_shaped_cpp_example_runs_on_gpu(tmp_path):
    import subprocess
    exe = _compile_shim_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Iterations:" in r.stdout


def test_multi_gpu_example_compiles_and_the_c_defaults_match_the_python_ones(tmp_path):
    """The in-process multi-device entry (swf_batch_create_sharded + swf_solve_batches) compiles from plain C++ against
    include/swf_solver.h; swf_default_options fills the struct flat.default_options() fills.  Without a GPU the example reports it."""
    import subprocess
    exe = _compile_shim_example(tmp_path, "shim_multi_gpu")
    from rtk_visual_inertial_navigation_amd.flat import OptionsC, default_options
    o = OptionsC()
    solver.lib().swf_default_options.restype = None
    solver.lib().swf_default_options(ctypes.byref(o))
    d = default_options()
    for f, _ in OptionsC._fields_:
        assert getattr(o, f) == getattr(d, f), f
    if solver.device_count() == 0:
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 1 and "no HIP device" in r.stdout
        w = synth.make_window(2, K=3, F=5, S=0, seed=3)
        with pytest.raises(solver.SwfError):
            solver.ShardedBatchSolver([w])


@pytest.mark.gpu
def test_multi_gpu_example_runs_on_gpu(tmp_path):
    import subprocess
    exe = _compile_shim_example(tmp_path, "shim_multi_gpu")
    for mask in ("0", "1"):
        r = subprocess.run([exe, mask], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "37 of 37 windows solved" in r.stdout


@pytest.mark.gpu
def test_sharded_batches_from_one_process_equal_one_batch_bitwise():
    """swf_batch_create_sharded / swf_solve_batches: the windows dealt to the visible devices (here possibly one) and, separately, to two
    batches created on the same device with swf_batch_create_on and solved through ONE swf_solve_batches call, give what one batch
    gives, bit for bit (windows are independent units; a window's arithmetic does not depend on its batch)."""
    ws = [synth.make_window(3, K=5 + (i % 3), F=20 + 3 * i, S=5, seed=70 + i) for i in range(7)]
    ref = [w.copy() for w in ws]
    bs = solver.BatchSolver(ref); sm_ref = bs.solve(default_options()); bs.close()
    sh = [w.copy() for w in ws]
    S = solver.ShardedBatchSolver(sh, device_mask=0)
    assert sum(c for _, c in S.parts) == len(ws) and [f for f, _ in S.parts] == sorted(f for f, _ in S.parts)
    assert all(b.device() in range(solver.device_count()) for b in S.batches)
    sm = S.solve(default_options()); S.close()
    two = [w.copy() for w in ws]
    b0, b1 = solver.BatchSolver(two[:3], device=0), solver.BatchSolver(two[3:], device=solver.device_count() - 1)
    hs = (ctypes.c_void_p * 2)(b0._h, b1._h)
    opt = default_options()
    assert solver.lib().swf_solve_batches(hs, ctypes.c_int32(2), ctypes.byref(opt)) == 0
    b0.download_state(); b1.download_state()
    sm2 = b0.summaries() + b1.summaries()
    b0.close(); b1.close()
    for a, b_, c, s0, s1, s2 in zip(ref, sh, two, sm_ref, sm, sm2):
        assert s0.final_cost == s1.final_cost == s2.final_cost and s0.num_iterations == s1.num_iterations == s2.num_iterations
        for k in ("pose", "sb", "lm", "sc"):
            assert np.array_equal(a.a[k], b_.a[k]) and np.array_equal(a.a[k], c.a[k])


def test_problem_bookkeeping_without_gpu():
    """ceres::Problem bookkeeping semantics that need no device: implicit AddParameterBlock on
    AddResidualBlock, HasParameterBlock / IsParameterBlockConstant / ParameterBlockSize, cascade of
    RemoveParameterBlock, size mismatch refused."""
    w = synth.make_window(3, K=4, F=6, S=2, seed=2)
    P, blocks = solver.problem_from_window(w)
    assert P.NumParameterBlocks() == w.n_blocks
    n_fac = w.a["proj_idx"].size // 3 + 3 + 8 + 8 + 1 + 1
    assert P.NumResidualBlocks() == n_fac
    ex = blocks[w.bid_pose(4)]
    assert P.IsParameterBlockConstant(ex) and P.ParameterBlockSize(ex) == 7
    P.SetParameterBlockVariable(ex); assert not P.IsParameterBlockConstant(ex)
    pose1 = blocks[w.bid_pose(1)]
    n_touch = int((w.a["proj_idx"].reshape(-1, 3)[:, 0] == 1).sum()) + 2 + 2 + 2      # obs + 2 IMU + 2 CP + 2 PR
    P.RemoveParameterBlock(pose1)
    assert P.NumResidualBlocks() == n_fac - n_touch and not P.HasParameterBlock(pose1)
    with pytest.raises(solver.SwfError):
        P.AddProjection(blocks[w.bid_sb(0)], ex, blocks[w.bid_lm(0)], [0.0, 0.0])   # a 9-block where a pose is expected
    P.close()


def test_problem_query_surface_without_gpu():
    """GetResidualBlocks / GetResidualBlocksForParameterBlock / GetParameterBlocks / GetParameterBlocksForResidualBlock and the
    is_use flag, as GlobalMarge walks them (R/swf/swf_image.cpp:350-367): creation / insertion order, removed blocks disappear,
    RemoveParameterBlock cascades."""
    w = synth.make_window(3, K=4, F=6, S=2, seed=2)
    P, blocks = solver.problem_from_window(w)
    ids = P.GetResidualBlocks()
    assert ids == sorted(ids) and len(ids) == P.NumResidualBlocks()
    pb = P.GetParameterBlocks()
    assert len(pb) == P.NumParameterBlocks() == w.n_blocks
    assert all(any(q is b for q in blocks) for b in pb)
    # the first residual block is the first projection factor: pose, extrinsic, landmark in AddResidualBlock order
    p_, e_, l_ = w.a["proj_idx"].reshape(-1, 3)[0]
    got = P.GetParameterBlocksForResidualBlock(ids[0])
    assert got[0] is blocks[w.bid_pose(p_)] and got[1] is blocks[w.bid_pose(e_)] and got[2] is blocks[w.bid_lm(l_)]
    # per parameter block: every factor listed for a block lists the block back
    lm0 = blocks[w.bid_lm(0)]
    f_lm0 = P.GetResidualBlocksForParameterBlock(lm0)
    assert len(f_lm0) == int((w.a["proj_idx"].reshape(-1, 3)[:, 2] == 0).sum()) and len(f_lm0) > 0
    for f in f_lm0:
        assert any(q is lm0 for q in P.GetParameterBlocksForResidualBlock(f))
    # is_use round trip; removal
    assert P.IsResidualBlockUsed(f_lm0[0])
    P.SetResidualBlockUsed(f_lm0[0], False); assert not P.IsResidualBlockUsed(f_lm0[0])
    P.RemoveResidualBlock(f_lm0[0])
    assert f_lm0[0] not in P.GetResidualBlocks() and P.GetResidualBlocksForParameterBlock(lm0) == f_lm0[1:]
    with pytest.raises(solver.SwfError):
        P.GetParameterBlocksForResidualBlock(f_lm0[0])
    P.RemoveParameterBlock(lm0)
    assert not any(q is lm0 for q in P.GetParameterBlocks())
    with pytest.raises(solver.SwfError):
        P.GetResidualBlocksForParameterBlock(lm0)
    # a long-lived problem stays bounded: the slots of removed residual blocks are handed out again (the estimator adds and removes
    # landmarks every frame); the ids of live blocks do not move
    freed = set(f_lm0)
    n_before, live_before = P.NumResidualBlocks(), set(P.GetResidualBlocks())
    assert not (freed & live_before)
    lm1 = blocks[w.bid_lm(1)]
    new = [P.AddProjection(blocks[w.bid_pose(0)], blocks[w.bid_pose(4)], lm1, [0.01 * k, 0.0]) for k in range(len(freed) + 2)]
    assert set(new[:len(freed)]) == freed and all(f > max(live_before | freed) for f in new[len(freed):])
    assert P.NumResidualBlocks() == n_before + len(new) and set(P.GetResidualBlocks()) == live_before | set(new)
    assert all(any(q is lm1 for q in P.GetParameterBlocksForResidualBlock(f)) for f in new)
    P.close()
