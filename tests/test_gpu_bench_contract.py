"""The driver's bench contract, exercised on a miniature workload in-process (bench.py is a script: argv in, ONE JSON line
out): field names / types, the roofline and brute_force / pruned objects, and internal consistency of the numbers."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    from houv_amd import solver
    monkeypatch.setattr(solver, "PRUNED", solver.PRUNED)          # bench.main selects the search globally: restore it afterwards
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    lines = [l for l in buf.getvalue().splitlines() if l.startswith("{")]
    assert len(lines) == 1, buf.getvalue()
    return json.loads(lines[0])


def test_bench_line_contract(monkeypatch):
    d = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--pairs", "8", "--points", "768", "--kernel", "26",
              "--iters", "40", "--no-cpu-baseline", "--no-chamfer-op"], monkeypatch)
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict)):
        assert isinstance(d[key], typ), key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 8 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]          # pairs/s = pairs / time
    from houv_amd import _lib
    r = d["roofline"]
    assert r["bound"] == "valu" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    # frac = executed flops / peak: flops really executed, so it cannot exceed 1 -- nor the ~0.5 the instruction mix allows
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 0.6
    assert r["frac_executed_flops"] == r["frac"] and "executed fp32 flops" in r["frac_definition"]
    assert "solve_kernel<256, 3, 4, 2, 1>" in r["kernel"]          # the product default = the pruned search; derived from the launches
    assert "pruned" in d["config"]["solver"]
    assert r["launches"] > 0 and r["kernel_time_share"] <= 1.0
    # traffic comes from a committed PMC pass and is only passed on when that pass profiled THIS library build
    assert "traffic" in r and "traffic_source" in r and r["library_build_id"] == _lib.build_id()
    if r["traffic"] is None:
        assert "not reported" in r["traffic_source"] or "no " in r["traffic_source"]
    else:
        assert _lib.build_id() in r["traffic_source"] and 0 < r["valu_issue_occupancy_lower_bound"] <= 1.0
    assert 1.0 < r["sustained_clock_ghz"] < 2.6                       # measured inside the kernel (s_memtime / s_memrealtime)
    ps = r["pruned_search"]
    assert 0 < ps["share_of_point_pairs_evaluated"] < 1 and 0 < ps["lane_utilisation_of_the_walk"] <= 1
    b = d["brute_force"]                                              # every timed batch again through the brute-force sweep
    assert b["bit_identical_to_timed_run"] is True and b["batches_compared"] == 2 and b["unit"] == "pairs/s"
    br = b["roofline"]
    assert "solve_kernel<256, 3, 4, 0, 1>" in br["kernel"] and 0 < br["frac"] < 0.6
    assert 0 < br["valu_issue_slot_model"]["occupancy_at_sustained_clock"] < 1.1
    assert "pruned_search" not in br


def test_bench_brute_solver_flag(monkeypatch):
    """--solver brute times north_star's brute-force formulation; the other leg is then the pruned search."""
    d = _run(["--gpus", "1", "--steps", "1", "--warmup", "1", "--pairs", "8", "--points", "768", "--kernel", "26",
              "--iters", "30", "--no-cpu-baseline", "--no-chamfer-op", "--solver", "brute"], monkeypatch)
    assert "solve_kernel<256, 3, 4, 0, 1>" in d["roofline"]["kernel"] and d["config"]["solver"].startswith("brute")
    assert d["pruned"]["bit_identical_to_timed_run"] is True and "4, 2, 1>" in d["pruned"]["roofline"]["kernel"]
    # up to 256 points both searches are the brute-force kernel: nothing to compare, and the line says so
    d = _run(["--gpus", "1", "--steps", "1", "--warmup", "1", "--pairs", "8", "--points", "256", "--kernel", "26",
              "--iters", "20", "--no-cpu-baseline", "--no-chamfer-op"], monkeypatch)
    assert "solve_kernel<256, 1, 4, 0, 1>" in d["roofline"]["kernel"] and "skipped" in d["brute_force"]


def test_bench_one_rank_through_rccl():
    """the RCCL branch has to run at least once.  bench.py --gpus 1 --force-process-group initialises the
    "nccl" (= RCCL) process group at world size 1 on cuda:0 exactly as an N > 1 rank does (device_id given), so the
    all_gather_into_tensor of the transforms, the barrier and the MAX all-reduce of the time run on HBM tensors, and the
    self-diagnosis object `ranks` (per-rank seconds, retried pairs, gather time) is on the line.  Run as a subprocess
    started from the forkserver: a hung collective must not take pytest with it."""
    import multiprocessing as mp
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-process-group", "--backend", "nccl",
           "--steps", "2", "--warmup", "1", "--pairs", "8", "--points", "512", "--kernel", "26", "--iters", "40",
           "--no-cpu-baseline", "--no-chamfer-op", "--no-other-solver"]
    ctx = mp.get_context("forkserver")
    q = ctx.Queue()
    proc = ctx.Process(target=_launch, args=(cmd, ROOT, env, q))
    proc.start()
    rc, out, err = q.get(timeout=700)
    proc.join(timeout=60)
    assert rc == 0, err
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and "RCCL" in d["config"]["parallelism"]
    rk = d["ranks"]
    assert len(rk["per_rank_s"]) == 1 and rk["per_rank_s"][0] > 0 and rk["gather_ms"]["max"] > 0
    assert rk["per_rank_retried_pairs"][0] >= 0 and rk["load_balance_mean_over_max"] == 1.0


def _launch(cmd, cwd, env, q):
    import subprocess
    res = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    q.put((res.returncode, res.stdout, res.stderr[-3000:]))


def test_bench_two_ranks_as_the_driver_launches_it():
    """The N > 1 launch line of the driver (python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...), rehearsed
    with two ranks on the ONE GPU of the test box: --backend gloo --single-device swap RCCL for gloo and put both ranks on
    cuda:0; everything else (rank-local batches, one all-gather of every step's transforms, barrier + max-over-ranks
    timing, rank 0 printing the single JSON line) is the code path of the 8-GPU run.  The launcher is started from the
    forkserver (conftest.py: a process that never touched the GPU), not forked from this one."""
    import multiprocessing as mp
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device",
           "--steps", "2", "--warmup", "1", "--pairs", "8", "--points", "512", "--kernel", "26", "--iters", "40"]
    ctx = mp.get_context("forkserver")
    q = ctx.Queue()
    proc = ctx.Process(target=_launch, args=(cmd, ROOT, env, q))
    proc.start()
    rc, out, err = q.get(timeout=700)
    proc.join(timeout=60)
    assert rc == 0, err
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]                             # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2
    assert abs(d["value"] - 2 * 8 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]      # whole-job pairs/s: both ranks' pairs
    assert "dp2" in d["config"]["parallelism"] and d["config"]["pairs_per_gpu"] == 8
    assert "brute_force" not in d and "cpu_baseline" not in d and "chamfer_op" not in d     # N = 1 extras stay off
    assert d["quality"]["hypothesis_iterations_per_pair"] >= 26 * 40
    rk = d["ranks"]                                                  # an N > 1 record explains its own efficiency
    assert len(rk["per_rank_s"]) == 2 and len(rk["per_rank_retried_pairs"]) == 2 and rk["gather_ms"]["max"] >= rk["gather_ms"]["mean"] > 0
    assert 0 < rk["load_balance_mean_over_max"] <= 1.0
