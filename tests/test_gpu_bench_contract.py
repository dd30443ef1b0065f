"""The driver's bench contract, exercised on a miniature workload in-process (bench.py is a script: argv in, ONE JSON line
out): field names / types, the roofline and pruned objects, and internal consistency of the numbers."""
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
    d = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--pairs", "8", "--points", "512", "--kernel", "26",
              "--iters", "40", "--no-cpu-baseline", "--no-chamfer-op"], monkeypatch)
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict)):
        assert isinstance(d[key], typ), key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 8 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]          # pairs/s = pairs / time
    r = d["roofline"]
    assert r["bound"] == "valu" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1.0      # physical: cannot exceed the issue slots
    assert "solve_kernel<256, 2, 4, false, 1>" in r["kernel"]                                # derived from the launches, not hard-coded
    assert r["launches"] > 0 and r["kernel_time_share"] <= 1.0 and "traffic" in r
    p = d["pruned"]
    assert p["bit_identical_to_brute_force"] is True and p["unit"] == "pairs/s" and p["steps"] >= 5


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
    assert "pruned" not in d and "cpu_baseline" not in d and "chamfer_op" not in d          # N = 1 extras stay off
    assert d["quality"]["hypothesis_iterations_per_pair"] >= 26 * 40
