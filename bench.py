#!/usr/bin/env python3
"""bench.py -- HOUV registration throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one full `solve_model` pass (registration/models/houv.py:142-206: base-0 stage of K=64 restarts x 200
Adam iterations, then the data-dependent retry stages at bases 1..3) over one batch of synthetic MVP-shaped pairs
resident in HBM.  N=1 workload = BASELINE.json configs[1]: 2048x2048-point pairs, batch 256.  For N>1 every rank
solves its own batch of the same size (weak scaling; pairs are independent, no data-path collective) and the ranks
exchange the per-pair (R,t) of all timed steps with ONE RCCL all-gather inside the timed region.  Rank 0 prints one JSON line.

Extra objects on the line:
  roofline      the dominant kernel (houv::solve_kernel, brute-force sweep), timed live with HIP events on its launch stream;
                VALU-issue bound: `frac` is the share of the chip's VALU issue slots the two sweeps occupy
  pruned        the same batches through the opt-in EXACT pruned search (houv_solve_iterate_pruned): pairs/s, us per
                hypothesis-iteration, and whether every transform came out bit-identical to the brute-force run
  chamfer_op    the stand-alone Chamfer op at the same cloud size (BASELINE metric's "Chamfer HBM GB/s" half)
  cpu_baseline  the CPU oracle (oracle/houv_ref_cpu.py, the reference's PyTorch-CPU formulation) on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3     # MI355X fp32 vector == fp32-input MFMA peak (MI355X_MICROARCH.md, chip-level table)
HBM_PEAK_GBPS = 8000.0
FLOP_PER_EVAL = 8            # 3 sub + 1 mul + 2 fma per squared distance
PEAK_CLOCK_HZ = 2.4e9        # the clock the 157.3 TFLOP/s peak is quoted at: 1024 SIMDs x 64 lanes x 2 flop / 2 clk
N_SIMD = 1024
# VALU issue slots (one slot = one full-rate wave64 instruction = 2 clk of a SIMD) the sweep spends per point pair and wave:
#   4-metric sweep: 3 v_sub + 2 v_mul + 4 v_fma = 9 slots, 2 v_min3 at half rate = 4 slots, sub-tile tracking
#                   (v_cmp + v_cndmask per query, metric and 32 references, both half rate) 0.5              -> 13.5
#   1-metric sweep: 3 v_sub + 1 v_mul + 2 v_fma = 6 slots, 1/2 v_min3 = 1 slot, tracking 0.125              ->  7.125
# Only v_add/sub/mul/fma/mov issue at one wave64 instruction per 2 clk on gfx950; v_min/max/min3/med3, v_cmp, v_cndmask and
# the integer ops take 4 (scripts/ubench/{valu_rate,misc_rate}.hip: profiles/r01_valu_rate.txt, r02_instr_rates.txt)
SLOTS_PER_PAIR = {True: 13.5, False: 7.125}
EXEC_FLOP_PER_PAIR = {True: 13.0, False: 8.0}     # flops the fused sweep really executes per point pair (mins not counted)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=256, help="pairs per GPU per step (BASELINE cfg2: 256)")
    ap.add_argument("--points", type=int, default=2048)
    ap.add_argument("--kernel", type=int, default=64, help="restarts per pair (houv.py:142 default)")
    ap.add_argument("--iters", type=int, default=200, help="Adam iterations per stage (houv.py:142 default)")
    ap.add_argument("--dcp", action="store_true",
                    help="BASELINE configs[4]: DCP feature head (fp32 MFMA GEMMs, random-init weights) + HOUV loss of its answer")
    ap.add_argument("--icp", action="store_true", help="BASELINE configs[3]: ICP refinement (threshold 0.02, <=500 its) after HOUV")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo = CPU rehearsal of the N>1 control path")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--solver", default="brute", choices=["brute", "pruned"],
                    help="brute = the brute-force sweep north_star specifies (default, what the roofline is defined on); "
                         "pruned = opt-in exact search (bit-identical outputs on the same clouds, Morton-sorted inputs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pruned", action="store_true", help="skip the `pruned` leg (same batches through the exact pruned search)")
    ap.add_argument("--no-chamfer-op", action="store_true")
    return ap.parse_args()


def cpu_baseline(points, kernel, iters, inst_iters_per_pair):
    """Time the oracle's predict_model (float64 expanded-form Chamfer + autograd + torch Adam = the reference's
    PyTorch-CPU path) on 1 pair x 8 restarts... see `sample`; scale linearly in hypothesis-iterations (cost is
    exactly linear in them) to the work the GPU run did per pair."""
    from houv_amd import synthetic
    from oracle import houv_ref_cpu as orc
    # the 1-GPU box grants this job 16 host cores (oversubscribing all visible cores is slower)
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    src, tgt, _ = synthetic.make_pairs(1, points, seed=4242)
    k_s, it_s = 26, 1                      # 26 = the fewest restarts reset_weight accepts (houv.py:47-51)
    orc.predict_model(src, tgt, kernel=k_s, num_epochs=1)          # warm-up (allocator, threads)
    t0 = time.time()
    n_done = 0
    while time.time() - t0 < 12.0:
        orc.predict_model(src, tgt, kernel=k_s, num_epochs=it_s)
        n_done += k_s * it_s
    dt = time.time() - t0
    per = dt / n_done
    return {"value": 1.0 / (per * inst_iters_per_pair), "unit": "pairs/s", "cores": cores,
            "kind": "port (oracle restatement: oracle/houv_ref_cpu.py, pinned bit for bit to the reference's own outputs)",
            "sample": f"{n_done} hypothesis-iterations of oracle.predict_model at {points}x{points} points "
                      f"({dt:.1f} s, {per:.3f} s each), scaled linearly to the {inst_iters_per_pair:.0f} "
                      "hypothesis-iterations per pair the GPU run executed",
            "seconds_per_hypothesis_iteration": per}


def pmc_traffic(wg_iters_per_launch, points):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (bench.py cannot run
    under the profiler itself): profiles/r02_pmc_traffic.json holds FETCH_SIZE / WRITE_SIZE per workgroup-iteration,
    collected in separate --pmc runs and corrected as MI355X_MICROARCH.md prescribes; scaled to this run's launches."""
    path = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")
    if points != 2048 or not os.path.exists(path):
        return None, "no PMC pass for this configuration (profiles/r02_pmc_traffic.json covers 2048-point clouds)"
    with open(path) as f:
        t = json.load(f)
    per = t["fetch_bytes_per_wg_iter"] + t["write_bytes_per_wg_iter"]
    return per * wg_iters_per_launch, t["note"]


def pruned_leg(args, dev, batches, timed, brute_answers, net):
    """The same batches through the opt-in exact pruned search: at most the first five timed batches are compared bit for
    bit with the brute-force answers (that pass doubles as warm-up) and then timed (repeated until >= 5 steps), so the
    leg stays ~30 s whatever --steps is."""
    from houv_amd import solver
    from houv_amd.models.houv import predict_model
    if args.points > 2048:
        return {"skipped": "the pruned search serves clouds of <= 2048 points (64-bit visit masks)"}

    def solve(s, t):
        from houv_amd.models.houv import HOUV
        ans, _, _ = solver.best_of_k_with_retry(
            lambda ss, tt, base: predict_model(net if base in (0, 3) else HOUV.blank_like(net), ss, tt, kernel=args.kernel,
                                               num_epochs=args.iters, angle_base=base), s, t)
        return ans
    old, solver.PRUNED = solver.PRUNED, True
    try:
        solver.LAUNCH_LOG = []
        timed = timed[:5]
        identical = all(bool(torch.equal(solve(batches[b][0], batches[b][1]), brute_answers[b])) for b in timed)   # + warm-up
        reps = max(1, -(-5 // len(timed)))
        solver.LAUNCH_LOG = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for b in timed:
                solve(batches[b][0], batches[b][1])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        log, solver.LAUNCH_LOG = solver.LAUNCH_LOG, None
    finally:
        solver.PRUNED = old
    log = [e for e in log if e[2] == args.pairs * args.kernel]          # base-stage launches (retry stages overlap each other)
    k_ms = sum(e0.elapsed_time(e1) for e0, e1, *_ in log)
    inst_iters = sum(n * it for _, _, n, it, *_ in log)
    steps = reps * len(timed)
    return {"solver": "houv_solve_iterate_pruned (exact: previous-NN upper bound + sub-tile bounding boxes)",
            "value": args.pairs * steps / dt, "unit": "pairs/s", "steps": steps, "ms_per_step": dt * 1e3 / steps,
            "us_per_hypothesis_iteration": k_ms * 1e3 / max(inst_iters, 1),
            "bit_identical_to_brute_force": identical,
            "batches_compared": len(timed),
            "compared": "ans[P,4,4] of the first (up to five) timed batches, torch.equal against the brute-force run of the same batch"}


def chamfer_op_probe(dev, points):
    """The stand-alone Chamfer op (houv_chamfer_forward) at the bench's cloud size, B' = 4096 instances.
    Timed as a train of back-to-back launches after a warm-up: inside an isolated 3-ms launch the clock is still ramping
    (GRBM_GUI_ACTIVE: 1.9 GHz there, 2.3 GHz in the 1.4-s solve launches)."""
    from houv_amd import ops
    B, reps = 4096, 30
    a = torch.rand(B, points, 3, device=dev) - 0.5
    b = torch.rand(B, points, 3, device=dev) - 0.5
    d1 = torch.empty(B, points, device=dev); d2 = torch.empty_like(d1)
    i1 = torch.empty(B, points, dtype=torch.int32, device=dev); i2 = torch.empty_like(i1)
    for _ in range(10):
        ops.chamfer_forward(a, b, d1, d2, i1, i2)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.chamfer_forward(a, b, d1, d2, i1, i2)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3 / reps)
    t = float(np.min(ts))
    flops = 2.0 * B * points * points * FLOP_PER_EVAL
    byts = B * (2 * points * 12 + 2 * points * 8)            # inputs 2*N*12 B + outputs 2*N*8 B
    direct = os.environ.get("HOUV_CHAMFER_DIRECT") == "1"
    return {"kernel": "houv::chamfer_nn_kernel<4> (direct sweep)" if direct else
            "houv::chamfer_nn_filter_kernel<8> (expanded-form filter + exact recovery, bit-exact)",
            "batch": B, "points": points, "ms_per_launch": t * 1e3, "launches_timed": 3 * reps,
            "point_pairs_per_s": 2.0 * B * points * points / t,
            "tflops_8flop_accounting": flops / t / 1e12, "frac_fp32_peak_8flop_accounting": flops / t / 1e12 / FP32_PEAK_TFLOPS,
            "algorithmic_GBps": byts / t / 1e9, "frac_hbm_peak": byts / t / 1e9 / HBM_PEAK_GBPS,
            "note": "VALU-issue bound; the 8-flop accounting prices the direct-difference form -- the filtered sweep executes "
                    "3 FMAs per pair plus the exact recovery (DESIGN.md 3.2); the HBM figure is reported because BASELINE's "
                    "metric asks for it (a brute-force 2048^2 sweep cannot exceed ~2.4 % of HBM peak)"}


def bench_dcp(args, dev, world, rank):
    """configs[4]: one step = DCP forward (DGCNN + Transformer pointer + SVD head) over the batch, then the HOUV robust
    Chamfer loss (Predict_loss, houv.py:209-222) of the source cloud moved by DCP's answer.  Weights are random-init
    (the reference ships no checkpoint), so only throughput and the loss plumbing are meaningful, not accuracy."""
    from houv_amd import ops, synthetic
    from houv_amd.models.dcp import Model
    from houv_amd.models.houv import Predict_loss
    P = args.pairs
    torch.manual_seed(2021)
    net = Model(None, pairs_per_chunk=int(os.environ.get("HOUV_DCP_CHUNK", 32))).to(dev)
    batches = []
    for b in range(args.steps + args.warmup):
        s, t, _ = synthetic.make_pairs(P, args.points, seed=2021, first_id=(b * world + rank) * P)
        batches.append((s.to(dev), t.to(dev)))
    losses = []

    def step(b):
        s, t = batches[b]
        T12 = net(s, t)
        moved = torch.bmm(s, T12[:, :3, :3].transpose(1, 2)) + T12[:, :3, 3].unsqueeze(1)
        with torch.no_grad():
            loss, min_1 = Predict_loss(moved, t)
        losses.append(float(loss.mean()))

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        step(w)
    ops.GEMM_LOG = []
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    sync()
    dt = time.perf_counter() - t0
    log, ops.GEMM_LOG = ops.GEMM_LOG, None
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    g_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in log)
    g_fl = sum(f for _, _, f in log)
    out = {
        "metric": "registration pairs/sec (2048-pt partial pairs)", "value": P * world * args.steps / dt,
        "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"DCP feature head (random-init weights) + HOUV Predict_loss, {args.points}-pt pairs, "
                               f"batch {P}/GPU (BASELINE configs[4])", "pairs_per_gpu": P, "points": args.points,
                   "parallelism": f"dp{world}"},
        "quality": {"mean_houv_loss_of_dcp_answer": float(np.mean(losses[-args.steps:]))},
        "roofline": {"kernel": "houv::gemm_f32_kernel + houv::attention_f32_kernel (v_mfma_f32_32x32x2_f32)", "bound": "mfma",
                     "achieved": g_fl / (g_ms * 1e-3) / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": g_fl / (g_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, "traffic": None, "launches": len(log),
                     "avg_launch_ms": g_ms / max(len(log), 1), "kernel_time_share": g_ms * 1e-3 / dt},
    }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU fallback)"
    if args.single_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from houv_amd import distributed as hd
    from houv_amd import solver, synthetic
    from houv_amd.models.houv import HOUV

    if args.dcp:
        return bench_dcp(args, dev, world, rank)
    P = args.pairs
    n_total = P * world
    solver.PRUNED = args.solver == "pruned"
    # synthetic MVP-shaped pairs, a different slice per rank and per step; resident in HBM before timing starts
    n_batches = args.steps + args.warmup
    batches = []
    for b in range(n_batches):
        s, t, pose = synthetic.make_pairs(P, args.points, seed=2021, first_id=(b * world + rank) * P)
        # the order of the points of a cloud carries no meaning: sort both clouds along a Morton curve once, outside
        # the timed region, so that the brute-force run and the `pruned` leg (which needs spatially compact
        # sub-tiles) see IDENTICAL inputs and their outputs can be compared bit for bit
        batches.append((solver.morton_sort(s.to(dev)), solver.morton_sort(t.to(dev)), pose.to(dev)))
    net = HOUV(P * args.kernel, 0).to(dev)
    results = []

    from houv_amd.models.houv import predict_model

    def solve_on_device(s, t):
        # solve_model (houv.py:142-206) without its final .cpu()/print: transforms stay in HBM for the all-gather
        ans, _, _ = solver.best_of_k_with_retry(
            lambda ss, tt, base: predict_model(net if base in (0, 3) else HOUV.blank_like(net), ss, tt, kernel=args.kernel,
                                               num_epochs=args.iters, angle_base=base), s, t)
        return ans

    def step(b):
        s, t, _ = batches[b]
        ans = solve_on_device(s, t)
        if args.icp:
            from houv_amd.icp import icp_refine
            ans = icp_refine(s, t, ans)
            ans[:, 3, :] = 0.0                                               # keep the results layout of houv.py:187-195
        results.append((b, ans))

    def gather_all():
        """The path's single collective (north_star: "a single RCCL all-gather of per-pair (R,t)"): every rank's
        transforms of ALL its steps in one [steps*P, 12] all-gather -- no per-step synchronisation between ranks,
        exactly like the reference, whose shards only meet in the final --combine (run_test.sh:21-23)."""
        if world == 1 or not results:
            return [a for _, a in results]
        mine = torch.cat([a for _, a in results], 0)
        full = hd.gather_transforms(mine if args.backend == "nccl" else mine.cpu(), mine.shape[0] * world).to(dev)
        per_rank = full.reshape(world, len(results), P, 4, 4)
        return [per_rank[:, i].reshape(world * P, 4, 4) for i in range(len(results))]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    solver.LAUNCH_LOG = []
    for w in range(args.warmup):
        step(w)
    gather_all()
    results.clear()
    n_warm_launches = len(solver.LAUNCH_LOG)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    gathered = gather_all()
    sync()
    dt = time.perf_counter() - t0
    log_all, solver.LAUNCH_LOG = solver.LAUNCH_LOG, None
    log = log_all[n_warm_launches:]
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- accuracy of what was just timed (sanity, rank-local batch of the last step) ----
    from houv_amd.train_utils import rotation_error, translation_error
    b_last = results[-1][0]
    full = gathered[-1]
    pose = batches[b_last][2]
    mine = full[rank * P:(rank + 1) * P] if world > 1 else full
    r_err = rotation_error(mine[:, :3, :3], pose[:, :3, :3])
    t_err = translation_error(mine[:, :3, 3], pose[:, :3, 3])

    # ---- dominant kernel: live HIP-event timing on the launch stream ----
    from houv_amd import _lib
    # the retry stages of a step run concurrently on side streams (solver.CONCURRENT_RETRIES): their launches overlap
    # each other, so the kernel's roofline is taken from the base-stage launches (all P*K hypotheses, alone on the GPU),
    # which are 88 % of the hypothesis-iterations; `kernel_time_share` counts every launch's hypothesis-iterations
    all_inst_iters = sum(n * it for _, _, n, it, *_ in log)
    log_retry = [e for e in log if e[2] != P * args.kernel]
    log = [e for e in log if e[2] == P * args.kernel]
    k_ms = sum(e0.elapsed_time(e1) for e0, e1, *_ in log)
    inst_iters = sum(n * it for _, _, n, it, *_ in log)
    evals = sum(n * it * (8.0 if v else 2.0) * N * M for _, _, n, it, N, M, v in log)   # (4 metrics | 1) x 2 directions x N x M
    flops_alg = evals * FLOP_PER_EVAL
    slots = sum(n * it * 2.0 * N * M / 64.0 * SLOTS_PER_PAIR[bool(v)] for _, _, n, it, N, M, v in log)   # wave-level issue slots
    flops_exec = sum(n * it * 2.0 * N * M * EXEC_FLOP_PER_PAIR[bool(v)] for _, _, n, it, N, M, v in log)
    secs = k_ms * 1e-3
    frac_valu = slots / (secs * PEAK_CLOCK_HZ / 2.0 * N_SIMD)          # a SIMD offers one issue slot per 2 clk
    # the kernel the log's launches ran: template arguments from the variant table and the launch's view flag
    kinds = sorted({(_lib.solve_variant(N, M, args.solver == "pruned"), 4 if v else 1) for *_, N, M, v in log})
    kname = ", ".join("houv::solve_kernel<%d, %d, %d, %s, 1>" % (b, q, nm, "true" if args.solver == "pruned" else "false")
                      for (b, q), nm in kinds)
    wg_iters_per_launch = inst_iters / max(len(log), 1)
    traffic, traffic_note = pmc_traffic(wg_iters_per_launch, args.points)
    roofline = {
        "kernel": kname, "bound": "valu",
        "bound_note": "compute bound on the fp32 VALU ISSUE rate (no MFMA, ~0 algorithmic HBM bytes: 2048^2 brute force has "
                      "~820 flop/byte). `achieved` counts every VALU issue slot the two sweeps need as one FMA slot "
                      "(64 lanes x 2 flop; half-rate v_min3 = 2 slots), so achieved/peak = the share of the chip's "
                      "issue slots at the 2.4 GHz peak clock that the sweeps occupy; the chip sustains ~2.2 GHz under "
                      "this load, i.e. the same work fills ~9 % more of the slots actually offered",
        "achieved": frac_valu * FP32_PEAK_TFLOPS, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": frac_valu,
        "executed_tflops": flops_exec / secs / 1e12,                    # fp32 flops really executed (13 per 4-metric point pair)
        "achieved_algorithmic": flops_alg / secs / 1e12,                # SURVEY 8(d) accounting: 8 separate 8-flop sweeps
        "frac_algorithmic": flops_alg / secs / 1e12 / FP32_PEAK_TFLOPS,
        "frac_algorithmic_note": "credits the fused 4-metric sweep as four separate 8-flop evaluations (SURVEY.md 8d); it "
                                 "says how much work a non-fused formulation would have needed, not how busy the ALUs are",
        "traffic": traffic, "traffic_note": traffic_note,
        "launches": len(log), "avg_launch_ms": k_ms / max(len(log), 1),
        # all launches of the process (warm-up included) = what `rocprofv3 --kernel-trace --stats` averages over
        "launches_incl_warmup": len(log_all),
        "avg_launch_ms_incl_warmup": sum(e0.elapsed_time(e1) for e0, e1, *_ in log_all) / max(len(log_all), 1),
        "us_per_hypothesis_iteration": k_ms * 1e3 / max(inst_iters, 1),
        "launches_note": "base-stage launches (P*K hypotheses each); %d retry-stage launches ran concurrently on side "
                         "streams and are not in these sums" % len(log_retry),
        "kernel_time_share": secs / dt,
        "hypothesis_iterations_share": inst_iters / max(all_inst_iters, 1),
    }
    if args.solver == "pruned":
        roofline["pruned_note"] = ("--solver pruned: `achieved` still prices the brute-force sweep's issue slots, so frac > 1 "
                                   "only says that evaluations were provably skipped; it is not a roofline claim")
    out = {
        "metric": "registration pairs/sec (2048-pt partial pairs)", "value": n_total * args.steps / dt,
        "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"HOUV solve_model, {args.points}x{args.points}-pt pairs, batch {P}/GPU, "
                               f"K={args.kernel} restarts x {args.iters} Adam iterations + retry stages "
                               "(BASELINE configs[1])",
                   "pairs_per_gpu": P, "points": args.points, "kernel": args.kernel, "iters": args.iters,
                   "icp_refine": bool(args.icp), "solver": args.solver,
                   "parallelism": f"dp{world} (pair shards, one {'RCCL' if args.backend == 'nccl' else 'gloo'} "
                                  "all-gather of [steps*P,12] per rank)"},
        "quality": {"mean_rot_err_deg": float(r_err.mean()), "median_rot_err_deg": float(r_err.median()),
                    "mean_trans_err": float(t_err.mean()),
                    "hypothesis_iterations_per_pair": all_inst_iters / (P * args.steps)},
        "roofline": roofline,
    }
    if rank == 0 and world == 1:
        if args.solver == "brute" and not args.no_pruned and not args.icp:
            timed = [b for b, _ in results]
            out["pruned"] = pruned_leg(args, dev, batches, timed, {b: a for b, a in results}, net)
            out["pruned"]["speedup_over_brute_force"] = out["pruned"]["value"] / out["value"]
        if not args.no_chamfer_op:
            out["chamfer_op"] = chamfer_op_probe(dev, args.points)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.points, args.kernel, args.iters, all_inst_iters / (P * args.steps))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
