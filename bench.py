#!/usr/bin/env python3
"""bench.py -- HOUV registration throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one full `solve_model` pass (registration/models/houv.py:142-206: base-0 stage of K=64 restarts x 200
Adam iterations, then the data-dependent retry stages at bases 1..3) over one batch of synthetic MVP-shaped pairs
resident in HBM.  N=1 workload = BASELINE.json configs[1]: 2048x2048-point pairs, batch 256.  For N>1 every rank
solves its own batch of the same size (weak scaling; pairs are independent, no data-path collective) and the ranks
exchange the per-pair (R,t) of all timed steps with ONE RCCL all-gather inside the timed region.  Rank 0 prints one JSON line.

`value` = the DEFAULT product path: the exact pruned nearest-neighbour search (houv_solve_iterate_pruned; --solver brute times the
brute-force sweep instead).  Extra objects on the line:
  roofline      the dominant kernel of the timed run (houv::solve_kernel<512, 4, 4, 2, 1>), timed live with HIP events on its launch
                stream; `frac` = EXECUTED fp32 flops of the nearest-neighbour evaluations / 157.3 TFLOP/s (see `frac_definition`);
                `pruned_search` = what the search visited; `sustained_clock_ghz` = measured inside the kernel; `traffic` and the
                `valu_*` figures = the committed PMC pass, printed only when it profiled the loaded library build
  brute_force   EVERY timed batch again through the brute-force sweep (houv_solve_iterate): pairs/s, whether every transform came out
                bit-identical to the timed run, and that kernel's own roofline (+ the VALU issue-slot model)
  ranks         N > 1 (or --force-process-group): per-rank seconds, retried pairs, gather time, load balance
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

BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak (MI355X_MICROARCH.md, Matrix cores)
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
    ap.add_argument("--solver", default="pruned", choices=["brute", "pruned"],
                    help="pruned = the product default: exact pruned search, bit-identical to the brute-force sweep on the same "
                         "clouds; brute = the brute-force sweep north_star specifies (then the other leg is `pruned`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-solver", action="store_true",
                    help="skip the leg that re-runs every timed batch through the other search (bit-identity + its roofline)")
    ap.add_argument("--force-process-group", action="store_true",
                    help="initialise torch.distributed even at world size 1, so that the RCCL all-gather path runs on one GPU")
    ap.add_argument("--no-chamfer-op", action="store_true")
    return ap.parse_args()


def cpu_baseline(points, kernel, iters, inst_iters_per_pair):
    """Time the oracle's predict_model (float64 expanded-form Chamfer + autograd + torch Adam = the reference's
    PyTorch-CPU path) on TWO pairs x 26 restarts for >= 30 s; scale linearly in hypothesis-iterations (cost is
    exactly linear in them) to the work the GPU run did per pair."""
    from houv_amd import synthetic
    from oracle import houv_ref_cpu as orc
    # the 1-GPU box grants this job 16 host cores (oversubscribing all visible cores is slower)
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    src, tgt, _ = synthetic.make_pairs(2, points, seed=4242)
    k_s, it_s = 26, 1                      # 26 = the fewest restarts reset_weight accepts (houv.py:47-51)
    orc.predict_model(src[:1], tgt[:1], kernel=k_s, num_epochs=1)          # warm-up (allocator, threads)
    t0 = time.time()
    n_done, per_pair = 0, [0, 0]
    while time.time() - t0 < 30.0 or min(per_pair) == 0:
        i = int(per_pair[1] < per_pair[0])
        orc.predict_model(src[i:i + 1], tgt[i:i + 1], kernel=k_s, num_epochs=it_s)
        per_pair[i] += k_s * it_s
        n_done += k_s * it_s
    dt = time.time() - t0
    per = dt / n_done
    return {"value": 1.0 / (per * inst_iters_per_pair), "unit": "pairs/s", "cores": cores,
            "kind": "port (oracle restatement: oracle/houv_ref_cpu.py, pinned bit for bit to the reference's own outputs)",
            "sample": f"{n_done} hypothesis-iterations of oracle.predict_model over 2 pairs at {points}x{points} points "
                      f"({per_pair[0]} + {per_pair[1]}; {dt:.1f} s, {per:.3f} s each), scaled linearly to the "
                      f"{inst_iters_per_pair:.0f} hypothesis-iterations per pair the GPU run executed",
            "seconds_per_hypothesis_iteration": per}


def committed_pmc(kernel_name, points):
    """The committed rocprofv3 PMC passes of the fused loop (bench.py cannot run under the profiler itself):
    profiles/r03_pmc_solve.json holds, per kernel instantiation, the counters of separate --pmc runs at the bench's launch
    shape together with the houv_build_id() of the library they profiled.  Returns (entry | None, note): the numbers are
    only passed on when that build id is the loaded library's -- a profile of other code is not this run's traffic."""
    from houv_amd import _lib
    path = os.path.join(ROOT, "profiles", "r03_pmc_solve.json")
    if not os.path.exists(path):
        return None, "no committed PMC pass (profiles/r03_pmc_solve.json)"
    with open(path) as f:
        t = json.load(f)
    if t.get("build_id") != _lib.build_id():
        return None, (f"committed PMC pass profiled library build {t.get('build_id')}, the loaded library is "
                      f"{_lib.build_id()}: not reported")
    e = t.get("kernels", {}).get(kernel_name)
    if e is None or t.get("points") != points:
        return None, f"profiles/r03_pmc_solve.json has no pass for {kernel_name} at {points} points"
    return e, (f"profiles/r03_pmc_solve.json (committed rocprofv3 --pmc passes of library build {t['build_id']}, git "
               f"{t.get('git_head', '?')}; NOT measured in this run): separate FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE "
               "doubled per profiles/r02_pmc_calib.txt, per workgroup-iteration, scaled by this run's launches")


def solve_stats_pass(solve, batch):
    """One more base stage of one batch with the library's counters on (houv_debug_set("solve_stats")), OUTSIDE the timed
    region: sub-tile visits the pruned sweeps asked for / executed, and the shader clock the chip sustained inside the
    kernel (s_memtime over s_memrealtime, two stamps per workgroup launch)."""
    from houv_amd import _lib
    buf = torch.zeros(8, dtype=torch.int64, device=batch[0].device)
    _lib.debug_set("solve_stats", buf.data_ptr())
    try:
        solve(batch[0], batch[1])
        torch.cuda.synchronize()
    finally:
        _lib.debug_set("solve_stats", 0)
    v = [int(x) for x in buf.cpu()]
    return {"asked": v[0], "steps": v[1], "pruned_wave_sweeps": v[2], "brute_wave_sweeps": v[3],
            "clock_ghz": (v[4] / v[5] * 0.1) if v[5] else None}


def kernel_roofline(log, log_all, dt, all_inst_iters, n_retry_launches, stats, points):
    """`roofline` of the solve_kernel instantiation whose base-stage launches are in `log` (HIP events on the launch stream).
    VALU bound (no MFMA, ~0 algorithmic HBM bytes).  frac = executed fp32 flops of the nearest-neighbour evaluations / peak."""
    from houv_amd import _lib
    pruned = bool(log[0][7]) if log else False
    k_ms = sum(e0.elapsed_time(e1) for e0, e1, *_ in log)
    secs = k_ms * 1e-3
    inst_iters = sum(n * it for _, _, n, it, *_ in log)
    pairs_brute = sum(n * it * 2.0 * N * M for _, _, n, it, N, M, *_ in log)          # point pairs of two full sweeps
    views = bool(log[0][6]) if log else True
    flop_pp = EXEC_FLOP_PER_PAIR[views]
    kinds = sorted({(_lib.solve_variant(N, M, pruned, with_mode=True), 4 if v else 1) for _, _, _, _, N, M, v, *_ in log})
    kname = ", ".join("houv::solve_kernel<%d, %d, %d, %d, 1>" % (b, q, nm, mode) for (b, q, mode), nm in kinds)
    r = {"kernel": kname, "bound": "valu", "unit": "TFLOP/s", "peak": FP32_PEAK_TFLOPS}
    if pruned and stats and stats["pruned_wave_sweeps"]:
        # executed point pairs: every sub-tile a lane asked for = 32 evaluations (+ the brute-force first iteration of a stage)
        q = max(kinds[0][0][1], 1)                                          # queries (= sub-tile lists) per lane
        tiles_per_query = stats["asked"] / (stats["pruned_wave_sweeps"] * 64.0 * q)
        steps_per_wave_sweep = stats["steps"] / float(stats["pruned_wave_sweeps"])
        ntile = -(-points // (32 if points <= 2048 else 64))                 # visit-mask bits: sub-tiles, or super-tiles above 2048 points
        # the first iteration of a stage has no remembered neighbours and sweeps everything: one iteration of every first launch
        brute_iters = sum(n for _, _, n, _, _, _, _, _, first in log if first)
        share_pruned = 1.0 - brute_iters / float(max(inst_iters, 1))
        useful_share = share_pruned * tiles_per_query / ntile + (1.0 - share_pruned)
        pairs_exec = pairs_brute * useful_share
        r["pruned_search"] = {
            "sub_tiles_asked_per_query_and_sweep": tiles_per_query,
            "sub_tiles_per_cloud": ntile,
            "wave_steps_per_sweep": steps_per_wave_sweep,
            "lane_utilisation_of_the_walk": stats["asked"] / (stats["steps"] * 64.0) if stats["steps"] else None,
            "share_of_point_pairs_evaluated": useful_share,
            "source": "houv_debug_set(\"solve_stats\") counters of one extra base stage outside the timed region"}
    else:
        pairs_exec = pairs_brute
    flops_exec = pairs_exec * flop_pp
    flops_alg = sum(n * it * (8.0 if v else 2.0) * N * M for _, _, n, it, N, M, v, *_ in log) * FLOP_PER_EVAL
    r["achieved"] = flops_exec / secs / 1e12
    r["frac"] = r["achieved"] / FP32_PEAK_TFLOPS
    r["frac_definition"] = ("executed fp32 flops of the nearest-neighbour evaluations (13 per 4-metric point pair, 8 per single-"
                            "metric pair; min / compare / select / list-walk instructions are not flops) / 157.3 TFLOP/s fp32 vector peak"
                            + ("; the pruned search evaluates only the sub-tiles its bounds cannot exclude, so this is NOT the share "
                               "of a brute-force sweep's work -- see pruned_search.share_of_point_pairs_evaluated" if pruned else ""))
    r["executed_tflops"] = r["achieved"]
    r["frac_executed_flops"] = r["frac"]
    r["achieved_algorithmic"] = flops_alg / secs / 1e12
    r["frac_algorithmic"] = r["achieved_algorithmic"] / FP32_PEAK_TFLOPS
    r["frac_algorithmic_note"] = ("SURVEY.md 8(d) accounting: 8 separate brute-force sweeps of 8 flop per evaluation per hypothesis-"
                                  "iteration; says how much work the reference's formulation needs for the same result, not how busy "
                                  "the ALUs are (the fused 4-metric sweep shares dx,dy,dz; the pruned search skips evaluations)")
    if not pruned:
        slots = pairs_brute / 64.0 * SLOTS_PER_PAIR[views]
        clock = (stats or {}).get("clock_ghz") or PEAK_CLOCK_HZ / 1e9
        r["valu_issue_slot_model"] = {
            "slots_per_point_pair": SLOTS_PER_PAIR[views],
            "occupancy_at_sustained_clock": slots / (secs * clock * 1e9 / 2.0 * N_SIMD),
            "occupancy_at_2.4GHz": slots / (secs * PEAK_CLOCK_HZ / 2.0 * N_SIMD),
            "note": "a MODEL, not a measurement: issue slots the two sweeps need (full-rate op = 1 slot = 2 clk of a SIMD, half-rate "
                    "v_min3 / v_cmp / v_cndmask = 2) over the slots offered; round 2 reported the 2.4 GHz figure as `frac`"}
    r["sustained_clock_ghz"] = (stats or {}).get("clock_ghz")
    r["sustained_clock_source"] = ("s_memtime / s_memrealtime stamped around every workgroup's loop of one extra base stage "
                                   "(houv_debug_set(\"solve_stats\")), outside the timed region")
    wg_iters_per_launch = inst_iters / max(len(log), 1)
    pmc, pmc_note = committed_pmc(kname, points)
    if pmc is not None:
        r["traffic"] = (pmc["fetch_bytes_per_wg_iter"] + pmc["write_bytes_per_wg_iter"]) * wg_iters_per_launch
        # every VALU instruction takes >= 2 clk of its SIMD: a LOWER bound of the issue-slot occupancy that cannot exceed 1
        r["valu_clk_per_instruction"] = pmc["simd_clk_per_wg_iter"] / pmc["valu_insts_per_wg_iter"]
        r["valu_issue_occupancy_lower_bound"] = 2.0 / r["valu_clk_per_instruction"]
        r["pmc"] = {k: pmc[k] for k in pmc}
    else:
        r["traffic"] = None
    r["traffic_source"] = pmc_note
    r["library_build_id"] = _lib.build_id()
    r.update({"launches": len(log), "avg_launch_ms": k_ms / max(len(log), 1),
              "us_per_hypothesis_iteration": k_ms * 1e3 / max(inst_iters, 1),
              "launches_note": "base-stage launches (P*K hypotheses each); %d retry-stage launches ran concurrently on side "
                               "streams and are not in these sums" % n_retry_launches,
              "kernel_time_share": secs / dt, "hypothesis_iterations_share": inst_iters / max(all_inst_iters, 1)})
    if log_all is not None:
        # all launches of this kernel by the process so far (warm-up and the counters' extra stage included) = what
        # `rocprofv3 --kernel-trace --stats` averages over
        r["launches_incl_warmup"] = len(log_all)
        r["avg_launch_ms_incl_warmup"] = sum(e0.elapsed_time(e1) for e0, e1, *_ in log_all) / max(len(log_all), 1)
    return r


def other_solver_leg(args, dev, batches, timed, answers, make_solve, P):
    """Every timed batch once more through the OTHER search (brute force when the product default ran, and vice versa):
    torch.equal of every ans[P,4,4] against the timed run, and that search's own timing + roofline."""
    from houv_amd import solver
    other = not solver.PRUNED
    if not solver.uses_pruned(args.points, args.points, True):
        return {"skipped": "both searches run the brute-force kernel at this cloud size (the pruned search serves 257..2048 points)"}
    old, solver.PRUNED = solver.PRUNED, other
    try:
        solve = make_solve()
        solve(batches[timed[0]][0], batches[timed[0]][1])                    # warm-up
        solver.LAUNCH_LOG = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = [solve(batches[b][0], batches[b][1]) for b in timed]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        log, solver.LAUNCH_LOG = solver.LAUNCH_LOG, None
        identical = [bool(torch.equal(o, answers[b])) for o, b in zip(outs, timed)]
        stats = solve_stats_pass(lambda s, t: solver.run_stage(
            s, t, solver.houv_init_params(P * args.kernel), args.kernel, min(args.iters, 50), angle_base=0, trans_mode=0,
            use_views=True, f64_params=False, lr=0.01), batches[timed[-1]])
    finally:
        solver.PRUNED = old
    all_ii = sum(n * it for _, _, n, it, *_ in log)
    base = [e for e in log if e[2] == P * args.kernel]
    return {"solver": "houv_solve_iterate_pruned (exact pruned search)" if other else "houv_solve_iterate (brute-force sweep, north_star's formulation)",
            "value": P * len(timed) / dt, "unit": "pairs/s", "steps": len(timed), "ms_per_step": dt * 1e3 / len(timed),
            "bit_identical_to_timed_run": all(identical), "batches_compared": len(identical),
            "compared": "ans[P,4,4] of EVERY timed batch, torch.equal against the timed run of the same (spatially sorted) batch",
            "roofline": kernel_roofline(base, None, dt, all_ii, len(log) - len(base), stats, args.points)}


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
    return {"kernel": "houv::chamfer_nn_filter_kernel<8> (expanded-form filter + exact recovery, bit-exact)",
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
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)                    # the model replays each chunk's launches as one HIP graph (models/dcp.py)
    sync()
    dt = time.perf_counter() - t0
    # kernel times for `roofline`: ONE more step, outside the timed region, launched eagerly with an event pair around every
    # houv_gemm_f32 / houv_attention_f32 call (events cannot be recorded inside a graph replay)
    n_timed = len(losses)
    ops.GEMM_LOG = []
    step(args.warmup + args.steps - 1)
    sync()
    log, ops.GEMM_LOG = ops.GEMM_LOG, None
    del losses[n_timed:]
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
        # the GEMMs and the attention run on the bf16 matrix pipe with every fp32 product summed from SIX bf16 part products
        # (houv_split.h): the pipe executes 6 bf16 flops per useful fp32 flop, so `achieved` / `frac` price the executed work against
        # the dense bf16 peak and `useful_fp32_tflops` says what the model got out of it (the fp32-input MFMA peak is 157.3)
        "roofline": {"kernel": "houv::gemm_split_kernel<.., 6> + houv::attention_split_kernel (v_mfma_f32_32x32x16_bf16, six bf16 "
                               "part products per fp32 product)", "bound": "mfma",
                     "achieved": 6.0 * g_fl / (g_ms * 1e-3) / 1e12, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": 6.0 * g_fl / (g_ms * 1e-3) / 1e12 / BF16_PEAK_TFLOPS,
                     "useful_fp32_tflops": g_fl / (g_ms * 1e-3) / 1e12,
                     "useful_over_fp32_input_mfma_peak": g_fl / (g_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                     "note": "HIP-event time of every houv_gemm_f32 / houv_attention_f32 call of one step (the attention's split pre-pass included); "
                             "the short-K 1x1 convolutions of the DGCNN are HBM-bound inside it (DESIGN.md 9.2)",
                     "traffic": None, "launches": len(log),
                     "avg_launch_ms": g_ms / max(len(log), 1), "kernel_time_share": g_ms * 1e-3 / (dt / args.steps),
                     "kernel_times_from": "one extra eager step outside the timed region (the timed steps replay HIP graphs)",
                     "hip_graphs": bool(net.use_graphs and net._graphs)},
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
    if world > 1 or args.force_process_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
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
    solver.PRUNED = args.solver == "pruned"          # the product default (houv_amd.solver.PRUNED) unless --solver brute
    use_pruned = solver.uses_pruned(args.points, args.points)
    # synthetic MVP-shaped pairs, a different slice per rank and per step; resident in HBM before timing starts
    n_batches = args.steps + args.warmup
    batches = []
    for b in range(n_batches):
        s, t, pose = synthetic.make_pairs(P, args.points, seed=2021, first_id=(b * world + rank) * P)
        # The order of the points of a cloud carries no meaning.  The pruned search sorts both clouds into k-d leaves of 32 points
        # (solver.run_stage does it; sorting a sorted cloud is the identity); sorting them here, once, outside the timed
        # region makes the brute-force leg see IDENTICAL inputs, so that the two searches can be compared bit for bit
        leaf = solver.sort_leaf(args.points, args.points)
        batches.append((solver.spatial_sort(s.to(dev), leaf), solver.spatial_sort(t.to(dev), leaf), pose.to(dev)))
    results = []

    from houv_amd.models.houv import predict_model

    def make_solve():
        net = HOUV(P * args.kernel, 0).to(dev)

        def solve_on_device(s, t):
            # solve_model (houv.py:142-206) without its final .cpu()/print: transforms stay in HBM for the all-gather
            ans, _, retry = solver.best_of_k_with_retry(
                lambda ss, tt, base: predict_model(net if base in (0, 3) else HOUV.blank_like(net), ss, tt,
                                                   kernel=args.kernel, num_epochs=args.iters, angle_base=base), s, t)
            solve_on_device.retried += int(retry.numel())
            return ans
        solve_on_device.retried = 0
        return solve_on_device
    solve_on_device = make_solve()

    def step(b):
        s, t, _ = batches[b]
        ans = solve_on_device(s, t)
        if args.icp:
            from houv_amd.icp import icp_refine
            ans = icp_refine(s, t, ans)
            ans[:, 3, :] = 0.0                                               # keep the results layout of houv.py:187-195
        results.append((b, ans))

    gather_ms = [0.0]

    def gather_all():
        """The path's single collective (north_star: "a single RCCL all-gather of per-pair (R,t)"): every rank's
        transforms of ALL its steps in one [steps*P, 12] all-gather -- no per-step synchronisation between ranks,
        exactly like the reference, whose shards only meet in the final --combine (run_test.sh:21-23)."""
        if not dist.is_initialized() or not results:
            return [a for _, a in results]
        mine = torch.cat([a for _, a in results], 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        full = hd.gather_transforms(mine if args.backend == "nccl" else mine.cpu(), mine.shape[0] * world).to(dev)
        torch.cuda.synchronize()
        gather_ms[0] = (time.perf_counter() - t0) * 1e3
        per_rank = full.reshape(world, len(results), P, 4, 4)
        return [per_rank[:, i].reshape(world * P, 4, 4) for i in range(len(results))]

    def sync():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    solver.LAUNCH_LOG = []
    for w in range(args.warmup):
        step(w)
    gather_all()
    results.clear()
    solve_on_device.retried = 0
    n_warm_launches = len(solver.LAUNCH_LOG)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    torch.cuda.synchronize()
    my_solve_s = time.perf_counter() - t0                                  # this rank's own compute, before the collective
    gathered = gather_all()
    sync()
    dt = time.perf_counter() - t0
    log_all, solver.LAUNCH_LOG = solver.LAUNCH_LOG, None
    log = log_all[n_warm_launches:]
    per_rank = None
    if dist.is_initialized():
        cdev = dev if args.backend == "nccl" else "cpu"
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # self-diagnosis of an N > 1 record: where did each rank's time go? (stragglers can only come from the retry stages)
        mine = torch.tensor([my_solve_s, float(solve_on_device.retried), gather_ms[0],
                             float(sum(n * it for _, _, n, it, *_ in log))], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu().numpy()
        per_rank = {"per_rank_s": [float(x) for x in allr[:, 0]], "per_rank_retried_pairs": [int(x) for x in allr[:, 1]],
                    "per_rank_hypothesis_iterations": [float(x) for x in allr[:, 3]],
                    "gather_ms": {"max": float(allr[:, 2].max()), "mean": float(allr[:, 2].mean())},
                    "solve_s": {"max": float(allr[:, 0].max()), "mean": float(allr[:, 0].mean())},
                    "load_balance_mean_over_max": float(allr[:, 0].mean() / allr[:, 0].max()),
                    "note": "per_rank_s = a rank's own solve time before the collective; whole-job time = max over ranks "
                            "+ the all-gather; the collective moves steps*P*64 B per rank"}

    # ---- accuracy of what was just timed (sanity, rank-local batch of the last step) ----
    from houv_amd.train_utils import rotation_error, translation_error
    b_last = results[-1][0]
    full = gathered[-1]
    pose = batches[b_last][2]
    mine = full[rank * P:(rank + 1) * P] if world > 1 else full
    r_err = rotation_error(mine[:, :3, :3], pose[:, :3, :3])
    t_err = translation_error(mine[:, :3, 3], pose[:, :3, 3])

    # ---- dominant kernel: live HIP-event timing on the launch stream ----
    # the retry stages of a step run concurrently on side streams (solver.CONCURRENT_RETRIES): their launches overlap
    # each other, so the kernel's roofline is taken from the base-stage launches (all P*K hypotheses, alone on the GPU),
    # which are 88 % of the hypothesis-iterations; `kernel_time_share` counts every launch's hypothesis-iterations
    all_inst_iters = sum(n * it for _, _, n, it, *_ in log)
    log_base = [e for e in log if e[2] == P * args.kernel]
    stats = None
    if rank == 0:
        solver.LAUNCH_LOG = []
        stats = solve_stats_pass(lambda s, t: solver.run_stage(
            s, t, solver.houv_init_params(P * args.kernel), args.kernel, min(args.iters, 50), angle_base=0, trans_mode=0,
            use_views=True, f64_params=False, lr=0.01), batches[b_last])
        log_all = log_all + solver.LAUNCH_LOG          # the profiler sees this launch too (launches_incl_warmup)
        solver.LAUNCH_LOG = None
    roofline = kernel_roofline(log_base, log_all, dt, all_inst_iters, len(log) - len(log_base), stats, args.points)
    out = {
        "metric": "registration pairs/sec (2048-pt partial pairs)", "value": n_total * args.steps / dt,
        "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"HOUV solve_model, {args.points}x{args.points}-pt pairs, batch {P}/GPU, "
                               f"K={args.kernel} restarts x {args.iters} Adam iterations + retry stages "
                               "(BASELINE configs[1])",
                   "pairs_per_gpu": P, "points": args.points, "kernel": args.kernel, "iters": args.iters,
                   "icp_refine": bool(args.icp),
                   "solver": ("pruned: exact pruned nearest-neighbour search, the product default -- bit-identical to the "
                              "brute-force sweep on the same clouds (see `brute_force`)") if use_pruned else
                             "brute: the brute-force sweep of north_star's formulation",
                   "parallelism": f"dp{world} (pair shards, one {'RCCL' if args.backend == 'nccl' else 'gloo'} "
                                  "all-gather of [steps*P,12] per rank)"},
        "quality": {"mean_rot_err_deg": float(r_err.mean()), "median_rot_err_deg": float(r_err.median()),
                    "mean_trans_err": float(t_err.mean()),
                    "hypothesis_iterations_per_pair": all_inst_iters / (P * args.steps),
                    "note": "synthetic pairs, 20 % of them with up to 180 deg of relative rotation: not comparable with the "
                            "reference's MVP leaderboard figures"},
        "roofline": roofline,
    }
    if per_rank is not None:
        out["ranks"] = per_rank
    if rank == 0 and world == 1:
        if not args.no_other_solver and not args.icp:
            timed = [b for b, _ in results]
            leg = other_solver_leg(args, dev, batches, timed, {b: a for b, a in results}, make_solve, P)
            out["brute_force" if args.solver == "pruned" else "pruned"] = leg
            if "value" in leg:
                leg["ratio_timed_run_over_this"] = out["value"] / leg["value"]
        if not args.no_chamfer_op:
            out["chamfer_op"] = chamfer_op_probe(dev, args.points)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.points, args.kernel, args.iters, all_inst_iters / (P * args.steps))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()

if __name__ == "__main__":
    main()
