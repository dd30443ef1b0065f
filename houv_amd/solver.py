"""Host driver of the fused HOUV loop: parameter initialisation (host numpy, exactly as the reference draws
it), chunked launches of houv_solve_iterate, and the best-of-K + angle-window retry logic shared by
``solve_model`` (registration/models/houv.py:142-206) and ``solve`` (registration/train_utils.py:467-572)."""
import os

import numpy as np
import weakref

import torch

from . import ops

RETRY_THRESHOLD = 0.030      # houv.py:156, train_utils.py:494 (strict >)
ITERS_PER_LAUNCH = 50        # bound single-launch duration; state round-trips through HBM (192 B/hypothesis)

# bench.py sets this to a list to collect (start_event, end_event, hypotheses, iterations, N, M, use_views, pruned, first) per
# houv_solve_iterate[_pruned] launch: HIP events recorded on the stream the kernel is launched on.
LAUNCH_LOG = None

# the 26 non-zero {-1,0,1}^3 axes in the reference's loop order (houv.py:44-51)
LATTICE_AXES = np.array([(x, y, z) for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1)
                         if (x, y, z) != (0, 0, 0)], dtype=np.float64)


def houv_init_params(n_inst, seed=2021):
    """HOUV.reset_weight (houv.py:40-61): re-seed numpy before every draw; rows 0..25 of V are the lattice
    axes (n_inst < 26 raises, as the reference's unchecked assignment does).  Returns float64 [n,8] whose
    entries are the reference's float32 values."""
    np.random.seed(seed)
    V = np.random.randn(n_inst, 3)
    if n_inst < 26:
        raise IndexError(f"index {n_inst} is out of bounds for axis 0 with size {n_inst}")   # houv.py:50
    V[:26] = LATTICE_AXES
    np.random.seed(seed)
    a = np.random.randn(n_inst, 1)
    np.random.seed(seed)
    c = np.random.randn(n_inst, 3)
    np.random.seed(seed)
    s = np.random.randn(n_inst, 1)
    return np.concatenate([V, a, c, s], axis=1).astype(np.float32).astype(np.float64)


def solve_twin_init_params(n_inst):
    """getPredict_angle (train_utils.py:381-386): float64 draws from the GLOBAL numpy RNG in the order
    V, angle, tran_c, tran_s, angle_XYZ (the last is never used but advances the generator)."""
    V = np.random.randn(n_inst, 3)
    a = np.random.randn(n_inst, 1)
    c = np.random.randn(n_inst, 3)
    s = np.random.randn(n_inst, 1)
    np.random.randn(n_inst, 3)
    return np.concatenate([V, a, c, s], axis=1)


# The DEFAULT search since round 3 is the exact pruned one (houv_solve_iterate_pruned): on the same clouds it returns the
# brute-force sweep's result BIT FOR BIT (tests/test_gpu_solve.py::test_pruned_search_is_bit_identical_to_brute_force,
# bench.py's `brute_force` leg compares every timed batch) at about half the time.  It wants spatially compact
# 32-point sub-tiles, so run_stage reorders both clouds into k-d leaves of 32 points first (kd_sort) -- point order carries no meaning to
# the loss; only the fp32 summation order changes with it.  ``houv_amd.solver.PRUNED = False`` or HOUV_SOLVER=brute
# selects the brute-force sweep (unsorted clouds, the round-1/2 default); clouds of up to 256 points take it in any case.
PRUNED = os.environ.get("HOUV_SOLVER", "pruned").strip().lower() != "brute"
PRUNED_MAX_POINTS = 4096     # the fused kernel's limit; above 2048 points a visit-mask bit covers a 64-point super-tile
PRUNED_MIN_POINTS = 257      # up to 256 points (8 sub-tiles, one point per lane) the library itself runs the brute-force kernel
                             # (houv_solve_variant reports prune mode 0); from 257 on the search pays (profiles/r03_sizes.txt)


def uses_pruned(N, M, pruned=None):
    """Whether run_stage takes the pruned kernel for clouds of N and M points under the current / given switch."""
    pruned = PRUNED if pruned is None else pruned
    return bool(pruned) and PRUNED_MIN_POINTS <= max(N, M) <= PRUNED_MAX_POINTS


def morton_sort(cloud):
    """Reorder every cloud [P,N,3] along a 30-bit Morton curve (spatially compact 32-point sub-tiles make the pruned
    search effective; the result of a solve does not depend on the order of the points beyond fp32 summation order)."""
    lo = cloud.min(dim=1, keepdim=True)[0]
    hi = cloud.max(dim=1, keepdim=True)[0]
    q = ((cloud - lo) / (hi - lo + 1e-9) * 1023.0).to(torch.int64).clamp_(0, 1023)

    def spread(x):
        x = (x | (x << 16)) & 0x030000FF
        x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3
        x = (x | (x << 2)) & 0x09249249
        return x
    code = spread(q[..., 0]) | (spread(q[..., 1]) << 1) | (spread(q[..., 2]) << 2)
    order = torch.argsort(code, dim=1, stable=True)     # stable: sorting a sorted cloud again is the identity
    return torch.gather(cloud, 1, order.unsqueeze(2).expand(-1, -1, 3)).contiguous()


KD_RULE = "area"             # split axis: "area" = the one whose halves have the smallest projected box areas (default: -3 % kernel
                             # time, the view metrics search in projections) | "extent" = the longest one


def _split_segments(seg, cut, rule):
    """One k-d split of every segment of ``seg`` [S, L, 3] at position ``cut``: each segment reordered (stably) along its split axis."""
    S, L, _ = seg.shape
    if rule == "area":
        # the view terms search in the three axis-dropped projections: pick the split whose two halves have the smallest
        # summed projected box areas (ties: the lowest axis) -- ~5 % fewer sub-tile visits than the longest-axis rule
        best = out = None
        for ax in range(3):
            o = torch.argsort(seg[..., ax], dim=1, stable=True)
            cand = torch.gather(seg, 1, o.unsqueeze(2).expand(-1, -1, 3))

            def area(x):
                e = x.max(dim=1)[0] - x.min(dim=1)[0]
                return e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 0] * e[:, 2]
            cost = area(cand[:, :cut]) + area(cand[:, cut:])
            if best is None:
                best, out = cost, cand
            else:
                take = cost < best
                best = torch.where(take, cost, best)
                out = torch.where(take.view(S, 1, 1), cand, out)
        return out
    ext = seg.max(dim=1)[0] - seg.min(dim=1)[0]                          # [S,3]
    axis = ext.argmax(dim=1)                                             # first maximum wins ties: deterministic
    key = torch.gather(seg, 2, axis.view(S, 1, 1).expand(-1, L, 1))[..., 0]
    order = torch.argsort(key, dim=1, stable=True)
    return torch.gather(seg, 1, order.unsqueeze(2).expand(-1, -1, 3))


def kd_sort(cloud, leaf=32, rule=None):
    """Reorder every cloud [P,N,3] so that consecutive runs of ``leaf`` points -- the kernel's 32-point sub-tiles -- are the leaves
    of a balanced k-d tree: the point range is halved (at a multiple of ``leaf``) along the axis ``KD_RULE`` picks, recursively.  Leaves of a
    k-d tree have tighter boxes than runs of a Morton curve (whose 32-runs straddle the curve's jumps), so the pruned search visits
    fewer sub-tiles.  The order is CANONICAL: it starts from the lexicographic (x, y, z) order and only uses stable sorts, so it is
    a function of the point SET -- sorting a sorted cloud again (or any permutation of it) gives the same order, which is what
    lets bench.py hand identical clouds to both searches.  The segments of one tree level that have the same length and cut are
    split in ONE batched call (all of them when N is leaf times a power of two: 6 calls instead of 63 at 2048 points)."""
    P, N, _ = cloud.shape
    rule = rule or KD_RULE
    for ax in (2, 1, 0):                                    # lexicographic by (x, y, z): canonical starting order
        order = torch.argsort(cloud[..., ax], dim=1, stable=True)
        cloud = torch.gather(cloud, 1, order.unsqueeze(2).expand(-1, -1, 3))
    segs = [(0, N)]
    while True:
        nxt, groups = [], {}
        for (a, b) in segs:
            tiles = -(-(b - a) // leaf)
            if tiles <= 1:
                nxt.append((a, b))
                continue
            mid = a + (tiles - tiles // 2) * leaf                                # left half gets the extra tile; a multiple of leaf
            groups.setdefault((b - a, mid - a), []).append(a)
            nxt += [(a, mid), (mid, b)]
        if not groups:
            break
        for (length, cut), starts in groups.items():
            if len(starts) == 1:
                a = starts[0]
                cloud[:, a:a + length] = _split_segments(cloud[:, a:a + length], cut, rule)
            else:                                                                # same shape: one batched split
                idx = (torch.tensor(starts, device=cloud.device).view(-1, 1) + torch.arange(length, device=cloud.device).view(1, -1)).reshape(-1)
                seg = cloud[:, idx].reshape(P * len(starts), length, 3)
                cloud[:, idx] = _split_segments(seg, cut, rule).reshape(P, len(starts) * length, 3)
        segs = nxt
    return cloud.contiguous()


SPATIAL_SORT = "kd"          # "kd" (balanced k-d leaves, round 3) | "morton" (rounds 1-2)


_SORTED = {}                 # data_ptr -> (version, shape, device, order, leaf, weakref) of tensors spatial_sort itself produced


def spatial_sort(cloud, leaf=32):
    """The point order the pruned search wants (spatially compact 32-point sub-tiles; ``leaf`` = 64 for clouds of more than 2048
    points, whose visit masks are over 64-point super-tiles).  A tensor this function returned is
    recognised (the same tensor object: address, version counter, shape, weak reference) and handed back as it is: callers that keep their clouds sorted (bench.py,
    the drivers' batches) do not pay for the sort again in every stage."""
    if _sorted_leaf(cloud) == leaf:
        return cloud
    out = kd_sort(cloud, leaf) if SPATIAL_SORT == "kd" else morton_sort(cloud)
    _mark_sorted(out, leaf)
    return out


def sort_leaf(N, M):
    """Leaf size of the spatial sort for clouds of N and M points: the pruned search's visit-mask granularity."""
    return 32 if max(N, M) <= 2048 else 64


def _sorted_leaf(cloud):
    """Leaf size a (contiguous) tensor was spatially sorted with, or None when it is not known to be sorted."""
    m = _SORTED.get(cloud.data_ptr()) if cloud.is_contiguous() else None
    # the entry must be about THIS tensor object: a freed tensor's address is soon re-used by another of the same shape, and an
    # unsorted cloud taken for a sorted one is still solved exactly but in another summation order (and slower)
    ok = m is not None and m[5]() is cloud and m[:4] == (cloud._version, tuple(cloud.shape), cloud.device, SPATIAL_SORT + KD_RULE)
    return m[4] if ok else None


def _mark_sorted(cloud, leaf):
    """Record that every cloud of this (contiguous) tensor is in spatial_sort order -- e.g. a row subset of a sorted batch."""
    if len(_SORTED) > 512:
        for key in [k for k, v in _SORTED.items() if v[5]() is None]:
            del _SORTED[key]
        if len(_SORTED) > 512:
            _SORTED.clear()
    _SORTED[cloud.data_ptr()] = (cloud._version, tuple(cloud.shape), cloud.device, SPATIAL_SORT + KD_RULE, leaf, weakref.ref(cloud))


FUSED_MAX_POINTS = 4096     # both clouds of a hypothesis live in LDS inside the fused kernel (houv_solve_iterate)


def _run_stage_unfused(src, tgt, params, K, n_iters, *, angle_base, trans_mode, use_views, f64_params, lr, want_grad,
                       want_cd, alpha):
    """Clouds too large for the fused kernel's LDS: the loop in the reference's own shape -- differentiable forward
    (houv.py:94-103 / train_utils.py:113-148), Chamfer through the stand-alone HIP op (houv_chamfer_forward/backward,
    any size), torch.topk, autograd and torch.optim.Adam -- everything on the GPU, ~an order of magnitude slower per
    iteration than the fused kernel.  Same return convention as run_stage."""
    import math
    from .model_utils_completion import calc_cd_percent, loss_view
    from .train_utils import rotation, translation
    dev = src.device
    P, N, _ = src.shape
    n = P * K
    dt = torch.float64 if f64_params else torch.float32
    p0 = torch.as_tensor(params, dtype=torch.float64).to(dev)
    leaves = [p0[:, 0:3].to(dt).clone().requires_grad_(True), p0[:, 3:4].to(dt).clone().requires_grad_(True),
              p0[:, 4:7].to(dt).clone().requires_grad_(True), p0[:, 7:8].to(dt).clone().requires_grad_(True)]
    opt = torch.optim.Adam(leaves, lr=lr)
    srck = src.repeat_interleave(K, dim=0)      # houv.py:111-112 replicates the clouds K-fold as well
    tgtk = tgt.repeat_interleave(K, dim=0)
    out = None
    for it in range(n_iters):
        V, a, c, sc = (l.float() for l in leaves)      # the pose is evaluated in fp32, as in the fused kernel
        angle = torch.sin(a * math.pi) * math.pi / 8 + math.pi / 8 + angle_base * math.pi / 4
        R = rotation(angle, V)
        sigma = torch.sin(sc * math.pi) * 0.125 + 0.125 if trans_mode == 0 else torch.sin(sc * math.pi)
        T = translation(c, sigma)
        moved = torch.bmm(srck, R.transpose(1, 2)) + T
        cds = [calc_cd_percent(moved, tgtk, percent=alpha)]
        min_1, _ = torch.min(torch.stack(cds[0], dim=1), dim=1)
        loss = min_1 * 6
        if use_views:
            for d in range(3):
                cds.append(loss_view(moved, tgtk, dim=d))
                v, _ = torch.min(torch.stack(cds[-1], dim=1), dim=1)
                loss = loss + v
        opt.zero_grad()
        loss.mean().backward()
        if it == n_iters - 1:
            out = dict(score=min_1.detach().float(), loss=loss.detach().float(), R=R.detach().float().contiguous(),
                       T=T.detach().float()[:, 0].contiguous(),
                       last_params=torch.cat([l.detach() for l in leaves], dim=1).double())
            if want_grad:
                out["grad"] = torch.cat([l.grad for l in leaves], dim=1).float()
            if want_cd:
                cd = torch.zeros((n, 8), dtype=torch.float32, device=dev)
                for m, (c0, c1) in enumerate(cds):
                    cd[:, 2 * m], cd[:, 2 * m + 1] = c0.detach(), c1.detach()
                out["cd"] = cd
        opt.step()
    state = torch.zeros((n, 24), dtype=torch.float64, device=dev)
    state[:, :8] = torch.cat([l.detach() for l in leaves], dim=1).double()
    state[:, 8:16] = torch.cat([opt.state[l]["exp_avg"] for l in leaves], dim=1).double()
    state[:, 16:24] = torch.cat([opt.state[l]["exp_avg_sq"] for l in leaves], dim=1).double()
    return out, state


def run_stage(src, tgt, params, K, n_iters, *, angle_base, trans_mode, use_views, f64_params, lr,
              iters_per_launch=None, want_grad=False, want_cd=False, alpha=0.5, pruned=None, want_last_params=False):
    """Run ``n_iters`` optimisation iterations for P*K hypotheses.  params: float64 [P*K,8] (numpy or tensor).
    Returns (out dict of the last forward, state tensor [P*K,24] fp64 after n_iters Adam steps).
    ``want_last_params``: also return, as out["last_params"] (fp64 [P*K,8]), the parameters the LAST forward read, i.e.
    before the final Adam step (the last launch is then exactly one iteration; chunking is bit-neutral)."""
    if n_iters < 1:
        raise ValueError("num_epochs must be >= 1 (the reference reads the last iteration's outputs)")
    src = src.contiguous().float()
    tgt = tgt.contiguous().float()
    P, N, _ = src.shape
    if use_views and tgt.shape[1] != N:
        # loss_view multiplies the target by a mask shaped like the moved cloud (model_utils_completion.py:158-163):
        # the reference raises on N != M whenever the view terms are on.
        raise RuntimeError(f"The size of tensor a ({tgt.shape[1]}) must match the size of tensor b ({N}) at "
                           "non-singleton dimension 1")
    dev = src.device
    n = P * K
    p = torch.as_tensor(params, dtype=torch.float64)
    if tuple(p.shape) != (n, 8):
        raise ValueError(f"params must be [{n},8]")
    if max(N, tgt.shape[1]) > FUSED_MAX_POINTS:
        from . import _lib
        _lib.require_gpu(src, tgt)
        return _run_stage_unfused(src, tgt, p, K, n_iters, angle_base=angle_base, trans_mode=trans_mode,
                                  use_views=use_views, f64_params=f64_params, lr=lr, want_grad=want_grad,
                                  want_cd=want_cd, alpha=alpha)
    state = torch.zeros((n, 24), dtype=torch.float64, device=dev)
    state[:, :8] = p.to(dev)
    k_full = int(N * alpha)            # model_utils_completion.py:85-86 with percent = alpha
    k_view = int(N * 1)
    step = iters_per_launch or ITERS_PER_LAUNCH
    pruned = PRUNED if pruned is None else pruned
    nn_ws = None
    if uses_pruned(N, tgt.shape[1], pruned):
        leaf = sort_leaf(N, tgt.shape[1])
        src, tgt = spatial_sort(src, leaf), spatial_sort(tgt, leaf)
        nn_ws = ops.solve_workspace(n, N, tgt.shape[1], dev)
    done, out = 0, None
    last_params = None
    while done < n_iters:
        it = min(step, n_iters - done)
        if want_last_params:
            if done == n_iters - 1:
                last_params = state[:, :8].clone()
            else:
                it = min(it, n_iters - 1 - done)
        last = done + it == n_iters
        if LAUNCH_LOG is not None:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev1 = torch.cuda.Event(enable_timing=True)
            ev0.record(torch.cuda.current_stream(dev))
        out = ops.solve_iterate(src, tgt, state, K, steps_done=done, n_iters=it, angle_base=angle_base,
                                trans_mode=trans_mode, use_views=use_views, f64_params=f64_params, k_full=k_full,
                                k_view=k_view, lr=lr, loss_scale=1.0 / n, want_grad=want_grad and last,
                                want_cd=want_cd and last, nn_ws=nn_ws,
                                ws_valid="verify" if pruned == "verify" else done > 0)
        if LAUNCH_LOG is not None:
            ev1.record(torch.cuda.current_stream(dev))
            LAUNCH_LOG.append((ev0, ev1, n, it, N, tgt.shape[1], bool(use_views), nn_ws is not None, done == 0))
        done += it
    if want_last_params:
        out["last_params"] = last_params
    return out, state


# The three retry stages (angle windows 45-90 / 90-135 / 135-180 degrees) are independent solves of the same retried
# pairs: launched on three side streams they share the GPU instead of each leaving most of it idle (a retry set of 10
# pairs x 64 restarts is 640 workgroups on a chip that holds 512 at a time: two rounds for 1.25 rounds of work, three
# times over).  The replacements are still applied in the reference's order 1, 2, 3.
CONCURRENT_RETRIES = True
_retry_streams = {}


def _side_streams(dev):
    key = (dev.type, dev.index)
    if key not in _retry_streams:
        _retry_streams[key] = [torch.cuda.Stream(device=dev) for _ in range(3)]
    return _retry_streams[key]


def best_of_k_with_retry(stage_fn, src, tgt):
    """houv.py:152-197 / train_utils.py:488-545.  ``stage_fn(src, tgt, base) -> (score[B,K], R[B,K,3,3], T[B,K,3])``.
    Base-0 stage; pairs whose best score is > 0.030 are re-solved in the 45-90/90-135/135-180 degree windows
    and replaced where strictly better.  Returns ans[B,4,4] (row 3 left all-zero, as the reference leaves it),
    score, and the retried pair indices.  One host sync per stage (the retry set is data dependent)."""
    B = src.shape[0]
    score, R, T = stage_fn(src, tgt, 0)
    best, _ = score.topk(1, dim=1, largest=False, sorted=True)          # NaN hypotheses sort last
    retry = torch.nonzero(best[:, 0] > RETRY_THRESHOLD).reshape(-1)
    if retry.numel() > 0:
        s_add, t_add = src[retry].contiguous(), tgt[retry].contiguous()
        for sub, whole in ((s_add, src), (t_add, tgt)):        # pairs picked out of a sorted batch are sorted: the three retry
            leaf = _sorted_leaf(whole)                          # stages need not sort them again
            if leaf is not None:
                _mark_sorted(sub, leaf)
        outs = {}
        if CONCURRENT_RETRIES and src.is_cuda:
            main = torch.cuda.current_stream(src.device)
            side = _side_streams(src.device)
            for base in range(1, 4):
                side[base - 1].wait_stream(main)                        # s_add / t_add were produced on the main stream
                with torch.cuda.stream(side[base - 1]):
                    outs[base] = stage_fn(s_add, t_add, base)
            for base in range(1, 4):
                main.wait_stream(side[base - 1])
                for x in outs[base]:
                    x.record_stream(main)
            s_add.record_stream(side[0]); s_add.record_stream(side[1]); s_add.record_stream(side[2])
            t_add.record_stream(side[0]); t_add.record_stream(side[1]); t_add.record_stream(side[2])
        else:
            for base in range(1, 4):
                outs[base] = stage_fn(s_add, t_add, base)
        for base in range(1, 4):
            score_a, R_a, T_a = outs[base]
            best_a, _ = score_a.topk(1, dim=1, largest=False, sorted=True)
            flag = torch.nonzero((best_a < best[retry]).reshape(-1)).reshape(-1)
            ge = retry[flag]
            R[ge] = R_a[flag]
            score[ge] = score_a[flag]
            T[ge] = T_a[flag]
            best[ge] = best_a[flag]
    _, k = score.topk(1, dim=1, largest=False, sorted=True)
    pick = k[:, 0]
    rows = torch.arange(B, device=score.device)
    ans = torch.zeros((B, 4, 4), dtype=torch.float32, device=score.device)
    ans[:, :3, :3] = R[rows, pick]
    ans[:, :3, 3] = T[rows, pick]
    return ans, score, retry
