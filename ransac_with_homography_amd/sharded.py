"""Multi-GPU form of the path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

  * RANSAC shards the hypothesis index range [0, K) into contiguous slices, one per rank, over
    replicated correspondences (3 KB).  Each rank reduces its slice to the two packed keys of
    rwh_score_count; ONE all-reduce(MAX) of 2 x int64 yields the global winner with the reference's
    first-index tie-break and early-exit semantics (ransac.py:186-202).  16 bytes: pure latency.
  * The warp shards by output-row tiles (one image) or by image (a batch); rows and images are
    independent, so there is NO collective on that path and outputs stay sharded.
  * Batched RANSAC (many image pairs) shards by problem: each rank runs `run_batch` on its contiguous slice of the
    problem list, no collective; results stay sharded unless the caller asks for an all-gather of the (small) results.

The scoring backend is injectable so that the sharding / reduction logic is testable on CPU
ranks (the CPU tests plug the oracle in; the product default is the HIP kernels).
"""
import numpy as np


def shard_range(total, rank, world):
    """Contiguous slice [begin, end) of `total` items owned by `rank`; sizes differ by at most 1."""
    base, rem = divmod(int(total), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def _dist():
    import torch.distributed as dist
    return dist


def gpu_score_slice(pa, pb, idx_slice, th, loss, need, hyp_base):
    """Product backend: this rank's slice of the index table through the SAME native driver a one-GPU `RANSAC.run` uses
    (`rwh_ransac_run` with `hyp_base` = the slice's offset: upload, K1 + K2, the settle step with the reference's solver on
    every hypothesis that can decide the slice -- repeated-index samples on host threads while the GPU searches --, the accept
    rules), which returns the slice's two packed keys (round 4; rounds 2-3 ran the step-by-step Python driver here).
    A hypothesis that can decide the GLOBAL search can decide its own slice (the slice's best is not above the global best),
    so settling needs no exchange and the path keeps its ONE collective.  Without numpy's LAPACK by address: the Python twin.
    Returns the keys (include/rwh.h, rwh_score_count) as a 2 x int64 tensor on the GPU."""
    import torch
    from . import _lapack, kernels
    from .ransac import HOST_THREADS, RESCORE_MARGIN, _settle_on_host, presettle, repeated_rows
    k = idx_slice.shape[0]
    w0 = w1 = 0
    if k:
        idx_host = np.ascontiguousarray(np.asarray(idx_slice)[:, :4], dtype=np.int32)
        pa_host, pb_host = pa.cpu().numpy(), pb.cpu().numpy()          # (before the launch: a copy after it would wait for the search)
        addr = _lapack.dgesdd_address()
        gesv = None if loss == "fwd" else _lapack.dgesv_address()
        if addr is not None and (loss == "fwd" or gesv is not None):
            ws = kernels.RunWorkspace(pa.shape[0], k, pa.device)
            res = kernels.ransac_run(np.ascontiguousarray(pa_host, dtype=np.float32), np.ascontiguousarray(pb_host, dtype=np.float32), idx_host,
                                     th, loss, need, RESCORE_MARGIN, ws, addr, HOST_THREADS, dgesv=gesv, hyp_base=hyp_base, want_keys=True)
            return torch.from_numpy(res[7].copy()).to(pa.device)
        ws = kernels.SearchWorkspace(k, pa.shape[0], pa.device, want_masks=False)
        kernels.ransac_search(pa, pb, torch.from_numpy(idx_host).to(pa.device), th, loss, need, ws, hyp_base=hyp_base)
        pre = presettle(pa, pb, pa_host, pb_host, idx_host, np.flatnonzero(repeated_rows(idx_host)), th, loss)   # while the GPU searches
        counts_host, flags_host = ws.counts_flags()
        winner, early, count, _, _, _ = _settle_on_host(pa, pb, pa_host, pb_host, idx_host, counts_host, flags_host, need, th, loss,
                                                        RESCORE_MARGIN, pre=pre, H_dev=ws.H, flags_dev=ws.flags)
        if winner is not None:
            inv = 0xFFFFFFFF - (hyp_base + winner)
            w0 = (int(count) << 32) | inv
            w1 = inv if early else 0
    return torch.tensor([w0, w1], dtype=torch.int64, device=pa.device)


def ransac_sharded(pa, pb, idx_table, th, loss, need, score_slice=gpu_score_slice, group=None):
    """Sharded hypothesis search.  `idx_table`: the full K x 4 host index table (every rank draws the
    same table from the same seed).  Returns (winner_index | None, early_exit) on every rank."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    k = idx_table.shape[0]
    b, e = shard_range(k, rank, world)
    best = score_slice(pa, pb, idx_table[b:e], th, loss, need, b)
    if not isinstance(best, torch.Tensor):
        best = torch.as_tensor(np.asarray(best, dtype=np.int64))
    if world > 1:
        dist.all_reduce(best, op=dist.ReduceOp.MAX, group=group)  # the one collective of the path
    from .kernels import decode_best
    winner, _, early = decode_best(best.cpu().numpy(), k)
    return winner, early


def winner_inliers(pa, pb, idx_row, th, loss):
    """Inlier indices of one hypothesis, recomputed locally by every rank (deterministic): H by the reference's own
    solver on the host (`ransac.svd_hypotheses`, one 8 x 9 SVD), scored by K2."""
    import torch
    from . import kernels
    from .ransac import svd_hypotheses
    H = svd_hypotheses(pa.cpu().numpy(), pb.cpu().numpy(), np.asarray(idx_row).reshape(1, 4))
    hinv = None if loss == "fwd" else torch.from_numpy(kernels.host_inverses(H)).to(pa.device)      # numpy's own inverse (ransac.py:74)
    counts, masks, _ = kernels.score_count(torch.from_numpy(H).to(pa.device), pa, pb, th, loss, 1 << 30,
                                           kernels.new_best(pa.device), hinv=hinv)
    bits = np.unpackbits(masks[0].cpu().numpy().view(np.uint8), bitorder="little")[:pa.shape[0]]
    return np.nonzero(bits)[0].astype(np.int64), int(counts[0])


def warp_row_shard(src, inv_h, grid, bound_hw, interp, out_dtype, group=None):
    """This rank's output-row tile of one warp: returns (tensor [rows, out_w, C], (row_begin, row_end)).
    The source is replicated; nothing is exchanged."""
    from . import kernels
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    rows = shard_range(grid.out_h, rank, world)
    return kernels.warp_backward(src, inv_h, grid, bound_hw, interp, out_dtype, rows=rows), rows


def run_batch_sharded(datas, gather=False, solve=None, group=None, **kw):
    """Many-pair RANSAC over the ranks: rank r solves problems shard_range(len(datas), r, world) with
    `ransac.run_batch(slice, **kw)` (or the injected `solve`), in one GPU submission per rank.  Device sampling stays
    a function of (seed, GLOBAL problem index, hypothesis): pass `seed`, and every problem gets the table it would get
    in a single-rank run.  Returns (results of this rank's slice, (begin, end)); with gather=True every rank receives
    the full list instead (one all_gather of small Python objects: a 3x3 matrix and an index array per problem)."""
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    b, e = shard_range(len(datas), rank, world)
    if solve is None:
        from .ransac import run_batch as solve
    mine = solve(datas[b:e], problem_base=b, **kw) if e > b else []
    if not gather or world == 1:
        return mine, (b, e)
    parts = [None] * world
    dist.all_gather_object(parts, mine, group=group)
    return [r for part in parts for r in part], (0, len(datas))
