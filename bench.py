#!/usr/bin/env python3
"""bench.py -- backward-warp Mpixels/s (+ RANSAC hypotheses/s) on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W

Run as that plain command with N > 1 it starts its own ranks (one process per GPU, `torch.distributed.run` on
127.0.0.1, before anything touches the GPU in the parent); under `torch.distributed.run` (the driver's launch) it is
a rank.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
  transformImageH-style warp of 3840x2160 RGB u8 frames with the survey's mild-perspective Hs,
  bilinear, uint8 out (output 2028x3771 per frame, the reference's auto-bounds geometry).
  One "step" = ONE launch of the warp kernel over a batch of FRAMES distinct frames already
  resident in HBM (batch source + destination = 1.5 GB > the 256 MB Infinity Cache, so the
  traffic is HBM traffic, not cache traffic).
  N > 1: every rank warps its own batch (shard by image, no collective): weak scaling (`value`).
Extra keys of the same JSON line:
  warp_8k       the north_star's 8K warp: 8 distinct 7680x4320 frames per launch, same kernel, with its own kernel
                time / achieved GB/s / fraction of the HBM roofline (N = 1);
  config5_batch_1080p  BASELINE config 5's warp: 512 distinct 1080p frames sharded by image, one launch per rank.
  strong_8k     the same 8 frames with the OUTPUT ROWS sharded over the ranks (sharded.warp_row_shard's row tiles, no
                collective): fixed total work = the strong-scaling leg;
  ransac        hypotheses/s on matchespoints (10 000 and 100 000 hypotheses at N = 1; 100 000 sharded over the ranks with
                the ONE all-reduce(MAX) of 2 x int64 at N > 1) and what every rank decoded from the reduced keys;
  rccl          backend, the world size every rank observed, whether all ranks agree on the winner;
  roofline / cpu_baseline   as the contract asks (the CPU leg runs on rank 0 at N = 1 only).
"""
import argparse
import contextlib
import io
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H_S = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
PREWARM_MS = 150.0     # untimed clock ramp before the warm-up steps (see run_rank)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=32, help="4K frames per step (per GPU)")
    ap.add_argument("--src", default="3840x2160", help="source frame WxH (default: the BASELINE 4K configuration)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="headline + roofline only (no 8K / RANSAC / config 4 legs)")
    return ap.parse_args(argv)


def cpu_baseline(frames_note, src_w, src_h):
    """The numpy CPU path (oracle = bit-identical restatement of the reference) on ONE frame of the
    same workload, min of <= 16 after 1 warm-up, single process (numpy's elementwise ops are single
    threaded).  Plus the RANSAC loop body on 2000 hypotheses."""
    from oracle import rwh_oracle as orc
    img = np.random.default_rng(1234).integers(0, 256, (src_h, src_w, 3), dtype=np.uint8)
    ts = []
    out = None
    for i in range(17):     # 1 warm-up + up to 16 timed frames, bounded at ~25 s: 12-25 s of CPU work depending on the host
        t0 = time.perf_counter()
        out, _, _ = orc.transform_image_h(img, H_S)
        ts.append(time.perf_counter() - t0)
        if i >= 3 and sum(ts) > 25.0:
            break
    t = min(ts[1:])
    mpix = out.shape[0] * out.shape[1] / 1e6
    z = np.load(os.path.join(ROOT, "tests", "golden", "matchespoints.npz"))
    X, Y = z["ptsA"].T, z["ptsB"].T
    np.random.seed(0)
    idx = np.random.randint(0, X.shape[1], (2000, 4))
    orc.ransac_table(X, Y, idx[:50], th=5, method="fwd")
    t0 = time.perf_counter()
    orc.ransac_table(X, Y, idx, th=5, method="fwd")
    tr = time.perf_counter() - t0
    return {"value": round(mpix / t, 3), "unit": "Mpix/s", "cores": 1, "kind": "port",
            "sample": "1 frame %dx%d RGB u8 -> %dx%d (of the %s), numpy oracle, min of %d after warm-up; "
                      "os.cpu_count=%d OPENBLAS_NUM_THREADS=%s" % (src_w, src_h, out.shape[0], out.shape[1], frames_note, len(ts) - 1,
                                                                   os.cpu_count(), os.environ.get("OPENBLAS_NUM_THREADS", "unset")),
            "ransac_hyp_per_s": round(2000 / tr, 1), "ransac_sample": "2000 hypotheses x 185 correspondences, fwd"}


# ------------------------------------------------------------------------------------------------------------------
# The device side of a rank.  Everything bench.py asks of the GPU goes through this object, so that the rank function
# (timing protocol, sharding, collectives, the JSON line) can be driven on CPU ranks with a stand-in (tests/).
# ------------------------------------------------------------------------------------------------------------------
class GpuBackend:
    collective = "nccl"

    def __init__(self, local_rank, rehearsal=False):
        import torch
        self.torch = torch
        # Rehearsal on a one-GPU box (never used by the driver): RWH_BENCH_REHEARSAL=1 maps every rank to cuda:0 and
        # swaps RCCL for gloo (RCCL refuses two ranks on one device); the all-reduces then go through host tensors.
        self.rehearsal = rehearsal
        if rehearsal:
            local_rank = 0
            self.collective = "gloo"
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        from ransac_with_homography_amd import kernels, sharded
        from ransac_with_homography_amd.homography import _bounds
        self.kernels, self.sharded, self._bounds = kernels, sharded, _bounds

    def init_args(self):
        return {} if self.rehearsal else {"device_id": self.dev}

    def sync(self):
        self.torch.cuda.synchronize()

    def tensor(self, values, dtype="int64"):
        return self.torch.tensor(values, dtype=getattr(self.torch, dtype), device="cpu" if self.rehearsal else self.dev)

    # ---- warp workloads ----
    def make_warp(self, frames, src_w, src_h, seed, rows_of=None):
        """-> (step(), out_h, out_w, rows, plan): `frames` distinct src_w x src_h frames resident in HBM; step() = one launch.
        rows_of = (rank, world): this rank's output-row tile only (sharded.shard_range, what warp_row_shard launches)."""
        torch, k = self.torch, self.kernels
        gen = torch.Generator(device=self.dev).manual_seed(seed)
        src = torch.randint(0, 256, (frames, src_h, src_w, 3), dtype=torch.uint8, device=self.dev, generator=gen)
        min_x, min_y, out_w, out_h = self._bounds(src_h, src_w, H_S, 0)
        grid = k.Grid(min_x, min_x + out_w - 1, out_w, min_y, min_y + out_h - 1, out_h)
        inv = np.linalg.inv(H_S)
        rows = (0, out_h) if rows_of is None else self.sharded.shard_range(out_h, *rows_of)
        dst = torch.empty((frames, rows[1] - rows[0], out_w, 3), dtype=torch.uint8, device=self.dev)
        src[:, 0, 0, :] = 0  # texel (0,0) is blanked once (homography.py:126-130); the timed launches are the warp kernel alone
        plan = k.warp_plan((frames, src_h, src_w, 3), torch.uint8, inv, grid, (src_h, src_w), "bilinear", torch.uint8, rows=rows)

        def step():
            k.warp_backward(src, inv, grid, (src_h, src_w), "bilinear", torch.uint8, zero_origin=False, rows=rows, out=dst)
        return step, out_h, out_w, rows, plan, (src, dst, grid, inv)

    def events(self):
        torch = self.torch
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def clock_probe(self, ms):
        """One wavefront on a side stream that counts shader cycles for `ms` ms (rwh_lab_clock_probe): .mhz() afterwards."""
        return self.kernels.ClockProbe(ms)

    def load_facts(self, step, seconds=1.2):
        """What the chip does under `seconds` of back-to-back step() launches: the shader clock it holds (in-kernel cycle
        counter against the 100 MHz constant clock) and the board power meanwhile, against the power cap -- read from the
        amdgpu hwmon node of THIS device (power1_average / power1_cap, microwatts).  No child process: a program that holds the
        GPU must not fork + exec a tool (rounds 2-3 polled rocm-smi here; under rocprofv3 the box refused every one of those
        execs and the profiled passes silently carried no power figure).
        MI355X lowers its clock when a kernel reaches the cap; then the kernel's bound is energy per pixel."""
        import threading
        samples, stop = [], threading.Event()
        node, cap_node = self.kernels.power_node(), self.kernels.power_cap_node()

        def read_uw(path):
            try:
                with open(path) as f:
                    return int(f.read().strip()) * 1e-6
            except (OSError, ValueError, TypeError):
                return None

        def sampler():
            while not stop.is_set():
                w = read_uw(node) if node else None
                if w is not None:
                    samples.append(w)
                time.sleep(0.02)
        th = threading.Thread(target=sampler)
        th.start()
        t_end = time.perf_counter() + seconds
        probe = self.clock_probe(min(1000.0 * seconds * 0.6, 1500.0))
        cur = self.torch.cuda.current_stream()
        while time.perf_counter() < t_end:
            for _ in range(50):
                step()
            cur.synchronize()              # (the launch stream only: a device-wide synchronize would wait for the probe)
        stop.set()
        th.join()
        tail = sorted(samples[len(samples) // 2:])
        return {"sclk_mhz": round(probe.mhz(), 0), "power_w": round(tail[len(tail) // 2], 1) if tail else None,
                "power_cap_w": read_uw(cap_node) if cap_node else None, "power_samples": len(tail),
                "power_source": node or "no hwmon node for this device"}

    # ---- RANSAC workloads ----
    def _matches(self):
        z = np.load(os.path.join(ROOT, "tests", "golden", "matchespoints.npz"))
        return self.torch.from_numpy(z["ptsA"]).to(self.dev), self.torch.from_numpy(z["ptsB"]).to(self.dev)

    def make_search(self, K, b, e, settle=False):
        """Hypotheses [b, e) of the K-row numpy table (seed 0) on matchespoints: key reset + K1 + K2 + argmax in one call.
        settle=True (the N > 1 leg): sharded.gpu_score_slice -- the same search PLUS this rank's host settle step (the
        reference's solver for flagged / near-best hypotheses), i.e. the path that carries the bit-exact guarantee."""
        torch, k = self.torch, self.kernels
        pa, pb = self._matches()
        need = k.need_count(185, 70, 4)
        np.random.seed(0)
        table = np.random.randint(0, 185, (K, 4))
        if settle:
            idx_host = np.ascontiguousarray(table[b:e], dtype=np.int32)

            def search_settled():
                best = self.sharded.gpu_score_slice(pa, pb, idx_host, 5.0, "fwd", need, b)
                return best.cpu() if self.rehearsal else best
            return search_settled
        idx_dev = torch.from_numpy(table[b:e].astype(np.int32)).to(self.dev)
        ws = k.SearchWorkspace(e - b, 185, self.dev, want_masks=False)

        def search():
            k.ransac_search(pa, pb, idx_dev, 5.0, "fwd", need, ws, hyp_base=b)
            return ws.best.cpu() if self.rehearsal else ws.best
        return search

    def batched_search_leg(self, sync_all):
        """rwh_ransac_batched: 64 independent copies of the problem x 10 000 hypotheses each in ONE submission, device
        Philox sampling -- the throughput figure once the per-run launch + readback latency is amortised."""
        torch, k = self.torch, self.kernels
        pa, pb = self._matches()
        need = k.need_count(185, 70, 4)
        P, K = 64, 10000
        offs = torch.arange(0, 185 * (P + 1), 185, dtype=torch.int32, device=self.dev)
        pa_b, pb_b = pa.repeat(P, 1), pb.repeat(P, 1)
        needs = torch.full((P,), need, dtype=torch.int32, device=self.dev)
        bws = k.BatchWorkspace(P, K, 185, self.dev, want_masks=False)

        def batched_step():
            k.ransac_batched(pa_b, pb_b, offs, needs, 5.0, "fwd", bws, seed=2024)
            return bws.best.cpu()

        for _ in range(2):
            batched_step()
        sync_all()
        t0 = time.perf_counter()
        R = 10
        for _ in range(R):
            bb = batched_step()
        sync_all()
        tr = time.perf_counter() - t0
        cnts = [k.decode_best(bb[p].numpy(), K)[1] for p in range(P)]
        return {"batched P=%d K=%d" % (P, K): {
            "hyp_per_s": round(P * K * R / tr, 1), "us_per_run": round(tr / R * 1e6, 1),
            "pair_evals_per_s": round(P * K * R * 185 / tr, 1), "winner_count_min_max": [int(min(cnts)), int(max(cnts))],
            "sampling": "device Philox4x32-10, 4 distinct correspondences (non-parity mode)"}}


def synthetic_correspondences(M, seed=42):
    """SURVEY 8(d) scaling set: Hs-projected uniform points in [0, 4000)^2 + N(0, 1 px) noise + 40 % uniform outliers."""
    rng = np.random.default_rng(seed)
    Hs = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
    A = rng.uniform(0, 4000, (M, 2))
    P = np.c_[A, np.ones(M)] @ Hs.T
    B = P[:, :2] / P[:, 2:] + rng.normal(0, 1.0, (M, 2))
    out = rng.random(M) < 0.4
    B[out] = rng.uniform(0, 4000, (int(out.sum()), 2))
    return A.astype(np.float32), B.astype(np.float32), rng


def _scaling_set_leg(self, sync_all):
    """K = 10 000 hypotheses on the synthetic sets of SURVEY 8(d): M = 1 024 and 16 384 correspondences (points streamed
    per hypothesis instead of held in registers)."""
    torch, k = self.torch, self.kernels
    report = {}
    K = 10000
    for M in (1024, 16384):
        A, B, rng = synthetic_correspondences(M)
        pa, pb = torch.from_numpy(A).to(self.dev), torch.from_numpy(B).to(self.dev)
        idx = torch.from_numpy(rng.integers(0, M, (K, 4)).astype(np.int32)).to(self.dev)
        need = k.need_count(M, 70, 4)
        ws = k.SearchWorkspace(K, M, self.dev, want_masks=False)

        def step():
            k.ransac_search(pa, pb, idx, 3.0, "fwd", need, ws)
            return ws.best.cpu()

        for _ in range(2):
            step()
        sync_all()
        t0 = time.perf_counter()
        R = 10
        for _ in range(R):
            best = step()
        sync_all()
        tr = time.perf_counter() - t0
        winner, cnt, early = k.decode_best(best.numpy(), K)
        report["M=%d K=%d" % (M, K)] = {"hyp_per_s": round(K * R / tr, 1), "us_per_run": round(tr / R * 1e6, 1),
                                         "pair_evals_per_s": round(K * R * M / tr, 1), "winner": winner, "winner_count": cnt,
                                         "inlier_fraction_planted": 0.6}
    return {"scaling_set": report}


def _parity_path_leg(self):
    """`RANSAC.run` end to end -- the path that carries the bit-exact guarantee (winner, count, inlier list equal the
    reference's): numpy sampling, uploads, K1 + K2, the reference's own SVD on the host for every sample K1 flags (repeated
    index: 3.3 %; ill-conditioned: ~2 %) and for every hypothesis near the decision, K2 on those, the accept rules, the host
    refit.  The host SVDs are LAPACK dgesdd calls of ~8 us each that OpenBLAS serialises: they are the cost."""
    import ransac as rs
    z = np.load(os.path.join(ROOT, "tests", "golden", "matchespoints.npz"))
    X, Y = z["ptsA"].T.copy(), z["ptsB"].T.copy()
    out = {}
    for K, reps in ((1000, 10), (10000, 5), (100000, 2)):
        def run():
            np.random.seed(0)
            with contextlib.redirect_stdout(io.StringIO()):
                r = rs.RANSAC(rs.HomoModel(th=5, d=70, n=4), k=K)
                return r, r.run([X, Y], method="fwd")
        run()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r, (H, inl, cnt) = run()
            ts.append(time.perf_counter() - t0)
        t = sorted(ts)[len(ts) // 2]                      # median: one scheduling hiccup of a host thread is not the path's cost
        out["K=%d" % K] = {"ms_per_run": round(t * 1e3, 3), "ms_per_run_mean": round(sum(ts) / len(ts) * 1e3, 3), "hyp_per_s": round(K / t, 1),
                           "winner": r.last_run["winner"], "inliers": int(cnt),
                           "host_solved_hypotheses": r.last_run.get("host_settled"), "host_rounds": r.last_run.get("host_rounds"),
                           "flagged_by_k1": r.last_run.get("flagged")}
    # dense / clustered correspondences (the round-3 verdict's recipe: a Gaussian cloud around (500, 500), sigma 3 / 10 / 40 px, and two
    # clusters; 30 % outliers): most samples are ill-conditioned there -- what the settle step hands to the host is the cost
    rng = np.random.default_rng(7)
    for name, kind, M, sigma in (("cloud_sigma3", "cloud", 1400, 3.0), ("cloud_sigma10", "cloud", 2000, 10.0), ("cloud_sigma40", "cloud", 2900, 40.0),
                                 ("two_clusters_sigma5", "two", 2000, 5.0)):
        G = rng.normal(500, sigma, (M, 2)) if kind == "cloud" else np.array([[400., 450.], [620., 560.]])[rng.integers(0, 2, M)] + rng.normal(0, sigma, (M, 2))
        P = np.c_[G, np.ones(M)] @ H_S.T
        Bp = P[:, :2] / P[:, 2:3] + rng.normal(0, 1.0, (M, 2))
        o = rng.random(M) < 0.3
        Bp[o] = rng.uniform(Bp.min(), Bp.max(), (int(o.sum()), 2))
        Xd, Yd = G.astype(np.float32).T.copy(), Bp.astype(np.float32).T.copy()
        K = 10000

        def run_d():
            np.random.seed(0)
            with contextlib.redirect_stdout(io.StringIO()):
                r = rs.RANSAC(rs.HomoModel(th=5, d=95, n=4), k=K)
                return r, r.run([Xd, Yd], method="fwd")
        for _ in range(4):          # (a new problem size: page-locked workspace, allocator blocks and the host pool's threads settle in 3-4 calls)
            run_d()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            r, (H, inl, cnt) = run_d()
            ts.append(time.perf_counter() - t0)
        lr = r.last_run
        idx = np.asarray(lr["idx"])[:, :4]
        rep = int(((idx[:, 0] == idx[:, 1]) | (idx[:, 0] == idx[:, 2]) | (idx[:, 0] == idx[:, 3]) | (idx[:, 1] == idx[:, 2]) |
                   (idx[:, 1] == idx[:, 3]) | (idx[:, 2] == idx[:, 3])).sum())
        out.setdefault("dense_sets K=10000", {})[name] = {
            "correspondences": M, "ms_per_run": round(sorted(ts)[len(ts) // 2] * 1e3, 3), "inliers": int(cnt),
            "flagged_by_k1_share": round(lr.get("flagged", 0) / K, 4), "host_solved_share": round(lr.get("host_settled", 0) / K, 4),
            "repeated_index_share": round(rep / K, 4), "count_intervals": lr.get("intervals")}
    out["note"] = ("RANSAC.run: bit-exact inlier sets (tests: 19 reference runs + g10 + g12 + dense clouds vs the oracle).  The K=... entries beside "
                   "this one time rwh_ransac_search alone (K1 + K2 + argmax + 16-byte readback): the raw search, no host solver.  "
                   "host_solved = repeated-index samples (always) + what the count intervals of round 4 leave undecided.")
    return out


GpuBackend.parity_path_leg = _parity_path_leg
GpuBackend.scaling_set_leg = _scaling_set_leg
GpuBackend.other_kernels_leg = lambda self, w, h, nb: _other_kernels_leg(self, w, h, nb)
GpuBackend.config4_leg = lambda self: _config4_leg(self)


def timed(backend, step, steps, warmup, sync_all, prewarm_ms=0.0):
    """W untimed warm-up steps, then EXACTLY K timed steps bracketed by barrier + synchronize.  -> (wall seconds, kernel ms
    per step by HIP events on the launch stream)."""
    if prewarm_ms:
        t_ramp = time.perf_counter()
        while (time.perf_counter() - t_ramp) * 1e3 < prewarm_ms:
            for _ in range(10):
                step()
            backend.sync()
    for _ in range(warmup):
        step()
    sync_all()
    ev0, ev1 = backend.events()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        step()
    ev1.record()
    sync_all()
    return time.perf_counter() - t0, ev0.elapsed_time(ev1) / steps


def run_rank(args, backend, dist=None):
    """One rank of the benchmark.  `dist`: torch.distributed, already initialised when world > 1 (None: single rank).
    Returns the JSON line (a dict) on rank 0, None elsewhere."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    src_w, src_h = (int(v) for v in args.src.lower().split("x"))
    k, sharded = backend.kernels, backend.sharded

    def sync_all():
        if world > 1:
            dist.barrier()
        backend.sync()

    def max_over_ranks(seconds):
        if world == 1:
            return seconds
        tt = backend.tensor([seconds], "float64")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    # ---- headline: weak scaling, every rank warps its own batch ---------------------------------------------------
    # Clock ramp: an idle MI355X needs tens of milliseconds of work before it runs at its sustained clocks (measured:
    # 0.60 ms per step in the first 10 steps after the Python set-up, 0.49 ms from ~50 steps on).  Keep the GPU busy for
    # PREWARM_MS first, untimed like the warm-up steps, so that K timed steps measure the steady state whatever K is.
    B = args.frames
    step, out_h, out_w, _, plan, keep = backend.make_warp(B, src_w, src_h, 1234 + rank)
    elapsed, kernel_ms = timed(backend, step, args.steps, args.warmup, sync_all, PREWARM_MS)
    elapsed = max_over_ranks(elapsed)
    value = world * B * out_h * out_w / 1e6 * args.steps / elapsed
    alg_bytes = B * (3 * src_h * src_w + 3 * out_h * out_w)
    load = backend.load_facts(step) if hasattr(backend, "load_facts") else {}       # every rank: its own chip's clock and power
    per_rank_load = None
    if world > 1 and load:
        t = backend.tensor([0.0] * (2 * world), "float64")
        t[2 * rank], t[2 * rank + 1] = float(load.get("sclk_mhz") or 0.0), float(load.get("power_w") or 0.0)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        v = [float(x) for x in t.tolist()]
        per_rank_load = {"sclk_mhz": [round(x) for x in v[0::2]], "power_w": [round(x, 1) for x in v[1::2]]}
    del keep, step

    extras = {}
    if not args.no_extras:
        # ---- the other kernels of the warp entry point on the same workload (informational, N = 1) ----------------
        if world == 1:
            extras["other_warp_kernels"] = backend.other_kernels_leg(src_w, src_h, min(B, 16))
        # ---- the 8K warp (north_star's target configuration) and its strong-scaling form -------------------------
        F8 = 8
        FS8 = 8 if world == 1 else 64          # strong leg: at N > 1 a step must stay a bandwidth measurement (8 frames / 8 ranks = 43 us: launch latency)
        if world == 1:
            step8, oh8, ow8, _, plan8, keep8 = backend.make_warp(F8, 7680, 4320, 4321)
            _, ms8 = timed(backend, step8, 20, 5, sync_all, PREWARM_MS)
            by8 = F8 * 3 * (4320 * 7680 + oh8 * ow8)
            extras["warp_8k"] = {"workload": "%d frames 7680x4320 RGB u8 -> %dx%d u8 per launch" % (F8, oh8, ow8), "kernel": plan8,
                                 "kernel_ms": round(ms8, 4), "mpix_per_s": round(F8 * oh8 * ow8 / ms8 / 1e3, 1),
                                 "algorithmic_bytes_per_launch": by8, "achieved": round(by8 / ms8 / 1e6, 1), "unit": "GB/s",
                                 "frac": round(by8 / ms8 / 1e6 / HBM_PEAK_GBS, 4),
                                 "read_only_frac": round(F8 * 3 * 4320 * 7680 / ms8 / 1e6 / HBM_PEAK_GBS, 4)}
            del keep8, step8
        steps8, oh8, ow8, rows8, plan8, keep8 = backend.make_warp(FS8, 7680, 4320, 4321, rows_of=(rank, world))
        el8, ms8 = timed(backend, steps8, 10, 3, sync_all, PREWARM_MS)
        el8 = max_over_ranks(el8)
        extras["strong_8k"] = {"workload": "%d frames 7680x4320 -> %dx%d, output rows sharded over %d rank(s) (row tiles, no collective)"
                                           % (FS8, oh8, ow8, world), "scaling": "strong", "rows_of_rank0": list(rows8),
                               "mpix_per_s": round(FS8 * oh8 * ow8 / 1e6 * 10 / el8, 1), "ms_per_step": round(el8 / 10 * 1e3, 4),
                               "kernel": plan8}
        del keep8, steps8
        # ---- BASELINE config 5's warp: 512 distinct 1080p frames, sharded by image over the ranks, one launch each ------
        F5 = 512 // world
        step5, oh5, ow5, _, plan5, keep5 = backend.make_warp(F5, 1920, 1080, 5000 + rank)
        el5, ms5 = timed(backend, step5, 5, 2, sync_all, PREWARM_MS)
        el5 = max_over_ranks(el5)
        by5 = F5 * 3 * (1080 * 1920 + oh5 * ow5)
        extras["config5_batch_1080p"] = {"workload": "512 frames 1920x1080 RGB u8 -> %dx%d u8, %d per rank in ONE launch (sharded by image, no collective)"
                                                     % (oh5, ow5, F5), "scaling": "strong", "kernel": plan5, "kernel_ms_rank0": round(ms5, 4),
                                         "mpix_per_s": round(world * F5 * oh5 * ow5 / 1e6 * 5 / el5, 1),
                                         "frac_rank0": round(by5 / ms5 / 1e6 / HBM_PEAK_GBS, 4)}
        del keep5, step5
        # ---- RANSAC ---------------------------------------------------------------------------------------------
        extras.update(ransac_legs(backend, dist, world, rank, sync_all, max_over_ranks))
        if world == 1:
            c4 = backend.config4_leg()
            if c4:
                extras["config4_panorama_8k"] = c4

    seen = [world]
    if world > 1:     # what every rank believes the world to be (the driver checks the RCCL job really had N ranks)
        t = backend.tensor([0] * world)
        t[rank] = dist.get_world_size()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        seen = [int(v) for v in t.tolist()]
    if rank != 0:
        return None
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "kernel": plan, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes_per_launch": alg_bytes,
            "kernel_ms": round(kernel_ms, 4),
            "read_only_frac": round(B * 3 * src_h * src_w / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    roof.update(profile_facts(plan, B, (src_w, src_h), kernel_ms, load.get("sclk_mhz")))
    roof.update(load)
    if per_rank_load:
        roof["per_rank"] = per_rank_load
    if "warp_8k" in extras:      # the configuration the north_star's 0.70 is worded on, inside the roofline object
        roof["frac_8k"], roof["kernel_ms_8k"] = extras["warp_8k"]["frac"], extras["warp_8k"]["kernel_ms"]
    at_cap = bool(load.get("power_w") and load.get("power_cap_w") and load["power_w"] >= 0.97 * load["power_cap_w"])
    # what binds, from the run's own measurements + the committed counters: at the power cap with the VALU >= 90 % busy the kernel's
    # time follows its arithmetic energy ("valu+power"); below that it is the memory system ("hbm": peak = HBM3E, the contract's roofline)
    roof["bound"] = "valu+power" if (at_cap and (roof.get("valu_frac_pmc") or 0) > 0.9) else "hbm"
    if load.get("power_w") and load.get("power_cap_w") and load["power_w"] >= 0.97 * load["power_cap_w"]:
        # the launches run AT the board's power cap and the chip holds a clock below its 2.4 GHz maximum: the bound that
        # binds is energy per pixel (HBM 41 %, VALU 45 %, LDS 10 % of the energy of a launch)
        roof["limiter"] = ("board power cap: %.0f W of %.0f W, shader clock held at %.0f MHz of 2400; " % (load["power_w"], load["power_cap_w"], load["sclk_mhz"])
                           + (roof.get("limiter") or ""))
    line = {
        "metric": "backward-warp Mpixels/sec (+ RANSAC hypotheses/sec)", "value": round(value, 1), "unit": "Mpix/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8 (f64 coordinates, f32 blend)", "data": "synthetic",
        "config": {"workload": "transformImageH warp %dx%d RGB u8 bilinear -> %dx%d u8, %d distinct frames per "
                               "GPU per step, Hs mild perspective" % (src_w, src_h, out_h, out_w, B),
                   "frames_per_step_per_gpu": B, "sharding": "by image, no collective",
                   "untimed_clock_ramp_ms": PREWARM_MS},
        "roofline": roof,
        "rccl": {"backend": backend.collective if world > 1 else None, "world_size_seen_by_rank": seen},
    }
    line.update(extras)
    if "ransac" in line and "all_ranks_decoded" in line["ransac"].get("K=100000", {}):
        line["rccl"]["all_ranks_agree_on_winner"] = len(set(line["ransac"]["K=100000"]["all_ranks_decoded"])) == 1
    if not args.no_cpu and world == 1:
        line["cpu_baseline"] = cpu_baseline("%d-frame batch" % B, src_w, src_h)
    return line


PMC_JSON = "profiles/r04_pmc.json"


def profile_facts(kernel, frames, src_wh, kernel_ms, sclk_mhz=None):
    """Counter-derived facts of the dominant kernel from the committed rocprofv3 PMC summary of THIS command
    (PMC_JSON; counters cannot be read from inside the run): HBM traffic per launch and the VALU roofline --
    `valu`: VALU instructions per output pixel (SQ_INSTS_VALU / waves / 512 px x 64 lanes = lane-instructions per pixel)
    against 1024 SIMDs x 16 lanes x the shader clock MEASURED IN THIS RUN (every VALU instruction of this kernel's mix
    issues in >= 4 cycles per wavefront; only plain float32 / 32-bit integer adds, multiplies and moves take 2)."""
    path = os.path.join(ROOT, PMC_JSON)
    if not os.path.exists(path):
        return {}
    p = json.load(open(path)).get(kernel)
    if not p or tuple(p.get("src", ())) != tuple(src_wh):
        return {}
    out = {"traffic": int(p["hbm_bytes_per_launch"] * frames / p["frames"]) if p.get("hbm_bytes_per_launch") else None,
           "traffic_source": PMC_JSON + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"}
    if p.get("valu_insts_per_wave"):
        # instructions x measured issue cost / kernel cycles: waves per SIMD x instructions per wave x cycles per wave-
        # instruction (tools/valu_rates, this kernel's mix), against the kernel time of THIS run at the PMC pass's clock
        waves_per_simd = p["waves_per_launch"] * frames / p["frames"] / 1024.0
        cyc = waves_per_simd * p["valu_insts_per_wave"] * p["cycles_per_valu_inst"]
        out["valu_frac"] = round(cyc / (kernel_ms * 1e-3 * p["sclk_hz"]), 4)
        if sclk_mhz:
            waves = p["waves_per_launch"] * frames / p["frames"]
            lane_insts = waves * p["valu_insts_per_wave"] * 64.0
            peak = 1024 * 16 * sclk_mhz * 1e6
            out["valu"] = {"insts_per_wave": p["valu_insts_per_wave"], "lane_insts_per_pixel": round(p["valu_insts_per_wave"] / 8.0, 2),
                           "achieved": round(lane_insts / (kernel_ms * 1e-3) / 1e12, 2), "peak": round(peak / 1e12, 2),
                           "unit": "T lane-instructions/s at the measured %.0f MHz" % sclk_mhz,
                           "frac": round(lane_insts / (kernel_ms * 1e-3) / peak, 4)}
        out["valu_frac_pmc"] = p.get("valu_busy_frac")
        out["ta_busy_frac_pmc"] = p.get("ta_busy_frac")
        out["valu_insts_per_wave"] = p["valu_insts_per_wave"]
        out["limiter"] = p.get("limiter")
    return out


def pmc_row(kernel):
    """The committed counter row (PMC_JSON, `other_kernels`: rocprofv3 --pmc passes of this command) of a kernel that is not the
    headline one: which unit is busiest, i.e. what limits it (round-3 verdict item 3)."""
    path = os.path.join(ROOT, PMC_JSON)
    if not os.path.exists(path) or not kernel:
        return {}
    for key, row in json.load(open(path)).get("other_kernels", {}).items():
        if key.startswith(kernel + "(") or key.startswith(kernel + " |"):
            return {"pmc": {"valu_busy": row.get("valu_busy_frac"), "l2_busy": row.get("tcc_busy_frac"), "ta_busy": row.get("ta_busy_frac"),
                            "lds_busy": row.get("lds_busy_frac"), "l1_waiting_for_l2": row.get("tcp_pending_stall_frac"),
                            "hbm_bytes_per_launch": row.get("hbm_bytes_per_launch"), "l2_write_requests": row.get("l2_write_requests"),
                            "limiter": row.get("limiter"), "source": PMC_JSON}}
    return {}


def _other_kernels_leg(backend, src_w, src_h, nb):
    torch, k = backend.torch, backend.kernels
    step, out_h, out_w, _, _, (src, _, grid, inv) = backend.make_warp(nb, src_w, src_h, 99)
    other = {}
    for name, interp, dt, out_bytes in (("nearest_u8", "nn", torch.uint8, 3), ("bilinear_f32_out", "bilinear", torch.float32, 12)):
        d2 = torch.empty((nb, out_h, out_w, 3), dtype=dt, device=backend.dev)

        def step2():
            k.warp_backward(src, inv, grid, (src_h, src_w), interp, dt, zero_origin=False, out=d2)
        _, ms = timed(backend, step2, 20, 20, backend.sync, PREWARM_MS)
        byt = nb * (3 * src_h * src_w + out_bytes * out_h * out_w)
        other[name] = {"kernel": k.warp_plan((nb, src_h, src_w, 3), torch.uint8, inv, grid, (src_h, src_w), interp, dt),
                       "mpix_per_s": round(nb * out_h * out_w / ms / 1e3, 1), "ms_per_launch": round(ms, 4), "frames": nb,
                       "achieved_GBps": round(byt / ms / 1e6, 1), "frac_of_hbm_peak": round(byt / ms / 1e6 / HBM_PEAK_GBS, 4)}
        if hasattr(backend, "load_facts"):      # what limits it: the clock the chip holds and the board power under this kernel
            other[name].update(backend.load_facts(step2, 0.7))
        del d2
    # minification (scanner mode shrinks A4 scans).  1.5x: a 64 x 8 patch's footprint (98 x 14 texels) no longer fits a 5 KB window, the
    # patch is staged by halves; 2.5x: s^2 = 6.25 staged texels per output pixel against 4 taps used -- staging stops paying at ~1.8x,
    # masked gathers (whose cost per OUTPUT pixel does not depend on s, while the roofline's bytes per output pixel grow as s^2)
    for zname, zs, zpath in (("zoom_out_1p5", 1.5, "staged by half patches (16-wide halves fit the window up to ~1.8x)"),
                             ("zoom_out_2p5", 2.5, "masked gathers (a staged window would hold 6.25 texels per output pixel for the 4 it uses)")):
        zin = np.array([[zs, 0.0, 0.0], [0.0, zs, 0.0], [0.0, 0.0, 1.0]])       # inv(H): output pixel -> source texel
        zw, zh = int(src_w / zs), int(src_h / zs)
        zgrid = k.Grid(0, zw - 1, zw, 0, zh - 1, zh)
        dz = torch.empty((nb, zh, zw, 3), dtype=torch.uint8, device=backend.dev)

        def stepz():
            k.warp_backward(src, zin, zgrid, (src_h, src_w), "bilinear", torch.uint8, zero_origin=False, out=dz)
        _, ms = timed(backend, stepz, 20, 20, backend.sync, PREWARM_MS)
        byt = nb * 3 * (src_h * src_w + zh * zw)
        other[zname] = {"kernel": k.warp_plan((nb, src_h, src_w, 3), torch.uint8, zin, zgrid, (src_h, src_w), "bilinear", torch.uint8),
                        "path": zpath, "mpix_per_s": round(nb * zh * zw / ms / 1e3, 1), "ms_per_launch": round(ms, 4), "frames": nb,
                        "achieved_GBps": round(byt / ms / 1e6, 1), "frac_of_hbm_peak": round(byt / ms / 1e6 / HBM_PEAK_GBS, 4)}
        if hasattr(backend, "load_facts"):
            other[zname].update(backend.load_facts(stepz, 0.5))
        del dz
    # 4-channel images (the reference's RGBA = float32 with the alpha plane of addAlpha; uint8 RGBA for completeness): the
    # generic gather kernel -- 16-byte float32 texels gather well, there is no staged kernel for them
    nb4 = min(nb, 8)
    for name, sdt in (("rgba_f32_bilinear", torch.float32), ("rgba_u8_bilinear", torch.uint8)):
        s4 = (torch.rand((nb4, src_h, src_w, 4), device=backend.dev) * 255).to(sdt)
        d4 = torch.empty((nb4, out_h, out_w, 4), dtype=sdt, device=backend.dev)

        def step4():
            k.warp_backward(s4, inv, grid, (src_h, src_w), "bilinear", sdt, zero_origin=False, out=d4)
        _, ms = timed(backend, step4, 10, 5, backend.sync, PREWARM_MS)
        byt = nb4 * 4 * s4.element_size() * (src_h * src_w + out_h * out_w)
        other[name] = {"kernel": k.warp_plan((nb4, src_h, src_w, 4), sdt, inv, grid, (src_h, src_w), "bilinear", sdt),
                       "mpix_per_s": round(nb4 * out_h * out_w / ms / 1e3, 1), "ms_per_launch": round(ms, 4), "frames": nb4,
                       "achieved_GBps": round(byt / ms / 1e6, 1), "frac_of_hbm_peak": round(byt / ms / 1e6 / HBM_PEAK_GBS, 4)}
        del s4, d4
    for row in other.values():
        row.update(pmc_row(row.get("kernel")))
    return other


def ransac_legs(backend, dist, world, rank, sync_all, max_over_ranks):
    """BASELINE config 3 is K = 10 000 on matchespoints; one run costs ~35 us of launch + 16-byte readback latency whatever
    K is, so K = 100 000 (config 5's size) is reported beside it.  N > 1: the hypothesis range is sharded over the ranks and
    ONE all-reduce(MAX) of 2 x int64 picks the winner (the raw K1 + K2 search: the parity API's host settle step,
    ransac.RANSAC.run, is timed in config4_panorama_8k)."""
    k, sharded = backend.kernels, backend.sharded
    report = {}
    for K in ((10000, 100000) if world == 1 else (100000,)):
        b, e = sharded.shard_range(K, rank, world)
        # () -> this rank's two packed keys (2 x int64) on the collective's device; N > 1: with the rank's host settle step
        search = backend.make_search(K, b, e, settle=True) if world > 1 else backend.make_search(K, b, e)

        def ransac_step():
            best = search()
            if world > 1:
                dist.all_reduce(best, op=dist.ReduceOp.MAX)   # the ONE collective of the sharded RANSAC: 2 x int64, MAX
            return best.cpu()             # the 16-byte result reaches the host: launch + readback latency included

        for _ in range(3):
            ransac_step()
        sync_all()
        t0 = time.perf_counter()
        R = 20
        for _ in range(R):
            best = ransac_step()
        sync_all()
        tr = max_over_ranks(time.perf_counter() - t0)
        winner, cnt, early = k.decode_best(best.numpy(), K)
        entry = {"hyp_per_s": round(K * R / tr, 1), "us_per_run": round(tr / R * 1e6, 1),
                 "pair_evals_per_s": round(K * R * 185 / tr, 1), "winner": winner, "winner_count": cnt}
        if world > 1:
            t = backend.tensor([0] * world)
            t[rank] = -1 if winner is None else winner
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            entry["all_ranks_decoded"] = [int(v) for v in t.tolist()]
        report["K=%d" % K] = entry

    if world == 1 and hasattr(backend, "parity_path_leg"):
        report["parity_path"] = backend.parity_path_leg()
    if world == 1:
        report.update(backend.batched_search_leg(sync_all))
        if hasattr(backend, "scaling_set_leg"):
            report.update(backend.scaling_set_leg(sync_all))
    report["correspondences"] = 185
    report["includes"] = ("K1 (also clears the keys) + K2 + argmax pass + 16-byte readback per run (the raw search; `parity_path` times RANSAC.run)" if world == 1 else
                          "per rank: K1 + K2 + argmax pass over its slice + the host settle step (reference SVD for flagged / near-best hypotheses), "
                          "then the ONE all-reduce(max) of 2 x int64 + 16-byte readback: the bit-exact path")
    return {"ransac": report}


def _config4_leg(backend):
    """BASELINE config 4: foto1A / foto1B upsampled x8 (8192 x 5464), RANSAC (app.py parameters, threshold scaled) +
    warp + composite, end to end."""
    torch = backend.torch
    fpath = os.path.join(ROOT, "tests", "golden", "img_foto1.npz")
    if not os.path.exists(fpath):
        return None
    import homography as hg
    import ransac as rs
    torch.cuda.empty_cache()
    f = np.load(fpath)
    z = np.load(os.path.join(ROOT, "tests", "golden", "matchespoints.npz"))
    A8 = np.ascontiguousarray(np.repeat(np.repeat(f["A"], 8, axis=0), 8, axis=1))
    B8 = np.ascontiguousarray(np.repeat(np.repeat(f["B"], 8, axis=0), 8, axis=1))
    X8, Y8 = (z["ptsA"] * 8).T.copy(), (z["ptsB"] * 8).T.copy()
    runner = {}

    def search():
        np.random.seed(0)
        with contextlib.redirect_stdout(io.StringIO()):   # RANSAC.run prints the reference's "Warning::" line (ransac.py:204)
            r = rs.RANSAC(rs.HomoModel(th=32, d=95, n=4), k=1500)
            runner["r"] = r
            return r.run([X8, Y8], method="fwd")

    search()
    t0 = time.perf_counter()
    for _ in range(5):
        H8, inl8, c8 = search()
    t_search = (time.perf_counter() - t0) / 5
    A8d, B8d = torch.from_numpy(A8).to(backend.dev), torch.from_numpy(B8).to(backend.dev)
    res, res_ev = {}, {}
    for mode, kw in (("paste", {}), ("rate_blend", {"blending": "Rate", "blendrate": 0.2})):
        with contextlib.redirect_stdout(io.StringIO()):
            for _ in range(20):                 # clock ramp + allocator warm-up (the canvas is a fresh 250 MB tensor per call)
                canvas = hg.stitchPanorama(B8d, A8d, H8, **kw)
            torch.cuda.synchronize()
            ev0, ev1 = backend.events()
            t0 = time.perf_counter()
            ev0.record()
            for _ in range(20):
                canvas = hg.stitchPanorama(B8d, A8d, H8, **kw)
            ev1.record()
            torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / 20           # wall clock per Python call (host geometry, allocation, launch)
        res_ev[mode] = ev0.elapsed_time(ev1) / 20             # GPU time per call by HIP events on the launch stream
    t_hosts = []
    for _ in range(7):      # the first two calls page-lock the staging ring and the two result blocks (_xfer); then steady state
        A8c = A8.copy()     # (stitchPanorama blanks texel (0,0) of the caller's imgT like the reference: hand it a copy, untimed)
        t0 = time.perf_counter()
        out_np = hg.stitchPanorama(B8, A8c, H8)
        t_hosts.append(time.perf_counter() - t0)
    t_host = min(t_hosts[2:])
    last = runner["r"].last_run
    return {"images": "%dx%d + %dx%d RGB u8" % (A8.shape[1], A8.shape[0], B8.shape[1], B8.shape[0]),
            "canvas": "%dx%d" % (out_np.shape[1], out_np.shape[0]), "inliers": int(c8),
            "ransac_k1500_ms": round(t_search * 1e3, 3), "ransac_host_settled_hypotheses": last.get("host_settled"),
            "stitch_resident_ms": round(res["paste"] * 1e3, 3), "stitch_rate_blend_resident_ms": round(res["rate_blend"] * 1e3, 3),
            "stitch_gpu_ms": round(res_ev["paste"], 4), "stitch_rate_blend_gpu_ms": round(res_ev["rate_blend"], 4),
            "stitch_canvas_mpix_per_s": round(out_np.shape[0] * out_np.shape[1] / res["paste"] / 1e6, 1),
            "stitch_from_host_arrays_ms": round(t_host * 1e3, 1), "stitch_from_host_arrays_first_ms": round(t_hosts[0] * 1e3, 1),
            "stitch_from_host_arrays_steady_median_ms": round(sorted(t_hosts[2:])[len(t_hosts[2:]) // 2] * 1e3, 1),
            "note": "RANSAC.run incl. numpy sampling, uploads, readback, the host SVD settle step and the host refit; stitch = "
                    "compositor kernel on resident tensors (tensors in: the fast kernels; *_resident_ms = wall clock per Python call, *_gpu_ms = HIP events); from host arrays (exact float64 "
                    "kernel) adds 2 x 134 MB up + canvas down over PCIe (page-locked staging ring + host copy threads; uploads, composition by row tiles and the download overlapped: homography._stitch_pipelined); "
                    "stitch_from_host_arrays_ms = the MINIMUM of calls 3-7 (warm staging ring, cached page-locked result blocks), _first_ms = the cold first call, _steady_median_ms = the median of calls 3-7"}


# ------------------------------------------------------------------------------------------------------------------
def self_launch(args, argv):
    """`python bench.py --gpus N` typed as is: start N rank processes (one per GPU) with torch.distributed.run on the
    loopback interface and pass their output through.  Nothing in this process has touched the GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started inside a job of WORLD_SIZE=%d" % (args.gpus, world))
    import torch.distributed as dist
    rehearsal = os.environ.get("RWH_BENCH_REHEARSAL") == "1"
    backend = GpuBackend(int(os.environ.get("LOCAL_RANK", "0")), rehearsal)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend.collective, **backend.init_args())
    line = run_rank(args, backend, dist if world > 1 else None)
    if line is not None:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
