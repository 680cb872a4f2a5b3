#!/usr/bin/env python3
"""bench.py -- backward-warp Mpixels/s (+ RANSAC hypotheses/s) on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
  transformImageH-style warp of 3840x2160 RGB u8 frames with the survey's mild-perspective Hs,
  bilinear, uint8 out (output 2028x3771 per frame, the reference's auto-bounds geometry).
  One "step" = ONE launch of the warp kernel over a batch of FRAMES distinct frames already
  resident in HBM (batch source + destination = 1.5 GB > the 256 MB Infinity Cache, so the
  traffic is HBM traffic, not cache traffic).
  N > 1: every rank warps its own batch (shard by image, no collective): weak scaling.
Also reported in the same JSON line (extra keys): RANSAC hypotheses/s on matchespoints
(10 000 hypotheses per GPU-step at N=1; 100 000 sharded over the ranks with the one
all-reduce at N>1), the roofline of the dominant kernel, and the numpy CPU path timed on this
box's host cores on a bounded sample.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H_S = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
SRC_H, SRC_W = 2160, 3840
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
PREWARM_MS = 150.0    # untimed clock ramp before the warm-up steps (see main)


def cpu_baseline(frames_note):
    """The numpy CPU path (oracle = bit-identical restatement of the reference) on ONE frame of the
    same workload, min of 3 after 1 warm-up, single process (numpy's elementwise ops are single
    threaded).  Plus the RANSAC loop body on 300 hypotheses."""
    from oracle import rwh_oracle as orc
    img = np.random.default_rng(1234).integers(0, 256, (SRC_H, SRC_W, 3), dtype=np.uint8)
    ts = []
    out = None
    for i in range(9):      # 1 warm-up + 8 timed frames: ~8-25 s of CPU work depending on the host
        t0 = time.perf_counter()
        out, _, _ = orc.transform_image_h(img, H_S)
        ts.append(time.perf_counter() - t0)
        if i >= 3 and sum(ts) > 20.0:
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
                      "os.cpu_count=%d OPENBLAS_NUM_THREADS=%s" % (SRC_W, SRC_H, out.shape[0], out.shape[1], frames_note, len(ts) - 1,
                                                                   os.cpu_count(), os.environ.get("OPENBLAS_NUM_THREADS", "unset")),
            "ransac_hyp_per_s": round(2000 / tr, 1), "ransac_sample": "2000 hypotheses x 185 correspondences, fwd"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=32, help="4K frames per step (per GPU)")
    ap.add_argument("--src", default="3840x2160", help="source frame WxH (default: the BASELINE 4K configuration)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    args = ap.parse_args()
    global SRC_H, SRC_W
    SRC_W, SRC_H = (int(v) for v in args.src.lower().split("x"))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    # Rehearsal on a one-GPU box (never used by the driver): RWH_BENCH_REHEARSAL=1 maps every rank to cuda:0 and
    # swaps RCCL for gloo (RCCL refuses two ranks on one device); the all-reduces then go through host tensors.
    rehearsal = os.environ.get("RWH_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def all_reduce_max(t):
        """In-place MAX all-reduce of a GPU tensor (RCCL; via the host under the gloo rehearsal)."""
        if rehearsal:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)

    from ransac_with_homography_amd import kernels, sharded
    from ransac_with_homography_amd.homography import _bounds

    # ---- warp workload ---------------------------------------------------------------------
    B = args.frames
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    src = torch.randint(0, 256, (B, SRC_H, SRC_W, 3), dtype=torch.uint8, device=dev, generator=gen)
    min_x, min_y, out_w, out_h = _bounds(SRC_H, SRC_W, H_S, 0)
    grid = kernels.Grid(min_x, min_x + out_w - 1, out_w, min_y, min_y + out_h - 1, out_h)
    inv = np.linalg.inv(H_S)
    dst = torch.empty((B, out_h, out_w, 3), dtype=torch.uint8, device=dev)

    src[:, 0, 0, :] = 0  # texel (0,0) is blanked once (homography.py:126-130); the timed launches are the warp kernel alone

    def step():
        kernels.warp_backward(src, inv, grid, (SRC_H, SRC_W), "bilinear", torch.uint8, zero_origin=False, out=dst)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Clock ramp: an idle MI355X needs tens of milliseconds of work before it runs at its sustained clocks (measured:
    # 0.60 ms per step in the first 10 steps after the Python set-up, 0.49 ms from ~50 steps on).  Keep the GPU busy for
    # PREWARM_MS first, untimed like the warm-up steps, so that K timed steps measure the steady state whatever K is.
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < PREWARM_MS:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    sync_all()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    sync_all()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_reduce_max(tt)
        elapsed = float(tt.item())
    mpix_step = B * out_h * out_w / 1e6
    value = world * mpix_step * args.steps / elapsed

    # ---- the other kernels of the warp entry point on the same workload (informational, N = 1) -----------------------
    other = {}
    if world == 1:
        for name, interp, dt, out_bytes in (("nearest_u8", "nn", torch.uint8, 3), ("bilinear_f32_out", "bilinear", torch.float32, 12)):
            nb = min(B, 16)
            d2 = torch.empty((nb, out_h, out_w, 3), dtype=dt, device=dev)
            s2 = src[:nb]

            def step2():
                kernels.warp_backward(s2, inv, grid, (SRC_H, SRC_W), interp, dt, zero_origin=False, out=d2)

            for _ in range(20):
                step2()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                step2()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            byt = nb * (3 * SRC_H * SRC_W + out_bytes * out_h * out_w)
            other[name] = {"mpix_per_s": round(nb * out_h * out_w / ms / 1e3, 1), "ms_per_launch": round(ms, 4), "frames": nb,
                           "achieved_GBps": round(byt / ms / 1e6, 1), "frac_of_hbm_peak": round(byt / ms / 1e6 / HBM_PEAK_GBS, 4)}
            del d2

    # ---- RANSAC workload -----------------------------------------------------------------------
    # BASELINE config 3 is K = 10 000 on matchespoints; one run costs ~80 us of launch + 16-byte readback latency on
    # this system whatever K is, so K = 100 000 (config 5's size) is reported beside it.  N > 1: the hypothesis range
    # is sharded over the ranks and ONE all-reduce(MAX) of 2 x int64 picks the winner.
    z = np.load(os.path.join(ROOT, "tests", "golden", "matchespoints.npz"))
    pa, pb = torch.from_numpy(z["ptsA"]).to(dev), torch.from_numpy(z["ptsB"]).to(dev)
    need = kernels.need_count(185, 70, 4)
    ransac_report = {}
    for K in ((10000, 100000) if world == 1 else (100000,)):
        np.random.seed(0)
        idx_table = np.random.randint(0, 185, (K, 4))
        b, e = sharded.shard_range(K, rank, world)
        idx_dev = torch.from_numpy(idx_table[b:e].astype(np.int32)).to(dev)
        ws = kernels.SearchWorkspace(e - b, 185, dev, want_masks=False)

        def ransac_step():
            kernels.ransac_search(pa, pb, idx_dev, 5.0, "fwd", need, ws, hyp_base=b)   # key reset + K1 + K2, one call
            if world > 1:
                all_reduce_max(ws.best)   # the ONE collective of the sharded RANSAC: 2 x int64, MAX
            return ws.best.cpu()          # the 16-byte result reaches the host: launch + readback latency included

        for _ in range(3):
            ransac_step()
        sync_all()
        t0 = time.perf_counter()
        R = 20
        for _ in range(R):
            best = ransac_step()
        sync_all()
        tr = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([tr], dtype=torch.float64, device=dev)
            all_reduce_max(tt)
            tr = float(tt.item())
        winner, cnt, early = kernels.decode_best(best.numpy(), K)
        ransac_report["K=%d" % K] = {"hyp_per_s": round(K * R / tr, 1), "us_per_run": round(tr / R * 1e6, 1),
                                     "pair_evals_per_s": round(K * R * 185 / tr, 1), "winner": winner, "winner_count": cnt}

    # Batched mode (rwh_ransac_batched): 64 independent copies of the problem x 10 000 hypotheses each in ONE submission,
    # device Philox sampling -- the throughput figure once the per-run launch + readback latency is amortised.
    if world == 1:
        P, K = 64, 10000
        offs = torch.arange(0, 185 * (P + 1), 185, dtype=torch.int32, device=dev)
        pa_b, pb_b = pa.repeat(P, 1), pb.repeat(P, 1)
        needs = torch.full((P,), need, dtype=torch.int32, device=dev)
        bws = kernels.BatchWorkspace(P, K, 185, dev, want_masks=False)

        def batched_step():
            kernels.ransac_batched(pa_b, pb_b, offs, needs, 5.0, "fwd", bws, seed=2024)
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
        cnts = [kernels.decode_best(bb[p].numpy(), K)[1] for p in range(P)]
        ransac_report["batched P=%d K=%d" % (P, K)] = {
            "hyp_per_s": round(P * K * R / tr, 1), "us_per_run": round(tr / R * 1e6, 1),
            "pair_evals_per_s": round(P * K * R * 185 / tr, 1), "winner_count_min_max": [int(min(cnts)), int(max(cnts))],
            "sampling": "device Philox4x32-10, 4 distinct correspondences (non-parity mode)"}

    # BASELINE config 4: foto1A / foto1B upsampled x8 (8192 x 5464), RANSAC (app.py parameters, threshold scaled) +
    # fused warp/paste, end to end.  Reported beside the headline; the stitch kernel is the EXACT float64 compositor
    # (bit-identical canvases), not the fast warp kernel.
    config4 = None
    fpath = os.path.join(ROOT, "tests", "golden", "img_foto1.npz")
    if world == 1 and os.path.exists(fpath):
        import homography as hg
        import ransac as rs
        del src, dst
        torch.cuda.empty_cache()
        f = np.load(fpath)
        A8 = np.ascontiguousarray(np.repeat(np.repeat(f["A"], 8, axis=0), 8, axis=1))
        B8 = np.ascontiguousarray(np.repeat(np.repeat(f["B"], 8, axis=0), 8, axis=1))
        X8, Y8 = (z["ptsA"] * 8).T.copy(), (z["ptsB"] * 8).T.copy()

        def search():
            np.random.seed(0)
            with contextlib.redirect_stdout(io.StringIO()):   # RANSAC.run prints the reference's "Warning::" line (ransac.py:204)
                return rs.RANSAC(rs.HomoModel(th=32, d=95, n=4), k=1500).run([X8, Y8], method="fwd")

        search()
        t0 = time.perf_counter()
        for _ in range(5):
            H8, inl8, c8 = search()
        t_search = (time.perf_counter() - t0) / 5
        A8d, B8d = torch.from_numpy(A8).to(dev), torch.from_numpy(B8).to(dev)
        canvas = hg.stitchPanorama(B8d, A8d, H8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            canvas = hg.stitchPanorama(B8d, A8d, H8)
        torch.cuda.synchronize()
        t_res = (time.perf_counter() - t0) / 5
        t0 = time.perf_counter()
        out_np = hg.stitchPanorama(B8, A8.copy(), H8)
        t_host = time.perf_counter() - t0
        config4 = {"images": "%dx%d + %dx%d RGB u8" % (A8.shape[1], A8.shape[0], B8.shape[1], B8.shape[0]),
                   "canvas": "%dx%d" % (out_np.shape[1], out_np.shape[0]), "inliers": int(c8),
                   "ransac_k1500_ms": round(t_search * 1e3, 3), "stitch_resident_ms": round(t_res * 1e3, 3),
                   "stitch_canvas_mpix_per_s": round(out_np.shape[0] * out_np.shape[1] / t_res / 1e6, 1),
                   "stitch_from_host_arrays_ms": round(t_host * 1e3, 1),
                   "note": "RANSAC.run incl. numpy sampling, uploads, readback and the host refit; stitch = exact float64 "
                           "compositor kernel on resident tensors; from host arrays adds 2 x 134 MB up + canvas down over PCIe"}
        del A8d, B8d, canvas

    if rank == 0:
        alg_bytes = B * (3 * SRC_H * SRC_W + 3 * out_h * out_w)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            t32 = json.load(open(tpath)).get("warp_rgb8_bilinear_u8_bytes_per_launch")   # measured: 32 4K frames per launch
            traffic = int(t32 * B / 32) if (t32 and (SRC_W, SRC_H) == (3840, 2160)) else None
        line = {
            "metric": "backward-warp Mpixels/sec (+ RANSAC hypotheses/sec)", "value": round(value, 1), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8 (f64 coordinates, f32 blend)", "data": "synthetic",
            "config": {"workload": "transformImageH warp %dx%d RGB u8 bilinear -> %dx%d u8, %d distinct frames per "
                                   "GPU per step, Hs mild perspective" % (SRC_W, SRC_H, out_h, out_w, B),
                       "frames_per_step_per_gpu": B, "sharding": "by image, no collective",
                       "untimed_clock_ramp_ms": PREWARM_MS},
            "roofline": {"bound": "hbm", "kernel": "rwh::warp_rgb8_fast8<unsigned char, 6>", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel_ms": round(kernel_ms, 4),
                         "limiter": "VALU issue: 440 VALU instructions per 512-pixel wave, VALU active 81-84 % of the kernel "
                                    "(rocprofv3 PMC, profiles/r01_pmc_valu.txt); HBM traffic = 1.0003 x algorithmic",
                         "read_only_frac": round(B * 3 * SRC_H * SRC_W / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "ransac": dict(ransac_report, correspondences=185,
                           includes="K1 (also clears the keys) + K2 + argmax pass%s + 16-byte readback per run" % (" + all-reduce(max)" if world > 1 else "")),
        }
        if other:
            line["other_warp_kernels"] = other
        if config4:
            line["config4_panorama_8k"] = config4
        if not args.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline("%d-frame batch" % B)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
