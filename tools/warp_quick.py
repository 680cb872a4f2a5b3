"""Developer probe: time the uint8 bilinear warp of the BASELINE workload per kernel kind (rwh_lab_tune).
   python tools/warp_quick.py [kinds...]      kinds: 0 = the library's own choice, 5/6/7 = patch width 32/64/128, 13/14 = 32/64 staged by halves (default: 0 5 6 7)
   RWH_LIB=<path to another build of librwh_hip.so> (ablation builds), N=<timed launches>; prints the shader clock the
   chip held during the timed launches (rwh_lab_clock_probe)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_with_homography_amd import _lib, kernels
from ransac_with_homography_amd import homography as hg
if os.environ.get("RWH_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["RWH_LIB"])
H_S = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
if "ROT" in os.environ or "SCALE" in os.environ:      # rotation by ROT degrees / zoom by SCALE about the image centre instead
    t, sc = np.deg2rad(float(os.environ.get("ROT", "0"))), float(os.environ.get("SCALE", "1"))
    W0, H0 = (int(v) for v in os.environ.get("SRC", "3840x2160").split("x"))
    c, s_ = np.cos(t) * sc, np.sin(t) * sc
    H_S = np.array([[c, -s_, W0 / 2 - c * W0 / 2 + s_ * H0 / 2], [s_, c, H0 / 2 - s_ * W0 / 2 - c * H0 / 2], [0, 0, 1.0]])
frames = int(os.environ.get("FRAMES", "32"))
W, Hh = (int(v) for v in os.environ.get("SRC", "3840x2160").split("x"))
dev = _lib.require_gpu()
g = torch.Generator(device="cpu").manual_seed(1)
src = torch.randint(0, 256, (frames, Hh, W, 3), dtype=torch.uint8, generator=g).to(dev)
if os.environ.get("DATA", "noise") != "noise":       # DATA=photo: smooth synthetic frames (low-frequency waves + +-3 of noise); DATA=zeros
    # (a power-limited kernel's speed depends on how many bits toggle: uniform noise is the worst case, photographs are smooth)
    if os.environ["DATA"] == "zeros": src.zero_()
    else:
        yy, xx = torch.meshgrid(torch.arange(Hh, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
        for f in range(frames):
            for c in range(3):
                ph = 0.37 * f + 1.1 * c
                img = 128 + 70 * torch.sin(xx * (0.004 + 0.0007 * c) + ph) * torch.cos(yy * (0.006 + 0.0005 * f / frames) - ph) + 35 * torch.sin((xx + yy) * 0.021 + ph)
                img = img + torch.randint(-3, 4, (Hh, W), device=dev).float()
                src[f, :, :, c] = img.clamp(0, 255).to(torch.uint8)
        del yy, xx
# (rotation / zoom: the output grid is the source's own rectangle; BOUNDS=1: the warped image's bounding box, as the API takes it)
mx, my, ow, oh = (0, 0, W, Hh) if (("ROT" in os.environ or "SCALE" in os.environ) and not os.environ.get("BOUNDS")) else hg._bounds(Hh, W, H_S, 0)
ow = int(os.environ.get("OUTW", ow))          # OUTW=<n>: cut the output grid to n columns (store-alignment experiments)
grid = kernels.Grid(mx, mx + ow - 1, ow, my, my + oh - 1, oh)
inv = np.linalg.inv(H_S)
out = torch.empty((frames, oh, ow, 3), dtype=torch.uint8, device=dev)
lib = _lib.load()
ref = None
# MF=<a,b,..>: frames per block of the multi-frame kernel (rwh_lab_tune RWH_TUNE_WARP_FRAMES: 0 = the library's choice, 1 = the one-frame kernel)
for kind, mf in [(int(a), int(m)) for a in (sys.argv[1:] or ["0", "5", "6", "7"]) for m in os.environ.get("MF", "0").split(",")]:
    assert lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_SHAPE, kind) == 0
    assert lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_FRAMES, mf) == 0
    for _ in range(60): kernels.warp_backward(src, inv, grid, (Hh, W), "bilinear", torch.uint8, zero_origin=False, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = int(os.environ.get("N", "200"))
    probe = kernels.ClockProbe(0.6 * n * 0.45 * frames / 32)      # ends inside the timed launches
    e0.record()
    for _ in range(n): kernels.warp_backward(src, inv, grid, (Hh, W), "bilinear", torch.uint8, zero_origin=False, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    by = frames * (Hh * W * 3 + oh * ow * 3)
    same = "" if ref is None else (" identical to first kind: %s" % bool(torch.equal(ref, out)))
    if ref is None: ref = out.clone()
    import hashlib
    digest = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12] if os.environ.get("SHA") else ""      # SHA=1: digest of the output
    plan = kernels.warp_plan((frames, Hh, W, 3), torch.uint8, inv, grid, (Hh, W), "bilinear", torch.uint8)
    watts = ""
    if os.environ.get("POWER"):          # POWER=1: 3 more seconds of back-to-back launches with rocm-smi sampled meanwhile
        import threading, time, glob
        samples, stop = [], threading.Event()
        node = kernels.power_node()           # amdgpu hwmon file of this device (no child process: this program holds the GPU)
        def sampler():
            while not stop.is_set() and node:
                try: samples.append(int(open(node).read()) * 1e-6)
                except (OSError, ValueError): pass
                time.sleep(0.02)
        th = threading.Thread(target=sampler); th.start()
        t_end = time.time() + 3.0
        p2 = kernels.ClockProbe(1500.0)
        while time.time() < t_end:
            for _ in range(200): kernels.warp_backward(src, inv, grid, (Hh, W), "bilinear", torch.uint8, zero_origin=False, out=out)
            torch.cuda.current_stream().synchronize()      # not the device: that would wait for the clock probe
        stop.set(); th.join()
        tail = sorted(samples[len(samples) // 2:])
        watts = "  power %.0f W (median of %d samples, max %.0f) sclk(1.5 s) %.0f MHz" % (tail[len(tail) // 2] if tail else 0, len(tail), max(samples or [0]), p2.mhz())
    print(plan, end="  ")
    print("kind %2d mf %3d  %.4f ms per %d frames = %.2f us/frame  %.0f GB/s = %.3f of 8 TB/s  sclk %.0f MHz%s" % (kind, mf, ms, frames, ms * 1e3 / frames, by / ms / 1e6, by / ms / 1e6 / 8000, probe.mhz(), same) + watts + (" sha1 " + digest if digest else ""), flush=True)
