"""Developer probe: what the chip does over SECONDS of sustained warp launches -- kernel time per batch of 100 launches (HIP
events), the shader clock in consecutive 100 ms windows (rwh_lab_clock_probe kernels queued back to back on a side stream),
rocm-smi power / sclk samples.   python tools/sustain_probe.py [seconds] [interp: bilinear|nn]"""
import os, sys, time, subprocess, threading, re, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_with_homography_amd import _lib, kernels
from ransac_with_homography_amd import homography as hg
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
interp = sys.argv[2] if len(sys.argv) > 2 else "bilinear"
H_S = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
dev = _lib.require_gpu(); lib = _lib.load()
W, Hh, frames = 3840, 2160, 32
src = torch.randint(0, 256, (frames, Hh, W, 3), dtype=torch.uint8, device=dev)
mx, my, ow, oh = hg._bounds(Hh, W, H_S, 0)
grid = kernels.Grid(mx, mx + ow - 1, ow, my, my + oh - 1, oh); inv = np.linalg.inv(H_S)
out = torch.empty((frames, oh, ow, 3), dtype=torch.uint8, device=dev)
step = lambda: kernels.warp_backward(src, inv, grid, (Hh, W), interp, torch.uint8, zero_origin=False, out=out)
for _ in range(20): step()
torch.cuda.synchronize()
time.sleep(1.0)                                   # start from an idle chip
nprobe = int(secs * 10) + 5
pout = torch.zeros((nprobe, 2), dtype=torch.int64, device=dev)
side = torch.cuda.Stream()
samples, stop = [], threading.Event()
def sampler():
    while not stop.is_set():
        o = subprocess.run("rocm-smi --showpower --showclocks", shell=True, capture_output=True, text=True).stdout
        p = re.search(r"Power \(W\): ([0-9.]+)", o); c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", o)
        samples.append((time.perf_counter(), float(p.group(1)) if p else -1, int(c.group(1)) if c else -1))
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter()
for i in range(nprobe):
    lib.rwh_lab_clock_probe(ctypes.c_void_p(pout[i].data_ptr()), 100.0, ctypes.c_void_p(side.cuda_stream))
evs = []
while time.perf_counter() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): step()
    e1.record(); evs.append((time.perf_counter() - t0, e0, e1))
    if len(evs) > 4: evs[-4][2].synchronize()      # stay a few batches ahead, not unboundedly
torch.cuda.synchronize(); t_load_end = time.perf_counter() - t0
stop.set(); th.join()
side.synchronize()
p = pout.cpu().numpy()
print("load ran %.2f s; clock per 100 ms window (MHz):" % t_load_end)
print(" ".join("%.0f" % (100.0 * a / b) if b else "-" for a, b in p))
print("kernel ms per launch, per batch of 100 (time s: ms):")
print(" ".join("%.1f:%.4f" % (t, a.elapsed_time(b) / 100) for t, a, b in evs[::max(1, len(evs) // 40)]))
print("rocm-smi (time s: W, sclk MHz):")
print(" ".join("%.1f:%.0f,%d" % (t - t0, w, c) for t, w, c in samples))
