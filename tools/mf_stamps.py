"""Lab probe (needs a library built with -DRWH_LAB_STAMPS: tools/build_variant.sh stamps -DRWH_LAB_STAMPS): cycles per phase of the
multi-frame warp kernel's interior waves -- taps + blend | wait for the prefetched chunks | expand into LDS | stores + next loads.
   RWH_LIB=tools/labbuild/librwh_stamps.so MF=4 python tools/mf_stamps.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_with_homography_amd import _lib, kernels
from ransac_with_homography_amd import homography as hg
_lib.LIB_PATH = os.path.abspath(os.environ["RWH_LIB"])
H_S = np.array([[1.02, 0.01, 5.0], [0.015, 0.98, 7.0], [1e-5, 2e-5, 1.0]])
frames, W, Hh = 32, 3840, 2160
dev = _lib.require_gpu()
src = torch.randint(0, 256, (frames, Hh, W, 3), dtype=torch.uint8).to(dev)
mx, my, ow, oh = hg._bounds(Hh, W, H_S, 0)
grid = kernels.Grid(mx, mx + ow - 1, ow, my, my + oh - 1, oh)
inv = np.linalg.inv(H_S)
out = torch.empty((frames, oh, ow, 3), dtype=torch.uint8, device=dev)
lib = _lib.load()
for mf in [int(m) for m in os.environ.get("MF", "4,8").split(",")]:
    assert lib.rwh_lab_tune(_lib.RWH_TUNE_WARP_FRAMES, mf) == 0
    for _ in range(30): kernels.warp_backward(src, inv, grid, (Hh, W), "bilinear", torch.uint8, zero_origin=False, out=out)
    torch.cuda.synchronize()
    st = out.view(-1)[: 8192 * 4 * 4 * 8].view(torch.int64).cpu().numpy().reshape(-1, 4)
    st = st[(st[:, 0] > 0) & (st[:, 0] < 1e8)]                    # interior waves of the first 8192 blocks
    per = st / float(mf)
    print("mf %d: %d waves; cycles per frame: taps+blend %.0f | wait %.0f | expand %.0f | stores+issue %.0f | total %.0f   (medians; p90 total %.0f)" %
          (mf, len(st), *np.median(per, axis=0), np.median(per.sum(1)), np.quantile(per.sum(1), 0.9)), flush=True)
