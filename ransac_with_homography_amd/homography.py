"""MI355X-backed mirror of the reference's `homography.py` call surface.

Same names, argument meaning, defaults, return types and error behaviour as
choice17/ransac_with_homography `homography.py`; the per-pixel work (grid ->
inverse homography -> divide -> mask -> gather -> lerp, homography.py:108-209)
runs in the hand-written HIP kernel K3 behind `rwh_warp_backward` (include/rwh.h).

numpy arrays in -> numpy arrays out (uploaded / downloaded around the kernel);
torch-ROCm tensors in -> torch-ROCm tensors out (no host round trip: bilinear
results are float32 instead of the reference's float64, nn keeps the dtype).

What stays on the host, as in SURVEY.md 8(a) row a3: the O(1) 3x3 / 8x8 solves
(`calcHomographyLinear`, `inv(H)`, output bounding box) in numpy, exactly the
reference's arithmetic.  There is no CPU implementation of the warp here: without
librwh_hip.so and a GPU the warps raise `RwhUnavailable`.

Documented divergences from the reference:
  * numpy arrays in: a coordinate exactly on the last column / row, or a scan-mode `res` beyond the image, raises the
    reference's IndexError (`rwh_warp_index_check` tells where the reference would index past the image; fixture g15).
    Tensors in (the fast kernels): the +1 bilinear tap is clamped instead (its weight is 0 there) and scan-mode bounds
    larger than the source are clipped to the source;
  * with the fast kernels (torch tensors in, or EXACT = False) the bilinear blend is float32 on
    float64-derived weights (<= 1e-4 relative to the reference's float64 blend, typically 3e-7) and
    uint8 results can differ by 1 LSB where the float64 value sits within ~1e-5 of an integer; with the
    exact kernel (numpy arrays in, the default) results are bit-identical to the reference's;
  * `cylindericlMap` / `cylindricalWarp` / `cylindericalTransform` (dead code in
    the reference, needs OpenCV) are not provided beyond an import-compatible stub.
"""
import os

import numpy as np

from . import _lib, _xfer, kernels

# Which warp kernels serve this module:
#   EXACT = None (default)  numpy arrays in  -> the float64 "exact" kernel: results bit-identical to the reference's
#                                               float64 arrays and truncated uint8 images (a host round trip dominates
#                                               the call anyway);
#                           torch tensors in -> the fast float32-blend kernels (<= 1e-4 relative, uint8 within 1 LSB);
#   EXACT = True / False    force one or the other (environment: RWH_EXACT=1 / RWH_EXACT=0).
EXACT = {"1": True, "0": False}.get(os.environ.get("RWH_EXACT", ""), None)

__all__ = [
    "calc_corresp", "calc_correspLinear", "calc_correspCollective", "calc_correspLinearCollective",
    "calcHomography", "calcHomographyLinear", "calcH", "nearestNeighbor", "bilinear", "convertfunc",
    "wrapPerspective", "wrapPerspectiveScan", "perspectiveTransform", "transformImage", "transformImageH",
    "BLENDDIR", "addAlpha", "stitchPanorama", "cylindericlMap",
]


# ------------------------------------------------------------------ design matrices (host, O(N)) ----
def _pair_rows(u, v, sign):
    """Shared builder.  sign=-1: DLT rows [-x,-y,-1,0,0,0, x*x', y*x', x'] (homography.py:4-14, 30-46);
    sign=+1: linear rows [x,y,1,0,0,0,-x*x',-y*x'] with b = (x', y') (homography.py:16-28, 48-69).
    Products are formed in the input dtype, storage is float32, as in the reference."""
    u = np.asarray(u)
    v = np.asarray(v)
    n = u.shape[0]
    x, y, xp, yp = u[:, 0], u[:, 1], v[:, 0], v[:, 1]
    if sign < 0:
        a = np.zeros((n, 18), dtype=np.float32)
        a[:, 0] = -x; a[:, 1] = -y; a[:, 2] = -1
        a[:, 6] = x * xp; a[:, 7] = y * xp; a[:, 8] = xp
        a[:, 12] = -x; a[:, 13] = -y; a[:, 14] = -1
        a[:, 15] = x * yp; a[:, 16] = y * yp; a[:, 17] = yp
        return a.reshape(2 * n, 9)
    a = np.zeros((n, 16), dtype=np.float32)
    b = np.zeros((n, 2), dtype=np.float32)
    a[:, 0] = x; a[:, 1] = y; a[:, 2] = 1
    a[:, 6] = -x * xp; a[:, 7] = -y * xp
    a[:, 11] = x; a[:, 12] = y; a[:, 13] = 1
    a[:, 14] = -x * yp; a[:, 15] = -y * yp
    b[:, 0] = xp; b[:, 1] = yp
    return a.reshape(2 * n, 8), b.reshape(2 * n, 1)


def _four_rows(u, v):
    """The 4-point builders index rows 0..3 of both arrays by hand (homography.py:6-13, 18-27): with fewer rows they fail at
    the first index that is not there."""
    u, v = np.asarray(u), np.asarray(v)
    for a in (u, v):
        if a.shape[0] < 4:
            raise IndexError("index %d is out of bounds for axis 0 with size %d" % (a.shape[0], a.shape[0]))
    return u[:4], v[:4]


def calc_corresp(u, v):
    """8 x 9 DLT matrix of 4 pairs, float32 (homography.py:4-14)."""
    return _pair_rows(*_four_rows(u, v), -1)


def calc_correspLinear(u, v):
    """(8 x 8 A, 8 x 1 b), float32 (homography.py:16-28)."""
    return _pair_rows(*_four_rows(u, v), +1)


def calc_correspCollective(u, v):
    """2N x 9 DLT matrix (homography.py:30-46)."""
    return _pair_rows(u, v, -1)


def calc_correspLinearCollective(u, v):
    """(2N x 8 A, 2N x 1 b) (homography.py:48-69)."""
    return _pair_rows(u, v, +1)


# ------------------------------------------------------------------------------------- solvers ----
def calcHomography(u, v, collective=False):
    """DLT homography, float32 3x3 with h33 == 1 (homography.py:71-88): the reference's own arithmetic on the host --
    float32 DLT matrix -> numpy.linalg.svd -> last right-singular vector / its 9th element.  One 8 x 9 SVD costs less than
    the upload + launch + download a GPU solve of a single sample would, and it IS the reference's solver: samples with
    a repeated index return LAPACK's null vector like the reference does.  (The batched GPU generator K1 serves
    `RANSAC.run`'s thousands of samples per call, with this solver as its arbiter: ransac._settle_on_host.)"""
    u = np.asarray(u)
    v = np.asarray(v)
    mat = calc_correspCollective(u, v) if collective else calc_corresp(u, v)
    _, _, vt = np.linalg.svd(mat)
    h = vt[-1].reshape(3, 3)
    return h / h.item(8)


def calcHomographyLinear(u, v, collective=False):
    """Normal-equation homography with h33 == 1, float64 3x3 (homography.py:90-105).
    O(1) host work (called once per scanner apply / RANSAC run)."""
    A, b = calc_correspLinearCollective(u, v) if collective else calc_correspLinear(u, v)
    h = np.linalg.inv(A.T @ A) @ (A.T @ b)
    return np.array([[h.item(0), h.item(1), h.item(2)],
                     [h.item(3), h.item(4), h.item(5)],
                     [h.item(6), h.item(7), 1]])


calcH = calcHomographyLinear  # name used by BASELINE.json's north_star


# ------------------------------------------------------------------------------- warp plumbing ----
def _is_tensor(x):
    return type(x).__module__.startswith("torch")


def _blank_origin(img):
    """Mirror homography.py:112-116 / 126-130 on the caller's HOST array (the kernel does the
    same to the device copy): channels 0..2, and 3 when there are exactly 4."""
    img[0, 0, 0] = 0
    img[0, 0, 1] = 0
    img[0, 0, 2] = 0
    if img.shape[2] == 4:
        img[0, 0, 3] = 0


def _to_device(img, host=None):
    """-> (GPU tensor [H,W,C] uint8|float32, was_numpy, result dtype for nn)."""
    import torch
    dev = _lib.require_gpu()
    if _is_tensor(img):
        t = img
        if not t.is_cuda:
            t = t.to(dev)
        if t.dtype not in (torch.uint8, torch.float32):
            t = t.to(torch.float32)
        return t.contiguous(), False, None
    src, np_dtype = host if host is not None else _host_src(img)      # (`host`: what _host_src already returned for this image)
    return _xfer.to_device(src, dev), True, np_dtype


def _host_src(img):
    """numpy image -> (the uint8 / float32 array the kernels read, the caller's dtype)."""
    a = np.asarray(img)
    if a.ndim != 3:
        raise ValueError("not enough values to unpack (expected 3, got %d)" % a.ndim)  # img.shape unpack
    if a.shape[2] not in (3, 4):
        raise IndexError("index out of bounds: the warp supports 3 or 4 channels")
    return (a if a.dtype in (np.uint8, np.float32) else a.astype(np.float32)), a.dtype


def _warp(img, H, grid, bound_hw, convert, u8_out):
    """Common body of wrapPerspective / wrapPerspectiveScan from `invH = inv(H)` on."""
    import torch
    if convert not in kernels.INTERP:
        raise KeyError(convert)  # convertfunc[convert], homography.py:179 / 208
    inv_h = np.linalg.inv(np.asarray(H, dtype=np.float64))  # homography.py:172 / 203 (raises LinAlgError)
    host = None
    if not _is_tensor(img) and PIPELINE_MIN_BYTES is not None:
        # a large host array through the exact kernels: upload, kernel by output-row tiles and download overlapped (_warp_pipelined)
        a, np_dtype = host = _host_src(img)
        exact = True if EXACT is None else bool(EXACT)
        t_dtype = torch.uint8 if a.dtype == np.uint8 else torch.float32
        out_dtype = t_dtype if convert == "nn" else (torch.uint8 if u8_out else torch.float64 if exact else torch.float32)
        out_bytes = grid.out_h * grid.out_w * a.shape[2] * torch.empty(0, dtype=out_dtype).element_size()
        # (the result of a pipelined call is ONE page-locked block: capped like _xfer.to_host's, beyond it the plain path)
        if a.shape[0] >= 3 and a.shape[1] >= 3 and a.nbytes + out_bytes >= PIPELINE_MIN_BYTES * PIPELINE_WARP_FACTOR and grid.out_h >= 64 \
                and out_bytes <= _xfer.PINNED_RESULT_MAX:
            dev = _lib.require_gpu()
            a = np.ascontiguousarray(a)
            flag = kernels.warp_index_check(a.shape[:2], inv_h, grid, bound_hw, convert, dev)
            res = _warp_pipelined(a, inv_h, grid, bound_hw, convert, out_dtype, exact, dev)
            if res is not None:       # (None: page-locked memory for the result could not be had -- the plain path below)
                _blank_origin(img)
                bits = int(flag.item())
                if bits:
                    kernels.raise_like_reference(bits, a.shape[:2])
                if convert == "nn":
                    return res if res.dtype == np_dtype else res.astype(np_dtype)
                return res if (u8_out or res.dtype == np.float64) else res.astype(np.float64)
            host = (a, np_dtype)
    src, was_numpy, np_dtype = _to_device(img, host)
    true_hw = (int(src.shape[0]), int(src.shape[1]))
    if true_hw[0] < 3 or true_hw[1] < 3:      # the kernels want 3 x 3 texels at least: zero rows / columns beyond the bounds, which
        pad = torch.zeros((max(true_hw[0], 3), max(true_hw[1], 3), src.shape[2]), dtype=src.dtype, device=src.device)   # stay (h, w)
        pad[:true_hw[0], :true_hw[1]] = src
        src = pad
    exact = was_numpy if EXACT is None else bool(EXACT)
    if convert == "nn":
        out_dtype = src.dtype
    elif exact:
        out_dtype = torch.uint8 if u8_out else torch.float64
    else:
        out_dtype = torch.uint8 if u8_out else torch.float32
    # numpy arrays in (the reference's own callers): where the reference's interpolator would index past the image -- a coordinate
    # exactly on the last column / row, a scan `res` beyond the image, a NaN coordinate -- raise its IndexError (rwh.h)
    flag = kernels.warp_index_check(true_hw, inv_h, grid, bound_hw, convert, src.device) if was_numpy else None
    out = kernels.warp_backward(src, inv_h, grid, bound_hw, convert, out_dtype, zero_origin=True, exact=exact)
    if not was_numpy:
        return out
    _blank_origin(img)
    res = _xfer.to_host(out)
    bits = int(flag.item())
    if bits:
        kernels.raise_like_reference(bits, true_hw)
    if convert == "nn":
        return res if res.dtype == np_dtype else res.astype(np_dtype)
    return res if (u8_out or res.dtype == np.float64) else res.astype(np.float64)  # the reference's bilinear yields float64


def _bounds(h, w, H, boundary):
    """Output bounding box, homography.py:143-163 (host numpy, 4 points)."""
    bnd = np.array([[0, w - 1, w - 1, 0],
                    [0, 0, h - 1, h - 1],
                    [1., 1, 1, 1]])
    bnd_n = np.asarray(H) @ bnd
    bnd_n /= bnd_n[-1, :]
    max_x = int(np.max(bnd_n[0, :])); min_x = int(np.min(bnd_n[0, :]))
    max_y = int(np.max(bnd_n[1, :])); min_y = int(np.min(bnd_n[1, :]))
    if boundary:
        min_x = max(min_x, 0)
        min_y = max(min_y, 0)
    return min_x, min_y, max_x - min_x + 1, max_y - min_y + 1


def _wrap_perspective(img, H, convert, boundary, u8_out):
    h, w, _ = img.shape
    min_x, min_y, max_w, max_h = _bounds(h, w, H, boundary)
    if max_w <= 0 or max_h <= 0:
        raise ValueError("Number of samples, %d, must be non-negative." % min(max_w, max_h))  # np.linspace
    grid = kernels.Grid(min_x, min_x + max_w - 1, max_w, min_y, min_y + max_h - 1, max_h)
    return _warp(img, H, grid, (h, w), convert, u8_out), min_x, min_y


def wrapPerspective(img, H, convert='nn', boundary=0, crop=True):
    """Auto-bounds backward warp (homography.py:142-184).  Returns (img_n, min_x, min_y);
    img_n is float64 for 'bilinear' and keeps the image dtype for 'nn'."""
    if not crop:
        raise NotImplementedError("crop=False is dead, shape-inconsistent code in the reference (homography.py:180-183)")
    return _wrap_perspective(img, H, convert, boundary, False)


def wrapPerspectiveScan(img, H, res, convert='nn'):
    """Fixed-resolution backward warp (homography.py:186-209): grid linspace(0,w,w) x
    linspace(0,h,h); the bounds test uses `res` (clipped to the source here)."""
    h, w = res
    grid = kernels.Grid(0, w, w, 0, h, h)
    return _warp(img, H, grid, (h, w), convert, False), 0, 0


perspectiveTransform = wrapPerspective  # name used by BASELINE.json's north_star


def _sample(z_t, img, h, w, mh, mw, convert):
    """Common body of the two interpolators on caller-computed coordinates: `rwh_sample_points` (the reference's float64
    arithmetic, bit-identical results).  Like the reference it blanks texel (0,0) of the caller's image and, for
    'bilinear', writes 0 into the masked columns of the caller's z_t (homography.py:131-132: `z_t = z_t.T` is a view)."""
    import torch
    dev = _lib.require_gpu()
    chn = img.shape[2]
    if _is_tensor(z_t):
        zx, zy = z_t[0].to(dev, torch.float64).contiguous(), z_t[1].to(dev, torch.float64).contiguous()
    else:
        z_t = np.asarray(z_t)
        zx = torch.from_numpy(np.ascontiguousarray(z_t[0], dtype=np.float64)).to(dev)
        zy = torch.from_numpy(np.ascontiguousarray(z_t[1], dtype=np.float64)).to(dev)
    src, was_numpy, np_dtype = _to_device(img)
    out = kernels.sample_points(src, zx, zy, (h, w), convert, zero_origin=True).reshape(mh, mw, chn)
    if convert == 'bilinear' and not _is_tensor(z_t) and z_t.dtype.kind == 'f':
        mask = (z_t[0] > w - 1) | (z_t[0] < 0) | (z_t[1] > h - 1) | (z_t[1] < 0)
        z_t[0:2, mask] = 0
    if not was_numpy:
        return out
    _blank_origin(img)
    res = out.cpu().numpy()
    # the reference indexes the image with every coordinate its mask lets through: raise where it would (see _warp, rwh.h)
    ih_, iw_ = int(img.shape[0]), int(img.shape[1])
    if convert == 'nn':
        xi, yi = (zx + 0.5).to(torch.int64), (zy + 0.5).to(torch.int64)
        u = (xi >= 0) & (xi <= w - 1) & (yi >= 0) & (yi <= h - 1) & ~torch.isnan(zx) & ~torch.isnan(zy)
        bits = (1 if bool((u & (xi > iw_ - 1)).any()) else 0) | (2 if bool((u & (yi > ih_ - 1)).any()) else 0)
    else:
        u = ~((zx > w - 1) | (zx < 0) | (zy > h - 1) | (zy < 0))
        nan = u & (torch.isnan(zx) | torch.isnan(zy))
        fin = u & ~nan
        bits = (4 if bool(nan.any()) else 0) | (1 if bool((fin & (zx.floor() + 1 > iw_ - 1)).any()) else 0) | \
               (2 if bool((fin & (zy.floor() + 1 > ih_ - 1)).any()) else 0)
    if bits:
        kernels.raise_like_reference(bits, (ih_, iw_))
    return res if (convert == 'bilinear' or res.dtype == np_dtype) else res.astype(np_dtype)


def nearestNeighbor(z_t, img, h, w, mh, mw):
    """homography.py:108-121 on caller-computed coordinates z_t (3 x N, dehomogenised): (z_t + 0.5) truncated to int32,
    mask on the integers, gather; returns mh x mw x C in the image's dtype."""
    return _sample(z_t, img, h, w, mh, mw, 'nn')


def bilinear(z_t, img, h, w, mh, mw):
    """homography.py:123-138 on caller-computed coordinates: mask on the float64 coordinates, truncation, float64 lerps;
    returns mh x mw x C float64."""
    return _sample(z_t, img, h, w, mh, mw, 'bilinear')


convertfunc = {'nn': nearestNeighbor, 'bilinear': bilinear}


def transformImage(img, u, v, box=None, method='bilinear'):
    """Homography from 4 corner pairs + warp + uint8 + crop (homography.py:211-228).
    u, v: 3 x 4 homogeneous corners; `box=(h,w)` selects the scanner's fixed-resolution mode.
    The uint8 truncation is fused into the kernel (dst_dtype U8)."""
    u = np.asarray(u)
    v = np.asarray(v)
    H = calcHomographyLinear(u.T[:, :2], v.T[:, :2])
    if box is None:
        imgn, mx, my = _wrap_perspective(img, H, method, 0, True)
    else:
        h, w = box
        imgn = _warp(img, H, kernels.Grid(0, w, w, 0, h, h), (h, w), method, True)
        mx = my = 0
    if not _is_tensor(imgn) and imgn.dtype != np.uint8:
        imgn = imgn.astype(np.uint8)
    sx = int(v[0, 0] - mx); sy = int(v[1, 0] - my)
    ex = int(v[0, 2] - mx); ey = int(v[1, 2] - my)
    return imgn[sy:ey + 1, sx:ex + 1, :]


def transformImageH(img, H, method='bilinear'):
    """Warp by a given H (homography.py:230-242): uint8 for 3-channel results, 4-channel
    results stay floating point.  Returns (img, mx, my)."""
    if img.shape[2] == 3:
        imgn, mx, my = _wrap_perspective(img, H, method, 0, True)
        if not _is_tensor(imgn) and imgn.dtype != np.uint8:
            imgn = imgn.astype(np.uint8)
        return imgn, mx, my
    return _wrap_perspective(img, H, method, 0, False)


# ------------------------------------------------------------------------- alpha + compositor ----
class BLENDDIR(object):
    LEFT = 0
    RIGHT = 1
    TOP = 2
    DOWN = 3


def addAlpha(img, method="Rate", rate=0.2, direction=BLENDDIR.LEFT, alphaOnly=False):
    """Append (or return) an alpha plane, float32 (homography.py:250-286).  'Rate' is the only
    mode the reference finishes; its 'Gradient' branch (half-implemented, prints
    'not implement yet') is reproduced for the LEFT/RIGHT ramps it defines."""
    h, w, c = img.shape
    rate += 1e-10

    def ramp():
        if direction == BLENDDIR.RIGHT and alphaOnly:
            x = np.linspace(w - 1, 0, w); y = np.linspace(h - 1, 0, h)
        elif direction == BLENDDIR.LEFT:
            x = np.linspace(0, w - 1, w); y = np.linspace(0, h - 1, h)
        else:
            raise UnboundLocalError("local variable 'br' referenced before assignment")
        xx, yy = np.meshgrid(x, y)
        return (xx + yy) / (w + h) * 0.5

    if not alphaOnly:
        imgn = np.zeros((h, w, c + 1), dtype=np.float32)
        imgn[:, :, :c] = img
        if method == 'Rate':
            print(rate)
            imgn[:, :, c] = rate
        elif method == 'Gradient':
            imgn[:, :, c] = ramp()
            print('not implement yet')
        return imgn
    alpha = np.zeros((h, w, 1), dtype=np.float32)
    if method == 'Rate':
        print(rate)
        alpha[:, :, 0] = rate
    elif method == 'Gradient':
        alpha[:, :, 0] = ramp()
        print('not implement yet')
    return alpha


def _stitch_geometry(wt, ht, wq, hq, mx, my):
    """Paste rectangles and canvas size, homography.py:303-321."""
    tsx = 0; tsy = 0; tex = wt - 1; tey = ht - 1
    qsx = 0; qsy = 0; qex = wq - 1; qey = hq - 1
    if mx < 0 and my < 0:
        qsx = -mx; qsy = -my; qex = -mx + wq - 1; qey = -my + hq - 1
    elif mx < 0:
        tsy = my; tey = my + ht - 1
        qsx = -mx; qex = -mx + wq - 1
    elif my < 0:
        tsx = mx; tex = mx + wt - 1
        qsy = -my; qey = -my + hq - 1
    else:
        tsx = mx; tsy = my; tex = mx + wt - 1; tey = my + ht - 1
    return (tsx, tsy, tex, tey), (qsx, qsy, qex, qey), (max(tex + 1, qex + 1), max(tey + 1, qey + 1))


def _as_uint8_image(img, what):
    """stitchPanorama's kernels take uint8 RGB -- what cv2.imread hands the reference's own pipeline (ransac.py:236-243).  An image
    of another dtype whose values ARE uint8 values (a uint8 photograph converted to float) gives the reference the same canvas as its
    uint8 form -- the warp interpolates the same numbers, numpy's assignment into the uint8 canvas truncates the same way -- and is
    converted; anything else (fractions, values outside 0..255) is refused loudly rather than composited on the host."""
    if _is_tensor(img):
        import torch
        if img.dtype == torch.uint8:
            return img
        u = img.to(torch.uint8)
        if bool((u.to(img.dtype) == img).all()):
            return u
    else:
        a = np.asarray(img)
        if a.dtype == np.uint8:
            return img
        with np.errstate(invalid="ignore"):
            u = a.astype(np.uint8)
            if np.array_equal(u.astype(a.dtype), a):
                return u
    raise NotImplementedError("stitchPanorama: %s is not uint8 and holds values that are not uint8 values; the MI355X compositor takes "
                              "uint8 RGB images (the reference composites such images in numpy, homography.py:296-338)" % what)


# host arrays: below this many bytes (inputs + result) the plain upload / kernel / download (None: never pipelined).  Measured: a stitch
# gains from ~90 MB on (two 4K frames 3.0 -> 2.8 ms, two 12 MP photographs 4.2 -> 3.8, two 8K frames 9.4 -> 6.4, config 4's 8192 x 5464 pair
# 11.4 -> 7.7); a single warp only from 8K frames on (4.7 -> 3.8 ms; 4K: 1.5 ms plain against 2.4 -- two dozen tiles cost more than they hide)
PIPELINE_MIN_BYTES = 64 << 20
PIPELINE_WARP_FACTOR = 2.5         # a single warp is pipelined from PIPELINE_MIN_BYTES x this on


def _rows_needed(ih, xs, y_lo, y_hi, src_h):
    """Source rows [0, n) a warp samples for output rows y_lo .. y_hi (grid coordinates) and columns xs[0] .. xs[1]: the four
    corners through inv(H) -- a projective map with W > 0 on a convex region takes its extremes at the vertices -- plus the
    +1 tap and a margin; all rows when the horizon touches the region."""
    X = np.array([xs[0], xs[1], xs[0], xs[1]], dtype=np.float64)
    Y = np.array([y_lo, y_lo, y_hi, y_hi], dtype=np.float64)
    W = ih[2, 0] * X + ih[2, 1] * Y + ih[2, 2]
    if not (W > 0).all():
        return src_h
    ymax = float(np.max((ih[1, 0] * X + ih[1, 1] * Y + ih[1, 2]) / W))
    return src_h if not np.isfinite(ymax) else int(min(src_h, max(0.0, np.floor(ymax) + 3)))


def _pipeline(dev, uploads, need, bounds, compose, result, host):
    """The three PCIe legs of a host-array call overlapped (full duplex).  `uploads`: {tag: (flat uint8 host array, flat uint8
    device tensor)}; `need`[i]: {tag: bytes of that array row tile i reads}; `bounds`: the tiles' row boundaries in `result`
    (a device tensor whose rows go down into `host`, a page-locked tensor of the same shape); compose(i, r0, r1) enqueues tile
    i's kernel on the current stream.  The arrays go up interleaved in the order the tiles need them, a tile is composed as
    soon as its bytes have landed, finished tiles go down in 8 MB pieces issued BETWEEN the upload chunks: this runtime
    overlaps the two directions only when their copies alternate in the queues (two 256 MB copies on two streams: 9.4 ms; the
    same bytes as alternating 8 MB pieces: 5.9 ms)."""
    import torch
    from collections import deque
    cur = torch.cuda.current_stream(dev)
    comp, down = _xfer.side_stream(dev, "compose"), _xfer.side_stream(dev, "download")
    comp.wait_stream(cur)
    down.wait_stream(cur)
    nt = len(bounds) - 1
    size = {t: u[0].size for t, u in uploads.items()}
    need = [{t: min(int(n.get(t, 0)), size[t]) for t in uploads} for n in need]
    for i in range(1, nt):                      # tiles launch in order: what an earlier tile needed has arrived
        for t in uploads:
            need[i][t] = max(need[i][t], need[i - 1][t])
    C = _xfer.CHUNK if max(size.values()) >= (64 << 20) else (2 << 20)      # finer pieces for single frames
    tasks, sent = [], {t: 0 for t in uploads}

    def push(tag, upto):
        while sent[tag] < upto:
            m = min(C, size[tag] - sent[tag])
            tasks.append((uploads[tag][0], uploads[tag][1], sent[tag], m, tag))
            sent[tag] += m
    for i in range(nt):
        for t in uploads:
            push(t, need[i][t])
    for t in uploads:
        push(t, size[t])
    have = {t: 0 for t in uploads}
    last_ev = {t: None for t in uploads}
    state = {"next": 0}
    pieces = deque()                            # (row0, row1, event of the tile's kernel) still to go down
    row_bytes = result[0].numel() * result.element_size() if result.shape[0] else 1

    def launch_ready():
        while state["next"] < nt and all(have[t] >= need[state["next"]][t] for t in uploads):
            i = state["next"]
            r0, r1 = int(bounds[i]), int(bounds[i + 1])
            with torch.cuda.stream(comp):
                for ev in last_ev.values():
                    if ev is not None:
                        comp.wait_event(ev)
                compose(i, r0, r1)
                done = torch.cuda.Event()
                done.record(comp)
            step = max(1, C // row_bytes)
            for a0 in range(r0, r1, step):
                pieces.append((a0, min(r1, a0 + step), done))
            state["next"] += 1

    def send_down(n):
        with torch.cuda.stream(down):
            while n > 0 and pieces:
                a0, a1, done = pieces.popleft()
                down.wait_event(done)
                host[a0:a1].copy_(result[a0:a1], non_blocking=True)
                n -= 1

    def got(tag, nbytes, ev):
        have[tag] = nbytes
        last_ev[tag] = ev
        launch_ready()
        send_down(2)                            # two pieces down per chunk up: the directions alternate in the DMA queues
    _xfer.upload_tasks(tasks, dev, on_chunk=got, join=False)
    launch_ready()
    send_down(1 << 30)
    down.synchronize()
    cur.wait_stream(comp)
    cur.wait_stream(down)
    return host.numpy()


def _stitch_pipelined(imgQ, imgT, inv_h, mx, my, wt, ht, tsx, tsy, qsx, qsy, fh, fw, mode, blendrate, dev):
    """stitchPanorama from host arrays through `_pipeline`: the canvas by row tiles (rwh_stitch_panorama_rows, the exact kernel), a
    tile as soon as the rows of imgT it samples and the rows of imgQ it covers have arrived.  -> the canvas as a host array
    (page-locked, like _xfer.to_host's)."""
    import torch
    tT = np.ascontiguousarray(imgT).reshape(-1).view(np.uint8)
    tQ = np.ascontiguousarray(imgQ).reshape(-1).view(np.uint8)
    h, w = int(imgT.shape[0]), int(imgT.shape[1])
    hq, wq = int(imgQ.shape[0]), int(imgQ.shape[1])
    canvas = torch.empty((fh, fw, 3), dtype=torch.uint8, device=dev)
    try:
        host = torch.empty((fh, fw, 3), dtype=torch.uint8, pin_memory=True)
    except RuntimeError:          # no page-locked memory for the canvas: the caller takes the plain path
        return None
    t_flat = torch.empty(tT.size, dtype=torch.uint8, device=dev)
    q_flat = torch.empty(tQ.size, dtype=torch.uint8, device=dev)
    t_dev, q_dev = t_flat.view(h, w, 3), q_flat.view(hq, wq, 3)
    nt = int(max(1, min(24, fh // 128)))
    bounds = np.linspace(0, fh, nt + 1).astype(np.int64)
    ih = np.asarray(inv_h, dtype=np.float64)
    need = []
    for i in range(nt):
        r0, r1 = int(bounds[i]), int(bounds[i + 1])
        lo, hi = max(r0, tsy), min(r1, tsy + ht)
        rows_t = _rows_needed(ih, (mx, mx + wt - 1), my + lo - tsy, my + hi - 1 - tsy, h) if lo < hi else 0
        rows_q = int(max(0, min(hq, r1 - qsy))) if (r1 > qsy and r0 < qsy + hq) else 0
        need.append({"t": rows_t * w * 3, "q": rows_q * wq * 3})
    need[0]["t"] = max(need[0]["t"], 3)         # the first tile blanks texel (0,0) of imgT: behind the chunk that carries it

    def compose(i, r0, r1):
        kernels.stitch_panorama_rows(t_dev, q_dev, inv_h, (mx, my), (wt, ht), (tsx, tsy), (qsx, qsy), canvas, (r0, r1), mode, blendrate,
                                     zero_origin=(i == 0))
    return _pipeline(dev, {"t": (tT, t_flat), "q": (tQ, q_flat)}, need, bounds, compose, canvas, host)


def _warp_pipelined(a, inv_h, grid, bound_hw, convert, out_dtype, exact, dev):
    """One warp from a host array through `_pipeline`: output row tiles (rwh_warp_backward's row_begin / row_end), a tile as soon
    as the source rows it samples have arrived.  a: contiguous uint8 / float32 H x W x C host array.  -> host array."""
    import torch
    flat = a.reshape(-1).view(np.uint8)
    h, w, c = (int(v) for v in a.shape)
    t_dtype = torch.uint8 if a.dtype == np.uint8 else torch.float32
    s_flat = torch.empty(flat.size, dtype=torch.uint8, device=dev)
    src = s_flat.view(t_dtype).view(h, w, c)
    oh, ow = grid.out_h, grid.out_w
    result = torch.empty((oh, ow, c), dtype=out_dtype, device=dev)
    try:
        host = torch.empty((oh, ow, c), dtype=out_dtype, pin_memory=True)
    except RuntimeError:          # no page-locked memory for a result of this size: the caller takes the plain path
        return None
    nt = int(max(1, min(24, oh // 64)))
    bounds = np.linspace(0, oh, nt + 1).astype(np.int64)
    ih = np.asarray(inv_h, dtype=np.float64)
    row_b = w * c * a.itemsize

    def gy(r):
        return grid.y_last if r == oh - 1 else grid.y0 + r * grid.step_y
    need = [{"s": _rows_needed(ih, (grid.x0, grid.x_last), gy(int(bounds[i])), gy(int(bounds[i + 1]) - 1), h) * row_b} for i in range(nt)]
    need[0]["s"] = max(need[0]["s"], c * a.itemsize)      # the first tile blanks texel (0,0)

    def compose(i, r0, r1):
        kernels.warp_backward(src, inv_h, grid, bound_hw, convert, out_dtype, zero_origin=(i == 0), rows=(r0, r1), out=result[r0:r1], exact=exact)
    return _pipeline(dev, {"s": (flat, s_flat)}, need, bounds, compose, result, host)


def stitchPanorama(imgQ, imgT, H, method='bilinear', blending=False, blendrate=0.2):
    """Warp imgT by H and composite it with imgQ on a common canvas (homography.py:288-338).  `method` is ignored
    exactly as in the reference (always bilinear).

    ONE fused kernel (`rwh_stitch_panorama`): alpha plane, warp, paste / alpha blend per canvas pixel; nothing intermediate (RGBA
    float32 image, float64 warp, float32 canvas) is materialised.  `blending` False / 'Rate' / 'Gradient' / any other truthy value
    (for which the reference's addAlpha leaves the alpha plane at 0: blend mode 3) on uint8 RGB images -- or images of another
    dtype that hold uint8 values, converted; round 4 removed the host compositor that used to take the rest.  numpy arrays in (or EXACT = True): the reference's float64 arithmetic, canvas bit-identical to the
    reference's; torch tensors in (or EXACT = False): the staged fast warp kernel with the compositor as its epilogue,
    canvas within 1 LSB."""
    import torch
    paste = not blending                                     # homography.py:298 / 322: `if blending:`
    if imgQ.shape[2] != 3 or imgT.shape[2] != 3:
        raise NotImplementedError("stitchPanorama: 3-channel images (the reference pastes a %d-channel warp into its canvas only when both "
                                  "images have 3 channels, homography.py:296-338)" % imgT.shape[2])
    tens = _is_tensor(imgQ) or _is_tensor(imgT)
    caller_imgT = imgT
    imgQ, imgT = _as_uint8_image(imgQ, "imgQ"), _as_uint8_image(imgT, "imgT")
    if blending == 'Rate':
        print(blendrate + 1e-10)   # addAlpha prints the rate it stores (homography.py:257)
    elif blending == 'Gradient':
        print('not implement yet')  # homography.py:266
    h, w, _ = imgT.shape
    mx, my, wt, ht = _bounds(h, w, H, 0)
    if wt <= 0 or ht <= 0:
        raise ValueError("Number of samples, %d, must be non-negative." % min(wt, ht))
    hq, wq, _ = imgQ.shape
    (tsx, tsy, tex, tey), (qsx, qsy, qex, qey), (fw, fh) = _stitch_geometry(wt, ht, wq, hq, mx, my)
    inv_h = np.linalg.inv(np.asarray(H, dtype=np.float64))
    dev = _lib.require_gpu()
    exact = (not tens) if EXACT is None else bool(EXACT)    # numpy in: the bit-identical float64 kernel; tensors in: the fast one
    # 'Gradient': the alpha ramp, exact kernel only; any other truthy value: addAlpha leaves the alpha plane at 0 (mode 3)
    mode = 0 if paste else 1 if blending == 'Rate' else 2 if blending == 'Gradient' else 3
    # large host arrays through the exact kernel: uploads, composition by row tiles and the download overlapped (_stitch_pipelined)
    pipelined = (not tens) and exact and PIPELINE_MIN_BYTES is not None and fh * fw * 3 <= _xfer.PINNED_RESULT_MAX and \
        (np.asarray(imgT).nbytes + np.asarray(imgQ).nbytes + fh * fw * 3) >= PIPELINE_MIN_BYTES
    if tens:
        t_dev = imgT.to(dev).contiguous()
        q_dev = imgQ.to(dev).contiguous()
        if blending:
            t_dev = t_dev.clone()      # addAlpha copies: the caller's imgT keeps its texel (0,0)
    elif not pipelined:
        t_dev = _xfer.to_device(imgT, dev)      # staged through page-locked buffers by several host threads (_xfer)
        q_dev = _xfer.to_device(imgQ, dev)
    # (numpy in: transformImageH's bilinear warp of imgT raises IndexError in the reference where it indexes past the image)
    flag = None if tens else kernels.warp_index_check((h, w), inv_h, kernels.Grid(mx, mx + wt - 1, wt, my, my + ht - 1, ht), (h, w), "bilinear", dev)
    if pipelined:
        res = _stitch_pipelined(imgQ, imgT, inv_h, mx, my, wt, ht, tsx, tsy, qsx, qsy, fh, fw, mode, blendrate, dev)
        if res is not None:
            if not blending:
                _blank_origin(caller_imgT)
            bits = int(flag.item())
            if bits:
                kernels.raise_like_reference(bits, (h, w))
            return res
        t_dev = _xfer.to_device(imgT, dev)      # (no page-locked memory for the canvas: upload / kernel / download)
        q_dev = _xfer.to_device(imgQ, dev)
    out = kernels.stitch_panorama(t_dev, q_dev, inv_h, (mx, my), (wt, ht), (tsx, tsy), (qsx, qsy), (fh, fw),
                                  mode, blendrate, zero_origin=True, fast=not exact)
    if tens:
        return out
    if not blending:
        _blank_origin(caller_imgT)     # transformImageH -> bilinear blanks the caller's texel (0,0) in the paste path
    res = _xfer.to_host(out)
    bits = int(flag.item())
    if bits:
        kernels.raise_like_reference(bits, (h, w))
    return res


def cylindericlMap(img, f=1600):
    """Import-compatibility stub: ransac.py:3 imports this name but no live code calls it
    (its only call sites are commented out, ransac.py:249-251).  Needs cv2.remap."""
    raise NotImplementedError("cylindrical warps are dead code in the reference and out of scope (SURVEY.md C11)")
