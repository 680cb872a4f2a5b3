"""Tensor-level wrappers over the C ABI (include/rwh.h): torch-ROCm tensors in,
torch-ROCm tensors out, work enqueued on torch's current HIP stream.

torch is plumbing here (device memory, streams); every computation below happens
inside librwh_hip.so.
"""
import ctypes
import threading
import math

import numpy as np
import torch

from . import _lib
from ._lib import (RWH_BILINEAR, RWH_F32, RWH_F64, RWH_LOSS, RWH_NEAREST, RWH_U8, RWH_WARP_EXACT, RWH_WARP_ZERO_ORIGIN, check)

_DTYPE = {torch.uint8: RWH_U8, torch.float32: RWH_F32, torch.float64: RWH_F64}
INTERP = {"nn": RWH_NEAREST, "bilinear": RWH_BILINEAR}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _dev_check(*tensors):
    for t in tensors:
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous()):
            raise ValueError("expected contiguous tensors on the GPU")


class Grid:
    """Output sampling grid with numpy.linspace semantics on both axes
    (reference homography.py:166-167 and 197-198)."""

    __slots__ = ("x0", "step_x", "x_last", "out_w", "y0", "step_y", "y_last", "out_h")

    def __init__(self, x_start, x_stop, out_w, y_start, y_stop, out_h):
        self.out_w, self.out_h = int(out_w), int(out_h)
        self.x0, self.x_last = float(x_start), float(x_stop)
        self.y0, self.y_last = float(y_start), float(y_stop)
        # numpy.linspace: step = (stop - start) / (num - 1)
        self.step_x = (self.x_last - self.x0) / (self.out_w - 1) if self.out_w > 1 else 0.0
        self.step_y = (self.y_last - self.y0) / (self.out_h - 1) if self.out_h > 1 else 0.0
        if self.out_w == 1:
            self.x_last = self.x0
        if self.out_h == 1:
            self.y_last = self.y0


def warp_backward(src, inv_h, grid, bound_hw, interp, out_dtype, zero_origin=True, rows=None, out=None, exact=False):
    """Launch K3.  `src`: [B,H,W,C] or [H,W,C] uint8/float32 GPU tensor.
    Returns a tensor [B,rows,out_w,C] (or without B) of `out_dtype` holding
    output rows `rows=(begin,end)` (default: all).  `inv_h`: inv(H) 3x3 for the whole batch, or [B,3,3] with one
    inverse per image (same output grid for all).  `exact=True` selects the float64 kernel that
    reproduces the reference's arithmetic bit for bit (out_dtype float64 / uint8 for bilinear)."""
    lib = _lib.load()
    _dev_check(src)
    squeeze = src.dim() == 3
    if squeeze:
        src = src.unsqueeze(0)
    B, H, W, C = src.shape
    if src.dtype not in _DTYPE or out_dtype not in _DTYPE:
        raise ValueError("unsupported image dtype")
    r0, r1 = (0, grid.out_h) if rows is None else rows
    if out is None:
        out = torch.empty((B, r1 - r0, grid.out_w, C), dtype=out_dtype, device=src.device)
    else:
        _dev_check(out)
        assert out.dtype == out_dtype and out.numel() == B * (r1 - r0) * grid.out_w * C
    ih = np.ascontiguousarray(inv_h, dtype=np.float64)
    n_h = 1 if ih.size == 9 else ih.size // 9          # one inverse for the batch, or [B,3,3]: one per image
    assert ih.size == 9 * n_h and n_h in (1, B), "inv_h: 3x3, or one 3x3 per image of the batch"
    ih = ih.reshape(9 * n_h)
    st = lib.rwh_warp_backward(
        _ptr(src), H, W, C, _DTYPE[src.dtype], src.stride(0) * src.element_size(), B,
        ih.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n_h,
        grid.x0, grid.step_x, grid.x_last, grid.y0, grid.step_y, grid.y_last,
        grid.out_h, grid.out_w, int(bound_hw[0]), int(bound_hw[1]), INTERP[interp],
        _ptr(out), _DTYPE[out_dtype], (r1 - r0) * grid.out_w * C * out.element_size(),
        r0, r1, (RWH_WARP_ZERO_ORIGIN if zero_origin else 0) | (RWH_WARP_EXACT if exact else 0), _lib.stream_ptr())
    check(st, "rwh_warp_backward")
    return out[0] if squeeze else out


def sample_points(img, xs, ys, bound_hw, interp, zero_origin=True):
    """Launch the interpolator on precomputed coordinates (rwh_sample_points): img [H,W,C] uint8|float32 GPU tensor, xs / ys
    [N] float64 GPU tensors -> [N, C] (image dtype for 'nn', float64 for 'bilinear')."""
    lib = _lib.load()
    _dev_check(img, xs, ys)
    assert xs.dtype == torch.float64 and ys.dtype == torch.float64 and xs.numel() == ys.numel()
    H, W, C = img.shape
    n = xs.numel()
    out_dtype = img.dtype if interp == "nn" else torch.float64
    out = torch.empty((n, C), dtype=out_dtype, device=img.device)
    check(lib.rwh_sample_points(_ptr(img), H, W, C, _DTYPE[img.dtype], _ptr(xs), _ptr(ys), n, int(bound_hw[0]), int(bound_hw[1]),
                                INTERP[interp], _ptr(out), _DTYPE[out_dtype], RWH_WARP_ZERO_ORIGIN if zero_origin else 0,
                                _lib.stream_ptr()), "rwh_sample_points")
    return out


def warp_index_check(src_hw, inv_h, grid, bound_hw, interp, device):
    """rwh_warp_index_check: enqueue the test "would the reference raise IndexError on this warp?" and return the device flag
    (int32 [1]; read it after the warp: 1 = an index past the last column, 2 = past the last row, 4 = a NaN coordinate)."""
    lib = _lib.load()
    ih = np.ascontiguousarray(inv_h, dtype=np.float64).reshape(9)
    flag = torch.empty(1, dtype=torch.int32, device=device)
    check(lib.rwh_warp_index_check(int(src_hw[0]), int(src_hw[1]), ih.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                   grid.x0, grid.step_x, grid.x_last, grid.y0, grid.step_y, grid.y_last, grid.out_h, grid.out_w,
                                   int(bound_hw[0]), int(bound_hw[1]), INTERP[interp], _ptr(flag), _lib.stream_ptr()), "rwh_warp_index_check")
    return flag


def raise_like_reference(bits, src_hw):
    """The IndexError numpy raises inside the reference's interpolators (homography.py:117-121, 133-135) for these flag bits."""
    if bits & 4:
        raise IndexError("index -2147483648 is out of bounds for axis 0 with size %d" % src_hw[0])
    if bits & 2:
        raise IndexError("index %d is out of bounds for axis 0 with size %d" % (src_hw[0], src_hw[0]))
    if bits & 1:
        raise IndexError("index %d is out of bounds for axis 1 with size %d" % (src_hw[1], src_hw[1]))


def warp_plan(src_shape, src_dtype, inv_h, grid, bound_hw, interp, out_dtype, rows=None, exact=False):
    """Name of the kernel `warp_backward` launches for this configuration (rwh_warp_plan: the library's own dispatch,
    nothing is launched and no GPU is needed).  src_shape: (B, H, W, C) or (H, W, C)."""
    lib = _lib.load()
    B, H, W, C = (1,) + tuple(src_shape) if len(src_shape) == 3 else tuple(src_shape)
    ih = np.ascontiguousarray(inv_h, dtype=np.float64)
    n_h = 1 if ih.size == 9 else ih.size // 9
    r0, r1 = (0, grid.out_h) if rows is None else rows
    buf = ctypes.create_string_buffer(128)
    check(lib.rwh_warp_plan(H, W, C, _DTYPE[src_dtype], B, ih.reshape(-1).ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n_h,
                            grid.x0, grid.step_x, grid.x_last, grid.y0, grid.step_y, grid.y_last, grid.out_h, grid.out_w,
                            int(bound_hw[0]), int(bound_hw[1]), INTERP[interp], _DTYPE[out_dtype], r0, r1,
                            RWH_WARP_EXACT if exact else 0, buf, 128), "rwh_warp_plan")
    return buf.value.decode()


def dlt4_batched(pts_a, pts_b, idx):
    """Launch K1.  pts_*: [M,2] float32, idx: [K,4] int32 -> (H [K,9] float32, flags [K] uint8)."""
    lib = _lib.load()
    _dev_check(pts_a, pts_b, idx)
    assert pts_a.dtype == torch.float32 and pts_b.dtype == torch.float32 and idx.dtype == torch.int32
    M, K = pts_a.shape[0], idx.shape[0]
    assert pts_a.shape == (M, 2) and pts_b.shape == (M, 2) and idx.shape == (K, 4)
    H = torch.empty((K, 9), dtype=torch.float32, device=idx.device)
    flags = torch.empty((K,), dtype=torch.uint8, device=idx.device)
    check(lib.rwh_dlt4_batched(_ptr(pts_a), _ptr(pts_b), M, _ptr(idx), K, _ptr(H), _ptr(flags), _lib.stream_ptr()),
          "rwh_dlt4_batched")
    return H, flags


def new_best(device):
    """Zeroed accumulator for rwh_score_count's packed argmax keys (2 x int64)."""
    return torch.zeros(2, dtype=torch.int64, device=device)


_scratch_best = {}


def scratch_best(device):
    """A 2 x int64 accumulator nobody reads (settle-step calls of K2 want counts and masks only): allocated once per device,
    never cleared -- saves the zero-fill launch of `new_best` per call."""
    key = (device.type, device.index)
    if key not in _scratch_best:
        _scratch_best[key] = torch.zeros(2, dtype=torch.int64, device=device)
    return _scratch_best[key]


def score_count(H, pts_a, pts_b, th, loss, need, best, hyp_base=0, want_masks=True, want_err=False, hinv=None):
    """Launch K2.  Returns (counts [K] int32, masks [K,ceil(M/64)] int64 or None, err [K,M] float32 or None);
    `best` (from new_best) is updated in place with atomic max.  hinv: [K, 9] float32 on the device, numpy.linalg.inv of every
    row of H as the reference would compute it ('backward' / 'reproj': rwh_score_count_inv), or None (the kernel's own)."""
    lib = _lib.load()
    _dev_check(H, pts_a, pts_b, best)
    K, M = H.shape[0], pts_a.shape[0]
    words = (M + 63) // 64
    counts = torch.empty((K,), dtype=torch.int32, device=H.device)
    masks = torch.empty((K, words), dtype=torch.int64, device=H.device) if want_masks else None
    err = torch.empty((K, M), dtype=torch.float32, device=H.device) if want_err else None
    if hinv is not None:
        assert hinv.dtype == torch.float32 and hinv.is_contiguous() and tuple(hinv.shape) == (K, 9) and hinv.device == H.device
    check(lib.rwh_score_count_inv(_ptr(H), _ptr(hinv) if hinv is not None else None, _ptr(pts_a), _ptr(pts_b), M, K, float(th),
                                  RWH_LOSS[loss], int(need), int(hyp_base), _ptr(counts), _ptr(masks) if want_masks else None,
                                  _ptr(best), _ptr(err) if want_err else None, _lib.stream_ptr()), "rwh_score_count_inv")
    return counts, masks, err


def score_interval(H, rows, flags, pts_a, pts_b, th, coord_scale, delta0, delta1):
    """rwh_score_interval: count intervals [lo, hi] of the listed rows of H ([K, 9] float32 on the GPU) under 'fwd' -- the pairs that
    are inliers for every / for some H within the perturbation budget (delta0, or delta1 for rows flagged RWH_HYP_ILLCOND in
    `flags`, uint8 [K] on the GPU or None).  rows: host int array.  -> (lo, hi) host int64 arrays."""
    lib = _lib.load()
    _dev_check(H, pts_a, pts_b)
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    n = int(rows.shape[0])
    if n == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    d_rows = torch.from_numpy(rows).to(H.device)
    out = torch.empty((2, n), dtype=torch.int32, device=H.device)
    check(lib.rwh_score_interval(_ptr(H), _ptr(d_rows), n, _ptr(flags) if flags is not None else None, _ptr(pts_a), _ptr(pts_b),
                                 int(pts_a.shape[0]), float(th), float(coord_scale), float(delta0), float(delta1), _ptr(out[0]), _ptr(out[1]),
                                 _lib.stream_ptr()), "rwh_score_interval")
    o = out.cpu().numpy().astype(np.int64)
    return o[0], o[1]


def host_inverses(H_rows):
    """numpy.linalg.inv of every float32 3 x 3 in `H_rows` ([n, 9] host array) exactly as the reference computes it inside its
    loop (ransac.py:74: float64 LAPACK, cast to float32); a singular matrix gives NaNs instead of numpy's LinAlgError."""
    H3 = np.ascontiguousarray(H_rows, dtype=np.float32).reshape(-1, 3, 3)
    with np.errstate(all="ignore"):
        try:
            return np.linalg.inv(H3).reshape(-1, 9)
        except np.linalg.LinAlgError:
            out = np.full((H3.shape[0], 9), np.nan, dtype=np.float32)
            for i in range(H3.shape[0]):
                try:
                    out[i] = np.linalg.inv(H3[i]).reshape(9)
                except np.linalg.LinAlgError:
                    pass
            return out


class SearchWorkspace:
    """Device buffers of one RANSAC search (K hypotheses over M correspondences), reusable across runs."""

    def __init__(self, k, m, device, want_masks=True):
        self.k, self.m = k, m
        self.H = torch.empty((k, 9), dtype=torch.float32, device=device)
        # counts (int32) and flags (uint8) share one buffer so that the host reads both back with ONE copy (`counts_flags`)
        self.cf = torch.empty((5 * k + 16,), dtype=torch.uint8, device=device)
        self.counts = self.cf[:4 * k].view(torch.int32)
        self.flags = self.cf[4 * k:5 * k]
        self.masks = torch.empty((k, (m + 63) // 64), dtype=torch.int64, device=device) if want_masks else None
        self.best = torch.zeros(2, dtype=torch.int64, device=device)

    def counts_flags(self):
        """-> (counts int32 [k], flags uint8 [k]) as host arrays, one device-to-host copy."""
        raw = self.cf.cpu().numpy()
        return raw[:4 * self.k].view(np.int32), raw[4 * self.k:5 * self.k]


def ransac_search(pts_a, pts_b, idx, th, loss, need, ws, hyp_base=0, reset_best=True):
    """K1 + K2 in one library call (rwh_ransac_search); results land in the workspace `ws`."""
    lib = _lib.load()
    _dev_check(pts_a, pts_b, idx)
    K, M = idx.shape[0], pts_a.shape[0]
    assert K <= ws.k and M == ws.m and idx.dtype == torch.int32
    check(lib.rwh_ransac_search(_ptr(pts_a), _ptr(pts_b), M, _ptr(idx), K, float(th), RWH_LOSS[loss], int(need), int(hyp_base),
                                _ptr(ws.H), _ptr(ws.flags), _ptr(ws.counts), _ptr(ws.masks) if ws.masks is not None else None,
                                _ptr(ws.best), 1 if reset_best else 0, _lib.stream_ptr()), "rwh_ransac_search")
    return ws


_pinned_run_ws = threading.local()      # per thread: two threads inside rwh_ransac_run (ctypes drops the GIL) must not share a host workspace


class RunWorkspace:
    """Buffers of one `rwh_ransac_run` call: a device workspace (fresh per run: `RANSAC.last_run` keeps views into it) and a
    page-locked host workspace (cached per (m, k): page-locking is the expensive part), laid out by rwh_ransac_run_layout."""
    D_H, D_COUNTS, D_FLAGS, D_MASKS, D_END, H_COUNTS, H_FLAGS, H_CNTSET, H_END = 3, 4, 5, 6, 11, 13, 14, 15, 19

    def __init__(self, m, k, device):
        lib = _lib.load()
        self.m, self.k, self.words = int(m), int(k), (int(m) + 63) // 64
        off = (ctypes.c_longlong * 27)()
        n = lib.rwh_ransac_run_layout(self.m, self.k, off, 27)
        if n != 27:
            check(n if n < 0 else _lib_invalid(), "rwh_ransac_run_layout")
        self.off = list(off)
        self.dev = torch.empty(self.off[self.D_END], dtype=torch.uint8, device=device)
        key = (self.m, self.k)
        cache = _pinned_run_ws.__dict__.setdefault("ws", {})
        if key not in cache:
            if len(cache) > 8:
                cache.clear()
            cache[key] = torch.empty(self.off[self.H_END], dtype=torch.uint8, pin_memory=True)
        self.host = cache[key]

    def _dview(self, which, nbytes, dtype):
        return self.dev[self.off[which]:self.off[which] + nbytes].view(dtype)

    @property
    def H(self):
        return self._dview(self.D_H, 36 * self.k, torch.float32).reshape(self.k, 9)

    @property
    def counts(self):
        return self._dview(self.D_COUNTS, 4 * self.k, torch.int32)

    @property
    def flags(self):
        return self._dview(self.D_FLAGS, self.k, torch.uint8)

    @property
    def masks(self):
        return self._dview(self.D_MASKS, 8 * self.words * self.k, torch.int64).reshape(self.k, self.words)

    def host_counts(self, settled=False):
        """K2's raw counts (or, settled=True, the counts after the settle step) as a host int32 array (a copy)."""
        o = self.off[self.H_CNTSET if settled else self.H_COUNTS]
        return self.host[o:o + 4 * self.k].numpy().view(np.int32).copy()

    def host_flags(self):
        o = self.off[self.H_FLAGS]
        return self.host[o:o + self.k].numpy().copy()


def _lib_invalid():
    return -1


def ransac_run(pts_a, pts_b, idx, th, loss, need, margin_cap, ws, dgesdd, threads, dgesv=None, hyp_base=0, want_keys=False):
    """rwh_ransac_run: upload + search + settle + accept rules in ONE native call.  pts_a / pts_b: float32 [M, 2] HOST arrays,
    idx: int32 [K, 4] host array (hyp_base: the global index of its first row when it is a slice of a sharded search).
    -> (winner | None, early, count, host_solved, rounds, flagged, mask_words uint64 [words]); want_keys: + (the slice's two
    packed keys as int64 [2], hypotheses given a count interval)."""
    lib = _lib.load()
    assert pts_a.dtype == np.float32 and pts_b.dtype == np.float32 and idx.dtype == np.int32
    assert pts_a.flags.c_contiguous and pts_b.flags.c_contiguous and idx.flags.c_contiguous
    assert pts_a.shape == (ws.m, 2) and pts_b.shape == (ws.m, 2) and idx.shape == (ws.k, 4)
    out = np.zeros(8, dtype=np.int32)
    keys = np.zeros(2, dtype=np.uint64)
    mask = np.zeros(ws.words, dtype=np.uint64)
    check(lib.rwh_ransac_run(pts_a.ctypes.data, pts_b.ctypes.data, ws.m, idx.ctypes.data, ws.k, float(th), RWH_LOSS[loss], int(need),
                             int(margin_cap), ctypes.c_void_p(dgesdd), ctypes.c_void_p(dgesv or 0), int(threads), _ptr(ws.dev),
                             ctypes.c_void_p(ws.host.data_ptr()), int(hyp_base),
                             out.ctypes.data, keys.ctypes.data, mask.ctypes.data, _lib.stream_ptr()), "rwh_ransac_run")
    winner = int(out[0])
    res = ((winner if winner >= 0 else None), bool(out[1]), int(out[2]), int(out[3]), int(out[4]), int(out[5]), mask)
    return res + (keys.view(np.int64), int(out[6])) if want_keys else res


class BatchWorkspace:
    """Device buffers of one batched search: P problems x K hypotheses (rwh_ransac_batched)."""

    def __init__(self, n_problems, k, m_max, device, want_masks=True):
        self.p, self.k, self.m_max = n_problems, k, m_max
        self.words = (m_max + 63) // 64
        n = n_problems * k
        self.idx = torch.empty((n_problems, k, 4), dtype=torch.int32, device=device)
        self.H = torch.empty((n_problems, k, 9), dtype=torch.float32, device=device)
        self.flags = torch.empty((n_problems, k), dtype=torch.uint8, device=device)
        self.counts = torch.empty((n_problems, k), dtype=torch.int32, device=device)
        self.masks = torch.empty((n_problems, k, self.words), dtype=torch.int64, device=device) if want_masks else None
        self.best = torch.zeros((n_problems, 2), dtype=torch.int64, device=device)


def ransac_batched(pts_a, pts_b, offsets, needs, th, loss, ws, seed=None, idx=None, problem_base=0, early_stop=False):
    """P independent RANSAC searches in one library call (rwh_ransac_batched, include/rwh.h).

    pts_a/pts_b: [total,2] float32 (the problems' correspondences concatenated), offsets: [P+1] int32,
    needs: [P] int32 -- all on the GPU.  Either `idx` ([P,K,4] int32, problem-local indices, the caller's
    sampler: parity with `ransac_search`) or `seed` (device Philox sampling, non-parity; `problem_base` = global
    index of the first problem when a longer list is sharded over calls) must be given.
    Results land in `ws` (ws.idx holds the samples actually used)."""
    lib = _lib.load()
    _dev_check(pts_a, pts_b, offsets, needs)
    assert (seed is None) != (idx is None), "give exactly one of seed= (device sampling) and idx= (caller's samples)"
    P = offsets.shape[0] - 1
    assert P == ws.p and needs.shape[0] == P and offsets.dtype == torch.int32 and needs.dtype == torch.int32
    flags = _lib.RWH_BATCH_EARLY_STOP if early_stop else 0     # hypotheses after a problem's early exit are skipped (count -1)
    if idx is None:
        flags |= _lib.RWH_BATCH_DEVICE_SAMPLING
    else:
        assert tuple(idx.shape) == (P, ws.k, 4) and idx.dtype == torch.int32
        ws.idx.copy_(idx)
    check(lib.rwh_ransac_batched(_ptr(pts_a), _ptr(pts_b), _ptr(offsets), P, ws.m_max, ws.k, _ptr(ws.idx),
                                 int(seed or 0) & 0xFFFFFFFFFFFFFFFF, int(problem_base), float(th), RWH_LOSS[loss], _ptr(needs), _ptr(ws.H),
                                 _ptr(ws.flags), _ptr(ws.counts), _ptr(ws.masks) if ws.masks is not None else None,
                                 _ptr(ws.best), flags, _lib.stream_ptr()), "rwh_ransac_batched")
    return ws


def project_points(h9, pts, inverse):
    """Launch the projection kernel: h9 [9] float32, pts [M,2] float32 -> [3,M] float32."""
    lib = _lib.load()
    _dev_check(h9, pts)
    M = pts.shape[0]
    out = torch.empty((3, M), dtype=torch.float32, device=pts.device)
    check(lib.rwh_project_points(_ptr(h9), _ptr(pts), M, 1 if inverse else 0, _ptr(out), _lib.stream_ptr()),
          "rwh_project_points")
    return out


def project_points_ex(h9, pts3):
    """General projection (rwh_project_points_ex): h9 [9], pts3 [3,M], both float32 or both float64 -> [3,M] same dtype."""
    lib = _lib.load()
    _dev_check(h9, pts3)
    assert h9.dtype == pts3.dtype and h9.dtype in (torch.float32, torch.float64) and pts3.shape[0] == 3
    out = torch.empty_like(pts3)
    check(lib.rwh_project_points_ex(_ptr(h9), _ptr(pts3), pts3.shape[1], _DTYPE[h9.dtype], _ptr(out), _lib.stream_ptr()),
          "rwh_project_points_ex")
    return out


def stitch_panorama(img_t, img_q, inv_h, grid_origin, warp_wh, t_origin, q_origin, canvas_hw, blend, rate, zero_origin=True, fast=False):
    """Launch the fused compositor: img_t / img_q [H,W,3] uint8 GPU tensors -> canvas [fh,fw,3] uint8.  fast=True: the
    staged warp kernel with the compositor epilogue (within 1 LSB); False: the exact float64 kernel (bit-identical).
    blend: 0 / False paste, 1 / True 'Rate', 2 'Gradient' (exact kernel only)."""
    lib = _lib.load()
    _dev_check(img_t, img_q)
    assert img_t.dtype == torch.uint8 and img_q.dtype == torch.uint8 and img_t.shape[2] == 3 and img_q.shape[2] == 3
    fh, fw = canvas_hw
    out = torch.empty((fh, fw, 3), dtype=torch.uint8, device=img_t.device)
    ih = np.ascontiguousarray(inv_h, dtype=np.float64).reshape(9)
    check(lib.rwh_stitch_panorama(_ptr(img_t), img_t.shape[0], img_t.shape[1], _ptr(img_q), img_q.shape[0], img_q.shape[1],
                                  ih.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), int(grid_origin[0]), int(grid_origin[1]),
                                  int(warp_wh[0]), int(warp_wh[1]), int(t_origin[0]), int(t_origin[1]), int(q_origin[0]),
                                  int(q_origin[1]), int(fh), int(fw), int(blend), float(rate), _ptr(out),
                                  (RWH_WARP_ZERO_ORIGIN if zero_origin else 0) | (_lib.RWH_STITCH_FAST if fast else 0),
                                  _lib.stream_ptr()), "rwh_stitch_panorama")
    return out


def stitch_panorama_rows(img_t, img_q, inv_h, grid_origin, warp_wh, t_origin, q_origin, out, rows, blend, rate, zero_origin=False):
    """Canvas rows [rows[0], rows[1]) of the exact compositor into `out` ([fh, fw, 3] uint8 on the device): rwh_stitch_panorama_rows,
    on torch's current stream."""
    lib = _lib.load()
    _dev_check(img_t, img_q, out)
    fh, fw = int(out.shape[0]), int(out.shape[1])
    ih = np.ascontiguousarray(inv_h, dtype=np.float64).reshape(9)
    check(lib.rwh_stitch_panorama_rows(_ptr(img_t), img_t.shape[0], img_t.shape[1], _ptr(img_q), img_q.shape[0], img_q.shape[1],
                                       ih.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), int(grid_origin[0]), int(grid_origin[1]),
                                       int(warp_wh[0]), int(warp_wh[1]), int(t_origin[0]), int(t_origin[1]), int(q_origin[0]),
                                       int(q_origin[1]), fh, fw, int(blend), float(rate), _ptr(out), int(rows[0]), int(rows[1]),
                                       RWH_WARP_ZERO_ORIGIN if zero_origin else 0, _lib.stream_ptr()), "rwh_stitch_panorama_rows")


def decode_best(best_words, k_total):
    """Unpack the two argmax words (host ints) -> (winner_index, count, early_exit).
    Word 1 (first index reaching `need`) takes precedence, like the reference's
    `break` (ransac.py:186-190); otherwise word 0 = max count, lowest index."""
    w0, w1 = int(best_words[0]), int(best_words[1])
    if w1 != 0:
        return 0xFFFFFFFF - w1, None, True
    if (w0 >> 32) == 0:  # nothing ever scored > 0: the reference keeps no model (ransac.py:199)
        return None, 0, False
    return 0xFFFFFFFF - (w0 & 0xFFFFFFFF), w0 >> 32, False


def need_count(m, d, n):
    """ransac.py:169,186: count >= m*d/100 + n, as an integer threshold."""
    return int(math.ceil(m * d / 100 + n))


class ClockProbe:
    """Shader clock held while other kernels run (rwh_lab_clock_probe): one wavefront on a side stream that stays resident
    for `ms` milliseconds of the 100 MHz constant clock and counts shader cycles meanwhile.
        p = ClockProbe(ms); ... enqueue the kernels being measured ...; mhz = p.mhz()"""

    def __init__(self, ms):
        lib = _lib.load()
        self.out = torch.zeros(2, dtype=torch.int64, device=_lib.require_gpu())
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())        # the zero fill above
        check(lib.rwh_lab_clock_probe(_ptr(self.out), float(ms), ctypes.c_void_p(self.stream.cuda_stream)), "rwh_lab_clock_probe")

    def mhz(self):
        self.stream.synchronize()
        cyc, ticks = (int(v) for v in self.out.cpu())
        return 100.0 * cyc / ticks if ticks else float("nan")


def power_node(device_index=None):
    """Path of the amdgpu hwmon file holding this device's board power in microwatts (power1_average / power1_input), or
    None.  Read it with open(): a process that has initialised the GPU must not fork + exec a tool like rocm-smi."""
    import glob
    try:
        idx = torch.cuda.current_device() if device_index is None else int(device_index)
        bus = torch.cuda.get_device_properties(idx).pci_bus_id
        dom = getattr(torch.cuda.get_device_properties(idx), "pci_domain_id", 0)
        dev = getattr(torch.cuda.get_device_properties(idx), "pci_device_id", 0)
        pci = "%04x:%02x:%02x.0" % (dom, bus, dev)
    except Exception:
        return None
    for leaf in ("power1_average", "power1_input"):
        hits = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*/%s" % (pci, leaf))
        if hits:
            return hits[0]
    return None


def power_cap_node(device_index=None):
    """The same device's power cap (microwatts), or None."""
    node = power_node(device_index)
    if not node:
        return None
    import os
    cap = os.path.join(os.path.dirname(node), "power1_cap")
    return cap if os.path.exists(cap) else None
