"""MI355X-backed mirror of the reference's `ransac.py` call surface.

`HomoModel`, `RANSAC` and `stitching` keep the reference's names, constructor
arguments, defaults, return values and data layout (features x observations,
ransac.py:159-166).  The k-iteration Python loop of `RANSAC.run`
(ransac.py:176-202) is replaced by ONE library call, `rwh_ransac_search`, which enqueues

    K1  rwh_dlt4_batched   all k four-point DLT hypotheses        (ransac.py:178-180)
    K2  rwh_score_count    k x M reprojection errors, `err < th`,  (ransac.py:182-184)
                           wavefront ballot + popcount, and the
                           order-independent form of the accept
                           rules (packed argmax keys)             (ransac.py:186-202)

followed by the reference's own final N-point refit on the host
(ransac.py:206-211 -> calcHomographyLinear, O(1)).

Sampling parity: the k x 4 index table is drawn from numpy's global legacy
generator exactly as k successive `np.random.randint(0, M, 4)` calls would
(ransac.py:177), and the generator is left where the reference would leave it
(an early exit at iteration e consumes only e+1 draws).

Solver parity (ransac.py:177 draws WITH replacement; homography.py:81-87): a sample with a repeated
index gives a rank-deficient 8 x 9 system, for which the reference still takes whatever null vector
LAPACK's SVD returns and lets it compete for the running best; an ill-conditioned sample (three collinear
source points, equal coordinates at different indices) gives K1 and LAPACK unrelated H; and on ~2 % of
ordinary samples the two round to neighbouring float32 H.  K1 flags the first two kinds
(RWH_HYP_REPEATED / _SINGULAR / _ILLCOND) and `_settle_on_host` re-derives, with the reference's own
arithmetic (float32 DLT matrix -> LAPACK dgesdd, the routine numpy.linalg.svd calls -> /h[8]: `svd_hypotheses`),
every flagged hypothesis and every unflagged one whose count is within a margin of the best (or of the
early-exit count), re-scores those rows with K2 (bit-exact given H) and only then applies the accept rules.
The repeated-index samples are known before anything is launched: `presettle` solves them on the host
while the GPU searches; the whole driver is one native call (`rwh_ransac_run`).  k = 1500 at M = 185: ~90 host solves,
0.36-0.42 ms per run end to end.

Error behaviour (fixture g14, written by the unmodified reference): k = 0 raises UnboundLocalError (ransac.py:203 reads a
variable only the loop assigns), a run in which no hypothesis has an inlier indexes with np.where(None) (an error from numpy
2.1 on), n < 4 raises IndexError after the first draw, a sample holding a NaN coordinate raises LinAlgError at ITS iteration
unless an earlier one took the early exit, a winner with fewer inliers than the refit accepts fails the refit's assertion --
each with numpy's generator left where the reference leaves it.

There is no CPU implementation of the loop here: without librwh_hip.so and a GPU
`RANSAC.run` raises `RwhUnavailable`.
"""
import ctypes
import os

import numpy as np

from . import _lapack, _lib, kernels
from .homography import (_pair_rows, calcHomography, calcHomographyLinear, cylindericlMap,  # noqa: F401
                         stitchPanorama)

# Which hypotheses need the reference's own solver (`_settle_on_host`):
#   * every sample K1 flags: a repeated index (LAPACK's null vector of the rank-deficient system is arbitrary -- and it is what
#     the reference uses), a non-finite result, an ill-conditioned sample (RWH_HYP_ILLCOND: collinear triples, equal
#     coordinates at different indices, |h33| tiny; K1's elimination and LAPACK then round to different float32 H);
#   * second line: every unflagged hypothesis whose count is within a margin of a decision (the best count, the early-exit
#     count).  An unflagged K1 H differs from LAPACK's by float32 round-off on 1.8 % of natural samples, which moves its
#     inlier count by <= 2 on all ~130 000 golden hypotheses and by <= 8 on the lattice / cluster stress sets of
#     tests/golden/g12_illcond.npz (measured with a float64 emulation of K1, tools/README).  The margin shrinks with the
#     best count (a count of 17 cannot move by 8): `_margin(best)`.
RESCORE_MARGIN = 8


def _margin(best, cap):
    """Margin around a decision count `best`: min(cap, 3 + best // 16), nondecreasing in `best` and growing by at most
    1 per 16 counts, so `best - _margin(best)` is nondecreasing too (a hypothesis inside the margin of a larger best is
    inside the margin of every smaller one: what lets each shard settle on its own, sharded.gpu_score_slice)."""
    return min(int(cap), 3 + int(best) // 16)


LVL = 0


def DEBUG(*args):
    if LVL >= 1:
        print("[DEBUG]", *args)


def _weak_threshold(th):
    """`err_total < self.th` (ransac.py:183) compares float32 errors with `th`:
    Python scalars are weak (compared as float32), numpy float64 scalars promote
    the comparison to float64.  Return the double the kernel must compare with."""
    if type(th) in (int, float, bool) or isinstance(th, (np.float32, np.float16, np.integer)):
        return float(np.float32(th))
    return float(th)


def _points_rows(P):
    """features x observations (2 x M or 3 x M) -> contiguous M x 2 float32.

    The search kernels take float32 correspondences -- what the reference's own pipeline hands to RANSAC.run (ransac.py:263-267:
    np.float32 keypoints; matchespoints.npy is float32).  On float64 or integer arrays the reference forms the DLT products in THAT
    dtype (homography.py:6-13) and -- even when every value is a float32 value -- the distances in float64 (ransac.py:78-82:
    float32 projection minus a float64 / integer target), so a borderline pair can fall the other way than in a float32 search.
    Round 3 cast such arrays with a warning; they are now refused: pass `X.astype(np.float32)` to get the float32 search."""
    P = np.asarray(P)
    if P.dtype != np.float32:
        raise TypeError("RANSAC: %s correspondences: the MI355X search computes in float32 like the reference's own pipeline "
                        "(ransac.py:263-267); on this input the reference forms its DLT products in %s and its distances in float64, "
                        "which this path does not reproduce -- pass float32 arrays (X.astype(np.float32))" % (P.dtype, P.dtype))
    return np.ascontiguousarray(P.T[:, :2])


def legacy_randint_table(m, k, n, want64=True):
    """np.random.randint(0, m, (k, n)) from numpy's GLOBAL legacy generator -- the stream ransac.py:177 consumes, one draw
    of n per iteration -- with the generator left exactly where that call leaves it.  -> (int64 [k, n] or None, int32 [k, n]).

    Native form (`rwh_host_legacy_randint`, host code of librwh_hip.so): MT19937 + numpy's masked rejection in a tight loop on
    the state `np.random.get_state()` hands out, written back with `set_state()`: 0.4 ms for 400 000 draws where numpy's own
    call takes 2.1 ms (39 % of a k = 100 000 run).  Identical output and generator position (tests/test_settle_cpu.py);
    any other bit generator, a missing library or an unexpected state layout take numpy's own call."""
    m, k, n = int(m), int(k), int(n)
    if k * n >= 4096 and 1 <= m < 2 ** 31 and os.path.exists(_lib.LIB_PATH):
        try:
            st = np.random.get_state()
            if st[0] == "MT19937" and len(st[1]) == 624 and 0 <= int(st[2]) <= 624:
                key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
                pos = ctypes.c_int32(int(st[2]))
                out32 = np.empty((k, n), dtype=np.int32)
                out64 = np.empty((k, n), dtype=np.int64) if want64 else None
                rc = _lib.load().rwh_host_legacy_randint(key.ctypes.data, ctypes.byref(pos), m, k * n, out32.ctypes.data,
                                                         out64.ctypes.data if want64 else None)
                if rc == 0:
                    np.random.set_state((st[0], key, int(pos.value), st[3], st[4]))
                    return out64, out32
        except (_lib.RwhUnavailable, OSError, ValueError, TypeError):
            pass
    t = np.random.randint(0, m, (k, n))
    return t, np.ascontiguousarray(t, dtype=np.int32)


def _host_thread_share():
    """Host threads of the settle step's LAPACK loop: this process's share of the cores it may run on -- one rank per GPU under
    torchrun (LOCAL_WORLD_SIZE), so 8 ranks on a 256-thread host take 32 each -- capped at 32 (k = 100 000 on matchespoints: 3 257
    repeated-index SVDs take 1.0 ms on 16 threads, 0.55 ms on 32, 0.44 ms on 64: tools/run_phases.py)."""
    try:
        cpus = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cpus = os.cpu_count() or 1
    try:
        ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        ranks = 1
    return max(1, min(32, cpus // ranks))


HOST_THREADS = _host_thread_share()
FORCE_PYTHON_DRIVER = False        # tests: RANSAC.run through the Python twin of rwh_ransac_run


def svd_hypotheses(pts_a, pts_b, idx_rows, threads=None):
    """The reference's 4-point solve (homography.py:4-14, 71-88) for n samples at once on the host:
    float32 DLT matrices -> LAPACK dgesdd (float64 inside, cast back to float32) -> last right-singular vector / its
    9th element.  -> float32 [n, 9].

    Native form: `rwh_host_dlt4_svd` runs the loop in librwh_hip.so's host code on `threads` cores, calling -- by address
    -- the very dgesdd numpy.linalg.svd calls (`_lapack.dgesdd_address`): the same numbers bit for bit
    (tests/test_settle_cpu.py), ~6 us per sample per core instead of ~11 us under the interpreter lock.  If that symbol
    cannot be found (another numpy build) or the library is missing, the stacked numpy call below does the same work."""
    idx_rows = np.ascontiguousarray(np.asarray(idx_rows).reshape(-1, 4), dtype=np.int32)
    n = idx_rows.shape[0]
    addr = _lapack.dgesdd_address()
    if addr is not None and n and os.path.exists(_lib.LIB_PATH):
        pa = np.ascontiguousarray(pts_a, dtype=np.float32)
        pb = np.ascontiguousarray(pts_b, dtype=np.float32)
        out = np.empty((n, 9), dtype=np.float32)
        st = _lib.load().rwh_host_dlt4_svd(pa.ctypes.data, pb.ctypes.data, pa.shape[0], idx_rows.ctypes.data, n,
                                           ctypes.c_void_p(addr), int(threads or HOST_THREADS), out.ctypes.data)
        if st == 0:
            return out
    flat = idx_rows.reshape(-1)
    mats = _pair_rows(pts_a[flat], pts_b[flat], -1).reshape(-1, 8, 9)
    with np.errstate(all="ignore"):          # h[8] == 0 divides like the reference does (inf / nan rows score 0)
        _, _, vt = np.linalg.svd(mats)
        h = vt[:, -1, :]
        return np.ascontiguousarray(h / h[:, 8:9])


def repeated_rows(idx):
    """Samples of a K x >=4 index table whose first four indices are not distinct (what K1 flags RWH_HYP_REPEATED):
    known to the host before anything is launched."""
    a, b, c, d = (np.asarray(idx)[:, i] for i in range(4))
    return (a == b) | (a == c) | (a == d) | (b == c) | (b == d) | (c == d)


class _Settled(object):
    """Hypotheses re-derived with the reference's solver so far: index -> (H row, count, where its mask lives)."""

    def __init__(self, k):
        self.done = np.zeros(k, dtype=bool)
        self.batches = []          # (indices, H float32 [n, 9], mask tensor [n, words] on the device)
        self.where = {}

    def add(self, cand, H, masks):
        b = len(self.batches)
        self.batches.append((cand, H, masks))
        self.done[cand] = True
        for j, i in enumerate(cand.tolist()):
            self.where[i] = (b, j)

    def mask_words(self, i):
        if i not in self.where:
            return None
        b, j = self.where[i]
        return self.batches[b][2][j].cpu().numpy()

    def rows(self):
        return {i: self.batches[b][1][j] for i, (b, j) in self.where.items()}


def _score_settled(H, pa_dev, pb_dev, th, method):
    """K2 on hypotheses the host solved (H: [n, 9] float32 host array) -> (counts, masks) device tensors.  'backward' and
    'reproj' project through numpy.linalg.inv(H) (ransac.py:74): these rows get numpy's own inverse, not the kernel's
    elimination, which rounds apart from LAPACK's on nearly singular H (rwh.h, rwh_score_count_inv)."""
    import torch
    dev = pa_dev.device
    hd = torch.from_numpy(H).to(dev)
    hinv = None if method == "fwd" else torch.from_numpy(kernels.host_inverses(H)).to(dev)
    cnt, msk, _ = kernels.score_count(hd, pa_dev, pb_dev, th, method, 1 << 30, kernels.scratch_best(dev), hinv=hinv)
    return cnt, msk


def presettle(pa_dev, pb_dev, pa, pb, idx_host, rows, th, method):
    """First part of the settle step, for samples the HOST can name before the GPU has said anything (repeated indices):
    their reference H by `svd_hypotheses` -- host time that overlaps the search already enqueued on the GPU -- then K2 on
    those rows, enqueued behind the search.  Returns (rows, H, counts tensor, masks tensor) for `_settle_on_host(pre=...)`."""
    import torch
    rows = np.asarray(rows, dtype=np.int64)
    if rows.size == 0:
        return None
    H = svd_hypotheses(pa, pb, idx_host[rows][:, :4])
    cnt, msk = _score_settled(H, pa_dev, pb_dev, th, method)
    return rows, H, cnt, msk


# 'fwd' searches (round 4): which hypotheses need the reference's own solver is decided by count INTERVALS (rwh_score_interval),
# not by a flat margin -- the native driver's rule (csrc/rwh_run.hip), mirrored in `_settle_on_host`.
IV_NEAR = 32                               # hypotheses within this many counts of the best / of `need` get an interval
IV_DELTA0, IV_DELTA1 = 2.0 ** -20, 2.0 ** -18     # perturbation budgets (natural scale of an entry): unflagged / RWH_HYP_ILLCOND rows


def _settle_on_host(pa_dev, pb_dev, pa, pb, idx_host, counts, flags, need, th, method, margin, stats=None, pre=None, H_dev=None,
                    flags_dev=None):
    """Accept rules of ransac.py:186-202 over K1/K2's results, exact with respect to the reference's solver.

    counts / flags: host copies of K2's counts and K1's flags for the k hypotheses of `idx_host`; `pre`: what `presettle`
    returned (its counts are read here).  Hypotheses are "settled" (H from the host SVD, count + mask from K2 on that H)
    in rounds until every hypothesis that can take part in the decision is settled:
      * nothing after the first hypothesis that certainly reaches `need` is ever looked at by the reference (`break`,
        ransac.py:186-190);
      * inside that prefix, 'fwd' with K1's H at hand (`H_dev`): the samples whose H says nothing (repeated index, non-finite,
        RWH_HYP_DEGENERATE) and, of the candidates -- RWH_HYP_ILLCOND samples, hypotheses within IV_NEAR of the best or of
        `need`, each with a count interval [lo, hi] from rwh_score_interval --, those whose interval is not a point and reaches
        the best lower bound or `need`;
      * 'backward' / 'reproj' (or no H_dev): every flagged hypothesis, every unflagged one within `_margin` of `need` or of
        the best trustworthy count (the rule of rounds 2-3).
    Usually ONE round beyond `pre`.  Returns (winner | None, early, count, mask_words | None, H_rows, counts): mask_words
    is the winner's uint64 mask if the winner was settled here (None: take K2's own mask for it), H_rows maps settled
    index -> float32[9], counts is the int64 count table with the settled entries replaced."""
    import torch
    k = counts.shape[0]
    counts = counts.astype(np.int64)
    st = _Settled(k)
    flags = np.asarray(flags)
    suspect = flags != 0
    n_rounds = 0
    if pre is not None:
        rows, H, cnt, msk = pre
        counts[rows] = cnt.cpu().numpy()
        st.add(rows, H, msk)
    with np.errstate(invalid="ignore"):
        cscale = float(max(1.0, np.abs(pa).max())) if pa.size else 1.0
    use_iv = method == "fwd" and H_dev is not None and cscale < 1e30
    n_iv = 0
    if use_iv:
        always = (flags & (_lib.RWH_HYP_REPEATED | _lib.RWH_HYP_SINGULAR | _lib.RWH_HYP_DEGENERATE)) != 0
        lo, hi = counts.copy(), counts.copy()              # not a candidate: its count is taken as it is
        free = ~always
        best0 = int(counts[free].max()) if free.any() else 0
        cand0 = np.flatnonzero(~st.done & free & (((flags & _lib.RWH_HYP_ILLCOND) != 0) | (counts >= best0 - IV_NEAR) | (counts >= need - IV_NEAR)))
        if cand0.size:
            if flags_dev is None:
                flags_dev = torch.from_numpy(np.ascontiguousarray(flags, dtype=np.uint8)).to(H_dev.device)
            lo[cand0], hi[cand0] = kernels.score_interval(H_dev, cand0, flags_dev, pa_dev, pb_dev, th, cscale, IV_DELTA0, IV_DELTA1)
        n_iv = int(cand0.size)
    while True:
        settled = st.done
        if use_iv:
            sure = np.where(settled, counts >= need, ~always & (lo >= need))
        else:
            sure = np.where(settled, counts >= need, ~suspect & (counts >= need + _margin(need, margin)))
        hit = np.flatnonzero(sure)
        end = int(hit[0]) + 1 if hit.size else k
        c = counts[:end]
        open_ = ~settled[:end]
        if use_iv:
            v = np.where(settled[:end], c, np.where(always[:end], 0, lo[:end]))
            best = int(v.max()) if v.size else 0
            cand = np.flatnonzero(open_ & (always[:end] | ((lo[:end] < hi[:end]) & ((hi[:end] >= best) | (hi[:end] >= need)))))
        else:
            trusted = settled | ~suspect                  # counts that mean something: the reference's, or K1's within a margin
            tc = c[trusted[:end]]
            best = int(tc.max()) if tc.size else 0
            cand = np.flatnonzero(open_ & (suspect[:end] | (c >= best - _margin(best, margin)) | (c >= need - _margin(need, margin))))
        if cand.size == 0:
            break
        n_rounds += 1
        H = svd_hypotheses(pa, pb, idx_host[cand][:, :4])
        cnt, msk = _score_settled(H, pa_dev, pb_dev, th, method)
        counts[cand] = cnt.cpu().numpy()
        st.add(cand, H, msk)
    if stats is not None:
        stats["host_settled"] = int(st.done.sum())
        stats["host_rounds"] = n_rounds
        stats["flagged"] = int(suspect.sum())
        stats["intervals"] = n_iv
    hit = np.flatnonzero(c >= need)
    if hit.size:
        w, early = int(hit[0]), True
    elif end and c.max() > 0:
        w, early = int(np.argmax(c)), False               # first index of the maximum (ransac.py:199: strict >)
    else:
        return None, False, 0, None, st.rows(), counts
    return w, early, int(c[w]), st.mask_words(w), st.rows(), counts


class Model(object):
    __slots__ = ('val', 'th', 'd', 'n')

    def fit(self, X, Y):
        raise NotImplementedError

    def fwd(self, X):
        raise NotImplementedError

    def dist(self, predY, trueY):
        raise NotImplementedError


class HomoModel(Model):
    """ransac.py:22-98."""

    def __init__(self, th=5, d=50, n=4):
        self.th = th
        self.d = d
        self.n = n
        self.val = np.empty((3, 3), dtype=np.float32)

    def fit(self, X, Y, collective=False):
        """ransac.py:30-53.  X, Y: 2 x n or 3 x n (n == self.n, or more with collective=True)."""
        nx, mx = X.shape
        ny, my = Y.shape
        assert ((mx == my) and (mx == self.n)) or ((mx == my) and (mx > self.n) and collective), \
            "invalid data size should be %d" % self.n
        assert (nx == ny) and nx in [2, 3], "invalid input dimension for row numbers"
        if collective:
            self.val = calcHomographyLinear(X.T[:, :2], Y.T[:, :2], True)
        else:
            self.val = calcHomography(X.T[:, :2], Y.T[:, :2], False)
        return self.val

    # -- projection helpers: one launch of the projection kernel each -------------------------
    def _project(self, P, inverse):
        """ransac.py:55-76 with numpy's dtype rules: a 2-row input becomes a float32 3 x M with ones (ransac.py:59-60), a
        3-row input is used as it is (third row included); `val @ x` is float32 only if both operands are, float64
        otherwise; reproj goes through numpy.linalg.inv(val) in val's dtype (host, 3 x 3)."""
        import torch
        P = np.asarray(P)
        nrow, m = P.shape
        assert nrow in [2, 3], "invalid input dimension for row numbers"
        dev = _lib.require_gpu()
        val = np.asarray(self.val)
        if nrow == 2:
            x = np.ones((3, m), dtype=np.float32)
            x[:2, :] = P
        else:
            x = P
        if m < 2:
            # no point: numpy's empty product.  ONE point: numpy hands a 3 x 3 by 3 x 1 product to BLAS's matrix-VECTOR routine,
            # whose sums round differently from the matrix-matrix routine every other width takes (and the kernels reproduce):
            # the reference's own expression on the host (ransac.py:63-64 / 74-76), nine multiply-adds
            h = np.linalg.inv(val) if inverse else val
            y = h @ x
            return y / (y[-1, :] + 1e-10)
        if val.dtype == np.float32 and x.dtype == np.float32 and nrow == 2:
            # the RANSAC loop's own case: float32 H, w == 1, inverse by the kernel's float64 LU (bit-identical to numpy's)
            if inverse:
                val = np.linalg.inv(val)                  # ransac.py:74: numpy's own float32 inverse (LinAlgError on a singular val);
            h9 = torch.from_numpy(np.ascontiguousarray(val).reshape(9)).to(dev)      # the kernel then projects forward through it
            pts = torch.from_numpy(np.ascontiguousarray(x[:2].T)).to(dev)
            return kernels.project_points(h9, pts, False).cpu().numpy()
        if inverse:
            val = np.linalg.inv(val)                      # ransac.py:74: float64 inside, result in val's dtype
        dt = np.result_type(val.dtype, x.dtype)
        if dt not in (np.float32, np.float64):
            dt = np.dtype(np.float64)
        h9 = torch.from_numpy(np.ascontiguousarray(val, dtype=dt).reshape(9)).to(dev)
        pts3 = torch.from_numpy(np.ascontiguousarray(x, dtype=dt)).to(dev)
        return kernels.project_points_ex(h9, pts3).cpu().numpy()

    def fwd(self, X):
        """val @ [X;1] / (row2 + 1e-10), 3 x M in numpy's result dtype (ransac.py:55-64)."""
        return self._project(X, False)

    def reproj(self, Y):
        """inv(val) @ [Y;1] / (row2 + 1e-10) (ransac.py:66-76)."""
        return self._project(Y, True)

    def dist(self, predY, trueY):
        """Column-wise L2 distance (ransac.py:78-82); two arrays in, one out: host numpy."""
        delta = (predY - trueY)
        delta = np.sum(delta * delta, axis=0)
        return np.sqrt(delta)

    def computeLoss(self, X, Y, method="reproj"):
        """Per-correspondence loss, float32 [M] (ransac.py:84-98), from the scorer kernel."""
        import torch
        if method not in _lib.RWH_LOSS:
            exit("Invalid method!")  # ransac.py:97
        if method != "fwd":
            np.linalg.inv(self.val)                       # ransac.py:74 raises LinAlgError on a singular val
        if np.asarray(self.val).dtype != np.float32 or np.asarray(X).dtype != np.float32 or np.asarray(Y).dtype != np.float32 \
                or X.shape[0] != 2 or Y.shape[0] != 2 or X.shape[1] < 2:
            # not the RANSAC loop's float32 case (e.g. model.val is the float64 refit after run()): numpy computes these in
            # float64 -- the reference's own composition of fwd / reproj / dist (ransac.py:84-96), projections on the GPU
            err = None
            if method in ("fwd", "reproj"):
                err = self.dist(self.fwd(X)[:2, :], Y)
            if method in ("backward", "reproj"):
                back = self.dist(self.reproj(Y)[:2, :], X)
                err = back if err is None else err + back
            return err
        dev = _lib.require_gpu()
        h9 = torch.from_numpy(np.ascontiguousarray(self.val, dtype=np.float32).reshape(1, 9)).to(dev)
        pa = torch.from_numpy(_points_rows(X)).to(dev)
        pb = torch.from_numpy(_points_rows(Y)).to(dev)
        best = kernels.new_best(dev)
        hinv = None if method == "fwd" else torch.from_numpy(kernels.host_inverses(np.asarray(self.val, dtype=np.float32).reshape(1, 9))).to(dev)
        _, _, err = kernels.score_count(h9, pa, pb, 0.0, method, 1 << 30, best, want_masks=False, want_err=True, hinv=hinv)
        return err[0].cpu().numpy()


class RANSAC(object):
    """ransac.py:137-213."""

    __slots__ = ('model', 'th', 'd', 'n', 'k', 'last_run', 'rescore_margin')

    def __init__(self, model, k=1000):
        self.model = model
        self.th = model.th
        self.d = model.d
        self.n = model.n
        self.k = k
        self.last_run = None
        self.rescore_margin = RESCORE_MARGIN

    def computeLoss(self, X, Y, method="reproj"):
        return self.model.computeLoss(X, Y, method)

    def run(self, data, method="reproj"):
        """Returns (finalModel float64 3x3, (inlier_indices,), count) like ransac.py:159-213 and
        sets `model.val`.  `self.last_run` keeps diagnostics (winner index, early-exit flag,
        per-hypothesis flags/counts tensors) that the reference does not expose."""
        import torch
        X, Y = data
        nx, mx = X.shape
        ny, my = Y.shape
        assert mx == my, "data observation not consistent!"
        if method not in _lib.RWH_LOSS:
            exit("Invalid method!")
        k = int(self.k)
        if k <= 0:      # the loop body never runs and ransac.py:203 reads the count it would have assigned
            raise UnboundLocalError("local variable 'lenalsoIninears' referenced before assignment")
        if self.n < 4:
            # the first iteration draws its sample (ransac.py:177), then ransac.py:180 -> homography.py:9: calc_corresp reads u[3]
            np.random.randint(0, mx, self.n)
            raise IndexError("index 3 is out of bounds for axis 0 with size %d" % self.n)
        dev = _lib.require_gpu()
        need = mx * self.d / 100 + self.n

        # sampling: identical stream to k successive randint(0, mx, n) calls (ransac.py:177); the model is fitted on the
        # first four of the n sampled correspondences (ransac.py:180 -> homography.py:4-14)
        rng_state = np.random.get_state()
        idx_host, idx_n32 = legacy_randint_table(mx, k, self.n, want64=False)
        if idx_host is None:
            idx_host = idx_n32        # (int32: the same values; numpy's own call would have returned int64)

        pa_host, pb_host = _points_rows(X), _points_rows(Y)
        # A sample that holds a NaN coordinate makes the reference's SVD raise LinAlgError at ITS iteration (homography.py:81:
        # LAPACK's dgesdd rejects a matrix with a NaN) -- unless an earlier iteration has taken the early exit.  Such samples are
        # known from the index table: search the iterations before the first of them, and raise where the reference would.
        # (+-Inf coordinates pass LAPACK's check: those samples are solved like any other flagged sample.)
        first_bad = None
        with np.errstate(invalid="ignore"):
            maybe_nan = bool(np.isnan(pa_host.sum() + pb_host.sum()))      # one reduction per run (also NaN for Inf - Inf: sorted out below)
        if maybe_nan:
            nonfinite = np.isnan(pa_host).any(axis=1) | np.isnan(pb_host).any(axis=1)
            bad_rows = nonfinite[idx_host[:, :4]].any(axis=1)
            if bad_rows.any():
                first_bad = int(np.argmax(bad_rows))
                if first_bad == 0:
                    np.random.set_state(rng_state)
                    np.random.randint(0, mx, (1, self.n))
                    raise np.linalg.LinAlgError("SVD did not converge")
                k = first_bad
                idx_host = idx_host[:k]
        idx32 = np.ascontiguousarray(idx_n32[:k, :4])
        need_i = kernels.need_count(mx, self.d, self.n)
        th = _weak_threshold(self.th)
        addr = _lapack.dgesdd_address()
        gesv = None if method == "fwd" else _lapack.dgesv_address()
        if addr is not None and not FORCE_PYTHON_DRIVER and (method == "fwd" or gesv is not None):
            # the whole driver in ONE native call (rwh_ransac_run, csrc/rwh_run.hip): upload, K1 + K2 + argmax, the settle step
            # (repeated-index samples solved on host threads while the GPU searches), the accept rules
            ws = kernels.RunWorkspace(mx, k, dev)
            try:
                winner, early, totalfit, n_set, n_rounds, n_flagged, mask_words, _keys, n_iv = kernels.ransac_run(
                    pa_host, pb_host, idx32, th, method, need_i, self.rescore_margin, ws, addr, HOST_THREADS,
                    dgesv=gesv, want_keys=True)
            except _lib.RwhError:
                # LAPACK refused a sample (info != 0: e.g. an Inf coordinate times 0 is a NaN in the DLT matrix): the step-by-step
                # twin reaches numpy.linalg.svd, which raises the reference's LinAlgError for it
                ws = None
        else:
            ws = None
        if ws is not None:
            counts_host = ws.host_counts(settled=True)
            stats = {"raw_counts": ws.host_counts(), "host_settled": n_set, "host_rounds": n_rounds, "flagged": n_flagged, "intervals": n_iv}
            Hs, flags, counts, masks = ws.H, ws.flags, ws.counts, ws.masks
            settled_rows = _SettledRows(pa_host, pb_host, idx32)
        else:
            winner, early, totalfit, mask_words, settled_rows, counts_host, stats, (Hs, flags, counts, masks) = self._run_python_driver(
                pa_host, pb_host, idx_host, idx32, k, mx, need_i, th, method, dev)

        if early:  # leave the generator where the reference's `break` would
            np.random.set_state(rng_state)
            np.random.randint(0, mx, (winner + 1, self.n))
        elif first_bad is not None:  # no early exit before the sample whose SVD fails
            np.random.set_state(rng_state)
            np.random.randint(0, mx, (first_bad + 1, self.n))
            raise np.linalg.LinAlgError("SVD did not converge")
        elif k and counts_host[k - 1] < need:  # ransac.py:203-204 tests the LAST iteration's count
            print("Warning:: fitting model does not exceed required threshold %d vs %d" % (totalfit, need))

        if winner is None:
            # ransac.py:206 with inliers_pos_final = None: whatever the installed numpy makes of np.where(None) -- an empty index
            # (then the refit asserts, ransac.py:38) up to numpy 2.0, ValueError from 2.1 on
            inliers = np.where(None)
            totalfit = 0
        else:
            words = mask_words
            bits = np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")[:mx]
            inliers = (np.nonzero(bits)[0].astype(np.int64),)
            totalfit = np.int64(totalfit)
        self.last_run = {"winner": winner, "early_exit": early, "counts": counts, "flags": flags,
                         "hypotheses": Hs, "idx": idx_host, "settled": settled_rows, **stats}
        inliers_x = X[:, inliers[0]]
        inliers_y = Y[:, inliers[0]]
        DEBUG("Fitting final model using all inliers")
        finalModel = self.model.fit(inliers_x, inliers_y, collective=True)
        self.model.val = finalModel
        return finalModel, inliers, totalfit


    def _run_python_driver(self, pa_host, pb_host, idx_host, idx32, k, mx, need_i, th, method, dev):
        """The same driver step by step from Python (rounds 2-3; used when numpy's LAPACK cannot be taken by address, and by
        tests that compare the two drivers): one upload, rwh_ransac_search, `presettle` while the GPU searches, one readback,
        `_settle_on_host`."""
        import torch
        if k == 0:      # no iteration: the reference keeps no model and its refit fails (ransac.py:206-208)
            e = torch.empty(0, dtype=torch.int32, device=dev)
            return None, False, 0, None, {}, np.zeros(0, np.int32), {"raw_counts": np.zeros(0, np.int32), "host_settled": 0, "host_rounds": 0,
                                                                     "flagged": 0}, (e, e.to(torch.uint8), e, e)
        blob = torch.from_numpy(np.concatenate([pa_host.reshape(-1).view(np.uint8), pb_host.reshape(-1).view(np.uint8),
                                                idx32.reshape(-1).view(np.uint8)])).to(dev)
        nb = 8 * mx
        pa = blob[:nb].view(torch.float32).reshape(mx, 2)
        pb = blob[nb:2 * nb].view(torch.float32).reshape(mx, 2)
        idx = blob[2 * nb:].view(torch.int32).reshape(k, 4)
        ws = kernels.SearchWorkspace(k, mx, dev)
        kernels.ransac_search(pa, pb, idx, th, method, need_i, ws)             # enqueued; the host goes on
        pre = presettle(pa, pb, pa_host, pb_host, idx_host, np.flatnonzero(repeated_rows(idx_host)), th, method)
        counts_host, flags_host = ws.counts_flags()                           # one readback for counts + flags
        stats = {"raw_counts": counts_host.copy()}           # K2 on K1's own H, before the settle step
        winner, early, totalfit, mask_words, settled_rows, counts_host = _settle_on_host(
            pa, pb, pa_host, pb_host, idx_host, counts_host, flags_host, need_i, th, method, self.rescore_margin, stats, pre=pre,
            H_dev=ws.H, flags_dev=ws.flags)
        if winner is not None and mask_words is None:
            mask_words = ws.masks[winner].cpu().numpy()
        return winner, early, totalfit, mask_words, settled_rows, counts_host, stats, (ws.H, ws.flags, ws.counts, ws.masks)


class _SettledRows(object):
    """`RANSAC.last_run["settled"]` of the native driver: index -> the H the settle step gives that hypothesis, i.e. the
    reference's own (host SVD), computed on demand."""

    def __init__(self, pa, pb, idx32):
        self._pa, self._pb, self._idx = pa, pb, idx32

    def __getitem__(self, i):
        return svd_hypotheses(self._pa, self._pb, self._idx[int(i):int(i) + 1])[0]


class DeviceProblems(object):
    """Correspondences that already live on the GPU (the hand-off of SURVEY.md 8f row f-3): the problems' matches
    concatenated, pts_a / pts_b float32 [total, 2] torch tensors on the device (the `matchespoints` layout, points in rows),
    `sizes` = correspondences per problem (host ints).  `run_batch` takes it in place of the list of [X, Y] arrays."""

    def __init__(self, pts_a, pts_b, sizes):
        import torch
        self.sizes = [int(m) for m in sizes]
        assert pts_a.is_cuda and pts_b.is_cuda and pts_a.dtype == torch.float32 and pts_b.dtype == torch.float32
        assert tuple(pts_a.shape) == tuple(pts_b.shape) == (sum(self.sizes), 2)
        self.pts_a, self.pts_b = pts_a.contiguous(), pts_b.contiguous()

    def __len__(self):
        return len(self.sizes)


def run_batch(datas, th=5, d=50, n=4, k=1000, method="reproj", seed=0, idx=None, problem_base=0, refit=True, info=None):
    """RANSAC over MANY image pairs in one GPU submission (the reference has no
    counterpart: its RANSAC.run handles one pair per call, ransac.py:159-213).

    datas: list of [X, Y] (each 2 x M_p or 3 x M_p, the `RANSAC.run` layout), or a `DeviceProblems` (correspondences
    already on the GPU: nothing but the winners' inlier masks -- and, with refit=True, the points for the host refit --
    comes back).  Returns a list of `(finalModel float64 3x3, (inlier_indices,), count)` -- per problem exactly what
    `RANSAC.run` returns for the same samples, including the final N-point refit on the host (ransac.py:206-211);
    `finalModel` is None for a problem whose winner has too few inliers to refit (where `RANSAC.run` raises
    AssertionError) and with refit=False (no correspondence then visits the host at all).

    Sampling: by default on the device (Philox4x32-10 keyed by `seed`, four distinct correspondences per hypothesis):
    a documented NON-PARITY mode -- the reference draws with replacement from numpy's global legacy generator
    (ransac.py:177).  Pass `idx` (list of [k,4] integer arrays, one per problem) to supply the samples yourself; each
    problem's result then equals `RANSAC.run` on that table bit for bit.  The early-exit rule (ransac.py:186-190)
    holds per problem either way: the first hypothesis whose count reaches M*d/100 + n wins -- and in the device-sampling
    mode it also stops the work: scorer waves of later hypotheses of that problem skip (RWH_BATCH_EARLY_STOP; with the
    caller's tables every hypothesis is scored, because the host settle step may move the exit).  `problem_base`: global
    index of datas[0] when a longer problem list is split over several calls (sharded.run_batch_sharded), so that the
    device sampler draws the tables of the unsplit run.  `info`: optional dict, receives "scored" (hypotheses actually
    scored per problem) and "early" (per problem: did it exit early)."""
    import torch
    if method not in _lib.RWH_LOSS:
        exit("Invalid method!")
    if n < 4:
        raise IndexError("index 3 is out of bounds for axis 0 with size %d" % n)   # as RANSAC.run (homography.py:9)
    dev = _lib.require_gpu()
    P = len(datas)
    if P == 0:
        return []
    on_device = isinstance(datas, DeviceProblems)
    if on_device:
        sizes = datas.sizes
        pa, pb = datas.pts_a, datas.pts_b
    else:
        sizes = []
        for X, Y in datas:
            assert X.shape[1] == Y.shape[1], "data observation not consistent!"
            sizes.append(X.shape[1])
        pa = torch.from_numpy(np.concatenate([_points_rows(X) for X, _ in datas])).to(dev)
        pb = torch.from_numpy(np.concatenate([_points_rows(Y) for _, Y in datas])).to(dev)
    offsets = np.zeros(P + 1, dtype=np.int32)
    offsets[1:] = np.cumsum(sizes)
    needs_host = [kernels.need_count(m, d, n) for m in sizes]
    needs = torch.tensor(needs_host, dtype=torch.int32, device=dev)
    ws = kernels.BatchWorkspace(P, int(k), max(max(sizes), 1), dev)
    if idx is not None:
        if len(idx) != P:
            raise ValueError("idx: one [k, n] table per problem (%d tables for %d problems)" % (len(idx), P))
        tables = []
        for p_, t in enumerate(idx):        # numpy's own indexing rules (ransac.py:178 `data[:, idx]`): negative indices wrap, others raise
            t = np.asarray(t)
            if t.ndim != 2 or t.shape[0] != int(k) or t.shape[1] < 4:
                raise ValueError("idx[%d]: expected a [%d, >= 4] integer array, got shape %s" % (p_, int(k), t.shape))
            m_ = sizes[p_]
            if t.size and (int(t.min()) < -m_ or int(t.max()) >= m_):
                bad = int(t.max()) if int(t.max()) >= m_ else int(t.min())
                raise IndexError("index %d is out of bounds for axis 1 with size %d" % (bad, m_))
            tables.append(np.where(t < 0, t + m_, t))
        idx = tables
        table = torch.from_numpy(np.stack([t[:, :4].astype(np.int32) for t in idx])).to(dev)   # fit on the first four
        kernels.ransac_batched(pa, pb, torch.from_numpy(offsets).to(dev), needs, _weak_threshold(th), method, ws, idx=table)
    else:
        kernels.ransac_batched(pa, pb, torch.from_numpy(offsets).to(dev), needs, _weak_threshold(th), method, ws, seed=seed,
                               problem_base=problem_base, early_stop=True)
    pa_host = pb_host = None
    if idx is not None or (refit and on_device):
        pa_host, pb_host = pa.cpu().numpy(), pb.cpu().numpy()
    if idx is not None:
        # the caller's tables may hold repeated indices (numpy's sampler draws with replacement): settle every problem
        # with the reference's solver, exactly as RANSAC.run does
        counts_host = ws.counts.cpu().numpy()
        flags_host = ws.flags.cpu().numpy()
        winners, win_counts, host_masks = [], [], []
        for p in range(P):
            o0, o1 = int(offsets[p]), int(offsets[p + 1])
            w, early, cnt, words, _, _ = _settle_on_host(pa[o0:o1], pb[o0:o1], pa_host[o0:o1], pb_host[o0:o1], np.asarray(idx[p])[:, :4],
                                                     counts_host[p], flags_host[p], needs_host[p], _weak_threshold(th),
                                                     method, RESCORE_MARGIN, H_dev=ws.H[p], flags_dev=ws.flags[p])
            winners.append((w, None, early)); win_counts.append(cnt); host_masks.append(words)
    else:
        best = ws.best.cpu().numpy()
        winners = [kernels.decode_best(best[p], int(k)) for p in range(P)]
        host_masks = [None] * P
    rows = torch.tensor([[p, w[0] if w[0] is not None else 0] for p, w in enumerate(winners)], device=dev)
    win_masks = ws.masks[rows[:, 0], rows[:, 1]].cpu().numpy()          # one gather, one copy for all problems
    if idx is None:
        win_counts = ws.counts[rows[:, 0], rows[:, 1]].cpu().numpy()
    if info is not None:
        info["scored"] = (ws.counts >= 0).sum(dim=1).cpu().numpy()
        info["early"] = [bool(w[2]) for w in winners]
    out = []
    for p, (winner, _, early) in enumerate(winners):
        model = HomoModel(th=th, d=d, n=n)
        if winner is None or int(win_counts[p]) == 0:
            inliers, total = (np.array([], dtype=np.int64),), 0
        else:
            words = host_masks[p] if host_masks[p] is not None else win_masks[p]
            bits = np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")[:sizes[p]]
            inliers, total = (np.nonzero(bits)[0].astype(np.int64),), np.int64(win_counts[p])
        H = None
        if refit:
            if on_device:
                o0 = int(offsets[p])
                X, Y = pa_host[o0:o0 + sizes[p]].T, pb_host[o0:o0 + sizes[p]].T
            else:
                X, Y = datas[p]
            try:
                H = model.fit(X[:, inliers[0]], Y[:, inliers[0]], collective=True)
            except AssertionError:      # fewer inliers than a refit needs: RANSAC.run would raise here (ransac.py:38)
                H = None
        out.append((H, inliers, total))
    return out


def _match_features(trainImg, queryImg):
    """ORB + brute-force Hamming matcher of ransac.py:252-267.  This is OpenCV C++ and outside the
    GPU path; it runs only when cv2 is installed."""
    try:
        import cv2
    except ImportError as e:
        raise ImportError("stitching() needs OpenCV for ORB/BFMatcher, or pass matches=(ptsA, ptsB) "
                          "(float32 [N,2] each, the matchespoints.npy layout)") from e
    trainImg_gray = cv2.cvtColor(trainImg, cv2.COLOR_RGB2GRAY)
    queryImg_gray = cv2.cvtColor(queryImg, cv2.COLOR_RGB2GRAY)
    descriptor = cv2.ORB_create()
    kpsA, featuresA = descriptor.detectAndCompute(trainImg_gray, None)
    kpsB, featuresB = descriptor.detectAndCompute(queryImg_gray, None)
    bf = cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=True)
    found = sorted(bf.match(featuresA, featuresB), key=lambda m: m.distance)
    kpsA = np.float32([kp.pt for kp in kpsA])
    kpsB = np.float32([kp.pt for kp in kpsB])
    return np.float32([kpsA[m.queryIdx] for m in found]), np.float32([kpsB[m.trainIdx] for m in found])


def stitching(trainImg, queryImg, ransacMet="fwd", th=5, d=70, n=4, k=1000, blending=False, blendrate=0.2,
              mode=None, override=0, cylinderT=1, matches=None):
    """Panorama pipeline of ransac.py:235-283: matches -> RANSAC homography -> warp + composite.
    `matches=(ptsA, ptsB)` injects precomputed correspondences so the
    GPU path works without OpenCV; everything else keeps the reference's signature."""
    if override != 0:
        import cv2
        status, imgn = cv2.Stitcher_create().stitch([trainImg, queryImg])
        return imgn
    ptsA, ptsB = matches if matches is not None else _match_features(trainImg, queryImg)
    ptsA = np.asarray(ptsA, dtype=np.float32)
    ptsB = np.asarray(ptsB, dtype=np.float32)
    if mode is None:
        model = HomoModel(th=th, d=d, n=4)
        H, inliers, _len = RANSAC(model, k=k).run([ptsA.T, ptsB.T], method=ransacMet)
    else:
        import cv2
        H, status = cv2.findHomography(ptsA, ptsB, cv2.RANSAC, 4)
    return stitchPanorama(queryImg, trainImg, H=H, blending=blending, blendrate=blendrate)
