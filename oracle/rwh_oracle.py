"""CPU oracle (numpy restatement) of the reference's RANSAC-homography + backward-warp path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  This is *not* the product
and the product never imports it.

Every function cites the reference lines it restates (paths are relative to
the read-only reference checkout, `choice17/ransac_with_homography`).  The
restatement performs the same numpy operations in the same order and dtypes so
that its results are bit-identical to the reference's when both run on the same
numpy/BLAS build; `tests/test_oracle_golden.py` pins that against fixtures the
reference itself produced (`tests/golden/make_golden.py`).

Layout conventions are the reference's:
  * 4-point / N-point solvers take `u`, `v` as N x 2 arrays (points in rows);
  * the RANSAC model takes features x observations arrays (2 x M or 3 x M);
  * warps take H x W x C images and return (image, min_x, min_y).
"""
import numpy as np

# --------------------------------------------------------------------------
# Design matrices  (reference homography.py:4-69)
# --------------------------------------------------------------------------

def dlt_matrix(u, v):
    """2N x 9 DLT matrix, float32.

    Rows per pair (x,y)->(x',y'):  [-x,-y,-1, 0,0,0, x*x', y*x', x']
                                   [ 0, 0, 0,-x,-y,-1, x*y', y*y', y']
    Products are formed in the dtype of the inputs (float32 inputs -> float32
    rounded products) and the whole matrix is then stored as float32.
    4-point form: homography.py:4-14; N-point form: homography.py:30-46
    (identical values, so one builder serves both).
    """
    n = u.shape[0]
    a = np.zeros((n, 18), dtype=np.float32)
    x, y = u[:, 0], u[:, 1]
    xp, yp = v[:, 0], v[:, 1]
    a[:, 0] = -x
    a[:, 1] = -y
    a[:, 2] = -1
    a[:, 6] = x * xp
    a[:, 7] = y * xp
    a[:, 8] = xp
    a[:, 12] = -x
    a[:, 13] = -y
    a[:, 14] = -1
    a[:, 15] = x * yp
    a[:, 16] = y * yp
    a[:, 17] = yp
    return a.reshape(2 * n, 9)


def linear_system(u, v):
    """2N x 8 matrix A and 2N x 1 right-hand side b (h33 == 1), float32.

    Rows per pair: [x,y,1,0,0,0,-x*x',-y*x'] -> x' ; [0,0,0,x,y,1,-x*y',-y*y'] -> y'.
    homography.py:16-28 (4-point) and homography.py:48-69 (N-point).
    """
    n = u.shape[0]
    a = np.zeros((n, 16), dtype=np.float32)
    b = np.zeros((n, 2), dtype=np.float32)
    x, y = u[:, 0], u[:, 1]
    xp, yp = v[:, 0], v[:, 1]
    a[:, 0] = x
    a[:, 1] = y
    a[:, 2] = 1
    a[:, 6] = -x * xp
    a[:, 7] = -y * xp
    a[:, 11] = x
    a[:, 12] = y
    a[:, 13] = 1
    a[:, 14] = -x * yp
    a[:, 15] = -y * yp
    b[:, 0] = xp
    b[:, 1] = yp
    return a.reshape(2 * n, 8), b.reshape(2 * n, 1)


# --------------------------------------------------------------------------
# Solvers  (reference homography.py:71-105)
# --------------------------------------------------------------------------

def calc_homography(u, v, collective=False):
    """DLT: last right-singular vector of the 2N x 9 matrix, divided by its 9th
    element.  numpy.linalg.svd computes in float64 and casts the factors back to
    the float32 of its input, so the result is float32.  homography.py:71-88.
    Non-collective: the 4-point builder reads rows 0..3 only (homography.py:4-14) -- an n-point sample with n > 4
    (HomoModel(n=6)) is fitted on its first four pairs, and n < 4 raises IndexError like `u[3,0]` does there;
    collective: all N pairs (homography.py:30-46).
    """
    if not collective:
        if u.shape[0] < 4:
            raise IndexError("index 3 is out of bounds for axis 0 with size %d" % u.shape[0])
        u, v = u[:4], v[:4]
    mat = dlt_matrix(u, v)
    _, _, vt = np.linalg.svd(mat)
    h = vt[-1].reshape(3, 3)
    return h / h.item(8)


def calc_homography_linear(u, v, collective=False):
    """Normal equations with h33 == 1: inv(A^T A) @ (A^T b), all products in
    float32 (the inverse is computed in float64 inside numpy and cast back);
    the 3x3 is assembled from Python floats -> float64.  homography.py:90-105.
    """
    A, b = linear_system(u, v)
    h = np.linalg.inv(A.T @ A) @ (A.T @ b)
    return np.array([[h.item(0), h.item(1), h.item(2)],
                     [h.item(3), h.item(4), h.item(5)],
                     [h.item(6), h.item(7), 1]])


# --------------------------------------------------------------------------
# Interpolators  (reference homography.py:108-140)
# --------------------------------------------------------------------------

def _blank_origin_texel(img):
    """The reference zeroes texel (0,0) of the CALLER's image so that masked
    coordinates, redirected to (0,0), sample black.  homography.py:110-116 and
    126-130: channels 0..2 always, channel 3 only when there are exactly 4."""
    img[0, 0, 0] = 0
    img[0, 0, 1] = 0
    img[0, 0, 2] = 0
    if img.shape[2] == 4:
        img[0, 0, 3] = 0


def nearest_neighbor(z_t, img, h, w, mh, mw):
    """homography.py:108-121.  trunc(coord + 0.5) as int32, mask on the integer
    coordinates, gather.  Output dtype == image dtype."""
    zi = (z_t + 0.5).astype(np.int32).T
    chn = img.shape[2]
    _blank_origin_texel(img)
    outside = (zi[:, 0] > w - 1) | (zi[:, 0] < 0) | (zi[:, 1] > h - 1) | (zi[:, 1] < 0)
    zi[outside, 0:2] = 0
    return img[zi[:, 1], zi[:, 0], :].reshape(mh, mw, chn)


def bilinear(z_t, img, h, w, mh, mw):
    """homography.py:123-138.  Mask on the float coordinates, truncate, lerp in
    x then in y in float64.  `z_t` (3 x N float64) is modified in place, like
    the reference does through its transposed view."""
    z = z_t.T
    chn = img.shape[2]
    _blank_origin_texel(img)
    outside = (z[:, 0] > w - 1) | (z[:, 0] < 0) | (z[:, 1] > h - 1) | (z[:, 1] < 0)
    z[outside, 0:2] = 0
    zi = z.astype(np.int32)
    fr = z - zi
    fx = fr[:, 0:1]
    fy = fr[:, 1:2]
    xi = zi[:, 0]
    yi = zi[:, 1]
    top = img[yi, xi, :] * (1 - fx) + img[yi, xi + 1, :] * fx
    bot = img[yi + 1, xi, :] * (1 - fx) + img[yi + 1, xi + 1, :] * fx
    out = top * (1 - fy) + bot * fy
    return out.reshape(mh, mw, chn)


INTERPOLATORS = {'nn': nearest_neighbor, 'bilinear': bilinear}


# --------------------------------------------------------------------------
# Backward warps  (reference homography.py:142-242)
# --------------------------------------------------------------------------

def output_bounds(h, w, H, boundary=0):
    """Output bounding box of an h x w image under H.  homography.py:143-163:
    corners (0,0),(w-1,0),(w-1,h-1),(0,h-1) -> H -> dehomogenise -> int()
    truncation of the min / max; `boundary` truthy clamps the minima at 0."""
    corners = np.array([[0, w - 1, w - 1, 0],
                        [0, 0, h - 1, h - 1],
                        [1., 1, 1, 1]])
    p = H @ corners
    p /= p[-1, :]
    max_x = int(np.max(p[0, :]))
    min_x = int(np.min(p[0, :]))
    max_y = int(np.max(p[1, :]))
    min_y = int(np.min(p[1, :]))
    if boundary:
        min_x = max(min_x, 0)
        min_y = max(min_y, 0)
    return min_x, min_y, max_x - min_x + 1, max_y - min_y + 1


def _source_coords(H, x0, x1, nx, y0, y1, ny):
    """Grid of output coordinates -> inv(H) -> divide by the third row, all in
    float64.  homography.py:166-174 / 197-205."""
    xs = np.linspace(x0, x1, nx)
    ys = np.linspace(y0, y1, ny)
    xv, yv = np.meshgrid(xs, ys)
    z = np.dstack([xv, yv, np.ones((ny, nx))]).reshape([nx * ny, 3]).T
    z_t = np.linalg.inv(H) @ z
    z_t /= z_t[-1, :]
    return z_t


def wrap_perspective(img, H, convert='nn', boundary=0):
    """homography.py:142-184 with crop=True (the only live form).  Returns
    (warped, min_x, min_y); bilinear output is float64, nn keeps the dtype."""
    h, w, _ = img.shape
    min_x, min_y, max_w, max_h = output_bounds(h, w, H, boundary)
    z_t = _source_coords(H, min_x, min_x + max_w - 1, max_w, min_y, min_y + max_h - 1, max_h)
    out = INTERPOLATORS[convert](z_t, img, h, w, max_h, max_w)
    return out, min_x, min_y


def wrap_perspective_scan(img, H, res, convert='nn'):
    """homography.py:186-209.  Fixed output resolution res=(h,w); the grid is
    linspace(0,w,w) x linspace(0,h,h) (step w/(w-1)) and the bounds test uses
    `res`, not the source size."""
    h, w = res
    z_t = _source_coords(H, 0, w, w, 0, h, h)
    out = INTERPOLATORS[convert](z_t, img, h, w, h, w)
    return out, 0, 0


def transform_image(img, u, v, box=None, method='bilinear'):
    """homography.py:211-228: 4-point linear solve, warp (auto-bounds or scan),
    truncate to uint8, crop to the destination quad's corner 0 .. corner 2."""
    H = calc_homography_linear(u.T[:, :2], v.T[:, :2])
    if box is None:
        imgn, mx, my = wrap_perspective(img, H, convert=method)
    else:
        imgn, mx, my = wrap_perspective_scan(img, H, box, convert=method)
    imgn = imgn.astype(np.uint8)
    sx = int(v[0, 0] - mx)
    sy = int(v[1, 0] - my)
    ex = int(v[0, 2] - mx)
    ey = int(v[1, 2] - my)
    return imgn[sy:ey + 1, sx:ex + 1, :]


def transform_image_h(img, H, method='bilinear'):
    """homography.py:230-242: auto-bounds warp; uint8 truncation only for
    3-channel results (4-channel results stay float64)."""
    imgn, mx, my = wrap_perspective(img, H, convert=method, boundary=0)
    if imgn.shape[2] == 3:
        return imgn.astype(np.uint8), mx, my
    return imgn, mx, my


# --------------------------------------------------------------------------
# Alpha + panorama compositor  (reference homography.py:250-338)
# --------------------------------------------------------------------------

def add_alpha_rate(img, rate=0.2):
    """homography.py:250-258, method 'Rate', alphaOnly False: float32 H x W x (C+1)
    with a constant alpha plane of rate + 1e-10."""
    h, w, c = img.shape
    rate = rate + 1e-10
    out = np.zeros((h, w, c + 1), dtype=np.float32)
    out[:, :, :c] = img
    out[:, :, c] = rate
    return out


def add_alpha_zero(img):
    """homography.py:250-267 with a `method` that is neither 'Rate' nor 'Gradient' (stitchPanorama passes its `blending`
    argument through, homography.py:298: e.g. True): the alpha plane is never written and stays 0."""
    h, w, c = img.shape
    out = np.zeros((h, w, c + 1), dtype=np.float32)
    out[:, :, :c] = img
    return out


def add_alpha_gradient(img):
    """homography.py:250-266, method 'Gradient', direction LEFT, alphaOnly False:
    the alpha plane is the float64 ramp (x + y) / (w + h) * 0.5 stored as float32."""
    h, w, c = img.shape
    out = np.zeros((h, w, c + 1), dtype=np.float32)
    out[:, :, :c] = img
    xx, yy = np.meshgrid(np.linspace(0, w - 1, w), np.linspace(0, h - 1, h))
    out[:, :, c] = (xx + yy) / (w + h) * 0.5
    return out


def stitch_geometry(wt, ht, wq, hq, mx, my):
    """Canvas geometry of homography.py:303-321.  Returns the inclusive paste
    rectangles (tsx,tsy,tex,tey) for the warped image, (qsx,qsy,qex,qey) for
    the query image, and the canvas size (fw, fh)."""
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
    fw = max(tex + 1, qex + 1)
    fh = max(tey + 1, qey + 1)
    return (tsx, tsy, tex, tey), (qsx, qsy, qex, qey), (fw, fh)


def stitch_panorama(imgQ, imgT, H, blending=False, blendrate=0.2):
    """homography.py:288-338.  Warp imgT by H (always bilinear), place it and
    imgQ on a common canvas; 'Rate' / 'Gradient' blending alpha-averages the
    overlap in float32, otherwise imgQ is pasted over the warped imgT."""
    if blending == 'Gradient':
        imgT = add_alpha_gradient(imgT)
    elif blending == 'Rate':
        imgT = add_alpha_rate(imgT, blendrate)
    elif blending:
        imgT = add_alpha_zero(imgT)
    img_t, mx, my = transform_image_h(imgT, H)
    ht, wt, ct = img_t.shape
    hq, wq, _ = imgQ.shape
    (tsx, tsy, tex, tey), (qsx, qsy, qex, qey), (fw, fh) = stitch_geometry(wt, ht, wq, hq, mx, my)
    if blending:
        can = np.zeros((fh, fw, ct), dtype=np.float32)
        can[qsy:qey + 1, qsx:qex + 1, :3] = imgQ[:, :, :3].astype(np.float32)
        can[:, :, 3] += 1e-10
        if blending == 'Rate':
            can[qsy:qey + 1, qsx:qex + 1, 3] = 1 + 1e-10 - blendrate
        else:
            can[qsy:qey + 1, qsx:qex + 1, 3] = 1
        win = can[tsy:tey + 1, tsx:tex + 1]
        base = win[:, :, 3:4] + img_t[:, :, 3:4]
        can[tsy:tey + 1, tsx:tex + 1, :3] = \
            (win[:, :, 3:4] / base) * win[:, :, :3] + (img_t[:, :, 3:4] / base) * img_t[:, :, :3]
        return can[:, :, :3].astype(np.uint8)
    can = np.zeros((fh, fw, ct), dtype=np.uint8)
    can[tsy:tey + 1, tsx:tex + 1, :] = img_t
    can[qsy:qey + 1, qsx:qex + 1, :] = imgQ
    return can


# --------------------------------------------------------------------------
# Homography model: projection, distance, loss  (reference ransac.py:22-98)
# --------------------------------------------------------------------------

def _homogeneous(P):
    """ransac.py:58-62 / 69-73: 2 x M -> float32 3 x M with a row of ones;
    3 x M is used as given."""
    if P.shape[0] == 2:
        x = np.ones((3, P.shape[1]), dtype=np.float32)
        x[:2, :] = P
        return x
    return P


def project_fwd(val, X):
    """ransac.py:55-64: y = val @ [X;1], divided by (y[2] + 1e-10)."""
    y = val @ _homogeneous(X)
    return y / (y[-1, :] + 1e-10)


def project_back(val, Y):
    """ransac.py:66-76: same through inv(val)."""
    inv = np.linalg.inv(val)
    x = inv @ _homogeneous(Y)
    return x / (x[-1, :] + 1e-10)


def l2_dist(pred, true):
    """ransac.py:78-82."""
    d = pred - true
    return np.sqrt(np.sum(d * d, axis=0))


def compute_loss(val, X, Y, method="reproj"):
    """ransac.py:84-98: 'fwd' | 'backward' | 'reproj' (sum of both)."""
    if method == "fwd":
        return l2_dist(project_fwd(val, X)[:2, :], Y)
    if method == "backward":
        return l2_dist(project_back(val, Y)[:2, :], X)
    if method == "reproj":
        e = l2_dist(project_fwd(val, X)[:2, :], Y)
        e += l2_dist(project_back(val, Y)[:2, :], X)
        return e
    raise SystemExit("Invalid method!")


def fit_minimal(X, Y):
    """HomoModel.fit, non-collective (ransac.py:52): DLT on the sampled columns."""
    return calc_homography(X.T[:, :2], Y.T[:, :2], False)


def fit_all(X, Y):
    """HomoModel.fit, collective (ransac.py:50): N-point normal equations."""
    return calc_homography_linear(X.T[:, :2], Y.T[:, :2], True)


# --------------------------------------------------------------------------
# RANSAC driver  (reference ransac.py:159-213)
# --------------------------------------------------------------------------

def ransac_run(X, Y, th=5, d=50, n=4, k=1000, method="reproj"):
    """Sequential driver, consuming the global legacy numpy RNG exactly like the
    reference (one randint(0, M, n) per iteration, ransac.py:177).

    Early exit when count >= M*d/100 + n (ransac.py:169,186-190), otherwise the
    running best with a strict '>' (first index wins ties, ransac.py:199-202);
    final N-point refit on the winner's inliers (ransac.py:206-211).
    Returns (H_refit float64 3x3, (inlier_indices,), count, winner_iteration).
    """
    M = X.shape[1]
    assert M == Y.shape[1], "data observation not consistent!"
    need = M * d / 100 + n
    best = 0
    best_mask = None
    best_it = -1
    for it in range(k):
        idx = np.random.randint(0, M, n)
        val = fit_minimal(X[:, idx], Y[:, idx])
        mask = compute_loss(val, X, Y, method) < th
        cnt = np.sum(mask)
        if cnt >= need:
            best, best_mask, best_it = cnt, mask, it
            break
        if cnt > best:
            best, best_mask, best_it = cnt, mask, it
    if k <= 0:          # ransac.py:203 reads `lenalsoIninears`, which only the loop body assigns
        raise UnboundLocalError("local variable 'lenalsoIninears' referenced before assignment")
    inliers = np.where(best_mask)
    # HomoModel.fit's size check (ransac.py:38): exactly n columns, or more than n with collective=True -- a winner with
    # fewer inliers than the sample size n (possible for n > 4: the model is fitted on four points) fails here
    m = inliers[0].size
    assert m == n or m > n, "invalid data size should be %d" % n
    Hf = fit_all(X[:, inliers[0]], Y[:, inliers[0]])
    return Hf, inliers, best, best_it


def ransac_table(X, Y, idx_table, th=5, method="fwd"):
    """Batch form used to check the GPU kernels hypothesis by hypothesis:
    for every row of `idx_table` (K x 4 sample indices) return the per-hypothesis
    H (K x 9 float32) and inlier count (K int32).  Same arithmetic as one loop
    iteration of `ransac_run`."""
    K = idx_table.shape[0]
    Hs = np.empty((K, 9), dtype=np.float32)
    counts = np.empty(K, dtype=np.int32)
    for i in range(K):
        idx = idx_table[i]
        val = fit_minimal(X[:, idx], Y[:, idx])
        Hs[i] = val.reshape(9)
        counts[i] = np.sum(compute_loss(val, X, Y, method) < th)
    return Hs, counts


def select_winner(counts, need):
    """Batch equivalent of the driver's accept rules: the first
    index with count >= need if any, else the lowest index of the maximum.
    Returns (index, early_exit_flag)."""
    hit = np.nonzero(counts >= need)[0]
    if hit.size:
        return int(hit[0]), True
    return int(np.argmax(counts)), False
