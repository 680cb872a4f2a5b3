"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the RANSAC-homography + backward-warp path.

Nothing in the product package (`ransac_with_homography_amd/`, the top-level
`homography.py` / `ransac.py` drop-in modules) may import from here.  Allowed
importers: `tests/`, `__graft_entry__.smoke()`, and the `cpu_baseline` leg of
`bench.py` (where it is the thing timed as "the numpy CPU path", never the
thing shipped).

Pinning: `oracle.rwh_oracle` is pinned bit-for-bit against outputs of the
reference's own `homography.py` / `ransac.py` run in the build container
(`tests/golden/make_golden.py` -> `tests/golden/*.npz`); see
`tests/test_oracle_golden.py`.
"""
