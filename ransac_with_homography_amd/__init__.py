"""MI355X-native RANSAC-homography + backward-warp path.

Drop-in for the hot path of choice17/ransac_with_homography: import
`ransac_with_homography_amd.homography` / `.ransac` (or the top-level `homography`
/ `ransac` shim modules of this repository) where the reference's modules were
imported.  The compute lives in librwh_hip.so (hand-written HIP for gfx950, C ABI
in include/rwh.h); this package is the host-side mirror of the reference's API.
"""
from ._lib import RwhError, RwhUnavailable, LIB_PATH  # noqa: F401

__version__ = "0.1.0"
