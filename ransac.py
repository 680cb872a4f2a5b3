"""Drop-in module: `from ransac import stitching` (reference app.py:10) resolves to the
MI355X-backed implementation; `HomoModel`, `RANSAC`, `Model` likewise."""
from ransac_with_homography_amd.ransac import (DEBUG, LVL, HomoModel, Model, RANSAC, stitching, run_batch, DeviceProblems,  # noqa: F401
                                               calcHomography, calcHomographyLinear, cylindericlMap,
                                               stitchPanorama)
