"""Drop-in module: `from homography import transformImage` (reference app.py:9) and
`from homography import calcHomographyLinear, calcHomography, stitchPanorama, cylindericlMap`
(reference ransac.py:3) resolve to the MI355X-backed implementations."""
from ransac_with_homography_amd.homography import *  # noqa: F401,F403
from ransac_with_homography_amd.homography import __all__  # noqa: F401
