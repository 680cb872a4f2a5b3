/* Plain-C consumer of include/rwh.h: proves the boundary is a C ABI (no C++ types, no torch).
 * Built and run by tests/test_cabi_cpu.py with gcc; needs no GPU (argument validation only). */
#include <stdio.h>
#include <string.h>
#include "rwh.h"

int main(void) {
    double ih[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (rwh_abi_version() != RWH_ABI_VERSION) { printf("abi mismatch\n"); return 1; }
    if (strcmp(rwh_strerror(RWH_E_UNSUPPORTED), "unsupported dtype/channel/size combination") != 0) return 2;
    /* NULL device pointers must be rejected before anything touches a device */
    if (rwh_warp_backward(NULL, 8, 8, 3, RWH_U8, 192, 1, ih, 1, 0, 1, 7, 0, 1, 7, 8, 8, 8, 8, RWH_BILINEAR,
                          NULL, RWH_U8, 192, 0, 8, RWH_WARP_ZERO_ORIGIN, NULL) != RWH_E_INVALID) return 3;
    /* an empty row tile is a no-op even with NULL buffers (a rank that owns no rows) */
    if (rwh_warp_backward(NULL, 8, 8, 3, RWH_U8, 192, 1, ih, 1, 0, 1, 7, 0, 1, 7, 8, 8, 8, 8, RWH_BILINEAR,
                          NULL, RWH_U8, 192, 4, 4, 0, NULL) != RWH_OK) return 4;
    if (rwh_dlt4_batched(NULL, NULL, 4, NULL, 1, NULL, NULL, NULL) != RWH_E_INVALID) return 5;
    if (rwh_score_count(NULL, NULL, NULL, 4, 1, 5.0, RWH_LOSS_FWD, 3, 0, NULL, NULL, NULL, NULL, NULL) != RWH_E_INVALID) return 6;
    if (rwh_ransac_search(NULL, NULL, 4, NULL, 1, 5.0, RWH_LOSS_FWD, 3, 0, NULL, NULL, NULL, NULL, NULL, 0, NULL) != RWH_E_INVALID) return 7;
    if (rwh_project_points(NULL, NULL, 4, 0, NULL, NULL) != RWH_E_INVALID) return 8;
    if (rwh_stitch_panorama(NULL, 8, 8, NULL, 8, 8, ih, 0, 0, 8, 8, 0, 0, 0, 0, 8, 8, 0, 0.2, NULL, 0, NULL) != RWH_E_INVALID) return 9;
    printf("cabi ok\n");
    return 0;
}
