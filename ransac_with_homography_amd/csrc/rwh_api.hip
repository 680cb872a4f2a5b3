// Version / error-string entry points of librwh_hip.so (see include/rwh.h).
#include "rwh_common.h"

namespace rwh {
// Lab overrides (rwh_lab_tune): 0 = the library's own choice.  Process-wide, set by tests / tools only.
int g_force_warp_shape = 0;
int g_force_score_hpw = 0;
int g_score_exact_only = 0;
}  // namespace rwh

extern "C" int rwh_abi_version(void) { return RWH_ABI_VERSION; }

extern "C" int rwh_lab_tune(int knob, int value) {
    if (knob == RWH_TUNE_WARP_SHAPE && (value == 0 || (value >= 5 && value <= 7))) { rwh::g_force_warp_shape = value; return RWH_OK; }
    if (knob == RWH_TUNE_SCORE_HPW && value >= 0 && value <= 64) { rwh::g_force_score_hpw = value; return RWH_OK; }
    if (knob == RWH_TUNE_SCORE_EXACT && (value == 0 || value == 1)) { rwh::g_score_exact_only = value; return RWH_OK; }
    return RWH_E_INVALID;
}

extern "C" const char* rwh_strerror(int code) {
    switch (code) {
        case RWH_OK: return "ok";
        case RWH_E_INVALID: return "invalid argument";
        case RWH_E_UNSUPPORTED: return "unsupported dtype/channel/size combination";
        case RWH_E_LAUNCH: return "HIP launch failure";
        default: return "unknown error";
    }
}
