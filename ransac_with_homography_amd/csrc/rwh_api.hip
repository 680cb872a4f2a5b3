// Version / error-string entry points of librwh_hip.so (see include/rwh.h).
#include "rwh_common.h"

extern "C" int rwh_abi_version(void) { return RWH_ABI_VERSION; }

extern "C" const char* rwh_strerror(int code) {
    switch (code) {
        case RWH_OK: return "ok";
        case RWH_E_INVALID: return "invalid argument";
        case RWH_E_UNSUPPORTED: return "unsupported dtype/channel/size combination";
        case RWH_E_LAUNCH: return "HIP launch failure";
        default: return "unknown error";
    }
}
