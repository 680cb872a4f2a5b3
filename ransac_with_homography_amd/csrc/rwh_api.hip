// Version / error-string entry points of librwh_hip.so (see include/rwh.h).
#include "rwh_common.h"

namespace rwh {
// Lab overrides (rwh_lab_tune): 0 = the library's own choice.  Process-wide, set by tests / tools only.
int g_force_warp_shape = 0;
int g_force_score_hpw = 0;
int g_score_exact_only = 0;
int g_force_warp_frames = 0;
}  // namespace rwh

extern "C" int rwh_abi_version(void) { return RWH_ABI_VERSION; }

extern "C" int rwh_lab_tune(int knob, int value) {
    if (knob == RWH_TUNE_WARP_SHAPE && (value == 0 || (value >= 5 && value <= 7) || value == 13 || value == 14)) { rwh::g_force_warp_shape = value; return RWH_OK; }
    if (knob == RWH_TUNE_SCORE_HPW && value >= 0 && value <= 64) { rwh::g_force_score_hpw = value; return RWH_OK; }
    if (knob == RWH_TUNE_SCORE_EXACT && (value == 0 || value == 1)) { rwh::g_score_exact_only = value; return RWH_OK; }
    if (knob == RWH_TUNE_WARP_FRAMES && ((value >= 0 && value <= 64) || (value >= 102 && value <= 164))) { rwh::g_force_warp_frames = value; return RWH_OK; }
    return RWH_E_INVALID;
}

namespace rwh {
// One wave that sits on a CU for `ticks` of the 100 MHz constant clock and reports how many shader-clock cycles went by
// meanwhile: d_out[0] = delta s_memtime (shader cycles), d_out[1] = delta s_memrealtime (100 MHz ticks).
// Launched on a side stream next to the kernels being timed it measures the clock the chip HOLDS under that load
// (MI355X lowers its clock under sustained load; bench.py reports roofline.sclk_mhz from it).
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, unsigned long long ticks) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r = r0;
    // bounded: every iteration sleeps ~0.6 us at most, and the loop ends after `ticks` (the host caps it at 2 s)
    for (unsigned i = 0; i < (1u << 24) && r - r0 < ticks; ++i) {
        __builtin_amdgcn_s_sleep(100);
        r = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r - r0; }
}
}  // namespace rwh

extern "C" int rwh_lab_clock_probe(uint64_t* d_out, double milliseconds, void* stream) {
    if (!d_out || !(milliseconds > 0.0) || milliseconds > 2000.0) return RWH_E_INVALID;
    hipLaunchKernelGGL(rwh::clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<unsigned long long*>(d_out), (unsigned long long)(milliseconds * 1e5));
    return rwh::check_launch();
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
