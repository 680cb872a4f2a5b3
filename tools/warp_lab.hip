// Developer microbenchmark for the warp kernels (NOT part of the product, not built by build()).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -ffp-contract=off tools/warp_lab.hip -o tools/warp_lab
//   gpurun -- ./tools/warp_lab [frames] [case-filter]
// Times the RGB-u8 bilinear kernels on the BASELINE geometry (3840x2160 -> 2028x3771 auto-bounds grid and a
// 3840x2160 fixed grid) and reports how many output bytes differ from the generic (reference-order,
// texel-exact) kernel.
#include "../ransac_with_homography_amd/csrc/rwh_warp.hip"
#include "warp_pipe_experiment.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

using namespace rwh;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void fill_random(uint32_t* p, size_t n_words, uint32_t seed) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n_words; i += stride) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = x;
    }
}

__global__ void diff_count(const unsigned char* a, const unsigned char* b, size_t n, unsigned long long* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long d1 = 0, d2 = 0;
    for (; i < n; i += stride) { const int d = abs((int)a[i] - (int)b[i]); d1 += d == 1; d2 += d > 1; }
    if (d1) atomicAdd(out, d1);
    if (d2) atomicAdd(out + 1, d2);
}

// calibration: plain copies of the same volume (what the memory system delivers with no arithmetic)
__global__ __launch_bounds__(256) void copy16(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n) {
    size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}
__global__ __launch_bounds__(256) void copy12(const unsigned char* __restrict__ s, unsigned char* __restrict__ d, size_t n12) {
    size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i < n12) { pk3 v; __builtin_memcpy(&v, s + i * 12, 12); __builtin_memcpy(d + i * 12, &v, 12); }
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16;
    const char* filter = argc > 2 ? argv[2] : nullptr;
    const int SH = 2160, SW = 3840;
    const double H[9] = {1.02, 0.01, 5.0, 0.015, 0.98, 7.0, 1e-5, 2e-5, 1.0};
    double A[3][6], ih[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { A[i][j] = H[3 * i + j]; A[i][3 + j] = i == j; }
    for (int c = 0; c < 3; ++c) {
        int p = c; for (int i = c + 1; i < 3; ++i) if (fabs(A[i][c]) > fabs(A[p][c])) p = i;
        for (int j = 0; j < 6; ++j) { double t = A[c][j]; A[c][j] = A[p][j]; A[p][j] = t; }
        const double d = A[c][c]; for (int j = 0; j < 6; ++j) A[c][j] /= d;
        for (int i = 0; i < 3; ++i) if (i != c) { const double f = A[i][c]; for (int j = 0; j < 6; ++j) A[i][j] -= f * A[c][j]; }
    }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) ih[3 * i + j] = A[i][3 + j];
    const size_t src_bytes = (size_t)SH * SW * 3;
    unsigned char *src, *dst, *ref;
    CK(hipMalloc(&src, src_bytes * B + 64));
    CK(hipMalloc(&dst, (size_t)3840 * 2160 * 3 * B + 64));
    CK(hipMalloc(&ref, (size_t)3840 * 2160 * 3 * B + 64));
    fill_random<<<4096, 256>>>(reinterpret_cast<uint32_t*>(src), src_bytes * B / 4, 1234u);
    CK(hipDeviceSynchronize());
    unsigned long long* d_diff; CK(hipMalloc(&d_diff, 16));
    hipEvent_t e0s, e1s; CK(hipEventCreate(&e0s)); CK(hipEventCreate(&e1s));
    {
        const size_t nbytes = (size_t)2028 * 3771 * 3 * B;   // copy as many bytes as one launch writes (reads the same amount)
        for (int which = 0; which < 2; ++which) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0s));
                if (which == 0) copy16<<<(unsigned)((nbytes / 16 + 255) / 256), 256>>>((const uint4*)src, (uint4*)dst, nbytes / 16);
                else copy12<<<(unsigned)((nbytes / 12 + 255) / 256), 256>>>(src, dst, nbytes / 12);
                CK(hipEventRecord(e1s)); CK(hipEventSynchronize(e1s));
                float ms; CK(hipEventElapsedTime(&ms, e0s, e1s)); best = ms < best ? ms : best;
            }
            printf("copy %s: %zu MB read + same written: %.3f ms = %.1f GB/s\n", which ? "12 B/lane" : "16 B/lane", nbytes >> 20, best, 2.0 * nbytes / best / 1e6);
        }
    }
    struct Geo { int x0, y0, w, h; const char* name; } geo[2] = {{5, 7, 3771, 2028, "auto 3771x2028"}, {0, 0, 3840, 2160, "fixed 3840x2160"}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int g = 0; g < 2; ++g) {
        const Geo G = geo[g];
        auto call = [&](int interp, int dst_dtype, bool generic) {
            // the C ABI picks the fast kernel; `generic` forces the generic one by going through launch()
            if (!generic)
                return rwh_warp_backward(src, SH, SW, 3, RWH_U8, (int64_t)src_bytes, B, ih, 1, G.x0, 1.0, G.x0 + G.w - 1, G.y0, 1.0,
                                         G.y0 + G.h - 1, G.h, G.w, SH, SW, interp, dst, dst_dtype, (int64_t)G.w * G.h * 3, 0, G.h, 0, nullptr);
            WarpArgs a;
            a.src = src; a.dst = dst; a.src_img_stride = (long long)src_bytes; a.dst_img_stride = (long long)G.w * G.h * 3;
            for (int i = 0; i < 9; ++i) a.ih[i] = ih[i];
            a.x0 = G.x0; a.step_x = 1.0; a.x_last = G.x0 + G.w - 1; a.y0 = G.y0; a.step_y = 1.0; a.y_last = G.y0 + G.h - 1;
            a.src_h = SH; a.src_w = SW; a.bound_h = SH; a.bound_w = SW; a.out_h = G.h; a.out_w = G.w; a.row_begin = 0; a.rows = G.h;
            a.tiles_x = (G.w + 255) / 256; a.tiles_y = (G.h + 3) / 4; a.nblocks = a.tiles_x * a.tiles_y * B; a.cpx = (a.nblocks + 7) / 8;
            return launch(warp_generic<unsigned char, 3, unsigned char, RWH_BILINEAR>, a, (hipStream_t)0);
        };
        auto fast = [&](int variant, int group) {
            WarpArgs a;
            a.src = src; a.dst = dst; a.src_img_stride = (long long)src_bytes; a.dst_img_stride = (long long)G.w * G.h * 3;
            a.src_h = SH; a.src_w = SW; a.bound_h = SH; a.bound_w = SW; a.out_h = G.h; a.out_w = G.w; a.row_begin = 0; a.rows = G.h;
            return launch_fast(a, ih, G.x0, 1.0, G.y0, 1.0, RWH_U8, B, (hipStream_t)0, variant, group,
                               variant == 2 ? warp_rgb8_pipe<unsigned char> : nullptr);
        };
        struct Case { const char* name; std::function<int()> run; };
        std::vector<Case> cases = {
            {"generic u8 (reference order)", [&] { return call(RWH_BILINEAR, RWH_U8, true); }},
            {"fast u8 (C ABI default)", [&] { return call(RWH_BILINEAR, RWH_U8, false); }},
            {"exact u8 (float64, RWH_WARP_EXACT)", [&] {
                return rwh_warp_backward(src, SH, SW, 3, RWH_U8, (int64_t)src_bytes, B, ih, 1, G.x0, 1.0, G.x0 + G.w - 1, G.y0, 1.0,
                                         G.y0 + G.h - 1, G.h, G.w, SH, SW, RWH_BILINEAR, dst, RWH_U8, (int64_t)G.w * G.h * 3, 0, G.h, RWH_WARP_EXACT, nullptr); }},
            {"fast u8 px4", [&] { return fast(0, 1); }},
            {"fast u8 px8", [&] { return fast(1, 1); }},
            {"fast u8 pipe G=4", [&] { return fast(2, 4); }},
            {"fast u8 pipe G=8", [&] { return fast(2, 8); }},
            {"fast u8 pipe G=16", [&] { return fast(2, 16); }},
        };
        const double bytes = (double)B * (src_bytes + (double)G.w * G.h * 3);
        const size_t nbytes = (size_t)G.w * G.h * 3 * B;
        printf("== geometry %s, %d frames, %.1f MB algorithmic ==\n", G.name, B, bytes / 1e6);
        for (auto& c : cases) {
            if (filter && &c != &cases[0] && !strstr(c.name, filter)) continue;
            CK(hipMemset(dst, 0, nbytes));
            for (int i = 0; i < 2; ++i) if (c.run() != 0) { printf("launch failed\n"); return 1; }
            CK(hipEventRecord(e0));
            const int reps = 10;
            for (int i = 0; i < reps; ++i) c.run();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            if (&c == &cases[0]) CK(hipMemcpy(ref, dst, nbytes, hipMemcpyDeviceToDevice));
            CK(hipMemset(d_diff, 0, 16));
            diff_count<<<2048, 256>>>(dst, ref, nbytes, d_diff);
            unsigned long long hd[2]; CK(hipMemcpy(hd, d_diff, 16, hipMemcpyDeviceToHost));
            printf("%-30s %8.3f ms  %7.1f GB/s  %6.2f us/frame  diff1=%llu diff>1=%llu of %zu\n", c.name, ms, bytes / ms / 1e6, ms * 1e3 / B, hd[0], hd[1], nbytes);
        }
    }
    return 0;
}
