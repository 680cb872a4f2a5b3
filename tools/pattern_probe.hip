// Developer microbenchmark (NOT part of the product): what do the warp's MEMORY patterns cost without its arithmetic?
// 32 x 4K RGB u8 frames -> 3771 x 2028 outputs (the bench geometry); every output tile loads its source footprint and stores its
// pixels; loaded data goes through LDS and is xor-folded into what is stored, so nothing is optimised away.
//   load patterns   W : one window per WAVE  (64 x 8 patch: 12 rows x <= 18 chunks of 12 B, 3 rows per instruction)        = the product kernel
//                   B : one window per BLOCK (128 x 16 tile: 19 rows x 34 chunks of 12 B over 256 threads)
//                   B16: the block window with 16-byte loads (26 per row)
//                   W2: one window per wave, 128 x 8 patches (2 waves per tile): 11 rows x 34 chunks
//   store patterns  S : 2 x 12 B per lane, 96-byte row segments (the product kernel)   T : the tile's rows as 16-byte pieces (24 lanes x 16 B per row)
//   hipcc -O3 --offload-arch=gfx950 tools/pattern_probe.hip -o tools/pattern_probe && gpurun -- ./tools/pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
struct __attribute__((packed)) pk3 { unsigned a, b, c; };
struct __attribute__((packed)) pk4 { unsigned a, b, c, d; };
constexpr int SW = 3840, SH = 2160, OW = 3771, OH = 2028, TX = (OW + 127) / 128, TY = (OH + 15) / 16;
struct Args { const unsigned char* src; unsigned char* dst; long long sstride, dstride; unsigned nblocks, cpx; int frames; };

template <int LOADS, int STORES>     // LOADS: 0 none, 1 W, 2 B, 3 B16, 4 W2;  STORES: 0 none, 1 S, 2 T
__global__ __launch_bounds__(256) void pattern(const Args a) {
    __shared__ __attribute__((aligned(16))) unsigned lds[20 * 140 + 64];
    const unsigned b = blockIdx.x;
    const unsigned logical = (b & 7u) * a.cpx + (b >> 3);
    if (logical >= a.nblocks) return;
    const unsigned img = logical / (TX * TY), t = logical % (TX * TY), ty = t / TX, tx = t % TX;
    const int tcol = min((int)tx * 128, OW - 128), trow = min((int)ty * 16, OH - 16);
    const unsigned char* simg = a.src + (long long)img * a.sstride;
    unsigned char* dimg = a.dst + (long long)img * a.dstride;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned pitch = SW * 3;
    unsigned acc = 0;
    if (LOADS == 1) {           // per-wave window: the wave's 64 x 8 patch (2 x 2 in the tile)
        const int px = tcol + (wave & 1) * 64, py = trow + (wave >> 1) * 8;
        const int x0 = ((int)(px * 1.0183f) + 3) & ~3, y0 = (int)(py * 1.0651f) + 5;
        const int srow = lane / 21, scol = lane - 21 * srow;
        const bool on = (srow < 3) & (scol < 18);
        pk3 v[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) { v[p] = pk3{0, 0, 0}; if (on) __builtin_memcpy(&v[p], simg + (size_t)(min(y0 + 3 * p + srow, SH - 1)) * pitch + (size_t)(x0 + 4 * scol) * 3, 12); }
#pragma unroll
        for (int p = 0; p < 4; ++p) if (on) { lds[wave * 700 + (3 * p + srow) * 54 + 3 * scol] = v[p].a; lds[wave * 700 + (3 * p + srow) * 54 + 3 * scol + 1] = v[p].b; lds[wave * 700 + (3 * p + srow) * 54 + 3 * scol + 2] = v[p].c; }
    }
    if (LOADS == 2) {           // per-block window: 19 rows x 34 chunks of 12 B
        const int x0 = ((int)(tcol * 1.0183f) + 3) & ~3, y0 = (int)(trow * 1.0651f) + 5;
        pk3 v[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int c = tid + 256 * p, r = c / 34, k = c - 34 * r;
            v[p] = pk3{0, 0, 0};
            if (c < 19 * 34) __builtin_memcpy(&v[p], simg + (size_t)(min(y0 + r, SH - 1)) * pitch + (size_t)(x0 + 4 * k) * 3, 12);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) { const int c = tid + 256 * p; if (c < 19 * 34) { lds[3 * c] = v[p].a; lds[3 * c + 1] = v[p].b; lds[3 * c + 2] = v[p].c; } }
    }
    if (LOADS == 3) {           // per-block window, 16-byte loads: 19 rows x 26 pieces
        const int x0 = ((int)(tcol * 1.0183f) + 3) & ~3, y0 = (int)(trow * 1.0651f) + 5;
        pk4 v[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int c = tid + 256 * p, r = c / 26, k = c - 26 * r;
            v[p] = pk4{0, 0, 0, 0};
            if (c < 19 * 26) __builtin_memcpy(&v[p], simg + (size_t)(min(y0 + r, SH - 1)) * pitch + (size_t)x0 * 3 + 16 * k, 16);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) { const int c = tid + 256 * p; if (c < 19 * 26) { lds[4 * c] = v[p].a; lds[4 * c + 1] = v[p].b; lds[4 * c + 2] = v[p].c; lds[4 * c + 3] = v[p].d; } }
    }
    if (LOADS == 4) {           // per-wave window of a 128 x 8 patch (waves 0, 1 of the tile; waves 2, 3 idle here: half the waves, same bytes)
        if (wave < 2) {
            const int py = trow + wave * 8;
            const int x0 = ((int)(tcol * 1.0183f) + 3) & ~3, y0 = (int)(py * 1.0651f) + 5;
            pk3 v[6];
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                const int c = lane + 64 * p, r = c / 34, k = c - 34 * r;
                v[p] = pk3{0, 0, 0};
                if (c < 11 * 34) __builtin_memcpy(&v[p], simg + (size_t)(min(y0 + r, SH - 1)) * pitch + (size_t)(x0 + 4 * k) * 3, 12);
            }
#pragma unroll
            for (int p = 0; p < 6; ++p) { const int c = lane + 64 * p; if (c < 11 * 34) { lds[wave * 1400 + 3 * c] = v[p].a; lds[wave * 1400 + 3 * c + 1] = v[p].b; lds[wave * 1400 + 3 * c + 2] = v[p].c; } }
        }
    }
    __syncthreads();
    acc = lds[(tid * 7) % 2048] ^ lds[(tid * 13 + 5) % 2048];
    if (STORES == 1) {          // the product kernel's stores: lane = 4 px (12 B), runs 32 px apart, patch rows 8 lanes wide
        const int px = tcol + (wave & 1) * 64, py = trow + (wave >> 1) * 8, prow = lane >> 3, pq = lane & 7;
        unsigned char* d = dimg + ((size_t)(py + prow) * OW + px + 4 * pq) * 3;
        pk3 w = {acc, acc * 3u, acc * 5u};
        __builtin_memcpy(d, &w, 12);
        w.a ^= 0x55u;
        __builtin_memcpy(d + 96, &w, 12);
    } else if (STORES == 2) {   // the tile's 16 rows x 384 B as 16-byte pieces: 24 per row, 384 pieces over 256 threads
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int c = tid + 256 * p, r = c / 24, k = c - 24 * r;
            if (c < 16 * 24) { pk4 w = {acc, acc * 3u, acc * 5u, acc * 7u}; __builtin_memcpy(dimg + ((size_t)(trow + r) * OW + tcol) * 3 + 16 * k, &w, 16); }
        }
    } else if (STORES == 3) {   // float32 output, the product's re-dealt layout: per run, lane l of a patch row stores pieces l, 8 + l, 16 + l of the row's 384 B
        const int px = tcol + (wave & 1) * 64, py = trow + (wave >> 1) * 8, prow = lane >> 3, pq = lane & 7;
        unsigned char* d = dimg + ((size_t)(py + prow) * OW + px) * 12;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int v = 0; v < 3; ++v) { pk4 w = {acc + h, acc * 3u, acc * 5u + v, acc * 7u}; __builtin_memcpy(d + 384 * h + 16 * (8 * v + pq), &w, 16); }
    } else if (STORES == 4) {   // float32 output as whole tile rows: 16 rows x 1536 B = 1536 pieces of 16 B, 6 per thread, 96 consecutive lanes per row
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            const int c = tid + 256 * p, r = c / 96, k = c - 96 * r;
            pk4 w = {acc + p, acc * 3u, acc * 5u, acc * 7u};
            __builtin_memcpy(dimg + ((size_t)(trow + r) * OW + tcol) * 12 + 16 * k, &w, 16);
        }
    } else if (acc == 0x12345678u) dimg[tid] = 1;
}

__global__ __launch_bounds__(256) void copy16(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

int main() {
    const int F = 32;
    const size_t sb = (size_t)SW * SH * 3, db = (size_t)OW * OH * 12;     // (room for float32 outputs; the uint8 patterns use a quarter)
    unsigned char *src, *dst;
    CK(hipMalloc(&src, sb * F + 4096)); CK(hipMalloc(&dst, db * F + 4096));
    { std::vector<unsigned> h(1 << 22); unsigned s = 1; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
      for (size_t o = 0; o < sb * F; o += h.size() * 4) CK(hipMemcpy(src + o, h.data(), std::min(h.size() * 4, sb * F - o), hipMemcpyHostToDevice)); }
    Args a{src, dst, (long long)sb, (long long)db, (unsigned)(TX * TY * F), (unsigned)((TX * TY * F + 7) / 8), F};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char* name; void (*k)(const Args); double rd, wr; };
    const Case cases[] = {
        {"loads W   (per-wave 64x8 windows: product)", pattern<1, 0>, 1, 0}, {"loads B   (per-block 128x16 window, 12 B)", pattern<2, 0>, 1, 0},
        {"loads B16 (per-block window, 16 B)", pattern<3, 0>, 1, 0}, {"loads W2  (per-wave 128x8 windows)", pattern<4, 0>, 1, 0},
        {"stores S  (12 B, 96-byte segments: product)", pattern<0, 1>, 0, 1}, {"stores T  (16 B pieces of 384-byte tile rows)", pattern<0, 2>, 0, 1},
        {"stores F  (float32 out, product's re-dealt 16 B)", pattern<0, 3>, 0, 4}, {"stores FT (float32 out, whole tile rows)", pattern<0, 4>, 0, 4},
        {"W + F (float32-output product pattern)", pattern<1, 3>, 1, 4}, {"W + FT", pattern<1, 4>, 1, 4}, {"B + FT", pattern<2, 4>, 1, 4},
        {"W + S (product pattern)", pattern<1, 1>, 1, 1}, {"B + S", pattern<2, 1>, 1, 1}, {"B + T", pattern<2, 2>, 1, 1}, {"B16 + T", pattern<3, 2>, 1, 1}, {"W2 + S", pattern<4, 1>, 1, 1},
    };
    for (int rep = 0; rep < 2; ++rep)
        for (const auto& c : cases) {
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(c.k, dim3(8 * a.cpx), dim3(256), 0, 0, a);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            const int N = 100;
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(c.k, dim3(8 * a.cpx), dim3(256), 0, 0, a);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= N;
            const double bytes = F * (c.rd * sb + c.wr * (double)OW * OH * 3);
            printf("%-48s %.4f ms per 32 frames = %5.2f us/frame  %6.0f GB/s algorithmic\n", c.name, ms, ms * 1e3 / F, bytes / ms / 1e6);
        }
    const size_t n16 = (sb * F) / 16;
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(copy16, dim3(8192), dim3(256), 0, 0, (const uint4*)src, (uint4*)dst, std::min(n16, db * F / 16));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(copy16, dim3(8192), dim3(256), 0, 0, (const uint4*)src, (uint4*)dst, std::min(n16, db * F / 16));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 50;
    printf("%-48s %.4f ms = %6.0f GB/s (read + write)\n", "plain copy, 16 B per lane", ms, 2.0 * std::min(n16 * 16, db * F) / ms / 1e6);
    return 0;
}
