// Host-side helper of the RANSAC settle step (no device code in this file): the reference's own 4-point solve
//   calc_corresp (homography.py:4-14) -> numpy.linalg.svd -> v[-1] / v[-1][8]   (homography.py:71-88)
// for a batch of samples, on host threads.  The SVD is LAPACK's dgesdd -- THE SAME routine, taken by address from the
// OpenBLAS the caller's numpy links (ransac_with_homography_amd/_lapack.py finds the symbol), called with numpy's own
// arguments (jobz = 'A', column-major copy, workspace from a size query) -- so every H equals numpy's bit for bit
// (tests/test_settle_cpu.py); what this file adds is the loop in native code, off the interpreter lock, on several cores.
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include "rwh.h"

namespace {
typedef long long lint;   // ILP64 Fortran integer (numpy >= 2 bundles scipy-openblas64)
typedef void (*dgesdd_t)(const char* jobz, const lint* m, const lint* n, double* a, const lint* lda, double* s, double* u,
                         const lint* ldu, double* vt, const lint* ldvt, double* work, const lint* lwork, lint* iwork,
                         lint* info, size_t jobz_len);

void solve_range(dgesdd_t dgesdd, const float* pa, const float* pb, const int32_t* idx, int begin, int end, float* out, int* bad) {
    const lint M = 8, N = 9;
    double a[72], s[8], u[64], vt[81], wq = 0;
    lint iwork[64], info = 0, lwork = -1;
    dgesdd("A", &M, &N, a, &M, s, u, &M, vt, &N, &wq, &lwork, iwork, &info, 1);      // workspace size query, as numpy does
    lwork = wq > 1 ? (lint)wq : 1;
    std::vector<double> work((size_t)lwork);
    for (int t = begin; t < end; ++t) {
        float mat[8][9];
        for (int i = 0; i < 4; ++i) {
            const int id = idx[4 * t + i];
            const float x = pa[2 * id], y = pa[2 * id + 1], xp = pb[2 * id], yp = pb[2 * id + 1];
            float* r0 = mat[2 * i];
            float* r1 = mat[2 * i + 1];
            r0[0] = -x; r0[1] = -y; r0[2] = -1.f; r0[3] = 0.f; r0[4] = 0.f; r0[5] = 0.f;
            r0[6] = x * xp; r0[7] = y * xp; r0[8] = xp;                                 // float32 products (homography.py:6-13)
            r1[0] = 0.f; r1[1] = 0.f; r1[2] = 0.f; r1[3] = -x; r1[4] = -y; r1[5] = -1.f;
            r1[6] = x * yp; r1[7] = y * yp; r1[8] = yp;
        }
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 9; ++j) a[i + 8 * j] = (double)mat[i][j];               // column-major, float64 (numpy.linalg)
        dgesdd("A", &M, &N, a, &M, s, u, &M, vt, &N, work.data(), &lwork, iwork, &info, 1);
        if (info != 0) *bad = 1;                                                        // numpy would raise LinAlgError
        float h[9];
        for (int j = 0; j < 9; ++j) h[j] = (float)vt[8 + 9 * j];                        // last row of V^T, cast to float32
        const float h8 = h[8];
        for (int j = 0; j < 9; ++j) out[9 * t + j] = h[j] / h8;                         // float32 division (homography.py:87)
    }
}
}  // namespace

extern "C" int rwh_host_dlt4_svd(const float* pts_a, const float* pts_b, int m, const int32_t* idx_rows, int n,
                                 void* dgesdd_ilp64, int threads, float* out_h) {
    if (!pts_a || !pts_b || !idx_rows || !out_h || !dgesdd_ilp64 || m <= 0 || n < 0) return RWH_E_INVALID;
    for (long long i = 0; i < 4ll * n; ++i)
        if (idx_rows[i] < 0 || idx_rows[i] >= m) return RWH_E_INVALID;
    if (n == 0) return RWH_OK;
    dgesdd_t f = reinterpret_cast<dgesdd_t>(dgesdd_ilp64);
    int nt = threads < 1 ? 1 : threads;
    if (nt > (n + 11) / 12) nt = (n + 11) / 12;            // a thread is worth starting for a dozen samples (~0.1 ms of LAPACK)
    int bad = 0;
    if (nt <= 1) {
        solve_range(f, pts_a, pts_b, idx_rows, 0, n, out_h, &bad);
    } else {
        std::vector<std::thread> pool;
        std::vector<int> bads((size_t)nt, 0);
        for (int w = 1; w < nt; ++w)
            pool.emplace_back(solve_range, f, pts_a, pts_b, idx_rows, (int)((long long)n * w / nt), (int)((long long)n * (w + 1) / nt), out_h, &bads[(size_t)w]);
        solve_range(f, pts_a, pts_b, idx_rows, 0, (int)((long long)n / nt), out_h, &bads[0]);
        for (auto& th : pool) th.join();
        for (int b : bads) bad |= b;
    }
    return bad ? RWH_E_LAUNCH : RWH_OK;
}
