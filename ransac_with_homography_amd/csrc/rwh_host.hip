// Host-side helper of the RANSAC settle step (no device code in this file): the reference's own 4-point solve
//   calc_corresp (homography.py:4-14) -> numpy.linalg.svd -> v[-1] / v[-1][8]   (homography.py:71-88)
// for a batch of samples, on host threads.  The SVD is LAPACK's dgesdd -- THE SAME routine, taken by address from the
// OpenBLAS the caller's numpy links (ransac_with_homography_amd/_lapack.py finds the symbol), called with numpy's own
// arguments (jobz = 'A', column-major copy, workspace from a size query) -- so every H equals numpy's bit for bit
// (tests/test_settle_cpu.py); what this file adds is the loop in native code, off the interpreter lock, on several cores.
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <atomic>
#include <thread>
#include <unistd.h>
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

// A small pool of host worker threads, started on first use and kept for the life of the process (the only state this
// library keeps): starting a thread costs ~15-20 us, as much as two SVDs, and RANSAC.run asks for two batches of a few dozen
// per call.  Chunks of a batch are handed out under a mutex; the caller works on chunks too and returns when all are done.
class HostPool {
  public:
    // never destroyed: workers may outlive main().  A forked child gets a pool of its own on first use: the parent's worker
    // threads do not exist in it (and its mutexes may have been held at the fork).  Creation is serialised (two threads on the
    // very first call used to be able to create two pools, one of them leaked with its workers).
    static HostPool& get() {
        static std::atomic<HostPool*> cur{nullptr};
        static std::atomic<long> owner{0};
        static std::atomic<long> maker{0};          // pid of the thread creating a pool, 0 = nobody (not a std::mutex: one held
        const long me = (long)getpid();             //  across a fork would stay locked in the child, whose holder does not exist)
        HostPool* p = cur.load(std::memory_order_acquire);
        if (p && owner.load(std::memory_order_acquire) == me) return *p;
        for (;;) {
            long holder = 0;
            if (maker.compare_exchange_weak(holder, me, std::memory_order_acquire)) break;
            if (holder != 0 && holder != me && maker.compare_exchange_weak(holder, me, std::memory_order_acquire)) break;   // the parent's, at a fork
            std::this_thread::yield();
        }
        p = cur.load(std::memory_order_acquire);
        if (!p || owner.load(std::memory_order_acquire) != me) {
            p = new HostPool;
            cur.store(p, std::memory_order_release);
            owner.store(me, std::memory_order_release);
        }
        maker.store(0, std::memory_order_release);
        return *p;
    }
    // run job(chunk) for chunk = 0 .. n_chunks-1 on up to `threads` threads (the caller included)
    void run(int n_chunks, int threads, const std::function<void(int)>& job) {
        std::lock_guard<std::mutex> serial(run_mu_);                            // one batch at a time
        {
            std::unique_lock<std::mutex> lk(mu_);
            const int want = threads - 1 < n_chunks - 1 ? threads - 1 : n_chunks - 1;
            while ((int)n_workers_ < want) { std::thread([this] { loop(); }).detach(); ++n_workers_; }
            job_ = &job; n_chunks_ = n_chunks; next_ = 0; done_ = 0; limit_ = want; active_ = 0;
        }
        cv_work_.notify_all();
        for (;;) {
            int c;
            { std::lock_guard<std::mutex> lk(mu_); if (next_ >= n_chunks_) break; c = next_++; }
            job(c);
            { std::lock_guard<std::mutex> lk(mu_); ++done_; }
        }
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [this] { return done_ == n_chunks_; });
        job_ = nullptr;
    }

  private:
    void loop() {
        for (;;) {
            int c;
            const std::function<void(int)>* job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [this] { return job_ && next_ < n_chunks_ && active_ < limit_; });
                c = next_++; job = job_; ++active_;
            }
            for (;;) {
                (*job)(c);
                std::lock_guard<std::mutex> lk(mu_);
                ++done_;
                if (next_ < n_chunks_) { c = next_++; continue; }
                --active_;
                if (done_ == n_chunks_) cv_done_.notify_all();
                break;
            }
        }
    }
    std::mutex mu_, run_mu_;
    std::condition_variable cv_work_, cv_done_;
    const std::function<void(int)>* job_ = nullptr;
    int n_chunks_ = 0, next_ = 0, done_ = 0, limit_ = 0, active_ = 0;
    size_t n_workers_ = 0;
};

extern "C" int rwh_host_dlt4_svd(const float* pts_a, const float* pts_b, int m, const int32_t* idx_rows, int n,
                                 void* dgesdd_ilp64, int threads, float* out_h) {
    if (!pts_a || !pts_b || !idx_rows || !out_h || !dgesdd_ilp64 || m <= 0 || n < 0) return RWH_E_INVALID;
    for (long long i = 0; i < 4ll * n; ++i)
        if (idx_rows[i] < 0 || idx_rows[i] >= m) return RWH_E_INVALID;
    if (n == 0) return RWH_OK;
    dgesdd_t f = reinterpret_cast<dgesdd_t>(dgesdd_ilp64);
    int nt = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
    // samples per hand-out: ~50 us of LAPACK for small batches, larger pieces (fewer trips through the pool's mutex) for big ones
    int CHUNK = n / (8 * nt);
    CHUNK = CHUNK < 6 ? 6 : (CHUNK > 64 ? 64 : CHUNK);
    const int n_chunks = (n + CHUNK - 1) / CHUNK;
    if (nt <= 1 || n_chunks <= 1) {
        int bad = 0;
        solve_range(f, pts_a, pts_b, idx_rows, 0, n, out_h, &bad);
        return bad ? RWH_E_LAUNCH : RWH_OK;
    }
    std::vector<int> bads((size_t)n_chunks, 0);
    const std::function<void(int)> job = [&](int c) {
        const int b = c * CHUNK, e = b + CHUNK < n ? b + CHUNK : n;
        solve_range(f, pts_a, pts_b, idx_rows, b, e, out_h, &bads[(size_t)c]);
    };
    HostPool::get().run(n_chunks, nt, job);
    int bad = 0;
    for (int b : bads) bad |= b;
    return bad ? RWH_E_LAUNCH : RWH_OK;
}

// n x (numpy.linalg.inv of a float32 3 x 3, ransac.py:74): float64 dgesv on the identity -- the routine and the library
// numpy.linalg.inv itself calls -- cast back to float32.  A singular matrix (numpy raises LinAlgError) gives NaNs here.
extern "C" int rwh_host_inv3(const float* h, int n, void* dgesv_ilp64, float* out) {
    if (!h || !out || !dgesv_ilp64 || n < 0) return RWH_E_INVALID;
    typedef void (*dgesv_t)(const int64_t*, const int64_t*, double*, const int64_t*, int64_t*, double*, const int64_t*, int64_t*);
    dgesv_t f = reinterpret_cast<dgesv_t>(dgesv_ilp64);
    const int64_t N = 3;
    for (int t = 0; t < n; ++t) {
        double a[9], b[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        int64_t ipiv[3], info = 0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) a[i + 3 * j] = (double)h[9 * (size_t)t + 3 * i + j];      // column-major, float64 (numpy.linalg)
        f(&N, &N, a, &N, ipiv, b, &N, &info);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) out[9 * (size_t)t + 3 * i + j] = info == 0 ? (float)b[i + 3 * j] : __builtin_nanf("");
    }
    return RWH_OK;
}


// ---- the sample table of RANSAC.run, drawn exactly as numpy's legacy generator draws it (ransac.py:177) ---------------------
// k successive np.random.randint(0, m, n) calls consume the global RandomState (MT19937) like ONE randint(0, m, (k, n)) call does
// (SURVEY A.4).  For a range below 2^32 numpy's legacy randint takes one 32-bit output per draw, masks it with the smallest
// 2^b - 1 >= m - 1 and rejects values above m - 1 (numpy/random/src/distributions: random_bounded_uint64_fill with
// use_masked -> buffered_bounded_masked_uint32); m == 1 consumes nothing.  numpy's own loop costs ~5 ns per draw behind the
// interpreter and the generator's lock -- 2.1 ms of a 5.3 ms RANSAC.run at k = 100 000 --; this one is the same arithmetic in
// a tight loop.  `key` / `pos`: RandomState.get_state()[1:3], updated in place for set_state().  out32 (k x n, the kernel's
// index table) and / or out64 (the int64 array numpy would have returned) may be NULL.
namespace {
inline void mt19937_refill(uint32_t* mt) {
    constexpr int N = 624, M = 397;
    constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, A = 0x9908b0dfu;
    int kk = 0;
    for (; kk < N - M; ++kk) { const uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER); mt[kk] = mt[kk + M] ^ (y >> 1) ^ ((y & 1u) ? A : 0u); }
    for (; kk < N - 1; ++kk) { const uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER); mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? A : 0u); }
    const uint32_t y = (mt[N - 1] & UPPER) | (mt[0] & LOWER);
    mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
}
}  // namespace

extern "C" int rwh_host_legacy_randint(uint32_t* key, int32_t* pos, int64_t m, int64_t count, int32_t* out32, int64_t* out64) {
    if (!key || !pos || m <= 0 || m > 0x7fffffffll || count < 0 || *pos < 0 || *pos > 624) return RWH_E_INVALID;
    const uint32_t rng = (uint32_t)(m - 1);
    if (rng == 0) {                                   // numpy: "rng == 0: out = off", no draw
        for (int64_t i = 0; i < count; ++i) { if (out32) out32[i] = 0; if (out64) out64[i] = 0; }
        return RWH_OK;
    }
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    int p = *pos;
    int64_t i = 0;
    // whole blocks of the state while they cannot over-run the table (every raw output yields at most one draw): tempering and
    // a branch-free compaction -- the rejection branch (28 % taken at m = 185) is what makes the obvious loop 7 ns per draw
    while (count - i >= 624) {
        if (p == 624) { mt19937_refill(key); p = 0; }
        uint32_t t[624];
        const int nraw = 624 - p;
        for (int q = 0; q < nraw; ++q) {
            uint32_t y = key[p + q];
            y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
            t[q] = y & mask;
        }
        p = 624;
        if (out32 && out64) {
            int64_t j = i;
            for (int q = 0; q < nraw; ++q) { out32[j] = (int32_t)t[q]; out64[j] = (int64_t)t[q]; j += (t[q] <= rng); }
            i = j;
        } else if (out32) {
            int64_t j = i;
            for (int q = 0; q < nraw; ++q) { out32[j] = (int32_t)t[q]; j += (t[q] <= rng); }
            i = j;
        } else if (out64) {
            int64_t j = i;
            for (int q = 0; q < nraw; ++q) { out64[j] = (int64_t)t[q]; j += (t[q] <= rng); }
            i = j;
        } else {
            for (int q = 0; q < nraw; ++q) i += (t[q] <= rng);
        }
    }
    for (; i < count; ++i) {                              // the tail, draw by draw
        uint32_t v;
        do {
            if (p == 624) { mt19937_refill(key); p = 0; }
            uint32_t y = key[p++];
            y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
            v = y & mask;
        } while (v > rng);
        if (out32) out32[i] = (int32_t)v;
        if (out64) out64[i] = (int64_t)v;
    }
    *pos = p;
    return RWH_OK;
}
