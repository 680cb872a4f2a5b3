// rwh_ransac_run: the host driver of RANSAC.run (ransac.py:159-213 without the final refit) in native code.
//
// What the reference does per call is a Python loop over k samples; what this library does is ONE search on the GPU
// (rwh_ransac_search: K1 + K2 + argmax) plus the "settle" step that makes the result exact with respect to the reference's
// solver: every sample K1 flags (repeated index, non-finite, ill-conditioned) and every hypothesis whose count is within a
// margin of a decision gets the reference's own H -- LAPACK dgesdd, rwh_host_dlt4_svd -- and is re-scored by K2; then the
// accept rules (first count >= need wins and stops, else the first maximum) are applied.  Rounds 2-3 drove this from
// Python (ransac._settle_on_host: still there for the batched and sharded forms, and as this function's twin in the CPU
// tests); here the same logic runs without the interpreter between its ~30 small steps: one upload, the repeated-index
// samples solved on host threads WHILE the GPU searches, one readback, usually one more solve / score / readback round.
//
// Buffers are the caller's (the function allocates nothing persistent): a device workspace and a page-locked host
// workspace, laid out by rwh_ransac_run_layout.  The function synchronises the stream (its results are host values).
#include <algorithm>
#include <climits>
#include <cstring>
#include <vector>
#ifdef RWH_RUN_STAMPS
#include <chrono>
#include <cstdio>
#endif
#include "rwh_common.h"

namespace {
// sections of the two workspaces (byte offsets, 256-byte aligned)
enum { D_PA, D_PB, D_IDX, D_H, D_COUNTS, D_FLAGS, D_MASKS, D_BEST, D_HSET, D_CNTSET, D_MASKSET, D_END,
       H_UP, H_COUNTS, H_FLAGS, H_CNTSET, H_HSET, H_MASK, H_ROWS, H_END,
       D_HINVSET, H_HINVSET,               // (round 3, appended: the settled hypotheses' inverses; both lie before D_END / H_END)
       D_ROWS, D_LO, D_HI, H_LO, H_HI, N_OFF };   // (round 4: candidate rows and their count intervals, rwh_score_interval)

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

void layout(int m, int k, long long* off) {
    const size_t words = (size_t)(m + 63) / 64, K = (size_t)(k > 0 ? k : 1), M = (size_t)m;
    size_t d = 0;
    off[D_PA] = (long long)d; d += 8 * M;                 // pts_a, pts_b, idx: ONE contiguous upload (8-byte / 16-byte aligned pieces)
    off[D_PB] = (long long)d; d += 8 * M;
    d = (d + 15) & ~(size_t)15;
    off[D_IDX] = (long long)d; d = up256(d + 16 * K);
    off[D_H] = (long long)d; d = up256(d + 36 * K);
    off[D_COUNTS] = (long long)d; d += 4 * K;             // counts then flags: ONE contiguous readback
    off[D_FLAGS] = (long long)d; d = up256(d + K);
    off[D_MASKS] = (long long)d; d = up256(d + 8 * words * K);
    off[D_BEST] = (long long)d; d = up256(d + 64);
    off[D_HSET] = (long long)d; d = up256(d + 36 * K);
    off[D_CNTSET] = (long long)d; d = up256(d + 4 * K);
    off[D_MASKSET] = (long long)d; d = up256(d + 8 * words * K);
    off[D_HINVSET] = (long long)d; d = up256(d + 36 * K);
    off[D_ROWS] = (long long)d; d = up256(d + 4 * K);
    off[D_LO] = (long long)d; d = up256(d + 4 * K);
    off[D_HI] = (long long)d; d = up256(d + 4 * K);
    off[D_END] = (long long)d;
    size_t h = 0;
    off[H_UP] = (long long)h; h = up256(h + (size_t)(off[D_IDX] - off[D_PA]) + 16 * K);
    off[H_COUNTS] = (long long)h; h += 4 * K;
    off[H_FLAGS] = (long long)h; h = up256(h + K);
    off[H_CNTSET] = (long long)h; h = up256(h + 4 * K);
    off[H_HSET] = (long long)h; h = up256(h + 36 * K);
    off[H_MASK] = (long long)h; h = up256(h + 8 * words);
    off[H_ROWS] = (long long)h; h = up256(h + 16 * K);
    off[H_HINVSET] = (long long)h; h = up256(h + 36 * K);
    off[H_LO] = (long long)h; h += 4 * K;                 // lo then hi: ONE contiguous readback
    off[H_HI] = (long long)h; h = up256(h + 4 * K);
    off[H_END] = (long long)h;
}

#ifdef RWH_RUN_STAMPS       // lab build: where the host time of one call goes (stderr)
struct Stamps {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
    void at(const char* what) {
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[run] %-28s +%7.1f us  (%7.1f)\n", what, std::chrono::duration<double, std::micro>(now - last).count(),
                std::chrono::duration<double, std::micro>(now - t0).count());
        last = now;
    }
};
#define STAMP(w) stamps.at(w)
#else
#define STAMP(w) ((void)0)
#endif

inline int margin_of(int best, int cap) { const int v = 3 + (best > 0 ? best : 0) / 16; return v < cap ? v : cap; }
}  // namespace

extern "C" int rwh_ransac_run_layout(int m, int k, long long* offsets, int n_offsets) {
    if (!offsets || m <= 0 || k < 0 || n_offsets < N_OFF) return RWH_E_INVALID;
    layout(m, k, offsets);
    return N_OFF;
}

// Settle step under 'fwd' (round 4): which hypotheses need the reference's own solver is decided by COUNT INTERVALS
// (rwh_score_interval), not by a flat margin.  Candidates -- every RWH_HYP_ILLCOND sample that is not degenerate, every other
// hypothesis within IV_NEAR counts of the best or of `need` -- get [lo, hi]; host-solved are the repeated-index / non-finite /
// degenerate samples (their H says nothing) and the candidates that are uncertain (lo < hi) AND able to matter (hi reaches the
// best lo, or `need`).  IV_DELTA0 / IV_DELTA1: the perturbation budgets, in units of an entry's natural scale, for unflagged /
// ill-conditioned samples: 16 and 64 float32 ulps (no containment failure at 8 ulps on 13 families x 20 000 samples;
// profiles/r04_lab_notes.txt).  'backward' / 'reproj' keep the margin rule of rounds 2-3 (an interval through the inverse is
// too wide to be useful).
static constexpr int IV_NEAR = 32;
static constexpr double IV_DELTA0 = 0x1p-20, IV_DELTA1 = 0x1p-18;

extern "C" int rwh_ransac_run(const float* pts_a, const float* pts_b, int m, const int32_t* idx, int k, double th, int loss,
                              int need, int margin_cap, void* dgesdd_ilp64, void* dgesv_ilp64, int threads, void* d_ws, void* h_ws,
                              int64_t hyp_base, int32_t* out, uint64_t* out_keys, uint64_t* out_mask, void* stream) {
    if (!pts_a || !pts_b || !idx || !d_ws || !h_ws || !out || !out_mask || !dgesdd_ilp64 || m <= 0 || k < 0) return RWH_E_INVALID;
    if (loss < RWH_LOSS_FWD || loss > RWH_LOSS_REPROJ || margin_cap < 0 || hyp_base < 0 || hyp_base + k > 0xFFFFFFFFll) return RWH_E_INVALID;
    for (long long i = 0; i < 4ll * k; ++i)
        if (idx[i] < 0 || idx[i] >= m) return RWH_E_INVALID;
    hipStream_t s = static_cast<hipStream_t>(stream);
#ifdef RWH_RUN_STAMPS
    Stamps stamps;
#endif
    STAMP("validate idx");
    long long off[N_OFF];
    layout(m, k, off);
    unsigned char* D = static_cast<unsigned char*>(d_ws);
    unsigned char* Hh = static_cast<unsigned char*>(h_ws);
    const int words = (m + 63) / 64;
    float* d_pa = reinterpret_cast<float*>(D + off[D_PA]);
    float* d_pb = reinterpret_cast<float*>(D + off[D_PB]);
    int32_t* d_idx = reinterpret_cast<int32_t*>(D + off[D_IDX]);
    float* d_H = reinterpret_cast<float*>(D + off[D_H]);
    int32_t* d_counts = reinterpret_cast<int32_t*>(D + off[D_COUNTS]);
    uint8_t* d_flags = D + off[D_FLAGS];
    uint64_t* d_masks = reinterpret_cast<uint64_t*>(D + off[D_MASKS]);
    uint64_t* d_best = reinterpret_cast<uint64_t*>(D + off[D_BEST]);
    float* d_hset = reinterpret_cast<float*>(D + off[D_HSET]);
    int32_t* d_cntset = reinterpret_cast<int32_t*>(D + off[D_CNTSET]);
    uint64_t* d_maskset = reinterpret_cast<uint64_t*>(D + off[D_MASKSET]);
    int32_t* h_counts = reinterpret_cast<int32_t*>(Hh + off[H_COUNTS]);
    uint8_t* h_flags = Hh + off[H_FLAGS];
    int32_t* h_cntset = reinterpret_cast<int32_t*>(Hh + off[H_CNTSET]);
    float* h_hset = reinterpret_cast<float*>(Hh + off[H_HSET]);
    uint64_t* h_mask = reinterpret_cast<uint64_t*>(Hh + off[H_MASK]);
    int32_t* h_rows = reinterpret_cast<int32_t*>(Hh + off[H_ROWS]);
    float* d_hinvset = reinterpret_cast<float*>(D + off[D_HINVSET]);
    float* h_hinvset = reinterpret_cast<float*>(Hh + off[H_HINVSET]);
    int32_t* d_rows = reinterpret_cast<int32_t*>(D + off[D_ROWS]);
    int32_t* d_lo = reinterpret_cast<int32_t*>(D + off[D_LO]);
    int32_t* d_hi = reinterpret_cast<int32_t*>(D + off[D_HI]);
    int32_t* h_lo = reinterpret_cast<int32_t*>(Hh + off[H_LO]);
    int32_t* h_hi = reinterpret_cast<int32_t*>(Hh + off[H_HI]);
    // 'backward' / 'reproj' project through numpy.linalg.inv(H): the settled hypotheses get THAT inverse (LAPACK dgesv, rwh_host_inv3),
    // not the kernel's own elimination, which rounds apart from it on nearly singular H (rwh.h, rwh_score_count_inv)
    const bool host_inv = loss != RWH_LOSS_FWD && dgesv_ilp64 != nullptr;

    for (int i = 0; i < 8; ++i) out[i] = 0;
    out[0] = -1;
    if (out_keys) { out_keys[0] = 0; out_keys[1] = 0; }
    for (int w = 0; w < words; ++w) out_mask[w] = 0;
    if (k == 0) return RWH_OK;

    // ---- one upload: correspondences + index table; the search goes out behind it ---------------------------------------
    const size_t up_bytes = (size_t)(off[D_IDX] - off[D_PA]) + 16 * (size_t)k;
    unsigned char* up = Hh + off[H_UP];
    memcpy(up + (off[D_PA] - off[D_PA]), pts_a, 8 * (size_t)m);
    memcpy(up + (off[D_PB] - off[D_PA]), pts_b, 8 * (size_t)m);
    memcpy(up + (off[D_IDX] - off[D_PA]), idx, 16 * (size_t)k);
    if (hipMemcpyAsync(D + off[D_PA], up, up_bytes, hipMemcpyHostToDevice, s) != hipSuccess) return RWH_E_LAUNCH;
    int st = rwh_ransac_search(d_pa, d_pb, m, d_idx, k, th, loss, need, 0, d_H, d_flags, d_counts, d_masks, d_best, 1, s);
    if (st != RWH_OK) return st;
    if (hipMemcpyAsync(h_counts, d_counts, 5 * (size_t)k, hipMemcpyDeviceToHost, s) != hipSuccess) return RWH_E_LAUNCH;   // counts + flags
    STAMP("staging copy + enqueue");

    // ---- hypotheses settled so far: position in the Hset / cnt_set / mask_set tables, -1 = not settled -------------------
    // (per-thread scratch that keeps its pages between calls: four fresh 400 KB vectors per call at k = 100 000 cost ~0.15 ms of page faults)
    static thread_local std::vector<int> pos_tl, cnt_tl, lo_tl, hi_tl, act_tl;
    std::vector<int>&pos = pos_tl, &cnt = cnt_tl, &lo = lo_tl, &hi = hi_tl, &act = act_tl;
    pos.assign((size_t)k, -1);
    cnt.resize((size_t)k);
    int nset = 0, rounds = 0;
    auto settle = [&](int n_rows) -> int {      // h_rows[0..n_rows) -> H by the reference's solver, counts + masks by K2
        if (n_rows == 0) return RWH_OK;
        std::vector<int32_t> samples(4 * (size_t)n_rows);
        for (int j = 0; j < n_rows; ++j) memcpy(&samples[4 * (size_t)j], idx + 4 * (size_t)h_rows[j], 16);
        int r = rwh_host_dlt4_svd(pts_a, pts_b, m, samples.data(), n_rows, dgesdd_ilp64, threads, h_hset + 9 * (size_t)nset);
        if (r != RWH_OK) return r;
        if (hipMemcpyAsync(d_hset + 9 * (size_t)nset, h_hset + 9 * (size_t)nset, 36 * (size_t)n_rows, hipMemcpyHostToDevice, s) != hipSuccess) return RWH_E_LAUNCH;
        if (host_inv) {
            r = rwh_host_inv3(h_hset + 9 * (size_t)nset, n_rows, dgesv_ilp64, h_hinvset + 9 * (size_t)nset);
            if (r != RWH_OK) return r;
            if (hipMemcpyAsync(d_hinvset + 9 * (size_t)nset, h_hinvset + 9 * (size_t)nset, 36 * (size_t)n_rows, hipMemcpyHostToDevice, s) != hipSuccess) return RWH_E_LAUNCH;
        }
        r = rwh_score_count_inv(d_hset + 9 * (size_t)nset, host_inv ? d_hinvset + 9 * (size_t)nset : nullptr, d_pa, d_pb, m, n_rows, th, loss,
                                INT_MAX, 0, d_cntset + nset, d_maskset + (size_t)nset * words, d_best + 4, nullptr, s);
        if (r != RWH_OK) return r;
        if (hipMemcpyAsync(h_cntset + nset, d_cntset + nset, 4 * (size_t)n_rows, hipMemcpyDeviceToHost, s) != hipSuccess) return RWH_E_LAUNCH;
        return RWH_OK;
    };
    auto absorb = [&](int n_rows) {             // after the stream has been synchronised
        for (int j = 0; j < n_rows; ++j) { pos[(size_t)h_rows[j]] = nset + j; cnt[(size_t)h_rows[j]] = h_cntset[nset + j]; }
        nset += n_rows;
    };

    // ---- while the GPU searches: the samples the host already knows K1 will flag (a repeated index) -----------------------
    int n_rep = 0;
    for (int i = 0; i < k; ++i) {
        const int32_t* q = idx + 4 * (size_t)i;
        if ((q[0] == q[1]) | (q[0] == q[2]) | (q[0] == q[3]) | (q[1] == q[2]) | (q[1] == q[3]) | (q[2] == q[3])) h_rows[n_rep++] = i;
    }
    STAMP("vectors + repeated scan");
    st = settle(n_rep);
    if (st != RWH_OK) { (void)hipStreamSynchronize(s); return st; }
    STAMP("repeated-index SVDs");
    if (hipStreamSynchronize(s) != hipSuccess) return RWH_E_LAUNCH;
    STAMP("sync 1");
    memcpy(cnt.data(), h_counts, 4 * (size_t)k);
    absorb(n_rep);
    int n_flagged = 0;
    for (int i = 0; i < k; ++i) n_flagged += h_flags[i] != 0;

    STAMP("counts copy + flag count");
    // ---- 'fwd': count intervals for the candidates (see the comment above the function) ------------------------------------
    double cscale = 1.0;
    for (long long i = 0; i < 2ll * m; ++i) { const double v = pts_a[i] < 0 ? -(double)pts_a[i] : (double)pts_a[i]; if (v > cscale) cscale = v; }
    const bool use_iv = loss == RWH_LOSS_FWD && cscale < 1e30;            // (an Inf / NaN coordinate: the margin rule)
    const unsigned always_bits = use_iv ? (RWH_HYP_REPEATED | RWH_HYP_SINGULAR | RWH_HYP_DEGENERATE) : 0xFFu;
    int n_iv = 0;
    if (use_iv) {
        lo.resize((size_t)k); hi.resize((size_t)k);
        memcpy(lo.data(), cnt.data(), 4 * (size_t)k);                      // not a candidate: its count is taken as it is
        memcpy(hi.data(), cnt.data(), 4 * (size_t)k);
        int best0 = 0;
        for (int i = 0; i < k; ++i) {
            const int c = (h_flags[i] & always_bits) ? 0 : cnt[(size_t)i];
            best0 = c > best0 ? c : best0;
        }
        const int near_lim = (best0 < need ? best0 : need) - IV_NEAR;      // cnt >= best0 - NEAR or cnt >= need - NEAR
        for (int i = 0; i < k; ++i) {
            const unsigned f = h_flags[i];
            if (!(f & always_bits) && ((f & RWH_HYP_ILLCOND) || cnt[(size_t)i] >= near_lim) && pos[(size_t)i] < 0) h_rows[n_iv++] = i;
        }
        STAMP("lo/hi init + candidates");
        if (n_iv) {
            if (hipMemcpyAsync(d_rows, h_rows, 4 * (size_t)n_iv, hipMemcpyHostToDevice, s) != hipSuccess) return RWH_E_LAUNCH;
            st = rwh_score_interval(d_H, d_rows, n_iv, d_flags, d_pa, d_pb, m, th, cscale, IV_DELTA0, IV_DELTA1, d_lo, d_hi, s);
            if (st != RWH_OK) { (void)hipStreamSynchronize(s); return st; }
            if (hipMemcpyAsync(h_lo, d_lo, 4 * (size_t)n_iv, hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipMemcpyAsync(h_hi, d_hi, 4 * (size_t)n_iv, hipMemcpyDeviceToHost, s) != hipSuccess) return RWH_E_LAUNCH;
            if (hipStreamSynchronize(s) != hipSuccess) return RWH_E_LAUNCH;
            std::vector<int32_t> rows_iv(h_rows, h_rows + n_iv);          // (h_rows is reused by the rounds below)
            for (int j = 0; j < n_iv; ++j) { lo[(size_t)rows_iv[(size_t)j]] = h_lo[j]; hi[(size_t)rows_iv[(size_t)j]] = h_hi[j]; }
        }
    }

    STAMP("interval kernel round trip");
    // ---- rounds: settle every hypothesis that can take part in the decision (ransac._settle_on_host, same rules) ----------
    int end = k;
    if (use_iv) {
        // The same rules as the general loop below, without three passes over all k hypotheses per round.  Only a hypothesis that is
        // not settled and either carries an always-bit or has an open interval (lo < hi) can ever be taken: the ACTIVE list, a few
        // thousand of 100 000.  Every other one is fixed for the whole call: v[] = what it contributes to the running best (lo; 0 for
        // an unsettled always-bit sample; the settled count once settled -- kept in lo[]), `fs` = the first hypothesis that
        // certainly exits.
        act.clear();
        int fs = k;
        for (int i = k - 1; i >= 0; --i) {
            const bool settled = pos[(size_t)i] >= 0, always = (h_flags[i] & always_bits) != 0;
            int v;
            if (settled) v = cnt[(size_t)i];
            else if (always) v = 0;
            else v = lo[(size_t)i];
            if ((settled || !always) && v >= need) fs = i;
            if (!settled && (always || lo[(size_t)i] < hi[(size_t)i])) act.push_back(i);
            lo[(size_t)i] = v;                                  // from here on lo[] is v[]
        }
        std::reverse(act.begin(), act.end());                   // ascending, like the general loop's h_rows
        for (;;) {
            end = fs < k ? fs + 1 : k;
            int best = 0;
            for (int i = 0; i < end; ++i) best = lo[(size_t)i] > best ? lo[(size_t)i] : best;
            int n_rows = 0;
            size_t keep = 0;
            for (size_t a = 0; a < act.size(); ++a) {
                const int i = act[a];
                const bool take = i < end && ((h_flags[i] & always_bits) || hi[(size_t)i] >= best || hi[(size_t)i] >= need);
                if (take) h_rows[n_rows++] = i; else act[keep++] = i;
            }
            act.resize(keep);
            STAMP("round scans");
            if (n_rows == 0) break;
            ++rounds;
            st = settle(n_rows);
            if (st != RWH_OK) { (void)hipStreamSynchronize(s); return st; }
            if (hipStreamSynchronize(s) != hipSuccess) return RWH_E_LAUNCH;
            const int first_new = nset;
            absorb(n_rows);
            for (int j = 0; j < n_rows; ++j) {
                const int i = h_rows[j], c = h_cntset[first_new + j];
                lo[(size_t)i] = c;
                if (c >= need && i < fs) fs = i;
            }
            STAMP("round settle + sync");
        }
    } else
    for (;;) {
        end = k;
        const int m_need = margin_of(need, margin_cap);
        for (int i = 0; i < k; ++i) {
            bool sure;
            if (pos[(size_t)i] >= 0) sure = cnt[(size_t)i] >= need;
            else if (use_iv) sure = !(h_flags[i] & always_bits) && lo[(size_t)i] >= need;
            else sure = h_flags[i] == 0 && cnt[(size_t)i] >= need + m_need;
            if (sure) { end = i + 1; break; }
        }
        int best = 0;                                       // a lower bound of the best count the reference sees in the prefix
        for (int i = 0; i < end; ++i) {
            int v = 0;
            if (pos[(size_t)i] >= 0) v = cnt[(size_t)i];
            else if (use_iv) v = (h_flags[i] & always_bits) ? 0 : lo[(size_t)i];
            else v = h_flags[i] == 0 ? cnt[(size_t)i] : 0;
            if (v > best) best = v;
        }
        const int m_best = margin_of(best, margin_cap);
        int n_rows = 0;
        for (int i = 0; i < end; ++i) {
            if (pos[(size_t)i] >= 0) continue;
            bool take;
            if (use_iv) take = (h_flags[i] & always_bits) || (lo[(size_t)i] < hi[(size_t)i] && (hi[(size_t)i] >= best || hi[(size_t)i] >= need));
            else take = h_flags[i] != 0 || cnt[(size_t)i] >= best - m_best || cnt[(size_t)i] >= need - m_need;
            if (take) h_rows[n_rows++] = i;
        }
        STAMP("round scans");
        if (n_rows == 0) break;
        ++rounds;
        st = settle(n_rows);
        if (st != RWH_OK) { (void)hipStreamSynchronize(s); return st; }
        if (hipStreamSynchronize(s) != hipSuccess) return RWH_E_LAUNCH;
        absorb(n_rows);
        STAMP("round settle + sync");
    }

    // ---- the accept rules (ransac.py:186-202) over the prefix the reference looks at ------------------------------------
    int winner = -1, early = 0;
    for (int i = 0; i < end; ++i)
        if (cnt[(size_t)i] >= need) { winner = i; early = 1; break; }
    if (winner < 0) {
        int bestc = 0;
        for (int i = 0; i < end; ++i)
            if (cnt[(size_t)i] > bestc) { bestc = cnt[(size_t)i]; winner = i; }       // strict >: the first index of the maximum
    }
    out[0] = winner; out[1] = early; out[2] = winner >= 0 ? cnt[(size_t)winner] : 0; out[3] = nset; out[4] = rounds; out[5] = n_flagged;
    out[6] = n_iv;
    if (out_keys && winner >= 0) {
        // this slice's packed keys, as K2b packs them (the payload of the one all-reduce of a sharded search): word 0 = count and
        // inverted GLOBAL index of the slice's winner, word 1 = inverted global index of its first hypothesis that reaches `need`
        const unsigned long long gi = (unsigned long long)hyp_base + (unsigned long long)winner;
        out_keys[0] = ((unsigned long long)(unsigned)cnt[(size_t)winner] << 32) | (0xFFFFFFFFull - gi);
        out_keys[1] = early ? (0xFFFFFFFFull - gi) : 0ull;
    }
    if (winner >= 0) {
        const uint64_t* src = pos[(size_t)winner] >= 0 ? d_maskset + (size_t)pos[(size_t)winner] * words : d_masks + (size_t)winner * words;
        if (hipMemcpyAsync(h_mask, src, 8 * (size_t)words, hipMemcpyDeviceToHost, s) != hipSuccess) return RWH_E_LAUNCH;
        if (hipStreamSynchronize(s) != hipSuccess) return RWH_E_LAUNCH;
        for (int w = 0; w < words; ++w) out_mask[w] = h_mask[w];
    }
    STAMP("accept rules + mask readback");
    // for the caller's diagnostics: the settled counts replace K2's in the host copy (the device copy keeps K2's own)
    for (int i = 0; i < k; ++i) h_cntset[i] = cnt[(size_t)i];      // (h_cntset is free again: every batch has been absorbed)
    STAMP("settled counts out");
    return RWH_OK;
}
