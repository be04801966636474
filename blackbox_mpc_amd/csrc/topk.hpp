// Exact sorted top-k of a workgroup's LDS-resident rewards: tf.nn.top_k(sorted=True) semantics
// (cem.py:97-99) -- larger first, ties -> lower index first.
//
// MSD radix select on order-preserving 32-bit keys: <= 4 passes of 8-bit digits taken below the bits
// all keys share, each pass one LDS histogram (ds_add_u32) + one 256-bin scan by a single wave, i.e.
// O(N) work per pass instead of the O(N^2) comparison count of ranking by counting.  That pins the
// k-th key exactly; the <= k winners are then compacted and ranked among themselves (k^2 comparisons
// on 64-bit (key,index) words).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bbmpc {

constexpr int TOPK_HIST_WORDS = 528;   // histogram A (256 bins) + 16 control words + histogram B (256 bins)

// smaller key == better (larger reward); equal rewards <=> equal keys (-0 is folded onto +0 first)
__device__ __forceinline__ uint32_t reward_key(float r) {
    const uint32_t u = __float_as_uint(r + 0.0f);
    const uint32_t asc = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
    return ~asc;
}

// inclusive prefix sum over the 64 lanes: DPP row_shr scan inside each 16-lane row, row totals via readlane
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8
    const int t0 = __builtin_amdgcn_readlane(x, 15), t1 = __builtin_amdgcn_readlane(x, 31),
              t2 = __builtin_amdgcn_readlane(x, 47);
    const int row = lane >> 4;
    x += (row >= 1 ? t0 : 0) + (row >= 2 ? t1 : 0) + (row >= 3 ? t2 : 0);
    return (uint32_t)x;
}

// vals[N] in LDS (left untouched); eidx[k] out; hist[TOPK_HIST_WORDS] and ekeys[k] are LDS scratch.
// All threads of the workgroup must call (k <= nthr); ends with a barrier (eidx visible to everyone).
// Barrier budget: 1 (key range) + 2 per radix pass (usually 2 passes) + 1 (compaction) + 2 (ranking).
#ifdef BBMPC_KERNEL_DBG
__shared__ long long g_topk_dbg[16];
#define TOPK_DBG(i) do { if (tid == 0) g_topk_dbg[(i)] = (long long)wall_clock64(); } while (0)
#else
#define TOPK_DBG(i) do {} while (0)
#endif

struct TopkSel {
    uint32_t T;          // decided high bits of the k-th key
    uint32_t remaining;  // how many keys of the boundary bucket are winners
    uint32_t eq_total;   // how many keys the boundary bucket holds
    int top;             // number of low bits left undecided (early exit)
};

// First part of the selection, up to (not including) its first barrier: every thread reads only the elements
// n = tid, tid + nthr, ... -- a caller whose threads have just WRITTEN exactly those elements (the persistent kernel's
// rollout) runs it in front of the barrier it needs anyway and then calls block_topk_select(..., prepass_done = true).
__device__ __forceinline__ void block_topk_prepass(const float* vals, int N, int k, uint32_t* hist, int tid, int nthr) {
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t* hist2 = hist + 272;                // second histogram: zeroed while the other one is scanned
    // ---- key range: digits are taken from the highest bit where the smallest key and an UPPER BOUND B of the
    // k-th smallest key differ.  Rewards of one population mostly share sign + exponent, so a fixed top-8-bit
    // digit would send a whole wave's ds_add_u32 to one or two bins (serialised); and the worst rewards are far
    // outliers, so [min, max] would still squeeze everything that matters into a few bins.  Bound: every 16-lane
    // row contributes its two smallest keys (distinct elements); if that makes >= k elements, the largest of them
    // is >= the k-th smallest overall.  Keys that do not share the common prefix of (min, B) are > B and never
    // enter a histogram.
    uint32_t k1 = 0xFFFFFFFFu, kmaxl = 0u;           // this lane's smallest key (one element per lane) / largest
    for (int n = tid; n < N; n += nthr) {
        const uint32_t key = reward_key(vals[n]);
        k1 = min(k1, key);
        kmaxl = max(kmaxl, key);
    }
    uint32_t k2 = 0xFFFFFFFFu;                       // (k1 <= k2): two smallest of the lanes merged so far
    // mirror butterfly inside the 16-lane row: every step merges DISJOINT lane sets, so (min, second min) stay exact
#define BB_ROW_MIN2(ctrl) do {                                                                              \
        const uint32_t o1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k1, ctrl, 0xf, 0xf, false);             \
        const uint32_t o2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k2, ctrl, 0xf, 0xf, false);             \
        const uint32_t hi1 = max(k1, o1);                                                                         \
        k1 = min(k1, o1);                                                                                         \
        k2 = min(hi1, min(k2, o2));                                                                               \
    } while (0)
    BB_ROW_MIN2(0xB1); BB_ROW_MIN2(0x4E); BB_ROW_MIN2(0x141); BB_ROW_MIN2(0x140);
#undef BB_ROW_MIN2
#define BB_ROW_U32(op, v, ctrl) v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xf, 0xf, false))
    BB_ROW_U32(max, kmaxl, 0xB1); BB_ROW_U32(max, kmaxl, 0x4E); BB_ROW_U32(max, kmaxl, 0x141); BB_ROW_U32(max, kmaxl, 0x140);
#undef BB_ROW_U32
    // rows that hold fewer than two elements do not take part in the bound
    const int first_round = min(N, nthr);            // lanes tid < first_round hold an element
    const int rows_ok = first_round >= 2 ? min(nthr >> 4, ((first_round - 2) >> 4) + 1) : 0;
    const bool bound_ok = 2 * rows_ok >= k;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = wave * 4 + r;
        kmin = min(kmin, (uint32_t)__builtin_amdgcn_readlane((int)k1, 16 * r));
        const uint32_t b = bound_ok ? (uint32_t)__builtin_amdgcn_readlane((int)k2, 16 * r)
                                    : (uint32_t)__builtin_amdgcn_readlane((int)kmaxl, 16 * r);
        if (!bound_ok || row < rows_ok) kmax = max(kmax, b);
    }
    // per-wave slots instead of atomics; the min of wave w goes to hist[w] (low bins), the bound to hist[16+w],
    // both are consumed before the first histogram pass touches the bins
    if (lane == 0) { hist[wave] = kmin; hist[16 + wave] = kmax; }
    for (int i = tid; i < 256; i += nthr) hist2[i] = 0;          // first pass histograms into hist2
}

// Selection only: pins the boundary bucket of the k-th key.  Ends with a barrier; ctrl[3] (compaction cursor) is 0.
__device__ __forceinline__ TopkSel block_topk_select(const float* vals, int N, int k, uint32_t* hist, int tid, int nthr,
                                                     bool prepass_done = false) {
    const int lane = tid & 63, nw = nthr >> 6;
    TOPK_DBG(0);
    uint32_t* ctrl = hist + 256;                 // [0] bucket [1] wanted [2] bucket size [3] compaction cursor
    uint32_t* hist2 = hist + 272;                // second histogram: zeroed while the other one is scanned
    if (!prepass_done) {
        block_topk_prepass(vals, N, k, hist, tid, nthr);
        __syncthreads();
    }
    uint32_t kmin, kmax;
    TOPK_DBG(1);
    // combine the (<= 16) per-wave slots with ONE LDS round trip: lane w fetches wave w's pair, then a 16-lane DPP
    // reduction (a serial loop over the slots costs one LDS latency per wave)
    kmin = 0xFFFFFFFFu; kmax = 0u;
    if (lane < nw) { kmin = hist[lane]; kmax = hist[16 + lane]; }
#define BB_ROW_U32(op, v, ctrl) v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xf, 0xf, false))
    BB_ROW_U32(min, kmin, 0xB1); BB_ROW_U32(min, kmin, 0x4E); BB_ROW_U32(min, kmin, 0x141); BB_ROW_U32(min, kmin, 0x140);
    BB_ROW_U32(max, kmax, 0xB1); BB_ROW_U32(max, kmax, 0x4E); BB_ROW_U32(max, kmax, 0x141); BB_ROW_U32(max, kmax, 0x140);
#undef BB_ROW_U32
    kmin = (uint32_t)__builtin_amdgcn_readlane((int)kmin, 0);
    kmax = (uint32_t)__builtin_amdgcn_readlane((int)kmax, 0);
    const uint32_t diff = kmin ^ kmax;
    int top = diff ? 32 - __clz(diff) : 0;        // number of low bits that are not common to all keys
    uint32_t T = (top >= 32) ? 0u : (kmin >> top) << top;   // common high bits
    uint32_t remaining = (uint32_t)k;
    uint32_t eq_total = (uint32_t)N;              // top == 0: every key equal
    int pass = 0;
    // hist is still being read above by slower waves: the first pass uses hist2 (zeroed before the barrier),
    // later passes alternate; the array for pass p+1 is zeroed by the idle waves while wave 0 scans pass p.
    while (top > 0) {
        uint32_t* h = (pass & 1) ? hist : hist2;
        uint32_t* hn = (pass & 1) ? hist2 : hist;
        const int width = top >= 8 ? 8 : top;
        const int shift = top - width;
        const uint32_t dmask = (1u << width) - 1u;
        for (int n = tid; n < N; n += nthr) {
            const uint32_t key = reward_key(vals[n]);
            const bool in = (top >= 32) || ((key >> top) == (T >> top));
#ifdef BBMPC_EXP_NOATOMIC
            if (in) h[(key >> shift) & dmask] = 1u;
#else
            if (in) atomicAdd(&h[(key >> shift) & dmask], 1u);
#endif
        }
        __syncthreads();
        TOPK_DBG(2 + 2 * pass);
        if (tid < 64) {
            const uint4 c = *reinterpret_cast<const uint4*>(h + 4 * lane);
            const uint32_t s = c.x + c.y + c.z + c.w;
            const uint32_t incl = wave_incl_scan(s, lane);
            const uint32_t excl = incl - s;
            if (excl < remaining && remaining <= incl) {        // exactly one lane
                uint32_t cum = excl, b = 4 * lane, cnt = c.x;
                if (cum + c.x < remaining) { cum += c.x; b += 1; cnt = c.y;
                    if (cum + c.y < remaining) { cum += c.y; b += 1; cnt = c.z;
                        if (cum + c.z < remaining) { cum += c.z; b += 1; cnt = c.w; } } }
                ctrl[0] = b;
                ctrl[1] = remaining - cum;     // how many keys of this bucket are still wanted
                ctrl[2] = cnt;                 // how many keys the bucket holds
                ctrl[3] = 0;                   // compaction cursor
            }
        } else {
            for (int i = tid - 64; i < 256; i += nthr - 64) hn[i] = 0;
        }
        if (nthr == 64)
            for (int i = tid; i < 256; i += 64) hn[i] = 0;
        __syncthreads();
        TOPK_DBG(3 + 2 * pass);
        T |= ctrl[0] << shift;
        remaining = ctrl[1];
        eq_total = ctrl[2];
        top = shift;
        ++pass;
        if (eq_total == remaining) break;      // every key of the boundary bucket is a winner: the undecided
    }                                          // low bits no longer matter
    if (pass == 0) {                           // min == bound: at least k keys tie for best; no pass ran, so the
        if (tid == 0) ctrl[3] = 0;             // bucket size is unknown and nobody reset the cursor: force the
        __syncthreads();                       // exact tie path (lowest indices win)
        eq_total = remaining + 1u;
    }
    return TopkSel{T, remaining, eq_total, top};
}

// Winner test for element n given the selection (exact ties at the boundary: lowest indices win)
__device__ __forceinline__ bool topk_is_winner(const float* vals, int n, const TopkSel& s) {
    const uint32_t Tp = (s.top >= 32) ? 0u : (s.T >> s.top);
    const uint32_t key = reward_key(vals[n]);
    const uint32_t kp_ = (s.top >= 32) ? 0u : (key >> s.top);
    if (kp_ < Tp) return true;
    if (kp_ != Tp) return false;
    if (s.eq_total == s.remaining) return true;                  // the whole boundary bucket is in
    uint32_t before = 0;                                         // top == 0 here (rare): count equal keys ahead of n
    for (int m = 0; m < n; ++m) before += (reward_key(vals[m]) == s.T) ? 1u : 0u;
    return before < s.remaining;
}

// Finish A: the k winners in ASCENDING INDEX order (a deterministic order that needs no ranking): per-wave
// ballot + mbcnt prefix, wave totals through LDS.  Used where only the elite SET matters (CEM's refit).
// One barrier per round of nthr elements + the closing one.
__device__ __forceinline__ void block_topk_finish_indexed(const float* vals, int N, int k, int* eidx, uint32_t* hist,
                                                          const TopkSel& sel, int tid, int nthr) {
    const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    uint32_t* wtot = hist;                                       // [2][16] wave totals, double buffered per round
    uint32_t base = 0;
    int round = 0;
    for (int n0 = 0; n0 < N; n0 += nthr, ++round) {
        const int n = n0 + tid;
        const bool take = (n < N) && topk_is_winner(vals, n, sel);
        const unsigned long long bal = __ballot(take);
        const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        uint32_t* wt = wtot + (round & 1) * 16;
        if (lane == 0) wt[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        // lane w fetches wave w's total (one LDS round trip), 16-lane DPP prefix scan, results through readlane
        int x = (lane < nw) ? (int)wt[lane] : 0;
        x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
        x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
        x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
        x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane(x, 15);
        const uint32_t incl_w = (uint32_t)__builtin_amdgcn_readlane(x, __builtin_amdgcn_readfirstlane(wave));
        const uint32_t off = base + incl_w - (uint32_t)__popcll(bal);
        if (take) eidx[off + pre] = n;
        base += tot;
    }
    __syncthreads();
}

// Finish B: the k winners sorted (tf.nn.top_k(sorted=True)): atomic compaction + k x k ranking.
__device__ __forceinline__ void block_topk_finish_sorted(const float* vals, int N, int k, int* eidx, uint32_t* hist,
                                                         unsigned long long* ekeys, const TopkSel& sel, int tid, int nthr) {
    uint32_t* ctrl = hist + 256;
    const uint32_t T = sel.T, remaining = sel.remaining, eq_total = sel.eq_total;
    const int top = sel.top;
    // ---- compaction of the winners (any order); `top` low bits may be undecided after an early exit
    const uint32_t Tp = (top >= 32) ? 0u : (T >> top);
    for (int e = tid; e < k; e += nthr) eidx[e] = 0;             // rank counters for the next phase
    if (eq_total == remaining) {                                 // the whole boundary bucket is in (the usual case)
        for (int n = tid; n < N; n += nthr) {
            const uint32_t key = reward_key(vals[n]);
            const uint32_t kp_ = (top >= 32) ? 0u : (key >> top);
            if (kp_ <= Tp) {
                const uint32_t slot = atomicAdd(&ctrl[3], 1u);
                ekeys[slot] = ((unsigned long long)key << 32) | (uint32_t)n;
            }
        }
        __syncthreads();
    } else {
        // top == 0 here: more keys EQUAL the k-th one than are wanted, the lowest indices win.  The number of equal keys
        // ahead of element n comes from a ballot / prefix count per wave and the waves' totals (one barrier per round of
        // nthr elements) -- counting them one by one per element was 25-30 us when a hundred candidates tie, which is the
        // steady state of CMA-ES on the pendulum: its step size is never reset (cma_es.py:215-227 restores m and sigma at
        // an episode's start only), after 40 control steps sigma is 2e-5 and the 500 rewards take ten distinct values
        // (profiles/NOTES_r4.md)
        const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
        uint32_t* wtot = hist;                                   // [2][16] wave totals, double buffered per round
        uint32_t base = 0;
        int round = 0;
        for (int n0 = 0; n0 < N; n0 += nthr, ++round) {
            const int n = n0 + tid;
            const uint32_t key = n < N ? reward_key(vals[n]) : 0xFFFFFFFFu;
            const bool tie = n < N && key == T;
            const unsigned long long bal = __ballot(tie);
            const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            uint32_t* wt = wtot + (round & 1) * 16;
            if (lane == 0) wt[wave] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t ahead = base, tot = 0;
            for (int w = 0; w < nw; ++w) { const uint32_t c = wt[w]; ahead += (w < wave) ? c : 0u; tot += c; }
            const bool take = n < N && (key < T || (tie && ahead + pre < remaining));
            if (take) {
                const uint32_t slot = atomicAdd(&ctrl[3], 1u);
                ekeys[slot] = ((unsigned long long)key << 32) | (uint32_t)n;
            }
            base += tot;
        }
        __syncthreads();
    }
    // ---- rank the k winners among themselves with the whole workgroup: thread (e, chunk) counts how many
    // winners in its chunk precede winner e, partial counts meet in LDS (eidx doubles as the counter array).
    const int kp64 = (k + 63) & ~63;                 // e runs over full waves so chunk ids are wave-uniform
    const int nchunk = max(1, nthr / kp64);
    const int clen = (k + nchunk - 1) / nchunk;
    for (int t = tid; t < kp64 * nchunk; t += nthr) {
        const int e = t % kp64, c = t / kp64;
        if (e < k) {
            const unsigned long long mine = ekeys[e];
            const int o0 = c * clen, o1 = min(k, o0 + clen);
            int cnt = 0;
#pragma unroll 4
            for (int o = o0; o < o1; ++o) cnt += (ekeys[o] < mine) ? 1 : 0;
            if (cnt) atomicAdd((uint32_t*)&eidx[e], (uint32_t)cnt);
        }
    }
    __syncthreads();
    int myrank = -1, myidx = 0;
    if (tid < k) { myrank = eidx[tid]; myidx = (int)(uint32_t)(ekeys[tid] & 0xFFFFFFFFull); }
    __syncthreads();
    if (myrank >= 0) eidx[myrank] = myidx;
    __syncthreads();
}

// vals[N] in LDS (left untouched); eidx[k] out; hist[TOPK_HIST_WORDS] and ekeys[k] are LDS scratch.
// All threads of the workgroup must call (k <= nthr); ends with a barrier (eidx visible to everyone).
__device__ __forceinline__ void block_topk_sorted(const float* vals, int N, int k, int* eidx, uint32_t* hist,
                                                  unsigned long long* ekeys, int tid, int nthr) {
    const TopkSel sel = block_topk_select(vals, N, k, hist, tid, nthr);
    block_topk_finish_sorted(vals, N, k, eidx, hist, ekeys, sel, tid, nthr);
}

}  // namespace bbmpc
