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
__device__ __forceinline__ void block_topk_sorted(const float* vals, int N, int k, int* eidx, uint32_t* hist,
                                                  unsigned long long* ekeys, int tid, int nthr) {
    const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    uint32_t* ctrl = hist + 256;                 // [0] bucket [1] wanted [2] bucket size [3] compaction cursor
    uint32_t* hist2 = hist + 272;                // second histogram: zeroed while the other one is scanned
    // ---- key range: digits are taken from the highest bit where min and max differ.  Rewards of one
    // population mostly share sign + exponent, so a fixed top-8-bit digit would send a whole wave's ds_add_u32
    // to one or two bins (serialised); digits below the common prefix spread over the bins.
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    for (int n = tid; n < N; n += nthr) {
        const uint32_t key = reward_key(vals[n]);
        kmin = min(kmin, key);
        kmax = max(kmax, key);
    }
    // 16-lane DPP rows, then the four row results through readlane (a bpermute shuffle costs ~100 cycles/step)
#define BB_ROW_U32(op, v, ctrl) v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xf, 0xf, false))
    BB_ROW_U32(min, kmin, 0xB1); BB_ROW_U32(min, kmin, 0x4E); BB_ROW_U32(min, kmin, 0x141); BB_ROW_U32(min, kmin, 0x140);
    BB_ROW_U32(max, kmax, 0xB1); BB_ROW_U32(max, kmax, 0x4E); BB_ROW_U32(max, kmax, 0x141); BB_ROW_U32(max, kmax, 0x140);
#undef BB_ROW_U32
    kmin = min(min((uint32_t)__builtin_amdgcn_readlane((int)kmin, 0), (uint32_t)__builtin_amdgcn_readlane((int)kmin, 16)),
               min((uint32_t)__builtin_amdgcn_readlane((int)kmin, 32), (uint32_t)__builtin_amdgcn_readlane((int)kmin, 48)));
    kmax = max(max((uint32_t)__builtin_amdgcn_readlane((int)kmax, 0), (uint32_t)__builtin_amdgcn_readlane((int)kmax, 16)),
               max((uint32_t)__builtin_amdgcn_readlane((int)kmax, 32), (uint32_t)__builtin_amdgcn_readlane((int)kmax, 48)));
    // per-wave slots instead of atomics; the min of wave w goes to hist[w] (low bins), the max to hist[16+w],
    // both are consumed before the first histogram pass touches the bins
    if (lane == 0) { hist[wave] = kmin; hist[16 + wave] = kmax; }
    for (int i = tid; i < 256; i += nthr) hist2[i] = 0;          // first pass histograms into hist2
    __syncthreads();
    kmin = 0xFFFFFFFFu; kmax = 0u;
    for (int w = 0; w < nw; ++w) { kmin = min(kmin, hist[w]); kmax = max(kmax, hist[16 + w]); }
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
            if (in) atomicAdd(&h[(key >> shift) & dmask], 1u);
        }
        __syncthreads();
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
        T |= ctrl[0] << shift;
        remaining = ctrl[1];
        eq_total = ctrl[2];
        top = shift;
        ++pass;
        if (eq_total == remaining) break;      // every key of the boundary bucket is a winner: the undecided
    }                                          // low bits no longer matter
    if (pass == 0) {                           // all keys equal: no pass ran, nobody reset the cursor
        if (tid == 0) ctrl[3] = 0;
        __syncthreads();
    }
    // ---- compaction of the winners (any order); `top` low bits may be undecided after an early exit
    const uint32_t Tp = (top >= 32) ? 0u : (T >> top);
    for (int e = tid; e < k; e += nthr) eidx[e] = 0;             // rank counters for the next phase
    for (int n = tid; n < N; n += nthr) {
        const uint32_t key = reward_key(vals[n]);
        const uint32_t kp_ = (top >= 32) ? 0u : (key >> top);
        bool take = kp_ < Tp;
        if (kp_ == Tp) {
            if (eq_total == remaining) take = true;              // the whole boundary bucket is in
            else {                                               // top == 0 here: exact ties, lowest indices win
                uint32_t before = 0;
                for (int m = 0; m < n; ++m) before += (reward_key(vals[m]) == T) ? 1u : 0u;
                take = before < remaining;
            }
        }
        if (take) {
            const uint32_t slot = atomicAdd(&ctrl[3], 1u);
            ekeys[slot] = ((unsigned long long)key << 32) | (uint32_t)n;
        }
    }
    __syncthreads();
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

}  // namespace bbmpc
