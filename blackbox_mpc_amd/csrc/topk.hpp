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

#ifdef BBMPC_TOPK_DBG
__device__ long long g_topk_dbg[32];
#define TK(slot) do { if (tid == 0 && blockIdx.x == 0) g_topk_dbg[(slot)] = (long long)wall_clock64(); } while (0)
#else
#define TK(slot) do {} while (0)
#endif

constexpr int TOPK_HIST_WORDS = 272;   // 256 bins + 16 control words (256.. bucket, 257 wanted, 258 bucket size,
                                       // 259 compaction cursor, 260/261 key min/max)

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
// All threads of the workgroup must call; ends with a barrier (eidx visible to everyone).
__device__ __forceinline__ void block_topk_sorted(const float* vals, int N, int k, int* eidx, uint32_t* hist,
                                                  unsigned long long* ekeys, int tid, int nthr) {
    const int lane = tid & 63;
    TK(0);
    // Key range first: digits are taken from the highest bit where min and max differ.  Rewards of one
    // population mostly share sign + exponent, so a fixed top-8-bit digit would send a whole wave's
    // ds_add_u32 to one or two bins (serialised); digits below the common prefix spread over the bins.
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    for (int n = tid; n < N; n += nthr) {
        const uint32_t key = reward_key(vals[n]);
        kmin = min(kmin, key);
        kmax = max(kmax, key);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o, 64));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o, 64));
    }
    if (tid == 0) { hist[260] = 0xFFFFFFFFu; hist[261] = 0u; }
    __syncthreads();
    if (lane == 0) { atomicMin(&hist[260], kmin); atomicMax(&hist[261], kmax); }
    __syncthreads();
    kmin = hist[260];
    kmax = hist[261];
    TK(1);
    const uint32_t diff = kmin ^ kmax;
    int top = diff ? 32 - __clz(diff) : 0;        // number of low bits that are not common to all keys
    uint32_t T = (top >= 32) ? 0u : (kmin >> top) << top;   // common high bits
    uint32_t remaining = (uint32_t)k;
    if (top == 0 && tid == 0) hist[258] = (uint32_t)N;       // every key equal
    int passno = 0;
    while (top > 0) {
        TK(2 + 4 * passno);
        const int width = top >= 8 ? 8 : top;
        const int shift = top - width;
        const uint32_t dmask = (1u << width) - 1u;
        for (int i = tid; i < 256; i += nthr) hist[i] = 0;
        __syncthreads();
        for (int n = tid; n < N; n += nthr) {
            const uint32_t key = reward_key(vals[n]);
            const bool in = (top >= 32) || ((key >> top) == (T >> top));
            if (in) atomicAdd(&hist[(key >> shift) & dmask], 1u);
        }
        __syncthreads();
        TK(3 + 4 * passno);
        if (tid < 64) {
            const uint4 c = *reinterpret_cast<const uint4*>(hist + 4 * lane);
            const uint32_t s = c.x + c.y + c.z + c.w;
            const uint32_t incl = wave_incl_scan(s, lane);
            const uint32_t excl = incl - s;
            if (excl < remaining && remaining <= incl) {        // exactly one lane
                uint32_t cum = excl, b = 4 * lane, cnt = c.x;
                if (cum + c.x < remaining) { cum += c.x; b += 1; cnt = c.y;
                    if (cum + c.y < remaining) { cum += c.y; b += 1; cnt = c.z;
                        if (cum + c.z < remaining) { cum += c.z; b += 1; cnt = c.w; } } }
                hist[256] = b;
                hist[257] = remaining - cum;     // how many keys of this bucket are still wanted
                hist[258] = cnt;                 // how many keys the bucket holds
            }
        }
        TK(4 + 4 * passno);
        __syncthreads();
        TK(5 + 4 * passno);
        ++passno;
        (void)passno;
        T |= hist[256] << shift;
        remaining = hist[257];
        const bool whole_bucket = hist[258] == remaining;     // every key of the boundary bucket is a winner:
        top = shift;                                            // the undecided low bits no longer matter
        __syncthreads();
        if (whole_bucket) break;
    }
    // T is now the key of the k-th best
    TK(20);
    if (tid == 0) hist[259] = 0;
    __syncthreads();
    const uint32_t eq_total = hist[258];         // population members with exactly that key
    // `top` low bits may be undecided after an early exit; compare on the decided prefix
    const uint32_t Tp = (top >= 32) ? 0u : (T >> top);
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
            const uint32_t slot = atomicAdd(&hist[259], 1u);
            ekeys[slot] = ((unsigned long long)key << 32) | (uint32_t)n;
        }
    }
    __syncthreads();
    TK(21);
    // rank the k winners among themselves with the whole workgroup: thread (e, chunk) counts how many
    // winners in its chunk precede winner e, partial counts meet in LDS (eidx doubles as the counter array).
    for (int e = tid; e < k; e += nthr) eidx[e] = 0;
    __syncthreads();
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
    __syncthreads();                                 // callers guarantee k <= nthr
    if (myrank >= 0) eidx[myrank] = myidx;
    __syncthreads();
    TK(22);
}

}  // namespace bbmpc
