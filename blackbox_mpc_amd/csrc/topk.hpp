// Exact sorted top-k of a workgroup's LDS-resident rewards: tf.nn.top_k(sorted=True) semantics
// (cem.py:97-99) -- larger first, ties -> lower index first.
//
// MSD radix select on order-preserving 32-bit keys: 4 passes x 8 bits, each pass one LDS histogram
// (ds_add_u32) + one 256-bin scan by a single wave, i.e. O(N) work per pass instead of the O(N^2)
// comparison count of ranking by counting.  That pins the k-th key exactly; the <= k winners are then
// compacted and ranked among themselves (k^2 comparisons on 64-bit (key,index) words).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bbmpc {

constexpr int TOPK_HIST_WORDS = 272;   // 256 bins + 16 control words

// smaller key == better (larger reward); equal rewards <=> equal keys (-0 is folded onto +0 first)
__device__ __forceinline__ uint32_t reward_key(float r) {
    const uint32_t u = __float_as_uint(r + 0.0f);
    const uint32_t asc = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
    return ~asc;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// vals[N] in LDS (left untouched); eidx[k] out; hist[TOPK_HIST_WORDS] and ekeys[k] are LDS scratch.
// All threads of the workgroup must call; ends with a barrier (eidx visible to everyone).
__device__ __forceinline__ void block_topk_sorted(const float* vals, int N, int k, int* eidx, uint32_t* hist,
                                                  unsigned long long* ekeys, int tid, int nthr) {
    const int lane = tid & 63;
    uint32_t prefix = 0, remaining = (uint32_t)k;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < 256; i += nthr) hist[i] = 0;
        __syncthreads();
        for (int n = tid; n < N; n += nthr) {
            const uint32_t key = reward_key(vals[n]);
            if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
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
        __syncthreads();
        prefix = (prefix << 8) | hist[256];
        remaining = hist[257];
        __syncthreads();
    }
    const uint32_t T = prefix;                   // key of the k-th best
    const uint32_t eq_total = hist[258];         // population members with exactly that key
    if (tid == 0) hist[259] = 0;
    __syncthreads();
    for (int n = tid; n < N; n += nthr) {
        const uint32_t key = reward_key(vals[n]);
        bool take = key < T;
        if (key == T) {
            if (eq_total == remaining) take = true;              // every tied member is in
            else {                                               // rare: lowest indices among the ties win
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
    for (int e = tid; e < k; e += nthr) {
        const unsigned long long mine = ekeys[e];
        int rank = 0;
        for (int o = 0; o < k; ++o) rank += (ekeys[o] < mine) ? 1 : 0;
        eidx[rank] = (int)(uint32_t)(mine & 0xFFFFFFFFull);
    }
    __syncthreads();
}

}  // namespace bbmpc
