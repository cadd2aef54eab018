// topk_keys.h -- 64-bit order keys and LDS / in-register bitonic networks for top-k on gfx950.
//
// A candidate (score, row) is one uint64:  hi = order-preserving transform of the fp32 score,
// lo = ~row.  Larger key = better candidate under the library's canonical order
// (score descending, id ascending), so keys are totally ordered and top-k has no ties to break.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vs {

__device__ __forceinline__ uint32_t flip_f32(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unflip_f32(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}
__device__ __forceinline__ uint64_t make_key(float score, uint32_t row) {
    return ((uint64_t)flip_f32(score) << 32) | (uint32_t)(~row);
}
__device__ __forceinline__ float key_score(uint64_t k) { return unflip_f32((uint32_t)(k >> 32)); }
__device__ __forceinline__ uint32_t key_row(uint64_t k) { return ~(uint32_t)k; }

// Workgroup-wide bitonic sort, descending, of n (power of two) keys in LDS. NT = threads.
template <int NT>
__device__ __forceinline__ void wg_sort_desc(uint64_t* buf, int n, int tid) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < (n >> 1); t += NT) {
                const int i = 2 * t - (t & (stride - 1));
                const int j = i + stride;
                const bool asc = (i & size) != 0;
                const uint64_t a = buf[i], b = buf[j];
                if (asc ? (a > b) : (a < b)) {
                    buf[i] = b;
                    buf[j] = a;
                }
            }
        }
    }
    __syncthreads();
}

// Cut a candidate buffer to its K best WITHOUT sorting it: MSD radix select of the K-th largest key (8 bits a pass, two histograms
// alternating: two barriers a pass; it ends at the first pass whose bin holds one key -- keys are distinct, scores mostly are: 3 - 4
// passes), then the keys >= it leave for gdst[0, K) in no particular order.  A walk's cut only needs the SET of survivors and the new
// threshold; the bitonic sort it replaced was 66 barrier stages for 2048 keys (~ 35 k cycles a slot: 6 % of a 2.6 M-doc shard's walk,
// 11 % at 1 M docs -- round 6).
//   buf: 2 * NT keys in LDS (0 = empty slot), at least K + 1 of them non-zero; hist: (2 * 256 + 8) uint32 of LDS scratch; every thread
//   of the workgroup calls it; the caller's barrier separates the fill of buf from the call.  Returns the K-th largest key.
constexpr int kCutHistWords = 2 * 256 + 8;
// (NOT inlined: the walks call it on their rare path -- inlined, its registers and the compiler's spills around it land in the walks' hot loops)
template <int NT>
__device__ __attribute__((noinline)) uint64_t wg_cut_topk(const uint64_t* buf, int K, uint64_t* gdst, uint32_t* hist, int tid) {
    const int lane = tid & 63;
    const uint64_t k0 = buf[tid], k1 = buf[tid + NT];
    uint32_t* ctl = hist + 512;                   // [0] bin, [1] rank left inside it, [2] keys in it, [3] output counter, [4..5] the K-th key
    uint64_t prefix = 0, mask = 0;
    uint32_t want = (uint32_t)K;
    if (tid < 256) hist[tid] = 0u;
    if (tid == 0) ctl[3] = 0u;
    __syncthreads();
    // one key into a histogram; a wave whose live keys all share the digit (the top bytes of scores a threshold apart) adds once
    auto count = [&](uint32_t* h, bool on, uint32_t digit) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(on);
        if (m == 0ull) return;
        const uint32_t d0 = (uint32_t)__shfl((int)digit, __builtin_ctzll(m), 64);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(on && digit == d0);
        if (same == m) { if (lane == __builtin_ctzll(m)) atomicAdd(&h[d0], (uint32_t)__builtin_popcountll(m)); }
        else if (on) atomicAdd(&h[digit], 1u);
    };
    int pass = 0;
    for (int shift = 56; shift >= 0; shift -= 8, ++pass) {
        uint32_t* h = hist + 256 * (pass & 1);
        count(h, k0 != 0ull && (k0 & mask) == prefix, (uint32_t)(k0 >> shift) & 255u);
        count(h, k1 != 0ull && (k1 & mask) == prefix, (uint32_t)(k1 >> shift) & 255u);
        if (tid >= 256 && tid < 512) hist[256 * ((pass + 1) & 1) + (tid - 256)] = 0u;      // the other histogram, for the next pass (last read a barrier ago)
        __syncthreads();
        if (tid < 64) {
            // lane l holds bins 4 l .. 4 l + 3; keys in the bins of higher lanes, then the bin inside the lane where the rank falls
            const uint4 c = reinterpret_cast<const uint4*>(h)[tid];
            const uint32_t s = c.x + c.y + c.z + c.w;
            uint32_t incl = s;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_down((int)incl, o, 64); if (tid + o < 64) incl += t; }
            const uint32_t above = incl - s;
            if (above < want && want <= incl) {
                uint32_t a = above, b, cb;
                if (want <= a + c.w) { b = 3u; cb = c.w; }
                else { a += c.w; if (want <= a + c.z) { b = 2u; cb = c.z; }
                else { a += c.z; if (want <= a + c.y) { b = 1u; cb = c.y; } else { a += c.y; b = 0u; cb = c.x; } } }
                ctl[0] = 4u * (uint32_t)tid + b; ctl[1] = want - a; ctl[2] = cb;
            }
        }
        __syncthreads();
        prefix |= (uint64_t)ctl[0] << shift;
        mask |= (uint64_t)0xFFu << shift;
        want = ctl[1];
        if (ctl[2] == 1u) break;                  // (uniform) one key carries this prefix: the K-th
    }
    // survivors: every key whose decided bits are above the prefix, and the one (the K-th) that carries it
    const bool s0 = k0 != 0ull && (k0 & mask) >= prefix, s1 = k1 != 0ull && (k1 & mask) >= prefix;
    auto emit = [&](bool on, uint64_t key) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(on);
        if (m == 0ull) return;
        uint32_t base = 0u;
        if (lane == __builtin_ctzll(m)) base = atomicAdd(&ctl[3], (uint32_t)__builtin_popcountll(m));
        base = (uint32_t)__shfl((int)base, __builtin_ctzll(m), 64);
        if (on) {
            gdst[base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = key;
            if ((key & mask) == prefix) { ctl[4] = (uint32_t)key; ctl[5] = (uint32_t)(key >> 32); }
        }
    };
    emit(s0, k0);
    emit(s1, k1);
    __syncthreads();
    return ((uint64_t)ctl[5] << 32) | ctl[4];
}

// One wave sorts 256 keys held 4 per lane (element e = r*64 + lane), descending, no LDS, no barrier.
__device__ __forceinline__ void wave_sort256_desc(uint64_t (&k)[4], int lane) {
#pragma unroll
    for (int size = 2; size <= 256; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 64) {
                const int rs = stride >> 6;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if ((r & rs) == 0) {
                        const bool asc = ((r << 6) & size) != 0;
                        const uint64_t a = k[r], b = k[r | rs];
                        const bool sw = asc ? (a > b) : (a < b);
                        k[r] = sw ? b : a;
                        k[r | rs] = sw ? a : b;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint64_t mine = k[r];
                    const uint64_t other = __shfl_xor(mine, stride, 64);
                    const int e = (r << 6) | lane;
                    const bool asc = (e & size) != 0;
                    const bool lower = (lane & stride) == 0;
                    const bool want_max = (lower != asc);  // lower slot of a descending pair keeps the max
                    const bool take = want_max ? (other > mine) : (other < mine);
                    k[r] = take ? other : mine;
                }
            }
        }
    }
}

// A work item's final result for K <= 256: the K best of the candidate buffer, sorted descending, to out[0, K) (zeros behind the last).
// buf: the 2 * NT keys in LDS (0 = empty), cnt of them live; scratch_g: K keys of global scratch (the buffer's own backing store);
// cut to K by radix select when there are more, then ONE wave sorts them in registers.  Not inlined (rare path, see wg_cut_topk).
template <int NT>
__device__ __attribute__((noinline)) void wg_final_topk256(const uint64_t* buf, uint32_t cnt, int K, uint64_t* scratch_g, uint64_t* out, uint32_t* hist, int tid) {
    const bool cut = cnt > (uint32_t)K;
    if (cut) (void)wg_cut_topk<NT>(buf, K, scratch_g, hist, tid);
    else __syncthreads();
    const uint32_t n = min(cnt, (uint32_t)K);
    if (tid < 64) {
        uint64_t kk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const uint32_t e = (uint32_t)(r * 64 + tid); kk[r] = e < n ? (cut ? scratch_g[e] : buf[e]) : 0ull; }
        wave_sort256_desc(kk, tid);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int e = r * 64 + tid; if (e < K) out[e] = kk[r]; }
    }
}

}  // namespace vs
