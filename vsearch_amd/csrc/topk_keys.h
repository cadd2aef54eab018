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

}  // namespace vs
