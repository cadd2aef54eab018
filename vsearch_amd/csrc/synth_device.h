// synth_device.h -- device twin of vsearch_amd/synth.py / oracle/vs_oracle.c (vso_synth_csr).
// Integer-exact: every row is a pure function of (seed, global row id).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vs {

__host__ __device__ inline uint64_t sm64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline uint64_t hash2(uint64_t seed, uint64_t a) { return sm64(sm64(seed) ^ (a * 0xD1342543DE82EF95ull)); }
__host__ __device__ inline uint64_t hash3(uint64_t seed, uint64_t a, uint64_t b) { return sm64(hash2(seed, a) + b * 0x2545F4914F6CDD1Dull); }

__host__ __device__ inline uint32_t feistel16(uint32_t x, uint64_t key) {
    uint32_t L = x >> 8, R = x & 0xFF;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t k = (uint32_t)((key >> (16 * i)) & 0xFFFF);
        const uint32_t t = (R ^ k) * 0x9E3779B1u + k;
        const uint32_t F = (t >> 24) & 0xFF;
        const uint32_t nl = R;
        R = L ^ F;
        L = nl;
    }
    return (L << 8) | R;
}
__host__ __device__ inline uint32_t perm_col(uint64_t key, uint32_t j, uint32_t n_cols) {
    uint32_t x = feistel16(j, key);
    while (x >= n_cols) x = feistel16(x, key);
    return x;
}
// kind 2: skewed column popularity ("Zipf, s = 1"): popularity rank r (1 = most popular) is used by a row with probability
// min(1, c / r).  The rank space is cut into octaves [2^k, 2^(k+1)): a row takes ALL ranks of octaves 0..6 (ranks 1..127) and
// the same number of distinct ranks from each octave above (equal mass per octave = the 1/r law), drawn by a keyed affine
// bijection on the octave; a fixed bijection (the same for every seed: corpus and queries agree on what is popular) maps ranks to
// column ids.  Integer-exact like the rest of the generator.  Element j (0 <= j < len) of a row:
__host__ __device__ inline uint32_t synth_skew_col(uint64_t key, uint32_t j, uint32_t len, uint32_t n_cols) {
    constexpr uint32_t kHead = 127;
    uint32_t rank;
    uint32_t kmax = 0;
    while ((2u << kmax) <= n_cols) ++kmax;                    // highest octave that starts inside [1, n_cols]
    if (len <= kHead + 8 || kmax < 8) {
        rank = j + 1;                                         // degenerate shapes: the `len` most popular columns
    } else if (j < kHead) {
        rank = j + 1;
    } else {
        const uint32_t n_oct = kmax - 7 + 1, per = (len - kHead) / n_oct;
        const uint32_t t = j - kHead;
        uint32_t o = t / per;
        if (o > n_oct - 1) o = n_oct - 1;                     // the remainder goes to the last (largest) octave
        const uint32_t i = t - o * per, k = 7 + o;
        const uint32_t lo = 1u << k, size = (2u << k) <= n_cols + 1 ? lo : n_cols + 1 - lo, mask = lo - 1;
        const uint64_t h = sm64(key + 0x9E3779B97F4A7C15ull * (k + 1));
        const uint32_t odd = (uint32_t)h | 1u, add = (uint32_t)(h >> 32);
        uint32_t x = i;
        do { x = (x * odd + add) & mask; } while (x >= size);   // cycle walking keeps the affine map a bijection on [0, size)
        rank = lo + x;
    }
    return perm_col(0x5A495046534B4557ull, rank - 1, n_cols);
}
__host__ __device__ inline uint32_t synth_col(uint64_t key, uint32_t j, uint32_t len, int kind, uint32_t n_cols) {
    return kind == 2 ? synth_skew_col(key, j, len, n_cols) : perm_col(key, j, n_cols);
}
__host__ __device__ inline int64_t synth_row_len(uint64_t seed, int64_t row, int kind, int32_t nnz, int32_t n_cols) {
    int64_t len = nnz;
    if (kind == 1) {
        const uint64_t h = hash3(seed, (uint64_t)row, 0x4C454Eull);
        int64_t s = 0;
        for (int i = 0; i < 4; ++i) s += (int64_t)((h >> (16 * i)) & 0xFFFF);
        len = 1 + (s * (int64_t)(nnz - 1)) / (2 * 65536);
    }
    return len < n_cols ? len : n_cols;
}
__host__ __device__ inline float synth_val(uint64_t seed, int64_t row, uint32_t col, int val_law) {
    const uint64_t h = hash3(seed ^ 0x56414Cull, (uint64_t)row, (uint64_t)col);
    if (val_law == 0) return (164.0f + (float)(h % 49152ull)) / 16384.0f;   // exact: dyadic grid 2^-14
    if (val_law == 1) return (1.0f + (float)(h % 255ull)) / 64.0f;
    return 1.0f;
}

}  // namespace vs
