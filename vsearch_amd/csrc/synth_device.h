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
