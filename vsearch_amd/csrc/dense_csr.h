// dense_csr.h -- dense [B, V] fp32 -> CSR (Tensor.to_sparse_csr(), retriever.py:304) building blocks,
// shared by vs_dense_to_csr (index build) and the query sparsifier of the multi-query CSR scan.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vs {

constexpr int kSpThreads = 1024;

// block-wide exclusive scan of one int per thread (kSpThreads threads); scratch: 16 ints in LDS
__device__ __forceinline__ int block_excl_scan(int v, int* scratch, int tid, int* total) {
    const int lane = tid & 63, w = tid >> 6;
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    __syncthreads();
    if (lane == 63) scratch[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < kSpThreads / 64; ++i) {
        const int s = scratch[i];
        if (i < w) base += s;
        tot += s;
    }
    if (total) *total = tot;
    return base + incl - v;
}

// ---- Tensor.to_sparse_csr() (retriever.py:304): non-zeros of a dense [B, V] matrix ----------------
template <int UNUSED>
__global__ __launch_bounds__(kSpThreads) void count_nz_kernel(const float* x, int64_t ld, int32_t B, int32_t V, int64_t* counts) {
    __shared__ int scratch[32];
    const int tid = threadIdx.x;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        int c = 0;
        for (int i = tid; i < V; i += kSpThreads) c += x[(size_t)b * ld + i] != 0.f;
        int total = 0;
        block_excl_scan(c, scratch, tid, &total);
        if (tid == 0) counts[b] = total;
        __syncthreads();
    }
}

// rowptr = exclusive prefix sums of the per-row counts (ONE workgroup of kSpThreads threads; contiguous segment per thread)
template <int UNUSED>
__global__ __launch_bounds__(kSpThreads) void scan_counts_kernel(const int64_t* counts, int32_t B, int64_t* rowptr) {
    __shared__ int64_t wave_tot[kSpThreads / 64];
    if (blockIdx.x != 0) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int seg = (B + kSpThreads - 1) / kSpThreads;
    const int i0 = min(B, tid * seg), i1 = min(B, i0 + seg);
    int64_t mine = 0;
    for (int i = i0; i < i1; ++i) mine += counts[i];
    int64_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[w] = incl;
    __syncthreads();
    int64_t pos = incl - mine;
    for (int k = 0; k < w; ++k) pos += wave_tot[k];
    for (int i = i0; i < i1; ++i) { rowptr[i] = pos; pos += counts[i]; }
    if (i0 < B && i1 == B) rowptr[B] = pos;
}

// Ordered compaction, one workgroup per row, 1024 columns per step: coalesced reads, the position of a non-zero =
// row base + non-zeros in earlier steps + earlier waves (LDS) + earlier lanes (ballot / mbcnt).
template <int UNUSED>
__global__ __launch_bounds__(kSpThreads) void fill_csr_kernel(const float* x, int64_t ld, int32_t B, int32_t V, const int64_t* rowptr,
                                                              int32_t* cols, float* vals, int64_t cap) {
    __shared__ int wave_cnt[2][kSpThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        int64_t base = rowptr[b];
        if (rowptr[b + 1] == base) continue;                   // empty row (or one the planner left out): nothing to write
        int buf = 0;
        for (int c0 = 0; c0 < V; c0 += kSpThreads, buf ^= 1) {
            const int i = c0 + tid;
            const float v = i < V ? x[(size_t)b * ld + i] : 0.f;
            const bool nz = v != 0.f;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(nz);
            if (lane == 0) wave_cnt[buf][w] = __builtin_popcountll(bal);
            __syncthreads();                                   // (double-buffered: one barrier per step)
            int before = 0, total = 0;
#pragma unroll
            for (int k = 0; k < kSpThreads / 64; ++k) {
                const int n = wave_cnt[buf][k];
                before += k < w ? n : 0;
                total += n;
            }
            if (nz) {
                const int64_t pos = base + before + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                if (pos < cap) { cols[pos] = i; vals[pos] = v; }
            }
            base += total;
        }
        __syncthreads();
    }
}

}  // namespace vs
