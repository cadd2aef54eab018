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
// (8 loads per thread issued before the first is used: with one in flight a row's read is all latency)
template <int UNUSED>
__global__ __launch_bounds__(kSpThreads) void count_nz_kernel(const float* x, int64_t ld, int32_t B, int32_t V, int64_t* counts) {
    __shared__ int scratch[32];
    const int tid = threadIdx.x;
    constexpr int U = 8;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const float* xr = x + (size_t)b * ld;
        int c = 0;
        int i = tid;
        for (; i + (U - 1) * kSpThreads < V; i += U * kSpThreads) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = xr[i + u * kSpThreads];
#pragma unroll
            for (int u = 0; u < U; ++u) c += v[u] != 0.f;
        }
        for (; i < V; i += kSpThreads) c += xr[i] != 0.f;
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

// Ordered compaction, one workgroup per row.  A thread reads its column of up to 32 steps of 1024 columns in ONE batch of loads; the
// position of a non-zero = row base + non-zeros in earlier (step, wave) pairs -- an exclusive scan over the 32 x 16 ballot counts,
// which in (step, wave) order is the output order -- + earlier lanes of its wave (mbcnt).  3 barriers per 32 Ki columns, where the
// step-by-step version this replaces took one per 1024 columns, each behind the step's load.
constexpr int kFillSteps = 32;
template <int UNUSED>
__global__ __launch_bounds__(kSpThreads) void fill_csr_kernel(const float* x, int64_t ld, int32_t B, int32_t V, const int64_t* rowptr,
                                                              int32_t* cols, float* vals, int64_t cap) {
    constexpr int NW = kSpThreads / 64;
    __shared__ int wave_cnt[kFillSteps * NW];
    __shared__ int scratch[32];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        int64_t base = rowptr[b];
        if (rowptr[b + 1] == base) continue;                   // empty row (or one the planner left out): nothing to write
        const float* xr = x + (size_t)b * ld;
        for (int c0 = 0; c0 < V; c0 += kFillSteps * kSpThreads) {
            float v[kFillSteps];
#pragma unroll
            for (int st = 0; st < kFillSteps; ++st) {
                const int i = c0 + st * kSpThreads + tid;
                v[st] = i < V ? xr[i] : 0.f;
            }
#pragma unroll
            for (int st = 0; st < kFillSteps; ++st) {
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(v[st] != 0.f);
                if (lane == 0) wave_cnt[st * NW + w] = __builtin_popcountll(bal);
            }
            __syncthreads();
            int total = 0;
            const int mine = tid < kFillSteps * NW ? wave_cnt[tid] : 0;
            const int before = block_excl_scan(mine, scratch, tid, &total);
            if (tid < kFillSteps * NW) wave_cnt[tid] = before;
            __syncthreads();
#pragma unroll
            for (int st = 0; st < kFillSteps; ++st) {
                const bool nz = v[st] != 0.f;
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(nz);
                if (nz) {
                    const int64_t pos = base + wave_cnt[st * NW + w] + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    if (pos < cap) { cols[pos] = c0 + st * kSpThreads + tid; vals[pos] = v[st]; }
                }
            }
            base += total;
            __syncthreads();
        }
    }
}

}  // namespace vs
