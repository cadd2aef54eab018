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

template <int UNUSED>
__global__ void scan_counts_kernel(const int64_t* counts, int32_t B, int64_t* rowptr) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int64_t acc = 0;
        rowptr[0] = 0;
        for (int b = 0; b < B; ++b) { acc += counts[b]; rowptr[b + 1] = acc; }
    }
}

template <int UNUSED>
__global__ __launch_bounds__(kSpThreads) void fill_csr_kernel(const float* x, int64_t ld, int32_t B, int32_t V, const int64_t* rowptr,
                                                              int32_t* cols, float* vals, int64_t cap) {
    __shared__ int scratch[32];
    const int tid = threadIdx.x;
    const int seg = (V + kSpThreads - 1) / kSpThreads;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const int i0 = tid * seg, i1 = min(V, i0 + seg);
        int c = 0;
        for (int i = i0; i < i1; ++i) c += x[(size_t)b * ld + i] != 0.f;
        int64_t pos = rowptr[b] + block_excl_scan(c, scratch, tid, nullptr);
        for (int i = i0; i < i1; ++i) {
            const float v = x[(size_t)b * ld + i];
            if (v != 0.f) {
                if (pos < cap) { cols[pos] = i; vals[pos] = v; }
                ++pos;
            }
        }
        __syncthreads();
    }
}


}  // namespace vs
