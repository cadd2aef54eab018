// csr_scan.h -- device code of the CSR scoring pass (K1+K3 of SURVEY.md §2.1):
//   scores[b, r] = sum_j q[b, col[r, j]] * val[r, j]      (index.py:91, torch sparse-CSR addmm)
//   top-k per query over r                                 (index.py:92, Tensor.topk)
// without materialising the [B, N] score matrix.
//
// Layout streamed from HBM ("packets"): a row is a run of 16-byte column packets (8 x uint16 column
// ids) plus, for valued indexes, the matching 8 fp32 (32 B) or fp16 (16 B) values; rows are padded
// to whole packets with column id n_cols, whose slot in the LDS query image holds 0.
// A row is scored by G consecutive lanes (G*8 nnz per step, 16-byte coalesced loads); the partial
// sums are combined with xor-shuffles inside the wave.  The query's weights live in LDS as a
// dense fp32 image (118 KB for V = 29 523) gathered by column id.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "topk_keys.h"
#include "dense_csr.h"

namespace vs {

constexpr int kScanThreads = 1024;          // 16 waves = 4 per SIMD; LDS limits us to 1 workgroup per CU
constexpr int kScanWaves = kScanThreads / 64;
constexpr int kWaveCap = 256;               // wave-private candidate slots (4 per lane)
constexpr int kWgCap = kScanWaves * kWaveCap;   // 4096 keys = 32 KB
constexpr int kMaxKWave = 128;              // wave-private path: k <= 128
constexpr int kMaxKShared = 2048;           // shared-buffer path: k <= 2048

enum : int { VM_F32 = 0, VM_F16 = 1, VM_BIN = 2 };

struct ScanArgs {
    const uint32_t* pk_ptr;   // [n_rows + 1] packet offsets
    const uint4* cols;        // [n_packets] 8 x u16
    const void* vals;         // [n_packets * 8] fp32 | fp16 | null
    const float* q;           // [B, n_cols] fp32, contiguous (already rounded to the index dtype)
    int64_t n_rows;
    int32_t n_cols;
    int32_t B;
    int32_t k;
    int32_t nchunk;
    int64_t rows_per_chunk;
    uint64_t* cand;           // [B, nchunk, k] keys, sorted descending per (b, chunk)
    float* all_scores;        // scores-only mode: [B, n_rows]
    const uint64_t* upper;    // optional [B]: only keys < upper[b] qualify (multi-pass for k > 2048)
};

__host__ __device__ inline size_t scan_img_bytes(int32_t n_cols) {
    return (((size_t)n_cols + 1) * 4 + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t scan_lds_bytes(int32_t n_cols) {
    return scan_img_bytes(n_cols) + (size_t)kWgCap * 8 + 16;
}

// Partial score of one row for this lane's packets p0+lg, p0+lg+G, ...
template <int G, int VM>
__device__ __forceinline__ float row_partial(const ScanArgs& a, const float* img, uint32_t p0, uint32_t p1, int lg) {
    float acc0 = 0.f, acc1 = 0.f;
    const uint32_t padw = (uint32_t)a.n_cols | ((uint32_t)a.n_cols << 16);
    for (uint32_t p = p0 + lg; p < p1; p += 2 * G) {
        const bool two = (p + G) < p1;
        const uint32_t pb = two ? p + G : p;
        uint4 ca = a.cols[p];
        uint4 cb = a.cols[pb];
        if (!two) cb = make_uint4(padw, padw, padw, padw);
        if constexpr (VM == VM_F32) {
            const float4* v = reinterpret_cast<const float4*>(a.vals);
            const float4 a0 = v[2 * (size_t)p], a1 = v[2 * (size_t)p + 1];
            const float4 b0 = v[2 * (size_t)pb], b1 = v[2 * (size_t)pb + 1];
            acc0 = fmaf(a0.x, img[ca.x & 0xFFFF], acc0);
            acc0 = fmaf(a0.y, img[ca.x >> 16], acc0);
            acc0 = fmaf(a0.z, img[ca.y & 0xFFFF], acc0);
            acc0 = fmaf(a0.w, img[ca.y >> 16], acc0);
            acc0 = fmaf(a1.x, img[ca.z & 0xFFFF], acc0);
            acc0 = fmaf(a1.y, img[ca.z >> 16], acc0);
            acc0 = fmaf(a1.z, img[ca.w & 0xFFFF], acc0);
            acc0 = fmaf(a1.w, img[ca.w >> 16], acc0);
            acc1 = fmaf(b0.x, img[cb.x & 0xFFFF], acc1);
            acc1 = fmaf(b0.y, img[cb.x >> 16], acc1);
            acc1 = fmaf(b0.z, img[cb.y & 0xFFFF], acc1);
            acc1 = fmaf(b0.w, img[cb.y >> 16], acc1);
            acc1 = fmaf(b1.x, img[cb.z & 0xFFFF], acc1);
            acc1 = fmaf(b1.y, img[cb.z >> 16], acc1);
            acc1 = fmaf(b1.z, img[cb.w & 0xFFFF], acc1);
            acc1 = fmaf(b1.w, img[cb.w >> 16], acc1);
        } else if constexpr (VM == VM_F16) {
            const uint4* v = reinterpret_cast<const uint4*>(a.vals);
            const uint4 va = v[p], vb = v[pb];
            const __half2* ha = reinterpret_cast<const __half2*>(&va);
            const __half2* hb = reinterpret_cast<const __half2*>(&vb);
            const uint32_t cwa[4] = {ca.x, ca.y, ca.z, ca.w};
            const uint32_t cwb[4] = {cb.x, cb.y, cb.z, cb.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float2 fa = __half22float2(ha[i]);
                const float2 fb = __half22float2(hb[i]);
                acc0 = fmaf(fa.x, img[cwa[i] & 0xFFFF], acc0);
                acc0 = fmaf(fa.y, img[cwa[i] >> 16], acc0);
                acc1 = fmaf(fb.x, img[cwb[i] & 0xFFFF], acc1);
                acc1 = fmaf(fb.y, img[cwb[i] >> 16], acc1);
            }
        } else {
            const uint32_t cwa[4] = {ca.x, ca.y, ca.z, ca.w};
            const uint32_t cwb[4] = {cb.x, cb.y, cb.z, cb.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc0 += img[cwa[i] & 0xFFFF];
                acc0 += img[cwa[i] >> 16];
                acc1 += img[cwb[i] & 0xFFFF];
                acc1 += img[cwb[i] >> 16];
            }
        }
    }
    return acc0 + acc1;
}

template <int G>
__device__ __forceinline__ float group_sum(float x) {
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

__device__ __forceinline__ void load_image(const ScanArgs& a, float* img, int qi, int tid) {
    const float* qrow = a.q + (size_t)qi * a.n_cols;
    for (int i = tid; i < a.n_cols; i += kScanThreads) img[i] = qrow[i];
    if (tid == 0) img[a.n_cols] = 0.f;
}

// ---- scores-only pass: writes the dense [B, N] matrix the reference materialises (tests) --------
template <int G, int VM>
__global__ __launch_bounds__(kScanThreads) void csr_scan_scores(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* img = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int RPW = 64 / G;
    const int g = lane / G, lg = lane % G;
    const int64_t items = (int64_t)a.B * a.nchunk;
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int qi = (int)(item / a.nchunk), c = (int)(item % a.nchunk);
        const int64_t r0 = (int64_t)c * a.rows_per_chunk;
        const int64_t r1 = min(a.n_rows, r0 + a.rows_per_chunk);
        __syncthreads();
        load_image(a, img, qi, tid);
        __syncthreads();
        for (int64_t rb = r0 + (int64_t)w * RPW; rb < r1; rb += (int64_t)kScanWaves * RPW) {
            const int64_t row = rb + g;
            float acc = 0.f;
            if (row < r1) acc = row_partial<G, VM>(a, img, a.pk_ptr[row], a.pk_ptr[row + 1], lg);
            acc = group_sum<G>(acc);
            if (lg == 0 && row < r1) a.all_scores[(size_t)qi * a.n_rows + row] = acc;
        }
    }
}

// ---- fused scoring + top-k, k <= 128: wave-private candidate buffers, no barrier in the scan -----
// Each wave keeps its own <=256 candidate keys in LDS and prunes them to the best k with an
// in-register bitonic network; a wave's k-th best key is a lower bound of the global k-th best, so
// waves share the tightest bound through one LDS word and drop every row that cannot qualify.
template <int G, int VM>
__global__ __launch_bounds__(kScanThreads) void csr_scan_topk_wave(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* img = reinterpret_cast<float*>(smem);
    uint64_t* cand = reinterpret_cast<uint64_t*>(smem + scan_img_bytes(a.n_cols));
    unsigned long long* tau_sh = reinterpret_cast<unsigned long long*>(cand + kWgCap);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int RPW = 64 / G;
    const int g = lane / G, lg = lane % G;
    const int K = a.k;
    uint64_t* mybuf = cand + w * kWaveCap;
    const int64_t items = (int64_t)a.B * a.nchunk;

    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int qi = (int)(item / a.nchunk), c = (int)(item % a.nchunk);
        const int64_t r0 = (int64_t)c * a.rows_per_chunk;
        const int64_t r1 = min(a.n_rows, r0 + a.rows_per_chunk);
        __syncthreads();
        load_image(a, img, qi, tid);
        if (tid == 0) *tau_sh = 0ull;
        __syncthreads();
        const uint64_t upper = a.upper ? a.upper[qi] : ~0ull;
        uint64_t tau = 0;
        int cnt = 0;
        for (int64_t rb = r0 + (int64_t)w * RPW; rb < r1; rb += (int64_t)kScanWaves * RPW) {
            const int64_t row = rb + g;
            float acc = 0.f;
            if (row < r1) acc = row_partial<G, VM>(a, img, a.pk_ptr[row], a.pk_ptr[row + 1], lg);
            acc = group_sum<G>(acc);
            const uint64_t key = make_key(acc, (uint32_t)row);
            const uint64_t ts = __hip_atomic_load(tau_sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            tau = ts > tau ? ts : tau;
            bool pass = (lg == 0) && (row < r1) && (key > tau) && (key < upper);
            uint64_t m = __ballot(pass);
            if (m) {
                int n = __popcll(m);
                if (cnt + n > kWaveCap) {
                    // prune: keep the K best of this wave's candidates
                    uint64_t k4[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = (r << 6) | lane;
                        k4[r] = e < cnt ? mybuf[e] : 0ull;
                    }
                    wave_sort256_desc(k4, lane);
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const int e = (r << 6) | lane;
                        if (e < K) mybuf[e] = k4[r];
                    }
                    const uint64_t kth_src = ((K - 1) >> 6) ? k4[1] : k4[0];
                    const uint64_t kth = __shfl(kth_src, (K - 1) & 63, 64);
                    if (lane == 0) atomicMax(tau_sh, (unsigned long long)kth);
                    tau = kth > tau ? kth : tau;
                    cnt = K;
                    pass = pass && (key > tau);
                    m = __ballot(pass);
                    n = __popcll(m);
                }
                if (pass) mybuf[cnt + __popcll(m & ((1ull << lane) - 1ull))] = key;
                cnt += n;
            }
        }
        for (int i = cnt + lane; i < kWaveCap; i += 64) mybuf[i] = 0ull;
        wg_sort_desc<kScanThreads>(cand, kWgCap, tid);      // starts with a barrier
        uint64_t* out = a.cand + ((size_t)qi * a.nchunk + c) * (size_t)K;
        for (int i = tid; i < K; i += kScanThreads) out[i] = cand[i];
    }
}

// ---- fused scoring + top-k, 128 < k <= 2048: one shared 4096-slot buffer, barrier per superbatch --
template <int G, int VM>
__global__ __launch_bounds__(kScanThreads) void csr_scan_topk_shared(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* img = reinterpret_cast<float*>(smem);
    uint64_t* cand = reinterpret_cast<uint64_t*>(smem + scan_img_bytes(a.n_cols));
    int* cnt_sh = reinterpret_cast<int*>(cand + kWgCap);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int RPW = 64 / G;
    constexpr int RPI = kScanWaves * RPW;                 // rows per workgroup iteration
    constexpr int SB = (kWgCap - kMaxKShared) / RPI;      // iterations between prune checks
    const int g = lane / G, lg = lane % G;
    const int K = a.k;
    const int64_t items = (int64_t)a.B * a.nchunk;

    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int qi = (int)(item / a.nchunk), c = (int)(item % a.nchunk);
        const int64_t r0 = (int64_t)c * a.rows_per_chunk;
        const int64_t r1 = min(a.n_rows, r0 + a.rows_per_chunk);
        __syncthreads();
        load_image(a, img, qi, tid);
        if (tid == 0) *cnt_sh = 0;
        __syncthreads();
        const uint64_t upper = a.upper ? a.upper[qi] : ~0ull;
        uint64_t tau = 0;
        const int64_t iters = (r1 - r0 + RPI - 1) / RPI;
        for (int64_t it0 = 0; it0 < iters; it0 += SB) {
            const int64_t it1 = min(iters, it0 + SB);
            for (int64_t it = it0; it < it1; ++it) {
                const int64_t row = r0 + it * RPI + (int64_t)w * RPW + g;
                float acc = 0.f;
                if (row < r1) acc = row_partial<G, VM>(a, img, a.pk_ptr[row], a.pk_ptr[row + 1], lg);
                acc = group_sum<G>(acc);
                const uint64_t key = make_key(acc, (uint32_t)row);
                const bool pass = (lg == 0) && (row < r1) && (key > tau) && (key < upper);
                const uint64_t m = __ballot(pass);
                if (m) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(cnt_sh, __popcll(m));
                    base = __shfl(base, 0, 64);
                    if (pass) cand[base + __popcll(m & ((1ull << lane) - 1ull))] = key;
                }
            }
            __syncthreads();
            const int cnt = *cnt_sh;
            const bool last = it1 >= iters;
            if (last || cnt > kMaxKShared) {                // uniform: cnt read after the barrier
                for (int i = cnt + tid; i < kWgCap; i += kScanThreads) cand[i] = 0ull;
                wg_sort_desc<kScanThreads>(cand, kWgCap, tid);
                if (!last && cnt > K) {
                    tau = cand[K - 1];
                    __syncthreads();
                    if (tid == 0) *cnt_sh = K;
                }
            }
            __syncthreads();
        }
        if (iters == 0) {                                   // empty chunk: emit sentinels
            for (int i = tid; i < kWgCap; i += kScanThreads) cand[i] = 0ull;
            __syncthreads();
        }
        uint64_t* out = a.cand + ((size_t)qi * a.nchunk + c) * (size_t)K;
        for (int i = tid; i < K; i += kScanThreads) out[i] = cand[i];
    }
}

// ---- exact row score: fp32 products summed in fp64 (the numerics of the multi-query scan and of the postings walk) ------
// One wave per row, a lane takes whole packets p0 + lane, p0 + lane + 64, ...; `weight(c)` returns the query weight of column c.
template <int VM, class W>
__device__ __forceinline__ double row_sum_f64(const uint32_t* pk_ptr, const uint4* cols, const void* vals, uint32_t row, int lane, W weight) {
    const uint32_t p0 = pk_ptr[row], p1 = pk_ptr[row + 1];
    double sum = 0.0;
    for (uint32_t p = p0 + lane; p < p1; p += 64) {
        const uint4 cw = cols[p];
        const uint32_t cwv[4] = {cw.x, cw.y, cw.z, cw.w};
        float v[8];
        if constexpr (VM == VM_F32) {
            const float4* vp = reinterpret_cast<const float4*>(vals);
            const float4 v0 = vp[2 * (size_t)p], v1 = vp[2 * (size_t)p + 1];
            v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
        } else if constexpr (VM == VM_F16) {
            const uint4 hv = reinterpret_cast<const uint4*>(vals)[p];
            const __half2* h = reinterpret_cast<const __half2*>(&hv);
#pragma unroll
            for (int t = 0; t < 4; ++t) { const float2 f = __half22float2(h[t]); v[2 * t] = f.x; v[2 * t + 1] = f.y; }
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = 1.f;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const uint32_t c = (t & 1) ? (cwv[t >> 1] >> 16) : (cwv[t >> 1] & 0xFFFFu);
            sum += (double)(weight(c) * v[t]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    return sum;
}

// ---- exact pass for queries picked on the device (sel[i].x, i < sel_n[0]): the filter-and-refine search's unproven queries when
// the postings copy is lossy.  One query per pass like csr_scan_topk_shared, one wave per row, fp64 row sums.
template <int VM>
__global__ __launch_bounds__(kScanThreads) void exact_scan_topk_kernel(ScanArgs a, const int2* sel, const int32_t* sel_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* img = reinterpret_cast<float*>(smem);
    uint64_t* cand = reinterpret_cast<uint64_t*>(smem + scan_img_bytes(a.n_cols));
    int* cnt_sh = reinterpret_cast<int*>(cand + kWgCap);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int SB = (kWgCap - kMaxKShared) / kScanWaves;      // iterations between prune checks
    const int K = a.k;
    const int64_t items = (int64_t)sel_n[0] * a.nchunk;
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int qi = sel[item / a.nchunk].x, c = (int)(item % a.nchunk);
        const int64_t r0 = (int64_t)c * a.rows_per_chunk;
        const int64_t r1 = min(a.n_rows, r0 + a.rows_per_chunk);
        __syncthreads();
        load_image(a, img, qi, tid);
        if (tid == 0) *cnt_sh = 0;
        __syncthreads();
        uint64_t tau = 0;
        const int64_t iters = (r1 - r0 + kScanWaves - 1) / kScanWaves;
        for (int64_t it0 = 0; it0 < iters; it0 += SB) {
            const int64_t it1 = min(iters, it0 + SB);
            for (int64_t it = it0; it < it1; ++it) {
                const int64_t row = r0 + it * kScanWaves + w;
                if (row < r1) {
                    const double sum = row_sum_f64<VM>(a.pk_ptr, a.cols, a.vals, (uint32_t)row, lane, [&](uint32_t col) { return img[col]; });
                    const uint64_t key = make_key((float)sum, (uint32_t)row);
                    if (lane == 0 && key > tau) cand[atomicAdd(cnt_sh, 1)] = key;
                }
            }
            __syncthreads();
            const int cnt = *cnt_sh;
            const bool last = it1 >= iters;
            if (last || cnt > kMaxKShared) {                // uniform: cnt read after the barrier
                for (int i = cnt + tid; i < kWgCap; i += kScanThreads) cand[i] = 0ull;
                wg_sort_desc<kScanThreads>(cand, kWgCap, tid);
                if (!last && cnt > K) {
                    tau = cand[K - 1];
                    __syncthreads();
                    if (tid == 0) *cnt_sh = K;
                }
            }
            __syncthreads();
        }
        if (iters == 0) {
            for (int i = tid; i < kWgCap; i += kScanThreads) cand[i] = 0ull;
            __syncthreads();
        }
        uint64_t* out = a.cand + ((size_t)qi * a.nchunk + c) * (size_t)K;
        for (int i = tid; i < K; i += kScanThreads) out[i] = cand[i];
    }
}

// ---- merge: per query, [n_lists * k] keys (each list sorted or not) -> top-k ids + scores --------
struct MergeArgs {
    const uint64_t* cand;     // [B, n_cand]
    int64_t n_cand;
    int32_t B;
    int32_t k;
    int64_t id_offset;
    int64_t* out_ids;         // [B, out_ld], this call fills columns [col0, col0 + k)
    float* out_scores;        // same shape
    int64_t out_ld;
    int32_t col0;
    uint64_t* upper_out;      // optional [B]: receives the k-th key (next pass's exclusive upper bound)
    const uint64_t* upper_in; // optional [B]: only keys < upper_in[b] take part (passes after the first when k > 2048)
    int32_t run_len;          // > 0: the list is a sequence of descending-sorted runs of this length (per-chunk top-k lists)
    const int2* sel;          // optional device list of queries (x = query) with its length in sel_n[0]: merge only those
    const int32_t* sel_n;
};

// Leaves the K largest of src[0, n_cand) (0 = empty slot) in buf[0, K), sorted descending; buf holds kWgCap keys.
__device__ __forceinline__ void merge_select(const uint64_t* src, int64_t n_cand, int run_len, int K, const uint64_t* upper_in, uint64_t* buf,
                                             int* cnt_sh, int tid) {
    int64_t consumed = 0;
    int have = 0;                                        // buf[0..have) = best so far
    __syncthreads();
    // Sorted runs (many row chunks, small batch): the first p = ceil(K / runs) keys of every run hold >= K keys, so
    // their K-th largest is a lower bound of the answer's K-th key; typically only a few hundred candidates pass
    // it, and ONE small sort replaces n_cand / 4096 full-buffer rounds.
    bool done = false;
    if (run_len > 0 && !upper_in && n_cand > kWgCap && n_cand % run_len == 0) {
        const int runs = (int)(n_cand / run_len);
        const int p = min(run_len, (K + runs - 1) / runs);
        const int heads = runs * p;
        if (heads >= K && heads <= kWgCap) {
            int n2 = 64;
            while (n2 < heads) n2 <<= 1;
            for (int i = tid; i < n2; i += kScanThreads) buf[i] = i < heads ? src[(size_t)(i / p) * run_len + (i % p)] : 0ull;
            wg_sort_desc<kScanThreads>(buf, n2, tid);
            const uint64_t bound = buf[K - 1];
            __syncthreads();
            if (tid == 0) *cnt_sh = 0;
            __syncthreads();
            for (int64_t i = tid; i < n_cand; i += kScanThreads) {
                const uint64_t key = src[i];
                if (key >= bound && key != 0ull) {
                    const int pos = atomicAdd(cnt_sh, 1);
                    if (pos < kWgCap) buf[pos] = key;
                }
            }
            __syncthreads();
            const int cnt = *cnt_sh;
            if (cnt <= kWgCap) {
                int n3 = 64;
                while (n3 < cnt) n3 <<= 1;
                n3 = max(n3, 64);
                while (n3 < K) n3 <<= 1;
                for (int i = cnt + tid; i < n3; i += kScanThreads) buf[i] = 0ull;
                wg_sort_desc<kScanThreads>(buf, n3, tid);
                done = true;
            }
            __syncthreads();
        }
    }
    if (!done) do {
        const int room = kWgCap - have;
        const int64_t take = min((int64_t)room, n_cand - consumed);
        for (int i = tid; i < room; i += kScanThreads) {
            uint64_t key = i < take ? src[consumed + i] : 0ull;
            if (upper_in && key >= upper_in[0]) key = 0ull;
            buf[have + i] = key;
        }
        consumed += take;
        wg_sort_desc<kScanThreads>(buf, kWgCap, tid);
        have = K;
    } while (consumed < n_cand);
}

template <int UNUSED>
__global__ __launch_bounds__(kScanThreads) void merge_topk_kernel(MergeArgs a) {
    __shared__ uint64_t buf[kWgCap];
    __shared__ int cnt_sh;
    const int tid = threadIdx.x;
    const int K = a.k;                                       // K <= kMaxKShared
    const int n_q = a.sel_n ? a.sel_n[0] : a.B;              // optional query list (device): only those queries are merged
    for (int bi = blockIdx.x; bi < n_q; bi += gridDim.x) {
        const int b = a.sel ? a.sel[bi].x : bi;
        merge_select(a.cand + (size_t)b * a.n_cand, a.n_cand, a.run_len, K, a.upper_in ? a.upper_in + b : nullptr, buf, &cnt_sh, tid);
        for (int i = tid; i < K; i += kScanThreads) {
            const uint64_t key = buf[i];
            a.out_ids[(size_t)b * a.out_ld + a.col0 + i] = (int64_t)key_row(key) + a.id_offset;
            a.out_scores[(size_t)b * a.out_ld + a.col0 + i] = key_score(key);
            if (a.upper_out && i == K - 1) a.upper_out[b] = key;
        }
        __syncthreads();
    }
}

// ---- select: per query, n_cand keys (n_cand large, e.g. every row of a dense index) -> top-k ------------
// MSD radix refinement on LDS histograms (12-bit digits): each pass over the (L2-resident) keys fixes
// 12 more bits of the k-th largest key; as soon as the keys at or above the current bin fit the 4096-slot
// LDS buffer they are collected and sorted.  Keys are distinct (the low word is the row id), so the
// refinement always terminates.  Same MergeArgs / output convention as merge_topk_kernel.
template <int UNUSED>
__global__ __launch_bounds__(kScanThreads) void select_topk_kernel(MergeArgs a) {
    __shared__ uint64_t buf[kWgCap];
    __shared__ int hist[4096];
    __shared__ int scratch[32];
    __shared__ int s_bin, s_above, s_pop, s_cnt;
    const int tid = threadIdx.x;
    const int K = a.k;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const uint64_t* src = a.cand + (size_t)b * a.n_cand;
        const uint64_t upper = a.upper_in ? a.upper_in[b] : ~0ull;
        uint64_t prefix = 0;
        int pbits = 0, need = K, total_above = 0;
        for (;;) {
            const int width = min(12, 64 - pbits);
            const int bins = 1 << width;
            for (int i = tid; i < bins; i += kScanThreads) hist[i] = 0;
            __syncthreads();
            for (int64_t i = tid; i < a.n_cand; i += kScanThreads) {
                uint64_t key = src[i];
                if (key >= upper) key = 0ull;
                if (pbits == 0 || (key >> (64 - pbits)) == prefix) atomicAdd(&hist[(int)((key >> (64 - pbits - width)) & (uint64_t)(bins - 1))], 1);
            }
            __syncthreads();
            // thread t owns 4 bins in DESCENDING order: bins-1-4t .. bins-4-4t
            int mine = 0, hb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int bin = bins - 1 - (4 * tid + j);
                hb[j] = bin >= 0 ? hist[bin] : 0;
                mine += hb[j];
            }
            int above = block_excl_scan(mine, scratch, tid, nullptr);       // keys in strictly higher bins
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int bin = bins - 1 - (4 * tid + j);
                if (bin >= 0 && above < need && need <= above + hb[j]) { s_bin = bin; s_above = above; s_pop = hb[j]; }
                above += hb[j];
            }
            __syncthreads();
            total_above += s_above;
            need -= s_above;
            prefix = (prefix << width) | (uint64_t)s_bin;
            pbits += width;
            const int pop = s_pop;
            __syncthreads();
            if (total_above + pop <= kWgCap || pbits >= 64) break;
        }
        // collect keys whose top pbits are >= prefix
        if (tid == 0) s_cnt = 0;
        for (int i = tid; i < kWgCap; i += kScanThreads) buf[i] = 0ull;
        __syncthreads();
        for (int64_t i = tid; i < a.n_cand; i += kScanThreads) {
            uint64_t key = src[i];
            if (key >= upper) key = 0ull;
            if ((pbits >= 64 ? key : (key >> (64 - pbits))) >= prefix) {
                const int pos = atomicAdd(&s_cnt, 1);
                if (pos < kWgCap) buf[pos] = key;
            }
        }
        wg_sort_desc<kScanThreads>(buf, kWgCap, tid);
        for (int i = tid; i < K; i += kScanThreads) {
            const uint64_t key = buf[i];
            a.out_ids[(size_t)b * a.out_ld + a.col0 + i] = (int64_t)key_row(key) + a.id_offset;
            a.out_scores[(size_t)b * a.out_ld + a.col0 + i] = key_score(key);
            if (a.upper_out && i == K - 1) a.upper_out[b] = key;
        }
        __syncthreads();
    }
}

}  // namespace vs
