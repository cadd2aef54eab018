// sparsify.hip -- src/ir/utils/sparse.py (elu1p, build_topk_mask, build_bow_mask) and the mask
// stage of VDREncoder.embed (src/ir/encoder/vdr.py:152-169), plus Tensor.to_sparse_csr()
// (retriever.py:304) and the encoder head's pooling tail (vdr.py:73-75).
//
// One workgroup per embedding row: the row (V = 29 523 fp32 = 118 KB) sits in LDS as order keys; the
// k-th largest value is found by a 4-pass 8-bit radix select on lane-split LDS histograms; ties at the
// threshold go to the lowest column ids (torch.topk leaves them unspecified).
#include "common.h"
#include "topk_keys.h"
#include "dense_csr.h"

#include <algorithm>

using namespace vs;

namespace {


__device__ __forceinline__ float elu1p_dev(float x) { return x > 0.f ? x + 1.0f : expm1f(x) + 1.0f; }

__global__ void elu1p_kernel(const float* x, int64_t n, float* out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = elu1p_dev(x[i]);
}

// out[b, c] = elu1p(max_l logits[b, l, c])   (elu1p is monotone: == max_l elu1p(logits), vdr.py:73-75)
__global__ void head_pool_kernel(const float* logits, int32_t B, int32_t L, int32_t V, float* out) {
    const int64_t n = (int64_t)B * V;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / V, c = i % V;
        const float* p = logits + (size_t)b * L * V + c;
        float m = -INFINITY;
        for (int l = 0; l < L; ++l) m = fmaxf(m, p[(size_t)l * V]);
        out[i] = elu1p_dev(m);
    }
}

// out[b, c] = mean of the `topk` largest elu1p(logits[b, :, c])   (vdr.py:76-79: pooling = "mean" with pooling_topk)
constexpr int kPoolTopkMax = 32;
__global__ void head_pool_mean_topk_kernel(const float* logits, int32_t B, int32_t L, int32_t V, int32_t topk, float* out) {
    const int64_t n = (int64_t)B * V;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / V, c = i % V;
        const float* p = logits + (size_t)b * L * V + c;
        float top[kPoolTopkMax];                            // ascending: top[0] = smallest kept
        int have = 0;
        for (int l = 0; l < L; ++l) {
            const float v = p[(size_t)l * V];
            if (have < topk) {
                int j = have++;
                while (j > 0 && top[j - 1] > v) { top[j] = top[j - 1]; --j; }
                top[j] = v;
            } else if (v > top[0]) {
                int j = 0;
                while (j + 1 < topk && top[j + 1] < v) { top[j] = top[j + 1]; ++j; }
                top[j] = v;
            }
        }
        float sum = 0.f;
        for (int j = have - 1; j >= 0; --j) sum += elu1p_dev(top[j]);       // largest first: torch.topk(...).values.mean(1) adds in that order
        out[i] = sum / (float)have;
    }
}

struct MaskArgs {
    float* emb;            // [B, V] (ld) in/out; may be null when only `mask` is wanted from x
    const float* x;        // source values (== emb for in-place)
    int64_t ld;
    const int64_t* ids;    // [B, L] token ids or null
    int32_t B, L, V, vocab, shift;
    int32_t topk;          // >0: top-k; 0: none; <0: all
    int activate_lexical;
    int bow;
    uint8_t* mask;         // optional [B, V] output (build_topk_mask)
    int* flags;            // |1: token id out of range, |2: a row kept more elements than its slot run holds
    // CSR emission (mask_rows_fast_kernel only): the kept non-zero elements of row b as (column, value) pairs in column order at
    // slot_cols / slot_vals [b * slot_cap ...], their number in row_nnz[b]; null: off
    int64_t* row_nnz;
    int32_t* slot_cols;
    float* slot_vals;
    int32_t slot_cap;
    // ... and the CSR arrays the workgroups' runs end up in (the kernel's last step): rowptr [B + 1], cols / vals [csr_cap]; behind
    // `flags`: flags[1] = the workgroup ticket, 8-byte words 32 ..: the workgroups' totals -- zeroed by the host before the launch
    int64_t* csr_rowptr;
    int32_t* csr_cols;
    float* csr_vals;
    int64_t csr_cap;
};

// block-wide sum of one 64-bit value per thread (kSpThreads threads); red: kSpThreads / 64 slots in LDS
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long* red, int tid) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();                                   // the previous round's readers are done with `red`
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    unsigned long long tot = 0;
#pragma unroll
    for (int i = 0; i < kSpThreads / 64; ++i) tot += red[i];
    return tot;
}

// One workgroup per row.  Global traffic is one coalesced read and one coalesced write of the row; the
// select runs on the LDS copy (4-pass radix select, see below).  Ties at the threshold go to the lowest
// columns (torch.topk leaves them open).
__global__ __launch_bounds__(kSpThreads) void mask_rows_kernel(MaskArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* keys = reinterpret_cast<uint32_t*>(smem);                       // [V]
    const int vwords = (a.V + 31) / 32;
    uint32_t* lex = keys + ((a.V + 3) & ~3);                                   // [vwords] lexical bitmap
    uint32_t* tie = lex + ((vwords + 3) & ~3);                                 // [vwords] threshold-valued columns that are selected
    unsigned long long* red = reinterpret_cast<unsigned long long*>(tie + ((vwords + 3) & ~3));   // [16]
    int* scratch = reinterpret_cast<int*>(red + kSpThreads / 64);              // [32]
    int* sel_sh = scratch + 32;                                                // [4]
    int* hist = sel_sh + 4;                                                    // [256 * 8]
    const int tid = threadIdx.x;
    const int seg = (a.V + kSpThreads - 1) / kSpThreads;

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        __syncthreads();
        const float* xr = a.x ? a.x + (size_t)b * a.ld : nullptr;
        const bool need_keys = !a.bow && a.topk > 0;
        const bool need_vals = !a.bow && a.emb;                                // `emb *= mask` rewrites from the LDS copy
        if (need_keys || need_vals) for (int i = tid; i < a.V; i += kSpThreads) keys[i] = flip_f32(xr[i]);
        for (int i = tid; i < vwords; i += kSpThreads) { lex[i] = 0; tie[i] = 0; }
        __syncthreads();
        if (a.ids && (a.bow || a.activate_lexical)) {
            int bad = 0;
            for (int l = tid; l < a.L; l += kSpThreads) {
                const int64_t t = a.ids[(size_t)b * a.L + l];
                if (t < 0 || t >= a.vocab) bad = 1;
                else if (t >= a.shift) atomicOr(&lex[(t - a.shift) >> 5], 1u << ((t - a.shift) & 31));
            }
            if (bad) atomicOr(a.flags, 1);
        }
        __syncthreads();

        uint32_t T = 0;          // k-th largest key
        bool all_eq = true;      // every element equal to T is selected
        if (need_keys) {
            // 4-pass MSD radix select, 8 bits per pass.  Histogram bins are split 8 ways by lane: the top byte of an
            // order key is the fp32 exponent, the same for nearly every element of a row, and 64 lanes adding to ONE LDS
            // address serialise; the bin holding the k-th key is found with a block scan over the 256 bins (descending).
            uint32_t prefix = 0, pmask = 0;
            int remaining = a.topk, n_eq = 0;
            for (int shift = 24; shift >= 0; shift -= 8) {
                for (int i = tid; i < 256 * 8; i += kSpThreads) hist[i] = 0;
                __syncthreads();
                for (int i = tid; i < a.V; i += kSpThreads) {
                    const uint32_t kx = keys[i];
                    if ((kx & pmask) == prefix) atomicAdd(&hist[((kx >> shift) & 255u) * 8 + (tid & 7)], 1);
                }
                __syncthreads();
                int h = 0;                                   // thread t < 256 owns bin 255 - t
                if (tid < 256) {
                    const int* hb = hist + (255 - tid) * 8;
                    h = hb[0] + hb[1] + hb[2] + hb[3] + hb[4] + hb[5] + hb[6] + hb[7];
                }
                const int above = block_excl_scan(h, scratch, tid, nullptr);      // keys in strictly higher bins
                if (tid < 256 && above < remaining && remaining <= above + h) { sel_sh[0] = 255 - tid; sel_sh[1] = above; sel_sh[2] = h; }
                __syncthreads();
                prefix |= (uint32_t)sel_sh[0] << shift;
                pmask |= 255u << shift;
                remaining -= sel_sh[1];
                n_eq = sel_sh[2];
                __syncthreads();
            }
            T = prefix;
            const int r_eq = remaining;                                         // >= 1 elements equal to T are selected
            all_eq = n_eq == r_eq;
            if (!all_eq) {       // rare: rank the equal elements by column (contiguous segment per thread + block scan)
                const int i0 = tid * seg, i1 = min(a.V, i0 + seg);
                int my_eq = 0;
                for (int i = i0; i < i1; ++i) my_eq += keys[i] == T;
                int before = block_excl_scan(my_eq, scratch, tid, nullptr);
                for (int i = i0; i < i1; ++i)
                    if (keys[i] == T) {
                        if (before < r_eq) atomicOr(&tie[i >> 5], 1u << (i & 31));
                        ++before;
                    }
                __syncthreads();
            }
        }
        float bow_val = 1.0f;
        if (a.bow < 0) {         // bow with L2 normalisation: value = 1 / max(sqrt(count), 1e-12)
            unsigned long long c = 0;
            for (int i = tid; i < vwords; i += kSpThreads) c += __popc(lex[i]);
            c = block_sum_u64(c, red, tid);
            bow_val = 1.0f / fmaxf(sqrtf((float)c), 1e-12f);
        }
        for (int i = tid; i < a.V; i += kSpThreads) {
            const bool lx = (lex[i >> 5] >> (i & 31)) & 1u;
            bool sel;
            if (a.bow) sel = lx;
            else {
                bool tk;
                if (a.topk == 0) tk = false;
                else if (a.topk < 0) tk = true;
                else {
                    const uint32_t kx = keys[i];
                    tk = kx > T || (kx == T && (all_eq || ((tie[i >> 5] >> (i & 31)) & 1u)));
                }
                sel = tk || (a.activate_lexical && lx);
            }
            if (a.mask) a.mask[(size_t)b * a.V + i] = sel ? 1 : 0;
            if (a.emb) {
                float* e = a.emb + (size_t)b * a.ld + i;
                if (a.bow) *e = sel ? bow_val : 0.f;
                else if (!sel) *e = unflip_f32(keys[i]) * 0.f;         // `emb *= mask` (vdr.py:169): x * 0 keeps NaN / sign like torch
            }
        }
    }
}

#include "mask_rows_fast.h"

int check_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(VS_ENODEVICE, "no HIP device visible: libvsearch_hip has no CPU fallback");
    }
    if (device < 0 || device >= ndev) return fail(VS_EINVAL, "device %d out of range", device);
    VS_HIP(hipSetDevice(device));
    return VS_OK;
}

size_t mask_lds_bytes(int V) {
    const int vwords = (V + 31) / 32;
    return ((size_t)((V + 3) & ~3) + 2 * (size_t)((vwords + 3) & ~3) + 2 * (kSpThreads / 64) + 32 + 4 + 256 * 8) * 4;
}

// Runs mask_rows_kernel with host/device staging of emb / ids / mask.
int run_mask(MaskArgs a, const float* x_in, float* emb_io, const int64_t* ids, uint8_t* mask_out, int device, hipStream_t s) {
    VS_TRY(check_device(device));
    const size_t lds = mask_lds_bytes(a.V);
    if (lds > 160 * 1024) return fail(VS_EUNSUPPORTED, "V = %d needs %zu B of LDS (> 160 KiB)", a.V, lds);
    DevBuf st_x, st_ids, st_mask;
    DevBuf& flags = device_scratch(device, kScratchFlags);
    VS_TRY(flags.reserve(256));
    if (ids) VS_HIP(hipMemsetAsync(flags.p, 0, 4, s));
#ifdef MR_TIMING
    VS_HIP(hipMemsetAsync(flags.p, 0, 256, s));
#endif
    a.flags = flags.as<int>();
    const size_t row_span = a.B > 0 ? ((size_t)(a.B - 1) * a.ld + a.V) * 4 : 0;
    const float* src = emb_io ? emb_io : x_in;
    const bool x_host = src && !is_device_ptr(src);
    if (src) {
        const void* d = nullptr;
        VS_TRY(to_device(src, row_span, st_x, s, &d));
        a.x = (const float*)d;
        a.emb = emb_io ? (float*)d : nullptr;
    }
    if (ids) {
        const void* d = nullptr;
        VS_TRY(to_device(ids, (size_t)a.B * a.L * 8, st_ids, s, &d));
        a.ids = (const int64_t*)d;
    }
    const bool m_host = mask_out && !is_device_ptr(mask_out);
    if (mask_out) {
        if (m_host) { VS_TRY(st_mask.alloc((size_t)a.B * a.V)); a.mask = st_mask.as<uint8_t>(); }
        else a.mask = mask_out;
    }
    ProfScope prof("mask_rows", s);
    if (!a.bow && a.topk > 0 && a.x && a.V <= kMrCols) {
        // the encoder's case (mask_rows_fast.h): persistent workgroups, the next row's loads in flight through a row's select
        void (*kern)(MaskArgs) = a.V <= 4 * kMrStep ? mask_rows_fast_kernel<4> : a.V <= 8 * kMrStep ? mask_rows_fast_kernel<8> : a.V <= 15 * kMrStep ? mask_rows_fast_kernel<15> : mask_rows_fast_kernel<16>;
        int cus = 0;
        VS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        VS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mask_fast_lds_bytes()));
        hipLaunchKernelGGL(kern, dim3(std::max(1, std::min(a.B, cus))), dim3(kMrThreads), mask_fast_lds_bytes(), s, a);
    } else {
        VS_HIP(hipFuncSetAttribute((const void*)mask_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(mask_rows_kernel, dim3(std::min(a.B, 1024)), dim3(kSpThreads), lds, s, a);
    }
    VS_HIP(hipGetLastError());
#ifdef MR_TIMING
    {
        unsigned long long h[32];
        VS_HIP(hipMemcpyAsync(h, flags.p, 256, hipMemcpyDeviceToHost, s));
        VS_HIP(hipStreamSynchronize(s));
        static int calls = 0;
        if (++calls == 5) {
            const double rows = (double)a.B;
            fprintf(stderr, "[vsearch_hip] mask stage, cycles per row (wave 0): barrier+wait %.0f, pack %.0f, issue+clear+lex %.0f, pass A %.0f, pick A %.0f, pass B + pick B %.0f, masks %.0f, candidates %.0f (list %.0f, barrier %.0f, ranking %.0f, barrier %.0f), write %.0f\n",
                    h[1] / rows, h[2] / rows, h[3] / rows, h[4] / rows, h[5] / rows, h[6] / rows, h[7] / rows, h[8] / rows, h[12] / rows, h[13] / rows, h[14] / rows, h[15] / rows, h[9] / rows);
        }
    }
#endif
    if (emb_io && x_host) VS_HIP(hipMemcpyAsync(emb_io, st_x.p, row_span, hipMemcpyDeviceToHost, s));
    if (m_host) VS_HIP(hipMemcpyAsync(mask_out, st_mask.p, (size_t)a.B * a.V, hipMemcpyDeviceToHost, s));
    if (ids) {                     // a bad token id must surface as an error (the reference's scatter_ asserts on it)
        int hflags = 0;
        VS_HIP(hipMemcpyAsync(&hflags, flags.p, 4, hipMemcpyDeviceToHost, s));
        VS_HIP(hipStreamSynchronize(s));
        if (hflags & 1) return fail(VS_EINVAL, "token id out of range [0, %d)", a.vocab);
    } else if (x_host || m_host || st_x.p || !s) {
        VS_HIP(hipStreamSynchronize(s));                 // host buffers / staging die with this call; device-only + stream: asynchronous
    }
    return VS_OK;
}

// ---- rerank (retriever.py:137-147): scores[b, j] = <p_emb[b * k + j, :], q[b, :]>, then a stable descending sort ------------
// One wave per re-embedded passage; fp32 products summed in fp64 (the library's exact numerics); the passage rows are ~97 %
// zeros, the query row comes from L2.  `row0` = index (b * k + j) of the chunk's first row: the re-embedding can be streamed in
// batches, the dense [B * k, V] tensor of the reference never has to exist.
// A lane takes 4 consecutive columns per step (16-byte loads of the fp32 passage row and of the query row; a row's first element only
// has to be 4-byte aligned) and the steps are unrolled 4 deep: 4 KB of the row in flight per wave, where the one-dword-a-lane loop
// this replaces kept 256 B and moved 2.4 TB/s.  A zero passage element contributes nothing whatever the query holds (0 * inf).
template <class T>
__global__ __launch_bounds__(256) void rerank_scores_kernel(const T* p, int64_t ldp, int64_t n_rows, int64_t row0, const float* q, int64_t ldq, int32_t k,
                                                            int32_t V, float* scores) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n_rows; r += (int64_t)gridDim.x * 4) {
        const int64_t g = row0 + r;
        const T* pr = p + (size_t)r * ldp;
        const float* qr = q + (size_t)(g / k) * ldq;
        double sum = 0.0;
        auto add = [&](float v, float w) { sum += (double)(v != 0.f ? v * w : 0.f); };
        int c = lane * 4;
        if constexpr (sizeof(T) == 4) {
            constexpr int U = 4;
            for (; c + (U - 1) * 256 + 4 <= V; c += U * 256) {
                float4 pv[U], qv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    __builtin_memcpy(&pv[u], pr + c + u * 256, 16);            // (dword-aligned 16-byte loads)
                    __builtin_memcpy(&qv[u], qr + c + u * 256, 16);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) { add(pv[u].x, qv[u].x); add(pv[u].y, qv[u].y); add(pv[u].z, qv[u].z); add(pv[u].w, qv[u].w); }
            }
            for (; c + 4 <= V; c += 256) {
                float4 pv, qv;
                __builtin_memcpy(&pv, pr + c, 16);
                __builtin_memcpy(&qv, qr + c, 16);
                add(pv.x, qv.x); add(pv.y, qv.y); add(pv.z, qv.z); add(pv.w, qv.w);
            }
        }
        // fp16 rows, and the last V % 4 columns: element by element
        if constexpr (sizeof(T) == 2) {
            for (int e = lane; e < V; e += 64) add(__half2float(pr[e]), qr[e]);
        } else {
            for (int e = c; e < V && e < c + 4; ++e) add(pr[e], qr[e]);       // (only the lane whose 4 columns straddle V gets here with e < V)
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (lane == 0) scores[g] = (float)sum;
    }
}

// per query: (score desc, first-stage rank asc) -> ids of the hits in the new order + their scores.  k <= kSpThreads * 2.
__global__ __launch_bounds__(kSpThreads) void rerank_topk_kernel(const float* scores, const int64_t* hit_ids, int32_t B, int32_t k, int64_t* out_ids, float* out_scores) {
    __shared__ uint64_t keys[2 * kSpThreads];
    const int tid = threadIdx.x;
    int n2 = 64;
    while (n2 < k) n2 <<= 1;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        for (int i = tid; i < n2; i += kSpThreads) keys[i] = i < k ? make_key(scores[(size_t)b * k + i], (uint32_t)i) : 0ull;
        wg_sort_desc<kSpThreads>(keys, n2, tid);
        for (int i = tid; i < k; i += kSpThreads) {
            const uint64_t key = keys[i];
            out_ids[(size_t)b * k + i] = hit_ids[(size_t)b * k + key_row(key)];
            out_scores[(size_t)b * k + i] = key_score(key);
        }
    }
}

}  // namespace

extern "C" int vs_head_pool_mean_topk(const float* logits, int32_t B, int32_t L, int32_t V, int32_t topk, float* out, int device, void* stream) {
    if (!logits || !out || B <= 0 || L <= 0 || V <= 0) return fail(VS_EINVAL, "bad argument");
    if (topk <= 0 || topk > kPoolTopkMax) return fail(VS_EUNSUPPORTED, "pooling_topk = %d (1..%d)", topk, kPoolTopkMax);
    if (topk > L) return fail(VS_ERANGE, "selected index k out of range (pooling_topk = %d > %d positions)", topk, L);
    if (!is_device_ptr(logits) || !is_device_ptr(out)) return fail(VS_EINVAL, "vs_head_pool_mean_topk takes device pointers");
    VS_TRY(check_device(device));
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)B * V;
    hipLaunchKernelGGL(head_pool_mean_topk_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(n, 256), 16384)), dim3(256), 0, s, logits, B, L, V, topk, out);
    VS_HIP(hipGetLastError());
    if (!stream) VS_HIP(hipStreamSynchronize(s));
    return VS_OK;
}

extern "C" int vs_rerank_scores(const void* p_emb, int p_dtype, int64_t ldp, int64_t n_rows, int64_t row0, const float* q, int64_t ldq, int32_t B,
                                int32_t k, int32_t n_cols, float* scores, int device, void* stream) {
    if (!p_emb || !q || !scores || n_rows < 0 || row0 < 0 || B <= 0 || k <= 0 || n_cols <= 0 || ldp < n_cols || ldq < n_cols)
        return fail(VS_EINVAL, "bad argument");
    if (row0 + n_rows > (int64_t)B * k) return fail(VS_ERANGE, "rows %lld..%lld beyond B * k = %lld", (long long)row0, (long long)(row0 + n_rows), (long long)B * k);
    if (p_dtype != VS_F32 && p_dtype != VS_F16) return fail(VS_EINVAL, "p_dtype must be VS_F32 or VS_F16");
    if (!is_device_ptr(p_emb) || !is_device_ptr(q) || !is_device_ptr(scores)) return fail(VS_EINVAL, "vs_rerank_scores takes device pointers");
    VS_TRY(check_device(device));
    hipStream_t s = (hipStream_t)stream;
    if (n_rows > 0) {
        ProfScope prof("rerank", s);
        const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(n_rows, 4), 8192);
        if (p_dtype == VS_F32)
            hipLaunchKernelGGL(rerank_scores_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)p_emb, ldp, n_rows, row0, q, ldq, k, n_cols, scores);
        else
            hipLaunchKernelGGL(rerank_scores_kernel<__half>, dim3(grid), dim3(256), 0, s, (const __half*)p_emb, ldp, n_rows, row0, q, ldq, k, n_cols, scores);
    }
    VS_HIP(hipGetLastError());
    if (!stream) VS_HIP(hipStreamSynchronize(s));
    return VS_OK;
}

extern "C" int vs_rerank_topk(const float* scores, const int64_t* hit_ids, int32_t B, int32_t k, int64_t* out_ids, float* out_scores, int device, void* stream) {
    if (!scores || !hit_ids || !out_ids || !out_scores || B <= 0 || k <= 0) return fail(VS_EINVAL, "bad argument");
    if (k > 2 * kSpThreads) return fail(VS_EUNSUPPORTED, "rerank of k = %d hits per query (at most %d)", k, 2 * kSpThreads);
    if (!is_device_ptr(scores) || !is_device_ptr(hit_ids) || !is_device_ptr(out_ids) || !is_device_ptr(out_scores))
        return fail(VS_EINVAL, "vs_rerank_topk takes device pointers");
    VS_TRY(check_device(device));
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("rerank", s);
    hipLaunchKernelGGL(rerank_topk_kernel, dim3((unsigned)std::min(B, 4096)), dim3(kSpThreads), 0, s, scores, hit_ids, B, k, out_ids, out_scores);
    VS_HIP(hipGetLastError());
    if (!stream) VS_HIP(hipStreamSynchronize(s));
    return VS_OK;
}

extern "C" int vs_elu1p(const float* x, int64_t n, float* out, int device, void* stream) {
    if (!x || !out || n < 0) return fail(VS_EINVAL, "bad argument");
    VS_TRY(check_device(device));
    hipStream_t s = (hipStream_t)stream;
    DevBuf st_in, st_out;
    const void* dx = nullptr;
    VS_TRY(to_device(x, (size_t)n * 4, st_in, s, &dx));
    const bool o_host = !is_device_ptr(out);
    float* dout = out;
    if (o_host) { VS_TRY(st_out.alloc(std::max<size_t>((size_t)n * 4, 4))); dout = st_out.as<float>(); }
    if (n > 0) hipLaunchKernelGGL(elu1p_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(n, 256), 8192)), dim3(256), 0, s, (const float*)dx, n, dout);
    VS_HIP(hipGetLastError());
    if (o_host) VS_HIP(hipMemcpyAsync(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    if (o_host || !stream || st_in.p) VS_HIP(hipStreamSynchronize(s));
    return VS_OK;
}

extern "C" int vs_head_pool(const float* logits, int32_t B, int32_t L, int32_t V, float* out, int device, void* stream) {
    if (!logits || !out || B <= 0 || L <= 0 || V <= 0) return fail(VS_EINVAL, "bad argument");
    VS_TRY(check_device(device));
    hipStream_t s = (hipStream_t)stream;
    DevBuf st_in, st_out;
    const void* dx = nullptr;
    VS_TRY(to_device(logits, (size_t)B * L * V * 4, st_in, s, &dx));
    const bool o_host = !is_device_ptr(out);
    float* dout = out;
    if (o_host) { VS_TRY(st_out.alloc((size_t)B * V * 4)); dout = st_out.as<float>(); }
    hipLaunchKernelGGL(head_pool_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64((int64_t)B * V, 256), 8192)), dim3(256), 0, s, (const float*)dx, B, L, V, dout);
    VS_HIP(hipGetLastError());
    if (o_host) VS_HIP(hipMemcpyAsync(out, dout, (size_t)B * V * 4, hipMemcpyDeviceToHost, s));
    if (o_host || !stream || st_in.p) VS_HIP(hipStreamSynchronize(s));
    return VS_OK;
}

extern "C" int vs_topk_mask(const float* x, int32_t B, int32_t V, int64_t ld, int32_t k, uint8_t* mask, int device, void* stream) {
    if (!x || !mask || B <= 0 || V <= 0 || ld < V) return fail(VS_EINVAL, "bad argument");
    if (k < 0 || k > V) return fail(VS_ERANGE, "selected index k out of range (k = %d, V = %d)", k, V);
    MaskArgs a{};
    a.ld = ld; a.B = B; a.L = 0; a.V = V; a.vocab = V; a.shift = 0; a.topk = k; a.activate_lexical = 0; a.bow = 0;
    return run_mask(a, x, nullptr, nullptr, mask, device, (hipStream_t)stream);
}

extern "C" int vs_bow_mask(const int64_t* ids, int32_t B, int32_t L, int32_t vocab, int32_t shift, int norm, float* out,
                           int device, void* stream) {
    if (!ids || !out || B <= 0 || L <= 0 || vocab <= 0 || shift < 0 || shift >= vocab) return fail(VS_EINVAL, "bad argument");
    MaskArgs a{};
    a.V = vocab - shift; a.ld = a.V; a.B = B; a.L = L; a.vocab = vocab; a.shift = shift; a.topk = 0; a.activate_lexical = 1;
    a.bow = norm ? -1 : 1;
    return run_mask(a, nullptr, out, ids, nullptr, device, (hipStream_t)stream);
}

extern "C" int vs_embed_mask(float* emb, int64_t ld, const int64_t* ids, int32_t B, int32_t L, int32_t vocab, int32_t shift,
                             int32_t topk, int activate_lexical, int bow, int device, void* stream) {
    if (!emb || B <= 0 || vocab <= 0 || shift < 0 || shift >= vocab || ld < vocab - shift) return fail(VS_EINVAL, "bad argument");
    if ((bow || activate_lexical) && (!ids || L <= 0)) return fail(VS_EINVAL, "token ids required for bow / activate_lexical");
    const int V = vocab - shift;
    if (!bow && topk > V) return fail(VS_ERANGE, "selected index k out of range (k = %d, V = %d)", topk, V);
    MaskArgs a{};
    a.V = V; a.ld = ld; a.B = B; a.L = L; a.vocab = vocab; a.shift = shift; a.topk = topk; a.activate_lexical = activate_lexical; a.bow = bow ? 1 : 0;
    return run_mask(a, nullptr, emb, ids, nullptr, device, (hipStream_t)stream);
}

// VDREncoder.embed's mask stage FUSED with Tensor.to_sparse_csr() (vdr.py:152-169 + retriever.py:304; SURVEY 8(f1) "write CSR rows
// directly"): x [B, V] (the pooled activations, NOT modified) -> the CSR of  x * (topk_mask | lexical_mask).  The mask kernel ranks the
// kept elements while it still holds the row in registers and writes (column, value) pairs; the dense masked row is never written nor
// read back: one read of [B, V] instead of the five passes of vs_embed_mask + vs_dense_to_csr.
//   rowptr int64 [B + 1] (device), cols int32 / vals fp32 [cap] (device), cap >= B * (topk + L) suffices; nnz = rowptr[B].
//   VS_EUNSUPPORTED: outside the fast mask kernel's range (bow, topk <= 0, V > 32 Ki, topk + L > 8192) -- use vs_embed_mask + vs_dense_to_csr.
extern "C" int vs_embed_mask_to_csr(const float* x, int64_t ld, const int64_t* ids, int32_t B, int32_t L, int32_t vocab, int32_t shift, int32_t topk,
                                    int activate_lexical, int64_t* rowptr, int32_t* cols, float* vals, int64_t cap, int device, void* stream) {
    if (!x || !rowptr || !cols || !vals || B <= 0 || vocab <= 0 || shift < 0 || shift >= vocab || ld < vocab - shift || cap < 0) return fail(VS_EINVAL, "bad argument");
    if (activate_lexical && (!ids || L <= 0)) return fail(VS_EINVAL, "token ids required for activate_lexical");
    const int V = vocab - shift;
    if (topk <= 0 || V > kMrCols) return fail(VS_EUNSUPPORTED, "vs_embed_mask_to_csr serves top-k masks of V <= %d columns", kMrCols);
    if (topk > V) return fail(VS_ERANGE, "selected index k out of range (k = %d, V = %d)", topk, V);
    if (!is_device_ptr(x) || !is_device_ptr(rowptr) || !is_device_ptr(cols) || !is_device_ptr(vals) || (ids && !is_device_ptr(ids)))
        return fail(VS_EINVAL, "vs_embed_mask_to_csr takes device pointers");
    VS_TRY(check_device(device));
    hipStream_t s = (hipStream_t)stream;
    const int32_t slot_cap = (int32_t)std::min<int64_t>(V, (int64_t)topk + (activate_lexical ? L : 0));
    if (slot_cap > kMrStage) return fail(VS_EUNSUPPORTED, "vs_embed_mask_to_csr serves rows of topk + L <= %d kept elements", kMrStage);
    DevBuf& flags = device_scratch(device, kScratchFlags);
    DevBuf& counts = device_scratch(device, kScratchCounts);
    DevBuf& slots = device_scratch(device, kScratchCsrSlots);
    int cus = 0;
    VS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    const int grid = std::max(1, std::min(B, cus));
    const size_t flag_bytes = 256 + (size_t)grid * 8 * 5;                    // (+ developer builds: four clock readings per workgroup)                        // error bits, ticket | (developer builds: phase clocks) | the workgroups' totals
    VS_TRY(flags.reserve(flag_bytes));
    VS_TRY(counts.reserve((size_t)B * 8));
    VS_TRY(slots.reserve((size_t)B * slot_cap * 8));
    VS_HIP(hipMemsetAsync(flags.p, 0, flag_bytes, s));
    MaskArgs a{};
    a.x = x; a.emb = nullptr; a.ld = ld; a.ids = activate_lexical ? ids : nullptr;
    a.V = V; a.B = B; a.L = L; a.vocab = vocab; a.shift = shift; a.topk = topk; a.activate_lexical = activate_lexical; a.bow = 0;
    a.flags = flags.as<int>();
    a.row_nnz = counts.as<int64_t>();
    a.slot_cols = slots.as<int32_t>();
    a.slot_vals = reinterpret_cast<float*>(slots.as<int32_t>() + (size_t)B * slot_cap);
    a.slot_cap = slot_cap;
    a.csr_rowptr = rowptr; a.csr_cols = cols; a.csr_vals = vals; a.csr_cap = cap;
    {
        // ONE launch: select, rank, emit, and -- once the workgroups before it have published their totals -- every workgroup moves its run
        // to its place in cols / vals and writes its rows' rowptr entries (mask_rows_fast.h step 5)
        ProfScope prof("mask_to_csr", s);
        void (*kern)(MaskArgs) = V <= 4 * kMrStep ? mask_rows_fast_kernel<4, 1> : V <= 8 * kMrStep ? mask_rows_fast_kernel<8, 1> : V <= 15 * kMrStep ? mask_rows_fast_kernel<15, 1> : mask_rows_fast_kernel<16, 1>;
        VS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mask_fast_lds_bytes()));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kMrThreads), mask_fast_lds_bytes(), s, a);
        VS_HIP(hipGetLastError());
    }
#ifdef MR_TIMING
    {
        unsigned long long h[32];
        VS_HIP(hipMemcpyAsync(h, flags.p, 256, hipMemcpyDeviceToHost, s));
        VS_HIP(hipStreamSynchronize(s));
        static int calls = 0;
        if (++calls == 5) {
            const double rows = (double)B;
            std::vector<unsigned long long> rt((size_t)grid * 4);
            VS_HIP(hipMemcpy(rt.data(), (const char*)flags.p + 256 + (size_t)grid * 8, rt.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull;
            for (int i = 0; i < grid; ++i) t0 = std::min(t0, rt[4 * i]);
            std::vector<double> st, rd, en, pf;
            for (int i = 0; i < grid; ++i) { st.push_back((rt[4 * i] - t0) / 100.0); rd.push_back((rt[4 * i + 1] - t0) / 100.0); en.push_back((rt[4 * i + 2] - t0) / 100.0); pf.push_back((rt[4 * i + 3] - t0) / 100.0); }
            auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
            fprintf(stderr, "[vsearch_hip] mask -> CSR workgroups (us after the first start): start median %.1f max %.1f; rows done min %.1f median %.1f max %.1f; prefix known min %.1f median %.1f max %.1f; end min %.1f median %.1f max %.1f\n",
                    pct(st, 0.5), pct(st, 1.0), pct(rd, 0.0), pct(rd, 0.5), pct(rd, 1.0), pct(pf, 0.0), pct(pf, 0.5), pct(pf, 1.0), pct(en, 0.0), pct(en, 0.5), pct(en, 1.0));
            fprintf(stderr, "[vsearch_hip] mask -> CSR workgroups by ticket (rows done / prefix known / end):");
            for (int i = 0; i < grid; i += std::max(1, grid / 16)) fprintf(stderr, " %d: %.1f %.1f %.1f;", i, rd[i], pf[i], en[i]);
            fprintf(stderr, " %d: %.1f %.1f %.1f\n", grid - 1, rd[grid - 1], pf[grid - 1], en[grid - 1]);
            fprintf(stderr, "[vsearch_hip] mask -> CSR, cycles per row (wave 0): barrier+wait %.0f, pack %.0f, issue+clear+lex %.0f, pass A %.0f, pick A %.0f, pass B + pick B %.0f, masks %.0f, "
                    "candidates %.0f (list %.0f, barrier %.0f, ranking %.0f, barrier %.0f), totals %.0f, rank + stage %.0f, copy out %.0f; per workgroup: its place in the CSR arrays %.0f, kernel %.0f cycles\n", h[1] / rows, h[2] / rows, h[3] / rows, h[4] / rows, h[5] / rows, h[6] / rows, h[7] / rows, h[8] / rows,
                    h[12] / rows, h[13] / rows, h[14] / rows, h[15] / rows, h[9] / rows, h[10] / rows, h[11] / rows, h[16] / (double)grid, h[17] / (double)grid);
        }
    }
#endif
    int hflags = 0;
    VS_HIP(hipMemcpyAsync(&hflags, flags.p, 4, hipMemcpyDeviceToHost, s));
    VS_HIP(hipStreamSynchronize(s));
    if (hflags & 1) return fail(VS_EINVAL, "token id out of range [0, %d)", vocab);
    if (hflags & 2) return fail(VS_ERANGE, "a row kept more than topk + L = %d elements", slot_cap);
    if (hflags & 4) return fail(VS_ERANGE, "cols / vals hold %lld entries, the batch has more non-zeros", (long long)cap);
    return VS_OK;
}

extern "C" int vs_dense_to_csr(const float* x, int32_t B, int32_t V, int64_t ld, int64_t* rowptr, int32_t* cols, float* vals,
                               int64_t cap, int device, void* stream) {
    if (!x || !rowptr || B <= 0 || V <= 0 || ld < V) return fail(VS_EINVAL, "bad argument");
    if ((cols == nullptr) != (vals == nullptr)) return fail(VS_EINVAL, "cols and vals must be given together");
    VS_TRY(check_device(device));
    hipStream_t s = (hipStream_t)stream;
    DevBuf st_x, st_c, st_v;
    const void* dx = nullptr;
    VS_TRY(to_device(x, ((size_t)(B - 1) * ld + V) * 4, st_x, s, &dx));
    DevBuf& counts = device_scratch(device, kScratchCounts);
    DevBuf& d_rp = device_scratch(device, kScratchRowPtr);
    VS_TRY(counts.reserve((size_t)B * 8));
    VS_TRY(d_rp.reserve((size_t)(B + 1) * 8));
    const bool rp_dev = is_device_ptr(rowptr);
    if (!cols) {
        // sizing call: rowptr out
        ProfScope prof("dense_to_csr", s);
        hipLaunchKernelGGL(count_nz_kernel<0>, dim3(std::min(B, 2048)), dim3(kSpThreads), 0, s, (const float*)dx, ld, B, V, counts.as<int64_t>());
        int64_t* rp_out = rp_dev ? rowptr : d_rp.as<int64_t>();
        hipLaunchKernelGGL(scan_counts_kernel<0>, dim3(1), dim3(kSpThreads), 0, s, counts.as<int64_t>(), B, rp_out);
        VS_HIP(hipGetLastError());
        if (!rp_dev) {
            VS_HIP(hipMemcpyAsync(rowptr, d_rp.p, (size_t)(B + 1) * 8, hipMemcpyDeviceToHost, s));
            VS_HIP(hipStreamSynchronize(s));
        } else if (st_x.p || !s) {
            VS_HIP(hipStreamSynchronize(s));
        }
        return VS_OK;
    }
    // fill call: rowptr (from the sizing call) in, cols / vals out
    const int64_t* rp_in = rowptr;
    if (!rp_dev) {
        VS_HIP(hipMemcpyAsync(d_rp.p, rowptr, (size_t)(B + 1) * 8, hipMemcpyHostToDevice, s));
        rp_in = d_rp.as<int64_t>();
    }
    const bool o_host = !is_device_ptr(cols);
    int32_t* dc = cols;
    float* dv = vals;
    if (o_host) {
        VS_TRY(st_c.alloc(std::max<size_t>((size_t)cap * 4, 4)));
        VS_TRY(st_v.alloc(std::max<size_t>((size_t)cap * 4, 4)));
        dc = st_c.as<int32_t>();
        dv = st_v.as<float>();
    }
    {
        ProfScope prof("dense_to_csr", s);
        hipLaunchKernelGGL(fill_csr_kernel<0>, dim3(std::min(B, 2048)), dim3(kSpThreads), 0, s, (const float*)dx, ld, B, V, rp_in, dc, dv, cap);
    }
    VS_HIP(hipGetLastError());
    if (o_host && cap > 0) {
        VS_HIP(hipMemcpyAsync(cols, dc, (size_t)cap * 4, hipMemcpyDeviceToHost, s));
        VS_HIP(hipMemcpyAsync(vals, dv, (size_t)cap * 4, hipMemcpyDeviceToHost, s));
    }
    if (o_host || !rp_dev || st_x.p || !s) VS_HIP(hipStreamSynchronize(s));
    return VS_OK;
}

// ---- Retriever._build_bot_vectors (retriever.py:208-253): host-side integer set work ---------------
extern "C" int vs_bot_build(const int32_t* tokens, const int64_t* offsets, int64_t n_docs, int32_t vocab, int32_t shift,
                            int32_t max_token, int64_t* out_rowptr, int32_t* out_cols) {
    if (!offsets || !out_rowptr || n_docs < 0 || vocab <= 0 || shift < 0 || shift > vocab) return fail(VS_EINVAL, "bad argument");
    if (n_docs > 0 && !tokens && offsets[n_docs] > offsets[0]) return fail(VS_EINVAL, "tokens is NULL");
    std::vector<uint32_t> stamp((size_t)vocab, 0u);     // last doc (1-based) that touched a token id
    std::vector<int32_t> uniq;
    out_rowptr[0] = 0;
    for (int64_t d = 0; d < n_docs; ++d) {
        uniq.clear();
        const uint32_t tag = (uint32_t)(d % 0xFFFFFFFEull) + 1;
        if (tag == 1 && d > 0) std::fill(stamp.begin(), stamp.end(), 0u);
        for (int64_t p = offsets[d]; p < offsets[d + 1]; ++p) {
            const int32_t t = tokens[p];
            if (t < 0 || t >= vocab) return fail(VS_EINVAL, "token id %d out of range [0, %d) in doc %lld", t, vocab, (long long)d);
            if (stamp[t] == tag) continue;
            stamp[t] = tag;
            uniq.push_back(t);                           // first-unique order (index_utils.py:11-21)
            if (max_token > 0 && (int32_t)uniq.size() == max_token) break;
        }
        int64_t cnt = 0;
        for (int32_t t : uniq) cnt += t >= shift;
        out_rowptr[d + 1] = out_rowptr[d] + cnt;
        if (out_cols) {
            int32_t* dst = out_cols + out_rowptr[d];
            int64_t w = 0;
            for (int32_t t : uniq) if (t >= shift) dst[w++] = t - shift;
            std::sort(dst, dst + w);
        }
    }
    return VS_OK;
}
