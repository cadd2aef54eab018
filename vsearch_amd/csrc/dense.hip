// dense.hip -- dense Index container and search (index.py:25-44, :88-94; build: retriever.py:292-297)
//   scores[B, N] = Q[B, V] . P[N, V]^T  on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32
//   products accumulated like an fmaf chain), scores leave the kernel as 64-bit order keys and the
//   shared top-k merge selects per query.
#include "common.h"
#include "csr_scan.h"
#include "dense_csr.h"

#include <algorithm>

using namespace vs;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kDenseKC = 32;   // columns per K-chunk: each half-wave loads 16 contiguous floats per row

// copy [n_rows, n_cols] (fp32 | fp16, leading dim ld) -> padded fp32 [n_rows, ldp], tail zeroed;
// round_f16 emulates storing / casting to fp16 (index.py:42,89).
template <class T>
__global__ void pad_rows_kernel(const T* src, int64_t ld, int64_t n_rows, int32_t n_cols, int32_t ldp, int round_f16, float* dst) {
    const int64_t n = n_rows * ldp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ldp;
        const int32_t c = (int32_t)(i % ldp);
        float v = 0.f;
        if (c < n_cols) {
            if constexpr (sizeof(T) == 4) v = src[r * ld + c];
            else v = __half2float(src[r * ld + c]);
            if (round_f16) v = __half2float(__float2half_rn(v));
        }
        dst[i] = v;
    }
}

template <class T>
__global__ void unpad_rows_kernel(const float* src, int32_t ldp, int64_t n_rows, int32_t n_cols, int64_t ld, T* dst) {
    const int64_t n = n_rows * n_cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n_cols;
        const int32_t c = (int32_t)(i % n_cols);
        if constexpr (sizeof(T) == 4) dst[r * ld + c] = src[r * ldp + c];
        else dst[r * ld + c] = __float2half_rn(src[r * ldp + c]);
    }
}

// scores[B, N] = Q[B, V] . P[N, V]^T on the fp32 matrix cores.
// Workgroup = 4 waves arranged WM x WN; a wave owns TM x TN MFMA tiles of 32 x 32 (v_mfma_f32_32x32x2_f32), so the
// block tile is BM x BN = (WM*TM*32) queries x (WN*TN*32) docs.  <2,2,2,2> = 128 x 128 is the main shape; <1,4,1,1> =
// 32 x 128 blocks (a quarter of the work each) fill the last, partial round of the grid so the tail does not
// cost a whole extra round.  K is walked in chunks of 32 columns: both operand tiles are fetched with coalesced
// 16-byte loads into registers while the previous chunk is multiplied (register-staged prefetch), then written
// to a double-buffered LDS image with a 36-float row pitch -- with that pitch a 16-lane ds_read_b128 group
// touches all 64 banks exactly once.  The MFMA K order is permuted (half-wave h owns columns 16h..16h+15 of the
// chunk) so a lane's operands for 4 consecutive MFMA steps are one ds_read_b128.  Rows are padded to ldp
// (multiple of 32) with zeros.  Two-level summation: chains inside 512-column blocks, block sums added to `tot`.
constexpr int kDPitch = kDenseKC + 4;

// SK = 1 (split-K, the tail of a search: launch_dense_scores): blockIdx.z picks a slice of `cps` K chunks (a multiple of 16: whole
// summation blocks); the block writes the sum of every 512-column summation block to part[block][B][n_tail] instead of keys / scores.
template <int WM, int WN, int TM, int TN, int SK = 0>
__global__ __launch_bounds__(256) void dense_scores_kernel(const float* __restrict__ Q, const float* __restrict__ P, int32_t B,
                                                           int64_t N, int64_t n_begin, int32_t ldp, uint64_t* keys, float* scores,
                                                           int32_t pool_L, uint32_t* pool_out, float* part, int32_t cps, int64_t n_tail) {
    // pool_L > 0: encoder-head mode (vdr.py:72-75).  Q = LayerNorm'ed hidden states [B * pool_L, ldp]; the block covers rows
    // of ONE sequence (blockIdx.y = sequence * l_tiles + l_tile); instead of scores it emits the column-wise max over the
    // sequence positions, as order keys merged with atomicMax (rows past the sequence end repeat its last row).
    static_assert(WM * WN == 4 && WM * TM <= 4 && WN * TN <= 4, "4 waves per workgroup, tiles up to 128 x 128");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int JA = BM / 32, JB = BN / 32;               // float4 staged per thread for each operand
    __shared__ __attribute__((aligned(16))) float As[2][BM * kDPitch];      // double-buffered: one barrier per K chunk
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * kDPitch];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w / WN, wn = w % WN;
    const int h = lane >> 5, l31 = lane & 31;
    const int64_t n_blk = n_begin + (int64_t)blockIdx.x * BN;
    int b_blk = blockIdx.y * BM;
    int row_hi = B - 1, seq = 0;
    if (pool_L > 0) {
        const int l_tiles = (pool_L + BM - 1) / BM;
        seq = blockIdx.y / l_tiles;
        row_hi = seq * pool_L + pool_L - 1;
        b_blk = seq * pool_L + (blockIdx.y % l_tiles) * BM;
    }
    // global -> register staging: thread t covers rows (t/8) + 32 j, 16-byte column group t%8
    const int lr = tid >> 3, lc = (tid & 7) * 4;
    const float4* q4[JA];
    const float4* p4[JB];
#pragma unroll
    for (int j = 0; j < JA; ++j) q4[j] = reinterpret_cast<const float4*>(Q + (size_t)min(b_blk + lr + 32 * j, row_hi) * ldp + lc);
#pragma unroll
    for (int j = 0; j < JB; ++j) p4[j] = reinterpret_cast<const float4*>(P + (size_t)min(n_blk + lr + 32 * j, N - 1) * ldp + lc);
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 tot[TM][TN], acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) { tot[i][j] = zero; acc[i][j] = zero; }
    // staging registers: named scalars + `if constexpr` (as arrays -- even with unrolled constant indices -- the
    // compiler keeps them in scratch memory, which halves the kernel's speed)
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    ra0 = ra1 = ra2 = ra3 = rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int chunks = ldp / kDenseKC;
    const int a_off = (wm * TM * 32 + l31) * kDPitch + h * 16, b_off = (wn * TN * 32 + l31) * kDPitch + h * 16;
    const int wa = lr * kDPitch + lc;
#define VS_DENSE_FETCH(c)                                     \
    {                                                         \
        const int cn_ = min((c), chunks - 1) * (kDenseKC / 4);\
        ra0 = q4[0][cn_];                                     \
        if constexpr (JA > 1) ra1 = q4[JA > 1 ? 1 : 0][cn_];  \
        if constexpr (JA > 2) ra2 = q4[JA > 2 ? 2 : 0][cn_];  \
        if constexpr (JA > 3) ra3 = q4[JA > 3 ? 3 : 0][cn_];  \
        rb0 = p4[0][cn_];                                     \
        if constexpr (JB > 1) rb1 = p4[JB > 1 ? 1 : 0][cn_];  \
        if constexpr (JB > 2) rb2 = p4[JB > 2 ? 2 : 0][cn_];  \
        if constexpr (JB > 3) rb3 = p4[JB > 3 ? 3 : 0][cn_];  \
    }
#define VS_DENSE_STAGE(buf)                                                                            \
    {                                                                                                  \
        *reinterpret_cast<float4*>(As[buf] + wa) = ra0;                                                \
        if constexpr (JA > 1) *reinterpret_cast<float4*>(As[buf] + wa + 32 * kDPitch) = ra1;           \
        if constexpr (JA > 2) *reinterpret_cast<float4*>(As[buf] + wa + 64 * kDPitch) = ra2;           \
        if constexpr (JA > 3) *reinterpret_cast<float4*>(As[buf] + wa + 96 * kDPitch) = ra3;           \
        *reinterpret_cast<float4*>(Bs[buf] + wa) = rb0;                                                \
        if constexpr (JB > 1) *reinterpret_cast<float4*>(Bs[buf] + wa + 32 * kDPitch) = rb1;           \
        if constexpr (JB > 2) *reinterpret_cast<float4*>(Bs[buf] + wa + 64 * kDPitch) = rb2;           \
        if constexpr (JB > 3) *reinterpret_cast<float4*>(Bs[buf] + wa + 96 * kDPitch) = rb3;           \
    }
    int c_lo = 0, c_hi = chunks;
    if constexpr (SK != 0) { c_lo = (int)blockIdx.z * cps; c_hi = min(chunks, c_lo + cps); }      // (c_lo even: the buffer parity below holds)
    VS_DENSE_FETCH(c_lo)
    VS_DENSE_STAGE(0)
    VS_DENSE_FETCH(c_lo + 1)
    __syncthreads();
    for (int c = c_lo; c < c_hi; ++c) {
        const float* a_rd = As[c & 1] + a_off;
        const float* b_rd = Bs[c & 1] + b_off;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float4*>(a_rd + i * 32 * kDPitch + s4 * 4);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const float4*>(b_rd + j * 32 * kDPitch + s4 * 4);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
        // the other buffer was last read in iteration c-1 (every wave has passed that iteration's barrier)
        if (c + 1 < c_hi) {
            if (c & 1) VS_DENSE_STAGE(0) else VS_DENSE_STAGE(1)
        }
        __syncthreads();
        VS_DENSE_FETCH(c + 2)
        if ((c & 15) == 15 || c == c_hi - 1) {
            if constexpr (SK != 0) {
                // split-K: the block sum itself goes out -- part[summation block][B][n_tail] -- and splitk_reduce_kernel adds the blocks in
                // block order from zero: the SAME association as `tot += acc` below, so a document's score does not depend on whether
                // it fell into a main round or the tail, on the batch size, on the CU count or on how the index is sharded
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int64_t n = n_blk + (wn * TN + j) * 32 + l31;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int b = b_blk + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                            if (b < B && n < N) part[((size_t)(c >> 4) * B + b) * n_tail + (n - n_begin)] = acc[i][j][r];
                        }
                        acc[i][j] = zero;
                    }
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) { tot[i][j] += acc[i][j]; acc[i][j] = zero; }
            }
        }
    }
    if constexpr (SK != 0) return;
#undef VS_DENSE_FETCH
#undef VS_DENSE_STAGE
    if (pool_L > 0) {
        // column max over this wave's rows (all valid: clamped rows duplicate the last position), both half-waves, then
        // one atomicMax per column into the [sequences, N] key buffer
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float m = -INFINITY;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, tot[i][j][r]);
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            const int64_t n = n_blk + (wn * TN + j) * 32 + l31;
            if (h == 0 && n < N) atomicMax(pool_out + (size_t)seq * N + n, flip_f32(m));
        }
        return;
    }
    // C/D layout of a 32x32 tile: col = lane & 31 (doc), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (query)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int64_t n = n_blk + (wn * TN + j) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = b_blk + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (b < B && n < N) {
                    if (keys) keys[(size_t)b * N + n] = make_key(tot[i][j][r], (uint32_t)n);
                    if (scores) scores[(size_t)b * N + n] = tot[i][j][r];
                }
            }
        }
}

// the 512-column block sums of the tail documents, added from zero in block order -- exactly what the main kernel's `tot += acc` does
__global__ void splitk_reduce_kernel(const float* part, int32_t n_blocks, int32_t B, int64_t n_tail, int64_t n_begin, int64_t N, uint64_t* keys, float* scores) {
    const int64_t total = (int64_t)B * n_tail;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / n_tail, t = i % n_tail;
        float sum = 0.f;
        for (int z = 0; z < n_blocks; ++z) sum += part[(size_t)z * total + i];
        const int64_t n = n_begin + t;
        if (keys) keys[(size_t)b * N + n] = make_key(sum, (uint32_t)n);
        if (scores) scores[(size_t)b * N + n] = sum;
    }
}

// Launch plan: full rounds of 128 x 128 blocks (2 co-resident per CU); the remaining documents -- less than a round of blocks -- are
// split along K over the idle slots (128 x 128 blocks on slices of whole 512-column summation blocks; the block sums are added in
// block order by splitk_reduce_kernel: the main kernel's association), or, when K is too short to split, done by 32 x 128 blocks.  (C2, 100 k x 29 523, B = 256: 14 of 782 document tiles are
// left after three rounds; as 112 quarter blocks they took 0.75 ms of 11.9.)
int launch_dense_scores(vs_index* idx, const float* dq, int B, int ldp, uint64_t* keys, float* scores, hipStream_t s) {
    const int64_t N = idx->n_rows;
    const int64_t doc_tiles = ceil_div64(N, 128), q_tiles = ceil_div64(B, 128);
    const int64_t slots = (int64_t)idx->cu_count * 2;
    const int64_t full_rounds = (doc_tiles * q_tiles) / slots;
    int64_t main_doc_tiles = std::min(doc_tiles, (full_rounds * slots) / q_tiles);
    if (main_doc_tiles * 128 > N) main_doc_tiles = N / 128;
    if (main_doc_tiles > 0) {
        hipLaunchKernelGGL((dense_scores_kernel<2, 2, 2, 2>), dim3((unsigned)main_doc_tiles, (unsigned)q_tiles), dim3(256), 0, s, dq,
                           idx->mat.as<float>(), B, N, (int64_t)0, ldp, keys, scores, 0, (uint32_t*)nullptr, (float*)nullptr, 0, (int64_t)0);
        VS_HIP(hipGetLastError());
    }
    const int64_t n_begin = main_doc_tiles * 128;
    if (n_begin < N) {
        const int64_t n_tail = N - n_begin, tail_tiles = ceil_div64(n_tail, 128) * q_tiles;
        const int chunks = ldp / kDenseKC;
        int S = (int)std::min<int64_t>(slots / std::max<int64_t>(tail_tiles, 1), chunks / 16);
        // (the split's workspace holds one partial per 512-column summation block: n_sum_blocks x B x n_tail floats -- 58 slices at V =
        //  29 523.  ADVICE r4: when that passes 256 MB, or cannot be reserved, the 32 x 128 tail kernel does the tail -- same summation
        //  order, no workspace -- instead of failing the search)
        const int n_sum_blocks = ceil_div(chunks, 16);
        const size_t ws_bytes = (size_t)n_sum_blocks * B * n_tail * 4;
        bool split = main_doc_tiles > 0 && S >= 2 && ws_bytes <= ((size_t)256 << 20);
        if (split && idx->ws_fb.reserve(ws_bytes) != VS_OK) { (void)hipGetLastError(); split = false; }
        if (split) {
            const int cps = ceil_div(ceil_div(chunks, S), 16) * 16;
            S = ceil_div(chunks, cps);
            hipLaunchKernelGGL((dense_scores_kernel<2, 2, 2, 2, 1>), dim3((unsigned)ceil_div64(n_tail, 128), (unsigned)q_tiles, (unsigned)S), dim3(256), 0, s, dq,
                               idx->mat.as<float>(), B, N, n_begin, ldp, (uint64_t*)nullptr, (float*)nullptr, 0, (uint32_t*)nullptr, idx->ws_fb.as<float>(), cps, n_tail);
            VS_HIP(hipGetLastError());
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64((int64_t)B * n_tail, 256), 4096)), dim3(256), 0, s,
                               (const float*)idx->ws_fb.as<float>(), n_sum_blocks, B, n_tail, n_begin, N, keys, scores);
        } else {
            hipLaunchKernelGGL((dense_scores_kernel<1, 4, 1, 1>), dim3((unsigned)ceil_div64(n_tail, 128), (unsigned)ceil_div(B, 32)), dim3(256), 0, s,
                               dq, idx->mat.as<float>(), B, N, n_begin, ldp, keys, scores, 0, (uint32_t*)nullptr, (float*)nullptr, 0, (int64_t)0);
        }
        VS_HIP(hipGetLastError());
    }
    return VS_OK;
}

int prep_dense_queries(vs_index* idx, const void* q, int q_dtype, int64_t ldq, int B, int ldp, hipStream_t s, const float** out) {
    if (q_dtype != VS_F32 && q_dtype != VS_F16) return fail(VS_EINVAL, "q_dtype must be VS_F32 or VS_F16");
    const size_t esz = dtype_size(q_dtype);
    const void* dq = q;
    if (!is_device_ptr(q)) {
        const size_t bytes = ((size_t)(B - 1) * ldq + idx->n_cols) * esz;
        VS_TRY(idx->ws_misc.reserve(bytes));
        VS_HIP(hipMemcpyAsync(idx->ws_misc.p, q, bytes, hipMemcpyHostToDevice, s));
        dq = idx->ws_misc.p;
    }
    VS_TRY(idx->ws_q.reserve((size_t)B * ldp * 4));
    const int round_f16 = idx->store_dtype == VS_F16;
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64((int64_t)B * ldp, 256), 8192);
    if (q_dtype == VS_F32)
        hipLaunchKernelGGL((pad_rows_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)dq, ldq, (int64_t)B, idx->n_cols, ldp, round_f16, idx->ws_q.as<float>());
    else
        hipLaunchKernelGGL((pad_rows_kernel<__half>), dim3(grid), dim3(256), 0, s, (const __half*)dq, ldq, (int64_t)B, idx->n_cols, ldp, round_f16, idx->ws_q.as<float>());
    VS_HIP(hipGetLastError());
    *out = idx->ws_q.as<float>();
    return VS_OK;
}

inline int dense_ldp(int32_t n_cols) { return (n_cols + kDenseKC - 1) / kDenseKC * kDenseKC; }

}  // namespace


int vs_csr_append_rows(vs_index* idx, const void* rowptr, int rowptr_dtype, const void* colidx, int col_dtype,
                       const void* values, int val_dtype, int64_t n_rows);

namespace {

// rows of a CSR-backed dense index -> dense rows (zeros elsewhere)
template <class T>
__global__ void csr_rows_to_dense_kernel(const uint32_t* pk_ptr, const uint16_t* cols, const void* vals, int store_dtype, int32_t n_cols,
                                         int64_t row_begin, int64_t row_end, int64_t ld, T* dst) {
    const int lane = threadIdx.x & 63;
    const int64_t row = row_begin + (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= row_end) return;
    T* out = dst + (size_t)(row - row_begin) * ld;
    for (int c = lane; c < n_cols; c += 64) {
        if constexpr (sizeof(T) == 4) out[c] = 0.f;
        else out[c] = __float2half_rn(0.f);
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const int64_t d0 = (int64_t)pk_ptr[row] * 8, d1 = (int64_t)pk_ptr[row + 1] * 8;
    for (int64_t j = d0 + lane; j < d1; j += 64) {
        const uint16_t c = cols[j];
        if (c == (uint16_t)n_cols) continue;
        float v = 1.0f;
        if (store_dtype == VS_F32) v = reinterpret_cast<const float*>(vals)[j];
        else if (store_dtype == VS_F16) v = __half2float(reinterpret_cast<const __half*>(vals)[j]);
        if constexpr (sizeof(T) == 4) out[c] = v;
        else out[c] = __float2half_rn(v);
    }
}

int export_dense_from_csr(const vs_index* idx, void* mat, int dtype, int64_t ld) {
    const size_t esz = dtype_size(dtype);
    const bool dst_dev = is_device_ptr(mat);
    const int64_t rows_per_chunk = std::max<int64_t>(1, ((int64_t)256 << 20) / ((int64_t)ld * (int64_t)esz));
    DevBuf stage;
    for (int64_t r = 0; r < idx->n_rows; r += rows_per_chunk) {
        const int64_t nr = std::min(rows_per_chunk, idx->n_rows - r);
        void* ddst = (char*)mat + (size_t)r * ld * esz;
        if (!dst_dev) {
            VS_TRY(stage.reserve((size_t)nr * ld * esz));
            ddst = stage.p;
        }
        const unsigned grid = (unsigned)ceil_div64(nr, 4);
        if (dtype == VS_F32)
            hipLaunchKernelGGL((csr_rows_to_dense_kernel<float>), dim3(grid), dim3(256), 0, 0, idx->pk_ptr.as<uint32_t>(), idx->cols.as<uint16_t>(),
                               idx->vals.p, idx->store_dtype, idx->n_cols, r, r + nr, ld, (float*)ddst);
        else
            hipLaunchKernelGGL((csr_rows_to_dense_kernel<__half>), dim3(grid), dim3(256), 0, 0, idx->pk_ptr.as<uint32_t>(), idx->cols.as<uint16_t>(),
                               idx->vals.p, idx->store_dtype, idx->n_cols, r, r + nr, ld, (__half*)ddst);
        VS_HIP(hipGetLastError());
        VS_HIP(hipDeviceSynchronize());
        if (!dst_dev) {
            const size_t bytes = ((size_t)(nr - 1) * ld + idx->n_cols) * esz;
            VS_HIP(hipMemcpy((char*)mat + (size_t)r * ld * esz, stage.p, bytes, hipMemcpyDeviceToHost));
        }
    }
    return VS_OK;
}

// stage rows [r, r + nr) of the caller's matrix as contiguous fp32 [nr, n_cols] on the device
int stage_rows_f32(const void* mat, int dtype, int64_t ld, int32_t n_cols, int64_t r, int64_t nr, int round_f16, DevBuf& raw, DevBuf& f32) {
    const size_t esz = dtype_size(dtype);
    const char* src = (const char*)mat + (size_t)r * ld * esz;
    const void* dsrc = src;
    if (!is_device_ptr(mat)) {
        const size_t bytes = ((size_t)(nr - 1) * ld + n_cols) * esz;
        VS_TRY(raw.reserve(bytes));
        VS_HIP(hipMemcpy(raw.p, src, bytes, hipMemcpyHostToDevice));
        dsrc = raw.p;
    }
    VS_TRY(f32.reserve((size_t)nr * n_cols * 4));
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(nr * n_cols, 256), 16384);
    if (dtype == VS_F32)
        hipLaunchKernelGGL((pad_rows_kernel<float>), dim3(grid), dim3(256), 0, 0, (const float*)dsrc, ld, nr, n_cols, n_cols, round_f16, f32.as<float>());
    else
        hipLaunchKernelGGL((pad_rows_kernel<__half>), dim3(grid), dim3(256), 0, 0, (const __half*)dsrc, ld, nr, n_cols, n_cols, round_f16, f32.as<float>());
    VS_HIP(hipGetLastError());
    return VS_OK;
}

}  // namespace

extern "C" int vs_index_create_dense(const void* mat, int dtype, int store_dtype, int64_t n_rows, int32_t n_cols, int64_t ld,
                                     int device, vs_index** out) {
    if (!out) return fail(VS_EINVAL, "out is NULL");
    *out = nullptr;
    if (!mat || n_rows <= 0 || n_cols <= 0 || ld < n_cols) return fail(VS_EINVAL, "bad matrix / shape");
    if (dtype != VS_F32 && dtype != VS_F16) return fail(VS_EINVAL, "dtype must be VS_F32 or VS_F16");
    if (store_dtype != VS_F32 && store_dtype != VS_F16) return fail(VS_EINVAL, "store_dtype must be VS_F32 or VS_F16");
    if (n_rows >= (1ll << 32) - 1) return fail(VS_EUNSUPPORTED, "n_rows must fit in 32 bits");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(VS_ENODEVICE, "no HIP device visible: libvsearch_hip has no CPU fallback");
    }
    if (device < 0 || device >= ndev) return fail(VS_EINVAL, "device %d out of range", device);
    VS_HIP(hipSetDevice(device));
    vs_index* idx = new (std::nothrow) vs_index();
    if (!idx) return fail(VS_ENOMEM, "host allocation failed");
    struct Guard { vs_index* p; ~Guard() { if (p) vs_index_destroy(p); } } guard{idx};
    hipDeviceProp_t prop;
    VS_HIP(hipGetDeviceProperties(&prop, device));
    idx->device = device;
    idx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    idx->kind = VS_KIND_DENSE;
    idx->store_dtype = store_dtype;          // fp16 storage is emulated by rounding; the matrix stays fp32 on device (v0)
    idx->n_rows = n_rows;
    idx->n_cols = n_cols;
    idx->nnz = n_rows * (int64_t)n_cols;
    const int ldp = dense_ldp(n_cols);
    VS_TRY(idx->mat.alloc((size_t)n_rows * ldp * 4));
    const size_t esz = dtype_size(dtype);
    const int64_t rows_per_chunk = std::max<int64_t>(1, ((int64_t)256 << 20) / ((int64_t)ld * (int64_t)esz));
    DevBuf stage;
    for (int64_t r = 0; r < n_rows; r += rows_per_chunk) {
        const int64_t nr = std::min(rows_per_chunk, n_rows - r);
        const char* src = (const char*)mat + (size_t)r * ld * esz;
        const void* dsrc = src;
        if (!is_device_ptr(mat)) {
            const size_t bytes = ((size_t)(nr - 1) * ld + n_cols) * esz;
            VS_TRY(stage.reserve(bytes));
            VS_HIP(hipMemcpy(stage.p, src, bytes, hipMemcpyHostToDevice));
            dsrc = stage.p;
        }
        const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(nr * ldp, 256), 16384);
        float* dst = idx->mat.as<float>() + (size_t)r * ldp;
        if (dtype == VS_F32)
            hipLaunchKernelGGL((pad_rows_kernel<float>), dim3(grid), dim3(256), 0, 0, (const float*)dsrc, ld, nr, n_cols, ldp, store_dtype == VS_F16, dst);
        else
            hipLaunchKernelGGL((pad_rows_kernel<__half>), dim3(grid), dim3(256), 0, 0, (const __half*)dsrc, ld, nr, n_cols, ldp, store_dtype == VS_F16, dst);
        VS_HIP(hipGetLastError());
        VS_HIP(hipDeviceSynchronize());
    }
    guard.p = nullptr;
    *out = idx;
    return VS_OK;
}

// Sparsity-aware dense index: a "dense" index of VDR embeddings holds <= 768 non-zeros per 29 523-wide row
// (retriever.py:292-297 keeps them dense).  When the matrix density is <= max_density the rows are stored
// as CSR packets and searched by the CSR scan (identical sums: zeros contribute nothing) -- ~2.6 % of the
// bytes and none of the 2*B*V*N flops; otherwise this is vs_index_create_dense.
extern "C" int vs_index_create_dense_auto(const void* mat, int dtype, int store_dtype, int64_t n_rows, int32_t n_cols, int64_t ld,
                                          double max_density, int device, vs_index** out) {
    if (!out) return fail(VS_EINVAL, "out is NULL");
    *out = nullptr;
    if (!mat || n_rows <= 0 || n_cols <= 0 || ld < n_cols) return fail(VS_EINVAL, "bad matrix / shape");
    if (dtype != VS_F32 && dtype != VS_F16) return fail(VS_EINVAL, "dtype must be VS_F32 or VS_F16");
    if (store_dtype != VS_F32 && store_dtype != VS_F16) return fail(VS_EINVAL, "store_dtype must be VS_F32 or VS_F16");
    if (max_density <= 0.0 || n_cols > 65535) return vs_index_create_dense(mat, dtype, store_dtype, n_rows, n_cols, ld, device, out);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(VS_ENODEVICE, "no HIP device visible: libvsearch_hip has no CPU fallback");
    }
    if (device < 0 || device >= ndev) return fail(VS_EINVAL, "device %d out of range", device);
    VS_HIP(hipSetDevice(device));
    const size_t esz = dtype_size(dtype);
    const int64_t rows_per_chunk = std::max<int64_t>(1, std::min<int64_t>(((int64_t)256 << 20) / ((int64_t)ld * (int64_t)esz), 1 << 20));
    DevBuf raw, f32, counts;
    VS_TRY(counts.alloc((size_t)n_rows * 8));
    // pass 1: non-zeros per row (of the values as stored, i.e. after fp16 rounding if requested)
    for (int64_t r = 0; r < n_rows; r += rows_per_chunk) {
        const int64_t nr = std::min(rows_per_chunk, n_rows - r);
        VS_TRY(stage_rows_f32(mat, dtype, ld, n_cols, r, nr, store_dtype == VS_F16, raw, f32));
        hipLaunchKernelGGL(count_nz_kernel<0>, dim3((unsigned)std::min<int64_t>(nr, 2048)), dim3(kSpThreads), 0, 0, f32.as<float>(), (int64_t)n_cols,
                           (int32_t)nr, n_cols, counts.as<int64_t>() + r);
        VS_HIP(hipGetLastError());
        VS_HIP(hipDeviceSynchronize());
    }
    std::vector<int64_t> hc((size_t)n_rows);
    VS_HIP(hipMemcpy(hc.data(), counts.p, (size_t)n_rows * 8, hipMemcpyDeviceToHost));
    int64_t nnz = 0, packets = 0;
    for (int64_t r = 0; r < n_rows; ++r) { nnz += hc[r]; packets += (hc[r] + 7) / 8; }
    if ((double)nnz > max_density * (double)n_rows * (double)n_cols || packets >= (1ll << 32))
        return vs_index_create_dense(mat, dtype, store_dtype, n_rows, n_cols, ld, device, out);
    vs_index* idx = nullptr;
    VS_TRY(vs_index_create_reserved(n_rows, packets, n_cols, store_dtype, device, &idx));
    struct Guard { vs_index* p; ~Guard() { if (p) vs_index_destroy(p); } } guard{idx};
    idx->logical_dense = true;
    // pass 2: dense rows -> CSR (ordered compaction on the device) -> packets
    DevBuf d_rp, d_cols, d_vals;
    for (int64_t r = 0; r < n_rows; r += rows_per_chunk) {
        const int64_t nr = std::min(rows_per_chunk, n_rows - r);
        std::vector<int64_t> rp((size_t)nr + 1);
        rp[0] = 0;
        for (int64_t i = 0; i < nr; ++i) rp[i + 1] = rp[i] + hc[r + i];
        const int64_t cn = rp[nr];
        VS_TRY(d_rp.reserve(rp.size() * 8));
        VS_HIP(hipMemcpy(d_rp.p, rp.data(), rp.size() * 8, hipMemcpyHostToDevice));
        VS_TRY(d_cols.reserve(std::max<size_t>((size_t)cn * 4, 16)));
        VS_TRY(d_vals.reserve(std::max<size_t>((size_t)cn * 4, 16)));
        VS_TRY(stage_rows_f32(mat, dtype, ld, n_cols, r, nr, store_dtype == VS_F16, raw, f32));
        hipLaunchKernelGGL(fill_csr_kernel<0>, dim3((unsigned)std::min<int64_t>(nr, 2048)), dim3(kSpThreads), 0, 0, f32.as<float>(), (int64_t)n_cols,
                           (int32_t)nr, n_cols, d_rp.as<int64_t>(), d_cols.as<int32_t>(), d_vals.as<float>(), cn);
        VS_HIP(hipGetLastError());
        VS_HIP(hipDeviceSynchronize());
        VS_TRY(vs_csr_append_rows(idx, d_rp.p, VS_I64, d_cols.p, VS_I32, d_vals.p, VS_F32, nr));
    }
    guard.p = nullptr;
    *out = idx;
    return VS_OK;
}

extern "C" int vs_index_export_dense(const vs_index* idx, void* mat, int dtype, int64_t ld) {
    if (!idx || !mat) return fail(VS_EINVAL, "NULL argument");
    if (idx->kind != VS_KIND_DENSE && !idx->logical_dense) return fail(VS_EINVAL, "not a dense index");
    if (dtype != VS_F32 && dtype != VS_F16) return fail(VS_EINVAL, "dtype must be VS_F32 or VS_F16");
    if (ld < idx->n_cols) return fail(VS_EINVAL, "ld < n_cols");
    VS_HIP(hipSetDevice(idx->device));
    if (idx->logical_dense) return export_dense_from_csr(idx, mat, dtype, ld);
    const int ldp = dense_ldp(idx->n_cols);
    const size_t esz = dtype_size(dtype);
    const bool dst_dev = is_device_ptr(mat);
    const int64_t rows_per_chunk = std::max<int64_t>(1, ((int64_t)256 << 20) / ((int64_t)ld * (int64_t)esz));
    DevBuf stage;
    for (int64_t r = 0; r < idx->n_rows; r += rows_per_chunk) {
        const int64_t nr = std::min(rows_per_chunk, idx->n_rows - r);
        void* ddst = (char*)mat + (size_t)r * ld * esz;
        if (!dst_dev) {
            VS_TRY(stage.reserve((size_t)nr * ld * esz));
            ddst = stage.p;
        }
        const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(nr * idx->n_cols, 256), 16384);
        const float* src = idx->mat.as<float>() + (size_t)r * ldp;
        if (dtype == VS_F32) hipLaunchKernelGGL((unpad_rows_kernel<float>), dim3(grid), dim3(256), 0, 0, src, ldp, nr, idx->n_cols, ld, (float*)ddst);
        else hipLaunchKernelGGL((unpad_rows_kernel<__half>), dim3(grid), dim3(256), 0, 0, src, ldp, nr, idx->n_cols, ld, (__half*)ddst);
        VS_HIP(hipGetLastError());
        VS_HIP(hipDeviceSynchronize());
        if (!dst_dev) {
            // rows are written with stride ld inside the stage; copy the used span
            const size_t bytes = ((size_t)(nr - 1) * ld + idx->n_cols) * esz;
            VS_HIP(hipMemcpy((char*)mat + (size_t)r * ld * esz, stage.p, bytes, hipMemcpyDeviceToHost));
        }
    }
    return VS_OK;
}


namespace {
__global__ void pool_finish_kernel(const uint32_t* keys, int64_t n, float* out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = unflip_f32(keys[i]);
        out[i] = x > 0.f ? x + 1.0f : expm1f(x) + 1.0f;           // elu1p after the max (monotone), vdr.py:73-75
    }
}
}  // namespace

// Encoder head, fused (vdr.py:72-75):  out[b, v] = elu1p( max_l  hidden[b, l, :] . W[v, :] ).
// hidden: [B, L, H] fp32 device (already LayerNorm'ed), W: [V, H] fp32 device (caller passes word_emb[shift:]).
// The [B, L, V] logits tensor of the reference (1.9 GB at B = 64, L = 256) is never materialised.
extern "C" int vs_head_project_pool(const float* hidden, const float* W, int32_t B, int32_t L, int32_t H, int32_t V, float* out,
                                    int device, void* stream) {
    if (!hidden || !W || !out || B <= 0 || L <= 0 || H <= 0 || V <= 0) return fail(VS_EINVAL, "bad argument");
    if (H % kDenseKC != 0) return fail(VS_EUNSUPPORTED, "hidden size %d is not a multiple of %d", H, kDenseKC);
    if (!is_device_ptr(hidden) || !is_device_ptr(W) || !is_device_ptr(out)) return fail(VS_EINVAL, "vs_head_project_pool takes device pointers");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return fail(VS_ENODEVICE, "no HIP device visible"); }
    VS_HIP(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    DevBuf& keys = device_scratch(device, kScratchHeadKeys);                            // kept between calls: one per encoder batch
    VS_TRY(keys.reserve((size_t)B * V * 4));
    VS_HIP(hipMemsetAsync(keys.p, 0, (size_t)B * V * 4, s));            // key 0 < key of any real number
    {
        ProfScope prof("head_project_pool", s);
        if (L > 64) {
            const int l_tiles = (L + 127) / 128;
            hipLaunchKernelGGL((dense_scores_kernel<2, 2, 2, 2>), dim3((unsigned)ceil_div64(V, 128), (unsigned)(B * l_tiles)), dim3(256), 0, s, hidden, W,
                               B * L, (int64_t)V, (int64_t)0, H, (uint64_t*)nullptr, (float*)nullptr, L, keys.as<uint32_t>(), (float*)nullptr, 0, (int64_t)0);
        } else {
            const int l_tiles = (L + 31) / 32;
            hipLaunchKernelGGL((dense_scores_kernel<1, 4, 1, 1>), dim3((unsigned)ceil_div64(V, 128), (unsigned)(B * l_tiles)), dim3(256), 0, s, hidden, W,
                               B * L, (int64_t)V, (int64_t)0, H, (uint64_t*)nullptr, (float*)nullptr, L, keys.as<uint32_t>(), (float*)nullptr, 0, (int64_t)0);
        }
    }
    VS_HIP(hipGetLastError());
    hipLaunchKernelGGL(pool_finish_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64((int64_t)B * V, 256), 8192)), dim3(256), 0, s, keys.as<uint32_t>(),
                       (int64_t)B * V, out);
    VS_HIP(hipGetLastError());
    if (!s) VS_HIP(hipStreamSynchronize(s));                             // with a stream the call is asynchronous
    return VS_OK;
}

int vs_dense_search(vs_index* idx, const void* q, int q_dtype, int64_t ldq, int32_t B, int32_t k, int64_t id_offset,
                    int64_t* out_ids, float* out_scores, hipStream_t s) {
    const int ldp = dense_ldp(idx->n_cols);
    const float* dq = nullptr;
    VS_TRY(prep_dense_queries(idx, q, q_dtype, ldq, B, ldp, s, &dq));
    const bool out_dev = is_device_ptr(out_ids);
    if (out_dev != is_device_ptr(out_scores)) return fail(VS_EINVAL, "out_ids and out_scores must both be host or both device pointers");
    int64_t* d_ids = out_ids;
    float* d_scores = out_scores;
    if (!out_dev) {
        VS_TRY(idx->ws_out_ids.reserve((size_t)B * k * 8));
        VS_TRY(idx->ws_out_scores.reserve((size_t)B * k * 4));
        d_ids = idx->ws_out_ids.as<int64_t>();
        d_scores = idx->ws_out_scores.as<float>();
    }
    const int64_t N = idx->n_rows;
    const int bs_max = (int)std::max<int64_t>(1, std::min<int64_t>(B, ((int64_t)1 << 30) / (N * 8)));
    VS_TRY(idx->ws_cand.reserve((size_t)bs_max * N * 8));
    const int passes = ceil_div(k, kMaxKShared);               // k > 2048: select in slices below an exclusive upper-bound key
    DevBuf upper;
    if (passes > 1) VS_TRY(upper.alloc((size_t)bs_max * 8));
    for (int b0 = 0; b0 < B; b0 += bs_max) {
        const int bs = std::min(bs_max, B - b0);
        {
            ProfScope prof("dense_scores", s);
            VS_TRY(launch_dense_scores(idx, dq + (size_t)b0 * ldp, bs, ldp, idx->ws_cand.as<uint64_t>(), nullptr, s));
        }
        for (int pass = 0; pass < passes; ++pass) {
            const int col0 = pass * kMaxKShared;
            MergeArgs m{};
            m.cand = idx->ws_cand.as<uint64_t>();
            m.n_cand = N;
            m.B = bs;
            m.k = std::min(k - col0, kMaxKShared);
            m.id_offset = id_offset;
            m.out_ids = d_ids + (size_t)b0 * k;
            m.out_scores = d_scores + (size_t)b0 * k;
            m.out_ld = k;
            m.col0 = col0;
            m.upper_out = passes > 1 ? upper.as<uint64_t>() : nullptr;
            m.upper_in = pass > 0 ? upper.as<uint64_t>() : nullptr;
            ProfScope prof("merge_topk", s);
            if (N > 2 * kWgCap) hipLaunchKernelGGL(select_topk_kernel<0>, dim3(std::min(bs, idx->cu_count * 2)), dim3(kScanThreads), 0, s, m);
            else hipLaunchKernelGGL(merge_topk_kernel<0>, dim3(std::min(bs, idx->cu_count * 2)), dim3(kScanThreads), 0, s, m);
        }
        VS_HIP(hipGetLastError());
    }
    if (!out_dev) {
        VS_HIP(hipMemcpyAsync(out_ids, d_ids, (size_t)B * k * 8, hipMemcpyDeviceToHost, s));
        VS_HIP(hipMemcpyAsync(out_scores, d_scores, (size_t)B * k * 4, hipMemcpyDeviceToHost, s));
        VS_HIP(hipStreamSynchronize(s));
    }
    if (passes > 1) VS_HIP(hipStreamSynchronize(s));            // `upper` is freed on return
    return VS_OK;
}

int vs_dense_scores(vs_index* idx, const void* q, int q_dtype, int64_t ldq, int32_t B, float* out_scores, hipStream_t s) {
    const int ldp = dense_ldp(idx->n_cols);
    const float* dq = nullptr;
    VS_TRY(prep_dense_queries(idx, q, q_dtype, ldq, B, ldp, s, &dq));
    const bool out_dev = is_device_ptr(out_scores);
    float* d_scores = out_scores;
    const int64_t N = idx->n_rows;
    if (!out_dev) {
        VS_TRY(idx->ws_out_scores.reserve((size_t)B * N * 4));
        d_scores = idx->ws_out_scores.as<float>();
    }
    VS_TRY(launch_dense_scores(idx, dq, B, ldp, nullptr, d_scores, s));
    if (!out_dev) {
        VS_HIP(hipMemcpyAsync(out_scores, d_scores, (size_t)B * N * 4, hipMemcpyDeviceToHost, s));
        VS_HIP(hipStreamSynchronize(s));
    }
    return VS_OK;
}
