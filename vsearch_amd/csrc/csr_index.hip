// csr_index.hip -- SparseIndex / BoTIndex device container and its search entry points.
//   reference: src/ir/retriever/index.py:128-218 (containers), :88-94 (search)
#include "common.h"
#include "csr_scan.h"
#include "csr_scan_mq.h"
#include "bp_walk.h"
#include "bp_refine.h"
#include "bp_bin.h"
// The three experimental walks of round 3 (flat worklists, streamed flat walk, two accumulator sets: all correct, all slower than the
// list walk -- docs/EXPERIMENTS.md) are lab results, not components: compiled only with -DVS_EXPERIMENTAL_WALKS (make EXPERIMENTAL=1).
// (bp_flat.h itself stays in: the bag-of-token walk, bp_bin.h, shares its candidate handling)
#ifdef VS_EXPERIMENTAL_WALKS
#include "bp_stream.h"
#include "bp_duo.h"
#else
namespace vs { constexpr int kDuoMaxK = 0, kDuoEntCap = 0, kDuoQT = 4; }
#endif
#include "bp_quad.h"
#include "synth_device.h"

#include <algorithm>
#include <chrono>

using namespace vs;

// =================================================================================================
// format conversion kernels
// =================================================================================================
namespace {

template <class T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__half>(__half v) { return __half2float(v); }

// One wave per row: copies the row's (col, val) run into its packet range and pads the tail.
// flags[0] |= 1: column out of range;  flags[0] |= 2: non-unit value for a binary index.
template <class RP, class CI, class VT>
__global__ void fill_packets_kernel(const RP* rowptr, const CI* colidx, const VT* values, int64_t row_begin,
                                    int64_t row_end, int64_t src_base, const uint32_t* pk_ptr, uint16_t* cols,
                                    void* vals, int store_dtype, int32_t n_cols, int* flags) {
    const int lane = threadIdx.x & 63;
    const int64_t row = row_begin + (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= row_end) return;
    const int64_t s0 = (int64_t)rowptr[row - row_begin] - src_base;
    const int64_t len = (int64_t)rowptr[row - row_begin + 1] - (int64_t)rowptr[row - row_begin];
    const int64_t d0 = (int64_t)pk_ptr[row] * 8;
    const int64_t dn = ((int64_t)pk_ptr[row + 1] - pk_ptr[row]) * 8;
    int bad = 0;
    for (int64_t j = lane; j < dn; j += 64) {
        uint16_t c = (uint16_t)n_cols;
        float v = 0.f;
        if (j < len) {
            const int64_t ci = (int64_t)colidx[s0 + j];
            if (ci < 0 || ci >= n_cols) bad |= 1;
            c = (uint16_t)ci;
            v = values ? to_f32<VT>(values[s0 + j]) : 1.0f;
            if (store_dtype == VS_NONE && v != 1.0f) bad |= 2;
        }
        cols[d0 + j] = c;
        if (store_dtype == VS_F32) reinterpret_cast<float*>(vals)[d0 + j] = v;
        else if (store_dtype == VS_F16) reinterpret_cast<__half*>(vals)[d0 + j] = __float2half_rn(v);
    }
    if (bad) atomicOr(flags, bad);
}

// true nnz per row = entries whose column id is not the pad id
__global__ void row_nnz_kernel(const uint32_t* pk_ptr, const uint16_t* cols, int64_t n_rows, int32_t n_cols, int64_t* out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int64_t d0 = (int64_t)pk_ptr[row] * 8, d1 = (int64_t)pk_ptr[row + 1] * 8;
    int cnt = 0;
    for (int64_t j = d0 + lane; j < d1; j += 64) cnt += cols[j] != (uint16_t)n_cols;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) out[row] = cnt;
}

template <class VT>
__global__ void export_rows_kernel(const uint32_t* pk_ptr, const uint16_t* cols, const void* vals, int store_dtype,
                                   const int64_t* rowptr, int64_t row_begin, int64_t row_end, int64_t dst_base,
                                   int64_t* colidx, VT* values) {
    const int lane = threadIdx.x & 63;
    const int64_t row = row_begin + (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= row_end) return;
    const int64_t d0 = (int64_t)pk_ptr[row] * 8;
    const int64_t s0 = rowptr[row] - dst_base, len = rowptr[row + 1] - rowptr[row];
    for (int64_t j = lane; j < len; j += 64) {
        colidx[s0 + j] = cols[d0 + j];
        float v = 1.0f;
        if (store_dtype == VS_F32) v = reinterpret_cast<const float*>(vals)[d0 + j];
        else if (store_dtype == VS_F16) v = __half2float(reinterpret_cast<const __half*>(vals)[d0 + j]);
        if constexpr (sizeof(VT) == 4) values[s0 + j] = v;
        else values[s0 + j] = __float2half_rn(v);
    }
}

// Synthetic rows, one wave per row: set the row's distinct columns in an LDS bitmap, then emit them
// in ascending order by bitmap scan (no sort).  4 waves per workgroup, each with its own bitmap.
constexpr int kSynthWaves = 4;
__global__ __launch_bounds__(kSynthWaves * 64) void synth_rows_kernel(uint64_t seed, int64_t row0, int64_t n_rows,
                                                                     int32_t n_cols, int32_t nnz, int kind, int val_law,
                                                                     const uint32_t* pk_ptr, uint16_t* cols, void* vals,
                                                                     int store_dtype) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int words = (n_cols + 31) / 32;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t* bm = reinterpret_cast<uint32_t*>(smem) + (size_t)w * words;
    const int wpl = (words + 63) / 64;                       // bitmap words per lane
    for (int64_t r = (int64_t)blockIdx.x * kSynthWaves + w; r < n_rows; r += (int64_t)gridDim.x * kSynthWaves) {
        const int64_t row = row0 + r;
        const int64_t len = synth_row_len(seed, row, kind, nnz, n_cols);
        const uint64_t key = hash3(seed, (uint64_t)row, 0x4B4559ull);
        for (int i = lane; i < words; i += 64) bm[i] = 0;
        __builtin_amdgcn_wave_barrier();
        for (int64_t j = lane; j < len; j += 64) {
            const uint32_t c = synth_col(key, (uint32_t)j, (uint32_t)len, kind, (uint32_t)n_cols);
            atomicOr(&bm[c >> 5], 1u << (c & 31));
        }
        __builtin_amdgcn_wave_barrier();
        // lane owns words [lane*wpl, lane*wpl + wpl): count, exclusive-scan across lanes, emit
        const int wb = lane * wpl, we = min(words, wb + wpl);
        int mine = 0;
        for (int i = wb; i < we; ++i) mine += __popc(bm[i]);
        int incl = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        int pos = incl - mine;
        const int64_t d0 = (int64_t)pk_ptr[r] * 8;
        const int64_t dn = ((int64_t)pk_ptr[r + 1] - pk_ptr[r]) * 8;
        for (int i = wb; i < we; ++i) {
            uint32_t bits = bm[i];
            while (bits) {
                const int b = __ffs(bits) - 1;
                bits &= bits - 1;
                const uint32_t c = (uint32_t)i * 32 + b;
                cols[d0 + pos] = (uint16_t)c;
                const float v = kind == 1 ? 1.0f : synth_val(seed, row, c, val_law);
                if (store_dtype == VS_F32) reinterpret_cast<float*>(vals)[d0 + pos] = v;
                else if (store_dtype == VS_F16) reinterpret_cast<__half*>(vals)[d0 + pos] = __float2half_rn(v);
                ++pos;
            }
        }
        for (int64_t j = len + lane; j < dn; j += 64) {
            cols[d0 + j] = (uint16_t)n_cols;
            if (store_dtype == VS_F32) reinterpret_cast<float*>(vals)[d0 + j] = 0.f;
            else if (store_dtype == VS_F16) reinterpret_cast<__half*>(vals)[d0 + j] = __float2half_rn(0.f);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// q (fp32 | fp16, leading dim ldq) -> contiguous fp32 [B, n_cols]; round_f16: emulate
// `q_embs.type(self.vector.dtype)` for an fp16 index (index.py:89).
template <class T>
__global__ void prep_queries_kernel(const T* q, int64_t ldq, int32_t B, int32_t n_cols, int round_f16, float* out) {
    const int64_t n = (int64_t)B * n_cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / n_cols, c = i % n_cols;
        float v = to_f32<T>(q[b * ldq + c]);
        if (round_f16) v = __half2float(__float2half_rn(v));
        out[i] = v;
    }
}

// keys hold 32-bit ids: a candidate whose id does not fit (or the pad sentinel 2^32 - 1) is dropped, never aliased to another document
__global__ void keys_from_pairs_kernel(const int64_t* ids, const float* scores, int64_t n, uint64_t* keys) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t id = ids[i];
        keys[i] = (id >= 0 && id < 0xFFFFFFFFll) ? make_key(scores[i], (uint32_t)id) : 0ull;
    }
}

int pick_lanes_per_row(int64_t n_packets, int64_t n_rows) {
    const double mp = n_rows > 0 ? (double)n_packets / (double)n_rows : 1.0;   // packets per row
    int g = 4;
    while (g < 64 && mp / g > 4.0) g <<= 1;   // aim at <= 4 packets per lane per row: 96 packets -> 32 lanes, 11 -> 4
    return g;
}

int new_index(int device, vs_index** out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(VS_ENODEVICE, "no HIP device visible: libvsearch_hip has no CPU fallback");
    }
    if (device < 0 || device >= ndev) return fail(VS_EINVAL, "device %d out of range (have %d)", device, ndev);
    VS_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    VS_HIP(hipGetDeviceProperties(&prop, device));
    vs_index* idx = new (std::nothrow) vs_index();
    if (!idx) return fail(VS_ENOMEM, "host allocation failed");
    idx->device = device;
    idx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    *out = idx;
    return VS_OK;
}

int alloc_csr_storage(vs_index* idx) {
    VS_TRY(idx->pk_ptr.alloc((size_t)(idx->n_rows + 1) * 4));
    VS_TRY(idx->cols.alloc(std::max<size_t>((size_t)idx->n_packets * 16, 16)));
    if (idx->store_dtype == VS_F32) VS_TRY(idx->vals.alloc(std::max<size_t>((size_t)idx->n_packets * 32, 32)));
    if (idx->store_dtype == VS_F16) VS_TRY(idx->vals.alloc(std::max<size_t>((size_t)idx->n_packets * 16, 16)));
    idx->lanes_per_row = pick_lanes_per_row(idx->n_packets, idx->n_rows);
    return VS_OK;
}

template <class F>
int dispatch_rp_ci(int rp_dt, int ci_dt, F&& f) {
    if (rp_dt == VS_I64 && ci_dt == VS_I64) return f((const int64_t*)nullptr, (const int64_t*)nullptr);
    if (rp_dt == VS_I64 && ci_dt == VS_I32) return f((const int64_t*)nullptr, (const int32_t*)nullptr);
    if (rp_dt == VS_I32 && ci_dt == VS_I64) return f((const int32_t*)nullptr, (const int64_t*)nullptr);
    if (rp_dt == VS_I32 && ci_dt == VS_I32) return f((const int32_t*)nullptr, (const int32_t*)nullptr);
    return fail(VS_EINVAL, "rowptr/colidx dtype must be VS_I32 or VS_I64");
}

}  // namespace

// =================================================================================================
// creation
// =================================================================================================
namespace {

int validate_csr_args(const void* rowptr, int rowptr_dtype, const void* /*colidx*/, int col_dtype, const void* values, int val_dtype) {
    if (!rowptr) return fail(VS_EINVAL, "rowptr is NULL");
    if (values && val_dtype != VS_F32 && val_dtype != VS_F16) return fail(VS_EINVAL, "val_dtype must be VS_F32 or VS_F16");
    if ((rowptr_dtype != VS_I32 && rowptr_dtype != VS_I64) || (col_dtype != VS_I32 && col_dtype != VS_I64))
        return fail(VS_EINVAL, "rowptr/colidx dtype must be VS_I32 or VS_I64");
    return VS_OK;
}

// Appends n_chunk CSR rows behind the rows already in `idx` (capacity reserved beforehand).
int append_csr_rows(vs_index* idx, const void* rowptr, int rowptr_dtype, const void* colidx, int col_dtype,
                    const void* values, int val_dtype, int64_t n_chunk) {
    const int32_t n_cols = idx->n_cols;
    const int store_dtype = idx->store_dtype;
    // row pointers on the host (they size everything)
    const size_t rps = dtype_size(rowptr_dtype);
    std::vector<char> rp_host((size_t)(n_chunk + 1) * rps);
    if (is_device_ptr(rowptr)) VS_HIP(hipMemcpy(rp_host.data(), rowptr, rp_host.size(), hipMemcpyDeviceToHost));
    else memcpy(rp_host.data(), rowptr, rp_host.size());
    auto rp_at = [&](int64_t i) -> int64_t {
        return rowptr_dtype == VS_I64 ? reinterpret_cast<const int64_t*>(rp_host.data())[i]
                                      : (int64_t) reinterpret_cast<const int32_t*>(rp_host.data())[i];
    };
    const int64_t row_base = idx->n_rows;                      // global index of the chunk's first row
    if (row_base + n_chunk > idx->rows_cap) return fail(VS_EINVAL, "append exceeds the reserved %lld rows", (long long)idx->rows_cap);
    std::vector<uint32_t> pk((size_t)n_chunk + 1);
    int64_t acc = idx->n_packets;
    pk[0] = (uint32_t)acc;
    for (int64_t r = 0; r < n_chunk; ++r) {
        const int64_t len = rp_at(r + 1) - rp_at(r);
        if (len < 0 || len > n_cols) return fail(VS_EINVAL, "row %lld has %lld entries (n_cols = %d)", (long long)(row_base + r), (long long)len, n_cols);
        acc += (len + 7) / 8;
        if (acc >= (1ll << 32)) return fail(VS_EUNSUPPORTED, "index exceeds 2^32 packets on one device");
        pk[r + 1] = (uint32_t)acc;
    }
    if (acc > idx->packets_cap) return fail(VS_EINVAL, "append exceeds the reserved %lld packets", (long long)idx->packets_cap);
    const int64_t nnz_chunk = rp_at(n_chunk) - rp_at(0);
    if (nnz_chunk > 0 && !colidx) return fail(VS_EINVAL, "colidx is NULL");
    VS_HIP(hipMemcpy(idx->pk_ptr.as<uint32_t>() + row_base, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));

    DevBuf flags;
    VS_TRY(flags.alloc(4));
    VS_HIP(hipMemset(flags.p, 0, 4));
    const bool src_dev = is_device_ptr(colidx);
    const size_t cis = dtype_size(col_dtype), vsz = values ? dtype_size(val_dtype) : 0;
    const int64_t kMaxChunkNnz = 32ll << 20;
    DevBuf st_rp, st_ci, st_v;
    int64_t r = 0;
    while (r < n_chunk) {
        int64_t r_end = r + 1;
        while (r_end < n_chunk && rp_at(r_end + 1) - rp_at(r) <= kMaxChunkNnz && r_end - r < (1 << 22)) ++r_end;
        const int64_t base = rp_at(r), cnt = rp_at(r_end) - base;
        const void *d_rp = nullptr, *d_ci = nullptr, *d_v = nullptr;
        // rowptr slice always re-uploaded from the host copy (cheap) so the kernel can index it from 0
        VS_TRY(st_rp.reserve((size_t)(r_end - r + 1) * rps));
        VS_HIP(hipMemcpy(st_rp.p, rp_host.data() + (size_t)r * rps, (size_t)(r_end - r + 1) * rps, hipMemcpyHostToDevice));
        d_rp = st_rp.p;
        int64_t src_base = base;
        if (src_dev) {
            d_ci = colidx;
            d_v = values;
            src_base = rp_at(0);                                   // device arrays are indexed absolutely
        } else if (cnt > 0) {
            VS_TRY(st_ci.reserve((size_t)cnt * cis));
            VS_HIP(hipMemcpy(st_ci.p, (const char*)colidx + (size_t)(base - rp_at(0)) * cis, (size_t)cnt * cis, hipMemcpyHostToDevice));
            d_ci = st_ci.p;
            if (values) {
                VS_TRY(st_v.reserve((size_t)cnt * vsz));
                VS_HIP(hipMemcpy(st_v.p, (const char*)values + (size_t)(base - rp_at(0)) * vsz, (size_t)cnt * vsz, hipMemcpyHostToDevice));
                d_v = st_v.p;
            }
        }
        const int wpb = 4;
        const unsigned grid = (unsigned)ceil_div64(r_end - r, wpb);
        const int64_t g0 = row_base + r, g1 = row_base + r_end;       // global row range of this slice
        int rc = dispatch_rp_ci(rowptr_dtype, col_dtype, [&](auto* rp_t, auto* ci_t) -> int {
            using RP = std::remove_cv_t<std::remove_pointer_t<decltype(rp_t)>>;
            using CI = std::remove_cv_t<std::remove_pointer_t<decltype(ci_t)>>;
            if (values && val_dtype == VS_F16)
                hipLaunchKernelGGL((fill_packets_kernel<RP, CI, __half>), dim3(grid), dim3(wpb * 64), 0, 0, (const RP*)d_rp, (const CI*)d_ci,
                                   (const __half*)d_v, g0, g1, src_base, idx->pk_ptr.as<uint32_t>(), idx->cols.as<uint16_t>(),
                                   idx->vals.p, store_dtype, n_cols, flags.as<int>());
            else
                hipLaunchKernelGGL((fill_packets_kernel<RP, CI, float>), dim3(grid), dim3(wpb * 64), 0, 0, (const RP*)d_rp, (const CI*)d_ci,
                                   (const float*)d_v, g0, g1, src_base, idx->pk_ptr.as<uint32_t>(), idx->cols.as<uint16_t>(),
                                   idx->vals.p, store_dtype, n_cols, flags.as<int>());
            return VS_OK;
        });
        VS_TRY(rc);
        VS_HIP(hipGetLastError());
        VS_HIP(hipDeviceSynchronize());       // staging buffers are reused by the next slice
        r = r_end;
    }
    int hflags = 0;
    VS_HIP(hipMemcpy(&hflags, flags.p, 4, hipMemcpyDeviceToHost));
    if (hflags & 1) return fail(VS_EINVAL, "column index out of range [0, %d)", n_cols);
    if (hflags & 2) return fail(VS_EINVAL, "store_dtype VS_NONE (binary index) requires every value == 1");
    idx->n_rows = row_base + n_chunk;
    idx->n_packets = acc;
    idx->bp_ready = false;                 // the column-grouped copy no longer matches: rebuilt on the next sparse search
    idx->bp_tried = false;
    idx->nnz += nnz_chunk;
    idx->lanes_per_row = pick_lanes_per_row(idx->n_packets, idx->n_rows);
    return VS_OK;
}

int reserve_csr(vs_index* idx, int64_t rows_cap, int64_t packets_cap) {
    idx->rows_cap = rows_cap;
    idx->packets_cap = packets_cap;
    VS_TRY(idx->pk_ptr.alloc((size_t)(rows_cap + 1) * 4));
    VS_HIP(hipMemset(idx->pk_ptr.p, 0, 4));
    VS_TRY(idx->cols.alloc(std::max<size_t>((size_t)packets_cap * 16, 16)));
    if (idx->store_dtype == VS_F32) VS_TRY(idx->vals.alloc(std::max<size_t>((size_t)packets_cap * 32, 32)));
    if (idx->store_dtype == VS_F16) VS_TRY(idx->vals.alloc(std::max<size_t>((size_t)packets_cap * 16, 16)));
    return VS_OK;
}

int check_csr_shape(int64_t n_rows, int32_t n_cols, int store_dtype) {
    if (n_rows < 0 || n_cols <= 0) return fail(VS_EINVAL, "bad shape");
    if (n_cols > 65535) return fail(VS_EUNSUPPORTED, "n_cols = %d > 65535: column ids are stored as uint16", n_cols);
    if (n_rows >= (1ll << 32) - 1) return fail(VS_EUNSUPPORTED, "n_rows must fit in 32 bits");
    if (store_dtype != VS_F32 && store_dtype != VS_F16 && store_dtype != VS_NONE) return fail(VS_EINVAL, "bad store_dtype");
    return VS_OK;
}

}  // namespace

// internal (dense.hip): append rows to a reserved CSR index from device / host arrays
int vs_csr_append_rows(vs_index* idx, const void* rowptr, int rowptr_dtype, const void* colidx, int col_dtype,
                       const void* values, int val_dtype, int64_t n_rows) {
    return append_csr_rows(idx, rowptr, rowptr_dtype, colidx, col_dtype, values, val_dtype, n_rows);
}

extern "C" int vs_index_create_reserved(int64_t rows_cap, int64_t packets_cap, int32_t n_cols, int store_dtype, int device, vs_index** out) {
    if (!out) return fail(VS_EINVAL, "out is NULL");
    *out = nullptr;
    VS_TRY(check_csr_shape(rows_cap, n_cols, store_dtype));
    if (packets_cap < 0 || packets_cap >= (1ll << 32)) return fail(VS_EUNSUPPORTED, "packets_cap must be in [0, 2^32)");
    vs_index* idx = nullptr;
    VS_TRY(new_index(device, &idx));
    struct Guard { vs_index* p; ~Guard() { if (p) vs_index_destroy(p); } } guard{idx};
    idx->kind = VS_KIND_CSR;
    idx->store_dtype = store_dtype;
    idx->n_rows = 0;
    idx->n_cols = n_cols;
    VS_TRY(reserve_csr(idx, rows_cap, packets_cap));
    guard.p = nullptr;
    *out = idx;
    return VS_OK;
}

extern "C" int vs_index_append_csr(vs_index* idx, const void* rowptr, int rowptr_dtype, const void* colidx, int col_dtype,
                                   const void* values, int val_dtype, int64_t n_rows) {
    if (!idx || idx->kind != VS_KIND_CSR) return fail(VS_EINVAL, "not a CSR index");
    if (n_rows < 0) return fail(VS_EINVAL, "bad n_rows");
    VS_TRY(validate_csr_args(rowptr, rowptr_dtype, colidx, col_dtype, values, val_dtype));
    VS_HIP(hipSetDevice(idx->device));
    return append_csr_rows(idx, rowptr, rowptr_dtype, colidx, col_dtype, values, val_dtype, n_rows);
}

extern "C" int vs_index_create_csr(const void* rowptr, int rowptr_dtype, const void* colidx, int col_dtype,
                                   const void* values, int val_dtype, int store_dtype, int64_t n_rows, int32_t n_cols,
                                   int device, vs_index** out) {
    if (!out) return fail(VS_EINVAL, "out is NULL");
    *out = nullptr;
    VS_TRY(check_csr_shape(n_rows, n_cols, store_dtype));
    VS_TRY(validate_csr_args(rowptr, rowptr_dtype, colidx, col_dtype, values, val_dtype));
    // exact packet count from the row pointers
    const size_t rps = dtype_size(rowptr_dtype);
    std::vector<char> rp_host((size_t)(n_rows + 1) * rps);
    if (is_device_ptr(rowptr)) VS_HIP(hipMemcpy(rp_host.data(), rowptr, rp_host.size(), hipMemcpyDeviceToHost));
    else memcpy(rp_host.data(), rowptr, rp_host.size());
    int64_t packets = 0;
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t len = rowptr_dtype == VS_I64 ? reinterpret_cast<const int64_t*>(rp_host.data())[r + 1] - reinterpret_cast<const int64_t*>(rp_host.data())[r]
                                                   : (int64_t) reinterpret_cast<const int32_t*>(rp_host.data())[r + 1] - reinterpret_cast<const int32_t*>(rp_host.data())[r];
        if (len < 0 || len > n_cols) return fail(VS_EINVAL, "row %lld has %lld entries (n_cols = %d)", (long long)r, (long long)len, n_cols);
        packets += (len + 7) / 8;
    }
    if (packets >= (1ll << 32)) return fail(VS_EUNSUPPORTED, "index exceeds 2^32 packets on one device");
    vs_index* idx = nullptr;
    VS_TRY(vs_index_create_reserved(n_rows, packets, n_cols, store_dtype, device, &idx));
    int rc = append_csr_rows(idx, rowptr, rowptr_dtype, colidx, col_dtype, values, val_dtype, n_rows);
    if (rc != VS_OK) {
        vs_index_destroy(idx);
        return rc;
    }
    *out = idx;
    return VS_OK;
}

extern "C" int vs_index_create_synthetic(uint64_t seed, int64_t row0, int64_t n_rows, int32_t n_cols, int32_t nnz, int kind,
                                         int val_law, int store_dtype, int device, vs_index** out) {
    if (!out) return fail(VS_EINVAL, "out is NULL");
    *out = nullptr;
    if (n_rows < 0 || n_cols <= 0 || n_cols > 65535 || nnz <= 0 || nnz > n_cols) return fail(VS_EINVAL, "bad synthetic shape");
    if (kind < 0 || kind > 2) return fail(VS_EINVAL, "kind must be 0 (fixed nnz), 1 (bag-of-token) or 2 (fixed nnz, skewed column popularity)");
    if (kind == 1) store_dtype = VS_NONE;
    vs_index* idx = nullptr;
    VS_TRY(new_index(device, &idx));
    struct Guard { vs_index* p; ~Guard() { if (p) vs_index_destroy(p); } } guard{idx};
    idx->kind = VS_KIND_CSR;
    idx->store_dtype = store_dtype;
    idx->n_rows = n_rows;
    idx->n_cols = n_cols;
    std::vector<uint32_t> pk((size_t)n_rows + 1);
    int64_t acc = 0, nz = 0;
    pk[0] = 0;
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t len = synth_row_len(seed, row0 + r, kind, nnz, n_cols);
        nz += len;
        acc += (len + 7) / 8;
        if (acc >= (1ll << 32)) return fail(VS_EUNSUPPORTED, "index exceeds 2^32 packets on one device");
        pk[r + 1] = (uint32_t)acc;
    }
    idx->nnz = nz;
    idx->n_packets = acc;
    idx->rows_cap = n_rows;
    idx->packets_cap = acc;
    VS_TRY(alloc_csr_storage(idx));
    VS_HIP(hipMemcpy(idx->pk_ptr.p, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
    const int words = (n_cols + 31) / 32;
    const size_t lds = (size_t)kSynthWaves * words * 4;
    const unsigned grid = (unsigned)std::min<int64_t>(std::max<int64_t>(1, ceil_div64(n_rows, kSynthWaves)), (int64_t)idx->cu_count * 32);
    hipLaunchKernelGGL(synth_rows_kernel, dim3(grid), dim3(kSynthWaves * 64), lds, 0, seed, row0, n_rows, n_cols, nnz, kind, val_law,
                       idx->pk_ptr.as<uint32_t>(), idx->cols.as<uint16_t>(), idx->vals.p, store_dtype);
    VS_HIP(hipGetLastError());
    VS_HIP(hipDeviceSynchronize());
    guard.p = nullptr;
    *out = idx;
    return VS_OK;
}

// =================================================================================================
// info / export / destroy
// =================================================================================================
constexpr int kQT = 8;   // queries per pass of the multi-query scan (csr_scan_mq.h)

static int64_t csr_bytes_per_pass(const vs_index* idx) {
    const int64_t per_packet = 16 + (idx->store_dtype == VS_F32 ? 32 : idx->store_dtype == VS_F16 ? 16 : 0);
    return idx->n_packets * per_packet + (idx->n_rows + 1) * 4;
}

extern "C" int vs_index_info(const vs_index* idx, vs_index_info_t* o) {
    if (!idx || !o) return fail(VS_EINVAL, "NULL argument");
    memset(o, 0, sizeof(*o));
    o->kind = idx->logical_dense ? VS_KIND_DENSE : idx->kind;
    o->store_dtype = idx->store_dtype;
    o->n_rows = idx->n_rows;
    o->n_cols = idx->n_cols;
    o->device = idx->device;
    o->nnz = idx->nnz;
    o->n_packets = idx->n_packets;
    o->last_scan_bytes = idx->last_scan_bytes;
    o->last_path = idx->last_path;
    o->last_walk_postings = idx->last_walk_postings;
    o->head_columns = idx->bp_ready ? idx->bp_n_head : 0;
    o->postings_state = idx->bp_ready ? 1 : idx->bp_state;
    o->postings_walk = !idx->bp_ready ? -1 : idx->bp_quad ? 4 : (idx->store_dtype == VS_NONE && idx->bp_walk_pref != 0) ? 5 : 0;
    o->reserved0 = 0;
    if (idx->last_path == 3 && idx->last_plan_dev) {               // the filter search keeps its plan on the device: read it now
        int64_t hp[6] = {0, 0, 0, 0, 0, 0};
        VS_HIP(hipSetDevice(idx->device));
        VS_HIP(hipDeviceSynchronize());
        VS_HIP(hipMemcpy(hp, idx->last_plan_dev, sizeof(hp), hipMemcpyDeviceToHost));
        o->last_scan_bytes = hp[4] * idx->last_plan_rs + hp[2] * idx->last_plan_blocks * 4;      // records + one directory word per (entry, block)
        o->last_walk_postings = hp[5];
    }
    if (idx->last_path == 3 && idx->last_flags && idx->last_flags_n > 0) {
        std::vector<uint32_t> h((size_t)idx->last_flags_n);
        VS_HIP(hipSetDevice(idx->device));
        VS_HIP(hipDeviceSynchronize());
        VS_HIP(hipMemcpy(h.data(), idx->last_flags, h.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t f : h) o->last_fallbacks += f ? 1 : 0;
    }
    o->aux_bytes = idx->bp_ready ? (int64_t)(idx->bp_dir.bytes + idx->bp_base.bytes + idx->bp_rec.bytes + idx->bp_strip.bytes + idx->bp_ovf.bytes) : 0;
    if (idx->kind == VS_KIND_CSR) {
        o->bytes_per_pass = csr_bytes_per_pass(idx);
        o->device_bytes = (int64_t)(idx->pk_ptr.bytes + idx->cols.bytes + idx->vals.bytes);
        o->lanes_per_row = idx->lanes_per_row;
        o->queries_per_pass = idx->last_qt > 0 ? idx->last_qt : (idx->qt_pref == 1 ? 1 : kQT);
    } else {
        o->bytes_per_pass = idx->n_rows * (int64_t)idx->n_cols * (idx->store_dtype == VS_F16 ? 2 : 4);
        o->device_bytes = (int64_t)idx->mat.bytes;
        o->queries_per_pass = 0;
    }
    return VS_OK;
}

extern "C" void vs_index_destroy(vs_index* idx) {
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    delete idx;
}

extern "C" int vs_index_export_csr(const vs_index* idx, int64_t* rowptr, int64_t* colidx, void* values, int val_dtype) {
    if (!idx || !rowptr) return fail(VS_EINVAL, "NULL argument");
    if (idx->kind != VS_KIND_CSR) return fail(VS_EINVAL, "not a CSR index");
    if (values && val_dtype != VS_F32 && val_dtype != VS_F16) return fail(VS_EINVAL, "val_dtype must be VS_F32 or VS_F16");
    VS_HIP(hipSetDevice(idx->device));
    const int64_t n = idx->n_rows;
    DevBuf d_len;
    VS_TRY(d_len.alloc(std::max<size_t>((size_t)n * 8, 8)));
    if (n > 0) {
        hipLaunchKernelGGL(row_nnz_kernel, dim3((unsigned)ceil_div64(n, 4)), dim3(256), 0, 0, idx->pk_ptr.as<uint32_t>(),
                           idx->cols.as<uint16_t>(), n, idx->n_cols, d_len.as<int64_t>());
        VS_HIP(hipGetLastError());
    }
    std::vector<int64_t> rp((size_t)n + 1);
    VS_HIP(hipMemcpy(rp.data() + 1, d_len.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    rp[0] = 0;
    for (int64_t r = 0; r < n; ++r) rp[r + 1] += rp[r];
    if (is_device_ptr(rowptr)) VS_HIP(hipMemcpy(rowptr, rp.data(), rp.size() * 8, hipMemcpyHostToDevice));
    else memcpy(rowptr, rp.data(), rp.size() * 8);
    if (!colidx) return VS_OK;
    const int64_t nnz = rp[n];
    DevBuf d_rp;
    VS_TRY(d_rp.alloc(rp.size() * 8));
    VS_HIP(hipMemcpy(d_rp.p, rp.data(), rp.size() * 8, hipMemcpyHostToDevice));
    const bool dst_dev = is_device_ptr(colidx);
    const size_t vsz = values ? dtype_size(val_dtype) : 4;
    DevBuf st_ci, st_v;
    const int64_t kMaxChunkNnz = 32ll << 20;
    int64_t r = 0;
    while (r < n) {
        int64_t r_end = r + 1;
        while (r_end < n && rp[r_end + 1] - rp[r] <= kMaxChunkNnz) ++r_end;
        const int64_t base = rp[r], cnt = rp[r_end] - base;
        int64_t* o_ci;
        void* o_v;
        int64_t dst_base;
        if (dst_dev) {
            o_ci = colidx;
            o_v = values;
            dst_base = 0;
        } else {
            VS_TRY(st_ci.reserve(std::max<size_t>((size_t)cnt * 8, 8)));
            VS_TRY(st_v.reserve(std::max<size_t>((size_t)cnt * vsz, 8)));
            o_ci = st_ci.as<int64_t>();
            o_v = st_v.p;
            dst_base = base;
        }
        DevBuf dummy;
        if (!values && dst_dev) { VS_TRY(dummy.alloc(std::max<size_t>((size_t)nnz * 4, 8))); o_v = dummy.p; dst_base = 0; }
        const unsigned grid = (unsigned)ceil_div64(r_end - r, 4);
        if (values && val_dtype == VS_F16)
            hipLaunchKernelGGL((export_rows_kernel<__half>), dim3(grid), dim3(256), 0, 0, idx->pk_ptr.as<uint32_t>(), idx->cols.as<uint16_t>(),
                               idx->vals.p, idx->store_dtype, d_rp.as<int64_t>(), r, r_end, dst_base, o_ci, (__half*)o_v);
        else
            hipLaunchKernelGGL((export_rows_kernel<float>), dim3(grid), dim3(256), 0, 0, idx->pk_ptr.as<uint32_t>(), idx->cols.as<uint16_t>(),
                               idx->vals.p, idx->store_dtype, d_rp.as<int64_t>(), r, r_end, dst_base, o_ci, (float*)o_v);
        VS_HIP(hipGetLastError());
        VS_HIP(hipDeviceSynchronize());
        if (!dst_dev && cnt > 0) {
            VS_HIP(hipMemcpy(colidx + base, st_ci.p, (size_t)cnt * 8, hipMemcpyDeviceToHost));
            if (values) VS_HIP(hipMemcpy((char*)values + (size_t)base * vsz, st_v.p, (size_t)cnt * vsz, hipMemcpyDeviceToHost));
        }
        r = r_end;
    }
    return VS_OK;
}

// =================================================================================================
// search
// =================================================================================================
namespace {

struct ScanPlan {
    int nchunk;
    int64_t rows_per_chunk;
    int grid;
};

ScanPlan plan_scan(const vs_index* idx, int B) {
    ScanPlan p;
    const int64_t min_rows = 512;
    int64_t rpc = std::max<int64_t>(min_rows, ceil_div64(idx->n_rows, idx->cu_count));
    p.rows_per_chunk = rpc;
    p.nchunk = (int)std::max<int64_t>(1, ceil_div64(idx->n_rows, rpc));
    p.grid = (int)std::min<int64_t>((int64_t)B * p.nchunk, idx->cu_count);
    return p;
}

// Row chunks for `units` concurrent scans (query tiles, or single queries on the Qt = 1 path).  Work items = units x
// chunks.  Every item pays a table / image build and top-k sorts, and -- more important -- workgroups that sweep the
// SAME rows at the same time for different units share the stream through L2 / Infinity Cache, so chunks are as few
// and as long as still fill the CUs: many units -> 1-2 chunks, one unit -> one chunk per CU.
int choose_chunks(const vs_index* idx, int units, int max_nchunk) {
    const int cus = idx->cu_count;
    const int max_chunks = (int)std::max<int64_t>(1, std::min<int64_t>(max_nchunk, idx->n_rows / 512));
    int best = std::min(max_chunks, std::max(1, (cus + units - 1) / units));
    double best_eff = 0.0;
    for (int c = best; c <= max_chunks; ++c) {
        const int64_t it = (int64_t)units * c;
        const double eff = (double)it / (double)(((it + cus - 1) / cus) * cus);          // fill of the last round
        if (eff > best_eff + 1e-9) { best_eff = eff; best = c; }
        if (eff >= 0.92) break;
    }
    return best;
}

template <int G, int VM>
int launch_scan_g(int mode, const ScanArgs& a, int grid, size_t lds, hipStream_t s) {
    // mode 0: scores, 1: wave top-k, 2: shared top-k
    auto set_lds = [&](const void* f) -> int {
        VS_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        return VS_OK;
    };
    if (mode == 0) {
        VS_TRY(set_lds((const void*)csr_scan_scores<G, VM>));
        hipLaunchKernelGGL((csr_scan_scores<G, VM>), dim3(grid), dim3(kScanThreads), lds, s, a);
    } else if (mode == 1) {
        VS_TRY(set_lds((const void*)csr_scan_topk_wave<G, VM>));
        hipLaunchKernelGGL((csr_scan_topk_wave<G, VM>), dim3(grid), dim3(kScanThreads), lds, s, a);
    } else {
        VS_TRY(set_lds((const void*)csr_scan_topk_shared<G, VM>));
        hipLaunchKernelGGL((csr_scan_topk_shared<G, VM>), dim3(grid), dim3(kScanThreads), lds, s, a);
    }
    VS_HIP(hipGetLastError());
    return VS_OK;
}

template <int VM>
int launch_scan_vm(int g, int mode, const ScanArgs& a, int grid, size_t lds, hipStream_t s) {
    switch (g) {
        case 4: return launch_scan_g<4, VM>(mode, a, grid, lds, s);
        case 8: return launch_scan_g<8, VM>(mode, a, grid, lds, s);
        case 16: return launch_scan_g<16, VM>(mode, a, grid, lds, s);
        case 32: return launch_scan_g<32, VM>(mode, a, grid, lds, s);
        default: return launch_scan_g<64, VM>(mode, a, grid, lds, s);
    }
}

int launch_scan(const vs_index* idx, int mode, const ScanArgs& a, int grid, hipStream_t s) {
    const size_t lds = scan_lds_bytes(idx->n_cols);
    if (lds > 160 * 1024) return fail(VS_EUNSUPPORTED, "n_cols = %d needs %zu B of LDS (> 160 KiB)", idx->n_cols, lds);
    ProfScope prof(mode == 0 ? "csr_scan_scores" : "csr_scan_topk", s);
    if (idx->store_dtype == VS_F32) return launch_scan_vm<VM_F32>(idx->lanes_per_row, mode, a, grid, lds, s);
    if (idx->store_dtype == VS_F16) return launch_scan_vm<VM_F16>(idx->lanes_per_row, mode, a, grid, lds, s);
    return launch_scan_vm<VM_BIN>(idx->lanes_per_row, mode, a, grid, lds, s);
}

int prep_queries(vs_index* idx, const void* q, int q_dtype, int64_t ldq, int B, hipStream_t s, const float** out) {
    if (q_dtype != VS_F32 && q_dtype != VS_F16) return fail(VS_EINVAL, "q_dtype must be VS_F32 or VS_F16");
    const size_t esz = dtype_size(q_dtype);
    const void* dq = q;
    if (!is_device_ptr(q)) {
        // host queries: upload the [B, ldq] block
        const size_t bytes = ((size_t)(B - 1) * ldq + idx->n_cols) * esz;
        VS_TRY(idx->ws_misc.reserve(bytes));
        VS_HIP(hipMemcpyAsync(idx->ws_misc.p, q, bytes, hipMemcpyHostToDevice, s));
        dq = idx->ws_misc.p;
    }
    const int round_f16 = idx->store_dtype == VS_F16;
    if (q_dtype == VS_F32 && !round_f16 && ldq == idx->n_cols) {          // already what the kernels read: contiguous fp32 rows on the device
        *out = (const float*)dq;
        return VS_OK;
    }
    VS_TRY(idx->ws_q.reserve((size_t)B * idx->n_cols * 4));
    const int64_t n = (int64_t)B * idx->n_cols;
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(n, 256), 4096);
    if (q_dtype == VS_F32)
        hipLaunchKernelGGL((prep_queries_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)dq, ldq, B, idx->n_cols, round_f16, idx->ws_q.as<float>());
    else
        hipLaunchKernelGGL((prep_queries_kernel<__half>), dim3(grid), dim3(256), 0, s, (const __half*)dq, ldq, B, idx->n_cols, round_f16, idx->ws_q.as<float>());
    VS_HIP(hipGetLastError());
    *out = idx->ws_q.as<float>();
    return VS_OK;
}

}  // namespace

namespace {

template <int G, int VM, int U, int DN>
int launch_mq_gu(const MqArgs& a, int grid, size_t lds, hipStream_t s) {
    VS_HIP(hipFuncSetAttribute((const void*)csr_scan_topk_mq<G, VM, kQT, U, DN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((csr_scan_topk_mq<G, VM, kQT, U, DN>), dim3(grid), dim3(kScanThreads), lds, s, a);
    VS_HIP(hipGetLastError());
    return VS_OK;
}
template <int G, int VM>
int launch_mq_g(int u, bool shared_cols, const MqArgs& a, int grid, size_t lds, hipStream_t s) {
    if (shared_cols)          // the shared-column variant keeps 16 more registers live: at most 2 packets in flight
        return u <= 1 ? launch_mq_gu<G, VM, 1, 1>(a, grid, lds, s) : launch_mq_gu<G, VM, 2, 1>(a, grid, lds, s);
    if (u <= 1) return launch_mq_gu<G, VM, 1, 0>(a, grid, lds, s);
    if (u == 2) return launch_mq_gu<G, VM, 2, 0>(a, grid, lds, s);
    return launch_mq_gu<G, VM, 3, 0>(a, grid, lds, s);
}
template <int VM>
int launch_mq_vm(int g, int u, bool shared_cols, const MqArgs& a, int grid, size_t lds, hipStream_t s) {
    switch (g) {
        case 8: return launch_mq_g<8, VM>(u, shared_cols, a, grid, lds, s);
        case 16: return launch_mq_g<16, VM>(u, shared_cols, a, grid, lds, s);
        case 32: return launch_mq_g<32, VM>(u, shared_cols, a, grid, lds, s);
        default: return launch_mq_g<64, VM>(u, shared_cols, a, grid, lds, s);
    }
}

constexpr double kMqSharedOverlap = 40.0;   // columns shared by two queries above which the shared-column variant runs

// LDS entries left for the tile's weights once the fixed tables are placed
inline int mq_lanes(const vs_index* idx) { return std::max(idx->lanes_per_row, 8); }
inline int mq_acc_rows(const vs_index* idx) {          // accumulator rows x copies (see S in csr_scan_topk_mq)
    const int g = mq_lanes(idx);
    return kScanWaves * (64 / g) * (g >= 32 ? 4 : (g >= 16 ? 2 : 1));
}
int mq_vals_cap(const vs_index* idx) {
    const size_t fixed = mq_fixed_lds_bytes<kQT>(idx->n_cols, mq_acc_rows(idx));
    const size_t total = 160 * 1024;
    if (fixed + 1024 > total) return 0;
    return (int)((total - fixed) / 4);
}

// ---- blocked postings (bp_walk.h): second, column-grouped copy of the index for sparse queries ---------------------
constexpr int kBpExactQT = 4;     // queries per tile of the fp64 walk (its accumulators are twice as wide as the filter walk's)
constexpr int kBpBinQT = 8;       // queries per tile of the binary index's filter walk

bool bp_wanted(const vs_index* idx) {
    if (idx->bp_pref == 0 || idx->n_rows <= 0 || idx->n_packets <= 0) return false;
    if (((size_t)idx->n_cols + 1) * 4 + 4096 > 160 * 1024) return false;            // the builder keeps one counter per column in LDS
    if (idx->bp_pref == 1) return true;
    // pays off when the index is big enough for the per-block directory (4 (V + 1) bytes per block) to disappear
    if (idx->store_dtype == VS_NONE) return idx->n_rows >= 65536;
    return idx->n_rows >= 16384 && (double)idx->nnz / (double)idx->n_rows >= 256.0;
}

// value mode of the records: the index's own, or fp16 for the lossy filter copy of an fp32 index (bp_refine.h)
inline int bp_record_vm(const vs_index* idx) {
    if (idx->store_dtype == VS_NONE) return VM_BIN;
    return (idx->store_dtype == VS_F16 || idx->bp_quant) ? VM_F16 : VM_F32;
}

void bp_release(vs_index* idx) {
    idx->bp_dir.release(); idx->bp_base.release(); idx->bp_rec.release(); idx->bp_df.release(); idx->bp_vmax.release(); idx->bp_ovf.release();
    idx->bp_hmap.release(); idx->bp_strip.release();
    idx->bp_n_head = 0;
    idx->bp_quad = false;
    idx->bp_ready = false;
}

// valued index: QT queries per tile, blocks of <= 2048 documents (exact fp64 walk: QT = 4, filter walk: QT = 8);
// binary index: filter walk only, one lane per (short) list
constexpr int kFlRoundsF16 = 8, kFlRoundsF32 = 5;     // record loads in flight per lane (registers: 8 / 12 per record)
// which walk serves the fixed-point filter of this index: 0 = a list per lane group (bp_walk.h), 1 = flat worklists (bp_flat.h),
// 2 = the list walk on two accumulator sets, no block barrier (bp_duo.h), 3 = flat worklists, record loads software-pipelined (bp_stream.h)
int bp_walk_kind(const vs_index* idx) {
#ifdef VS_EXPERIMENTAL_WALKS
    const bool can = !idx->bp_quad && idx->store_dtype != VS_NONE && idx->bp_n_head == 0 && idx->bp_max_block_recs < ((int64_t)1 << kFlRecBits) - 4096;
    if (!can) return 0;
    return idx->bp_walk_pref < 0 || idx->bp_walk_pref > 3 ? 0 : idx->bp_walk_pref;
#else
    (void)idx;
    return 0;
#endif
}
template <int AM>
bool bp_flat_ok(const vs_index* idx, const BpArgs& a) { return AM == AM_FIX && !a.upper && bp_walk_kind(idx) >= 1; }
// walk 2 (two accumulator sets, bp_duo.h) serves this call: 4 query slots per tile, K' within its candidate buffers
bool bp_duo_ok(const vs_index* idx, int kp, const uint64_t* upper) { return bp_walk_kind(idx) == 2 && !upper && kp <= kDuoMaxK; }
template <int QT, int AM>
int launch_bp_walk(const vs_index* idx, const BpArgs& a, int grid, int ent_cap, hipStream_t s) {
    const int vm = bp_record_vm(idx);
    size_t lds = bp_lds_bytes<QT, AM, kBpRowsMax>(ent_cap, AM == AM_FIX ? a.n_head : 0);
    void (*kern)(BpArgs) = nullptr;
    if (idx->bp_quad) {
        // quad chunks (bp_quad.h): the fixed-point filter walk only
        if (AM != AM_FIX || a.upper || ent_cap > kBpEntCap) return fail(VS_EUNSUPPORTED, "quad postings serve the filter walk only");
        kern = a.timing ? bp_quad_topk<1> : bp_quad_topk<0>;
        lds = quad_lds_bytes();
#ifdef VS_EXPERIMENTAL_WALKS
    } else if (AM == AM_FIX && bp_duo_ok(idx, a.k, a.upper) && ent_cap <= kDuoEntCap) {
        if (vm == VM_F32) kern = bp_duo_topk<VM_F32, kBpNB, kBpRowsMax>;
        else kern = bp_duo_topk<VM_F16, kBpNBWide, kBpRowsMax>;
        lds = bp_duo_lds_bytes<kBpRowsMax>(ent_cap);
    } else if (bp_flat_ok<AM>(idx, a) && bp_walk_kind(idx) == 3) {
        if (vm == VM_F32) kern = bp_stream_topk<VM_F32, 3, kBpRowsMax>;
        else kern = bp_stream_topk<VM_F16, 4, kBpRowsMax>;
        lds = bp_stream_lds_bytes<kBpRowsMax>(ent_cap);
    } else if (bp_flat_ok<AM>(idx, a) && bp_walk_kind(idx) != 2) {
        // valued records, no dense strips, fixed-point filter: the flat walk (bp_flat.h)
        if (vm == VM_F32) { kern = bp_flat_topk<VM_F32, kFlRoundsF32, kBpRowsMax>; lds = bp_flat_lds_bytes<kFlRoundsF32, kBpRowsMax>(ent_cap); }
        else { kern = bp_flat_topk<VM_F16, kFlRoundsF16, kBpRowsMax>; lds = bp_flat_lds_bytes<kFlRoundsF16, kBpRowsMax>(ent_cap); }
#endif
    } else if (vm == VM_BIN && AM == AM_FIX && idx->bp_walk_pref != 0 && !a.upper && ent_cap <= kBpEntCap) {
        // bag-of-token index: the walk with the next block's records prefetched across the barrier (bp_bin.h); postings_walk = 0: the list walk
        kern = bp_bin_topk<kBpRowsMaxBin>;
        lds = bp_bin_lds_bytes<kBpRowsMaxBin>(ent_cap);
    } else if (vm == VM_BIN) {
        if (AM != AM_FIX) return fail(VS_EUNSUPPORTED, "binary postings serve the filter walk only");
        // one lane per list; the option picks the records in flight per lane = the size of the chunks dealt to the waves (8: 512
        // entries, 13 chunks a block on the Wiki21M shape, 16.4 k q/s; 4: 25 chunks for 16 waves, 13.0 k)
        kern = idx->bp_lanes == 4 ? bp_walk_topk<VM_BIN, kBpBinQT, AM_FIX, 1, kBpRowsMaxBin, 4> : bp_walk_topk<VM_BIN, kBpBinQT, AM_FIX, 1, kBpRowsMaxBin, 8>;
        lds = bp_lds_bytes<kBpBinQT, AM_FIX, kBpRowsMaxBin>(ent_cap);
    } else if (AM == AM_FIX && a.n_head > 0) {
        if constexpr (AM == AM_FIX) {
            if (vm == VM_F32) kern = idx->bp_lanes != 4 ? bp_walk_topk<VM_F32, QT, AM_FIX, 8, kBpRowsMax, kBpNB, 1> : bp_walk_topk<VM_F32, QT, AM_FIX, 4, kBpRowsMax, kBpNB, 1>;
            else kern = idx->bp_lanes != 4 ? bp_walk_topk<VM_F16, QT, AM_FIX, 8, kBpRowsMax, kBpNB, 1> : bp_walk_topk<VM_F16, QT, AM_FIX, 4, kBpRowsMax, kBpNB, 1>;
        }
    } else if (vm == VM_F32) {
        kern = idx->bp_lanes != 4 ? bp_walk_topk<VM_F32, QT, AM, 8, kBpRowsMax> : bp_walk_topk<VM_F32, QT, AM, 4, kBpRowsMax>;
    } else {
        kern = idx->bp_lanes != 4 ? bp_walk_topk<VM_F16, QT, AM, 8, kBpRowsMax, (AM == AM_FIX ? kBpNBWide : kBpNB)> : bp_walk_topk<VM_F16, QT, AM, 4, kBpRowsMax>;
    }
    if (lds > 160 * 1024) return fail(VS_EUNSUPPORTED, "postings walk needs %zu B of LDS", lds);
    VS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kScanThreads), lds, s, a);
    VS_HIP(hipGetLastError());
    return VS_OK;
}

int bp_build(vs_index* idx, hipStream_t s) {
    idx->bp_tried = true;
    bp_release(idx);
    // Documents per block.  Valued index: as many as give an average list ~50 postings (768-nnz documents, V = 29 523: 1920) --
    // a list is read by 8 lanes x 8 postings per round, and at 53 postings a list (2048 documents) 1 list in 15 needs a second
    // record per lane, at 50 (1920) 1 in 40: 152.2 vs 158.7 ms at 21 M docs (1792: 156.1) -- capped by what the accumulators
    // hold (2048) and kept a multiple of 128 (dense strips).  Binary index: 2048.
    // Quad chunks (bp_quad.h) -- the default copy of a valued index searched by filter + refine: a list is cut into 64-cell chunks of
    // one-dword postings (fp16 values).  Not for: a binary index; exact fp32 records ("postings_quant" = 0, signed / huge values);
    // the fp64 walk ("postings_filter" = 0); a corpus with head columns (their dense strips belong to the list walk: bp_build starts
    // over without quad when it finds any); the experimental walks 0 .. 3 ("postings_walk"), aligned or arranged records.
    const bool quad_pref = idx->store_dtype != VS_NONE && idx->bp_filter != 0 && idx->n_cols <= 32768 && (idx->bp_walk_pref == -1 || idx->bp_walk_pref == 4) && !idx->bp_no_quad &&
                           idx->bp_align_pref != 1 && idx->bp_arrange_pref != 1 && (idx->store_dtype == VS_F16 || idx->bp_quant_pref != 0);
    idx->bp_no_quad = false;
    auto auto_rows = [&]() -> int {
        if (idx->store_dtype == VS_NONE) return kBpRowsMaxBin;
        const double avg = idx->n_rows > 0 ? (double)idx->nnz / (double)idx->n_rows : 1.0;
        // (quad chunks: at 50 postings a list 1 list in 40 goes on in an overflow chunk; 2048 documents per block measured the same)
        const int r = (int)(50.0 * (double)idx->n_cols / std::max(avg, 1.0)) / 128 * 128;
        return std::max(256, std::min(r, kBpRowsMax));
    };
    idx->bp_rows = idx->bp_rows_pref > 0 ? std::min(idx->bp_rows_pref, idx->store_dtype == VS_NONE ? kBpRowsMaxBin : kBpRowsMax)
                   : idx->bp_rows_forced > 0 ? idx->bp_rows_forced : auto_rows();
    idx->bp_rows_forced = 0;
    const int64_t n_blocks = ceil_div64(idx->n_rows, idx->bp_rows);
    const int V = idx->n_cols;
    const int RS0 = bp_rec_bytes(idx->store_dtype == VS_F32 ? VM_F16 : idx->store_dtype == VS_F16 ? VM_F16 : VM_BIN);   // smallest record this index can get
    const size_t b_dir = (size_t)n_blocks * ((size_t)V + 1) * 4;
    size_t free_b = 0, total_b = 0;
    VS_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t margin = idx->bp_pref == 1 ? ((size_t)256 << 20) : ((size_t)4 << 30);  // leave room for scratch / other tensors
    auto no_room = [&](size_t need) {
        fprintf(stderr, "[vsearch_hip] blocked-postings copy not built: needs %.1f GB, %.1f GB of HBM free -- sparse queries use the CSR scan (3x slower)\n",
                (double)need / 1e9, (double)free_b / 1e9);
        bp_release(idx);
        idx->bp_state = 2;
        (void)hipGetLastError();
        return VS_OK;
    };
    // lower bound of the records: one per 8 non-zeros
    if (free_b < b_dir + (size_t)idx->n_packets * RS0 + margin) return no_room(b_dir + (size_t)idx->n_packets * RS0);
    DevBuf block_recs;
    if (idx->bp_dir.alloc(b_dir) != VS_OK || idx->bp_base.alloc((size_t)(n_blocks + 1) * 8) != VS_OK || idx->bp_df.alloc((size_t)V * 16) != VS_OK ||
        block_recs.alloc((size_t)n_blocks * 4) != VS_OK)
        return no_room(b_dir);
    VS_HIP(hipMemsetAsync(idx->bp_df.p, 0, (size_t)V * 16, s));
    unsigned long long* df_rec = idx->bp_df.as<unsigned long long>();
    unsigned long long* df_nnz = df_rec + V;
    const size_t lds = ((size_t)V + 1) * 4;
    const int grid = (int)std::min<int64_t>(n_blocks, (int64_t)idx->cu_count * 8);
    ProfScope prof("bp_build", s);
    // max |value| (bounds the products of the fixed-point walk) and "any value negative"; a binary index has no values
    uint32_t hv[2] = {0x3F800000u, 0u};
    if (idx->store_dtype != VS_NONE) {
        VS_TRY(idx->bp_vmax.alloc(8));
        VS_HIP(hipMemsetAsync(idx->bp_vmax.p, 0, 8, s));
        const int64_t nv = idx->n_packets * 8;
        const unsigned g = (unsigned)std::min<int64_t>(ceil_div64(nv, 256 * 16), (int64_t)idx->cu_count * 16);
        if (idx->store_dtype == VS_F32) hipLaunchKernelGGL(bp_vmax_kernel<VM_F32>, dim3(g), dim3(256), 0, s, (const void*)idx->vals.p, nv, idx->bp_vmax.as<uint32_t>());
        else hipLaunchKernelGGL(bp_vmax_kernel<VM_F16>, dim3(g), dim3(256), 0, s, (const void*)idx->vals.p, nv, idx->bp_vmax.as<uint32_t>());
        VS_HIP(hipGetLastError());
        VS_HIP(hipMemcpyAsync(hv, idx->bp_vmax.p, 8, hipMemcpyDeviceToHost, s));
    }
    VS_HIP(hipStreamSynchronize(s));
    float vmax_f;
    memcpy(&vmax_f, &hv[0], 4);
    const bool lossy_ok = hv[1] == 0u && vmax_f < 60000.f;       // fp16 copies of the values: non-negative, no overflow
    const bool quad = quad_pref && lossy_ok;                        // (non-negative values: a set sign bit marks a link)
    // Lossy filter copy of an fp32 index: values rounded to fp16 (4 instead of 6 bytes per posting); needs the filter-and-refine
    // search, non-negative values and no fp16 overflow
    idx->bp_quant = idx->store_dtype == VS_F32 && idx->bp_filter != 0 && idx->bp_quant_pref != 0 && lossy_ok;
    // option "postings_align" = 1: lists start on whole 128-byte lines (+16 % bytes for -1.5 % walk time at 8 lanes per list: off by default)
    idx->bp_al_shift = idx->bp_align_pref == 1 ? bp_align_shift(bp_record_vm(idx)) : 0;
    DevBuf ovf;
    VS_TRY(ovf.alloc(4));
    VS_HIP(hipMemsetAsync(ovf.p, 0, 4, s));
    if (quad) {
        // quad chunks: main chunk of column c = chunk c of its block, overflow chunks behind (the directory is the builder's only)
        const size_t b_main = (size_t)n_blocks * V * kQuadChunkBytes;
        if (free_b < b_dir + b_main + margin) return no_room(b_dir + b_main);
        VS_HIP(hipFuncSetAttribute((const void*)quad_count_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(quad_count_kernel<0>, dim3(grid), dim3(kScanThreads), lds, s, idx->pk_ptr.as<uint32_t>(), idx->cols.as<uint4>(), idx->n_rows, V, idx->bp_rows,
                           idx->bp_dir.as<uint32_t>(), block_recs.as<uint32_t>(), df_rec, df_nnz, ovf.as<int32_t>());
    } else {
        VS_HIP(hipFuncSetAttribute((const void*)bp_count_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(bp_count_kernel<0>, dim3(grid), dim3(kScanThreads), lds, s, idx->pk_ptr.as<uint32_t>(), idx->cols.as<uint4>(), idx->n_rows, V, idx->bp_rows,
                           idx->bp_dir.as<uint32_t>(), block_recs.as<uint32_t>(), df_rec, df_nnz, (const uint16_t*)nullptr, idx->bp_al_shift, ovf.as<int32_t>(), 3);
    }
    VS_STAGE("bp_count", s);
    // Head columns (skewed vocabularies): present in >= 1/4 of the documents -> dense strips instead of posting lists.  Valued
    // indexes with the filter search only (the strips hold fp16 values: the fp64 walk cannot use them).
    idx->bp_vmax_f = vmax_f;
    static const int head_env = getenv("VS_BP_HEAD") ? atoi(getenv("VS_BP_HEAD")) : -2;           // (developer override of "postings_head")
    if (head_env > -2) idx->bp_head_pref = head_env;
    if (idx->store_dtype != VS_NONE && idx->bp_filter != 0 && idx->bp_head_pref != 0 && idx->n_rows >= 4096 && lossy_ok && idx->bp_rows % 128 == 0 &&
        vmax_f >= 1.f / 64.f) {
        DevBuf nh;
        VS_TRY(nh.alloc(4));
        VS_TRY(idx->bp_hmap.alloc((size_t)V * 2));
        hipLaunchKernelGGL(bp_head_select_kernel<0>, dim3(1), dim3(kScanThreads), 0, s, df_nnz, V, (unsigned long long)ceil_div64(idx->n_rows, idx->bp_head_pref > 0 ? idx->bp_head_pref : 4), kBpHeadCap,
                           idx->bp_hmap.as<uint16_t>(), nh.as<int32_t>());
        VS_HIP(hipGetLastError());
        int32_t h_n = 0;
        VS_HIP(hipMemcpyAsync(&h_n, nh.p, 4, hipMemcpyDeviceToHost, s));
        VS_HIP(hipStreamSynchronize(s));
        idx->bp_n_head = h_n;
        if (h_n > 0 && quad) {                                  // head columns: records + dense strips (the list walk)
            idx->bp_no_quad = true;
            return bp_build(idx, s);
        }
        if (h_n > 0 && idx->bp_rows_pref <= 0 && idx->bp_rows < kBpRowsMax) {
            // a skewed corpus: its lists are long whatever the block size, and the dense strips and the per-block costs want the
            // largest blocks (zipf 21 M docs: 289 ms at 2048 documents per block, 301 at 1920) -- start over with those
            idx->bp_rows_forced = kBpRowsMax;
            idx->bp_no_quad = true;                               // (head columns: the restart must not try quad chunks again)
            return bp_build(idx, s);
        }
        if (h_n > 0) {
            // the directory again, without the head columns' lists
            VS_HIP(hipMemsetAsync(idx->bp_df.p, 0, (size_t)V * 16, s));
            hipLaunchKernelGGL(bp_count_kernel<0>, dim3(grid), dim3(kScanThreads), lds, s, idx->pk_ptr.as<uint32_t>(), idx->cols.as<uint4>(), idx->n_rows, V,
                               idx->bp_rows, idx->bp_dir.as<uint32_t>(), block_recs.as<uint32_t>(), df_rec, df_nnz, (const uint16_t*)idx->bp_hmap.as<uint16_t>(),
                               idx->bp_al_shift, ovf.as<int32_t>(), 3);
            VS_HIP(hipGetLastError());
            const size_t b_strip = (size_t)n_blocks * bp_head_pad(h_n) * idx->bp_rows * 2;
            VS_HIP(hipMemGetInfo(&free_b, &total_b));
            if (free_b < b_strip + margin || idx->bp_strip.alloc(b_strip) != VS_OK) return no_room(b_strip);
            VS_HIP(hipMemsetAsync(idx->bp_strip.p, 0, b_strip, s));
        } else {
            idx->bp_hmap.release();
        }
    }
    hipLaunchKernelGGL(bp_base_kernel<0>, dim3(1), dim3(kScanThreads), 0, s, block_recs.as<uint32_t>(), n_blocks, idx->bp_base.as<unsigned long long>());
    VS_HIP(hipGetLastError());
    VS_STAGE("bp_base", s);
    unsigned long long n_rec = 0;
    int32_t h_ovf = 0;
    VS_HIP(hipMemcpyAsync(&n_rec, idx->bp_base.as<unsigned long long>() + n_blocks, 8, hipMemcpyDeviceToHost, s));
    VS_HIP(hipMemcpyAsync(&h_ovf, ovf.p, 4, hipMemcpyDeviceToHost, s));
    std::vector<uint32_t> h_brecs((size_t)n_blocks);
    VS_HIP(hipMemcpyAsync(h_brecs.data(), block_recs.p, (size_t)n_blocks * 4, hipMemcpyDeviceToHost, s));
    VS_HIP(hipStreamSynchronize(s));
    idx->bp_max_block_recs = 0;
    for (uint32_t r : h_brecs) idx->bp_max_block_recs = std::max<int64_t>(idx->bp_max_block_recs, (int64_t)r);
    if (h_ovf) {                                                        // (2048 documents x 29 523 columns, all present, would do it)
        fprintf(stderr, "[vsearch_hip] blocked-postings copy not built: a block holds more records than a directory word addresses -- sparse queries use the CSR scan\n");
        bp_release(idx);
        idx->bp_state = 3;
        return VS_OK;
    }
    VS_STAGE("bp_vmax", s);
    const int RS = quad ? kQuadChunkBytes : bp_rec_bytes(bp_record_vm(idx));
    const size_t b_rec = ((size_t)n_rec + 2) * RS;                       // + one record: a lane past the last list's end re-reads "the record at the end"
    VS_HIP(hipMemGetInfo(&free_b, &total_b));
    if (free_b < b_rec + margin || idx->bp_rec.alloc(b_rec) != VS_OK) return no_room(b_rec);
    idx->bp_records = (int64_t)n_rec;
    VS_HIP(hipMemsetAsync(idx->bp_rec.p, 0, b_rec, s));                  // pad postings: document 0, value 0
    if (quad) {
        void (*fill)(const uint32_t*, const uint4*, const void*, int64_t, int32_t, int32_t, const uint32_t*, const unsigned long long*, uint32_t*) =
            idx->store_dtype == VS_F32 ? quad_fill_kernel<VM_F32> : quad_fill_kernel<VM_F16>;
        VS_HIP(hipFuncSetAttribute((const void*)fill, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(fill, dim3(grid), dim3(kScanThreads), lds, s, idx->pk_ptr.as<uint32_t>(), idx->cols.as<uint4>(), (const void*)idx->vals.p, idx->n_rows, V,
                           idx->bp_rows, idx->bp_dir.as<uint32_t>(), idx->bp_base.as<unsigned long long>(), idx->bp_rec.as<uint32_t>());
        VS_HIP(hipGetLastError());
        VS_STAGE("quad_fill", s);
        const size_t alds = (size_t)2 * 256 * (kQuadCells + 1) * 4;
        VS_HIP(hipFuncSetAttribute((const void*)quad_arrange_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)alds));
        hipLaunchKernelGGL(quad_arrange_kernel<0>, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div64((int64_t)n_rec, 256), (int64_t)idx->cu_count * 8))), dim3(256), alds, s,
                           idx->bp_rec.as<uint32_t>(), n_rec);
        VS_HIP(hipGetLastError());
        VS_STAGE("quad_arrange", s);
        idx->bp_quad = true;
        VS_HIP(hipStreamSynchronize(s));
        idx->bp_dir.release();                                           // (the chunks link to their overflow themselves)
        idx->bp_quant = idx->store_dtype == VS_F32;                      // fp16-rounded values of an fp32 index: the refine step's bound accounts for them
    } else {
        void (*fill)(const uint32_t*, const uint4*, const void*, int64_t, int32_t, int32_t, const uint32_t*, const unsigned long long*, char*, const uint16_t*, __half*, int32_t, int32_t) =
            idx->store_dtype == VS_F32 ? (idx->bp_quant ? bp_fill_kernel<VM_F32, VM_F16> : bp_fill_kernel<VM_F32, VM_F32>)
            : idx->store_dtype == VS_F16 ? bp_fill_kernel<VM_F16, VM_F16> : bp_fill_kernel<VM_BIN, VM_BIN>;
        VS_HIP(hipFuncSetAttribute((const void*)fill, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(fill, dim3(grid), dim3(kScanThreads), lds, s, idx->pk_ptr.as<uint32_t>(), idx->cols.as<uint4>(), (const void*)idx->vals.p, idx->n_rows, V,
                           idx->bp_rows, idx->bp_dir.as<uint32_t>(), idx->bp_base.as<unsigned long long>(), idx->bp_rec.as<char>(),
                           idx->bp_n_head > 0 ? (const uint16_t*)idx->bp_hmap.as<uint16_t>() : (const uint16_t*)nullptr, idx->bp_strip.as<__half>(), idx->bp_n_head,
                           idx->bp_al_shift);
    }
    VS_HIP(hipGetLastError());
    VS_STAGE("bp_fill", s);
    // bank-aware order inside the lists (option "postings_arrange" = 1; off by default: 4 M docs, list walk 29.83 -> 29.58 ms for 55 ms more
    // build time -- the scatter-add's bank conflicts are not what the walk waits for)
    if (!quad && idx->store_dtype != VS_NONE && idx->bp_arrange_pref == 1) {
        void (*arr)(const uint32_t*, const unsigned long long*, char*, int64_t, int32_t, int32_t) =
            bp_record_vm(idx) == VM_F32 ? bp_arrange_kernel<VM_F32> : bp_arrange_kernel<VM_F16>;
        const size_t alds = (size_t)256 * 64 * (2 + 4);
        VS_HIP(hipFuncSetAttribute((const void*)arr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)alds));
        hipLaunchKernelGGL(arr, dim3((unsigned)std::min<int64_t>(n_blocks, (int64_t)idx->cu_count * 8)), dim3(256), alds, s, idx->bp_dir.as<uint32_t>(),
                           idx->bp_base.as<unsigned long long>(), idx->bp_rec.as<char>(), n_blocks, V, idx->bp_al_shift);
        VS_HIP(hipGetLastError());
        VS_STAGE("bp_arrange", s);
    }
    VS_HIP(hipStreamSynchronize(s));                                     // `block_recs` is freed on return
    if (debug_sync_on()) {
        std::vector<unsigned long long> hb((size_t)n_blocks + 1);
        std::vector<uint32_t> hd((size_t)V + 1);
        (void)hipMemcpy(hb.data(), idx->bp_base.p, hb.size() * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hd.data(), idx->bp_dir.as<uint32_t>() + (size_t)(n_blocks - 1) * (V + 1), hd.size() * 4, hipMemcpyDeviceToHost);
        bool mono = true;
        for (size_t i = 0; i + 1 < hb.size(); ++i) mono = mono && hb[i] <= hb[i + 1];
        bool dmono = true;
        for (size_t i = 0; i + 2 < hd.size(); ++i) dmono = dmono && (hd[i] >> 12) <= (hd[i + 1] >> 12);
        fprintf(stderr, "[vsearch_hip] bp: rows %d blocks %lld records %llu base[1] %llu base[last] %llu monotone %d; last block dir end %u (block holds %llu) monotone %d\n",
                idx->bp_rows, (long long)n_blocks, n_rec, hb.size() > 1 ? hb[1] : 0ull, hb[n_blocks], (int)mono, (hd[V - 1] >> 12) << idx->bp_al_shift, hb[n_blocks] - hb[n_blocks - 1], (int)dmono);
    }
    idx->bp_ready = true;
    idx->bp_state = 1;
    return VS_OK;
}

// Builds the blocked-postings copy NOW when this index would get one at its first sparse search (vs_index_prepare: the 0.5 s of a
// 21 M-doc build then belong to load / move_to_device, not to a user's first retrieve).  Idempotent.
int csr_prepare_impl(vs_index* idx, hipStream_t s) {
    if (idx->kind != VS_KIND_CSR || idx->qt_pref == 1) return VS_OK;
    if (!bp_wanted(idx)) { if (!idx->bp_ready) idx->bp_state = 4; return VS_OK; }
    if (!idx->bp_ready && !idx->bp_tried) VS_TRY(bp_build(idx, s));
    return VS_OK;
}

// Error bound, in units of the fixed-point sums, of the dense (matrix-core) part of a filter score against the real sum of
// weight * scale * fp16 strip value over the head columns (all terms >= 0, the sum < 2^30):
//   weights split in two fp16 numbers: hi + lo misses <= 2^-22 of each weight                            -> 2^8
//   lo below the fp16 normal range (taken as flushed to zero): <= 2^-14 of an operand unit, x 2^16       -> 4 per column
//   strip values below the fp16 normal range (taken as flushed): value < 2^-14, weights sum < 2^30 / max -> 2^16 / max value
//   fp32 accumulation: 33 additions per k-step of 32 columns, each off by <= one ulp of a sum < 2^30     -> 33 * 128 per k-step
//   the lo column's own sums are 2^-11 of that; two truncations                                          -> 64 + 2
int32_t bp_head_slack(const vs_index* idx) {
    if (idx->bp_n_head <= 0) return 0;
    const int hp = bp_head_pad(idx->bp_n_head);
    return 256 + 4 * hp + (int32_t)ceilf(65536.f / idx->bp_vmax_f) + 33 * 128 * (hp / 32) + 66;
}

// chunks of the postings walk for `n_tiles` query tiles (see bp_filter_search)
int bp_choose_chunks(const vs_index* idx, int n_tiles, int64_t n_blocks, int plan_nchunk) {
    int nchunk = (int)std::min<int64_t>(choose_chunks(idx, n_tiles, plan_nchunk), n_blocks);
    // Big index, enough tiles: as FEW chunks as give every CU ONE work item.  Every tile sweeps its chunk's blocks in the same
    // order at the same pace, so with few chunks all tiles are within a few blocks of each other and each block is fetched from
    // HBM once for all of them (Infinity Cache); and every item pays its start-up (entry sort, the candidate flood until its
    // threshold rises) once.  21 M docs, 1024 queries, walk time: 2 chunks 159.0 ms (5 of 6 fresh processes; 167.3 in the sixth),
    // 4 chunks 169.4, 3 chunks (384 items on 256 CUs) 328.  (Before the waves took their chunks of a block dynamically, one item
    // per CU was unstable: 203 .. 233 ms against 207 at two per CU.)  Binary index: 2 chunks 63.5 ms, 4: 64.9, 8: 68.1.
    if (idx->n_rows >= (2 << 20) && (int64_t)n_tiles * 4 >= idx->cu_count) {
        int best = 1;
        double best_eff = 0.0;
        // (a skewed corpus -- one with head columns -- keeps two items per CU: its tiles differ in weight, and with one item
        //  each the heaviest tile's CU finishes alone: zipf 21 M docs 441 ms against 289)
        const int per_cu = idx->bp_n_head > 0 ? 2 : 1;
        const int c0 = (int)std::max<int64_t>(1, ceil_div64(per_cu * (int64_t)idx->cu_count, n_tiles));
        for (int c = c0; c <= c0 + 3; ++c) {
            const int64_t it = (int64_t)n_tiles * c;
            const double eff = (double)it / (double)(ceil_div64(it, idx->cu_count) * idx->cu_count);
            if (eff > best_eff + 1e-9) { best_eff = eff; best = c; }
            if (eff >= 0.95) break;
        }
        nchunk = (int)std::min<int64_t>(best, n_blocks);
    }
    if (idx->bp_chunks > 0) nchunk = (int)std::min<int64_t>(idx->bp_chunks, n_blocks);
    static const int chunks_env = getenv("VS_BP_CHUNKS") ? atoi(getenv("VS_BP_CHUNKS")) : 0;      // (developer override)
    if (chunks_env > 0) nchunk = (int)std::min<int64_t>(chunks_env, n_blocks);
    return std::max(1, nchunk);
}

// the filter-and-refine search takes this call: one pass, k + margin within the candidate buffers
bool bp_filter_ok(const vs_index* idx, int k, int col0, const uint64_t* upper) {
    return idx->bp_ready && idx->bp_filter != 0 && col0 == 0 && !upper && k + std::max(28, k / 4) <= kBpMaxK &&
           (idx->store_dtype == VS_NONE || idx->bp_vmax.p) && mq_vals_cap(idx) > 0;
}

// Filter and refine (bp_walk.h, bp_refine.h), WITHOUT a host synchronisation: the query tiles are planned on the device and the
// kernels read the tile count there; scratch is sized from upper bounds (B queries x the entry capacity of a tile); queries too
// dense for a tile, and queries whose top k the refine step cannot prove, are collected on the device and take the exact
// one-query scan, launched unconditionally (it returns at once when the list is empty).
int bp_filter_search(vs_index* idx, const float* dq, int32_t B, int32_t k, int64_t id_offset, int64_t* d_ids, float* d_scores, const ScanPlan& plan,
                     hipStream_t s, bool* done, int32_t out_ld) {
    const int V = idx->n_cols;
    const int kp = k + std::max(28, k / 4);
    const bool duo = bp_duo_ok(idx, kp, nullptr);
    static const int qt_env = getenv("VS_BP_QT") ? atoi(getenv("VS_BP_QT")) : 0;                       // (developer: smaller tiles on the 8-slot walk)
    const int qt = qt_env > 0 ? std::min(qt_env, kQT) : idx->store_dtype == VS_NONE ? kBpBinQT : (duo ? kDuoQT : kQT);
    // (dense strips: their weight matrix takes 16 KB of the LDS the entries would use)
    const int vals_cap = std::min(mq_vals_cap(idx), duo ? kDuoEntCap : (idx->bp_n_head > 0 ? kBpEntCap - 512 : kBpEntCap));
    const int64_t qcap = (int64_t)B * vals_cap;                               // bound of the batch's (query, column) entries that enter a tile
    const int64_t n_blocks = ceil_div64(idx->n_rows, idx->bp_rows);
    if (idx->bp_rows > (idx->store_dtype == VS_NONE ? kBpRowsMaxBin : kBpRowsMax)) return fail(VS_EINVAL, "postings_rows beyond the walk's block capacity");
    // scratch: per-query metadata (counts, qptr, plan, tiles, flags, scale, slack, weight sums), column frequencies, the sparse batch
    const size_t off_counts = 0, off_qptr = off_counts + (size_t)B * 8, off_plan = off_qptr + (size_t)(B + 1) * 8, off_tiles = off_plan + 64,
                 off_fb = off_tiles + (size_t)B * 8, off_scale = off_fb + (size_t)B * 8, off_slack = off_scale + (size_t)B * 4,
                 off_wsum = off_slack + (size_t)B * 4, off_flags = off_wsum + (size_t)B * 4, off_nfb = off_flags + (size_t)B * 4,
                 off_gtau = (off_nfb + 64 + 15) & ~(size_t)15, off_freq = (off_gtau + (size_t)B * 8 + 15) & ~(size_t)15;
    VS_TRY(idx->ws_mq_meta.reserve(off_freq + (size_t)(V + 4) * 4 + 8));
    char* meta = idx->ws_mq_meta.as<char>();
    unsigned long long* gtau = (unsigned long long*)(meta + off_gtau);
    int64_t* counts = (int64_t*)(meta + off_counts);
    int64_t* qptr = (int64_t*)(meta + off_qptr);
    int64_t* dplan = (int64_t*)(meta + off_plan);
    int2* tiles = (int2*)(meta + off_tiles);
    int2* fb_tiles = (int2*)(meta + off_fb);
    float* qscale = (float*)(meta + off_scale);
    int32_t* qslack = (int32_t*)(meta + off_slack);
    float* qwsum = (float*)(meta + off_wsum);
    uint32_t* flags = (uint32_t*)(meta + off_flags);
    int32_t* fb_n = (int32_t*)(meta + off_nfb);
    uint32_t* colfreq = (uint32_t*)(meta + off_freq);
    VS_TRY(idx->ws_mq_q.reserve(std::max<size_t>((size_t)qcap * 8, 16)));
    int32_t* qcols = idx->ws_mq_q.as<int32_t>();
    float* qvals = reinterpret_cast<float*>(qcols + qcap);
    // 1. sparsify the batch and plan the tiles, all on the device
    VS_HIP(hipMemsetAsync(gtau, 0, (size_t)(off_freq - off_gtau) + (size_t)(V + 4) * 4 + 8, s));      // thresholds + column frequencies (adjacent)
    hipLaunchKernelGGL(bp_count_colfreq_kernel<0>, dim3(std::min(B, 2048)), dim3(256), 0, s, dq, (int64_t)V, B, V, counts, colfreq);
    if (B <= kPlanFast) hipLaunchKernelGGL(bp_plan_fast_kernel<0>, dim3(1), dim3(256), 0, s, counts, B, qt, vals_cap, qptr, tiles, dplan, flags);
    else hipLaunchKernelGGL(bp_plan_kernel<0>, dim3(1), dim3(64), 0, s, counts, B, qt, vals_cap, qptr, tiles, dplan, flags);
    hipLaunchKernelGGL(fill_csr_kernel<0>, dim3(std::min(B, 2048)), dim3(kSpThreads), 0, s, dq, (int64_t)V, B, V, qptr, qcols, qvals, qcap);
    if (idx->bp_df.p)
        hipLaunchKernelGGL(bp_walk_kernel<0>, dim3(1), dim3(kScanThreads), 0, s, colfreq, idx->bp_df.as<unsigned long long>(),
                           idx->bp_df.as<unsigned long long>() + V, V, dplan + 4);
    hipLaunchKernelGGL(bp_qscale_kernel<0>, dim3((unsigned)ceil_div(B, 4)), dim3(256), 0, s, qptr, qvals, B, idx->bp_vmax.as<uint32_t>(),
                       idx->store_dtype == VS_NONE ? 1 : 0, (idx->bp_quant || idx->bp_n_head > 0) ? 1 : 0, qscale, qslack, qwsum, (const int32_t*)qcols,
                       idx->bp_n_head > 0 ? (const uint16_t*)idx->bp_hmap.as<uint16_t>() : (const uint16_t*)nullptr, bp_head_slack(idx));
    VS_HIP(hipGetLastError());
    VS_STAGE("sparsify", s);
    // 2. the walk.  Work items = (tile, chunk); the tile count lives on the device, the chunks follow its lower bound ceil(B / qt)
    const int n_tiles_est = ceil_div(B, qt);
    const int nchunk = bp_choose_chunks(idx, n_tiles_est, n_blocks, plan.nchunk);
    const int nchunk_fb = (int)std::min<int64_t>(n_blocks, 64);
    const int grid = (int)std::min<int64_t>((int64_t)B * nchunk, idx->cu_count);
    VS_TRY(idx->ws_mq_cand.reserve((size_t)idx->cu_count * kQT * std::max(kBpCap, kFlCap) * 8));
    VS_TRY(idx->ws_cand.reserve(std::max((size_t)B * nchunk * kp, (size_t)B * nchunk_fb * k) * 8));
    BpArgs a{};
    a.rows = idx->bp_rows;
    a.dir = idx->bp_dir.as<uint32_t>();
    a.al_shift = idx->bp_al_shift;
    a.base = idx->bp_base.as<unsigned long long>();
    a.rec = idx->bp_rec.as<char>();
    a.n_rows = idx->n_rows;
    a.n_cols = V;
    a.k = kp;
    a.nchunk = nchunk;
    a.blocks_per_chunk = ceil_div64(n_blocks, nchunk);
    a.qptr = qptr;
    a.qcols = qcols;
    a.qvals = qvals;
    a.tiles = tiles;
    a.n_tiles = 0;
    a.n_tiles_dev = reinterpret_cast<const int32_t*>(dplan);                  // plan[0] (little endian: the low word of the int64)
    a.ent_cap = vals_cap;
    a.cand = idx->ws_cand.as<uint64_t>();
    a.gcand = idx->ws_mq_cand.as<uint64_t>();
    a.qscale = qscale;
    a.gtau = gtau;
    a.df = idx->bp_df.p ? idx->bp_df.as<unsigned long long>() + V : nullptr;          // (second half of bp_df: non-zeros per column)
    a.hmap = idx->bp_n_head > 0 ? idx->bp_hmap.as<uint16_t>() : nullptr;
    a.strip = idx->bp_strip.as<__half>();
    a.n_head = idx->bp_n_head;
    if (a.n_head > 0) {
        int ve;
        (void)frexpf(idx->bp_vmax_f, &ve);                          // max value < 2^ve
        a.head_pre = ldexpf(1.f, ve - 16);                          // weight * scale < 2^30 / max value  ->  * 2^ve / 2^16 < 2^15: an fp16 number
        a.head_mul = ldexpf(1.f, 16 - ve);
    }
    idx->last_path = 3;
    idx->last_plan_dev = dplan;
    idx->last_plan_rs = idx->bp_quad ? kQuadChunkBytes : bp_rec_bytes(bp_record_vm(idx));
    idx->last_plan_blocks = n_blocks;
    // lock-step window of the walk's work items (all walks; kernels ignore it when not every item is resident)
    static const int pace_env = getenv("VS_BP_PACE") ? atoi(getenv("VS_BP_PACE")) : -1;
    // (off by default for the list walk: it costs it 10 %; the bag-of-token walk of bp_bin.h runs ahead of its memory and
    //  NEEDS it: free running 81 ms, window 16: 66.7, 32: 56.5, 48: 57.2, 64: 58.7, 128: 73 -- the list walk: 61.3)
    const bool bin_walk = idx->store_dtype == VS_NONE && idx->bp_walk_pref != 0;
    // (the two-set walk has no block barrier to keep its workgroups at one pace: 4 M docs 56.9 ms free running, window 1: 39.8, 2: 37.9, 4: 42.2)
    const int pace_w = pace_env >= 0 ? pace_env : (idx->bp_pace >= 0 ? idx->bp_pace : (bin_walk ? 32 : (duo ? 2 : (idx->bp_quad ? kQuadPaceDefault : 0))));
    if (pace_w > 0) {
        VS_TRY(idx->ws_pace.reserve((size_t)nchunk * a.blocks_per_chunk * 4));
        VS_HIP(hipMemsetAsync(idx->ws_pace.p, 0, (size_t)nchunk * a.blocks_per_chunk * 4, s));
        a.pace = idx->ws_pace.as<uint32_t>();
        a.pace_window = pace_w;
    }
    static const int knob_env = getenv("VS_BP_KNOB") ? atoi(getenv("VS_BP_KNOB")) : 0;
    a.knob = knob_env;
    static const bool debug_on = getenv("VS_BP_DEBUG") != nullptr;
    DevBuf dbg;
    if (debug_on) {
        VS_TRY(dbg.alloc(64));
        VS_HIP(hipMemsetAsync(dbg.p, 0, 64, s));
        a.debug = dbg.as<unsigned long long>();
    }
    static const bool timing_on = getenv("VS_BP_TIMING") != nullptr;            // developer aid: where the walk's wave-cycles go
    DevBuf timing;
    if (timing_on) {
        VS_TRY(timing.alloc(128 + (size_t)grid * 32));
        VS_HIP(hipMemsetAsync(timing.p, 0, 128 + (size_t)grid * 32, s));
        a.timing = timing.as<unsigned long long>();
    }
    {
        ProfScope prof("csr_scan_topk", s);
        VS_TRY((launch_bp_walk<kQT, AM_FIX>(idx, a, grid, vals_cap, s)));
    }
    if (debug_on) {
        unsigned long long d[8] = {0};
        VS_HIP(hipMemcpyAsync(d, dbg.p, 64, hipMemcpyDeviceToHost, s));
        VS_HIP(hipStreamSynchronize(s));
        fprintf(stderr, "[vsearch_hip] stream walk debug: %llu bad items (e.g. item %08llx, block of %llu records, block %llu, batch of %llu)\n", d[0], d[1] >> 32, d[1] & 0xFFFFFFFFull,
                d[2] >> 32, d[2] & 0xFFFFFFFFull);
    }
    if (timing_on) {
        unsigned long long h[16] = {0};
        VS_HIP(hipMemcpyAsync(h, timing.p, 128, hipMemcpyDeviceToHost, s));
        VS_HIP(hipStreamSynchronize(s));
        {
            std::vector<unsigned long long> wg((size_t)grid * 4);
            VS_HIP(hipMemcpy(wg.data(), timing.as<unsigned long long>() + 16, wg.size() * 8, hipMemcpyDeviceToHost));
            double tmin = 1e30, tmax = 0, tsum = 0;
            double xs[8] = {0}, xc[8] = {0}, xn[8] = {0};
            unsigned long long s_min = ~0ull, s_max = 0, e_max = 0;
            for (int i = 0; i < grid; ++i) {
                const double t = (double)wg[4 * i] * 1e-5, cyc = (double)wg[4 * i + 1];       // ms
                tmin = std::min(tmin, t); tmax = std::max(tmax, t); tsum += t;
                const int x = (int)(wg[4 * i + 2] & 7);
                xs[x] += t; xc[x] += cyc; xn[x] += 1;
                s_min = std::min(s_min, wg[4 * i + 1]); s_max = std::max(s_max, wg[4 * i + 1]); e_max = std::max(e_max, wg[4 * i + 1] + wg[4 * i]);
            }
            fprintf(stderr, "[vsearch_hip] walk: workgroup time min %.2f mean %.2f max %.2f ms; starts spread over %.2f ms, first start -> last end %.2f ms;", tmin, tsum / grid, tmax,
                    (double)(s_max - s_min) * 1e-5, (double)(e_max - s_min) * 1e-5);
            fprintf(stderr, "\n");
            if (getenv("VS_BP_TIMING_WG")) {
                std::vector<int> ord(grid);
                for (int i = 0; i < grid; ++i) ord[i] = i;
                std::sort(ord.begin(), ord.end(), [&](int x, int y) { return wg[4 * x] > wg[4 * y]; });
                for (int j = 0; j < std::min(grid, 12); ++j) {
                    const int i = ord[j];
                    const unsigned hw = (unsigned)wg[4 * i + 3];
                    fprintf(stderr, "   wg %3d: %.2f ms, %.1f Mcycles, xcc %d se %d sh %d cu %d simd %d\n", i, (double)wg[4 * i] * 1e-5, (double)wg[4 * i + 1] * 1e-6, (int)(wg[4 * i + 2] & 7),
                            (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3);
                }
            }
        }
        if (idx->bp_quad) {
            const double bw = (double)std::max<unsigned long long>(1, h[5]);
            fprintf(stderr, "[vsearch_hip] quad walk, cycles per block and wave: gathers back %.0f, scan %.0f, barrier %.0f, emit %.0f, barrier (+ rest of the plan) %.0f\n", (double)h[6] / bw, (double)h[7] / bw,
                    (double)h[8] / bw, (double)h[9] / bw, (double)h[3] / bw);
            h[12] = h[14] = 0;
        }
        if (h[12] | h[14]) {
            const double bw = (double)std::max<unsigned long long>(1, h[5]) / 16.0;
            fprintf(stderr, "[vsearch_hip] binary walk, cycles per block: wave 0 walk %.0f wait %.0f; wave 8 walk %.0f wait %.0f\n", (double)h[12] / bw, (double)h[13] / bw,
                    (double)h[14] / bw, (double)h[15] / bw);
        }
        if (bp_flat_ok<AM_FIX>(idx, a)) {        // the flat walk has no dense part: slot 3 carries 100 MHz ticks
            fprintf(stderr, "[vsearch_hip] flat walk: shader clock %.0f MHz\n", 100.0 * (double)(h[0] + h[1] + h[2] + h[4]) / (double)std::max<unsigned long long>(1, h[3]));
            h[3] = 0;
        }
        const double tot = (double)(h[0] + h[1] + h[2] + h[3] + h[4]);
        fprintf(stderr, "[vsearch_hip] walk wave-cycles: prologue %.1f %%, list walk %.1f %%, barrier wait %.1f %%, dense %.1f %%, epilogue %.1f %%; per block and wave: "
                        "walk %.0f wait %.0f dense %.0f epilogue %.0f cycles\n", 100.0 * h[0] / tot, 100.0 * h[1] / tot, 100.0 * h[2] / tot, 100.0 * h[3] / tot, 100.0 * h[4] / tot,
                (double)h[1] / (double)std::max<unsigned long long>(1, h[5]), (double)h[2] / (double)std::max<unsigned long long>(1, h[5]),
                (double)h[3] / (double)std::max<unsigned long long>(1, h[5]), (double)h[4] / (double)std::max<unsigned long long>(1, h[5]));
    }
    VS_STAGE("filter walk", s);
    // 3. refine: exact scores of the K' candidates, the proof, the flags
    RefineArgs r{};
    r.cand = a.cand;
    r.n_cand = (int64_t)nchunk * kp;
    r.run_len = kp;
    r.B = B; r.k = k; r.kp = kp;
    r.pk_ptr = idx->pk_ptr.as<uint32_t>();
    r.cols = idx->cols.as<uint4>();
    r.vals = idx->vals.p;
    r.n_cols = V;
    r.n_rows = idx->n_rows;
    r.q = dq;
    r.qscale = qscale;
    r.qslack = qslack;
    r.qwsum = qwsum;
    r.quant = (idx->bp_quant || idx->bp_n_head > 0) ? 1 : 0;       // fp16-rounded values in the records and / or the dense strips
    r.force_flag = idx->bp_force_fb ? 1 : 0;
    r.id_offset = id_offset;
    r.out_ids = d_ids;
    r.out_scores = d_scores;
    r.out_ld = out_ld;
    r.flags = flags;
    {
        ProfScope prof("refine_topk", s);
        // the query's dense row in LDS when it fits beside the candidate buffers (V <= ~30 k): one workgroup per CU then
        const bool img = refine_lds_bytes(V, 1) <= (size_t)160 * 1024;
        const size_t rlds = refine_lds_bytes(V, img ? 1 : 0);
        const int rgrid = std::min(B, idx->cu_count * (img ? 1 : 2));
        void (*rk)(RefineArgs) = idx->store_dtype == VS_F32 ? (img ? refine_topk_kernel<VM_F32, 1> : refine_topk_kernel<VM_F32, 0>)
                               : idx->store_dtype == VS_F16 ? (img ? refine_topk_kernel<VM_F16, 1> : refine_topk_kernel<VM_F16, 0>)
                                                            : (img ? refine_topk_kernel<VM_BIN, 1> : refine_topk_kernel<VM_BIN, 0>);
        VS_HIP(hipFuncSetAttribute((const void*)rk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds));
        hipLaunchKernelGGL(rk, dim3(rgrid), dim3(kScanThreads), rlds, s, r);
        VS_HIP(hipGetLastError());
    }
    VS_STAGE("refine", s);
    // 4. flagged queries (too dense for a tile, or unproven; normally none): exact one-query scan of the CSR packets + merge
    {
        ProfScope prof("exact_fallback", s);
        hipLaunchKernelGGL(fb_plan_kernel<0>, dim3(1), dim3(kScanThreads), 0, s, flags, B, fb_tiles, fb_n);
        ScanArgs sa{};
        sa.pk_ptr = idx->pk_ptr.as<uint32_t>();
        sa.cols = idx->cols.as<uint4>();
        sa.vals = idx->vals.p;
        sa.q = dq;
        sa.n_rows = idx->n_rows;
        sa.n_cols = V;
        sa.B = B;
        sa.k = k;
        sa.nchunk = nchunk_fb;
        sa.rows_per_chunk = ceil_div64(idx->n_rows, nchunk_fb);
        sa.cand = a.cand;
        const size_t slds = scan_lds_bytes(V);
        void (*ek)(ScanArgs, const int2*, const int32_t*) = idx->store_dtype == VS_NONE  ? exact_scan_topk_kernel<VM_BIN>
                                                            : idx->store_dtype == VS_F16 ? exact_scan_topk_kernel<VM_F16> : exact_scan_topk_kernel<VM_F32>;
        VS_HIP(hipFuncSetAttribute((const void*)ek, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds));
        hipLaunchKernelGGL(ek, dim3(idx->cu_count), dim3(kScanThreads), slds, s, sa, (const int2*)fb_tiles, (const int32_t*)fb_n);
        MergeArgs m{};
        m.cand = a.cand;
        m.n_cand = (int64_t)nchunk_fb * k;
        m.B = B;
        m.k = k;
        m.id_offset = id_offset;
        m.out_ids = d_ids;
        m.out_scores = d_scores;
        m.out_ld = out_ld;
        m.col0 = 0;
        m.run_len = k;
        m.sel = fb_tiles;
        m.sel_n = fb_n;
        hipLaunchKernelGGL(merge_topk_kernel<0>, dim3(std::min(B, idx->cu_count)), dim3(kScanThreads), 0, s, m);
        VS_HIP(hipGetLastError());
    }
    VS_STAGE("fallback", s);
    idx->last_flags = flags;
    idx->last_flags_n = B;
    *done = true;
    return VS_OK;
}

// Multi-query pass (Qt = kQT).  Returns VS_OK and sets *done = false when the batch does not qualify
// (a query denser than the LDS weight capacity): the caller then takes the dense-image path.
// One pass delivers ranks [col0, col0 + k) of every query into columns col0.. of the [B, out_ld] outputs; `upper`
// ([B], nullable) holds the exclusive upper-bound keys on entry and the k-th keys of this pass on return.
int mq_search(vs_index* idx, const float* dq, int32_t B, int32_t k, int64_t id_offset, int64_t* d_ids, float* d_scores,
              const ScanPlan& plan, hipStream_t s, bool* done, int32_t out_ld, int32_t col0, uint64_t* upper) {
    *done = false;
    // Filter and refine (bp_refine.h): the walk runs on int32 fixed-point sums and returns K' > k documents per query, the refine
    // kernel re-scores them exactly and proves the top k; unproven queries go through the fp64 walk.  Without it (option
    // "postings_filter" = 0, "search after" passes, k beyond the candidate buffers) every tile takes the fp64 walk.
    if (bp_filter_ok(idx, k, col0, upper)) return bp_filter_search(idx, dq, B, k, id_offset, d_ids, d_scores, plan, s, done, out_ld);
    const bool filter_only = idx->bp_quant || idx->bp_quad || idx->store_dtype == VS_NONE || idx->bp_n_head > 0;    // lossy / binary records, quad chunks, dense strips: the filter only
    const bool use_bp = idx->bp_ready && !filter_only;                         // the fp64 walk over exact records
    const int qt_plan = use_bp ? kBpExactQT : kQT;
    const int bp_cap = kBpEntCap / 2;
    const int vals_cap = use_bp ? std::min(mq_vals_cap(idx), bp_cap) : mq_vals_cap(idx);     // entries (non-zeros) one tile may hold
    if (vals_cap <= 0 || k > (use_bp ? kBpMaxK : kMaxKMq)) return VS_OK;     // (callers split larger k into passes)
    const int V = idx->n_cols;
    // 1. sparsify the batch: counts -> (qptr, tiles, plan) -> (qcols, qvals)
    const size_t off_counts = 0, off_qptr = off_counts + (size_t)B * 8, off_plan = off_qptr + (size_t)(B + 1) * 8,
                 off_tiles = off_plan + 64, off_freq = off_tiles + (((size_t)B * sizeof(int2) + 15) & ~(size_t)15);
    VS_TRY(idx->ws_mq_meta.reserve(off_freq + (size_t)(V + 4) * 4 + 8));
    char* meta = idx->ws_mq_meta.as<char>();
    int64_t* counts = (int64_t*)(meta + off_counts);
    int64_t* qptr = (int64_t*)(meta + off_qptr);
    int64_t* dplan = (int64_t*)(meta + off_plan);
    int2* tiles = (int2*)(meta + off_tiles);
    uint32_t* colfreq = (uint32_t*)(meta + off_freq);
    VS_HIP(hipMemsetAsync(colfreq, 0, (size_t)(V + 4) * 4 + 8, s));          // counts + the 64-bit overlap sum behind them
    hipLaunchKernelGGL(count_nz_kernel<0>, dim3(std::min(B, 2048)), dim3(kSpThreads), 0, s, dq, (int64_t)V, B, V, counts);
    hipLaunchKernelGGL(mq_colfreq_kernel<0>, dim3(std::min(B, 2048)), dim3(kSpThreads), 0, s, dq, (int64_t)V, B, V, colfreq);
    hipLaunchKernelGGL(mq_plan_kernel<0>, dim3(1), dim3(64), 0, s, counts, B, qt_plan, vals_cap, qptr, tiles, dplan, colfreq, V);
    VS_HIP(hipGetLastError());
    if (use_bp && idx->bp_df.p)
        hipLaunchKernelGGL(bp_walk_kernel<0>, dim3(1), dim3(kScanThreads), 0, s, colfreq, idx->bp_df.as<unsigned long long>(),
                           idx->bp_df.as<unsigned long long>() + V, V, dplan + 4);
    int64_t hplan[6] = {0, 0, 0, 0, 0, 0};
    VS_HIP(hipMemcpyAsync(hplan, dplan, sizeof(hplan), hipMemcpyDeviceToHost, s));
    VS_HIP(hipStreamSynchronize(s));
    if (hplan[1] > vals_cap) return VS_OK;                       // some query is too dense for the tile tables
    const int n_tiles = (int)hplan[0];
    const int64_t qnnz = hplan[2];
    VS_TRY(idx->ws_mq_q.reserve(std::max<size_t>((size_t)qnnz * 8, 16)));
    int32_t* qcols = idx->ws_mq_q.as<int32_t>();
    float* qvals = reinterpret_cast<float*>(qcols + qnnz);
    hipLaunchKernelGGL(fill_csr_kernel<0>, dim3(std::min(B, 2048)), dim3(kSpThreads), 0, s, dq, (int64_t)V, B, V, qptr, qcols, qvals, qnnz);
    VS_HIP(hipGetLastError());
    VS_STAGE("sparsify", s);
    if (debug_sync_on()) fprintf(stderr, "[vsearch_hip] plan: tiles %d qnnz %lld max %lld cap %d\n", n_tiles, (long long)qnnz, (long long)hplan[1], vals_cap);
    // 2. scan.  Work items = (tile, row chunk)
    int nchunk = choose_chunks(idx, n_tiles, plan.nchunk);
    if (use_bp) {
        // blocked postings: chunks are runs of blocks
        const int64_t n_blocks = ceil_div64(idx->n_rows, idx->bp_rows);
        nchunk = (int)std::min<int64_t>(nchunk, n_blocks);
        nchunk = bp_choose_chunks(idx, n_tiles, n_blocks, plan.nchunk);
        const int64_t blocks_per_chunk = ceil_div64(n_blocks, nchunk);
        const int64_t items = (int64_t)n_tiles * nchunk;
        const int grid = (int)std::min<int64_t>(items, idx->cu_count);
        VS_TRY(idx->ws_mq_cand.reserve((size_t)idx->cu_count * kQT * kBpCap * 8));
        VS_TRY(idx->ws_cand.reserve((size_t)B * nchunk * k * 8));
        const int RS = bp_rec_bytes(bp_record_vm(idx));
        BpArgs a{};
        a.rows = idx->bp_rows;
        a.dir = idx->bp_dir.as<uint32_t>();
        a.al_shift = idx->bp_al_shift;
    a.al_shift = idx->bp_al_shift;
        a.base = idx->bp_base.as<unsigned long long>();
        a.rec = idx->bp_rec.as<char>();
        a.n_rows = idx->n_rows;
        a.n_cols = V;
        a.k = k;
        a.nchunk = nchunk;
        a.blocks_per_chunk = blocks_per_chunk;
        a.qptr = qptr;
        a.qcols = qcols;
        a.qvals = qvals;
        a.tiles = tiles;
        a.n_tiles = n_tiles;
        a.ent_cap = vals_cap;
        a.cand = idx->ws_cand.as<uint64_t>();
        a.gcand = idx->ws_mq_cand.as<uint64_t>();
        a.upper = col0 > 0 ? upper : nullptr;
        // what this launch has to read: the records of the batch's (query, column) entries + one directory pair per entry and block
        idx->last_scan_bytes += hplan[4] * RS + qnnz * n_blocks * 4;
        idx->last_walk_postings += hplan[5];
        idx->last_path = 2;
        ProfScope prof("csr_scan_topk", s);
        VS_TRY((launch_bp_walk<kBpExactQT, AM_F64>(idx, a, grid, vals_cap, s)));
        VS_STAGE("fp64 walk", s);
    } else {
    const int64_t rows_per_chunk = ceil_div64(idx->n_rows, nchunk);
    const int64_t items = (int64_t)n_tiles * nchunk;
    const int grid = (int)std::min<int64_t>(items, idx->cu_count);
    VS_TRY(idx->ws_mq_cand.reserve((size_t)grid * kQT * kMqCap * 8));
    VS_TRY(idx->ws_cand.reserve((size_t)B * nchunk * k * 8));
    MqArgs a{};
    a.pk_ptr = idx->pk_ptr.as<uint32_t>();
    a.cols = idx->cols.as<uint4>();
    a.vals = idx->vals.p;
    a.n_rows = idx->n_rows;
    a.n_cols = V;
    a.k = k;
    a.nchunk = nchunk;
    a.rows_per_chunk = rows_per_chunk;
    a.qptr = qptr;
    a.qcols = qcols;
    a.qvals = qvals;
    a.tiles = tiles;
    a.n_tiles = n_tiles;
    a.vals_cap = vals_cap;
    a.cand = idx->ws_cand.as<uint64_t>();
    a.gcand = idx->ws_mq_cand.as<uint64_t>();
    a.upper = col0 > 0 ? upper : nullptr;
    const size_t lds = mq_fixed_lds_bytes<kQT>(V, mq_acc_rows(idx)) + (size_t)vals_cap * 4;
    idx->last_scan_bytes += (int64_t)n_tiles * csr_bytes_per_pass(idx);
    idx->last_path = 1;
    {
        ProfScope prof("csr_scan_topk", s);
        // packets per lane per trip: enough to cover an average row in one trip, at most 3
        const double ppr = idx->n_rows > 0 ? (double)idx->n_packets / (double)idx->n_rows : 1.0;
        const int u = std::max(1, std::min(3, (int)((ppr + mq_lanes(idx) - 1) / mq_lanes(idx))));
        // expected number of columns two queries of the batch share; uniform 776-nnz queries: ~20
        const double overlap = B > 1 ? (double)hplan[3] / ((double)B * (double)(B - 1)) : 0.0;
        const bool shared_cols = idx->mq_variant >= 0 ? idx->mq_variant == 1 : overlap > kMqSharedOverlap;
        int rc = idx->store_dtype == VS_F32 ? launch_mq_vm<VM_F32>(mq_lanes(idx), u, shared_cols, a, grid, lds, s)
               : idx->store_dtype == VS_F16 ? launch_mq_vm<VM_F16>(mq_lanes(idx), u, shared_cols, a, grid, lds, s)
                                            : launch_mq_vm<VM_BIN>(mq_lanes(idx), u, shared_cols, a, grid, lds, s);
        VS_TRY(rc);
    }
    }
    // 3. merge chunks
    MergeArgs m{};
    m.cand = idx->ws_cand.as<uint64_t>();
    m.n_cand = (int64_t)nchunk * k;
    m.B = B;
    m.k = k;
    m.id_offset = id_offset;
    m.out_ids = d_ids;
    m.out_scores = d_scores;
    m.out_ld = out_ld;
    m.col0 = col0;
    m.upper_out = upper;
    m.run_len = k;                                 // every chunk's list is sorted
    {
        ProfScope prof("merge_topk", s);
        hipLaunchKernelGGL(merge_topk_kernel<0>, dim3(std::min(B, idx->cu_count * 2)), dim3(kScanThreads), 0, s, m);
    }
    VS_HIP(hipGetLastError());
    *done = true;
    return VS_OK;
}

}  // namespace

int vs_csr_prepare(vs_index* idx, hipStream_t s) { return csr_prepare_impl(idx, s); }

int vs_csr_search(vs_index* idx, const void* q, int q_dtype, int64_t ldq, int32_t B, int32_t k, int64_t id_offset,
                  int64_t* out_ids, float* out_scores, hipStream_t s) {
    const float* dq = nullptr;
    VS_TRY(prep_queries(idx, q, q_dtype, ldq, B, s, &dq));
    const ScanPlan plan = plan_scan(idx, B);
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
    idx->last_qt = 1;
    idx->last_scan_bytes = 0;
    idx->last_walk_postings = 0;
    idx->last_flags = nullptr;
    idx->last_flags_n = 0;
    idx->last_plan_dev = nullptr;
    idx->last_path = 0;
    if (idx->qt_pref != 1) {
        if (!idx->bp_ready && !idx->bp_tried && bp_wanted(idx)) VS_TRY(bp_build(idx, s));
        if (!idx->bp_ready && !idx->bp_tried) idx->bp_state = 4;
        bool done = false;
        // k > kMaxKMq: "search after" passes of kMaxKMq ranks each (the k-th key of a pass is the next pass's exclusive
        // upper bound); large batches are cut so that the candidate scratch stays bounded
        // ranks one pass delivers: the whole k when the filter-and-refine search takes it (k + its margin within the candidate
        // buffers), else 1024 per pass of the fp64 postings walk (exact records only), else 512 per pass of the CSR scan
        const bool one_pass = idx->bp_ready && idx->bp_filter != 0 && k + std::max(28, k / 4) <= kBpMaxK;
        const int max_k = one_pass ? k : (idx->bp_ready && !idx->bp_quant && !idx->bp_quad && idx->store_dtype != VS_NONE && idx->bp_n_head == 0) ? kBpMaxK : kMaxKMq;
        const int mq_passes = ceil_div(k, max_k);
        const int kk_mq = std::min<int>(k, max_k);
        DevBuf mq_upper;
        if (mq_passes > 1) VS_TRY(mq_upper.alloc((size_t)B * 8));
        const size_t per_q_mq = (size_t)plan.nchunk * kk_mq * 8;
        const int bs_mq = (int)std::max<size_t>(1, std::min<size_t>((size_t)B, ((size_t)1 << 30) / per_q_mq));
        bool all = true;
        for (int pass = 0; pass < mq_passes && all; ++pass) {
            const int col0 = pass * max_k;
            const int kk = std::min(k - col0, max_k);
            for (int b0 = 0; b0 < B && all; b0 += bs_mq) {
                const int bs = std::min(bs_mq, B - b0);
                VS_TRY(mq_search(idx, dq + (size_t)b0 * idx->n_cols, bs, kk, id_offset, d_ids + (size_t)b0 * k, d_scores + (size_t)b0 * k, plan, s,
                                 &done, k, col0, mq_passes > 1 ? mq_upper.as<uint64_t>() + b0 : nullptr));
                all = all && done;
            }
        }
        if (mq_passes > 1) VS_HIP(hipStreamSynchronize(s));      // `mq_upper` is freed on return
        if (all) {
            idx->last_qt = kQT;
            if (!out_dev) {
                VS_HIP(hipMemcpyAsync(out_ids, d_ids, (size_t)B * k * 8, hipMemcpyDeviceToHost, s));
                VS_HIP(hipMemcpyAsync(out_scores, d_scores, (size_t)B * k * 4, hipMemcpyDeviceToHost, s));
                VS_HIP(hipStreamSynchronize(s));
            }
            return VS_OK;
        }
    }
    idx->last_scan_bytes = 0;                                   // one query per pass from here on
    idx->last_path = 0;
    const int passes = ceil_div(k, kMaxKShared);
    DevBuf upper;                                              // [B] exclusive upper-bound keys (multi-pass only)
    if (passes > 1) {
        VS_TRY(upper.alloc((size_t)B * 8));
        VS_HIP(hipMemsetAsync(upper.p, 0xFF, (size_t)B * 8, s));
    }
    // bound candidate scratch: process queries in sub-batches
    const int kk_max = std::min<int>(k, kMaxKShared);
    const size_t per_q = (size_t)plan.nchunk * kk_max * 8;
    const int bs_max = (int)std::max<size_t>(1, std::min<size_t>((size_t)B, ((size_t)512 << 20) / per_q));
    VS_TRY(idx->ws_cand.reserve(per_q * bs_max));
    // few, long row chunks when many queries run side by side (they share the index stream on chip)
    const int nchunk1 = choose_chunks(idx, std::min(B, bs_max), plan.nchunk);
    const int64_t rows_per_chunk1 = ceil_div64(idx->n_rows, nchunk1);
    for (int pass = 0; pass < passes; ++pass) {
        const int col0 = pass * kMaxKShared;
        const int kk = std::min(k - col0, kMaxKShared);
        for (int b0 = 0; b0 < B; b0 += bs_max) {
            const int bs = std::min(bs_max, B - b0);
            ScanArgs a{};
            a.pk_ptr = idx->pk_ptr.as<uint32_t>();
            a.cols = idx->cols.as<uint4>();
            a.vals = idx->vals.p;
            a.q = dq + (size_t)b0 * idx->n_cols;
            a.n_rows = idx->n_rows;
            a.n_cols = idx->n_cols;
            a.B = bs;
            a.k = kk;
            a.nchunk = nchunk1;
            a.rows_per_chunk = rows_per_chunk1;
            a.cand = idx->ws_cand.as<uint64_t>();
            a.upper = passes > 1 ? upper.as<uint64_t>() + b0 : nullptr;
            const int grid = (int)std::min<int64_t>((int64_t)bs * nchunk1, idx->cu_count);
            idx->last_scan_bytes += (int64_t)bs * csr_bytes_per_pass(idx);
            VS_TRY(launch_scan(idx, kk <= kMaxKWave ? 1 : 2, a, grid, s));
            MergeArgs m{};
            m.cand = a.cand;
            m.n_cand = (int64_t)nchunk1 * kk;
            m.B = bs;
            m.k = kk;
            m.id_offset = id_offset;
            m.out_ids = d_ids + (size_t)b0 * k;
            m.out_scores = d_scores + (size_t)b0 * k;
            m.out_ld = k;
            m.col0 = col0;
            m.run_len = kk;
            m.upper_out = passes > 1 ? upper.as<uint64_t>() + b0 : nullptr;
            {
                ProfScope prof("merge_topk", s);
                hipLaunchKernelGGL(merge_topk_kernel<0>, dim3(std::min(bs, idx->cu_count * 2)), dim3(kScanThreads), 0, s, m);
            }
            VS_HIP(hipGetLastError());
        }
    }
    if (!out_dev) {
        VS_HIP(hipMemcpyAsync(out_ids, d_ids, (size_t)B * k * 8, hipMemcpyDeviceToHost, s));
        VS_HIP(hipMemcpyAsync(out_scores, d_scores, (size_t)B * k * 4, hipMemcpyDeviceToHost, s));
        VS_HIP(hipStreamSynchronize(s));
    }
    if (passes > 1) VS_HIP(hipStreamSynchronize(s));          // `upper` is freed on return
    return VS_OK;
}

int vs_csr_scores(vs_index* idx, const void* q, int q_dtype, int64_t ldq, int32_t B, float* out_scores, hipStream_t s) {
    const float* dq = nullptr;
    VS_TRY(prep_queries(idx, q, q_dtype, ldq, B, s, &dq));
    const ScanPlan plan = plan_scan(idx, B);
    const bool out_dev = is_device_ptr(out_scores);
    float* d_scores = out_scores;
    const size_t bytes = (size_t)B * idx->n_rows * 4;
    if (!out_dev) {
        VS_TRY(idx->ws_out_scores.reserve(bytes));
        d_scores = idx->ws_out_scores.as<float>();
    }
    ScanArgs a{};
    a.pk_ptr = idx->pk_ptr.as<uint32_t>();
    a.cols = idx->cols.as<uint4>();
    a.vals = idx->vals.p;
    a.q = dq;
    a.n_rows = idx->n_rows;
    a.n_cols = idx->n_cols;
    a.B = B;
    a.k = 0;
    a.nchunk = plan.nchunk;
    a.rows_per_chunk = plan.rows_per_chunk;
    a.all_scores = d_scores;
    VS_TRY(launch_scan(idx, 0, a, plan.grid, s));
    if (!out_dev) {
        VS_HIP(hipMemcpyAsync(out_scores, d_scores, bytes, hipMemcpyDeviceToHost, s));
        VS_HIP(hipStreamSynchronize(s));
    }
    return VS_OK;
}

extern "C" int vs_merge_topk(const int64_t* cand_ids, const float* cand_scores, int32_t B, int64_t n_cand, int32_t k,
                             int64_t* out_ids, float* out_scores, int device, void* stream) {
    if (!cand_ids || !cand_scores || !out_ids || !out_scores) return fail(VS_EINVAL, "NULL argument");
    if (B <= 0 || k <= 0) return fail(VS_EINVAL, "B and k must be positive");
    if (k > n_cand) return fail(VS_ERANGE, "selected index k out of range (k = %d > %lld candidates)", k, (long long)n_cand);
    if (k > kMaxKShared) return fail(VS_EUNSUPPORTED, "vs_merge_topk supports k <= %d", kMaxKShared);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return fail(VS_ENODEVICE, "no HIP device visible"); }
    VS_HIP(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)B * n_cand;
    DevBuf st_ids, st_sc, o_ids, o_sc;
    DevBuf& keys = device_scratch(device, 3);                 // kept between calls: the sharded search merges once per batch
    const void *d_ids = nullptr, *d_sc = nullptr;
    VS_TRY(to_device(cand_ids, n * 8, st_ids, s, &d_ids));
    VS_TRY(to_device(cand_scores, n * 4, st_sc, s, &d_sc));
    VS_TRY(keys.reserve(n * 8));
    hipLaunchKernelGGL(keys_from_pairs_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64((int64_t)n, 256), 4096)), dim3(256), 0, s,
                       (const int64_t*)d_ids, (const float*)d_sc, (int64_t)n, keys.as<uint64_t>());
    VS_HIP(hipGetLastError());
    const bool out_dev = is_device_ptr(out_ids);
    int64_t* di = out_ids;
    float* ds = out_scores;
    if (!out_dev) {
        VS_TRY(o_ids.alloc((size_t)B * k * 8));
        VS_TRY(o_sc.alloc((size_t)B * k * 4));
        di = o_ids.as<int64_t>();
        ds = o_sc.as<float>();
    }
    MergeArgs m{};
    m.cand = keys.as<uint64_t>();
    m.n_cand = n_cand;
    m.B = B;
    m.k = k;
    m.id_offset = 0;
    m.out_ids = di;
    m.out_scores = ds;
    m.out_ld = k;
    m.col0 = 0;
    hipLaunchKernelGGL(merge_topk_kernel<0>, dim3(std::min(B, 512)), dim3(kScanThreads), 0, s, m);
    VS_HIP(hipGetLastError());
    if (!out_dev) {
        VS_HIP(hipMemcpyAsync(out_ids, di, (size_t)B * k * 8, hipMemcpyDeviceToHost, s));
        VS_HIP(hipMemcpyAsync(out_scores, ds, (size_t)B * k * 4, hipMemcpyDeviceToHost, s));
    }
    if (!out_dev || st_ids.p || st_sc.p || !s) VS_HIP(hipStreamSynchronize(s));      // host buffers / staging die here
    return VS_OK;
}

// =================================================================================================
// native shard files (".vsx"): the device format written / read verbatim -- no CSR round trip, no
// decompression; a 97 GB Wiki21M index loads at storage speed instead of through scipy's .npz
// (index.py:172-176 re-parses, slices and vstacks the shards on the host every time)
// =================================================================================================
namespace {
struct VsxHeader {
    char magic[8];            // "VSXCSR1\0"
    int32_t store_dtype, n_cols;
    int64_t n_rows, n_packets, nnz;
    int64_t reserved[4];
};

int copy_dev_to_file(FILE* f, const void* dev, size_t bytes) {
    const size_t chunk = (size_t)64 << 20;
    std::vector<char> host(std::min(bytes, chunk));
    for (size_t off = 0; off < bytes; off += chunk) {
        const size_t n = std::min(chunk, bytes - off);
        VS_HIP(hipMemcpy(host.data(), (const char*)dev + off, n, hipMemcpyDeviceToHost));
        if (fwrite(host.data(), 1, n, f) != n) return fail(VS_EINVAL, "short write");
    }
    return VS_OK;
}
// file -> device through two pinned 64 MB buffers: the read of chunk i + 1 overlaps the DMA of chunk i
struct PinnedPair {
    void* buf[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
    size_t chunk = (size_t)64 << 20;
    bool busy[2] = {false, false};
    int init() {
        for (int i = 0; i < 2; ++i) {
            VS_HIP(hipHostMalloc(&buf[i], chunk, hipHostMallocDefault));
            VS_HIP(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        }
        VS_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        return VS_OK;
    }
    ~PinnedPair() {
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        for (int i = 0; i < 2; ++i) {
            if (done[i]) (void)hipEventDestroy(done[i]);
            if (buf[i]) (void)hipHostFree(buf[i]);
        }
    }
};
int copy_file_to_dev(FILE* f, void* dev, size_t bytes, PinnedPair& pp, int* turn) {
    for (size_t off = 0; off < bytes; off += pp.chunk) {
        const size_t n = std::min(pp.chunk, bytes - off);
        const int b = *turn;
        if (pp.busy[b]) VS_HIP(hipEventSynchronize(pp.done[b]));              // the DMA that last used this buffer has finished
        if (fread(pp.buf[b], 1, n, f) != n) return fail(VS_EINVAL, "short read: truncated .vsx file");
        VS_HIP(hipMemcpyAsync((char*)dev + off, pp.buf[b], n, hipMemcpyHostToDevice, pp.stream));
        VS_HIP(hipEventRecord(pp.done[b], pp.stream));
        pp.busy[b] = true;
        *turn = b ^ 1;
    }
    return VS_OK;
}

// A loaded shard file is searched as is: check what the scan kernels rely on.  flags: |1 row pointers not monotone / last != packets,
// |2 a column id above n_cols, |4 a real column after a pad column inside a row, |8 pad column outside a row's last packet;
// nnz_out = non-pad column slots.
template <int UNUSED>
__global__ __launch_bounds__(256) void validate_packets_kernel(const uint32_t* pk_ptr, const uint16_t* cols, int64_t n_rows, int64_t n_packets, int32_t n_cols,
                                                               int* flags, unsigned long long* nnz_out) {
    unsigned long long nnz = 0;
    int bad = 0;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * 256) {
        const uint32_t p0 = pk_ptr[r], p1 = pk_ptr[r + 1];
        if (p1 < p0 || (int64_t)p1 > n_packets) { bad |= 1; continue; }
        bool seen_pad = false;
        for (uint32_t p = p0; p < p1; ++p) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t c = cols[(size_t)p * 8 + i];
                if (c > (uint32_t)n_cols) bad |= 2;
                if (c == (uint32_t)n_cols) {
                    seen_pad = true;
                    if (p + 1 != p1) bad |= 8;
                } else {
                    if (seen_pad) bad |= 4;
                    ++nnz;
                }
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (pk_ptr[0] != 0u || (int64_t)pk_ptr[n_rows] != n_packets)) bad |= 1;
    if (bad) atomicOr(flags, bad);
    for (int o = 32; o > 0; o >>= 1) nnz += __shfl_xor(nnz, o, 64);
    if ((threadIdx.x & 63) == 0 && nnz) atomicAdd(nnz_out, nnz);
}
}  // namespace

extern "C" int vs_index_save_native(const vs_index* idx, const char* path) {
    if (!idx || !path) return fail(VS_EINVAL, "NULL argument");
    if (idx->kind != VS_KIND_CSR) return fail(VS_EINVAL, "native shard files hold CSR indexes");
    VS_HIP(hipSetDevice(idx->device));
    FILE* f = fopen(path, "wb");
    if (!f) return fail(VS_EINVAL, "cannot open %s for writing", path);
    VsxHeader h{};
    memcpy(h.magic, "VSXCSR1", 8);
    h.store_dtype = idx->store_dtype;
    h.n_cols = idx->n_cols;
    h.n_rows = idx->n_rows;
    h.n_packets = idx->n_packets;
    h.nnz = idx->nnz;
    h.reserved[0] = idx->logical_dense ? 1 : 0;
    int rc = fwrite(&h, sizeof(h), 1, f) == 1 ? VS_OK : fail(VS_EINVAL, "short write");
    if (rc == VS_OK) rc = copy_dev_to_file(f, idx->pk_ptr.p, (size_t)(idx->n_rows + 1) * 4);
    if (rc == VS_OK) rc = copy_dev_to_file(f, idx->cols.p, (size_t)idx->n_packets * 16);
    if (rc == VS_OK && idx->store_dtype != VS_NONE) rc = copy_dev_to_file(f, idx->vals.p, (size_t)idx->n_packets * (idx->store_dtype == VS_F32 ? 32 : 16));
    fclose(f);
    return rc;
}

extern "C" int vs_index_load_native(const char* path, int device, vs_index** out) {
    if (!path || !out) return fail(VS_EINVAL, "NULL argument");
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) return fail(VS_EINVAL, "cannot open %s", path);
    struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
    VsxHeader h{};
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "VSXCSR1", 8) != 0) return fail(VS_EINVAL, "%s is not a vsearch native shard file", path);
    VS_TRY(check_csr_shape(h.n_rows, h.n_cols, h.store_dtype));
    if (h.n_packets < 0 || h.n_packets >= (1ll << 32) || h.nnz < 0 || h.nnz > h.n_packets * 8) return fail(VS_EINVAL, "corrupt header");
    // the payload must be exactly what the header announces
    const size_t b_ptr = (size_t)(h.n_rows + 1) * 4, b_cols = (size_t)h.n_packets * 16,
                 b_vals = h.store_dtype == VS_NONE ? 0 : (size_t)h.n_packets * (h.store_dtype == VS_F32 ? 32 : 16);
    {
        const long here = ftell(f);
        if (fseek(f, 0, SEEK_END) != 0) return fail(VS_EINVAL, "cannot seek in %s", path);
        const long long end = ftell(f);
        if (fseek(f, here, SEEK_SET) != 0) return fail(VS_EINVAL, "cannot seek in %s", path);
        if ((unsigned long long)end != sizeof(h) + b_ptr + b_cols + b_vals)
            return fail(VS_EINVAL, "%s: %lld bytes on disk, the header announces %zu (truncated or mismatched file)", path, end, sizeof(h) + b_ptr + b_cols + b_vals);
    }
    vs_index* idx = nullptr;
    VS_TRY(vs_index_create_reserved(h.n_rows, h.n_packets, h.n_cols, h.store_dtype, device, &idx));
    struct Guard { vs_index* p; ~Guard() { if (p) vs_index_destroy(p); } } guard{idx};
    {
        PinnedPair pp;
        VS_TRY(pp.init());
        int turn = 0;
        const auto t0 = std::chrono::steady_clock::now();
        VS_TRY(copy_file_to_dev(f, idx->pk_ptr.p, b_ptr, pp, &turn));
        VS_TRY(copy_file_to_dev(f, idx->cols.p, b_cols, pp, &turn));
        if (b_vals) VS_TRY(copy_file_to_dev(f, idx->vals.p, b_vals, pp, &turn));
        VS_HIP(hipStreamSynchronize(pp.stream));
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        idx->load_GBps = dt > 0 ? (double)(b_ptr + b_cols + b_vals) / dt / 1e9 : 0.0;
        if (getenv("VS_VERBOSE")) fprintf(stderr, "[vsearch_hip] %s: %.2f GB in %.2f s = %.2f GB/s (file -> pinned -> HBM)\n", path,
                                          (double)(b_ptr + b_cols + b_vals) / 1e9, dt, idx->load_GBps);
    }
    // the scan kernels index LDS tables by column id and walk packets by the row pointers: refuse a payload that would send them astray
    {
        DevBuf chk;
        VS_TRY(chk.alloc(16));
        VS_HIP(hipMemset(chk.p, 0, 16));
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div64(h.n_rows, 256), 8192));
        hipLaunchKernelGGL(validate_packets_kernel<0>, dim3(grid), dim3(256), 0, 0, idx->pk_ptr.as<uint32_t>(), idx->cols.as<uint16_t>(), h.n_rows, h.n_packets,
                           h.n_cols, chk.as<int>(), reinterpret_cast<unsigned long long*>(chk.as<char>() + 8));
        VS_HIP(hipGetLastError());
        struct { int flags; int pad; unsigned long long nnz; } res;
        VS_HIP(hipMemcpy(&res, chk.p, 16, hipMemcpyDeviceToHost));
        if (res.flags & 1) return fail(VS_EINVAL, "%s: row pointers are not monotone or do not end at %lld packets", path, (long long)h.n_packets);
        if (res.flags & 2) return fail(VS_EINVAL, "%s: column id above n_cols = %d", path, h.n_cols);
        if (res.flags & 12) return fail(VS_EINVAL, "%s: pad columns inside a row (rows are padded at their tail only)", path);
        if ((long long)res.nnz != h.nnz) return fail(VS_EINVAL, "%s: %llu non-zeros in the payload, the header says %lld", path, res.nnz, (long long)h.nnz);
    }
    idx->n_rows = h.n_rows;
    idx->n_packets = h.n_packets;
    idx->nnz = h.nnz;
    idx->logical_dense = h.reserved[0] == 1;
    idx->lanes_per_row = pick_lanes_per_row(idx->n_packets, idx->n_rows);
    guard.p = nullptr;
    *out = idx;
    return VS_OK;
}
