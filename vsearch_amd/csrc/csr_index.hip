// csr_index.hip -- SparseIndex / BoTIndex device container and its search entry points.
//   reference: src/ir/retriever/index.py:128-218 (containers), :88-94 (search)
#include "csr_internal.h"
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

namespace {
// a row slice's pk_ptr (still the source's offsets, first packet p0): out[0] = non-pad column slots of the slice (its nnz) ...
template <int UNUSED>
__global__ __launch_bounds__(256) void slice_count_kernel(const uint32_t* pk_ptr, int64_t n_rows, uint32_t p0, const uint16_t* cols, int32_t n_cols, unsigned long long* nnz_out) {
    unsigned long long nnz = 0;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * 256) {
        const uint32_t a = pk_ptr[r] - p0, b = pk_ptr[r + 1] - p0;
        if (b > a) {                                           // pads sit at the tail of a row's last packet only
            nnz += (unsigned long long)(b - a - 1) * 8ull;
#pragma unroll
            for (int i = 0; i < 8; ++i) nnz += cols[(size_t)(b - 1) * 8 + i] != (uint16_t)n_cols ? 1ull : 0ull;
        }
    }
    for (int o = 32; o > 0; o >>= 1) nnz += __shfl_xor(nnz, o, 64);
    if ((threadIdx.x & 63) == 0 && nnz) atomicAdd(nnz_out, nnz);
}
// ... then rebased to the slice's first packet (a second launch: every element on its own)
template <int UNUSED>
__global__ __launch_bounds__(256) void slice_rebase_kernel(uint32_t* pk_ptr, int64_t n_rows, uint32_t p0) {
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r <= n_rows; r += (int64_t)gridDim.x * 256) pk_ptr[r] -= p0;
}
}  // namespace

// Rows [row0, row0 + n_rows) of a CSR index as a NEW index on `device` (any GPU): the packets are copied device to device (a peer
// copy across GPUs), the host never sees them.  This is what row-range sharding is made of ("per-shard npz or
// row ranges"): one .npz / .vsx file, or an index built in memory, dealt over several GPUs.
extern "C" int vs_index_slice_rows(const vs_index* src, int64_t row0, int64_t n_rows, int device, vs_index** out) {
    if (!out) return fail(VS_EINVAL, "out is NULL");
    *out = nullptr;
    if (!src || src->kind != VS_KIND_CSR) return fail(VS_EINVAL, "not a CSR index");
    if (row0 < 0 || n_rows < 0 || row0 + n_rows > src->n_rows) return fail(VS_ERANGE, "rows [%lld, %lld) outside the index's %lld rows", (long long)row0, (long long)(row0 + n_rows), (long long)src->n_rows);
    VS_HIP(hipSetDevice(src->device));
    uint32_t pr[2] = {0u, 0u};
    VS_HIP(hipMemcpy(&pr[0], src->pk_ptr.as<uint32_t>() + row0, 4, hipMemcpyDeviceToHost));
    VS_HIP(hipMemcpy(&pr[1], src->pk_ptr.as<uint32_t>() + row0 + n_rows, 4, hipMemcpyDeviceToHost));
    const int64_t packets = (int64_t)pr[1] - (int64_t)pr[0];
    if (packets < 0) return fail(VS_EINVAL, "row pointers decrease");
    vs_index* idx = nullptr;
    VS_TRY(vs_index_create_reserved(n_rows, packets, src->n_cols, src->store_dtype, device, &idx));
    struct Guard { vs_index* p; ~Guard() { if (p) vs_index_destroy(p); } } guard{idx};
    const size_t vb = src->store_dtype == VS_F32 ? 32 : (src->store_dtype == VS_F16 ? 16 : 0);
    auto copy = [&](void* dst, const void* from, size_t bytes) -> int {
        if (bytes == 0) return VS_OK;
        if (device == src->device) VS_HIP(hipMemcpy(dst, from, bytes, hipMemcpyDeviceToDevice));
        else VS_HIP(hipMemcpyPeer(dst, device, from, src->device, bytes));
        return VS_OK;
    };
    VS_TRY(copy(idx->pk_ptr.p, src->pk_ptr.as<uint32_t>() + row0, (size_t)(n_rows + 1) * 4));
    VS_TRY(copy(idx->cols.p, src->cols.as<char>() + (size_t)pr[0] * 16, (size_t)packets * 16));
    if (vb) VS_TRY(copy(idx->vals.p, src->vals.as<char>() + (size_t)pr[0] * vb, (size_t)packets * vb));
    VS_HIP(hipSetDevice(device));
    DevBuf cnt;
    VS_TRY(cnt.alloc(8));
    VS_HIP(hipMemset(cnt.p, 0, 8));
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div64(n_rows + 1, 256), 4096));
    hipLaunchKernelGGL(slice_count_kernel<0>, dim3(grid), dim3(256), 0, 0, idx->pk_ptr.as<uint32_t>(), n_rows, pr[0], idx->cols.as<uint16_t>(), src->n_cols,
                       cnt.as<unsigned long long>());
    hipLaunchKernelGGL(slice_rebase_kernel<0>, dim3(grid), dim3(256), 0, 0, idx->pk_ptr.as<uint32_t>(), n_rows, pr[0]);
    VS_HIP(hipGetLastError());
    unsigned long long nnz = 0;
    VS_HIP(hipMemcpy(&nnz, cnt.p, 8, hipMemcpyDeviceToHost));
    idx->n_rows = n_rows;
    idx->n_packets = packets;
    idx->nnz = (int64_t)nnz;
    idx->logical_dense = src->logical_dense;
    idx->lanes_per_row = pick_lanes_per_row(idx->n_packets, idx->n_rows);
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
    o->postings_walk = !idx->bp_ready ? -1 : idx->bp_quad ? 4 : idx->bp_bq ? 6 : (idx->store_dtype == VS_NONE && idx->bp_walk_pref != 0) ? 5 : 0;
    o->last_packed_tiles = 0;
    if (idx->last_path == 3 && idx->last_plan_dev) {               // the filter search keeps its plan on the device: read it now
        int64_t hp[6] = {0, 0, 0, 0, 0, 0};
        VS_HIP(hipSetDevice(idx->device));
        VS_HIP(hipDeviceSynchronize());
        VS_HIP(hipMemcpy(hp, idx->last_plan_dev, sizeof(hp), hipMemcpyDeviceToHost));
        // records + one directory word per (entry, block); quad chunks: every chunk the walk reads (main chunks, empty ones included, and
        // overflow chunks -- quad_count_kernel counts them all), no directory
        o->last_scan_bytes = (idx->bp_quad || idx->bp_bq) ? hp[4] * idx->last_plan_rs : hp[4] * idx->last_plan_rs + hp[2] * idx->last_plan_blocks * 4;
        o->last_walk_postings = hp[5];
    }
    if (idx->last_path == 3 && idx->last_split_dev) {
        int32_t hs[2] = {0, 0};
        VS_HIP(hipSetDevice(idx->device));
        VS_HIP(hipDeviceSynchronize());
        VS_HIP(hipMemcpy(hs, idx->last_split_dev, sizeof(hs), hipMemcpyDeviceToHost));
        o->last_packed_tiles = hs[0];
    }
    if (idx->last_path == 3 && idx->last_flags && idx->last_flags_n > 0) {
        std::vector<uint32_t> h((size_t)idx->last_flags_n);
        VS_HIP(hipSetDevice(idx->device));
        VS_HIP(hipDeviceSynchronize());
        VS_HIP(hipMemcpy(h.data(), idx->last_flags, h.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t f : h) o->last_fallbacks += f ? 1 : 0;
    }
    o->aux_bytes = idx->bp_ready ? (int64_t)(idx->bp_dir.bytes + idx->bp_base.bytes + idx->bp_rec.bytes + idx->bp_strip.bytes) : 0;
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

ScanPlan plan_scan(const vs_index* idx, int B) {
    ScanPlan p;
    const int64_t min_rows = 512;
    int64_t rpc = std::max<int64_t>(min_rows, ceil_div64(idx->n_rows, idx->cu_count));
    p.rows_per_chunk = rpc;
    p.nchunk = (int)std::max<int64_t>(1, ceil_div64(idx->n_rows, rpc));
    p.grid = (int)std::min<int64_t>((int64_t)B * p.nchunk, idx->cu_count);
    return p;
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

// (the multi-query scan and its launches: mq_search.hip)

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
    DevBuf& keys = device_scratch(device, kScratchMergeKeys);                 // kept between calls: the sharded search merges once per batch
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
