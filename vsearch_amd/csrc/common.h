// common.h -- host-side plumbing shared by the translation units of libvsearch_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "vsearch_hip.h"

namespace vs {

// ---- error slot (thread-local; the only global mutable state besides the profiler) ------------
inline char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define VS_HIP(call)                                                                                  \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess)                                                                        \
            return vs::fail(e__ == hipErrorOutOfMemory ? VS_ENOMEM : VS_EHIP, "%s failed: %s (%s:%d)", \
                            #call, hipGetErrorString(e__), __FILE__, __LINE__);                       \
    } while (0)

#define VS_TRY(expr)                 \
    do {                             \
        int rc__ = (expr);           \
        if (rc__ != VS_OK) return rc__; \
    } while (0)

inline size_t dtype_size(int dt) {
    switch (dt) {
        case VS_F32: case VS_I32: return 4;
        case VS_F16: case VS_U16: return 2;
        case VS_I64: return 8;
        case VS_U8: return 1;
        default: return 0;
    }
}

// true if p is a device (or managed) pointer; host pointers unknown to HIP report false.
inline bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // clear sticky "invalid value" for plain host memory
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// RAII device buffer
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    int alloc(size_t n) {
        release();
        if (n == 0) return VS_OK;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(VS_ENOMEM, "hipMalloc(%zu bytes) failed: %s", n, hipGetErrorString(e));
        }
        bytes = n;
        return VS_OK;
    }
    int reserve(size_t n) { return n <= bytes ? VS_OK : alloc(n); }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Small per-device scratch buffers that outlive a call (never freed: hipFree synchronises the device, and a
// malloc/free pair per call costs more than the sparsify kernels themselves).  The library is driven by one host
// thread; a slot is owned by ONE entry point:
enum : int {
    kScratchFlags = 0,      // sparsify.hip: per-row flags of the mask stage
    kScratchCounts = 1,     // sparsify.hip: per-row non-zero counts
    kScratchRowPtr = 2,     // sparsify.hip: vs_dense_to_csr row pointers
    kScratchMergeKeys = 3,  // csr_index.hip: vs_merge_topk keys (kept between the calls of a sharded search)
    kScratchHeadKeys = 4,   // dense.hip: keys of the fused encoder head
    kScratchCsrSlots = 5,   // sparsify.hip: vs_embed_mask_to_csr slot runs
    kScratchHeadOut = 6,    // bp_search.hip: the head pre-pass's dense sums -- ONE per device, shared by every index on it (it can be tens of GB)
    kScratchHeadW = 7,      // bp_search.hip: the head pre-pass's weight operands
};
inline DevBuf& device_scratch(int device, int slot) {
    static DevBuf* pool = new DevBuf[16 * 8];
    return pool[(device & 15) * 8 + (slot & 7)];
}

// Copies `bytes` from src (host or device) into a device staging buffer if needed and returns a
// device pointer usable on `stream`.
inline int to_device(const void* src, size_t bytes, DevBuf& stage, hipStream_t stream, const void** out) {
    if (is_device_ptr(src)) {
        *out = src;
        return VS_OK;
    }
    VS_TRY(stage.reserve(bytes));
    VS_HIP(hipMemcpyAsync(stage.p, src, bytes, hipMemcpyHostToDevice, stream));
    *out = stage.p;
    return VS_OK;
}

// ---- profiler: hipEvent pairs around named kernel launches --------------------------------------
struct Profiler {
    struct Pending { std::string name; hipEvent_t a, b; };
    struct Acc { double ms = 0; int64_t n = 0; };
    bool on = false;
    std::mutex mu;
    std::vector<Pending> pending;
    std::map<std::string, Acc> acc;
    static Profiler& get() {
        static Profiler p;
        return p;
    }
    void begin(const char* name, hipStream_t s) {
        if (!on) return;
        Pending p;
        p.name = name;
        // (no system-scope fence when an event completes: the pair times kernels, nothing on the host reads their output through it --
        //  with the default flags the fence's cache write-back is inside the interval: ~ 15 us behind a kernel that wrote 100 MB)
        (void)hipEventCreateWithFlags(&p.a, hipEventDisableSystemFence);
        (void)hipEventCreateWithFlags(&p.b, hipEventDisableSystemFence);
        (void)hipEventRecord(p.a, s);
        std::lock_guard<std::mutex> g(mu);
        pending.push_back(p);
    }
    void end(hipStream_t s) {
        if (!on) return;
        std::lock_guard<std::mutex> g(mu);
        if (!pending.empty()) (void)hipEventRecord(pending.back().b, s);
    }
    void drain() {
        std::lock_guard<std::mutex> g(mu);
        for (auto& p : pending) {
            float ms = 0;
            if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
                acc[p.name].ms += ms;
                acc[p.name].n += 1;
            }
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
        pending.clear();
    }
};

struct ProfScope {
    hipStream_t s;
    ProfScope(const char* name, hipStream_t st) : s(st) { Profiler::get().begin(name, st); }
    ~ProfScope() { Profiler::get().end(s); }
};

// VS_DEBUG_SYNC=1: synchronise and report after each stage (finding the kernel behind a device fault)
inline bool debug_sync_on() {
    static const bool on = getenv("VS_DEBUG_SYNC") != nullptr;
    return on;
}
#define VS_STAGE(name, stream)                                                                        \
    do {                                                                                              \
        if (vs::debug_sync_on()) {                                                                    \
            hipError_t e__ = hipStreamSynchronize(stream);                                            \
            fprintf(stderr, "[vsearch_hip] stage %s: %s\n", name, hipGetErrorString(e__));            \
            fflush(stderr);                                                                           \
        }                                                                                             \
    } while (0)

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline uint32_t pow2_ceil(uint32_t x) {
    uint32_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

}  // namespace vs

// ---- the index handle ---------------------------------------------------------------------------
struct vs_index {
    int kind = VS_KIND_CSR;
    int device = 0;
    int store_dtype = VS_F32;
    int64_t n_rows = 0;
    int32_t n_cols = 0;
    int64_t nnz = 0;
    // CSR device format ("packets": 8 nnz, rows padded with column id n_cols whose query weight is 0)
    int64_t n_packets = 0;
    int64_t rows_cap = 0, packets_cap = 0;   // reserved capacity (vs_index_create_reserved / append)
    int lanes_per_row = 32;
    vs::DevBuf pk_ptr;   // uint32 [n_rows + 1]
    vs::DevBuf cols;     // uint16 [n_packets * 8]
    vs::DevBuf vals;     // fp32 / fp16 [n_packets * 8] (absent for binary)
    // blocked postings (bp_walk.h): column-grouped copy for sparse queries, built on first use when HBM allows
    vs::DevBuf bp_dir;   // uint32 [n_blocks, n_cols + 1]: first record of a column inside its block
    vs::DevBuf bp_base;  // uint64 [n_blocks + 1]: first record of a block
    vs::DevBuf bp_rec;   // records: 8 x uint16 document-in-block + 8 values (fp32 | fp16 | none)
    vs::DevBuf bp_df;    // uint64 [2][n_cols]: records / non-zeros per column over all blocks -- what a query entry walks
    int64_t bp_records = 0;
    vs::DevBuf bp_hmap;  // uint16 [n_cols]: strip index of a head column (dense strip), 0xFFFF otherwise; valid when bp_n_head > 0
    vs::DevBuf bp_strip; // fp16 [n_blocks][bp_n_head][bp_rows]: values of the head columns
    int bp_n_head = 0;
    float bp_vmax_f = 1.f;   // max |value| of the index (bp_build)
    int bp_head_gemm_pref = -1;   // option "postings_head_gemm": -1 auto (= 1), 1 = the head columns' part of the sums by the head pre-pass (bp_head.h: up to 1024 columns
                                  // present in >= 1/8 of the documents; from 1 M documents on, HBM permitting, 1536 in >= 1/16), 0 = multiplied inside the walk, tile by tile (up to 512 columns in >= 1/4)
    int bp_head_tiles = 0;        // option "postings_head_tiles": tiles per pass of the head pre-pass (0 = as many as the scratch HBM holds)
    int bp_head_product = -1;     // option "postings_head_product": the pre-pass's kernel: 0 = every wide wave loads its own operands, 1 = 2 x 2 waves share them
                                  // through an LDS ring (bp_head.h: head_gemm_lds_kernel), -1 auto (the ring from 256 queries a pass on)
    bool bp_head_gemm = false;    // this copy's strips are served by the head pre-pass
    int bp_head_pref = -1;   // option "postings_head": -1 auto (columns present in >= 1/4 of the documents, at most 512), 0 = none, N = share 1/N
    vs::DevBuf bp_vmax;  // [2] uint32: float bits of max |value| (bounds the fixed-point walk's products), any-value-negative flag
    int bp_lanes = 0;    // option "postings_lanes": lanes per posting list of a valued index (4 | 8, auto = 8); binary index: records in flight per lane (auto = 8)
    bool bp_force_fb = false;   // option "postings_force_fallback" (tests)
    int bp_rows_forced = 0;  // bp_build restarting itself with this block size (skewed corpus found): consumed by the next bp_build
    int bp_al_shift = 0;     // lists of the copy start on a multiple of 2^bp_al_shift records (bp_walk.h)
    int bp_align_pref = -1;  // option "postings_align": -1 auto (= 0), 0 = packed lists, 1 = lists start on whole 128-byte lines
    bool bp_quant = false;   // the records of this fp32 index hold fp16-rounded values (lossy filter copy, bp_refine.h)
    int bp_quant_pref = -1;  // option "postings_quant": -1 auto (on when the data allow it), 0 = keep fp32 values in the records
    int bp_filter = 1;   // option "postings_filter": 1 = int32 fixed-point walk + exact refine (default), 0 = fp64 walk only
    int64_t last_walk_postings = 0;      // postings (multiply-adds) the most recent search's walk visited
    const uint32_t* last_flags = nullptr; // device [last_flags_n]: queries of the most recent filter search that took the exact walk
    int last_flags_n = 0;
    const int32_t* last_split_dev = nullptr;   // device [2]: tiles of the most recent filter search on the packed / the int32 bag-of-token walk (bp_bq.h)
    const int64_t* last_plan_dev = nullptr;   // device plan of the most recent filter search ([2] entries, [4] records, [5] postings walked)
    int last_plan_rs = 0;
    int64_t last_plan_blocks = 0;
    int bp_rows = 2048;  // documents per block of the copy (picked at build time)
    bool bp_ready = false, bp_tried = false;
    int bp_state = 0;    // vs_index_info_t.postings_state: 0 not attempted, 1 ready, 2 no HBM room, 3 directory overflow, 4 not wanted (small index / option)
    int64_t last_scan_bytes = 0;   // bytes the scan kernels of the most recent search had to read (algorithmic, per path)
    int last_path = 0;             // 0 = one query per pass, 1 = 8-query CSR scan, 2 = blocked postings
    int bp_pref = -1;    // option "blocked_postings": -1 auto, 0 never, 1 always
    int bp_rows_pref = 0;// option "postings_rows": 0 = auto, else documents per block (multiple of 64, 256..2048); applies at the next build
    int bp_chunks = 0;   // option "postings_chunks": 0 = auto, else block runs per tile on the postings path
    int bp_walk_pref = -1;   // option "postings_walk": -1 auto (= 4 where it applies, else 0), 4 = quad chunks (bp_quad.h), 0 = one list per lane group (bp_walk.h), 1 = flat worklists (bp_flat.h), 2 = list walk on two accumulator sets (bp_duo.h), 3 = streamed flat walk (bp_stream.h)
    bool bp_quad = false;       // bp_rec holds quad chunks (bp_quad.h): 64-cell chunks of one-dword postings; dir / base count chunks
    int bp_bq_maxrow = 0;       // bag-of-token chunks: non-zeros of the longest row (an upper bound: packets x 8) -- what a document can match of a query at most
    int bp_packed_pref = -1;    // option "postings_packed": -1 / 1 = four query slots a tile on packed 16-bit sums where the batch allows it (bp_bq.h), 0 = two int32 slots
    bool bp_bq = false;         // bp_rec holds bag-of-token chunks (bp_bq.h): chunks of uint16 postings of a binary index; base counts chunks
    bool bp_no_quad = false;    // bp_build restarting itself without quad chunks (head columns found): consumed by the next bp_build
    int bp_arrange_pref = -1;   // option "postings_arrange": 1 = bank-aware order inside the lists (bp_arrange_kernel), -1 / 0 = as filled
    int bp_pace = -1;        // option "postings_pace": blocks a work item may run ahead of the slowest item of its chunk (-1 auto, 0 = free running)
    int64_t bp_max_block_recs = 0;   // records of the fullest block (the flat walk's items address 2^19)
    int mq_variant = -1; // option "mq_variant": -1 auto (from the batch's query overlap), 0 plain, 1 shared columns
    // dense
    vs::DevBuf mat;      // [n_rows, n_cols] store_dtype
    // scratch owned by the handle (grow-only)
    vs::DevBuf ws_q, ws_cand, ws_out_ids, ws_out_scores, ws_misc, ws_mq_meta, ws_mq_q, ws_mq_cand, ws_fb, ws_pace;
    bool logical_dense = false;   // dense Index stored as CSR packets (sparsity-aware dense index)
    int qt_pref = 0;     // 0 = auto (multi-query pass when the batch qualifies), 1 = force the dense-image pass
    int last_qt = 0;     // queries per pass of the most recent search
    int cu_count = 256;
    double load_GBps = 0.0;   // vs_index_load_native: file -> HBM rate of the load that created this handle
};
