// bp_search.hip -- blocked postings: the column-grouped second copy of a SparseIndex / BoTIndex (builder), the launches of its
// walks (bp_walk.h list walk, bp_quad.h quad walk, bp_bin.h bag-of-token walk) and the filter + refine search built on them.
//   reference: src/ir/retriever/index.py:88-94 (SparseIndex.search: q @ P^T, top k) -- same results, bit for bit
#include "csr_internal.h"
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
#include "bp_head.h"
#include "bp_bq.h"

#include <chrono>
#include <vector>

namespace vs {

// ---- blocked postings (bp_walk.h): second, column-grouped copy of the index for sparse queries ---------------------

bool bp_wanted(const vs_index* idx) {
    if (idx->bp_pref == 0 || idx->n_rows <= 0 || idx->n_packets <= 0) return false;
    if (((size_t)idx->n_cols + 1) * 4 + 4096 > 160 * 1024) return false;            // the builder keeps one counter per column in LDS
    if (idx->bp_pref == 1) return true;
    // pays off when the index is big enough for the per-block directory (4 (V + 1) bytes per block) to disappear
    if (idx->store_dtype == VS_NONE) return idx->n_rows >= 65536;
    return idx->n_rows >= 16384 && (double)idx->nnz / (double)idx->n_rows >= 256.0;
}

void bp_release(vs_index* idx) {
    idx->bp_dir.release(); idx->bp_base.release(); idx->bp_rec.release(); idx->bp_df.release(); idx->bp_vmax.release();
    idx->bp_hmap.release(); idx->bp_strip.release();
    idx->bp_n_head = 0;
    idx->bp_head_gemm = false;
    idx->bp_quad = false;
    idx->bp_bq = false;
    idx->bp_ready = false;
}

// valued index: QT queries per tile, blocks of <= 2048 documents (exact fp64 walk: QT = 4, filter walk: QT = 8);
// binary index: filter walk only, one lane per (short) list
#ifdef VS_EXPERIMENTAL_WALKS
constexpr int kFlRoundsF16 = 8, kFlRoundsF32 = 5;     // record loads in flight per lane (registers: 8 / 12 per record)
#endif
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
    } else if (idx->bp_bq) {
        // bag-of-token chunks (bp_bq.h): the fixed-point filter walk only
        if (AM != AM_FIX || a.upper || ent_cap > kBqEntCap) return fail(VS_EUNSUPPORTED, "bag-of-token chunks serve the filter walk only");
        const int bq_qt = bq_slots(idx->bp_rows);
        kern = bq_qt == 2 ? (a.timing ? bp_bq_topk<2, 1> : bp_bq_topk<2, 0>) : (a.timing ? bp_bq_topk<4, 1> : bp_bq_topk<4, 0>);
        lds = bq_lds_bytes(bq_qt);
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
    } else if (vm == VM_BIN && AM == AM_FIX && idx->bp_walk_pref != 0 && !a.upper && ent_cap <= kBpEntCap) {          // (records: postings_walk = 5, or no room for the chunks)
        // bag-of-token index: the walk with the next block's records prefetched across the barrier (bp_bin.h); postings_walk = 0: the list walk
        kern = bp_bin_topk<kBpRowsMaxBin>;
        lds = bp_bin_lds_bytes<kBpRowsMaxBin>(ent_cap);
    } else if (vm == VM_BIN) {
        if (AM != AM_FIX) return fail(VS_EUNSUPPORTED, "binary postings serve the filter walk only");
        // one lane per list; the option picks the records in flight per lane = the size of the chunks dealt to the waves (8: 512
        // entries, 13 chunks a block on the Wiki21M shape, 16.4 k q/s; 4: 25 chunks for 16 waves, 13.0 k)
        kern = idx->bp_lanes == 4 ? bp_walk_topk<VM_BIN, kBpBinQT, AM_FIX, 1, kBpRowsMaxBin, 4> : bp_walk_topk<VM_BIN, kBpBinQT, AM_FIX, 1, kBpRowsMaxBin, 8>;
        lds = bp_lds_bytes<kBpBinQT, AM_FIX, kBpRowsMaxBin>(ent_cap);
    } else if (AM == AM_FIX && a.n_head > 0 && a.head_out) {
        // head columns served by the head pre-pass (bp_head.h): the list walk adds its sums in the epilogue; no strip weights in LDS
        if constexpr (AM == AM_FIX) {
            if (vm == VM_F32) kern = idx->bp_lanes != 4 ? bp_walk_topk<VM_F32, QT, AM_FIX, 8, kBpRowsMax, kBpNB, 2> : bp_walk_topk<VM_F32, QT, AM_FIX, 4, kBpRowsMax, kBpNB, 2>;
            else kern = idx->bp_lanes != 4 ? bp_walk_topk<VM_F16, QT, AM_FIX, 8, kBpRowsMax, kBpNBWide, 2> : bp_walk_topk<VM_F16, QT, AM_FIX, 4, kBpRowsMax, kBpNB, 2>;
        }
        lds = bp_lds_bytes<QT, AM, kBpRowsMax>(ent_cap, 0);
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
    // Size gate (profiles/r05_quad_policy.txt): a block's main chunks cost n_cols x 256 bytes whatever the lists hold.
    // Measured at 4 M docs, 64 .. 768 nnz per document: the quad walk is FASTER than the record walk at every density (31.6 vs 39.7 ms at
    // 64 nnz: a chunk step per list against the list walk's per-list bookkeeping), so speed never argues against it -- memory does: at
    // 128 nnz the chunks are 4.8 x the CSR packets, 86 % of their cells empty.  With "postings_walk" = -1 (auto) quad chunks are chosen
    // when their main area stays within kQuadMaxRatio x the CSR bytes (>= ~205 nnz per document at V = 29 523); 4 forces them.
    const int64_t quad_blocks = ceil_div64(std::max<int64_t>(idx->n_rows, 1), kQuadRows);
    const double quad_main = (double)quad_blocks * (double)idx->n_cols * (double)kQuadChunkBytes;
    const double csr_bytes = (double)idx->n_packets * (16.0 + (idx->store_dtype == VS_F32 ? 32.0 : idx->store_dtype == VS_F16 ? 16.0 : 0.0)) + (double)(idx->n_rows + 1) * 4.0;
    const bool quad_dense_enough = idx->bp_walk_pref == 4 || quad_main <= kQuadMaxRatio * csr_bytes;
    const bool quad_pref = idx->store_dtype != VS_NONE && idx->bp_filter != 0 && idx->n_cols <= 32768 && (idx->bp_walk_pref == -1 || idx->bp_walk_pref == 4) && !idx->bp_no_quad &&
                           idx->bp_align_pref != 1 && idx->bp_arrange_pref != 1 && (idx->store_dtype == VS_F16 || idx->bp_quant_pref != 0) && quad_dense_enough;
    // Bag-of-token chunks (bp_bq.h) -- the default copy of a binary index searched by filter + refine: every (block, column) list is a
    // direct-mapped 64-byte chunk (n_blocks x n_cols x 64 bytes: 6.5 GB at 21 M docs against 3.6 GB of packets).  "postings_walk" = 5
    // keeps the records of bp_bin.h, 0 the list walk; when the chunks do not fit HBM the records are built instead.
    const bool bq_pref = idx->store_dtype == VS_NONE && idx->bp_filter != 0 && idx->n_cols <= 32768 && (idx->bp_walk_pref == -1 || idx->bp_walk_pref == 6) && !idx->bp_no_quad;
    idx->bp_no_quad = false;
    auto auto_rows = [&]() -> int {
        const double avg_nnz = idx->n_rows > 0 ? (double)idx->nnz / (double)idx->n_rows : 1.0;
        // (chunks: blocks of as many documents as keep kBqFill postings a list on average -- whole thousands up to kBqRowsMax, quarters
        //  of a thousand below 1024; bp_bq.h)
        if (bq_pref) {
            const int r = (int)std::min(65536.0, (double)kBqFill * (double)idx->n_cols / std::max(avg_nnz, 1.0));
            return std::max(256, std::min(r >= 1024 ? r / 1024 * 1024 : r / 256 * 256, kBqRowsMax));
        }
        if (idx->store_dtype == VS_NONE) return kBpRowsMaxBin;
        const double avg = idx->n_rows > 0 ? (double)idx->nnz / (double)idx->n_rows : 1.0;
        // (quad chunks: at 50 postings a list 1 list in 40 goes on in an overflow chunk; 2048 documents per block measured the same)
        const int r = (int)(50.0 * (double)idx->n_cols / std::max(avg, 1.0)) / 128 * 128;
        return std::max(256, std::min(r, kBpRowsMax));
    };
    idx->bp_rows = idx->bp_rows_pref > 0 ? std::min(idx->bp_rows_pref, bq_pref ? kBqRowsMax : idx->store_dtype == VS_NONE ? kBpRowsMaxBin : kBpRowsMax)
                   : idx->bp_rows_forced > 0 ? idx->bp_rows_forced : auto_rows();
    idx->bp_rows_forced = 0;
    const int64_t n_blocks = ceil_div64(idx->n_rows, idx->bp_rows);
    const int V = idx->n_cols;
    const int RS0 = bp_rec_bytes(idx->store_dtype == VS_F32 ? VM_F16 : idx->store_dtype == VS_F16 ? VM_F16 : VM_BIN);   // smallest record this index can get
    const size_t b_dir = (size_t)n_blocks * ((size_t)V + 1) * 4;
    size_t free_b = 0, total_b = 0;
    // (the head pre-pass's scratch of this device -- tens of GB, kept between searches -- is given back before a copy is sized: the next
    //  search on a head-column index takes it again, within what is then free)
    if (device_scratch(idx->device, kScratchHeadOut).bytes) { device_scratch(idx->device, kScratchHeadOut).release(); device_scratch(idx->device, kScratchHeadW).release(); }
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
    const bool bq = bq_pref;
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
        if (free_b < b_dir + b_main + margin) {
            // no room for the chunks: the record copy is smaller (4 bytes a posting against 256 a list) -- try that before giving
            // up to the CSR scan
            fprintf(stderr, "[vsearch_hip] quad chunks need %.1f GB, %.1f GB of HBM free: building the record copy instead\n", (double)(b_dir + b_main) / 1e9, (double)free_b / 1e9);
            idx->bp_no_quad = true;
            return bp_build(idx, s);
        }
        VS_HIP(hipFuncSetAttribute((const void*)(quad_count_kernel<kQuadCells, kQuadLinked>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((quad_count_kernel<kQuadCells, kQuadLinked>), dim3(grid), dim3(kScanThreads), lds, s, idx->pk_ptr.as<uint32_t>(), idx->cols.as<uint4>(), idx->n_rows, V, idx->bp_rows,
                           idx->bp_dir.as<uint32_t>(), block_recs.as<uint32_t>(), df_rec, df_nnz, ovf.as<int32_t>());
    } else if (bq) {
        const size_t b_main = (size_t)n_blocks * V * kBqChunkBytes;
        if (free_b < b_dir + b_main + margin) {
            fprintf(stderr, "[vsearch_hip] bag-of-token chunks need %.1f GB, %.1f GB of HBM free: building the record copy instead\n", (double)(b_dir + b_main) / 1e9, (double)free_b / 1e9);
            idx->bp_no_quad = true;
            return bp_build(idx, s);
        }
        VS_HIP(hipFuncSetAttribute((const void*)(quad_count_kernel<kBqCells, kBqLinked>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((quad_count_kernel<kBqCells, kBqLinked>), dim3(grid), dim3(kScanThreads), lds, s, idx->pk_ptr.as<uint32_t>(), idx->cols.as<uint4>(), idx->n_rows, V, idx->bp_rows,
                           idx->bp_dir.as<uint32_t>(), block_recs.as<uint32_t>(), df_rec, df_nnz, ovf.as<int32_t>());
    } else {
        VS_HIP(hipFuncSetAttribute((const void*)bp_count_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(bp_count_kernel<0>, dim3(grid), dim3(kScanThreads), lds, s, idx->pk_ptr.as<uint32_t>(), idx->cols.as<uint4>(), idx->n_rows, V, idx->bp_rows,
                           idx->bp_dir.as<uint32_t>(), block_recs.as<uint32_t>(), df_rec, df_nnz, (const uint16_t*)nullptr, idx->bp_al_shift, ovf.as<int32_t>());
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
        // head pre-pass (bp_head.h, default): columns in >= 1/8 of the documents, up to 1024 -- from 1 M documents on, where the HBM has
        // room: >= 1/16, up to 1536 (the proof's slack grows with the columns: on a few thousand documents the wider head costs
        // fallbacks, tests/test_gpu_filter.py::test_head_strips_across_value_ranges); multiplied inside the walk: >= 1/4, up to 512.  (Round 6: with the product through the LDS ring a head column costs ~ 22 us per 1024 queries at 21 M
        // docs, and the densest lists of the tail more: Zipf 21 M docs, columns 1024 / 1280 / 1536 / 1792 / 2048: 157.7 / 154.2 / 152.7 /
        // 151.3 / 150.9 ms per search, 9 GB of strips per 256 columns.)
        static const int gemm_env = getenv("VS_BP_HEAD_GEMM") ? atoi(getenv("VS_BP_HEAD_GEMM")) : -2;   // (developer override of "postings_head_gemm")
        if (gemm_env > -2) idx->bp_head_gemm_pref = gemm_env;
        const bool gemm = idx->bp_head_gemm_pref != 0 && kQT == 8;
        static const int cap_env = getenv("VS_BP_HEAD_CAP") ? atoi(getenv("VS_BP_HEAD_CAP")) : 0;      // (developer: another cap of the pre-pass's columns, <= 2048)
        // the larger cap where its strips, the tail's records (at most 4 bytes a posting) and a pass's scratch (48 GB at most) all fit
        VS_HIP(hipMemGetInfo(&free_b, &total_b));
        const size_t strips_hi = (size_t)n_blocks * kBpHeadCapGemm * idx->bp_rows * 2;
        const bool wide = idx->n_rows >= (1 << 20) && free_b >= strips_hi + (size_t)idx->nnz * 4 + ((size_t)48 << 30) + margin;
        const int cap_gemm = cap_env > 0 ? std::min(cap_env, 2048) : wide ? kBpHeadCapGemm : kBpHeadCapGemmLow;
        hipLaunchKernelGGL(bp_head_select_kernel<0>, dim3(1), dim3(kScanThreads), 0, s, df_nnz, V,
                           (unsigned long long)ceil_div64(idx->n_rows, idx->bp_head_pref > 0 ? idx->bp_head_pref : (gemm ? (wide ? 16 : 8) : 4)), gemm ? cap_gemm : kBpHeadCap,
                           idx->bp_hmap.as<uint16_t>(), nh.as<int32_t>());
        VS_HIP(hipGetLastError());
        int32_t h_n = 0;
        VS_HIP(hipMemcpyAsync(&h_n, nh.p, 4, hipMemcpyDeviceToHost, s));
        VS_HIP(hipStreamSynchronize(s));
        idx->bp_n_head = h_n;
        idx->bp_head_gemm = gemm && h_n > 0;
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
                               idx->bp_al_shift, ovf.as<int32_t>());
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
    if (bq && (idx->bp_max_block_recs > 65535 || idx->bp_max_block_recs - V > 32767)) {
        // (a chunk index is 16 bits of a descriptor, a link's payload 15: a block of extremely long lists keeps the records)
        idx->bp_no_quad = true;
        return bp_build(idx, s);
    }
    const int RS = quad ? kQuadChunkBytes : bq ? kBqChunkBytes : bp_rec_bytes(bp_record_vm(idx));
    const size_t b_rec = ((size_t)n_rec + 2) * RS;                       // + one record: a lane past the last list's end re-reads "the record at the end"
    VS_HIP(hipMemGetInfo(&free_b, &total_b));
    if (free_b < b_rec + margin || idx->bp_rec.alloc(b_rec) != VS_OK) {
        if (quad || bq) {                                                       // (overflow chunks took it past what is free: the record copy next)
            (void)hipGetLastError();
            idx->bp_no_quad = true;
            return bp_build(idx, s);
        }
        return no_room(b_rec);
    }
    idx->bp_records = (int64_t)n_rec;
    if (!bq) VS_HIP(hipMemsetAsync(idx->bp_rec.p, 0, b_rec, s));         // pad postings: document 0, value 0 (bag-of-token chunks: the fill writes the pads)
    if (bq) {
        VS_HIP(hipFuncSetAttribute((const void*)bq_fill_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(bq_fill_kernel<0>, dim3(grid), dim3(kScanThreads), lds, s, idx->pk_ptr.as<uint32_t>(), idx->cols.as<uint4>(), idx->n_rows, V, idx->bp_rows,
                           (uint32_t)bq_rmax(bq_slots(idx->bp_rows)), idx->bp_dir.as<uint32_t>(), idx->bp_base.as<unsigned long long>(), idx->bp_rec.as<uint16_t>());
        VS_HIP(hipGetLastError());
        VS_STAGE("bq_fill", s);
        {   // the longest row: bounds a document's sum in the packed walk (bp_bq.h)
            DevBuf mr;
            VS_TRY(mr.alloc(4));
            VS_HIP(hipMemsetAsync(mr.p, 0, 4, s));
            hipLaunchKernelGGL(bq_maxrow_kernel<0>, dim3((unsigned)std::min<int64_t>(ceil_div64(idx->n_rows, 256), (int64_t)idx->cu_count * 8)), dim3(256), 0, s,
                               idx->pk_ptr.as<uint32_t>(), idx->n_rows, mr.as<uint32_t>());
            VS_HIP(hipGetLastError());
            uint32_t h_mr = 0;
            VS_HIP(hipMemcpyAsync(&h_mr, mr.p, 4, hipMemcpyDeviceToHost, s));
            VS_HIP(hipStreamSynchronize(s));
            idx->bp_bq_maxrow = (int)std::min<uint32_t>(h_mr, 1u << 30);
        }
        // (the two spare records behind the array: pads too -- a lane never reads them, the allocation's tail is simply initialised)
        VS_HIP(hipMemsetAsync(idx->bp_rec.as<char>() + (size_t)n_rec * RS, 0, 2 * (size_t)RS, s));
        idx->bp_bq = true;
        VS_HIP(hipStreamSynchronize(s));
        idx->bp_dir.release();                                           // (the chunks link to their overflow themselves)
    } else if (quad) {
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
        if (!idx->bp_quad && !idx->bp_bq) (void)hipMemcpy(hd.data(), idx->bp_dir.as<uint32_t>() + (size_t)(n_blocks - 1) * (V + 1), hd.size() * 4, hipMemcpyDeviceToHost);   // (quad chunks: the directory is gone)
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
//   head pre-pass: the sums handed to the walk as uint16 in units of 2^14 (truncated)                    -> 2^14
int32_t bp_head_slack(const vs_index* idx) {
    if (idx->bp_n_head <= 0) return 0;
    const int hp = bp_head_pad(idx->bp_n_head);
    // (head pre-pass: its sums reach the walk truncated to units of 2^kHeadOutShift)
    return 256 + 4 * hp + (int32_t)ceilf(65536.f / idx->bp_vmax_f) + 33 * 128 * (hp / 32) + 66 + (idx->bp_head_gemm ? (1 << kHeadOutShift) : 0);
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
        // (... with the strips multiplied inside the walk.  Behind the head pre-pass the tiles are even again -- what differed was their
        //  share of head columns: 126 ms at one item per CU against 138 at two, 21 M docs x 1023 head columns)
        const int per_cu = idx->bp_n_head > 0 && !idx->bp_head_gemm ? 2 : 1;
        const int c0 = (int)std::max<int64_t>(1, ceil_div64(per_cu * (int64_t)idx->cu_count, n_tiles));
        for (int c = c0; c <= c0 + 3; ++c) {
            const int64_t it = (int64_t)n_tiles * c;
            const double eff = (double)it / (double)(ceil_div64(it, idx->cu_count) * idx->cu_count);
            if (eff > best_eff + 1e-9) { best_eff = eff; best = c; }
            if (eff >= 0.95) break;
        }
        nchunk = (int)std::min<int64_t>(best, n_blocks);
        // Quad chunks: a power of two (work items go to the XCDs round robin -- item % 8 -- and take chunk item % nchunk: a count that does
        // not divide 8 scatters every chunk over all XCDs: 3 chunks 225 ms, 5: 231, 6: 176 at 21 M docs x 1024 queries), and the SMALLEST
        // that fills the CUs: one item per CU.  Round 5 ran four chunks (two items per CU, a block swept by two XCDs' L2s instead of
        // four: 134.3 ms against 141.0 with two) -- with round 6's epilogue (uniform cut decision, cuts by selection) the order is
        // the other way round, in every fresh process: 21 M docs, B = 1024: 2 chunks 129.9 - 130.0 ms / 4: 134.5; B = 2048: 1 chunk
        // 258.7 / 2: 260.0 / 4: 269.3; B = 512: 4 chunks 67.4 / 8: 73.5 (2: half the CUs idle); B = 256: 8 chunks 36.6 / 16: 43.1;
        // 2.6 M docs (one rank's shard of an 8-GPU step), B = 1024: 2 chunks 16.73 ms / 4: 17.53 / 8: 19.45 -- every item pays its
        // start-up (entry sort, the candidate flood until its thresholds rise, the final cuts) once (profiles/r06_chunks.txt).
        if (idx->bp_quad) {
            int pick = 1;
            double pick_eff = 0.0;
            for (int c = 1; c <= 64; c *= 2) {
                const int64_t it = (int64_t)n_tiles * c;
                const double eff = (double)it / (double)(ceil_div64(it, idx->cu_count) * idx->cu_count);
                if (eff > pick_eff + 1e-9) { pick_eff = eff; pick = c; }
                if (eff >= 0.9) break;
            }
            nchunk = (int)std::min<int64_t>(pick, n_blocks);
        }
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
    // bag-of-token chunks in blocks of more than 4096 documents (two sum planes): tiles of FOUR queries -- the packed walk takes those whose
    // queries all qualify for 16-bit sums, the others are cut into two-slot tiles for the int32 walk (bq_split_kernel, bp_bq.h)
    const bool bq_pk = idx->bp_bq && bq_slots(idx->bp_rows) == 2 && idx->bp_packed_pref != 0 && idx->bp_bq_maxrow > 0;
    const int qt = bq_pk ? 4 : idx->bp_bq ? bq_slots(idx->bp_rows) : qt_env > 0 ? std::min(qt_env, kQT) : idx->store_dtype == VS_NONE ? kBpBinQT : (duo ? kDuoQT : kQT);
    // (dense strips: their weight matrix takes 16 KB of the LDS the entries would use)
    const int vals_cap = std::min(mq_vals_cap(idx), idx->bp_bq ? kBqEntCap : duo ? kDuoEntCap : (idx->bp_n_head > 0 ? kBpEntCap - 512 : kBpEntCap));
    const int64_t qcap = (int64_t)B * vals_cap;                               // bound of the batch's (query, column) entries that enter a tile
    const int64_t n_blocks = ceil_div64(idx->n_rows, idx->bp_rows);
    if (idx->bp_rows > (idx->bp_bq ? kBqRowsMax : idx->store_dtype == VS_NONE ? kBpRowsMaxBin : kBpRowsMax)) return fail(VS_EINVAL, "postings_rows beyond the walk's block capacity");
    // scratch: per-query metadata (counts, qptr, plan, tiles, flags, scale, slack, weight sums), column frequencies, the sparse batch
    const size_t off_counts = 0, off_qptr = off_counts + (size_t)B * 8, off_plan = off_qptr + (size_t)(B + 1) * 8, off_tiles = off_plan + 64,
                 off_fb = off_tiles + (size_t)B * 8, off_scale = off_fb + (size_t)B * 8, off_slack = off_scale + (size_t)B * 4,
                 off_wsum = off_slack + (size_t)B * 4, off_flags = off_wsum + (size_t)B * 4, off_nfb = off_flags + (size_t)B * 4,
                 off_t16 = off_nfb + 64, off_t2 = off_t16 + (size_t)B * 8, off_f16 = off_t2 + (size_t)B * 8, off_nsp = off_f16 + (size_t)B * 4,
                 off_gtau = (off_nsp + 64 + 15) & ~(size_t)15, off_freq = (off_gtau + (size_t)B * 8 + 15) & ~(size_t)15;
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
    int2* tiles16 = (int2*)(meta + off_t16);
    int2* tiles2 = (int2*)(meta + off_t2);
    uint32_t* flag16 = (uint32_t*)(meta + off_f16);
    int32_t* n_split = (int32_t*)(meta + off_nsp);
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
                       idx->bp_n_head > 0 ? (const uint16_t*)idx->bp_hmap.as<uint16_t>() : (const uint16_t*)nullptr, bp_head_slack(idx),
                       bq_pk ? idx->bp_bq_maxrow : 0, bq_pk ? flag16 : (uint32_t*)nullptr);
    if (bq_pk) hipLaunchKernelGGL(bq_split_kernel<0>, dim3(1), dim3(kScanThreads), 0, s, (const int2*)tiles, reinterpret_cast<const int32_t*>(dplan), (const uint32_t*)flag16,
                                  tiles16, tiles2, n_split);
    VS_HIP(hipGetLastError());
    VS_STAGE("sparsify", s);
    // 2. the walk.  Work items = (tile, chunk); the tile count lives on the device, the chunks follow its lower bound ceil(B / qt)
    // Head pre-pass (bp_head.h): the walk runs in PASSES over ranges of tiles, each behind the matrix product that writes the dense part
    // of its tiles' sums (64 KB per tile and block) to a scratch array -- as many tiles per pass as the scratch HBM has room for.
    const bool head_gemm = idx->bp_n_head > 0 && idx->bp_head_gemm;
    int tiles_per_pass = 0, n_pass = 1;
    const int head_ks = bp_head_pad(idx->bp_n_head) / 32;
    if (head_gemm) {
        const size_t per_tile = (size_t)n_blocks * (size_t)idx->bp_rows * 8 * 2;             // uint16 sums (bp_head.h)
        size_t free_b = 0, total_b = 0;
        VS_HIP(hipMemGetInfo(&free_b, &total_b));
        // ONE scratch per device, shared by every index on it (per index and sized from "what is free" the first search on a
        // head-column index took most of the free HBM for good): at most a quarter of the device's memory, 48 GB, and what is free minus a
        // margin; bp_build gives it back when a postings copy needs the room (head_scratch_release)
        DevBuf& head_out = device_scratch(idx->device, kScratchHeadOut);
        DevBuf& head_w = device_scratch(idx->device, kScratchHeadW);
        const size_t have = head_out.bytes, margin = (size_t)6 << 30;
        const size_t room = std::min<size_t>(free_b + have > margin ? free_b + have - margin : 0, total_b / 4);
        static const int tpp_env = getenv("VS_HEAD_TILES") ? atoi(getenv("VS_HEAD_TILES")) : 0;      // (developer: tiles per pass)
        int tpp = (int)std::min<size_t>(std::max<size_t>(have, std::min<size_t>(room, (size_t)48 << 30)) / std::max<size_t>(per_tile, 1), (size_t)ceil_div(B, qt));
        if (tpp_env > 0) tpp = std::min(tpp, tpp_env);
        if (idx->bp_head_tiles > 0) tpp = std::min(tpp, idx->bp_head_tiles);
        if (tpp >= 64) tpp = tpp / 64 * 64;                                   // whole groups of the product's workgroups (4 tile waves x 16 tiles)
        else if (tpp >= 2) tpp &= ~1;                                          // (a weight operand holds two tiles: passes start on even tiles)
        if (tpp < 1) return fail(VS_ENOMEM, "head pre-pass: no HBM for the dense sums of one tile (%.2f GB)", (double)per_tile / 1e9);
        tiles_per_pass = tpp;
        VS_TRY(head_out.reserve((size_t)tpp * per_tile));
        VS_TRY(head_w.reserve((size_t)((tpp + 1) / 2) * head_ks * 1024));
        n_pass = ceil_div(B, tpp);                                           // (a tile holds >= 1 query: passes beyond the batch's tiles return at once)
    }
    const int n_tiles_est = head_gemm ? std::min(tiles_per_pass, ceil_div(B, qt)) : ceil_div(B, qt);
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
    idx->last_split_dev = bq_pk ? n_split : nullptr;
    idx->last_plan_rs = idx->bp_quad ? kQuadChunkBytes : idx->bp_bq ? kBqChunkBytes : bp_rec_bytes(bp_record_vm(idx));
    idx->last_plan_blocks = n_blocks;
    // lock-step window of the walk's work items (all walks; kernels ignore it when not every item is resident)
    static const int pace_env = getenv("VS_BP_PACE") ? atoi(getenv("VS_BP_PACE")) : -1;
    // (off by default for the list walk: it costs it 10 %; the bag-of-token walk of bp_bin.h runs ahead of its memory and
    //  NEEDS it: free running 81 ms, window 16: 66.7, 32: 56.5, 48: 57.2, 64: 58.7, 128: 73 -- the list walk: 61.3)
    const bool bin_walk = idx->store_dtype == VS_NONE && idx->bp_walk_pref != 0 && !idx->bp_bq;
    // (the two-set walk has no block barrier to keep its workgroups at one pace: 4 M docs 56.9 ms free running, window 1: 39.8, 2: 37.9, 4: 42.2)
    const int pace_w = pace_env >= 0 ? pace_env : (idx->bp_pace >= 0 ? idx->bp_pace : (idx->bp_bq ? kBqPaceDefault : bin_walk ? 32 : (duo ? 2 : (idx->bp_quad ? kQuadPaceDefault : 0))));
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
    for (int pass = 0; pass < n_pass; ++pass) {
        if (head_gemm) {
            HeadArgs h{};
            h.strip = a.strip;
            h.wt = device_scratch(idx->device, kScratchHeadW).as<uint4>();
            h.out = device_scratch(idx->device, kScratchHeadOut).as<uint16_t>();
            h.tiles = tiles;
            h.n_tiles_dev = a.n_tiles_dev;
            h.tile0 = pass * tiles_per_pass;
            h.tile_cnt = tiles_per_pass;
            h.qptr = qptr; h.qcols = qcols; h.qvals = qvals; h.qscale = qscale;
            h.hmap = a.hmap;
            h.n_head = a.n_head;
            h.rows = idx->bp_rows;
            h.n_rows = idx->n_rows;
            h.n_blocks = n_blocks;
            h.head_pre = a.head_pre; h.head_mul = ldexpf(a.head_mul, -kHeadOutShift);      // (the product leaves its sums in units of 2^14)
            a.head_out = h.out;
            a.tile0 = h.tile0;
            a.tile_cnt = tiles_per_pass;
            if (a.pace && pass > 0) VS_HIP(hipMemsetAsync(idx->ws_pace.p, 0, (size_t)nchunk * a.blocks_per_chunk * 4, s));
            ProfScope prof("head_gemm", s);
            hipLaunchKernelGGL(head_weights_kernel<0>, dim3(tiles_per_pass), dim3(256), 0, s, h);
            // (developer: waves of a workgroup, documents x tiles; 1xx = wide waves.  21 M docs x 1023 head columns x 1024 queries: 114: 56.4 ms,
            //  24: 76.5, 122: ~ 60 -- the product is bound by operand traffic, and a wide wave moves a third fewer bytes per MFMA)
            //  Round 6: 2 x 2 wide waves that share their operands through an LDS ring, four k-steps of LDS-DMA in flight, a document run's
            //  tile groups side by side on one XCD (head_gemm_lds_kernel): 51.9 -> 37.1 ms.  A pass of fewer than 32 tiles leaves half of
            //  such a workgroup idle: those go to the direct kernel.)
            static const int shape_env = getenv("VS_HEAD_SHAPE") ? atoi(getenv("VS_HEAD_SHAPE")) : 0;
            const bool ring = shape_env == 222 || (shape_env == 0 && (idx->bp_head_product == 1 || (idx->bp_head_product < 0 && tiles_per_pass >= 32 && idx->cu_count % 8 == 0)));
            if (ring) {
                VS_HIP(hipFuncSetAttribute((const void*)head_gemm_lds_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
                hipLaunchKernelGGL(head_gemm_lds_kernel<0>, dim3(idx->cu_count), dim3(256), 163840, s, h);
            } else if (shape_env == 24) hipLaunchKernelGGL((head_gemm_kernel<2, 4>), dim3(idx->cu_count), dim3(512), 0, s, h);
            else hipLaunchKernelGGL((head_gemm_kernel<1, 4, 1>), dim3(idx->cu_count), dim3(256), 0, s, h);
            VS_HIP(hipGetLastError());
        }
        ProfScope prof("csr_scan_topk", s);
        if (bq_pk) {
            // the packed walk on the tiles that qualify, the int32 walk on the rest (cut to two slots); either returns at once without tiles
            BpArgs a16 = a, a2 = a;
            a16.tiles = tiles16; a16.n_tiles_dev = n_split;
            a2.tiles = tiles2; a2.n_tiles_dev = n_split + 1;
            void (*k16)(BpArgs) = a.timing ? bp_bq_topk<4, 1, 1> : bp_bq_topk<4, 0, 1>;
            void (*k2)(BpArgs) = a.timing ? bp_bq_topk<2, 1> : bp_bq_topk<2, 0>;
            const size_t lds = bq_lds_bytes(2);
            if (vals_cap > kBqEntCap) return fail(VS_EUNSUPPORTED, "bag-of-token chunks: tile entries beyond the table");
            VS_HIP(hipFuncSetAttribute((const void*)k16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            VS_HIP(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k16, dim3(grid), dim3(kScanThreads), lds, s, a16);
            hipLaunchKernelGGL(k2, dim3(grid), dim3(kScanThreads), lds, s, a2);
            VS_HIP(hipGetLastError());
        } else
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
    if (head_gemm) r.quant = 2;                                     // ... and the head pre-pass's weights rounded to one fp16 number each (bp_head.h)
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

int bp_exact_walk(const vs_index* idx, const BpArgs& a, int grid, int ent_cap, hipStream_t s) { return launch_bp_walk<kBpExactQT, AM_F64>(idx, a, grid, ent_cap, s); }

}  // namespace vs
