// csr_internal.h -- what csr_index.hip (containers, CSR scans, the search dispatcher) and bp_search.hip (blocked postings: builder,
// walk launches, filter + refine) share.  Internal to the library; the public boundary is include/vsearch_hip.h.
#pragma once
#include "common.h"
#include "csr_scan.h"
#include "csr_scan_mq.h"
#include "bp_walk.h"

#include <algorithm>

namespace vs {

constexpr int kQT = 8;   // queries per pass of the multi-query scan (csr_scan_mq.h) and per tile of the filter walks

struct ScanPlan {
    int nchunk;
    int64_t rows_per_chunk;
    int grid;
};

// Row chunks for `units` concurrent scans (query tiles, or single queries on the Qt = 1 path).  Work items = units x
// chunks.  Every item pays a table / image build and top-k sorts, and -- more important -- workgroups that sweep the
// SAME rows at the same time for different units share the stream through L2 / Infinity Cache, so chunks are as few
// and as long as still fill the CUs: many units -> 1-2 chunks, one unit -> one chunk per CU.
inline int choose_chunks(const vs_index* idx, int units, int max_nchunk) {
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

inline int64_t csr_bytes_per_pass(const vs_index* idx) {
    const int64_t per_packet = 16 + (idx->store_dtype == VS_F32 ? 32 : idx->store_dtype == VS_F16 ? 16 : 0);
    return idx->n_packets * per_packet + (idx->n_rows + 1) * 4;
}

// LDS entries left for a tile's weights once the fixed tables of the multi-query scan are placed
inline int mq_lanes(const vs_index* idx) { return std::max(idx->lanes_per_row, 8); }
inline int mq_acc_rows(const vs_index* idx) {          // accumulator rows x copies (see S in csr_scan_topk_mq)
    const int g = mq_lanes(idx);
    return kScanWaves * (64 / g) * (g >= 32 ? 4 : (g >= 16 ? 2 : 1));
}
inline int mq_vals_cap(const vs_index* idx) {
    const size_t fixed = mq_fixed_lds_bytes<kQT>(idx->n_cols, mq_acc_rows(idx));
    const size_t total = 160 * 1024;
    if (fixed + 1024 > total) return 0;
    return (int)((total - fixed) / 4);
}

constexpr int kBpExactQT = 4;     // queries per tile of the fp64 walk (its accumulators are twice as wide as the filter walk's)
constexpr int kBpBinQT = 8;       // queries per tile of the binary index's filter walk

// value mode of the records: the index's own, or fp16 for the lossy filter copy of an fp32 index (bp_refine.h)
inline int bp_record_vm(const vs_index* idx) {
    if (idx->store_dtype == VS_NONE) return VM_BIN;
    return (idx->store_dtype == VS_F16 || idx->bp_quant) ? VM_F16 : VM_F32;
}

// ---- bp_search.hip ------------------------------------------------------------------------------------------------------------
bool bp_wanted(const vs_index* idx);
int bp_build(vs_index* idx, hipStream_t s);
int csr_prepare_impl(vs_index* idx, hipStream_t s);
int bp_choose_chunks(const vs_index* idx, int n_tiles, int64_t n_blocks, int plan_nchunk);
bool bp_filter_ok(const vs_index* idx, int k, int col0, const uint64_t* upper);
int bp_filter_search(vs_index* idx, const float* dq, int32_t B, int32_t k, int64_t id_offset, int64_t* d_ids, float* d_scores, const ScanPlan& plan,
                     hipStream_t s, bool* done, int32_t out_ld);
// the exact (fp64) walk over lossless records: kBpExactQT queries per tile
int bp_exact_walk(const vs_index* idx, const BpArgs& a, int grid, int ent_cap, hipStream_t s);


// ---- mq_search.hip ------------------------------------------------------------------------------------------------------------
int mq_search(vs_index* idx, const float* dq, int32_t B, int32_t k, int64_t id_offset, int64_t* d_ids, float* d_scores,
              const ScanPlan& plan, hipStream_t s, bool* done, int32_t out_ld, int32_t col0, uint64_t* upper);

}  // namespace vs
