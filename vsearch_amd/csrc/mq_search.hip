// mq_search.hip -- the multi-query pass of SparseIndex.search: kQT queries per tile over the CSR packets (csr_scan_mq.h), or -- when the
// index keeps lossless blocked postings -- the fp64 walk (bp_walk.h); the filter + refine path (bp_search.hip) is tried first.
//   reference: src/ir/retriever/index.py:88-94
#include "csr_internal.h"

namespace vs {
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

}  // namespace

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
        VS_TRY(bp_exact_walk(idx, a, grid, vals_cap, s));
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

}  // namespace vs
