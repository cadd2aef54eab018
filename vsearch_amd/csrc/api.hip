// api.hip -- library-level entry points and the kind dispatch of Index.search (index.py:88-94).
#include "common.h"

#include <string>

using namespace vs;

int vs_csr_search(vs_index*, const void*, int, int64_t, int32_t, int32_t, int64_t, int64_t*, float*, hipStream_t);
int vs_csr_prepare(vs_index*, hipStream_t);
int vs_csr_scores(vs_index*, const void*, int, int64_t, int32_t, float*, hipStream_t);
int vs_dense_search(vs_index*, const void*, int, int64_t, int32_t, int32_t, int64_t, int64_t*, float*, hipStream_t);
int vs_dense_scores(vs_index*, const void*, int, int64_t, int32_t, float*, hipStream_t);

extern "C" int vs_version(void) { return 100; }   // 0.1.0

extern "C" const char* vs_last_error(void) { return err_buf(); }

extern "C" int vs_device_count(int32_t* out) {
    if (!out) return fail(VS_EINVAL, "out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *out = n;
    return VS_OK;
}

static int check_search_args(vs_index* idx, const void* q, int64_t ldq, int32_t B) {
    if (!idx || !q) return fail(VS_EINVAL, "NULL argument");
    if (B <= 0) return fail(VS_EINVAL, "B must be positive");
    if (ldq < idx->n_cols) return fail(VS_EINVAL, "query has %lld columns, index has %d", (long long)ldq, idx->n_cols);
    return VS_OK;
}

extern "C" int vs_index_search(vs_index* idx, const void* q, int q_dtype, int64_t ldq, int32_t B, int32_t k,
                               int64_t id_offset, int64_t* out_ids, float* out_scores, void* stream) {
    VS_TRY(check_search_args(idx, q, ldq, B));
    if (!out_ids || !out_scores) return fail(VS_EINVAL, "NULL output");
    if (k <= 0) return fail(VS_EINVAL, "k must be positive");
    if (k > idx->n_rows) return fail(VS_ERANGE, "selected index k out of range (k = %d > %lld rows)", k, (long long)idx->n_rows);
    VS_HIP(hipSetDevice(idx->device));
    hipStream_t s = (hipStream_t)stream;
    int rc = idx->kind == VS_KIND_CSR ? vs_csr_search(idx, q, q_dtype, ldq, B, k, id_offset, out_ids, out_scores, s)
                                      : vs_dense_search(idx, q, q_dtype, ldq, B, k, id_offset, out_ids, out_scores, s);
    if (rc != VS_OK) return rc;
    if (!stream) VS_HIP(hipStreamSynchronize(s));
    if (Profiler::get().on) Profiler::get().drain();
    return VS_OK;
}

extern "C" int vs_index_prepare(vs_index* idx, void* stream) {
    if (!idx) return fail(VS_EINVAL, "NULL index");
    VS_HIP(hipSetDevice(idx->device));
    if (idx->kind == VS_KIND_CSR) VS_TRY(vs_csr_prepare(idx, (hipStream_t)stream));
    if (!stream) VS_HIP(hipStreamSynchronize((hipStream_t)stream));
    return VS_OK;
}

extern "C" int vs_index_scores(vs_index* idx, const void* q, int q_dtype, int64_t ldq, int32_t B, float* out_scores, void* stream) {
    VS_TRY(check_search_args(idx, q, ldq, B));
    if (!out_scores) return fail(VS_EINVAL, "NULL output");
    VS_HIP(hipSetDevice(idx->device));
    hipStream_t s = (hipStream_t)stream;
    int rc = idx->kind == VS_KIND_CSR ? vs_csr_scores(idx, q, q_dtype, ldq, B, out_scores, s)
                                      : vs_dense_scores(idx, q, q_dtype, ldq, B, out_scores, s);
    if (rc != VS_OK) return rc;
    if (!stream) VS_HIP(hipStreamSynchronize(s));
    if (Profiler::get().on) Profiler::get().drain();
    return VS_OK;
}

extern "C" int vs_index_set_queries_per_pass(vs_index* idx, int qt) {
    if (!idx) return fail(VS_EINVAL, "NULL argument");
    if (qt != 0 && qt != 1) return fail(VS_EINVAL, "queries_per_pass: 0 = auto, 1 = dense-image pass");
    idx->qt_pref = qt;
    return VS_OK;
}

extern "C" int vs_index_set_option(vs_index* idx, const char* name, int value) {
    if (!idx || !name) return fail(VS_EINVAL, "NULL argument");
    const std::string n(name);
    if (n == "queries_per_pass") return vs_index_set_queries_per_pass(idx, value);
    if (n == "blocked_postings") {
        if (value < -1 || value > 1) return fail(VS_EINVAL, "blocked_postings: -1 = auto, 0 = off, 1 = on");
        idx->bp_pref = value;
        if (value == 0) { idx->bp_dir.release(); idx->bp_base.release(); idx->bp_rec.release(); idx->bp_ready = false; }
        idx->bp_tried = false;
        return VS_OK;
    }
    if (n == "postings_rows") {
        if (value != 0 && (value < 256 || value > 8192 || value % 64)) return fail(VS_EINVAL, "postings_rows: 0 = auto, else a multiple of 64 in 256..8192 (clipped to what the index's walk holds: 2048, bag-of-token chunks 8192)");
        idx->bp_rows_pref = value;
        idx->bp_dir.release(); idx->bp_base.release(); idx->bp_rec.release(); idx->bp_ready = false; idx->bp_tried = false;
        return VS_OK;
    }
    if (n == "postings_chunks") {
        if (value < 0 || value > 4096) return fail(VS_EINVAL, "postings_chunks: 0 = auto, else 1..4096");
        idx->bp_chunks = value;
        return VS_OK;
    }
    if (n == "postings_filter") {
        if (value < 0 || value > 1) return fail(VS_EINVAL, "postings_filter: 1 = fixed-point walk + exact refine, 0 = fp64 walk only");
        if (value != idx->bp_filter) { idx->bp_dir.release(); idx->bp_base.release(); idx->bp_rec.release(); idx->bp_ready = false; idx->bp_tried = false; }
        idx->bp_filter = value;
        return VS_OK;
    }
    if (n == "postings_align") {
        if (value < -1 || value > 1) return fail(VS_EINVAL, "postings_align: -1 = auto (packed), 0 = packed, 1 = lists start on 128-byte lines");
        if (value != idx->bp_align_pref) { idx->bp_dir.release(); idx->bp_base.release(); idx->bp_rec.release(); idx->bp_ready = false; idx->bp_tried = false; }
        idx->bp_align_pref = value;
        return VS_OK;
    }
    if (n == "postings_force_fallback") {          // tests: the refine step flags every query, so the exact pass decides all results
        idx->bp_force_fb = value != 0;
        return VS_OK;
    }
    if (n == "postings_quant") {
        if (value < -1 || value > 1) return fail(VS_EINVAL, "postings_quant: -1 = auto, 0 = off, 1 = on (when the data allow it)");
        idx->bp_quant_pref = value;
        idx->bp_dir.release(); idx->bp_base.release(); idx->bp_rec.release(); idx->bp_ready = false; idx->bp_tried = false;
        return VS_OK;
    }
    if (n == "postings_head") {
        if (value < -1 || value == 1 || value > 64)
            return fail(VS_EINVAL, "postings_head: -1 = auto (4), 0 = no dense strips, N in 2..64 = columns present in >= 1/N of the documents");
        if (value != idx->bp_head_pref) {
            idx->bp_dir.release(); idx->bp_base.release(); idx->bp_rec.release(); idx->bp_strip.release(); idx->bp_hmap.release();
            idx->bp_n_head = 0; idx->bp_ready = false; idx->bp_tried = false;
        }
        idx->bp_head_pref = value;
        return VS_OK;
    }
    if (n == "postings_head_gemm") {
        if (value < -1 || value > 1) return fail(VS_EINVAL, "postings_head_gemm: -1 = auto (on), 1 = head columns by the head pre-pass, 0 = inside the walk");
        if (value != idx->bp_head_gemm_pref) {
            idx->bp_dir.release(); idx->bp_base.release(); idx->bp_rec.release(); idx->bp_strip.release(); idx->bp_hmap.release();
            idx->bp_n_head = 0; idx->bp_ready = false; idx->bp_tried = false;
        }
        idx->bp_head_gemm_pref = value;
        return VS_OK;
    }
    if (n == "postings_head_product") {
        if (value < -1 || value > 1) return fail(VS_EINVAL, "postings_head_product: -1 = auto, 0 = operands straight from global memory, 1 = through the LDS ring");
        idx->bp_head_product = value;
        return VS_OK;
    }
    if (n == "postings_head_tiles") {
        if (value < 0 || value > 4096) return fail(VS_EINVAL, "postings_head_tiles: 0 = auto, else tiles per pass of the head pre-pass");
        idx->bp_head_tiles = value;
        return VS_OK;
    }
    if (n == "postings_lanes") {
        if (value != 0 && value != 4 && value != 8) return fail(VS_EINVAL, "postings_lanes: 0 = auto | 4 | 8");
        idx->bp_lanes = value;
        return VS_OK;
    }
    if (n == "postings_walk") {
        if (value < -1 || value > 6)
            return fail(VS_EINVAL, "postings_walk: -1 = auto, 4 = quad chunks (valued index), 6 = bag-of-token chunks / 5 = prefetched records (binary index), 0 = a list per lane group, "
                                   "1 = flat worklists, 2 = list walk on two accumulator sets, 3 = streamed flat walk");
#ifndef VS_EXPERIMENTAL_WALKS
        if (value >= 1 && value <= 3) return fail(VS_EUNSUPPORTED, "postings_walk %d: the experimental walks are not part of this build (make EXPERIMENTAL=1)", value);
#endif
        // quad chunks and records are different copies, and auto (-1) picks between them by size: any change rebuilds at the next search
        if (value != idx->bp_walk_pref) {
            idx->bp_dir.release(); idx->bp_base.release(); idx->bp_rec.release(); idx->bp_quad = false; idx->bp_bq = false; idx->bp_ready = false; idx->bp_tried = false;
        }
        idx->bp_walk_pref = value;
        return VS_OK;
    }
    if (n == "postings_arrange") {
        if (value < -1 || value > 1) return fail(VS_EINVAL, "postings_arrange: -1 = auto (off), 0 = off, 1 = on");
        if (value != idx->bp_arrange_pref) { idx->bp_dir.release(); idx->bp_base.release(); idx->bp_rec.release(); idx->bp_ready = false; idx->bp_tried = false; }
        idx->bp_arrange_pref = value;
        return VS_OK;
    }
    if (n == "postings_pace") {
        if (value < -1 || value > 4096) return fail(VS_EINVAL, "postings_pace: -1 = auto, 0 = free running, N = lock-step window in blocks");
        idx->bp_pace = value;
        return VS_OK;
    }
    if (n == "postings_packed") {
        if (value < -1 || value > 1) return fail(VS_EINVAL, "postings_packed: -1 = auto (on), 1 = packed 16-bit sums of the bag-of-token chunk walk where the batch allows it, 0 = int32 sums");
        idx->bp_packed_pref = value;
        return VS_OK;
    }
    if (n == "mq_variant") {
        if (value < -1 || value > 1) return fail(VS_EINVAL, "mq_variant: -1 = auto, 0 = plain, 1 = shared columns");
        idx->mq_variant = value;
        return VS_OK;
    }
    return fail(VS_EINVAL, "unknown option '%s'", name);
}

// ---- row-sharded search inside ONE process: one vs_index per GPU, no RCCL -- the exchange is B * k (id, score)
// pairs per shard, moved with peer copies over xGMI to the first shard's device and merged there.
struct vs_shard_group {
    std::vector<vs_index*> shards;
    std::vector<int64_t> row0;
    std::vector<hipStream_t> streams;
    std::vector<bool> owns_stream;            // (shards on one device share the first one's stream)
    std::vector<hipEvent_t> done;
    std::vector<vs::DevBuf*> q, ids, sc;      // per shard, on the shard's device
    vs::DevBuf all_ids, all_sc, out_ids, out_sc;   // on shards[0]'s device
    int64_t n_total = 0;
};

extern "C" void vs_shard_group_destroy(vs_shard_group* g) {
    if (!g) return;
    for (size_t i = 0; i < g->shards.size(); ++i) {
        (void)hipSetDevice(g->shards[i]->device);
        if (i < g->streams.size() && g->streams[i]) (void)hipStreamSynchronize(g->streams[i]);
        if (i < g->done.size() && g->done[i]) (void)hipEventDestroy(g->done[i]);
        if (i < g->q.size()) delete g->q[i];
        if (i < g->ids.size()) delete g->ids[i];
        if (i < g->sc.size()) delete g->sc[i];
    }
    for (size_t i = 0; i < g->streams.size(); ++i)
        if (g->streams[i] && i < g->owns_stream.size() && g->owns_stream[i]) { (void)hipSetDevice(g->shards[i]->device); (void)hipStreamDestroy(g->streams[i]); }
    if (!g->shards.empty()) (void)hipSetDevice(g->shards[0]->device);
    delete g;
}

extern "C" int vs_shard_group_create(vs_index* const* shards, int32_t n_shards, vs_shard_group** out) {
    if (!shards || !out || n_shards <= 0) return fail(VS_EINVAL, "bad argument");
    *out = nullptr;
    vs_shard_group* g = new vs_shard_group();
    struct Guard { vs_shard_group* p; ~Guard() { if (p) vs_shard_group_destroy(p); } } guard{g};
    int64_t row = 0;
    for (int i = 0; i < n_shards; ++i) {
        vs_index* s = shards[i];
        if (!s) return fail(VS_EINVAL, "shard %d is NULL", i);
        if (s->n_cols != shards[0]->n_cols) return fail(VS_EINVAL, "shard %d has %d columns, shard 0 has %d", i, s->n_cols, shards[0]->n_cols);
        g->shards.push_back(s);
        g->row0.push_back(row);
        row += s->n_rows;
        VS_HIP(hipSetDevice(s->device));
        // One stream per DEVICE: shards that share a GPU run one after the other.  Every walk is a persistent grid that wants the whole
        // GPU (one 160 KB workgroup per CU, work items in lock step): two of them at once only take each other's CUs, and their
        // lock-step windows wait for workgroups that are not resident (r3: 8 shards on one GPU cost 235 ms against 8 x 20).
        hipStream_t st = nullptr;
        for (int j = 0; j < i && !st; ++j)
            if (g->shards[j]->device == s->device) st = g->streams[j];
        g->owns_stream.push_back(st == nullptr);
        hipEvent_t ev = nullptr;
        if (!st) VS_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        g->streams.push_back(st);
        VS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        g->done.push_back(ev);
        g->q.push_back(new vs::DevBuf());
        g->ids.push_back(new vs::DevBuf());
        g->sc.push_back(new vs::DevBuf());
    }
    if (row >= 0xFFFFFFFFll) return fail(VS_EUNSUPPORTED, "a shard group addresses fewer than 2^32 - 1 documents (%lld given): merged ids are 32-bit", (long long)row);
    g->n_total = row;
    guard.p = nullptr;
    *out = g;
    return VS_OK;
}

extern "C" int vs_shard_group_search(vs_shard_group* g, const void* q, int q_dtype, int64_t ldq, int32_t B, int32_t k, int64_t* out_ids, float* out_scores) {
    if (!g || !q || !out_ids || !out_scores) return fail(VS_EINVAL, "NULL argument");
    if (B <= 0 || k <= 0) return fail(VS_EINVAL, "B and k must be positive");
    if (k > g->n_total) return fail(VS_ERANGE, "selected index k out of range (k = %d > %lld rows)", k, (long long)g->n_total);
    if (q_dtype != VS_F32 && q_dtype != VS_F16) return fail(VS_EINVAL, "q_dtype must be VS_F32 or VS_F16");
    const int n = (int)g->shards.size();
    const int V = g->shards[0]->n_cols;
    if (ldq < V) return fail(VS_EINVAL, "query has %lld columns, index has %d", (long long)ldq, V);
    const size_t esz = dtype_size(q_dtype);
    const size_t q_bytes = ((size_t)(B - 1) * ldq + V) * esz;
    const bool q_dev = is_device_ptr(q);
    int q_device = 0;
    if (q_dev) {
        hipPointerAttribute_t attr;
        VS_HIP(hipPointerGetAttributes(&attr, q));
        q_device = attr.device;
    }
    // Device-resident queries or outputs: the shards run on the group's OWN non-blocking streams, which nothing orders after the
    // stream that produced `q` (an encoder still writing it) or that last used the output buffers (an allocator that recycles them).
    // The caller's stream is not known here, so the devices concerned are synchronised first (documented in the header).
    const bool out_dev_early = is_device_ptr(out_ids);
    if (q_dev) { VS_HIP(hipSetDevice(q_device)); VS_HIP(hipDeviceSynchronize()); }
    if (out_dev_early && (!q_dev || q_device != g->shards[0]->device)) { VS_HIP(hipSetDevice(g->shards[0]->device)); VS_HIP(hipDeviceSynchronize()); }
    // 1. every shard scores the whole batch on its own GPU and stream (asynchronous on the postings filter path)
    std::vector<int> ki((size_t)n);
    int64_t k_tot = 0;
    for (int i = 0; i < n; ++i) {
        vs_index* s = g->shards[i];
        ki[i] = (int)std::min<int64_t>(k, s->n_rows);
        k_tot += ki[i];
        if (ki[i] == 0) continue;
        VS_HIP(hipSetDevice(s->device));
        const void* dq = q;
        if (!q_dev || q_device != s->device) {
            VS_TRY(g->q[i]->reserve(q_bytes));
            if (q_dev) VS_HIP(hipMemcpyPeerAsync(g->q[i]->p, s->device, q, q_device, q_bytes, g->streams[i]));
            else VS_HIP(hipMemcpyAsync(g->q[i]->p, q, q_bytes, hipMemcpyHostToDevice, g->streams[i]));
            dq = g->q[i]->p;
        }
        VS_TRY(g->ids[i]->reserve((size_t)B * ki[i] * 8));
        VS_TRY(g->sc[i]->reserve((size_t)B * ki[i] * 4));
        VS_TRY(vs_index_search(s, dq, q_dtype, ldq, B, ki[i], g->row0[i], g->ids[i]->as<int64_t>(), g->sc[i]->as<float>(), (void*)g->streams[i]));
        VS_HIP(hipEventRecord(g->done[i], g->streams[i]));
    }
    // 2. the exchange: every shard's [B, k_i] block lands in columns of the [B, sum k_i] candidate matrix on the first shard's GPU
    vs_index* s0 = g->shards[0];
    VS_HIP(hipSetDevice(s0->device));
    hipStream_t st0 = g->streams[0];
    VS_TRY(g->all_ids.reserve((size_t)B * k_tot * 8));
    VS_TRY(g->all_sc.reserve((size_t)B * k_tot * 4));
    int64_t col = 0;
    for (int i = 0; i < n; ++i) {
        if (ki[i] == 0) continue;
        if (i > 0) VS_HIP(hipStreamWaitEvent(st0, g->done[i], 0));
        VS_HIP(hipMemcpy2DAsync(g->all_ids.as<int64_t>() + col, (size_t)k_tot * 8, g->ids[i]->p, (size_t)ki[i] * 8, (size_t)ki[i] * 8, (size_t)B, hipMemcpyDefault, st0));
        VS_HIP(hipMemcpy2DAsync(g->all_sc.as<float>() + col, (size_t)k_tot * 4, g->sc[i]->p, (size_t)ki[i] * 4, (size_t)ki[i] * 4, (size_t)B, hipMemcpyDefault, st0));
        col += ki[i];
    }
    // 3. merge on the first shard's GPU (canonical order: identical to searching the unsharded index)
    const bool out_dev = is_device_ptr(out_ids);
    if (out_dev != is_device_ptr(out_scores)) return fail(VS_EINVAL, "out_ids and out_scores must both be host or both device pointers");
    int64_t* d_ids = out_ids;
    float* d_sc = out_scores;
    if (!out_dev) {
        VS_TRY(g->out_ids.reserve((size_t)B * k * 8));
        VS_TRY(g->out_sc.reserve((size_t)B * k * 4));
        d_ids = g->out_ids.as<int64_t>();
        d_sc = g->out_sc.as<float>();
    }
    VS_TRY(vs_merge_topk(g->all_ids.as<int64_t>(), g->all_sc.as<float>(), B, k_tot, k, d_ids, d_sc, s0->device, (void*)st0));
    if (!out_dev) {
        VS_HIP(hipMemcpyAsync(out_ids, d_ids, (size_t)B * k * 8, hipMemcpyDeviceToHost, st0));
        VS_HIP(hipMemcpyAsync(out_scores, d_sc, (size_t)B * k * 4, hipMemcpyDeviceToHost, st0));
    }
    VS_HIP(hipStreamSynchronize(st0));
    return VS_OK;
}

extern "C" int vs_profile_enable(int on) {
    Profiler::get().on = on != 0;
    return VS_OK;
}
extern "C" int vs_profile_reset(void) {
    Profiler::get().drain();
    std::lock_guard<std::mutex> g(Profiler::get().mu);
    Profiler::get().acc.clear();
    return VS_OK;
}
extern "C" int vs_profile_read(const char* kernel, double* total_ms, int64_t* launches) {
    if (!kernel) return fail(VS_EINVAL, "kernel is NULL");
    Profiler::get().drain();
    std::lock_guard<std::mutex> g(Profiler::get().mu);
    auto it = Profiler::get().acc.find(kernel);
    if (total_ms) *total_ms = it == Profiler::get().acc.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == Profiler::get().acc.end() ? 0 : it->second.n;
    return VS_OK;
}
