// api.hip -- library-level entry points and the kind dispatch of Index.search (index.py:88-94).
#include "common.h"

#include <string>

using namespace vs;

int vs_csr_search(vs_index*, const void*, int, int64_t, int32_t, int32_t, int64_t, int64_t*, float*, hipStream_t);
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
        if (value != 0 && (value < 256 || value > 2048 || value % 64)) return fail(VS_EINVAL, "postings_rows: 0 = auto, else a multiple of 64 in 256..2048");
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
    if (n == "postings_lanes") {
        if (value != 4 && value != 8) return fail(VS_EINVAL, "postings_lanes: 4 | 8");
        idx->bp_lanes = value;
        return VS_OK;
    }
    if (n == "mq_variant") {
        if (value < -1 || value > 1) return fail(VS_EINVAL, "mq_variant: -1 = auto, 0 = plain, 1 = shared columns");
        idx->mq_variant = value;
        return VS_OK;
    }
    return fail(VS_EINVAL, "unknown option '%s'", name);
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
