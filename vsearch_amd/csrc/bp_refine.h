// bp_refine.h -- the "refine" half of the filter-and-refine postings search (see AM_FIX in bp_walk.h).
//
// The fixed-point walk returns, per query, the K' > k documents with the largest APPROXIMATE scores.  Here those K'
// documents are re-scored from their CSR rows with the library's exact numerics (fp32 product, fp64 sum -- what the CSR
// pass and the fp64 walk compute), the exact top k is written out, and the kernel PROVES per query that no document
// outside the K' can belong to it:  a document the walk did not return has an approximate sum A <= A_cut (the K'-th best).
// The walk adds trunc(fl32(w S v)) per matched term: the truncation loses less than one unit per term (n = the query's non-zeros
// bounds the terms), the fp32 rounding of a product x at most 2^-24 |x| -- more than a unit once x >= 2^25, which a query of few or
// one dominant non-zero reaches -- but sum |x| <= sum |w| vmax S < 2^30 (the choice of S), so all roundings of a document together
// stay below 2^6 units.  The library's own score sums fl32(w v) in fp64: above the real sum by at most another 2^-24 sum |w v|,
// again < 2^6 units of 1 / S.  Hence a document outside the K' has a library score below (A_cut + n + 1 + 128) / S -- the slack
// bp_qscale_kernel stores; round-2 builds left the 128 out (a hole no test could reach, closed here).  If the exact
// k-th score is above that bound the query is done; otherwise its flag is set and an exact pass re-runs it.
//
// Lossy records: for an fp32 index the filter's copy stores the values ROUNDED TO fp16 (4 instead of 6 bytes per posting -- the
// walk is bound by the bytes a CU can pull through its L1, rocprofv3: ~23 B/clk/CU).  The rounding error is relative
// (<= 2^-11 per value, + 2^-25 absolute for fp16 subnormals), so it widens the bound by a factor, not by a constant; it
// needs non-negative values and weights (VDR embeddings are, vdr.py:73-75 elu1p >= 0), else the copy keeps fp32 values.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "bp_walk.h"

namespace vs {

// out[0] = max |value| of a valued index (float bits; bounds every product of the walk), out[1] = 1 if any value is negative
template <int VM>
__global__ __launch_bounds__(256) void bp_vmax_kernel(const void* vals, int64_t n, uint32_t* out_bits) {
    float m = 0.f;
    bool neg = false;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float v;
        if constexpr (VM == VM_F32) v = reinterpret_cast<const float*>(vals)[i];
        else v = __half2float(reinterpret_cast<const __half*>(vals)[i]);
        m = fmaxf(m, fabsf(v));
        neg = neg || v < 0.f;
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(m));       // non-negative floats order like their bits
    if (__builtin_amdgcn_ballot_w64(neg) && (threadIdx.x & 63) == 0) atomicOr(out_bits + 1, 1u);
}

// Per query: S = 2^e with (sum |w|) * vmax * S < 2^30 (no int32 sum can wrap), the slack n = non-zeros + 1 + 128 (truncations + the fp32 rounding of the products, see the header) in fixed-point
// units and sum |w|.  n = 0 marks a query whose walk is EXACT (binary index and every w * S an integer): its approximate order
// is the exact order, nothing to prove.  n = -1 marks a query the quantised records cannot bound (a negative weight): it
// goes straight to the exact pass.  One wave per query, fixed reduction order.
template <int UNUSED>
__global__ __launch_bounds__(256) void bp_qscale_kernel(const int64_t* qptr, const float* qvals, int32_t B, const uint32_t* vmax_bits, int binary, int quant,
                                                        float* qscale, int32_t* qslack, float* qwsum, const int32_t* qcols, const uint16_t* hmap, int32_t head_slack,
                                                        int32_t pk_maxrow = 0, uint32_t* flag16 = nullptr) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= B) return;
    const int64_t e0 = qptr[q], e1 = qptr[q + 1];
    float sum = 0.f;
    bool neg = false;
    int heads = 0;                                                   // entries on head columns (dense strips, bp_walk.h)
    for (int64_t e = e0 + lane; e < e1; e += 64) {
        sum += fabsf(qvals[e]);
        neg = neg || qvals[e] < 0.f;
        if (hmap && hmap[qcols[e]] != 0xFFFFu) ++heads;
    }
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); heads += __shfl_xor(heads, o, 64); }
    neg = __builtin_amdgcn_ballot_w64(neg) != 0ull;
    const float vmax = binary ? 1.f : __uint_as_float(vmax_bits[0]);
    const float bound = sum * vmax * 1.0001f;                        // the reduction above is not the walk's order: a hair of slack
    int e = 0;
    if (bound > 0.f && isfinite(bound)) {
        int be;
        (void)frexpf(bound, &be);                                    // bound < 2^be
        e = 30 - be;
    }
    e = max(-60, min(60, e));
    // Packed 16-bit sums of the bag-of-token chunk walk (bp_bq.h, flag16 != nullptr): the SMALLEST power-of-two scale that makes every weight
    // an integer, when a document's sum then stays below 2^16 -- a document matches at most min(longest row, n) of the query's columns,
    // each with at most the largest weight.  Weights >= 0 only (a negative one would borrow from the slot in the dword's upper half).
    if (flag16) {
        int need = -1000;                                            // the scale's exponent: - (exponent of the weights' lowest set bit)
        float wmax = 0.f;
        for (int64_t x = e0 + lane; x < e1; x += 64) {
            const float w = qvals[x];
            wmax = fmaxf(wmax, w);
            if (w > 0.f && isfinite(w)) {
                int we;
                const float m = frexpf(w, &we);                      // w = m 2^we, m in [0.5, 1): 24 mantissa bits
                const uint32_t mi = (uint32_t)ldexpf(m, 24);
                need = max(need, 24 - we - (__ffs((int)mi) - 1));
            } else if (w != 0.f) need = 1000;                        // NaN / inf: not here
        }
        for (int o = 32; o > 0; o >>= 1) { need = max(need, __shfl_xor(need, o, 64)); wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64)); }
        bool ok16 = binary != 0 && !neg && need >= -40 && need <= 40 && e1 > e0;
        if (ok16) {
            const double wi = (double)wmax * ldexp(1.0, need);       // the largest integer weight
            const double terms = (double)min((int64_t)max(pk_maxrow, 1), e1 - e0);
            ok16 = wi * terms < 65536.0;
        }
        if (ok16) e = need;
        if (lane == 0) flag16[q] = ok16 ? 1u : 0u;
    }
    const float S = ldexpf(1.f, e);
    bool exact = binary != 0;
    if (exact) {
        for (int64_t x = e0 + lane; x < e1; x += 64) {
            const float w = qvals[x] * S;
            exact = exact && (w == truncf(w));
        }
        exact = __builtin_amdgcn_ballot_w64(!exact) == 0ull;
    }
    if (lane == 0) {
        qscale[q] = S;
        qwsum[q] = sum * 1.0001f;
        // the dense part of a score (matrix cores, bp_walk.h): bp_head_slack
        const int64_t dense_slack = heads > 0 ? (int64_t)head_slack : 0;
        // one unit of truncation per term, + 2 x 2^6 for the fp32 rounding of the products (the walk's and the library's: header)
        qslack[q] = (exact || e1 == e0) ? 0 : (((quant || heads > 0) && neg) ? -1 : (int32_t)min((int64_t)1 << 24, e1 - e0 + 1 + 128 + dense_slack));
    }
}

struct RefineArgs {
    const uint64_t* cand;     // [B, n_cand] approximate keys (make_key_fix), runs of run_len sorted descending
    int64_t n_cand;
    int32_t run_len;
    int32_t B, k, kp;         // kp = K' documents re-scored per query (k <= kp <= kBpMaxK)
    const uint32_t* pk_ptr;   // the CSR packets of the same index
    const uint4* cols;
    const void* vals;
    int32_t n_cols;
    int64_t n_rows;
    const float* q;           // [B, n_cols] dense fp32 queries (already rounded to the index dtype)
    const float* qscale;      // [B]
    const int32_t* qslack;    // [B]
    const float* qwsum;       // [B] sum |w| (rounded up)
    int32_t force_flag;       // tests: flag every query
    int32_t quant;            // 1: the records hold fp16-rounded copies of non-negative fp32 values (lossy filter copy)
    int64_t id_offset;
    int64_t* out_ids;         // [B, out_ld]
    float* out_scores;
    int64_t out_ld;
    uint32_t* flags;          // [B] in: 2 = the query entered no tile; out: 1 = the top k could not be proven from K' candidates
};

// IMG = 1: the query's dense fp32 row sits in LDS for the re-scoring (118 KB at V = 29 523: with the candidate buffers 159 KB, one
// workgroup per CU).  The 98 k weight look-ups of a query's 128 candidates are random 4-byte reads: from L2 they ran at the CU's
// ~0.13 lines per clock (outstanding-miss limit) and made the kernel 0.63 ms per 1024 queries; from LDS they are a few microseconds.
// IMG = 0 (wider vocabularies): global look-ups.
__host__ __device__ inline size_t refine_lds_bytes(int32_t n_cols, int img) {
    return (img ? scan_img_bytes(n_cols) : 0) + (size_t)kWgCap * 8 + (size_t)kBpMaxK * 8 + 16;
}
template <int VM, int IMG>
__global__ __launch_bounds__(kScanThreads) void refine_topk_kernel(RefineArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_r[];
    float* img = reinterpret_cast<float*>(smem_r);
    uint64_t* buf = reinterpret_cast<uint64_t*>(smem_r + (IMG ? scan_img_bytes(a.n_cols) : 0));      // [kWgCap]
    uint64_t* ex = buf + kWgCap;                                                                       // [kBpMaxK]
    int* cnt_ptr = reinterpret_cast<int*>(ex + kBpMaxK);
    int& cnt_sh = *cnt_ptr;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int K = a.k, KP = a.kp;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        if (a.flags[b] == 2u) continue;                             // too dense for a tile: no candidates exist, the exact pass takes it
        const float* qrow = a.q + (size_t)b * a.n_cols;
        if constexpr (IMG != 0) {
            __syncthreads();                                          // (the previous query's look-ups are done)
            for (int i = tid; i <= a.n_cols; i += kScanThreads) img[i] = i < a.n_cols ? qrow[i] : 0.f;      // pad columns (id n_cols) carry no weight
        }
        merge_select(a.cand + (size_t)b * a.n_cand, a.n_cand, a.run_len, KP, nullptr, buf, &cnt_sh, tid);
        __syncthreads();
        const uint64_t cut_key = buf[KP - 1];                         // 0: fewer than K' documents exist -> every document is a candidate
        int pow2 = 64;
        while (pow2 < KP) pow2 <<= 1;
        for (int i = KP + tid; i < pow2; i += kScanThreads) ex[i] = 0ull;
        // exact scores: one wave per candidate, a lane takes whole packets (8 non-zeros), fp32 products summed in fp64
        for (int i = w; i < KP; i += kScanWaves) {
            const uint64_t key = buf[i];
            uint64_t out = 0ull;
            if (key != 0ull) {
                const uint32_t row = key_row(key);
                const double sum = row_sum_f64<VM>(a.pk_ptr, a.cols, a.vals, row, lane, [&](uint32_t col) {
                    if constexpr (IMG != 0) return img[col];
                    else return col < (uint32_t)a.n_cols ? qrow[col] : 0.f;                                   // pad columns (id n_cols) carry no weight
                });
                out = make_key((float)sum, row);
            }
            if (lane == 0) ex[i] = out;
        }
        wg_sort_desc<kScanThreads>(ex, pow2, tid);
        for (int i = tid; i < K; i += kScanThreads) {
            const uint64_t key = ex[i];
            a.out_ids[(size_t)b * a.out_ld + i] = (int64_t)key_row(key) + a.id_offset;
            a.out_scores[(size_t)b * a.out_ld + i] = key_score(key);
        }
        if (tid == 0) {
            const int32_t slack = a.qslack[b];
            bool ok = slack == 0 || (cut_key == 0ull && slack > 0);
            if (!ok && slack > 0) {
                double bound = ((double)key_fix(cut_key) + (double)slack) / (double)a.qscale[b];    // exact sums outside the K' stay below
                // Quantised records (values v >= 0 stored as fp16(v), weights w >= 0): |v - fp16(v)| <= 2^-11 v + 2^-25, so the exact sum
                // E of a document and the sum Q over its stored values satisfy E (1 - 2^-11) <= Q + 2^-25 sum |w|.
                if (a.quant) bound = (bound + (double)a.qwsum[b] * 0x1p-25) / (1.0 - 0x1p-11);
                if (a.quant >= 2) bound = bound / (1.0 - 0x1p-11);          // head weights rounded to ONE fp16 number: 2^-11 relative once more
                float bf = (float)bound;
                if ((double)bf < bound) bf = nextafterf(bf, INFINITY);
                ok = ex[K - 1] != 0ull && key_score(ex[K - 1]) > bf;
            }
            a.flags[b] = (ok && !a.force_flag) ? 0u : 1u;
        }
        __syncthreads();
    }
}

// Device-side tile plan of the filter search: row pointers of the sparse batch + greedy tiling (<= qt queries and <= vals_cap
// entries per tile, consecutive queries).  A query denser than vals_cap enters no tile: it gets no entries (qptr does not advance)
// and flag 2 -- it takes the exact one-query scan.  plan[0] = tiles, plan[1] = densest query, plan[2] = entries in tiles.
template <int UNUSED>
__global__ __launch_bounds__(64) void bp_plan_kernel(const int64_t* counts, int32_t B, int32_t qt, int32_t vals_cap, int64_t* qptr, int2* tiles, int64_t* plan,
                                                     uint32_t* flags) {
    // The greedy tiling is sequential.  ONE WAVE: 64 counts at a time sit in a register across the lanes, the loop reads them with
    // v_readlane and keeps its state in scalar registers; results go back through the lanes and leave coalesced.  (Still 0.15 ms
    // for 1024 queries -- ~60 scalar instructions per query from a lone wave; jump pointers per query, followed tile by tile, would
    // cut the sequential part eightfold.)
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x;
    int64_t acc = 0, mx = 0, nz = 0;
    int nt = 0, start = -1, cnt = 0;
    if (lane == 0) qptr[0] = 0;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int nb = min(64, B - b0);
        const int32_t c_l = lane < nb ? (int32_t)min(counts[b0 + lane], (int64_t)0x7FFFFFFF) : 0;
        int64_t q_l = 0;
        uint32_t f_l = 0;
        for (int i = 0; i < nb; ++i) {
            const int64_t c = (int64_t)__builtin_amdgcn_readlane(c_l, i);
            mx = c > mx ? c : mx;
            const bool dense = c > vals_cap;
            if (!dense && cnt > 0 && (cnt == qt || nz + c > vals_cap)) {          // the open tile is full: close it
                if (lane == 0) tiles[nt] = make_int2(start, cnt);
                ++nt;
                cnt = 0;
            }
            if (dense) {
                if (cnt > 0) {                                                       // tiles are runs of consecutive queries
                    if (lane == 0) tiles[nt] = make_int2(start, cnt);
                    ++nt;
                    cnt = 0;
                }
            } else {
                if (cnt == 0) { start = b0 + i; nz = 0; }
                ++cnt;
                nz += c;
                acc += c;
            }
            if (lane == i) { q_l = acc; f_l = dense ? 2u : 0u; }
        }
        if (lane < nb) { qptr[b0 + lane + 1] = q_l; flags[b0 + lane] = f_l; }
    }
    if (cnt > 0) {
        if (lane == 0) tiles[nt] = make_int2(start, cnt);
        ++nt;
    }
    if (lane == 0) {
        plan[0] = nt;
        plan[1] = mx;
        plan[2] = acc;
        plan[3] = 0;
    }
}

// The same plan for batches of up to kPlanFast queries (34 KB of LDS) in ~20 us: the counts, their running sums and, for every query, where a
// tile STARTING there would end (a jump pointer: at most qt look-aheads, all queries in parallel) are computed by the whole
// workgroup; only the walk along the jump pointers -- one step per TILE, not per query -- is sequential.
constexpr int kPlanFast = 4096;
template <int UNUSED>
__global__ __launch_bounds__(256) void bp_plan_fast_kernel(const int64_t* counts, int32_t B, int32_t qt, int32_t vals_cap, int64_t* qptr, int2* tiles,
                                                           int64_t* plan, uint32_t* flags) {
    __shared__ int32_t c_sh[kPlanFast];             // count, or -1 for a query too dense for a tile
    __shared__ int32_t nx_sh[kPlanFast];            // first query after the tile that starts here
    __shared__ long long seg_sh[256];
    __shared__ int mx_sh;
    if (blockIdx.x != 0) return;
    const int tid = threadIdx.x;
    if (tid == 0) mx_sh = 0;
    __syncthreads();
    int mx = 0;
    for (int b = tid; b < B; b += 256) {
        const int64_t c64 = counts[b];
        const int32_t c = (int32_t)min(c64, (int64_t)0x7FFFFFFF);
        mx = max(mx, c);
        const bool dense = c > vals_cap;
        c_sh[b] = dense ? -1 : c;
        flags[b] = dense ? 2u : 0u;
    }
    atomicMax(&mx_sh, mx);
    __syncthreads();
    // running sums of the counts that enter tiles: a contiguous segment per thread, then a scan of the 256 segment totals
    const int seg = (B + 255) / 256;
    const int i0 = min(B, tid * seg), i1 = min(B, i0 + seg);
    long long mine = 0;
    for (int i = i0; i < i1; ++i) mine += max(c_sh[i], 0);
    seg_sh[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        long long run = 0;
        for (int i = 0; i < 256; ++i) { const long long v = seg_sh[i]; seg_sh[i] = run; run += v; }
        qptr[0] = 0;
        plan[1] = mx_sh;
        plan[2] = run;
        plan[3] = 0;
    }
    __syncthreads();
    long long run = seg_sh[tid];
    for (int i = i0; i < i1; ++i) { run += max(c_sh[i], 0); qptr[i + 1] = run; }
    // jump pointers
    for (int b = tid; b < B; b += 256) {
        int j = b, cnt = 0;
        long long nz = 0;
        while (j < B && cnt < qt && c_sh[j] >= 0 && nz + c_sh[j] <= vals_cap) { nz += c_sh[j]; ++cnt; ++j; }
        nx_sh[b] = max(j, b + 1);
    }
    __syncthreads();
    if (tid == 0) {
        int nt = 0;
        for (int b = 0; b < B;) {
            if (c_sh[b] < 0) { ++b; continue; }
            const int j = nx_sh[b];
            tiles[nt++] = make_int2(b, j - b);
            b = j;
        }
        plan[0] = nt;
    }
}

// non-zeros per dense query row AND the batch's column frequencies (what the walk accounting needs) in one pass over the batch
template <int UNUSED>
__global__ __launch_bounds__(256) void bp_count_colfreq_kernel(const float* x, int64_t ld, int32_t B, int32_t V, int64_t* counts, uint32_t* colfreq) {
    __shared__ int part[4];
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        int c = 0;
        for (int i = threadIdx.x; i < V; i += 256)
            if (x[(size_t)b * ld + i] != 0.f) { ++c; atomicAdd(&colfreq[i], 1u); }
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) counts[b] = part[0] + part[1] + part[2] + part[3];
    }
}

// flagged queries -> one-query tiles for the exact pass (and the query list of the merge behind it)
template <int UNUSED>
__global__ __launch_bounds__(kScanThreads) void fb_plan_kernel(const uint32_t* flags, int32_t B, int2* tiles, int32_t* n_tiles) {
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += kScanThreads) {                      // ascending query order within a sweep is not needed: tiles are independent
        const int b = b0 + threadIdx.x;
        if (b < B && flags[b]) tiles[atomicAdd(&cnt, 1)] = make_int2(b, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) n_tiles[0] = cnt;
}

}  // namespace vs
