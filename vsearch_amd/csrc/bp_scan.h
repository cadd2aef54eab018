// bp_scan.h -- "blocked postings": a second, column-grouped copy of a long-row CSR index for SPARSE queries.
//
// The CSR scan (csr_scan_mq.h) looks every index non-zero up in the tile table, although only ~2.6 % x Qt of
// them carry a query weight -- and at Qt = 8 that lookup + hit handling (VALU issue), not HBM, bounds the pass.
// Here the rows are cut into blocks of up to kBpRows documents (sized at build time); inside a block the non-zeros
// are grouped by column (dir[b][c] .. dir[b][c+1] = the postings (document-in-block uint16, value) of column c).  A
// query tile then walks ONLY the posting lists of its own columns: every visited non-zero is a hit, a quad of lanes
// reads one list with 16-byte loads, and the products go to fp64 accumulators [document][query slot] in LDS (same
// numerics as the CSR pass: fp32 product, fp64 sum, order-independent).  One block = up to 1024 documents x 8 queries
// of accumulators (72 KB with the row pitch); after each block a thread finishes one document (8 sums -> order keys
// -> candidate buffers), exactly like the per-row epilogue of the CSR pass.  Bytes read per tile = the tile's share
// of the postings (~21 % of the index at 8 x 776 query non-zeros) + the directory entries.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "csr_scan.h"
#include "dense_csr.h"

namespace vs {

constexpr int kBpRows = 1024;         // most documents per block (= threads per workgroup: one document per thread at block end); the actual
                                      // count is picked at build time so that a column's list in a block averages ~22 postings: lists beyond
                                      // the 32 a quad takes per round cost the whole wave a second, mostly idle round
constexpr int kBpCap = 2048;          // candidate slots per (workgroup, query slot)
constexpr int kBpMaxK = kBpCap - kBpRows;
constexpr int kBpPitch = 9;           // accumulator row pitch in doubles (8 slots + 1): a document's row starts 72 B after its neighbour's, so
                                      // the adds of a wave spread over all LDS banks and the address is ONE mad (document * 72 + slot * 8)
constexpr int kBpEntCap = 7168;       // (query, column) entries per tile: 56 KB of LDS, and the 8192-slot entry sort must hold them
constexpr int kBpGroup = 4;           // lanes walking one posting list, 8 consecutive postings (16-byte loads) per lane and round
constexpr int kBpBatch = 4;           // posting lists whose loads are in flight together per group (the walk is latency-bound otherwise)

// First posting of block b: room for the block's non-zeros (its packets x 8) plus one pad posting per column.
__host__ __device__ inline size_t bp_block_base(uint32_t first_packet, int64_t b, int32_t n_cols) {
    return (size_t)first_packet * 8 + (size_t)b * (size_t)((n_cols + 2) & ~1);
}
__host__ __device__ inline size_t bp_postings_capacity(int64_t n_packets, int64_t n_blocks, int32_t n_cols) {
    return (size_t)n_packets * 8 + (size_t)n_blocks * (size_t)((n_cols + 2) & ~1) + 64;      // + slack: the scan's 16-byte loads may run past a list
}

// ---- builder: one workgroup per block; counts per column -> directory -> scatter ----------------------
template <int VM>
__global__ __launch_bounds__(kScanThreads) void bp_build_kernel(const uint32_t* pk_ptr, const uint4* cols, const void* vals, int64_t n_rows,
                                                                int32_t n_cols, int32_t rows, uint32_t* dir, uint16_t* pdoc, void* pval) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem);                  // [n_cols + 1]
    __shared__ int scratch[32];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t n_blocks = (n_rows + rows - 1) / rows;
    const int seg = (n_cols + 1 + kScanThreads - 1) / kScanThreads;
    for (int64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const int64_t r0 = b * rows, r1 = min(n_rows, r0 + rows);
        const uint32_t P0 = pk_ptr[r0], P1 = pk_ptr[r1];
        __syncthreads();
        for (int i = tid; i <= n_cols; i += kScanThreads) cnt[i] = 0;
        __syncthreads();
        for (uint32_t p = P0 + tid; p < P1; p += kScanThreads) {
            const uint4 cw = cols[p];
            const uint32_t cwv[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                atomicAdd(&cnt[cwv[i] & 0xFFFFu], 1u);                  // padding lands in cnt[n_cols], never read back
                atomicAdd(&cnt[cwv[i] >> 16], 1u);
            }
        }
        __syncthreads();
        if (tid == 0) cnt[n_cols] = 0;
        __syncthreads();
        {   // exclusive scan in column order; dir[b][c] = first posting of column c, dir[b][n_cols] = postings in the block
            const int i0 = tid * seg, i1 = min(n_cols + 1, i0 + seg);
            // every list starts on an even posting (4-byte aligned document ids: the scan uses 16-byte loads); an odd list
            // is followed by one pad posting (document 0, value 0 -- the arrays are zero-filled before the build)
            int mine = 0;
            for (int i = i0; i < i1; ++i) mine += ((int)cnt[i] + 1) & ~1;
            int off = block_excl_scan(mine, scratch, tid, nullptr);
            uint32_t* d = dir + (size_t)b * (n_cols + 1);
            for (int i = i0; i < i1; ++i) {
                const int c = ((int)cnt[i] + 1) & ~1;
                cnt[i] = (uint32_t)off;                                  // becomes the column's write cursor
                d[i] = (uint32_t)off;
                off += c;
            }
        }
        __syncthreads();
        const size_t base = bp_block_base(P0, b, n_cols);
        for (int64_t r = r0 + w; r < r1; r += kScanWaves) {
            const uint32_t p0 = pk_ptr[r], p1 = pk_ptr[r + 1];
            const uint16_t dl = (uint16_t)(r - r0);
            for (uint32_t p = p0 + lane; p < p1; p += 64) {
                const uint4 cw = cols[p];
                const uint32_t cwv[4] = {cw.x, cw.y, cw.z, cw.w};
                float v[8];
                if constexpr (VM == VM_F32) {
                    const float4* vp = reinterpret_cast<const float4*>(vals);
                    const float4 v0 = vp[2 * (size_t)p], v1 = vp[2 * (size_t)p + 1];
                    v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
                } else {
                    const uint4 hv = reinterpret_cast<const uint4*>(vals)[p];
                    const __half2* h = reinterpret_cast<const __half2*>(&hv);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float2 f = __half22float2(h[i]); v[2 * i] = f.x; v[2 * i + 1] = f.y; }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t c = (i & 1) ? (cwv[i >> 1] >> 16) : (cwv[i >> 1] & 0xFFFFu);
                    if (c < (uint32_t)n_cols) {
                        const uint32_t pos = atomicAdd(&cnt[c], 1u);
                        pdoc[base + pos] = dl;
                        if constexpr (VM == VM_F32) reinterpret_cast<float*>(pval)[base + pos] = v[i];
                        else reinterpret_cast<__half*>(pval)[base + pos] = __float2half(v[i]);
                    }
                }
            }
        }
    }
}

// df[c] = postings of column c over all blocks, pad postings included (what one query entry on column c streams)
template <int UNUSED>
__global__ void bp_df_kernel(const uint32_t* dir, int64_t n_blocks, int32_t n_cols, uint32_t* df) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cols) return;
    uint32_t acc = 0;
    for (int64_t b = 0; b < n_blocks; ++b) {
        const uint32_t* d = dir + (size_t)b * ((size_t)n_cols + 1);
        acc += d[c + 1] - d[c];
    }
    df[c] = acc;
}
// out[0] = sum over columns of (queries of the batch using the column) x df: the postings this batch's search walks
template <int UNUSED>
__global__ __launch_bounds__(kScanThreads) void bp_walk_kernel(const uint32_t* colfreq, const uint32_t* df, int32_t n_cols, int64_t* out) {
    __shared__ unsigned long long red[kScanThreads / 64];
    unsigned long long v = 0;
    for (int c = threadIdx.x; c < n_cols; c += kScanThreads) v += (unsigned long long)colfreq[c] * df[c];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int i = 0; i < kScanThreads / 64; ++i) t += red[i];
        out[0] = (int64_t)t;
    }
}

struct BpArgs {
    const uint32_t* pk_ptr;   // [n_rows + 1] (block b's postings start at bp_block_base(pk_ptr[b * rows], b))
    int32_t rows;             // documents per block (<= kBpRows)
    const uint32_t* dir;      // [n_blocks, n_cols + 1]
    const uint16_t* pdoc;
    const void* pval;
    int64_t n_rows;
    int32_t n_cols;
    int32_t k;
    int32_t nchunk;
    int64_t blocks_per_chunk;
    const int64_t* qptr;      // sparse queries (CSR over the batch) and the tile plan -- as MqArgs
    const int32_t* qcols;
    const float* qvals;
    const int2* tiles;
    int32_t n_tiles;
    int32_t ent_cap;          // LDS capacity for tile entries
    uint64_t* cand;           // [B, nchunk, k] output keys, sorted descending
    uint64_t* gcand;          // [grid, QT, kBpCap] scratch
    const uint64_t* upper;    // optional [B] exclusive upper bounds ("search after")
};

template <int QT>
__host__ __device__ inline size_t bp_lds_bytes(int ent_cap) {
    return (size_t)kBpRows * kBpPitch * 8 + (size_t)kBpCap * 8 + (size_t)QT * 16 + 64 * 4 + (size_t)ent_cap * 8;
}

template <int VM, int QT>
__global__ __launch_bounds__(kScanThreads) void bp_scan_topk(BpArgs a) {
    static_assert(kBpRows == kScanThreads, "one document per thread at block end");
    static_assert(kBpBatch == kBpGroup, "one directory-owning lane per list of a batch");
    static_assert(QT < kBpPitch && QT * kBpRows >= 8192, "row pitch covers the slots; the accumulator area holds the 8192-slot entry sort");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* acc = reinterpret_cast<double*>(smem);                                  // [kBpRows][kBpPitch]
    uint64_t* sortbuf = reinterpret_cast<uint64_t*>(acc + kBpRows * kBpPitch);      // [kBpCap]
    unsigned long long* tau = reinterpret_cast<unsigned long long*>(sortbuf + kBpCap);   // [QT]
    unsigned long long* upper_sh = tau + QT;                                        // [QT] exclusive upper bounds ("search after")
    int* scratch = reinterpret_cast<int*>(upper_sh + QT);                           // [48]
    unsigned int* ccnt = reinterpret_cast<unsigned int*>(scratch + 48);             // [QT]
    uint2* ent = reinterpret_cast<uint2*>(scratch + 64);                            // [ent_cap]: x = column | slot << 16, y = weight bits

    const int tid = threadIdx.x;
    const int gid = tid / kBpGroup, gl = tid % kBpGroup;
    constexpr int NG = kScanThreads / kBpGroup;                                     // 64 groups
    const int K = a.k;
    uint64_t* my_gcand = a.gcand + (size_t)blockIdx.x * QT * kBpCap;
    const int64_t n_blocks = (a.n_rows + a.rows - 1) / a.rows;
    const int64_t items = (int64_t)a.n_tiles * a.nchunk;
    const size_t dir_ld = (size_t)a.n_cols + 1;

    // Work items = (tile, chunk), taken round-robin (the host picks nchunk so that an XCD keeps to few chunks, see bp launch)
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int tile = (int)(item / a.nchunk), c = (int)(item % a.nchunk);
        const int q0 = a.tiles[tile].x, nq = a.tiles[tile].y;
        const int64_t b0 = (int64_t)c * a.blocks_per_chunk, b1 = min(n_blocks, b0 + a.blocks_per_chunk);
        __syncthreads();
        const int64_t e0 = a.qptr[q0], e1 = a.qptr[q0 + nq];
        const int n_ent = (int)(e1 - e0);
        // Entries sorted by column (the accumulator area doubles as the sort buffer): neighbouring quads then read neighbouring
        // directory words and neighbouring posting lists -- the directory is fetched once per block instead of one 64-byte
        // sector per entry, and the walk over the block's postings becomes a forward sweep with gaps.
        {
            uint64_t* skey = reinterpret_cast<uint64_t*>(acc);              // kBpRows * QT = 8192 slots
            for (int i = tid; i < kBpRows * QT; i += kScanThreads) {
                uint64_t key = 0;
                if (i < n_ent) {
                    const int64_t e = e0 + i;
                    int qs = 0;
                    while (e >= a.qptr[q0 + qs + 1]) ++qs;
                    key = ((uint64_t)(uint32_t)a.qcols[e] << 40) | ((uint64_t)qs << 32) | (uint64_t)__float_as_uint(a.qvals[e]);
                }
                skey[i] = key;
            }
            wg_sort_desc<kScanThreads>(skey, kBpRows * QT, tid);
            for (int i = tid; i < n_ent; i += kScanThreads) {
                const uint64_t key = skey[i];
                ent[i] = make_uint2((uint32_t)(key >> 40) | ((uint32_t)((key >> 32) & 0xFFu) << 16), (uint32_t)key);
            }
            __syncthreads();
        }
        for (int i = tid; i < kBpRows * kBpPitch; i += kScanThreads) acc[i] = 0.0;
        if (tid < QT) { tau[tid] = 0ull; ccnt[tid] = 0u; }
        if (tid < QT) upper_sh[tid] = (a.upper && tid < nq) ? a.upper[q0 + tid] : ~0ull;
        __syncthreads();

        for (int64_t b = b0; b < b1 || b == b0; ++b) {
            if (b < b1) {
                const uint32_t* dirb = a.dir + (size_t)b * dir_ld;
                const size_t base = bp_block_base(a.pk_ptr[b * a.rows], b, a.n_cols);
                // block-uniform base pointers + 32-bit byte offsets: the loads take the scalar-base form, no 64-bit address per lane
                const char* bdoc = reinterpret_cast<const char*>(a.pdoc + base);
                const char* bval = reinterpret_cast<const char*>(a.pval) + base * (VM == VM_F32 ? 4 : 2);
                // A slot = the quad's next kBpGroup entries; lane l owns the directory pair (first, end) of entry l of the slot and
                // fetches the next slot's pair while the current one is walked.  A lane takes 8 consecutive postings of a list per
                // round (one 16-byte load of document ids, two of values); the kBpBatch lists of a slot are loaded before any
                // multiply-add, so ~32 postings per lane are in flight.
                constexpr int SL = 1;                             // slots per trip (2 for the fp16 copy spills in the walk: measured 2.6x slower),
                constexpr int NB = kBpBatch * SL;                 // lists in flight per quad
                uint32_t nlo[SL], nhi[SL];
#pragma unroll
                for (int sl = 0; sl < SL; ++sl) {
                    const int e = gid + NG * (gl + kBpGroup * sl);
                    nlo[sl] = 0; nhi[sl] = 0;
                    if (e < n_ent) {
                        const uint32_t cc = ent[e].x & 0xFFFFu;
                        nlo[sl] = dirb[cc];
                        nhi[sl] = dirb[cc + 1];
                    }
                }
                for (int j = 0; gid + NG * (kBpGroup * j) < n_ent; j += SL) {
                    uint32_t clo[SL], chi[SL];
#pragma unroll
                    for (int sl = 0; sl < SL; ++sl) {
                        clo[sl] = nlo[sl]; chi[sl] = nhi[sl];
                        nlo[sl] = 0; nhi[sl] = 0;
                        const int e = gid + NG * (gl + kBpGroup * (j + SL + sl));
                        if (e < n_ent) {
                            const uint32_t cc = ent[e].x & 0xFFFFu;
                            nlo[sl] = dirb[cc];
                            nhi[sl] = dirb[cc + 1];
                        }
                    }
                    uint32_t pp[NB], o1[NB];
                    bool more = false;
#pragma unroll
                    for (int u = 0; u < NB; ++u) {                                       // owner lane u % 4 holds the pair of entry u % 4 of slot j + u / 4
                        const uint32_t lo = __shfl(clo[u / kBpGroup], u % kBpGroup, kBpGroup), hi = __shfl(chi[u / kBpGroup], u % kBpGroup, kBpGroup);
                        pp[u] = lo + 8u * gl; o1[u] = hi;
                        more = more || (pp[u] < o1[u]);
                    }
                    // rounds of 32 postings per list; lists longer than one round (popular columns) simply take more rounds
                    while (__builtin_amdgcn_ballot_w64(more)) {
                        uint4 dd[NB];
                        // values stay as loaded (fp32: 8 registers, fp16: 4 packed) until their list is consumed
                        uint4 rv[NB][VM == VM_F32 ? 2 : 1];
#pragma unroll
                        for (int u = 0; u < NB; ++u) {
                            if (pp[u] < o1[u]) {
                                dd[u] = *reinterpret_cast<const uint4*>(bdoc + pp[u] * 2u);
                                if constexpr (VM == VM_F32) {
                                    const float4* vp = reinterpret_cast<const float4*>(bval + pp[u] * 4u);
                                    const float4 v0 = vp[0], v1 = vp[1];
                                    rv[u][0] = make_uint4(__float_as_uint(v0.x), __float_as_uint(v0.y), __float_as_uint(v0.z), __float_as_uint(v0.w));
                                    rv[u][1] = make_uint4(__float_as_uint(v1.x), __float_as_uint(v1.y), __float_as_uint(v1.z), __float_as_uint(v1.w));
                                } else {
                                    rv[u][0] = *reinterpret_cast<const uint4*>(bval + pp[u] * 2u);
                                }
                            }
                        }
                        more = false;
#pragma unroll
                        for (int u = 0; u < NB; ++u) {
                            if (pp[u] < o1[u]) {
                                const uint2 en = ent[gid + NG * (u + kBpGroup * j)];     // (column | slot << 16, weight): one LDS broadcast per quad
                                const uint32_t qo = en.x >> 16;
                                const float wq = __uint_as_float(en.y);
                                const uint32_t nv = min(8u, o1[u] - pp[u]);
                                const uint32_t dw[4] = {dd[u].x, dd[u].y, dd[u].z, dd[u].w};
                                float vv[8];
                                if constexpr (VM == VM_F32) {
                                    vv[0] = __uint_as_float(rv[u][0].x); vv[1] = __uint_as_float(rv[u][0].y); vv[2] = __uint_as_float(rv[u][0].z);
                                    vv[3] = __uint_as_float(rv[u][0].w); vv[4] = __uint_as_float(rv[u][1].x); vv[5] = __uint_as_float(rv[u][1].y);
                                    vv[6] = __uint_as_float(rv[u][1].z); vv[7] = __uint_as_float(rv[u][1].w);
                                } else {
                                    const __half2* h = reinterpret_cast<const __half2*>(&rv[u][0]);
#pragma unroll
                                    for (int t = 0; t < 4; ++t) { const float2 f = __half22float2(h[t]); vv[2 * t] = f.x; vv[2 * t + 1] = f.y; }
                                }
#pragma unroll
                                for (int t = 0; t < 8; ++t) {
                                    const uint32_t d = (t & 1) ? (dw[t >> 1] >> 16) : (dw[t >> 1] & 0xFFFFu);
                                    // past the list's end the 16-byte loads picked up the next list's postings (valid documents of this
                                    // block) or zero padding: those lanes add 0.0 -- no branch per posting
                                    const float vsel = (uint32_t)t < nv ? vv[t] : 0.f;     // select on the fp32 value (one cndmask), not on the double
                                    const float prod = wq * vsel;
                                    atomicAdd(&acc[d * kBpPitch + qo], (double)prod);
                                }
                            }
                            pp[u] += 8u * kBpGroup;
                            more = more || (pp[u] < o1[u]);
                            __builtin_amdgcn_sched_barrier(0);       // one list at a time: 8 fp64 products live, not 32
                        }
                    }
                }
            }
            __syncthreads();
            if (b < b1) {       // one document per thread: its QT sums -> order keys -> candidates
                const int64_t row = b * a.rows + tid;
                double* pa = acc + (size_t)tid * kBpPitch;
                if (tid < a.rows && row < a.n_rows) {
#pragma unroll 1
                    for (int q = 0; q < nq; ++q) {                        // (slots >= nq are never written: a ragged tile skips them;
                        {                                                 //  not unrolled: 8 hoisted candidate-buffer addresses spill into the walk)
                            const double sum = pa[q];
                            pa[q] = 0.0;
                            const uint64_t key = make_key((float)sum, (uint32_t)row);
                            if (key > tau[q] && key < upper_sh[q]) {
                                const uint32_t pos = atomicAdd(&ccnt[q], 1u);
                                my_gcand[(size_t)q * kBpCap + pos] = key;
                            }
                        }
                    }
                }
            }
            __syncthreads();
            const bool last = b + 1 >= b1;
            for (int qs = 0; qs < nq; ++qs) {
                const uint32_t cnt = ccnt[qs];
                if (last || cnt > (uint32_t)(kBpCap - a.rows)) {
                    for (int i = tid; i < kBpCap; i += kScanThreads) sortbuf[i] = (uint32_t)i < cnt ? my_gcand[(size_t)qs * kBpCap + i] : 0ull;
                    wg_sort_desc<kScanThreads>(sortbuf, kBpCap, tid);
                    if (last) {
                        uint64_t* out = a.cand + ((size_t)(q0 + qs) * a.nchunk + c) * (size_t)K;
                        for (int i = tid; i < K; i += kScanThreads) out[i] = sortbuf[i];
                    } else if (cnt > (uint32_t)K) {
                        for (int i = tid; i < K; i += kScanThreads) my_gcand[(size_t)qs * kBpCap + i] = sortbuf[i];
                        if (tid == 0) {
                            tau[qs] = sortbuf[K - 1];
                            ccnt[qs] = (uint32_t)K;
                        }
                    }
                    __syncthreads();
                }
            }
            // (no barrier here: without a prune nothing was written, and the next block's epilogue -- the next writer of
            // ccnt / the candidate buffers -- sits behind the barrier that follows its walk)
            if (last) break;
        }
    }
}

}  // namespace vs
