// bp_bq.h -- "bag-of-token chunks": the column-grouped copy of a BINARY (bag-of-token) index as direct-mapped 32-byte chunks, and its
// walk -- the quad walk (bp_quad.h) re-cut for lists of ~6 postings without values (round 5).
//
// What bounds the record walk (bp_bin.h, 17.5 k q/s at 21 M docs): instructions.  A (tile, block) is 6 208 lists of ~6 postings; per
// list the walk reads a directory word, computes a record address, loads 16 bytes, unpacks 8 ids into 8 addresses, predicates the second
// record of 1 list in 6: ~ 1 700 instructions per wave and block, 2.3 of per-list bookkeeping for every posting (DESIGN r4 8.2).
//
// Here the list of (block b, column c) is chunk c of block b: 32 bytes = 16 cells of uint16 (document in the block); unused cells
// hold one of the 64 spare documents behind the block's 2048 (their sums are never read; spread, so that a wave's pad cells do not pile
// up on one LDS address).  No directory is read at search time: a tile's descriptor table {chunk | plane address << 16, integer
// weight} is built once per work item and serves all its blocks.  A wave step serves EIGHT lists -- an 8-lane group each, a lane
// loads dword i of the chunk (cells 2 i, 2 i + 1) -- with one ds_read_b64, one global_load_dword and 2 x (v_mad_u32_u16, ds_add_u32):
// ~ 10 instructions per 8 lists (tools/gen_bq_asm.py; the loop is one generated asm statement with counted waits).  A list of more
// than 16 postings (1 in 7 000) keeps 15 in its chunk, cell 15 LINKS to an overflow chunk behind the block's main chunks; a wave
// collects the links it meets in a list of its own and walks that after the table (as the quad walk does).
// Copy size: n_blocks x n_cols x 32 bytes (9.7 GB at 21 M docs, V = 29 523) against 3.6 GB of CSR packets.
//
// Same arithmetic as bp_bin_topk: integer weight (the query's weight x its power-of-two scale: exact for dyadic weights) added
// with ds_add_u32 into slot-major int32 planes -- sums, candidates and results are bit-identical (tests/test_gpu_filter.py).
#pragma once
#include "bp_bin.h"
#include "bp_quad.h"
#include "bp_bq_asm.h"

namespace vs {

constexpr int kBqCells = 16, kBqLinked = 15, kBqChunkBytes = 32;
constexpr int kBqPaceDefault = 0;                                     // lock-step window in blocks: off (21 M docs: free running 50.9 ms, window 4: 61.2, 16: 56.8, 32: 52.6)
constexpr int kBqStepDesc = 8;                                        // descriptors (lists) of a wave step
constexpr int kBqTableStep = kScanWaves * kBqStepDesc;               // descriptors of one step of all 16 waves: 128
constexpr uint32_t kBqPlane = (uint32_t)(kBpRowsMaxBin + kBinSpare) * 4u;     // bytes of a slot plane (bp_bin.h: 2048 documents + 66 spare)
static_assert(7u * kBqPlane < 65536u, "plane addresses fit the descriptor's high half");
__host__ __device__ constexpr size_t bq_fixed_lds() { return (size_t)8 * kBqPlane + (size_t)kFlCap * 8 + 8 * 16 + 32 * 4; }
__host__ __device__ constexpr size_t bq_lds_bytes() { return bq_fixed_lds() + (size_t)(kBpEntCap + kBqTableStep + kBqTableStep * kBqOverRead) * 8; }
static_assert(bq_lds_bytes() <= (size_t)160 * 1024, "the chunk walk's LDS");
// a wave's two link lists live in its share of the candidate sort buffer (32 KB / 16 waves = 2 KB): 64 descriptors each, of which
// 8 * kBqOverRead are the null ones a walk over-reads
constexpr int kBqListCap = 64 - kBqStepDesc * kBqOverRead, kBqListBytes = 64 * 8;
static_assert(kBqListCap >= 16 && 2 * kBqListBytes * kScanWaves <= kFlCap * 8, "the link lists fit the sort buffer");
// pad cell of (column c, cell j): one of the 64 spare documents, spread over lanes and neighbouring columns
__host__ __device__ constexpr uint16_t bq_pad(uint32_t c, uint32_t j) { return (uint16_t)(kBpRowsMaxBin + (((c & 7u) << 3) | (j >> 1))); }

// ---- builder: quad_count_kernel<16, 15> (postings per column -> overflow chunks, directory for the fill) -> bp_base_kernel -> fill --------
// fill: one workgroup per block: pad cells first (every main chunk, then the overflow chunks), then the block's non-zeros in arrival
// order, then the links
template <int UNUSED>
__global__ __launch_bounds__(kScanThreads) void bq_fill_kernel(const uint32_t* pk_ptr, const uint4* cols, int64_t n_rows, int32_t n_cols, int32_t rows,
                                                               const uint32_t* dir, const unsigned long long* base, uint16_t* rec) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* cur = reinterpret_cast<uint32_t*>(smem);                  // [n_cols + 1] postings placed so far
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t n_blocks = (n_rows + rows - 1) / rows;
    for (int64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const int64_t r0 = b * rows, r1 = min(n_rows, r0 + rows);
        const uint32_t* d = dir + (size_t)b * (n_cols + 1);
        __syncthreads();
        for (int i = tid; i <= n_cols; i += kScanThreads) cur[i] = 0u;
        uint16_t* brec = rec + (size_t)base[b] * kBqCells;
        const uint32_t n_over = d[n_cols] >> 12;                        // (quad_count_kernel: the block's overflow chunks)
        // pads: 8 cells (16 bytes) per thread and turn
        for (uint32_t i = tid; i < ((uint32_t)n_cols + n_over) * 2u; i += kScanThreads) {
            const uint32_t chunk = i >> 1, j0 = (i & 1u) * 8u;
            uint32_t wds[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) wds[k] = (uint32_t)bq_pad(chunk, j0 + 2 * k) | ((uint32_t)bq_pad(chunk, j0 + 2 * k + 1) << 16);
            reinterpret_cast<uint4*>(brec)[i] = make_uint4(wds[0], wds[1], wds[2], wds[3]);
        }
        __syncthreads();
        for (int64_t r = r0 + w; r < r1; r += kScanWaves) {
            const uint32_t p0 = pk_ptr[r], p1 = pk_ptr[r + 1];
            const uint16_t dl = (uint16_t)(r - r0);
            for (uint32_t p = p0 + lane; p < p1; p += 64) {
                const uint4 cw = cols[p];
                const uint32_t cwv[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t c = (i & 1) ? (cwv[i >> 1] >> 16) : (cwv[i >> 1] & 0xFFFFu);
                    if (c < (uint32_t)n_cols) {
                        const uint32_t pos = atomicAdd(&cur[c], 1u);
                        // chunk k of the list (0: the main chunk = chunk c, k >= 1: overflow chunk first + k - 1) holds 15 postings when
                        // another follows it, up to 16 when it is the last
                        const uint32_t wd = d[c], m = wd & kBpDirRecMask;
                        const uint32_t k = m ? min(pos / (uint32_t)kBqLinked, m) : 0u;
                        const size_t chunk = k == 0 ? (size_t)c : (size_t)n_cols + (wd >> 12) + (k - 1);
                        brec[chunk * kBqCells + (pos - k * (uint32_t)kBqLinked)] = dl;
                    }
                }
            }
        }
        // the links: chunk k of a list with overflow points at its overflow chunk k (index among the block's overflow chunks)
        __syncthreads();
        for (int c = tid; c < n_cols; c += kScanThreads) {
            const uint32_t wd = d[c], m = wd & kBpDirRecMask;
            for (uint32_t k = 0; k < m; ++k) {
                const size_t chunk = k == 0 ? (size_t)c : (size_t)n_cols + (wd >> 12) + (k - 1);
                brec[chunk * kBqCells + 15] = (uint16_t)(0x8000u | ((wd >> 12) + k));
            }
        }
    }
}

// ---- walk -----------------------------------------------------------------------------------------------------------
// Work items, tiles, thresholds and candidate handling as bp_bin_topk (epilogue: a thread finishes documents 2 t and 2 t + 1).
// BpArgs::rec = the chunks, base[b] = first chunk of block b.
template <int TM>          // TM = 1: phase clocks (VS_BP_TIMING)
__global__ __launch_bounds__(kScanThreads) void bp_bq_topk(BpArgs a) {
    constexpr int QT = 8, RMAX = kBpRowsMaxBin;
    static_assert(RMAX == 2 * kScanThreads, "a thread finishes documents 2 t and 2 t + 1");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* acc = reinterpret_cast<int32_t*>(smem);                                        // [QT][RMAX + spare], LDS address 0 (no static LDS in this kernel)
    uint64_t* sortbuf = reinterpret_cast<uint64_t*>(smem + (size_t)QT * kBqPlane);          // [kFlCap]; during a walk: the waves' link lists
    unsigned long long* tau = reinterpret_cast<unsigned long long*>(sortbuf + kFlCap);      // [8]
    unsigned long long* upper_sh = tau + 8;                                                 // [8]
    int* scratch = reinterpret_cast<int*>(upper_sh + 8);                                    // [16]
    unsigned int* ccnt = reinterpret_cast<unsigned int*>(scratch + 16);                     // [16]
    uint2* desc = reinterpret_cast<uint2*>(ccnt + 16);                                      // [n_static + 128 * kBqOverRead]
    const uint32_t desc_lds = (uint32_t)bq_fixed_lds();
    const uint32_t sort_lds = (uint32_t)QT * kBqPlane;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int K = a.k;
    uint64_t* my_gcand = a.gcand + (size_t)blockIdx.x * QT * kFlCap;
    const int64_t n_blocks = (a.n_rows + a.rows - 1) / a.rows;
    const int64_t items = (int64_t)(a.n_tiles_dev ? a.n_tiles_dev[0] : a.n_tiles) * a.nchunk;
    const unsigned long long k_rt0 = TM ? __builtin_amdgcn_s_memrealtime() : 0ull;
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    const uint32_t list_a = sort_lds + (uint32_t)wv * 2u * kBqListBytes, list_b = list_a + kBqListBytes;
    const uint32_t g8 = (uint32_t)(lane >> 3) * 8u, l4 = (uint32_t)(lane & 7) * 4u;

    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int tile = (int)(item / a.nchunk), c = (int)(item % a.nchunk);
        const int q0 = a.tiles[tile].x, nq = a.tiles[tile].y;
        [[maybe_unused]] long long tm = TM ? (long long)__builtin_readcyclecounter() : 0;
        [[maybe_unused]] uint32_t tacc[6] = {0u, 0u, 0u, 0u, 0u, 0u};
        auto lap = [&](int phase) {
            if constexpr (TM != 0) {
                const long long now = (long long)__builtin_readcyclecounter();
                tacc[phase] += (uint32_t)(now - tm);
                tm = now;
            }
        };
        const int64_t b0 = (int64_t)c * a.blocks_per_chunk, b1 = min(n_blocks, b0 + a.blocks_per_chunk);
        __syncthreads();
        const int64_t e0 = a.qptr[q0], e1 = a.qptr[q0 + nq];
        const int n_ent = (int)(e1 - e0);
        const int n_static = (n_ent + kBqTableStep - 1) / kBqTableStep * kBqTableStep;
        // the tile's entries sorted by column (the accumulator area doubles as the sort buffer) -> the descriptor table
        {
            uint64_t* skey = reinterpret_cast<uint64_t*>(acc);
            for (int i = tid; i < 8192; i += kScanThreads) {
                uint64_t key = 0;
                if (i < n_ent) {
                    const int64_t e = e0 + i;
                    int qs = 0;
                    while (e >= a.qptr[q0 + qs + 1]) ++qs;
                    const float w = a.qvals[e] * a.qscale[q0 + qs];               // power of two: exact
                    key = ((uint64_t)1 << 63) | ((uint64_t)(0xFFFFu - (uint32_t)a.qcols[e]) << 40) | ((uint64_t)qs << 32) | (uint64_t)__float_as_uint(w);
                }
                skey[i] = key;
            }
            wg_sort_desc<kScanThreads>(skey, 8192, tid);
            for (int i = tid; i < n_static + kBqTableStep * kBqOverRead; i += kScanThreads) {
                uint2 dsc = make_uint2(0u, 0u);                                   // null: chunk 0, plane 0, weight 0
                if (i < n_ent) {
                    const uint64_t key = skey[i];
                    const uint32_t col = 0xFFFFu - ((uint32_t)(key >> 40) & 0xFFFFu), qs = (uint32_t)(key >> 32) & 0xFFu;
                    dsc = make_uint2(col | ((qs * kBqPlane) << 16), (uint32_t)(int32_t)__uint_as_float((uint32_t)key));       // integer weight (bp_bin.h)
                }
                desc[i] = dsc;
            }
            __syncthreads();
        }
        for (int i = tid; i < (int)(QT * kBqPlane / 4); i += kScanThreads) acc[i] = 0;
        if (tid < 8) { tau[tid] = 0ull; ccnt[tid] = 0u; upper_sh[tid] = (a.upper && tid < nq) ? a.upper[q0 + tid] : ~0ull; }
        __syncthreads();
        const uint32_t trips = (uint32_t)(n_static / kBqTableStep);
        bool pace_off = false;
        unsigned long long base_cur = b0 < b1 ? a.base[b0] : 0ull;
        lap(0);
        for (int64_t b = b0; b < b1 || b == b0; ++b) {
            const bool have = b < b1;
            const int rows_b = have ? (int)min((int64_t)a.rows, a.n_rows - b * a.rows) : 0;
            if (have && trips > 0) {
                const char* brec = a.rec + (size_t)base_cur * kBqChunkBytes;
                // the overflow chunks a walk found: the wave's list `cur` holds n of them; their own links go to the other list
                auto chain = [&](uint32_t n, uint32_t cur, uint32_t nxt) {
                    // (a list has at most RMAX / 15 + 1 chunks: the bound keeps a corrupt link from hanging the GPU)
                    for (int depth = 0; n > 0 && depth < RMAX / kBqLinked + 2; ++depth) {
                        const uint32_t n_pad = (n + 7u) & ~7u;
                        if ((uint32_t)lane < n_pad - n + 8u * kBqOverRead)
                            reinterpret_cast<uint2*>(smem + cur)[n + (uint32_t)lane] = make_uint2(0u, 0u);      // null descriptors behind the list (smem = LDS address 0)
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        n = bq_list_asm(cur + g8, n_pad / 8u, brec, l4, (uint32_t)a.n_cols, nxt, (uint32_t)kBqListCap);
                        const uint32_t t = cur; cur = nxt; nxt = t;
                    }
                };
                const uint32_t n_link = bq_walk_asm(desc_lds + (uint32_t)wv * 64u + g8, trips, brec, l4, (uint32_t)a.n_cols, list_a, (uint32_t)kBqListCap);
                if (n_link <= (uint32_t)kBqListCap) {
                    chain(n_link, list_a, list_b);
                } else {
                    // more links than the list holds (a tile of very long lists): the table's chunks are added, their links are collected
                    // again, as many steps at a time as the list has room for
                    constexpr uint32_t kSeg = kBqListCap / 8;
                    for (uint32_t t0 = 0; t0 < trips; t0 += kSeg) {
                        const uint32_t n = bq_collect_asm(desc_lds + (uint32_t)wv * 64u + g8 + t0 * 1024u, min(kSeg, trips - t0), brec, l4, (uint32_t)a.n_cols, list_a, (uint32_t)kBqListCap);
                        chain(n, list_a, list_b);
                    }
                }
            }
            lap(1);
            if (a.pace && items <= (int64_t)gridDim.x && tid == 0 && have && !pace_off) {       // lock step (bp_walk.h)
                uint32_t* pc = a.pace + (size_t)c * a.blocks_per_chunk;
                const int64_t rel = b - b0;
                if (!((a.knob & 64) && blockIdx.x == 0))              // (VS_BP_KNOB=64, tests: workgroup 0 never reports -- every peer's wait must time out, not hang)
                    __hip_atomic_fetch_add(pc + rel, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (rel >= a.pace_window) {
                    const uint32_t need = (uint32_t)(items / a.nchunk);
                    if (!pace_wait(pc + rel - a.pace_window, need)) pace_off = true;
                }
            }
            if (b + 1 < b1) base_cur = a.base[b + 1];
            if (a.gtau && tid < nq) { const unsigned long long g = a.gtau[q0 + tid]; if (g > tau[tid]) tau[tid] = g; }
            lds_barrier();                                               // the block's sums are complete
            lap(2);
            {
                const int d = 2 * tid;
                uint32_t thi[QT];
                {
                    const uint4* t4 = reinterpret_cast<const uint4*>(tau);
#pragma unroll
                    for (int i = 0; i < QT / 2; ++i) { const uint4 t = t4[i]; thi[2 * i] = t.y; thi[2 * i + 1] = t.w; }
                }
                if (d < rows_b) {
                    const int64_t row = (int64_t)b * a.rows + d;
                    uint2 sums[QT];
#pragma unroll
                    for (int q = 0; q < QT; ++q) sums[q] = *reinterpret_cast<const uint2*>(acc + q * (RMAX + kBinSpare) + d);
#pragma unroll
                    for (int q = 0; q < QT; ++q) *reinterpret_cast<uint2*>(acc + q * (RMAX + kBinSpare) + d) = make_uint2(0u, 0u);
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        const uint32_t h0 = sums[q].x ^ 0x80000000u, h1 = sums[q].y ^ 0x80000000u;
                        if (q < nq && (h0 >= thi[q] || h1 >= thi[q])) {
                            const uint64_t k0 = ((uint64_t)h0 << 32) | (uint32_t)(~(uint32_t)row);
                            const uint64_t k1 = ((uint64_t)h1 << 32) | (uint32_t)(~(uint32_t)(row + 1));
                            const unsigned long long tq = tau[q], uq = upper_sh[q];
                            if (k0 > tq && k0 < uq) {
                                const uint32_t pos = atomicAdd(&ccnt[q], 1u);
                                my_gcand[(size_t)q * kFlCap + pos] = k0;
                            }
                            if (d + 1 < rows_b && k1 > tq && k1 < uq) {
                                const uint32_t pos = atomicAdd(&ccnt[q], 1u);
                                my_gcand[(size_t)q * kFlCap + pos] = k1;
                            }
                        }
                    }
                }
                lds_barrier();
                const bool last = b + 1 >= b1;
                uint32_t cnts[QT];
                {
                    const uint4* c4 = reinterpret_cast<const uint4*>(ccnt);
                    const uint4 c0 = c4[0], c1 = c4[1];
                    cnts[0] = c0.x; cnts[1] = c0.y; cnts[2] = c0.z; cnts[3] = c0.w; cnts[4] = c1.x; cnts[5] = c1.y; cnts[6] = c1.z; cnts[7] = c1.w;
                }
                bool any = last;
#pragma unroll
                for (int q = 0; q < QT; ++q) any = any || cnts[q] > (uint32_t)(kFlCap - RMAX);
                if (any) __syncthreads();                        // (the candidates pushed above are read back: a full barrier)
                if (any)
                for (int qs = 0; qs < nq; ++qs) {
                    const uint32_t cn = ccnt[qs];
                    if (last || cn > (uint32_t)(kFlCap - RMAX)) {
                        for (int i = tid; i < kFlCap; i += kScanThreads) sortbuf[i] = (uint32_t)i < cn ? my_gcand[(size_t)qs * kFlCap + i] : 0ull;
                        wg_sort_desc<kScanThreads>(sortbuf, kFlCap, tid);
                        if (last) {
                            uint64_t* out = a.cand + ((size_t)(q0 + qs) * a.nchunk + c) * (size_t)K;
                            for (int i = tid; i < K; i += kScanThreads) out[i] = sortbuf[i];
                        } else if (cn > (uint32_t)K) {
                            for (int i = tid; i < K; i += kScanThreads) my_gcand[(size_t)qs * kFlCap + i] = sortbuf[i];
                            if (tid == 0) {
                                const unsigned long long kth = sortbuf[K - 1];
                                if (kth > tau[qs]) tau[qs] = kth;
                                if (a.gtau && kth != 0ull) atomicMax(a.gtau + q0 + qs, kth);
                                ccnt[qs] = (uint32_t)K;
                            }
                        }
                        __syncthreads();
                    }
                }
            }
            lap(4);
            if constexpr (TM != 0) tacc[5] += 1u;
            if (b + 1 >= b1) break;
        }
        if constexpr (TM != 0) {
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 6; ++i) atomicAdd(a.timing + i, (unsigned long long)tacc[i]);
            }
        }
    }
    if (TM && threadIdx.x == 0) {
        a.timing[16 + 4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime() - k_rt0;
        a.timing[17 + 4 * blockIdx.x] = k_rt0;
        a.timing[18 + 4 * blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20);
        a.timing[19 + 4 * blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}

}  // namespace vs
