// bp_quad.h -- the "quad" postings copy and its walk: the default filter walk of a valued index (round 4).
//
// What round 3's list walk (bp_walk.h) was bound by: VALU issue.  profiles/r04_conflicts.txt: with every ds_add_u32 made
// conflict-free (a throw-away build) the LDS goes from 68 % to 28 % busy and the kernel time does not move; SQ_ACTIVE_INST_VALU is
// 79 % of all SIMD cycles -- 7.5 VALU instructions per ds_add, of which 3 do the work (multiply, truncate, address) and the rest
// manage list lengths, tails, per-lane predicates and the directory words of six lists in flight.
//
// Here a list is stored in CHUNKS of 256 bytes = 16 lanes x 4 postings, zero-padded; a posting is ONE dword (accumulator index of
// the document | fp16 value << 16); a wave step serves four chunks (one per 16-lane group) with one ds_read_b64 (descriptor), one
// global_load_dwordx4 and 4 x (v_fma_mix_f32, v_cvt_i32_f32, v_mad_u32_u16, ds_add_u32): 15 VALU instructions per step, nothing
// predicated (bp_quad_loop.h has the layout, tools/gen_quad_asm.py the loop).  The main chunk of column c is chunk c of every block, so
// a tile's descriptor table {column x 256 | slot row, weight} is built ONCE per work item and serves all its blocks; the waves take
// steps w, w + 16, ... of it: equal work by construction (the list walk needed a dynamic queue), and no directory is read at search
// time.  A list longer than its chunk goes on in an overflow chunk of its block that the chunk itself links to (its last two cells);
// a wave collects the links it meets in a small list of its own and walks that right after the table.  The builder deals the postings
// of a chunk so that the 16 a list adds in one instruction fall into 16 different LDS banks; two lists share a 32-lane half, so a
// bank takes at most 2 lanes = the cost of a conflict-free atomic (tools/microbench/lds_conflicts.hip).
// Measured on the inner loop alone (tools/microbench/quad_walk.hip): 6.3 - 6.7 cycles per chunk and CU against 8.9 per list
// visit of the list walk; unarranged chunks: 9.2.
//
// Same arithmetic as bp_walk_topk<VM_F16, 8, AM_FIX, ...>: fp32 weight x fp16 value in one v_fma_mix_f32, truncated, int32 sums --
// the candidate sets, and with them every result, are bit-identical (tests/test_gpu_filter.py runs both against the CSR scan).
#pragma once
#include "bp_walk.h"
#include "bp_quad_loop.h"
#include "bp_quad_asm.h"

#ifndef VS_QUAD_EPI
#define VS_QUAD_EPI 2          // how the epilogue reads and zeroes a document's sums: 0 = 8 reads + 8 writes, 1 = 8 ds_wrxchg_rtn_b32, 2 = 4 ds_wrxchg2_rtn_b32
#endif

namespace vs {

constexpr int kQuadRows = 2048;                                   // documents per block
constexpr int kQuadQT = 8;                                        // query slots per tile
constexpr int kQuadPer = kBpEntCap / kScanThreads;                // entries a thread owns: 7
constexpr size_t kQuadAccBytes = (size_t)(kQuadRows / 16) * kQuadGroupDw * 4;      // 73 728: accumulators, at LDS address 0
static_assert(kBpEntCap % kScanThreads == 0, "a thread owns a fixed number of entries");
static_assert(kQuadAccBytes >= kBpSortBytes, "the accumulator area holds the 8192-slot entry sort");
// LDS: accumulators | candidate sort buffer (during a walk: the waves' link lists) | thresholds, bounds, scratch, counters | descriptors
// (+ the null steps the loop over-reads)
__host__ __device__ constexpr size_t quad_fixed_lds() { return kQuadAccBytes + (size_t)kBpCap * 8 + (size_t)kQuadQT * 16 + 64 * 4; }
__host__ __device__ constexpr size_t quad_cut_lds() { return quad_fixed_lds() + (size_t)(kBpEntCap + 64 + 64 * kQuadOverRead) * 8; }        // the cut's histograms (wg_cut_topk), behind the table
__host__ __device__ constexpr size_t quad_lds_bytes() { return quad_cut_lds() + (size_t)kCutHistWords * 4; }
static_assert(quad_lds_bytes() <= (size_t)160 * 1024, "the quad walk's LDS");
static_assert(kBpCap == 2 * kScanThreads, "wg_cut_topk: two keys of the candidate buffer a thread");
static_assert(kQuadQT == 8 && (kQuadAccBytes + (size_t)kBpCap * 8) % 16 == 0, "the epilogue reads thresholds and counters 16 bytes at a time");
// a wave's two link lists live in its 1 KB of the sort buffer: 64 descriptors each, of which 4 * kQuadOverRead are the null ones a walk over-reads
constexpr int kQuadListCap = 64 - 4 * kQuadOverRead, kQuadListBytes = (kQuadListCap + 4 * kQuadOverRead) * 8;
static_assert(kQuadListCap >= 16, "room for links in a wave's list");
static_assert(2 * kQuadListBytes * kScanWaves <= kBpCap * 8, "the link lists fit the sort buffer");
// postings a chunk holds when its list goes on in another chunk (the last two cells are the link), overflow chunks of a list of n postings
constexpr int kQuadLinked = kQuadCells - 2;
constexpr double kQuadMaxRatio = 3.0;                              // auto policy: quad chunks while their main area is within this multiple of the CSR bytes (bp_build)
constexpr int kQuadPaceDefault = 0;                                // lock-step window in blocks (see the walk): off since the work items take FOUR block chunks
                                                                   // (round 5: free running 67.0 / 36.5 / 7.57 ms against 69.8 / 37.7 / 7.71 in lock step, 21 M docs B = 512 / 256, 1 M docs B = 1024)
// overflow chunks of a list of n postings: a chunk holds CELLS postings when it is the list's last, LINKED when another follows it
template <int CELLS, int LINKED>
__host__ __device__ constexpr uint32_t chunk_overflow(uint32_t n) { return n > (uint32_t)CELLS ? (n - (uint32_t)CELLS + (uint32_t)LINKED - 1u) / (uint32_t)LINKED : 0u; }
__host__ __device__ constexpr uint32_t quad_overflow_chunks(uint32_t n) { return chunk_overflow<kQuadCells, kQuadLinked>(n); }

// ---- builder --------------------------------------------------------------------------------------------------------
// A block's chunks: one MAIN chunk per column at chunk index = column -- a tile's descriptors for them are the same in every block,
// no directory is read at search time -- then the block's OVERFLOW chunks.  A list of more than 64 postings keeps 62 in its main
// chunk; the chunk's last two cells are a LINK to the overflow chunk that continues it (and so on): the last cell has the sign bit
// set -- postings have non-negative values -- and 14 payload bits, the cell before it 14 more, both with value 0 (as postings they
// add nothing, to a valid accumulator).  19 lists in 20 end in their main chunk on 768-nnz documents.
// pass 1, one workgroup per block: postings per column -> overflow chunks, directory (first overflow chunk << 12 | overflow chunks:
// the fill pass needs it, the search does not), block total (V + overflow chunks)
// (CELLS, LINKED: 64, 62 = quad chunks of a valued index; 16, 15 = the 32-byte chunks of a bag-of-token index, bp_bq.h)
template <int CELLS, int LINKED>
__global__ __launch_bounds__(kScanThreads) void quad_count_kernel(const uint32_t* pk_ptr, const uint4* cols, int64_t n_rows, int32_t n_cols, int32_t rows,
                                                                  uint32_t* dir, uint32_t* block_recs, unsigned long long* df_rec, unsigned long long* df_nnz,
                                                                  int32_t* overflow) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem);                  // [n_cols + 1]
    __shared__ int scratch[32];
    const int tid = threadIdx.x;
    const int64_t n_blocks = (n_rows + rows - 1) / rows;
    const int seg = (n_cols + kScanThreads - 1) / kScanThreads;
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
                atomicAdd(&cnt[cwv[i] & 0xFFFFu], 1u);                  // row padding lands in cnt[n_cols], never read
                atomicAdd(&cnt[cwv[i] >> 16], 1u);
            }
        }
        __syncthreads();
        const int i0 = min(n_cols, tid * seg), i1 = min(n_cols, i0 + seg);
        int mine = 0;
        for (int i = i0; i < i1; ++i) mine += (int)chunk_overflow<CELLS, LINKED>(cnt[i]);
        int tot = 0;
        int off = block_excl_scan(mine, scratch, tid, &tot);
        uint32_t* d = dir + (size_t)b * (n_cols + 1);
        for (int i = i0; i < i1; ++i) {
            const uint32_t c = cnt[i], r = chunk_overflow<CELLS, LINKED>(c);
            d[i] = bp_dir_pack((uint32_t)off, r);
            if (r > kBpDirRecMask || (uint32_t)off > kBpDirUnitMax) overflow[0] = 1;
            off += (int)r;
            atomicAdd(&df_rec[i], (unsigned long long)(1u + r));        // chunks an entry on this column reads in this block (the main chunk even when empty)
            if (c) atomicAdd(&df_nnz[i], (unsigned long long)c);
        }
        if (tid == 0) { d[n_cols] = bp_dir_pack((uint32_t)tot, 0u); block_recs[b] = (uint32_t)n_cols + (uint32_t)tot; }
    }
}

// fill: scatter the block's non-zeros into their lists' cells in arrival order (the array is zero-filled: unused cells add nothing)
template <int VS>
__global__ __launch_bounds__(kScanThreads) void quad_fill_kernel(const uint32_t* pk_ptr, const uint4* cols, const void* vals, int64_t n_rows, int32_t n_cols,
                                                                 int32_t rows, const uint32_t* dir, const unsigned long long* base, uint32_t* rec) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* cur = reinterpret_cast<uint32_t*>(smem);                  // [n_cols + 1] postings placed so far
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t n_blocks = (n_rows + rows - 1) / rows;
    for (int64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const int64_t r0 = b * rows, r1 = min(n_rows, r0 + rows);
        const uint32_t* d = dir + (size_t)b * (n_cols + 1);
        __syncthreads();
        for (int i = tid; i <= n_cols; i += kScanThreads) cur[i] = 0u;
        __syncthreads();
        uint32_t* brec = rec + (size_t)base[b] * kQuadCells;
        for (int64_t r = r0 + w; r < r1; r += kScanWaves) {
            const uint32_t p0 = pk_ptr[r], p1 = pk_ptr[r + 1];
            const uint32_t ai = quad_acc_index((uint32_t)(r - r0));
            for (uint32_t p = p0 + lane; p < p1; p += 64) {
                const uint4 cw = cols[p];
                const uint32_t cwv[4] = {cw.x, cw.y, cw.z, cw.w};
                uint32_t hb[8];                                         // fp16 bits of the 8 values
                if constexpr (VS == VM_F32) {
                    const float4* vp = reinterpret_cast<const float4*>(vals);
                    const float4 v0 = vp[2 * (size_t)p], v1 = vp[2 * (size_t)p + 1];
                    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) hb[i] = (uint32_t)__half_as_ushort(__float2half_rn(v[i]));
                } else {
                    const uint4 hv = reinterpret_cast<const uint4*>(vals)[p];
                    const uint32_t h4[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) hb[i] = (i & 1) ? (h4[i >> 1] >> 16) : (h4[i >> 1] & 0xFFFFu);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t c = (i & 1) ? (cwv[i >> 1] >> 16) : (cwv[i >> 1] & 0xFFFFu);
                    if (c < (uint32_t)n_cols) {
                        const uint32_t pos = atomicAdd(&cur[c], 1u);
                        // chunk k of the list (0: the main chunk = chunk c, k >= 1: overflow chunk first + k - 1 behind the main chunks)
                        // holds 62 postings when another follows it, up to 64 when it is the last
                        const uint32_t w = d[c], m = w & kBpDirRecMask;
                        const uint32_t k = m ? min(pos / (uint32_t)kQuadLinked, m) : 0u;
                        const size_t chunk = k == 0 ? (size_t)c : (size_t)n_cols + (w >> 12) + (k - 1);
                        const uint32_t hv = (hb[i] & 0x7FFFu) ? hb[i] : 0u;       // (-0 -> +0: the sign bit marks a link)
                        brec[chunk * kQuadCells + (pos - k * (uint32_t)kQuadLinked)] = ai | (hv << 16);
                    }
                }
            }
        }
        // the links: chunk k of a list with overflow points at its overflow chunk k
        __syncthreads();
        for (int c = tid; c < n_cols; c += kScanThreads) {
            const uint32_t w = d[c], m = w & kBpDirRecMask;
            for (uint32_t k = 0; k < m; ++k) {
                const size_t chunk = k == 0 ? (size_t)c : (size_t)n_cols + (w >> 12) + (k - 1);
                const uint32_t link = (uint32_t)n_cols + (w >> 12) + k;
                brec[chunk * kQuadCells + 62] = (link >> 14) & 0x3FFFu;
                brec[chunk * kQuadCells + 63] = 0x80000000u | (link & 0x3FFFu);
            }
        }
    }
}

// arrange: inside every chunk, deal the postings to the 16 x 4 grid (lane l, posting j -> dword 4 l + j) so that the 16 postings of
// a column j -- what the chunk's lanes add in ONE ds_add_u32 -- sit in 16 different LDS banks (bank = accumulator index mod 32 =
// document mod 32, whatever the slot).  Greedy: a posting goes to the emptiest column that does not hold its bank yet; when every
// column with room holds it (a bank with 5+ postings in the chunk) the conflict stays.  Unused cells (value 0) get the accumulator
// index of a document whose bank the column lacks: they add 0 to 16 more different banks.  A chunk is self-contained: one thread per
// chunk, 256 consecutive chunks (64 KB) staged through LDS with coalesced loads and stores.
template <int UNUSED>
__global__ __launch_bounds__(256) void quad_arrange_kernel(uint32_t* rec, unsigned long long n_chunks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LD = kQuadCells + 1;                                  // (odd pitch: the threads' cells fall into different banks)
    uint32_t* in = reinterpret_cast<uint32_t*>(smem);                   // [256][65]
    uint32_t* out = in + 256 * LD;                                      // [256][65]
    const int tid = threadIdx.x;
    for (unsigned long long c0 = (unsigned long long)blockIdx.x * 256; c0 < n_chunks; c0 += (unsigned long long)gridDim.x * 256) {
        const int nc = (int)min((unsigned long long)256, n_chunks - c0);
        const uint4* src = reinterpret_cast<const uint4*>(rec + c0 * kQuadCells);
        __syncthreads();
        for (int i = tid; i < nc * 16; i += 256) {
            const uint4 v = src[i];
            uint32_t* q = in + (i >> 4) * LD + (i & 15) * 4;
            q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
        }
        __syncthreads();
        if (tid < nc) {
            const uint32_t* mi = in + tid * LD;
            uint32_t* mo = out + tid * LD;
            uint32_t mask[4] = {0u, 0u, 0u, 0u}, fill[4] = {0u, 0u, 0u, 0u}, room[4] = {16u, 16u, 16u, 16u};
            const bool linked = (mi[63] >> 31) != 0u;                   // the last two cells are a link: they stay where they are
            if (linked) {
                mo[62] = mi[62]; mo[63] = mi[63];
                mask[2] |= 1u << (mi[62] & 31u); mask[3] |= 1u << (mi[63] & 31u);
                room[2] = 15u; room[3] = 15u;
            }
            for (int i = 0; i < (linked ? kQuadLinked : kQuadCells); ++i) {
                const uint32_t p = mi[i];
                if ((p & 0x7FFF0000u) == 0u) continue;                  // unused cell, or an explicit zero: adds nothing either way
                const uint32_t bit = 1u << (p & 31u);
                int pick = -1, any = -1;
                uint32_t best = 17u, best_any = 17u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (fill[j] < room[j] && fill[j] < best_any) { best_any = fill[j]; any = j; }
                    if (fill[j] < room[j] && !(mask[j] & bit) && fill[j] < best) { best = fill[j]; pick = j; }
                }
                if (pick < 0) pick = any;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j == pick) { mo[fill[j] * 4 + j] = p; mask[j] |= bit; ++fill[j]; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                while (fill[j] < room[j]) {
                    const uint32_t freeb = ~mask[j];
                    const uint32_t bk = freeb ? (uint32_t)(__ffs((int)freeb) - 1) : 0u;
                    mo[fill[j] * 4 + j] = quad_acc_index(bk);           // value 0
                    mask[j] |= 1u << bk;
                    ++fill[j];
                }
        }
        __syncthreads();
        uint4* dst = reinterpret_cast<uint4*>(rec + c0 * kQuadCells);
        for (int i = tid; i < nc * 16; i += 256) {
            const uint32_t* q = out + (i >> 4) * LD + (i & 15) * 4;
            dst[i] = make_uint4(q[0], q[1], q[2], q[3]);
        }
    }
}

// ---- walk -----------------------------------------------------------------------------------------------------------
// Work items, tiles, candidate handling and thresholds as bp_walk_topk (AM_FIX, 8 slots).  BpArgs::rec = the chunks, base[b] = first
// chunk of block b.
//
// Per item: the tile's entries are sorted by column and become the descriptor table {column x 256 | slot row, weight} -- the main chunk
// of column c is chunk c of EVERY block, so the table serves all blocks of the item and nothing is planned per block.  Per block: the
// waves walk the table (quad_walk_asm: wave w takes steps w, w + 16, ...); chunks that link to an overflow chunk leave a descriptor in
// the wave's own list, which the wave walks right after (quad_list_asm), and so on down the chain; one LDS barrier; the epilogue turns
// the block's sums into candidates.  The only global loads of a block are its chunks (and one scalar load of the next block's base).
typedef unsigned short quad_us2 __attribute__((ext_vector_type(2)));

template <int TM>          // TM = 1: phase clocks (VS_BP_TIMING)
__global__ __launch_bounds__(kScanThreads) void bp_quad_topk(BpArgs a) {
    constexpr int QT = kQuadQT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* acc = reinterpret_cast<int32_t*>(smem);                                        // LDS address 0 (no static LDS in this kernel)
    uint64_t* sortbuf = reinterpret_cast<uint64_t*>(smem + kQuadAccBytes);                  // [kBpCap]; during a walk: the link lists
    unsigned long long* tau = reinterpret_cast<unsigned long long*>(sortbuf + kBpCap);      // [QT]
    unsigned long long* upper_sh = tau + QT;                                                // [QT]
    int* scratch = reinterpret_cast<int*>(upper_sh + QT);                                   // [48]
    unsigned int* ccnt = reinterpret_cast<unsigned int*>(scratch + 48);                     // [QT]
    unsigned int* chi = ccnt + QT;                                                          // [QT] the counters' high halves at the end of the previous block (epilogue)
    uint2* desc = reinterpret_cast<uint2*>(scratch + 64);                                   // [n_static + 64 * kQuadOverRead]
    const uint32_t desc_lds = (uint32_t)quad_fixed_lds();                                   // its LDS byte address
    uint32_t* cut_hist = reinterpret_cast<uint32_t*>(smem + quad_cut_lds());                // [kCutHistWords]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int K = a.k;
    uint64_t* my_gcand = a.gcand + (size_t)blockIdx.x * QT * kBpCap;
    const int64_t n_blocks = (a.n_rows + a.rows - 1) / a.rows;
    const int64_t items = (int64_t)(a.n_tiles_dev ? a.n_tiles_dev[0] : a.n_tiles) * a.nchunk;
    const unsigned long long k_rt0 = TM ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // a barrier that orders LDS only (a __syncthreads() also waits for every global access in flight)
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // the wave's two link lists
    const uint32_t list_a = (uint32_t)kQuadAccBytes + (uint32_t)wv * 2u * kQuadListBytes, list_b = list_a + kQuadListBytes;
    const uint32_t g8 = (uint32_t)(lane >> 4) * 8u, s16 = (uint32_t)(lane & 15) * 16u;

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
        const int n_static = (n_ent + 63) & ~63;
        // the tile's entries sorted by column (the accumulator area doubles as the sort buffer) -> the descriptor table
        {
            uint64_t* skey = reinterpret_cast<uint64_t*>(acc);
            for (int i = tid; i < 8192; i += kScanThreads) {
                uint64_t key = 0;
                if (i < n_ent) {
                    const int64_t e = e0 + i;
                    int qs = 0;
                    while (e >= a.qptr[q0 + qs + 1]) ++qs;
                    const float w = a.qvals[e] * a.qscale[q0 + qs];                    // power of two: exact
                    key = ((uint64_t)1 << 63) | ((uint64_t)(uint32_t)a.qcols[e] << 40) | ((uint64_t)qs << 32) | (uint64_t)__float_as_uint(w);
                }
                skey[i] = key;
            }
            wg_sort_desc<kScanThreads>(skey, 8192, tid);
            for (int i = tid; i < n_static + 64 * kQuadOverRead; i += kScanThreads) {
                uint2 dsc = make_uint2(0u, 0u);                                         // null: chunk 0, weight 0
                if (i < n_ent) {
                    const uint64_t key = skey[i];
                    const uint32_t col = (uint32_t)(key >> 40) & 0xFFFFu, qs = (uint32_t)(key >> 32) & 0xFFu;
                    dsc = make_uint2((col << 8) | (qs * 16u), (uint32_t)key);
                }
                desc[i] = dsc;
            }
            __syncthreads();
        }
        for (int i = tid; i < (int)(kQuadAccBytes / 4); i += kScanThreads) acc[i] = 0;
        if (tid < QT) { tau[tid] = 0ull; ccnt[tid] = 0u; chi[tid] = 0u; }
        if (tid < QT) upper_sh[tid] = (a.upper && tid < nq) ? a.upper[q0 + tid] : ~0ull;
        __syncthreads();
        const uint32_t trips = (uint32_t)(n_static / 64);
        bool pace_off = false;                                           // (thread 0's: the lock step timed out once)
        // (the first chunk of a block is fetched a block ahead: s_waitcnt lgkmcnt(0) -- every LDS barrier -- waits for scalar loads too)
        unsigned long long base_cur = b0 < b1 ? a.base[b0] : 0ull;
        for (int64_t b = b0; b < b1 || b == b0; ++b) {
            const bool have = b < b1;
            const int rows_b = have ? (int)min((int64_t)a.rows, a.n_rows - b * a.rows) : 0;
            if (have && trips > 0) {
                const char* brec = a.rec + (size_t)base_cur * kQuadChunkBytes;
                // the overflow chunks a walk found: the wave's list `cur` holds n of them; their own links go to the other list
                auto chain = [&](uint32_t n, uint32_t cur, uint32_t nxt) {
                    // (a list has at most kQuadRows / 62 + 1 chunks: the bound keeps a corrupt link from hanging the GPU)
                    for (int depth = 0; n > 0 && depth < kQuadRows / kQuadLinked + 2; ++depth) {
                        const uint32_t n_pad = (n + 3u) & ~3u;
                        if ((uint32_t)lane < n_pad - n + 4u * kQuadOverRead)
                            reinterpret_cast<uint2*>(smem + cur)[n + (uint32_t)lane] = make_uint2(0u, 0u);      // null descriptors behind the list (smem = LDS address 0)
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        n = quad_list_asm(cur + g8, n_pad / 4u, brec, s16, nxt, (uint32_t)kQuadListCap);       // (n links out of n chunks at most: they fit)
                        const uint32_t t = cur; cur = nxt; nxt = t;
                    }
                };
                const uint32_t n_link = quad_walk_asm(desc_lds + (uint32_t)wv * 32u + g8, trips, brec, s16, list_a, (uint32_t)kQuadListCap);
                if (n_link <= (uint32_t)kQuadListCap) {
                    chain(n_link, list_a, list_b);
                } else {
                    // more links than the list holds (a tile of very long lists): the postings of the table's chunks are added, their links
                    // are collected again, as many steps at a time as the list has room for
                    constexpr uint32_t kSeg = kQuadListCap / 4;
                    for (uint32_t t0 = 0; t0 < trips; t0 += kSeg) {
                        const uint32_t n = quad_collect_asm(desc_lds + (uint32_t)wv * 32u + g8 + t0 * 512u, min(kSeg, trips - t0), brec, s16, list_a, (uint32_t)kQuadListCap);
                        chain(n, list_a, list_b);
                    }
                }
            }
            lap(1);
            // Lock step (BpArgs::pace): the tiles of a chunk of blocks sweep the same blocks; a workgroup that falls behind the pack loses
            // the L2 / Infinity Cache copies the others left behind and falls further behind (21 M docs, TWO block chunks: most workgroups
            // take 133 ms, a handful 165 - 170, and the launch ends with the last; with the four chunks of round 5 free running is stable
            // and faster -- the option "postings_pace" still turns the lock step on).  An item counts its arrival at the end of block j and waits while
            // the slowest item of its chunk has not reached block j - window (bounded: pace_wait, bp_walk.h).  Window 2 - 8 blocks: 141 - 143 ms,
            // 16: 152, 32: 161, free running: 134 - 166 (run to run).
            if (a.pace && items <= (int64_t)gridDim.x && tid == 0 && have && !pace_off) {
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
            // The NEXT block's cold start: the epilogue below drains every load a wave had in flight, and the next walk's first steps would
            // each wait for a line that leaves the L2.  One dword per lane from the 64 lines of this wave's first 8 steps (lane l: step
            // l >> 3, lane group (l >> 1) & 3, half l & 1 of the 256-byte chunk) brings them into the XCD's L2 while the epilogue runs;
            // the value is never used (the register is held until the loads are back: the compiler does not know it is a load's).
            // (measured, 21 M docs: 8 steps 127.2 ms, 4 steps 129.1, 16 steps 131.3 -- 8 steps of 32 workgroups are the XCD's 4 MB --, none 130.2;
            //  VS_BP_KNOB=512 turns it off)
            uint32_t pf = 0;
            if ((a.knob & 512) == 0 && b + 1 < b1 && (uint32_t)(lane >> 3) < trips) {
                const uint32_t dx = *reinterpret_cast<const uint32_t*>(smem + desc_lds + (uint32_t)(lane >> 3) * 512u + (uint32_t)wv * 32u + (uint32_t)((lane >> 1) & 3) * 8u);
                const char* pl = a.rec + (size_t)base_cur * kQuadChunkBytes + (dx & 0xFFFFFF00u) + (uint32_t)(lane & 1) * 128u;
                asm volatile("global_load_dword %0, %1, off" : "=v"(pf) : "v"(pl) : "memory");
            }
            lap(2);
            // epilogue: 1024 documents at a time, one per thread: its QT sums -> order keys -> candidates; prune when a buffer could overflow
            int32_t thr[QT];
            bool thr_stale = true;
            for (int d0 = 0; d0 < rows_b || d0 == 0; d0 += kScanThreads) {
                const int d = d0 + tid;
                const bool more = d0 + kScanThreads < rows_b;             // another round of this block follows
                const uint32_t inc = more ? 1u : 0x10000u;
                // (LDS instructions are what the epilogue costs -- 16 waves x 2 rounds a block: the 8 threshold halves in four 16-byte reads,
                //  a document's 8 sums read AND zeroed by four ds_wrxchg2_rtn_b32, the 8 counters below in two 16-byte reads: 10 instead of 32)
                // (the thresholds' score halves as SIGNED numbers a sum compares with directly, kept across the block's rounds: they change
                //  only in a cut -- a vector instruction costs a round ~ 16 cycles, and this was 16 of them and four LDS reads a round)
                if (d0 == 0 || thr_stale) {
                    const uint4* t4 = reinterpret_cast<const uint4*>(tau);
#pragma unroll
                    for (int i = 0; i < QT / 2; ++i) {
                        const uint4 t = t4[i];
                        thr[2 * i] = 2 * i < nq ? (int32_t)(t.y ^ 0x80000000u) : 0x7FFFFFFF;          // (a slot without a query: no sum reaches it)
                        thr[2 * i + 1] = 2 * i + 1 < nq ? (int32_t)(t.w ^ 0x80000000u) : 0x7FFFFFFF;
                    }
                    thr_stale = false;
                }
                if (d < rows_b) {
                    const int64_t row = b * a.rows + d;
                    int32_t sums[QT];
#if VS_QUAD_EPI == 0
                    int32_t* pa = acc + quad_acc_index((uint32_t)d);
#pragma unroll
                    for (int q = 0; q < QT; ++q) sums[q] = pa[q * 16];
#pragma unroll
                    for (int q = 0; q < QT; ++q) pa[q * 16] = 0;
#elif VS_QUAD_EPI == 1
                    {
                        const uint32_t pa = quad_acc_index((uint32_t)d) * 4u, zero = 0u;      // LDS byte address (the accumulators start at 0)
                        asm volatile("ds_wrxchg_rtn_b32 %0, %8, %9\n\tds_wrxchg_rtn_b32 %1, %8, %9 offset:64\n\tds_wrxchg_rtn_b32 %2, %8, %9 offset:128\n\t"
                                     "ds_wrxchg_rtn_b32 %3, %8, %9 offset:192\n\tds_wrxchg_rtn_b32 %4, %8, %9 offset:256\n\tds_wrxchg_rtn_b32 %5, %8, %9 offset:320\n\t"
                                     "ds_wrxchg_rtn_b32 %6, %8, %9 offset:384\n\tds_wrxchg_rtn_b32 %7, %8, %9 offset:448\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(sums[0]), "=&v"(sums[1]), "=&v"(sums[2]), "=&v"(sums[3]), "=&v"(sums[4]), "=&v"(sums[5]), "=&v"(sums[6]), "=&v"(sums[7])
                                     : "v"(pa), "v"(zero) : "memory");
                    }
#else
                    {
                        const uint32_t pa = quad_acc_index((uint32_t)d) * 4u, zero = 0u;
                        unsigned long long s01, s23, s45, s67;
                        asm volatile("ds_wrxchg2_rtn_b32 %0, %4, %5, %5 offset0:0 offset1:16\n\tds_wrxchg2_rtn_b32 %1, %4, %5, %5 offset0:32 offset1:48\n\t"
                                     "ds_wrxchg2_rtn_b32 %2, %4, %5, %5 offset0:64 offset1:80\n\tds_wrxchg2_rtn_b32 %3, %4, %5, %5 offset0:96 offset1:112\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(s01), "=&v"(s23), "=&v"(s45), "=&v"(s67) : "v"(pa), "v"(zero) : "memory");
                        sums[0] = (int32_t)(uint32_t)s01; sums[1] = (int32_t)(uint32_t)(s01 >> 32); sums[2] = (int32_t)(uint32_t)s23; sums[3] = (int32_t)(uint32_t)(s23 >> 32);
                        sums[4] = (int32_t)(uint32_t)s45; sums[5] = (int32_t)(uint32_t)(s45 >> 32); sums[6] = (int32_t)(uint32_t)s67; sums[7] = (int32_t)(uint32_t)(s67 >> 32);
                    }
#endif
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        if (sums[q] >= thr[q]) {
                            const uint32_t hi = (uint32_t)sums[q] ^ 0x80000000u;
                            const uint64_t key = ((uint64_t)hi << 32) | (uint32_t)(~(uint32_t)row);
                            if (key > tau[q] && key < upper_sh[q]) {
                                const uint32_t old = atomicAdd(&ccnt[q], inc);                 // (low half: first rounds, high half: last rounds -- below)
                                my_gcand[(size_t)q * kBpCap + (old & 0xFFFFu) + (old >> 16)] = key;
                            }
                        }
                    }
                }
                // Whether a buffer has to be cut is ONE decision of the workgroup (barriers sit behind it), read by every thread from the
                // counters behind the round's barrier -- so what a thread reads there must not depend on WHEN it reads.  A block's first
                // round counts its candidates in the LOW half of a slot's counter, its last round in the HIGH half (a candidate's place is
                // low + high of the value the atomic returns).  Behind the first round's barrier the waves that are already in the second
                // round change the high halves only: the count is low half + the high half as it stood at the end of the previous block
                // (chi[], written there by one thread, two barriers ago).  Behind a block's last round nobody pushes before the next
                // walk's barrier, which no wave passes before every wave has read: low + high as they are.
                // (Until round 6 a counter was one number: a wave already in the second round pushed between two waves' reads, and with a
                //  counter exactly at the limit -- 1024 after the first round of a work item's first block, where every document is a
                //  candidate -- the late reader went into the cut's barriers alone: whole blocks of candidates lost once other
                //  processes' waves on the CU stretched the window.  docs/EXPERIMENTS.md round 6, profiles/r06_prune_decision_race.txt.)
                lds_barrier();                                          // (the counters and sums are LDS; candidates other threads stored are read only when a prune follows)
                // (VS_BP_KNOB = 128 + 4096 n, tests: one wave reads the counters n x 512 cycles late -- the others are pushing the next round's
                //  candidates by then.  A SCALAR branch: s_sleep does not care about exec -- behind a vector condition every wave slept)
                if ((a.knob & 128) && __builtin_amdgcn_readfirstlane(wv) == 5)
                    for (int i = 0; i < (a.knob >> 12); ++i) __builtin_amdgcn_s_sleep(8);
                const bool last = b + 1 >= b1 && !more;
                // (every vector instruction of a thread costs the epilogue ~ 16 cycles a round -- 4 waves a SIMD, 4 cycles each: low + high
                //  of a counter is ONE v_dot2_u32_u16, the eight limits one compare of their maximum)
                uint32_t cmax;
                {
                    const uint4* c4 = reinterpret_cast<const uint4*>(ccnt);
                    const uint4 c0 = c4[0], c1 = c4[1];
                    const uint32_t w[QT] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                    uint32_t t[QT];
                    if (more) {
                        const uint4 h0 = c4[2], h1 = c4[3];              // chi[]
                        const uint32_t h[QT] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                        for (int q = 0; q < QT; ++q) t[q] = __builtin_amdgcn_udot2(__builtin_bit_cast(quad_us2, w[q]), quad_us2{1, 0}, h[q], false);
                    } else {
#pragma unroll
                        for (int q = 0; q < QT; ++q) t[q] = __builtin_amdgcn_udot2(__builtin_bit_cast(quad_us2, w[q]), quad_us2{1, 1}, 0u, false);
                        if (tid < QT) chi[tid] = ccnt[tid] >> 16;                    // (the next block's first round reads it two barriers from here)
                    }
                    cmax = max(max(max(t[0], t[1]), max(t[2], t[3])), max(max(t[4], t[5]), max(t[6], t[7])));
                }
                const bool any = last || cmax > (uint32_t)(kBpCap - kScanThreads);
                [[maybe_unused]] long long t_cut = 0;
                if constexpr (TM != 0) t_cut = (long long)__builtin_readcyclecounter();
                if (any) __syncthreads();                               // the candidates stored above become visible to the workgroup
                thr_stale = any;                                        // (a cut raises thresholds)
                if (any)
                for (int qs = 0; qs < nq; ++qs) {
                    // (a slot's count as the decision above took it: a wave that has left this loop is already pushing the next round's
                    //  candidates -- into the high halves)
                    const uint32_t cw = ccnt[qs], cnt = (cw & 0xFFFFu) + (more ? chi[qs] : (cw >> 16));
                    if (last || cnt > (uint32_t)(kBpCap - kScanThreads)) {
                        for (int i = tid; i < kBpCap; i += kScanThreads) sortbuf[i] = (uint32_t)i < cnt ? my_gcand[(size_t)qs * kBpCap + i] : 0ull;
                        if (last) {
                            // the item's result: its K best, sorted.  K <= 256 (k = 100: 128): cut to K by radix select, then ONE wave sorts
                            // them in registers (no barrier stages); else the workgroup's bitonic sort of the whole buffer
                            uint64_t* out = a.cand + ((size_t)(q0 + qs) * a.nchunk + c) * (size_t)K;
                            if (K <= 256) {
                                wg_final_topk256<kScanThreads>(sortbuf, cnt, K, my_gcand + (size_t)qs * kBpCap, out, cut_hist, tid);
                            } else {
                                wg_sort_desc<kScanThreads>(sortbuf, kBpCap, tid);
                                for (int i = tid; i < K; i += kScanThreads) out[i] = sortbuf[i];
                            }
                        } else if (cnt > (uint32_t)K) {
                            // (a cut needs the K best as a SET and the K-th key, not an order: radix select instead of the 66-stage sort)
                            const unsigned long long kth_sel = wg_cut_topk<kScanThreads>(sortbuf, K, my_gcand + (size_t)qs * kBpCap, cut_hist, tid);
                            if (tid == 0) {
                                const unsigned long long kth = kth_sel;
                                if (kth > tau[qs]) tau[qs] = kth;
                                if (a.gtau && kth != 0ull) atomicMax(a.gtau + q0 + qs, kth);
                                ccnt[qs] = (uint32_t)K;                     // (low half K, high half 0)
                                chi[qs] = 0u;
                            }
                        }
                        __syncthreads();
                    }
                }
                if constexpr (TM != 0) { if (any) tacc[3] += (uint32_t)((long long)__builtin_readcyclecounter() - t_cut); }      // (phase clocks: "dense" = inside the cuts)
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf) : : "memory");   // (the prefetch above has landed -- long ago)
            lap(4);
            if constexpr (TM != 0) tacc[5] += 1u;
            if (b + 1 >= b1) break;
        }
        if constexpr (TM != 0) {
            if ((tid & 63) == 0) {
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
