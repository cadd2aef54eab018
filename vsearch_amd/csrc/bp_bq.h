// bp_bq.h -- "bag-of-token chunks": the column-grouped copy of a BINARY (bag-of-token) index as direct-mapped chunks of uint16 cells, and
// its walk -- the quad walk (bp_quad.h) re-cut for short lists without values (round 5).
//
// What bounds the record walk (bp_bin.h, 17.5 k q/s at 21 M docs): instructions.  A (tile, block) is 6 208 lists of ~6 postings; per
// list the walk reads a directory word, computes a record address, loads 16 bytes, unpacks 8 ids into 8 addresses, predicates the second
// record of 1 list in 6: ~ 1 700 instructions per wave and block, 2.3 of per-list bookkeeping for every posting (DESIGN r4 8.2).
//
// Here the list of (block b, column c) is chunk c of block b: kBqCells cells of uint16 (document in the block); unused cells hold one
// of the kBqSpare spare documents behind a slot's plane (their sums are never read; spread, so that a wave's pad cells do not pile up
// on one LDS address).  No directory is read at search time: a tile's descriptor table {chunk | plane address << 16, integer weight} is
// built once per work item and serves all its blocks.  A wave step serves 64 / kBqGroupLanes lists -- a lane group each, a lane loads
// kBqLaneDwords dwords of the chunk -- with one ds_read_b64, one global_load and 2 x kBqLaneDwords x (v_mad_u32_u16, ds_add_u32)
// (tools/gen_bq_asm.py; the loop is one generated asm statement with counted waits).  A list longer than a chunk keeps kBqLinked
// postings in it, the last cell LINKS to an overflow chunk behind the block's main chunks; a wave collects the links it meets in a
// list of its own and walks that after the table (as the quad walk does).
//
// BLOCK SHAPE.  What the walk waits for is line requests (a CU's L1 gets ~ 0.19 lines a clock from L2 whatever they hold) and LDS adds;
// both are paid per LIST VISIT, and a visit serves as many (document, query) pairs as the block has documents -- the query slots only
// share the block's barriers.  So: as many documents per block as keep a list inside its chunk (18 postings a list on average for 32
// cells: 1 list in ~ 700 overflows), and as many query slots as the LDS then has room for: 2 slots of up to 8192 documents, or 4 of up
// to 4096 (the first form: 16-cell chunks, 2048 documents x 8 slots, 50.9 ms at 21 M docs x 1024 queries; this one: see DESIGN 4).
// Copy size: n_blocks x n_cols x 64 bytes (6.5 GB at 21 M docs, V = 29 523, 6144-document blocks) against 3.6 GB of CSR packets.
//
// Same arithmetic as bp_bin_topk: integer weight (the query's weight x its power-of-two scale: exact for dyadic weights) added
// with ds_add_u32 into slot-major int32 planes -- sums, candidates and results are bit-identical (tests/test_gpu_filter.py).
#pragma once
#include "bp_bin.h"
#include "bp_quad.h"
#include "bp_bq_asm.h"

#ifndef VS_BQ_DYN
#define VS_BQ_DYN 1             // the table's steps dealt to the waves: 1 = in batches from a counter in LDS (bq_dyn_asm), 0 = statically (wave w: steps w, w + 16, ...)
#endif
#ifndef VS_BQ_ZERO
#define VS_BQ_ZERO 1            // packed walk's epilogue: 1 = read AND zero a thread's 16 dwords of sums, 0 = never zero them, keep the last block's in registers (16 VGPRs live across the walk)
#endif

namespace vs {

constexpr int kBqCells = 2 * kBqGroupLanes * kBqLaneDwords, kBqLinked = kBqCells - 1, kBqChunkBytes = 2 * kBqCells;
constexpr int kBqRowsMax = 8192;                                      // documents per block, at most (2 query slots; <= 4096: 4 slots)
constexpr int kBqSpare = 256;                                         // spare documents behind a slot's plane: the pad cells' targets
constexpr int kBqFill = 18;                                           // postings a list should hold on average (of kBqLinked + 1 cells)
constexpr int kBqPaceDefault = 0;                                     // lock-step window in blocks: off (16-cell form, 21 M docs: free running 50.9 ms, window 4: 61.2, 16: 56.8, 32: 52.6)
constexpr int kBqStepDesc = 64 / kBqGroupLanes;                       // descriptors (lists) of a wave step
constexpr int kBqTableStep = kScanWaves * kBqStepDesc;               // descriptors of one step of all 16 waves
__host__ __device__ constexpr int bq_slots(int rows) { return rows > 4096 ? 2 : 4; }                 // query slots of a tile
__host__ __device__ constexpr int bq_rmax(int qt) { return qt == 2 ? 8192 : 4096; }                  // documents a slot's plane holds
__host__ __device__ constexpr uint32_t bq_plane(int qt) { return (uint32_t)(bq_rmax(qt) + kBqSpare) * 4u; }     // bytes of a slot plane
static_assert(3u * bq_plane(4) < 65536u && bq_plane(2) < 65536u, "plane addresses fit the descriptor's high half");
static_assert((size_t)2 * bq_plane(2) >= 65536 && (size_t)4 * bq_plane(4) >= 65536, "the planes double as the 8192-key entry sort buffer");
__host__ __device__ constexpr size_t bq_fixed_lds(int qt) { return (size_t)qt * bq_plane(qt) + (size_t)kFlCap * 8 + 8 * 16 + 32 * 4; }
// (query, column) entries per tile: whole thousands, as many as the LDS holds beside the planes of a four-slot tile, the candidate sort
// buffer and the null steps a walk over-reads (and the 8192-key entry sort: kBpEntCap)
constexpr int bq_ent_cap() {
    const int room = (160 * 1024 - (int)bq_fixed_lds(4)) / 8 - kBqTableStep * (1 + kBqOverRead);
    return room / 1024 * 1024 < kBpEntCap ? room / 1024 * 1024 : kBpEntCap;
}
constexpr int kBqEntCap = bq_ent_cap();
static_assert(kBqEntCap >= 4096, "a tile holds two queries of 2048 tokens");
constexpr int kBqBaseWin = 64;                                        // block bases staged in LDS at a time (behind the descriptor table)
__host__ __device__ constexpr size_t bq_base_lds(int qt) { return bq_fixed_lds(qt) + (size_t)(kBqEntCap + kBqTableStep + kBqTableStep * kBqOverRead) * 8; }
__host__ __device__ constexpr size_t bq_lds_bytes(int qt) { return bq_base_lds(qt) + (size_t)kBqBaseWin * 8; }
static_assert(bq_lds_bytes(2) <= (size_t)160 * 1024 && bq_lds_bytes(4) <= (size_t)160 * 1024, "the chunk walk's LDS");
// a wave's two link lists live in its share of the candidate sort buffer (32 KB / 16 waves = 2 KB): 128 descriptors each, of which
// kBqStepDesc * kBqOverRead are the null ones a walk over-reads
constexpr int kBqListSlots = kFlCap / kScanWaves / 2, kBqListCap = kBqListSlots - kBqStepDesc * kBqOverRead, kBqListBytes = kBqListSlots * 8;
static_assert(kBqListCap >= 16 && kBqListCap % kBqStepDesc == 0, "the link lists fit the sort buffer");
// pad cell of (column c, cell j): one of the spare documents behind the plane (pad0 = the plane's document capacity), spread over the
// cells of a chunk and over neighbouring columns
__host__ __device__ constexpr uint16_t bq_pad(uint32_t pad0, uint32_t c, uint32_t j) { return (uint16_t)(pad0 + (((c & 15u) << 4) | ((j >> 1) & 15u))); }
static_assert(kBqCells <= 64 && kBqRowsMax + kBqSpare < 32768, "a cell is a document below 2^15 or a link (bit 15)");

// ---- builder: quad_count_kernel<kBqCells, kBqLinked> (postings per column -> overflow chunks, directory for the fill) -> bp_base_kernel -> fill --
// fill: one workgroup per block: pad cells first (every main chunk, then the overflow chunks), then the block's non-zeros in arrival
// order, then the links
template <int UNUSED>
__global__ __launch_bounds__(kScanThreads) void bq_fill_kernel(const uint32_t* pk_ptr, const uint4* cols, int64_t n_rows, int32_t n_cols, int32_t rows, uint32_t pad0,
                                                               const uint32_t* dir, const unsigned long long* base, uint16_t* rec) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* cur = reinterpret_cast<uint32_t*>(smem);                  // [n_cols + 1] postings placed so far
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t n_blocks = (n_rows + rows - 1) / rows;
    constexpr uint32_t kParts = kBqCells / 8;                           // 16-byte parts of a chunk
    for (int64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const int64_t r0 = b * rows, r1 = min(n_rows, r0 + rows);
        const uint32_t* d = dir + (size_t)b * (n_cols + 1);
        __syncthreads();
        for (int i = tid; i <= n_cols; i += kScanThreads) cur[i] = 0u;
        uint16_t* brec = rec + (size_t)base[b] * kBqCells;
        const uint32_t n_over = d[n_cols] >> 12;                        // (quad_count_kernel: the block's overflow chunks)
        // pads: 8 cells (16 bytes) per thread and turn
        for (uint32_t i = tid; i < ((uint32_t)n_cols + n_over) * kParts; i += kScanThreads) {
            const uint32_t chunk = i / kParts, j0 = (i % kParts) * 8u;
            uint32_t wds[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) wds[k] = (uint32_t)bq_pad(pad0, chunk, j0 + 2 * k) | ((uint32_t)bq_pad(pad0, chunk, j0 + 2 * k + 1) << 16);
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
                        // chunk k of the list (0: the main chunk = chunk c, k >= 1: overflow chunk first + k - 1) holds kBqLinked postings
                        // when another follows it, up to kBqCells when it is the last
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
                brec[chunk * kBqCells + (kBqCells - 1)] = (uint16_t)(0x8000u | ((wd >> 12) + k));
            }
        }
    }
}

// ---- packed sums: which tiles qualify ---------------------------------------------------------------------------------
// longest row of the index in non-zeros (upper bound: its packets x 8) -- what a document can match of a query at most
template <int UNUSED>
__global__ __launch_bounds__(256) void bq_maxrow_kernel(const uint32_t* pk_ptr, int64_t n_rows, uint32_t* out) {
    uint32_t m = 0u;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * 256) m = max(m, pk_ptr[r + 1] - pk_ptr[r]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m * 8u);
}

// The plan's tiles (up to four queries each) -> the tiles of the packed walk (every query's sums fit 16 bits: flag16, bp_qscale_kernel)
// and two-slot tiles for the int32 walk (the others, cut in two).  One workgroup; order kept.
template <int UNUSED>
__global__ __launch_bounds__(kScanThreads) void bq_split_kernel(const int2* tiles, const int32_t* n_tiles_dev, const uint32_t* flag16, int2* tiles16, int2* tiles2, int32_t* n_out) {
    __shared__ int scratch[32];
    const int tid = threadIdx.x, nt = n_tiles_dev[0];
    const int per = (nt + kScanThreads - 1) / kScanThreads;
    const int t0 = min(nt, tid * per), t1 = min(nt, t0 + per);
    auto packed = [&](int2 t) { bool ok = true; for (int i = 0; i < t.y; ++i) ok = ok && flag16[t.x + i] != 0u; return ok; };
    int m16 = 0, m2 = 0;
    for (int t = t0; t < t1; ++t) { const int2 tl = tiles[t]; if (packed(tl)) ++m16; else m2 += tl.y > 2 ? 2 : 1; }
    int tot16 = 0, tot2 = 0;
    int o16 = block_excl_scan(m16, scratch, tid, &tot16);
    __syncthreads();
    int o2 = block_excl_scan(m2, scratch, tid, &tot2);
    for (int t = t0; t < t1; ++t) {
        const int2 tl = tiles[t];
        if (packed(tl)) tiles16[o16++] = tl;
        else {
            tiles2[o2++] = make_int2(tl.x, min(tl.y, 2));
            if (tl.y > 2) tiles2[o2++] = make_int2(tl.x + 2, tl.y - 2);
        }
    }
    if (tid == 0) { n_out[0] = tot16; n_out[1] = tot2; }
}

// ---- walk -----------------------------------------------------------------------------------------------------------
// Work items, tiles, thresholds and candidates as bp_bin_topk.  BpArgs::rec = the chunks, base[b] = first chunk of block b.
// Epilogue: thread t finishes documents 4 t .. 4 t + 3 (and 4096 + 4 t ..: two-plane tiles) of every slot: 16 dwords in registers from
// 4 ds_read_b128.  A block may hold more candidates than the candidate buffer has room for (a slot keeps
// kFlCap keys per workgroup, a block has up to 8192 documents): a push that finds no room stays PENDING in its thread, the buffer is
// sorted and cut to the best K' (which raises the threshold), and the pending sums are tested again -- a few turns in a work item's
// first block, none later.
//
// PACKED SUMS (PK = 1, round 6): the walk of a (tile, block) is traffic and per-block work -- half of its chunk requests miss the L2 with
// two query slots a tile (512 tiles each sweep their own chunks: 174 GB of HBM traffic a search), and 38 % of the wave-cycles are the
// barriers and the epilogue, paid per (tile, block).  Both halve when a tile holds FOUR slots on the same two planes: slots 2 p and
// 2 p + 1 share the dwords of plane p -- a descriptor's weight is pre-shifted by 16 (s & 1) bits, the add stays ONE ds_add_u32 (the
// generated loop is unchanged) and the low half cannot carry into the high one while every document's sum stays below 2^16.  That is a
// property of (index, query): a document matches at most min(longest row, query entries) of the query's columns, so
// min(longest row, n) x the largest integer weight < 65 536 is sufficient -- bp_qscale_kernel checks it per query (and picks the
// smallest power-of-two scale that makes the weights integers instead of the 2^30 one), bq_split_kernel sends the tiles whose queries
// all qualify here and cuts the others into two-slot tiles for the int32 kernel.  The epilogue takes a block's sums as the difference
// of the running dwords as before (mod 2^32 the carries of earlier blocks cancel) and splits the difference into its halves.
template <int QT, int TM, int PK = 0>          // QT = query slots (2: blocks of up to 8192 documents, 4: up to 4096; PK: 4 on two planes of 8192); TM = 1: phase clocks (VS_BP_TIMING)
__global__ __launch_bounds__(kScanThreads) void bp_bq_topk(BpArgs a) {
    constexpr int NP = PK ? QT / 2 : QT;                                                    // planes
    constexpr int RMAX = bq_rmax(NP), NH = RMAX / (4 * kScanThreads);
    constexpr uint32_t PLANE = bq_plane(NP);
    static_assert(NP * NH * 4 == 16 && (NP == 2 || NP == 4) && (!PK || QT == 4), "a thread finishes 16 dwords of sums");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* acc = reinterpret_cast<int32_t*>(smem);                                        // [NP][RMAX + spare], LDS address 0 (no static LDS in this kernel)
    uint64_t* sortbuf = reinterpret_cast<uint64_t*>(smem + (size_t)NP * PLANE);             // [kFlCap]; during a walk: the waves' link lists
    unsigned long long* tau = reinterpret_cast<unsigned long long*>(sortbuf + kFlCap);      // [8]
    unsigned long long* upper_sh = tau + 8;                                                 // [8]
    int* scratch = reinterpret_cast<int*>(upper_sh + 8);                                    // [16]
    unsigned int* ccnt = reinterpret_cast<unsigned int*>(scratch + 16);                     // [16]
    uint2* desc = reinterpret_cast<uint2*>(ccnt + 16);                                      // [n_static + kBqTableStep * kBqOverRead]
    unsigned long long* bases = reinterpret_cast<unsigned long long*>(smem + bq_base_lds(NP));      // [kBqBaseWin] first chunks of the item's next blocks
    const uint32_t desc_lds = (uint32_t)bq_fixed_lds(NP);
    const uint32_t sort_lds = (uint32_t)NP * PLANE;
    const uint32_t deal_lds = sort_lds + (uint32_t)kFlCap * 8u + 128u;                     // scratch[0]: the block's batch counter (bq_dyn_asm)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int K = a.k;
    uint64_t* my_gcand = a.gcand + (size_t)blockIdx.x * QT * kFlCap;
    const int64_t n_blocks = (a.n_rows + a.rows - 1) / a.rows;
    const int64_t items = (int64_t)(a.n_tiles_dev ? a.n_tiles_dev[0] : a.n_tiles) * a.nchunk;
    const unsigned long long k_rt0 = TM ? __builtin_amdgcn_s_memrealtime() : 0ull;
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    const uint32_t list_a = sort_lds + (uint32_t)wv * 2u * kBqListBytes, list_b = list_a + kBqListBytes;
    const uint32_t g8 = (uint32_t)(lane / kBqGroupLanes) * 8u, l4 = (uint32_t)(lane % kBqGroupLanes) * (uint32_t)(kBqLaneDwords * 4);
    [[maybe_unused]] constexpr uint32_t kWaveStep = kBqStepDesc * 8u;                    // bytes of a wave's descriptors of one step

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
                    const uint32_t wi = (uint32_t)(int32_t)__uint_as_float((uint32_t)key);                                   // integer weight (bp_bin.h)
                    if constexpr (PK != 0) dsc = make_uint2(col | (((qs >> 1) * PLANE) << 16), (wi & 0xFFFFu) << (16u * (qs & 1u)));      // two slots share a plane's dwords
                    else dsc = make_uint2(col | ((qs * PLANE) << 16), wi);
                }
                desc[i] = dsc;
            }
            __syncthreads();
        }
        for (int i = tid; i < (int)(NP * PLANE / 4); i += kScanThreads) acc[i] = 0;
        // the sums are never zeroed again: a thread finishes the SAME 16 cells in every block and keeps what they held after the last
        // one -- a block's sum is the difference (mod 2^32: exact), and the epilogue's LDS traffic is reads only
        uint32_t prev[NP][NH][4];
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) prev[q][h][j] = 0u;
        if (tid < 8) { tau[tid] = 0ull; ccnt[tid] = 0u; upper_sh[tid] = (a.upper && tid < nq) ? a.upper[q0 + tid] : ~0ull; }
        if (tid == 0) scratch[0] = 2 * kScanWaves;                       // the batch counter: batches w and w + 16 are the waves' to start with
        __syncthreads();
        const uint32_t trips = (uint32_t)(n_static / kBqTableStep);
        bool pace_off = false;
        // (a block's first chunk comes from a window of 64 staged in LDS: loaded per block behind the walk -- where it had to go after the
        //  round-5 finding below -- its latency sat in front of every block's barrier)
        lap(0);
        for (int64_t b = b0; b < b1 || b == b0; ++b) {
            const bool have = b < b1;
            const int rows_b = have ? (int)min((int64_t)a.rows, a.n_rows - b * a.rows) : 0;
            if (((b - b0) & (kBqBaseWin - 1)) == 0) {
                if (tid < kBqBaseWin) bases[tid] = b + tid < b1 ? a.base[b + tid] : 0ull;
                __syncthreads();
            }
            const unsigned long long base_cur = bases[(b - b0) & (kBqBaseWin - 1)];
            if (have && trips > 0) {
                const char* brec = a.rec + (size_t)base_cur * kBqChunkBytes;
                // the overflow chunks a walk found: the wave's list `cur` holds n of them; their own links go to the other list
                auto chain = [&](uint32_t n, uint32_t cur, uint32_t nxt) {
                    // (a list has at most RMAX / kBqLinked + 1 chunks: the bound keeps a corrupt link from hanging the GPU)
                    for (int depth = 0; n > 0 && depth < RMAX / kBqLinked + 2; ++depth) {
                        const uint32_t n_pad = (n + (uint32_t)kBqStepDesc - 1u) / (uint32_t)kBqStepDesc * (uint32_t)kBqStepDesc;
                        for (uint32_t i = (uint32_t)lane; i < n_pad - n + (uint32_t)(kBqStepDesc * kBqOverRead); i += 64u)
                            reinterpret_cast<uint2*>(smem + cur)[n + i] = make_uint2(0u, 0u);       // null descriptors behind the list (smem = LDS address 0)
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        n = bq_list_asm(cur + g8, n_pad / (uint32_t)kBqStepDesc, brec, l4, (uint32_t)a.n_cols, nxt, (uint32_t)kBqListCap);
                        const uint32_t t = cur; cur = nxt; nxt = t;
                    }
                };
#if VS_BQ_DYN
                // batches of kBqSets steps from a counter in LDS: the waves finish the block's table together whatever their memory luck
                // (the statement comes back early when the wave's link list wants walking: tools/gen_bq_asm.py build_dyn)
                {
                    const uint32_t nb = (uint32_t)n_static / (uint32_t)(kBqSets * kBqStepDesc);
                    uint32_t cur = (uint32_t)wv, nxt = (uint32_t)wv + (uint32_t)kScanWaves, pend = ~0u;
                    while (cur < nb) {
                        const uint32_t n_link = bq_dyn_asm(desc_lds + g8, nb, cur, nxt, pend, deal_lds, brec, l4, (uint32_t)a.n_cols, list_a, (uint32_t)kBqListCap);
                        chain(n_link, list_a, list_b);
                    }
                }
#else
                const uint32_t n_link = bq_walk_asm(desc_lds + (uint32_t)wv * kWaveStep + g8, trips, brec, l4, (uint32_t)a.n_cols, list_a, (uint32_t)kBqListCap);
                if (n_link <= (uint32_t)kBqListCap) {
                    chain(n_link, list_a, list_b);
                } else {
                    // more links than the list holds (a tile of very long lists): the table's chunks are added, their links are collected
                    // again, as many steps at a time as the list has room for
                    constexpr uint32_t kSeg = kBqListCap / kBqStepDesc;
                    for (uint32_t t0 = 0; t0 < trips; t0 += kSeg) {
                        const uint32_t n = bq_collect_asm(desc_lds + (uint32_t)wv * kWaveStep + g8 + t0 * (uint32_t)(kBqTableStep * 8), min(kSeg, trips - t0), brec, l4,
                                                          (uint32_t)a.n_cols, list_a, (uint32_t)kBqListCap);
                        chain(n, list_a, list_b);
                    }
                }
#endif
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
            // (round 5: the next base loaded ahead of the walk and held in a VGPR across it came back wrong in some waves when four processes
            //  shared the GPU: tests/test_gpu_search.py, docs/EXPERIMENTS.md -- hence the window in LDS)
            if (a.gtau && tid < nq) { const unsigned long long g = a.gtau[q0 + tid]; if (g > tau[tid]) tau[tid] = g; }
            lds_barrier();                                               // the block's sums are complete
            lap(2);
            if (tid == 0) scratch[0] = 2 * kScanWaves;                   // (the next block's batch counter: every wave has left the walk, the next one starts behind a barrier)
            {
                uint32_t thi[QT];
                {
                    const uint4* t4 = reinterpret_cast<const uint4*>(tau);
#pragma unroll
                    for (int i = 0; i < QT / 2; ++i) { const uint4 t = t4[i]; thi[2 * i] = t.y; thi[2 * i + 1] = t.w; }
                }
                // fast test first: the epilogue is bound by VALU issue (4 waves a SIMD), and behind a work item's first blocks hardly a
                // sum passes its threshold -- the largest of a thread's 4 sums of a (slot, half) against it (signed: the raw sum against
                // the biased threshold), one wave-uniform branch; only a wave with a hit works out which sums they are
                // (packed sums: both halves of a dword against their slots' thresholds in ONE saturating v_pk_sub_u16 -- a half stays
                //  non-zero exactly when its sum reaches the threshold)
                uint32_t dif[NP][NH][4];                                 // a block's sums: int32 (PK = 0) or two 16-bit halves (PK = 1)
                uint32_t pend = 0u;                                      // bit (q * NH + h) * 4 + j: a candidate not yet in the buffer
                bool hit = false;
                if constexpr (PK != 0) {
                    uint32_t tm1[NP];
                    bool fast_v = true;
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const int32_t tl = 2 * p < nq ? (int32_t)(thi[2 * p] ^ 0x80000000u) : 0x7FFFFFFF;
                        const int32_t th = 2 * p + 1 < nq ? (int32_t)(thi[2 * p + 1] ^ 0x80000000u) : 0x7FFFFFFF;
                        fast_v = fast_v && tl > 0 && th > 0;             // (a threshold of <= 0 lets every document pass: no short cut)
                        tm1[p] = (uint32_t)min(max(tl - 1, 0), 65535) | ((uint32_t)min(max(th - 1, 0), 65535) << 16);
                    }
                    const bool fast = __builtin_amdgcn_readfirstlane((int)fast_v) != 0;       // (the thresholds are the workgroup's: say so to the compiler)
                    // (fast: a dword's saturated difference against its two thresholds SAYS which halves pass -- the wave looks at a dword
                    //  only when one of its lanes has a passing half (the compare's mask is the ballot: no vector instruction more than the
                    //  OR it replaces), so a wave with one candidate in a block does one dword's worth of work, not 32 compares)
                    uint32_t any = 0u;
#pragma unroll
                    for (int p = 0; p < NP; ++p)
#pragma unroll
                        for (int h = 0; h < NH; ++h) {
                            uint4* pa = reinterpret_cast<uint4*>(acc + p * (RMAX + kBqSpare) + h * 4096 + 4 * tid);
                            const uint4 v = *pa;
                            if constexpr (VS_BQ_ZERO != 0) *pa = make_uint4(0u, 0u, 0u, 0u);
                            const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if constexpr (VS_BQ_ZERO != 0) dif[p][h][j] = vv[j];
                                else { dif[p][h][j] = vv[j] - prev[p][h][j]; prev[p][h][j] = vv[j]; }
                                uint32_t r;
                                asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(dif[p][h][j]), "v"(tm1[p]));
                                if (fast) {
                                    if (__builtin_amdgcn_ballot_w64(r != 0u) != 0ull) {
                                        if (r & 0xFFFFu) pend |= 1u << (((2 * p) * NH + h) * 4 + j);
                                        if (r >> 16) pend |= 1u << (((2 * p + 1) * NH + h) * 4 + j);
                                    }
                                } else any |= 1u;
                            }
                        }
                    hit = any != 0u;
                } else {
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        const int32_t thr = q < nq ? (int32_t)(thi[q] ^ 0x80000000u) : 0x7FFFFFFF;
#pragma unroll
                        for (int h = 0; h < NH; ++h) {
                            const uint4 v = *reinterpret_cast<const uint4*>(acc + q * (RMAX + kBqSpare) + h * 4096 + 4 * tid);
                            dif[q][h][0] = v.x - prev[q][h][0]; dif[q][h][1] = v.y - prev[q][h][1];
                            dif[q][h][2] = v.z - prev[q][h][2]; dif[q][h][3] = v.w - prev[q][h][3];
                            prev[q][h][0] = v.x; prev[q][h][1] = v.y; prev[q][h][2] = v.z; prev[q][h][3] = v.w;
                            hit = hit || max(max((int32_t)dif[q][h][0], (int32_t)dif[q][h][1]), max((int32_t)dif[q][h][2], (int32_t)dif[q][h][3])) >= thr;
                        }
                    }
                }
                // the sum of (slot q, half h, document j of the thread's four)
                auto sum_of = [&](int q, int h, int j) -> uint32_t {
                    if constexpr (PK != 0) { const uint32_t d = dif[q >> 1][h][j]; return (q & 1) ? d >> 16 : d & 0xFFFFu; }
                    else return dif[q][h][j];
                };
                if (__builtin_amdgcn_ballot_w64(hit) != 0ull) {
#pragma unroll
                    for (int q = 0; q < QT; ++q)
#pragma unroll
                        for (int h = 0; h < NH; ++h)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (q < nq && h * 4096 + 4 * tid + j < rows_b && (sum_of(q, h, j) ^ 0x80000000u) >= thi[q]) pend |= 1u << ((q * NH + h) * 4 + j);
                }
                lap(3);                                                  // (phase clocks: "dense" = reading the sums and testing them)
                const bool last = b + 1 >= b1;
                for (;;) {
                    if (pend != 0u) {
#pragma unroll
                        for (int q = 0; q < QT; ++q)
#pragma unroll
                            for (int h = 0; h < NH; ++h)
                                if (__builtin_amdgcn_ballot_w64((pend & (0xFu << ((q * NH + h) * 4))) != 0u) != 0ull)        // (wave-uniform: most groups of four hold nothing)
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const uint32_t bit = 1u << ((q * NH + h) * 4 + j);
                                    if (pend & bit) {
                                        const int64_t row = (int64_t)b * a.rows + h * 4096 + 4 * tid + j;
                                        const uint64_t key = ((uint64_t)(sum_of(q, h, j) ^ 0x80000000u) << 32) | (uint32_t)(~(uint32_t)row);
                                        bool keep = false;
                                        if (key > tau[q] && key < upper_sh[q]) {
                                            const uint32_t pos = atomicAdd(&ccnt[q], 1u);
                                            if (pos < (uint32_t)kFlCap) my_gcand[(size_t)q * kFlCap + pos] = key;
                                            else keep = true;               // no room: again after the buffer's cut
                                        }
                                        if (!keep) pend &= ~bit;
                                    }
                                }
                    }
                    lds_barrier();
                    uint32_t cnts[QT];
                    {
                        const uint4 c0 = *reinterpret_cast<const uint4*>(ccnt);
                        cnts[0] = c0.x; cnts[1] = c0.y;
                        if constexpr (QT == 4) { cnts[2] = c0.z; cnts[3] = c0.w; }
                    }
                    bool over = false, any = last;
#pragma unroll
                    for (int q = 0; q < QT; ++q) { over = over || cnts[q] > (uint32_t)kFlCap; any = any || cnts[q] > (uint32_t)(kFlCap / 2); }
                    if (!any) break;
                    __syncthreads();                                 // (the candidates pushed above are read back: a full barrier)
                    for (int qs = 0; qs < nq; ++qs) {
                        const uint32_t cn = min(cnts[qs], (uint32_t)kFlCap);
                        if (last || cn > (uint32_t)(kFlCap / 2)) {
                            for (int i = tid; i < kFlCap; i += kScanThreads) sortbuf[i] = (uint32_t)i < cn ? my_gcand[(size_t)qs * kFlCap + i] : 0ull;
                            wg_sort_desc<kScanThreads>(sortbuf, kFlCap, tid);
                            if (last && !over) {
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
                            } else if (tid == 0) {
                                ccnt[qs] = cn;
                            }
                            __syncthreads();
                        }
                    }
                    if (!over) break;
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
