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
// predicated (bp_quad_loop.h has the layout, tools/gen_quad_asm.py the loop).  Per (block, tile) the workgroup first PLANS: every
// thread turns its 7 (query, column) entries and the block's directory words (fetched a block ahead) into chunk descriptors
// {chunk offset | slot, weight} in LDS -- a prefix sum, no list is longer than one descriptor's 64 cells -- then the waves
// take steps w, w + 16, ...: equal work by construction (the list walk needed a dynamic queue).  The builder deals the postings
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

namespace vs {

constexpr int kQuadRows = 2048;                                   // documents per block
constexpr int kQuadQT = 8;                                        // query slots per tile
constexpr int kQuadPer = kBpEntCap / kScanThreads;                // entries a thread owns: 7
constexpr size_t kQuadAccBytes = (size_t)(kQuadRows / 16) * kQuadGroupDw * 4;      // 73 728: accumulators, at LDS address 0
static_assert(kBpEntCap % kScanThreads == 0, "a thread owns a fixed number of entries");
static_assert(kQuadAccBytes >= kBpSortBytes, "the accumulator area holds the 8192-slot entry sort");
// LDS: accumulators | candidate sort buffer | thresholds, bounds, scratch, counters | overflow bitmap of a block | descriptors (+ the
// null steps the loop over-reads)
__host__ __device__ constexpr size_t quad_fixed_lds() { return kQuadAccBytes + (size_t)kBpCap * 8 + (size_t)kQuadQT * 16 + 64 * 4; }
__host__ __device__ inline int quad_bitmap_words(int n_cols) { return ((n_cols + 31) / 32 + 15) & ~15; }
__host__ __device__ inline int quad_desc_cap(int n_cols) {
    return (int)((((size_t)160 * 1024 - quad_fixed_lds() - (size_t)quad_bitmap_words(n_cols) * 4) / 8 - 64 * kQuadOverRead) / 64 * 64);
}
__host__ __device__ inline size_t quad_lds_bytes(int n_cols) {
    return quad_fixed_lds() + (size_t)quad_bitmap_words(n_cols) * 4 + (size_t)(quad_desc_cap(n_cols) + 64 * kQuadOverRead) * 8;
}

// ---- builder --------------------------------------------------------------------------------------------------------
// A block's chunks: one MAIN chunk per column at chunk index = column (the first 64 postings of the column's list: 19 lists in 20 end
// there on 768-nnz documents), then the block's OVERFLOW chunks.  The main chunks need no directory -- a tile's descriptors for them
// are the same in every block -- and a bitmap says which columns of a block have overflow; only for those a directory word
// dir[b][c] = first overflow chunk (from the block's first overflow chunk) << 12 | overflow chunks is looked up.
// pass 1, one workgroup per block: postings per column -> overflow chunks, directory, bitmap, block total (V + overflow chunks)
template <int UNUSED>
__global__ __launch_bounds__(kScanThreads) void quad_count_kernel(const uint32_t* pk_ptr, const uint4* cols, int64_t n_rows, int32_t n_cols, int32_t rows,
                                                                  uint32_t* dir, uint32_t* ovf_bits, int32_t bm_words, uint32_t* block_recs,
                                                                  unsigned long long* df_rec, unsigned long long* df_nnz, int32_t* overflow) {
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
        auto ovf_of = [&](int i) -> uint32_t { return cnt[i] > (uint32_t)kQuadCells ? (cnt[i] - 1u) / (uint32_t)kQuadCells : 0u; };
        const int i0 = min(n_cols, tid * seg), i1 = min(n_cols, i0 + seg);
        int mine = 0;
        for (int i = i0; i < i1; ++i) mine += (int)ovf_of(i);
        int tot = 0;
        int off = block_excl_scan(mine, scratch, tid, &tot);
        uint32_t* d = dir + (size_t)b * (n_cols + 1);
        for (int i = i0; i < i1; ++i) {
            const uint32_t c = cnt[i], r = ovf_of(i);
            d[i] = bp_dir_pack((uint32_t)off, r);
            if (r > kBpDirRecMask || (uint32_t)off > kBpDirUnitMax) overflow[0] = 1;
            off += (int)r;
            if (c) {
                atomicAdd(&df_rec[i], (unsigned long long)(1u + r));
                atomicAdd(&df_nnz[i], (unsigned long long)c);
            }
        }
        uint32_t* bm = ovf_bits + (size_t)b * bm_words;
        for (int w = tid; w < bm_words; w += kScanThreads) {
            uint32_t bits = 0;
            for (int k = 0; k < 32; ++k) {
                const int c = 32 * w + k;
                if (c < n_cols && cnt[c] > (uint32_t)kQuadCells) bits |= 1u << k;
            }
            bm[w] = bits;
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
                        // the main chunk of column c is chunk c; posting 64 + o sits in overflow chunk first + o / 64 behind the main chunks
                        const size_t cell = pos < (uint32_t)kQuadCells ? (size_t)c * kQuadCells + pos
                                                                       : ((size_t)n_cols + (d[c] >> 12)) * kQuadCells + (pos - (uint32_t)kQuadCells);
                        brec[cell] = ai | (hb[i] << 16);
                    }
                }
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
            uint32_t mask[4] = {0u, 0u, 0u, 0u}, fill[4] = {0u, 0u, 0u, 0u};
            for (int i = 0; i < kQuadCells; ++i) {
                const uint32_t p = mi[i];
                if ((p & 0x7FFF0000u) == 0u) continue;                  // unused cell, or an explicit zero: adds nothing either way
                const uint32_t bit = 1u << (p & 31u);
                int pick = -1, any = -1;
                uint32_t best = 17u, best_any = 17u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (fill[j] < 16u && fill[j] < best_any) { best_any = fill[j]; any = j; }
                    if (fill[j] < 16u && !(mask[j] & bit) && fill[j] < best) { best = fill[j]; pick = j; }
                }
                if (pick < 0) pick = any;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j == pick) { mo[fill[j] * 4 + j] = p; mask[j] |= bit; ++fill[j]; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                while (fill[j] < 16u) {
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
// chunk of block b, dir / ovf_bits as the builder wrote them; BpArgs::gent = [grid][kBpEntCap] scratch for the item's sorted entries.
//
// Per item: the tile's entries are sorted by column and turned into the STATIC part of the descriptor table (main chunks: the same
// in every block).  Per block: the overflow descriptors of this block are appended behind it (`plan`: a prefix sum over the ~ 7 % of
// the entries whose bitmap bit is set), the waves walk the table (quad_walk_asm), then -- behind one LDS barrier -- the next block's
// bitmap goes to LDS, every thread tests its 7 entries and gathers the directory words of the flagged ones (they land under the
// epilogue), and the epilogue turns the block's sums into candidates.
template <int TM>          // TM = 1: phase clocks (VS_BP_TIMING)
__global__ __launch_bounds__(kScanThreads) void bp_quad_topk(BpArgs a) {
    constexpr int QT = kQuadQT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* acc = reinterpret_cast<int32_t*>(smem);                                        // LDS address 0 (no static LDS in this kernel)
    uint64_t* sortbuf = reinterpret_cast<uint64_t*>(smem + kQuadAccBytes);                  // [kBpCap]
    unsigned long long* tau = reinterpret_cast<unsigned long long*>(sortbuf + kBpCap);      // [QT]
    unsigned long long* upper_sh = tau + QT;                                                // [QT]
    int* scratch = reinterpret_cast<int*>(upper_sh + QT);                                   // [48]
    unsigned int* ccnt = reinterpret_cast<unsigned int*>(scratch + 48);                     // [QT]
    uint32_t* bitmap = reinterpret_cast<uint32_t*>(scratch + 64);                           // [bm_words] overflow bits of a block
    const int bm_words = quad_bitmap_words(a.n_cols);
    uint2* desc = reinterpret_cast<uint2*>(bitmap + bm_words);                              // [desc_cap + 64 * kQuadOverRead]
    const uint32_t desc_lds = (uint32_t)(quad_fixed_lds() + (size_t)bm_words * 4);          // its LDS byte address
    const int desc_cap = quad_desc_cap(a.n_cols);

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int K = a.k;
    uint64_t* my_gcand = a.gcand + (size_t)blockIdx.x * QT * kBpCap;
    const int64_t n_blocks = (a.n_rows + a.rows - 1) / a.rows;
    const int64_t items = (int64_t)(a.n_tiles_dev ? a.n_tiles_dev[0] : a.n_tiles) * a.nchunk;
    const size_t dir_ld = (size_t)a.n_cols + 1;
    const unsigned long long k_rt0 = TM ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // a barrier that orders LDS only: global loads stay in flight across it (a __syncthreads() waits for them: a DRAM round trip)
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // exclusive prefix sum over the wave's lanes on the VALU (DPP row shifts and broadcasts: no LDS traffic); *tot = the wave's sum
    auto wave_excl_scan = [&](int v, int* tot) {
        int x = v;
        x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);       // row_shr:1
        x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);       // row_shr:2
        x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);       // row_shr:4
        x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);       // row_shr:8: inclusive sums inside every row of 16
        x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);      // row_bcast:15 -> rows 1, 3
        x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);      // row_bcast:31 -> rows 2, 3
        *tot = __builtin_amdgcn_readlane(x, 63);
        return x - v;
    };

    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int tile = (int)(item / a.nchunk), c = (int)(item % a.nchunk);
        const int q0 = a.tiles[tile].x, nq = a.tiles[tile].y;
        [[maybe_unused]] long long tm = TM ? (long long)__builtin_readcyclecounter() : 0;
        [[maybe_unused]] uint32_t tacc[12] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};      // 6 .. 11: sub-phases of the plan
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
        // the tile's entries sorted by column (the accumulator area doubles as the sort buffer) -> the static descriptors (main chunk
        // of column c = chunk c of every block) and, for the overflow lookups, the global scratch
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
        if (tid < QT) { tau[tid] = 0ull; ccnt[tid] = 0u; }
        if (tid < QT) upper_sh[tid] = (a.upper && tid < nq) ? a.upper[q0 + tid] : ~0ull;
        // Entry ownership for the overflow lookups, the same for every block of the item: wave w owns the `per` x 64 consecutive
        // entries from w * per * 64, lane l the entries  w * per * 64 + 64 i + l  (i < per).  What a thread needs of an entry -- column,
        // slot, weight -- it re-reads from the entry's static descriptor in LDS: nothing is held in registers across the walk.
        const int per = (n_ent + kScanThreads - 1) / kScanThreads;
        const int ebase = wv * per * 64 + lane;
        // directory words of the thread's entries that overflow in block bb (its bitmap is in LDS), 0 for the others
        // (two blocks ahead: under load a directory gather takes ~ 15 k cycles to come back -- more than an epilogue hides)
        uint32_t nd[kQuadPer], nd2[kQuadPer];                            // of the next block to walk, of the one after
        auto fetch_dir = [&](int64_t bb, uint32_t (&nd)[kQuadPer]) {
            const uint32_t* dirn = a.dir + (size_t)bb * dir_ld;
#pragma unroll
            for (int i = 0; i < kQuadPer; ++i) {
                const int e = ebase + 64 * i;
                const uint32_t col = desc[min(e, n_static)].x >> 8;         // (beyond the tile: a null descriptor, never used)
                const bool flagged = i < per && e < n_ent && ((bitmap[col >> 5] >> (col & 31u)) & 1u);
                nd[i] = flagged ? dirn[col] : 0u;
            }
        };
        uint32_t bm_next = 0;                                            // the thread's word of a coming block's bitmap, fetched a block before it is needed
#pragma unroll
        for (int i = 0; i < kQuadPer; ++i) { nd[i] = 0u; nd2[i] = 0u; }
        if (b0 < b1) {
            for (int w = tid; w < bm_words; w += kScanThreads) bitmap[w] = a.ovf_bits[(size_t)b0 * bm_words + w];
            __syncthreads();
            fetch_dir(b0, nd);
            if (b0 + 1 < b1) {
                __syncthreads();
                for (int w = tid; w < bm_words; w += kScanThreads) bitmap[w] = a.ovf_bits[(size_t)(b0 + 1) * bm_words + w];
                __syncthreads();
                fetch_dir(b0 + 1, nd2);
            }
            if (b0 + 2 < b1 && tid < bm_words) bm_next = a.ovf_bits[(size_t)(b0 + 2) * bm_words + tid];
        }
        __syncthreads();
        // (the first chunk of a block is fetched a block ahead: a scalar load's DRAM round trip would otherwise sit in front of the first
        //  LDS barrier of the plan -- s_waitcnt lgkmcnt(0) waits for scalar loads too)
        unsigned long long base_cur = b0 < b1 ? a.base[b0] : 0ull;
        // One block: `ndb` holds the directory words gathered for it two blocks ago and, once the plan has read them, takes the gathers
        // of block b + 2.  The loop below alternates between two such arrays: neither is touched while its gathers are in flight.
        auto one_block = [&](const int64_t b, uint32_t (&ndb)[kQuadPer]) {
            const bool have = b < b1;
            const int rows_b = have ? (int)min((int64_t)a.rows, a.n_rows - b * a.rows) : 0;
            if (have) {
                const char* brec = a.rec + (size_t)base_cur * kQuadChunkBytes;
                // plan: this block's overflow descriptors behind the static part.  A thread's descriptors are numbered by a prefix sum;
                // round r of the walk takes numbers [r C, (r + 1) C) -- normally there is one round
                uint32_t cd[kQuadPer];
                int mine = 0;
#pragma unroll
                for (int i = 0; i < kQuadPer; ++i) { cd[i] = ndb[i]; mine += (int)(cd[i] & kBpDirRecMask); }      // (gathered two blocks ago)
                lap(6);
                int wtot = 0;
                int first_no = wave_excl_scan(mine, &wtot);
                if (lane == 0) scratch[wv] = wtot;
                lap(7);
                lds_barrier();
                lap(8);
                int n_ovf = 0;
#pragma unroll
                for (int u = 0; u < kScanWaves; ++u) { const int t = scratch[u]; first_no += u < wv ? t : 0; n_ovf += t; }
                const int C = (desc_cap - n_static) & ~63;
                // descriptors number [lo, hi) of the block's overflow -> the table behind the static part, null steps behind them
                auto emit = [&](int lo, int hi) {
                    int no = first_no;
#pragma unroll
                    for (int i = 0; i < kQuadPer; ++i) {
                        const int nch = (int)(cd[i] & kBpDirRecMask);
                        if (nch > 0 && no < hi && no + nch > lo) {
                            const uint2 sd = desc[ebase + 64 * i];         // the entry's static descriptor: slot row and weight
                            const uint32_t first = (uint32_t)a.n_cols + (cd[i] >> 12), so = sd.x & 0xFFu;
                            for (int j = max(0, lo - no); j < nch && no + j < hi; ++j)
                                desc[n_static + (no + j - lo)] = make_uint2(((first + (uint32_t)j) << 8) | so, sd.y);
                        }
                        no += nch;
                    }
                    const int n_end = n_static + (hi - lo), n_pad = (n_end + 63) & ~63;
                    for (int i = n_end + tid; i < n_pad + 64 * kQuadOverRead; i += kScanThreads) desc[i] = make_uint2(0u, 0u);
                    return n_pad;
                };
                if (n_ovf <= C) {
                    // the normal case: one walk over the static part and all of the block's overflow
                    const int n_pad = emit(0, n_ovf);
                    lap(9);
                    lds_barrier();
                    lap(3);                                              // (phase 3 = the plan, phase 1 = the walk proper)
                    if (n_pad > 0) quad_walk_asm(desc_lds + (uint32_t)(wv * 4 + (lane >> 4)) * 8u, (uint32_t)(n_pad / 64), brec, (uint32_t)(lane & 15) * 16u);
                    lap(1);
                } else {
                    // more overflow than the table holds (a tile of very long lists): the static part, then the overflow C descriptors at a time
                    for (int lo = 0; lo < n_ovf; lo += C) {
                        if (lo > 0) lds_barrier();                       // (the walk of the round before has read the table)
                        const int n_pad = emit(lo, min(n_ovf, lo + C));
                        lds_barrier();
                        const int from = lo == 0 ? 0 : n_static;
                        quad_walk_asm(desc_lds + (uint32_t)(from + wv * 4 + (lane >> 4)) * 8u, (uint32_t)((n_pad - from) / 64), brec, (uint32_t)(lane & 15) * 16u);
                    }
                    lap(1);
                }
                // the next block's bitmap -> LDS (its words were fetched a block ago)
                if (b + 2 < b1 && tid < bm_words) bitmap[tid] = bm_next;
            }
            if (a.gtau && tid < nq) { const unsigned long long g = a.gtau[q0 + tid]; if (g > tau[tid]) tau[tid] = g; }
            lds_barrier();                                               // the block's sums are complete; the bitmap is in place
            lap(2);
            if (b + 1 < b1) base_cur = a.base[b + 1];
            if (have && b + 2 < b1) {
                fetch_dir(b + 2, ndb);                                   // gathers of the flagged entries: a whole block to land
                if (b + 3 < b1 && tid < bm_words) bm_next = a.ovf_bits[(size_t)(b + 3) * bm_words + tid];
            }
            lap(0);                                                      // (phase 0 = issuing the directory gathers)
            // epilogue: 1024 documents at a time, one per thread: its QT sums -> order keys -> candidates; prune when a buffer could overflow
            for (int d0 = 0; d0 < rows_b || d0 == 0; d0 += kScanThreads) {
                const int d = d0 + tid;
                uint32_t thi[QT];
#pragma unroll
                for (int q = 0; q < QT; ++q) thi[q] = (uint32_t)(tau[q] >> 32);
                if (d < rows_b) {
                    const int64_t row = b * a.rows + d;
                    int32_t* pa = acc + quad_acc_index((uint32_t)d);
                    int32_t sums[QT];
#pragma unroll
                    for (int q = 0; q < QT; ++q) sums[q] = pa[q * 16];
#pragma unroll
                    for (int q = 0; q < QT; ++q) pa[q * 16] = 0;
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        const uint32_t hi = (uint32_t)sums[q] ^ 0x80000000u;
                        if (q < nq && hi >= thi[q]) {
                            const uint64_t key = ((uint64_t)hi << 32) | (uint32_t)(~(uint32_t)row);
                            if (key > tau[q] && key < upper_sh[q]) {
                                const uint32_t pos = atomicAdd(&ccnt[q], 1u);
                                my_gcand[(size_t)q * kBpCap + pos] = key;
                            }
                        }
                    }
                }
                lds_barrier();                                          // (the counters and sums are LDS; candidates other threads stored are read only when a prune follows)
                const bool last = b + 1 >= b1 && d0 + kScanThreads >= rows_b;
                uint32_t cnts[QT];
#pragma unroll
                for (int q = 0; q < QT; ++q) cnts[q] = ccnt[q];
                bool any = last;
#pragma unroll
                for (int q = 0; q < QT; ++q) any = any || cnts[q] > (uint32_t)(kBpCap - kScanThreads);
                if (any) __syncthreads();                               // the candidates stored above become visible to the workgroup
                if (any)
                for (int qs = 0; qs < nq; ++qs) {
                    const uint32_t cnt = ccnt[qs];
                    if (last || cnt > (uint32_t)(kBpCap - kScanThreads)) {
                        for (int i = tid; i < kBpCap; i += kScanThreads) sortbuf[i] = (uint32_t)i < cnt ? my_gcand[(size_t)qs * kBpCap + i] : 0ull;
                        wg_sort_desc<kScanThreads>(sortbuf, kBpCap, tid);
                        if (last) {
                            uint64_t* out = a.cand + ((size_t)(q0 + qs) * a.nchunk + c) * (size_t)K;
                            for (int i = tid; i < K; i += kScanThreads) out[i] = sortbuf[i];
                        } else if (cnt > (uint32_t)K) {
                            for (int i = tid; i < K; i += kScanThreads) my_gcand[(size_t)qs * kBpCap + i] = sortbuf[i];
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
        };
        for (int64_t b = b0;; b += 2) {
            one_block(b, nd);
            if (b + 1 >= b1) break;
            one_block(b + 1, nd2);
            if (b + 2 >= b1) break;
        }
        if constexpr (TM != 0) {
            if ((tid & 63) == 0) {
#pragma unroll
                for (int i = 0; i < 12; ++i) atomicAdd(a.timing + i, (unsigned long long)tacc[i]);
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
