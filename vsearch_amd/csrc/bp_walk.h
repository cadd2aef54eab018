// bp_walk.h -- "blocked postings": a second, column-grouped copy of a CSR index for SPARSE queries, and the walk over it.
//
// The CSR scan (csr_scan_mq.h) looks every index non-zero up in the tile table although only ~2.6 % x Qt of them carry a
// query weight.  Here the rows are cut into blocks of R documents (R <= 2048, picked at build time); inside a block the
// non-zeros are grouped by column into posting lists, and a query tile walks ONLY the lists of its own columns: every visited
// non-zero is a hit.  The products go to accumulators [document][query slot] in LDS; after each block a thread finishes its
// documents (sums -> order keys -> candidate buffers), like the per-row epilogue of the CSR pass.
//
// Layout ("records", all 16-byte aligned): a list is a run of RECORDS of 8 postings -- 8 uint16 document-in-block ids (16 B)
// followed by their 8 values (fp32: 32 B, fp16: 16 B, binary index: none) -- the last record zero-padded (document 0, value 0:
// adds nothing).  dir[b][c] = ONE word per list: (first record of column c in block b, in units of `align` records, relative
// to the block) << 12 | records of the list; base[b] = first record of block b.  Lists are packed (align = 1) or, option
// "postings_align", start on whole 128-byte lines (fp16 values: 4 x 32 B): the 7 records of an average list are then exactly
// two lines instead of 2.5 -- measured: 1.5 % less walk time for 16 % more bytes with 8 lanes per list, and the only layout
// on which 4 lanes per list keep up.  A lane takes whole records (one 16-byte load of ids + one/two of values, no per-posting bounds), a group
// of LG lanes takes consecutive records of one list, so a list of n postings is ~6 n contiguous bytes: with 128-byte cache
// lines that is what decides how many of the fetched bytes are used (rocprofv3, 21 M docs: the round-1 layout -- ids and values
// in separate arrays, 832-document blocks, 22-posting lists -- moved 6.7 TB through L2->L1 for 2.8 TB of postings and held the
// walk at the ~64 outstanding lines a CU can keep in flight).
//
// Accumulator modes:
//   AM_F64: fp32 product, fp64 sum (ds_add_f64) -- the library's exact numerics, same as the CSR pass.
//   AM_FIX: fp32 product of the PRE-SCALED weight (w * 2^e, e per query so that no sum can reach 2^31), truncated to int32 and
//           summed with ds_add_u32 -- 3.4x the ds_add_f64 rate on MI355X (tools/microbench/lds_scatter.hip: 5.0 vs 1.5 Tadd/s) and
//           half the LDS, hence twice the documents per block.  Integer sums are order-independent, and every truncation loses
//           less than one unit, so  |S * exact - A| < n  (n = matched terms) bounds the exact fp64 sum by the approximate one.
//           The walk then FILTERS: it returns the K' > k best documents by approximate score; refine_topk_kernel (bp_refine.h)
//           re-scores exactly those with the exact numerics and proves, per query, that no other document can reach the top k;
//           queries it cannot prove take the exact one-query CSR scan (exact_scan_topk_kernel).
//
// How a block is walked (bp_walk_topk): the tile's (column, slot, weight) entries, sorted by column, are cut into CHUNKS of one
// wave-slot each (8 lane groups x NB lists); a wave takes its first chunk by number and the following ones from a counter in LDS.
// A lane group loads one record per lane for each of its NB lists back to back (asm: global_load + per-list s_waitcnt), plus
// the second record a lane owes to the first of its lists that is longer than one round, then multiplies (v_fma_mix_f32: fp16
// value x fp32 weight), truncates and adds (ds_add_u32).  Directory words are fetched one chunk ahead.  With head columns
// (dense fp16 strips in MFMA operand order, bp_strip_index) the same queue also deals DENSE chunks: 64 documents x all head
// columns on v_mfma_f32_16x16x32_f16 against the tile's weights split in two fp16 numbers.  After the block's barrier a thread
// turns its two documents' sums into order keys and candidates; thresholds are shared between all work items of a query (gtau).
// What binds and what was tried: DESIGN.md 4 / 8.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <type_traits>

#include "csr_scan.h"
#include "dense_csr.h"

namespace vs {

enum : int { AM_F64 = 0, AM_FIX = 1 };

constexpr int kBpRowsMax = 2048;      // most documents per block of a valued index (8 query slots x int32 sums fill the LDS)
constexpr int kBpRowsMaxBin = 2048;   // binary (bag-of-token) index: same blocks (measured: 4096-document blocks with 4 query slots -- longer lists,
                                      // half the list visits -- lose to the doubled number of tiles: 11.8 k vs 14.1 k q/s on the Wiki21M shape)
constexpr int kBpCap = 2048;          // candidate slots per (workgroup, query slot): K' kept + 1024 new per epilogue round
constexpr int kBpMaxK = kBpCap - kScanThreads;
constexpr int kBpEntCap = 7168;       // (query, column) entries per tile: 56 KB of LDS, and the 8192-slot entry sort must hold them
constexpr int kBpNB = 4;              // posting lists whose loads are in flight together per lane group
constexpr int kBpNBWide = 6;          // ... for 32-byte records on 8-lane groups (the filter walk of a valued index): 6 x 8 + 8 registers of records
constexpr size_t kBpSortBytes = (size_t)8192 * 8;   // the entry sort of a tile (8192 slots) borrows the accumulator area

__host__ __device__ constexpr int bp_rec_bytes(int vm) { return vm == VM_F32 ? 48 : (vm == VM_F16 ? 32 : 16); }
// lists start on a multiple of 2^shift records: 4 x 32 B = one 128-byte line, 8 x 48 B = three; the one-record lists of a binary index stay packed
__host__ __device__ constexpr int bp_align_shift(int vm) { return vm == VM_F32 ? 3 : (vm == VM_F16 ? 2 : 0); }
// directory word of a list: first record (in alignment units, relative to the block) << 12 | records
constexpr uint32_t kBpDirRecMask = 0xFFFu, kBpDirUnitMax = 0xFFFFFu;
__host__ __device__ inline uint32_t bp_dir_pack(uint32_t unit, uint32_t recs) { return (unit << 12) | (recs & kBpDirRecMask); }

// ---- builder ------------------------------------------------------------------------------------------------------
// pass 1: one workgroup per block: postings per column -> records per column -> directory (record offsets) + block total;
//         also the per-column totals over all blocks (records and non-zeros: what a query entry on that column walks)
template <int UNUSED>
__global__ __launch_bounds__(kScanThreads) void bp_count_kernel(const uint32_t* pk_ptr, const uint4* cols, int64_t n_rows, int32_t n_cols, int32_t rows,
                                                                uint32_t* dir, uint32_t* block_recs, unsigned long long* df_rec,
                                                                unsigned long long* df_nnz, const uint16_t* hmap, int32_t al_shift, int32_t* overflow) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem);                  // [n_cols + 1]
    __shared__ int scratch[32];
    const int tid = threadIdx.x;
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
                atomicAdd(&cnt[cwv[i] & 0xFFFFu], 1u);                  // row padding lands in cnt[n_cols], dropped below
                atomicAdd(&cnt[cwv[i] >> 16], 1u);
            }
        }
        __syncthreads();
        if (tid == 0) cnt[n_cols] = 0;
        __syncthreads();
        const int i0 = tid * seg, i1 = min(n_cols + 1, i0 + seg);
        // (head columns -- hmap[c] != 0xFFFF -- live in the dense strips: no records, empty lists)
        auto recs_of = [&](int i) -> uint32_t { return (hmap && i < n_cols && hmap[i] != 0xFFFFu) ? 0u : (cnt[i] + 7u) >> 3; };       // 8 postings a record
        // a list starts on a multiple of 2^al_shift records: the scan runs in those units
        const uint32_t al_mask = (1u << al_shift) - 1u;
        int mine = 0;
        for (int i = i0; i < i1; ++i) mine += (int)((recs_of(i) + al_mask) >> al_shift);
        int tot = 0;
        int off = block_excl_scan(mine, scratch, tid, &tot);
        uint32_t* d = dir + (size_t)b * (n_cols + 1);
        for (int i = i0; i < i1; ++i) {
            const uint32_t c = cnt[i], r = recs_of(i);
            d[i] = bp_dir_pack((uint32_t)off, r);
            if (r > kBpDirRecMask || (uint32_t)off > kBpDirUnitMax) overflow[0] = 1;
            off += (int)((r + al_mask) >> al_shift);
            if (i < n_cols && c) {
                atomicAdd(&df_rec[i], (unsigned long long)r);
                atomicAdd(&df_nnz[i], (unsigned long long)c);
            }
        }
        if (tid == 0) block_recs[b] = (uint32_t)tot << al_shift;
    }
}

// Head columns: a column present in a large share of the documents (skewed vocabularies: the ~500 most popular columns of a
// Zipf(1) corpus cover 3/4 of all non-zeros) is cheaper as a DENSE strip -- one fp16 value per document of the block, and the
// block's [documents x head columns] strip times the tile's [head columns x slots] weights is a small GEMM for the matrix
// cores -- than as a 2048-posting list of scatter-adds (an LDS atomic costs ~20 lane-cycles).  Measured per block and tile:
// ~95 clocks per head column (the strip streams from L2 at ~26 TB/s chip-wide) against 4 800 p^2 for a list of density p, so
// the lists win below p ~ 1/7; the default threshold is 1/4.  hmap[c] = strip index of column c, 0xFFFF = ordinary column.
constexpr int kBpHeadCap = 512;          // multiplied inside the walk (their weights live in its LDS)
constexpr int kHeadOutShift = 14;        // the head pre-pass hands its sums over as uint16 in units of 2^14 (a sum < 2^30 fits; < 2^14 units lost: bp_head_slack)
constexpr int kBpHeadCapGemm = 1536;     // served by the head pre-pass (bp_head.h), where the HBM has room for their strips;
constexpr int kBpHeadCapGemmLow = 1024;  // ... where it is tight
// One workgroup.  Deterministic: when more than `cap` columns reach `thresh`, the threshold rises to the smallest document
// count that leaves at most `cap` of them; strip indexes follow column order.
template <int UNUSED>
__global__ __launch_bounds__(kScanThreads) void bp_head_select_kernel(const unsigned long long* df_nnz, int32_t n_cols, unsigned long long thresh, int32_t cap,
                                                                      uint16_t* hmap, int32_t* n_head) {
    __shared__ int cnt;
    __shared__ int part[kScanThreads];
    const int tid = threadIdx.x;
    const int per = (n_cols + kScanThreads - 1) / kScanThreads;
    const int c0 = min(n_cols, tid * per), c1 = min(n_cols, c0 + per);
    auto count_ge = [&](unsigned long long t) {
        __syncthreads();
        if (tid == 0) cnt = 0;
        __syncthreads();
        int n = 0;
        for (int c = c0; c < c1; ++c) n += df_nnz[c] >= t ? 1 : 0;
        if (n) atomicAdd(&cnt, n);
        __syncthreads();
        return cnt;
    };
    if (count_ge(thresh) > cap) {
        // smallest t in (thresh, 2^63] with count_ge(t) <= cap
        unsigned long long lo = thresh, hi = 1ull << 62;            // count_ge(lo) > cap, count_ge(hi) = 0
        while (hi - lo > 1) {
            const unsigned long long mid = lo + (hi - lo) / 2;
            if (count_ge(mid) > cap) lo = mid; else hi = mid;
        }
        thresh = hi;
    }
    int n = 0;
    for (int c = c0; c < c1; ++c) n += df_nnz[c] >= thresh ? 1 : 0;
    __syncthreads();
    part[tid] = n;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < kScanThreads; ++i) { const int v = part[i]; part[i] = run; run += v; }
        n_head[0] = run;
    }
    __syncthreads();
    int run = part[tid];
    for (int c = c0; c < c1; ++c) hmap[c] = df_nnz[c] >= thresh ? (uint16_t)run++ : (uint16_t)0xFFFFu;
}

// pass 2: exclusive scan of the block totals -> base[b] (records), base[n_blocks] = all records.  One workgroup.
template <int UNUSED>
__global__ __launch_bounds__(kScanThreads) void bp_base_kernel(const uint32_t* block_recs, int64_t n_blocks, unsigned long long* base) {
    __shared__ unsigned long long part[kScanThreads];
    const int tid = threadIdx.x;
    const int64_t per = (n_blocks + kScanThreads - 1) / kScanThreads;
    const int64_t i0 = min(n_blocks, (int64_t)tid * per), i1 = min(n_blocks, i0 + per);
    unsigned long long s = 0;
    for (int64_t i = i0; i < i1; ++i) s += block_recs[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < kScanThreads; ++i) { const unsigned long long v = part[i]; part[i] = run; run += v; }
        base[n_blocks] = run;
    }
    __syncthreads();
    unsigned long long run = part[tid];
    for (int64_t i = i0; i < i1; ++i) { base[i] = run; run += block_recs[i]; }
}

// The dense strips, stored in the order v_mfma_f32_16x16x32_f16 wants its A operand (16 documents x 32 head columns per
// instruction; lane l holds document l & 15, columns 8 * (l >> 4) .. + 7): 16-byte units
//   [block][k-step = column / 32][document / 16][lane = (column % 32) / 8 * 16 + document % 16], 8 halves (column % 8) each,
// so a wave reads one operand with ONE coalesced 1 KB load and the 8 operands of its 128 documents from 8 KB contiguous.
// n_head rounds up to 32 columns (the pad columns are zero).
__host__ __device__ inline int bp_head_pad(int n_head) { return (n_head + 31) & ~31; }
// head pre-pass output (bp_head.h): [tile of the pass][block][document / 16][slot 8][16 documents] int32 -> offset of (document d, slot 0)
__host__ __device__ inline size_t head_out_offset(int64_t tile_rel, int64_t n_blocks, int64_t b, int rows, int d) {
    return (((size_t)tile_rel * (size_t)n_blocks + (size_t)b) * (size_t)(rows / 16) + (size_t)(d >> 4)) * 128 + (size_t)(d & 15);
}
__host__ __device__ inline size_t bp_strip_index(int64_t b, int h, int dl, int n_head, int rows) {
    const size_t ks = (size_t)bp_head_pad(n_head) / 32, mb = (size_t)rows / 16;
    const size_t unit = (((size_t)b * ks + (size_t)(h >> 5)) * mb + (size_t)(dl >> 4)) * 64 + (size_t)(((h & 31) >> 3) * 16 + (dl & 15));
    return unit * 8 + (size_t)(h & 7);
}

// pass 3: scatter the non-zeros into their records (the array is zero-filled first: pad postings are document 0, value 0)
// VS = value mode of the CSR packets, VM = value mode of the records (VS = fp32, VM = fp16: the lossy filter copy, see bp_refine.h)
template <int VS, int VM>
__global__ __launch_bounds__(kScanThreads) void bp_fill_kernel(const uint32_t* pk_ptr, const uint4* cols, const void* vals, int64_t n_rows,
                                                               int32_t n_cols, int32_t rows, const uint32_t* dir, const unsigned long long* base,
                                                               char* rec, const uint16_t* hmap, __half* strip, int32_t n_head, int32_t al_shift) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* cur = reinterpret_cast<uint32_t*>(smem);                  // [n_cols + 1] write cursors in postings
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int RS = bp_rec_bytes(VM);
    const int64_t n_blocks = (n_rows + rows - 1) / rows;
    for (int64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const int64_t r0 = b * rows, r1 = min(n_rows, r0 + rows);
        const uint32_t* d = dir + (size_t)b * (n_cols + 1);
        __syncthreads();
        for (int i = tid; i <= n_cols; i += kScanThreads) cur[i] = ((d[i] >> 12) << al_shift) * 8u;
        __syncthreads();
        char* brec = rec + (size_t)base[b] * RS;
        for (int64_t r = r0 + w; r < r1; r += kScanWaves) {
            const uint32_t p0 = pk_ptr[r], p1 = pk_ptr[r + 1];
            const uint16_t dl = (uint16_t)(r - r0);
            for (uint32_t p = p0 + lane; p < p1; p += 64) {
                const uint4 cw = cols[p];
                const uint32_t cwv[4] = {cw.x, cw.y, cw.z, cw.w};
                float v[8];
                if constexpr (VS == VM_F32) {
                    const float4* vp = reinterpret_cast<const float4*>(vals);
                    const float4 v0 = vp[2 * (size_t)p], v1 = vp[2 * (size_t)p + 1];
                    v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
                } else if constexpr (VS == VM_F16) {
                    const uint4 hv = reinterpret_cast<const uint4*>(vals)[p];
                    const __half2* h = reinterpret_cast<const __half2*>(&hv);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float2 f = __half22float2(h[i]); v[2 * i] = f.x; v[2 * i + 1] = f.y; }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t c = (i & 1) ? (cwv[i >> 1] >> 16) : (cwv[i >> 1] & 0xFFFFu);
                    if (c < (uint32_t)n_cols && hmap && hmap[c] != 0xFFFFu) {
                        // head column: the strip in MFMA A-operand order (bp_strip_index)
                        if constexpr (VM != VM_BIN) strip[bp_strip_index(b, hmap[c], dl, n_head, rows)] = __float2half_rn(v[i]);
                    } else if (c < (uint32_t)n_cols) {
                        const uint32_t pos = atomicAdd(&cur[c], 1u);
                        char* rp = brec + (size_t)(pos >> 3) * RS;
                        reinterpret_cast<uint16_t*>(rp)[pos & 7u] = dl;
                        if constexpr (VM == VM_F32) reinterpret_cast<float*>(rp + 16)[pos & 7u] = v[i];
                        if constexpr (VM == VM_F16) reinterpret_cast<__half*>(rp + 16)[pos & 7u] = __float2half_rn(v[i]);
                    }
                }
            }
        }
        if constexpr (VM == VM_BIN) {
            // a binary posting has no value to zero: the pad postings of a list's last record point at the scratch row behind
            // the accumulators (document id = the block capacity), whatever they add is never read
            __syncthreads();
            for (int i = tid; i < n_cols; i += kScanThreads) {
                const uint32_t lim = (((d[i] >> 12) << al_shift) + (d[i] & kBpDirRecMask)) * 8u;
                for (uint32_t pos = cur[i]; pos < lim; ++pos) reinterpret_cast<uint16_t*>(brec + (size_t)(pos >> 3) * RS)[pos & 7u] = (uint16_t)kBpRowsMaxBin;
            }
        } else {
            // pad postings of a valued list carry value 0 (the array was zero-filled): they add nothing WHEREVER they point, so
            // they point at scattered documents of the block -- all on document 0, the pads of a wave's ds_add pile up on one
            // LDS bank (slot-major accumulators, bp_flat.h: every slot's document 0 is bank 0)
            __syncthreads();
            const uint32_t nb = (uint32_t)(r1 - r0);
            for (int i = tid; i < n_cols; i += kScanThreads) {
                const uint32_t lim = (((d[i] >> 12) << al_shift) + (d[i] & kBpDirRecMask)) * 8u;
                for (uint32_t pos = cur[i]; pos < lim; ++pos)
                    reinterpret_cast<uint16_t*>(brec + (size_t)(pos >> 3) * RS)[pos & 7u] = (uint16_t)(((uint32_t)i * 2654435761u + pos * 40503u) % nb);
            }
        }
    }
}

// pass 4 (valued records): bank-aware order INSIDE each list.  The walks give consecutive records of a list to consecutive lanes and
// add posting t of every record in one ds_add_u32; a document's LDS bank is its id mod 32 (slot-major sums; 9 * id + slot mod 32 in
// the padded layout of bp_walk_topk: the same classes).  Within a round of 8 records the 8 postings of position t are re-dealt so
// that their documents fall into DIFFERENT banks (greedy: a posting goes to the first position whose column has a free record
// and not yet its bank; 8 of 32 banks per column: it practically always finds one); pad postings (value 0) get documents of
// unused banks.  The lanes of one list then never collide -- 23 % of the lane pairs of a 32-lane group: Monte Carlo of the
// group's worst bank 3.27 -> 2.81 cycles.  Sums do not depend on the order of a list's postings.  One thread per list.
template <int VM>
__global__ __launch_bounds__(256) void bp_arrange_kernel(const uint32_t* dir, const unsigned long long* base, char* rec, int64_t n_blocks, int32_t n_cols,
                                                         int32_t al_shift) {
    static_assert(VM == VM_F16 || VM == VM_F32, "valued records");
    constexpr int RS = bp_rec_bytes(VM);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t* oid = reinterpret_cast<uint16_t*>(smem) + (size_t)threadIdx.x * 64;                               // [256][64] the re-dealt round
    uint32_t* oval = reinterpret_cast<uint32_t*>(smem + 256 * 64 * 2) + (size_t)threadIdx.x * 64;               // [256][64] (fp16 bits or fp32 bits)
    for (int64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const uint32_t* d = dir + (size_t)b * ((size_t)n_cols + 1);
        char* brec = rec + (size_t)base[b] * RS;
        for (int c = threadIdx.x; c < n_cols; c += 256) {
            const uint32_t w = d[c];
            const uint32_t first = (w >> 12) << al_shift, nrec = w & kBpDirRecMask;
            for (uint32_t r0 = 0; r0 < nrec; r0 += 8) {
                const int R = (int)min(8u, nrec - r0);
                char* rp = brec + (size_t)(first + r0) * RS;
                // greedy deal: column t = position in the record; its cells fill from record 0 up
                uint32_t mask[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};      // banks taken in column t
                uint32_t fill[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};      // cells taken in column t
                int start = 0;
                for (int j = 0; j < R; ++j) {
                    const uint4 iw = *reinterpret_cast<const uint4*>(rp + (size_t)j * RS);
                    const uint32_t iv[4] = {iw.x, iw.y, iw.z, iw.w};
                    uint32_t vv[8];
                    if constexpr (VM == VM_F16) {
                        const uint4 vw = *reinterpret_cast<const uint4*>(rp + (size_t)j * RS + 16);
                        const uint32_t v4[4] = {vw.x, vw.y, vw.z, vw.w};
#pragma unroll
                        for (int t = 0; t < 8; ++t) vv[t] = (t & 1) ? (v4[t >> 1] >> 16) : (v4[t >> 1] & 0xFFFFu);
                    } else {
                        const uint4 v0 = *reinterpret_cast<const uint4*>(rp + (size_t)j * RS + 16), v1 = *reinterpret_cast<const uint4*>(rp + (size_t)j * RS + 32);
                        vv[0] = v0.x; vv[1] = v0.y; vv[2] = v0.z; vv[3] = v0.w; vv[4] = v1.x; vv[5] = v1.y; vv[6] = v1.z; vv[7] = v1.w;
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const uint32_t v = vv[t];
                        if ((v & (VM == VM_F16 ? 0x7FFFu : 0x7FFFFFFFu)) == 0u) continue;      // a pad (or an explicit zero: adds nothing either)
                        const uint32_t id = (t & 1) ? (iv[t >> 1] >> 16) : (iv[t >> 1] & 0xFFFFu), bit = 1u << (id & 31u);
                        int pick = -1, any = -1;
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const int tt = (start + k) & 7;
                            const bool room = fill[tt] < (uint32_t)R;
                            if (room && any < 0) any = tt;
                            if (room && pick < 0 && !(mask[tt] & bit)) pick = tt;
                        }
                        if (pick < 0) pick = any;                                              // (every free column holds the bank already: a conflict stays)
                        oid[fill[pick] * 8 + pick] = (uint16_t)id;
                        oval[fill[pick] * 8 + pick] = v;
                        mask[pick] |= bit;
                        ++fill[pick];
                        start = (pick + 1) & 7;
                    }
                }
                // pads: value 0, a document of a bank the column does not hold yet (documents 0 .. 31 exist in every block)
                for (int t = 0; t < 8; ++t) {
                    while (fill[t] < (uint32_t)R) {
                        const uint32_t freeb = ~mask[t];
                        const uint32_t bk = freeb ? (uint32_t)(__ffs((int)freeb) - 1) : 0u;
                        oid[fill[t] * 8 + t] = (uint16_t)bk;
                        oval[fill[t] * 8 + t] = 0u;
                        mask[t] |= 1u << bk;
                        ++fill[t];
                    }
                }
                for (int j = 0; j < R; ++j) {
                    uint4 iw;
                    iw.x = oid[j * 8 + 0] | ((uint32_t)oid[j * 8 + 1] << 16); iw.y = oid[j * 8 + 2] | ((uint32_t)oid[j * 8 + 3] << 16);
                    iw.z = oid[j * 8 + 4] | ((uint32_t)oid[j * 8 + 5] << 16); iw.w = oid[j * 8 + 6] | ((uint32_t)oid[j * 8 + 7] << 16);
                    *reinterpret_cast<uint4*>(rp + (size_t)j * RS) = iw;
                    if constexpr (VM == VM_F16) {
                        uint4 vw;
                        vw.x = oval[j * 8 + 0] | (oval[j * 8 + 1] << 16); vw.y = oval[j * 8 + 2] | (oval[j * 8 + 3] << 16);
                        vw.z = oval[j * 8 + 4] | (oval[j * 8 + 5] << 16); vw.w = oval[j * 8 + 6] | (oval[j * 8 + 7] << 16);
                        *reinterpret_cast<uint4*>(rp + (size_t)j * RS + 16) = vw;
                    } else {
                        *reinterpret_cast<uint4*>(rp + (size_t)j * RS + 16) = make_uint4(oval[j * 8 + 0], oval[j * 8 + 1], oval[j * 8 + 2], oval[j * 8 + 3]);
                        *reinterpret_cast<uint4*>(rp + (size_t)j * RS + 32) = make_uint4(oval[j * 8 + 4], oval[j * 8 + 5], oval[j * 8 + 6], oval[j * 8 + 7]);
                    }
                }
            }
        }
    }
}

// out[0] = records, out[1] = non-zeros this batch's walk visits: sum over columns of (queries of the batch using the column) x df
template <int UNUSED>
__global__ __launch_bounds__(kScanThreads) void bp_walk_kernel(const uint32_t* colfreq, const unsigned long long* df_rec, const unsigned long long* df_nnz,
                                                               int32_t n_cols, int64_t* out) {
    __shared__ unsigned long long red[2][kScanThreads / 64];
    unsigned long long v = 0, z = 0;
    for (int c = threadIdx.x; c < n_cols; c += kScanThreads) { v += (unsigned long long)colfreq[c] * df_rec[c]; z += (unsigned long long)colfreq[c] * df_nnz[c]; }
    for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o, 64); z += __shfl_xor(z, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = v; red[1][threadIdx.x >> 6] = z; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0, u = 0;
        for (int i = 0; i < kScanThreads / 64; ++i) { t += red[0][i]; u += red[1][i]; }
        out[0] = (int64_t)t;
        out[1] = (int64_t)u;
    }
}

// ---- walk ---------------------------------------------------------------------------------------------------------
// Lock-step wait of the walks (BpArgs::pace): polls until *p >= need.  BOUNDED: lock step is a performance hint that assumes
// every work item of the launch is resident; when another kernel shares the GPU -- several shards of a shard group on one device, two
// processes, masked CUs -- the peers may never be scheduled, and an unbounded wait is a GPU hang.  After kPaceSpins polls (~ 4 ms) the
// caller gives up lock step for the rest of the launch (returns false).
constexpr int kPaceSpins = 1 << 14;
__device__ __forceinline__ bool pace_wait(const uint32_t* p, uint32_t need) {
    for (int spins = 0; spins < kPaceSpins; ++spins) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
        __builtin_amdgcn_s_sleep(8);
    }
    return false;
}

typedef unsigned short walk_us2 __attribute__((ext_vector_type(2)));

struct BpArgs {
    int32_t rows;             // documents per block (<= the kernel's RMAX)
    const uint32_t* dir;      // [n_blocks, n_cols + 1] one word per list (bp_dir_pack)
    int32_t al_shift;         // lists start on a multiple of 2^al_shift records
    const unsigned long long* base;   // [n_blocks + 1] first record of a block
    const char* rec;          // records
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
    const int32_t* n_tiles_dev; // optional: the tile count lives on the device (fallback plan built by a kernel); overrides n_tiles
    int32_t ent_cap;          // LDS capacity for tile entries
    uint64_t* cand;           // [B, nchunk, k] output keys, sorted descending
    uint64_t* gcand;          // [grid, QT, kBpCap] scratch
    const uint64_t* upper;    // optional [B] exclusive upper bounds ("search after")
    const unsigned long long* df;   // optional [n_cols]: non-zeros of every column over the whole index (list-length classes of the entry sort)
    unsigned long long* gtau; // optional [B], zeroed per search: the best K-th key any (tile, chunk) item has established for the query --
                              // a lower bound of the K-th best over all chunks, shared so that no item starts (or stays) cold
    const float* qscale;      // AM_FIX: [B] per-query power-of-two scale of the fixed-point sums
    const uint16_t* hmap;     // head columns (dense strips), n_head > 0 only: [n_cols] strip index or 0xFFFF
    const __half* strip;      // fp16 values of the head columns, MFMA operand order (bp_strip_index)
    int32_t n_head;
    float head_pre, head_mul; // powers of two: weights enter the fp16 operand as w * scale * head_pre (< 2^15), the sums leave as C * head_mul
    const uint16_t* head_out; // HD = 2 (head pre-pass, bp_head.h): the dense part of the sums, [tile - tile0][block][document / 16][slot][16] uint16 in units of 2^kHeadOutShift
    int32_t tile0, tile_cnt;  // tile_cnt > 0: this launch walks tiles [tile0, tile0 + tile_cnt) only (the passes of the head pre-pass)
    uint32_t* pace;           // optional [nchunk][blocks_per_chunk], zeroed per search: work items that have finished a block (flat walk: lock-step window)
    int32_t pace_window;      // blocks an item may run ahead of the slowest item of its chunk
    int32_t knob;             // developer switches (VS_BP_KNOB)
    unsigned long long* debug;    // optional (VS_BP_DEBUG=1): [8] consistency counters of the streamed walk
    unsigned long long* timing;   // optional (VS_BP_TIMING=1): [8] wave-cycles per phase, summed over waves: 0 item prologue, 1 list walk,
                                  // 2 wait at the barrier after the walk, 3 dense part, 4 epilogue; [5] = blocks x waves
};

// accumulators [RMAX + 1][QT + 1]: the extra row absorbs the pad postings of a binary list (document id RMAX)
template <int QT, int AM, int RMAX>
__host__ __device__ constexpr size_t bp_acc_bytes() { return (((size_t)(RMAX + 1) * (QT + 1) * (AM == AM_F64 ? 8 : 4)) + 15) & ~(size_t)15; }
// the tile's weights on the head columns as the MFMA B operand: [16 = slot + 8 * (hi | lo)][padded columns + 8] halves
__host__ __device__ inline size_t bp_head_lds(int n_head) { return n_head > 0 ? (size_t)16 * (bp_head_pad(n_head) + 8) * 2 : 0; }
template <int QT, int AM, int RMAX>
__host__ __device__ inline size_t bp_lds_bytes(int ent_cap, int n_head = 0) {
    return bp_acc_bytes<QT, AM, RMAX>() + (size_t)kBpCap * 8 + (size_t)QT * 16 + 64 * 4 + (size_t)kCutHistWords * 4 + (size_t)ent_cap * 8 + bp_head_lds(n_head);
}

__device__ __forceinline__ uint64_t make_key_fix(int32_t a, uint32_t row) {
    return ((uint64_t)((uint32_t)a ^ 0x80000000u) << 32) | (uint32_t)(~row);
}
__device__ __forceinline__ int32_t key_fix(uint64_t k) { return (int32_t)((uint32_t)(k >> 32) ^ 0x80000000u); }

// LDS byte offset of accumulator [document][slot]: document (one half of a packed id word) * pitch + slot offset, ONE instruction
__device__ __forceinline__ uint32_t acc_off_lo(uint32_t idw, uint32_t pitch, uint32_t so) {
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(idw), "v"(pitch), "v"(so));
    return r;
}
__device__ __forceinline__ uint32_t acc_off_hi(uint32_t idw, uint32_t pitch, uint32_t so) {
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(idw), "v"(pitch), "v"(so));
    return r;
}

// A record's loads as ONE asm statement each (global_load, block base in SGPRs + 32-bit byte offset); wait_loads<N> lets all but
// the N youngest loads land and ties the loaded registers to the wait, so that no use can be scheduled above it.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_rec48(u32x4& a, u32x4& b, u32x4& c, uint32_t off, unsigned long long base) {
    asm volatile("global_load_dwordx4 %0, %3, %4\n\tglobal_load_dwordx4 %1, %3, %4 offset:16\n\tglobal_load_dwordx4 %2, %3, %4 offset:32"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(off), "s"(base));
}
__device__ __forceinline__ void load_rec32(u32x4& a, u32x4& b, uint32_t off, unsigned long long base) {
    asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:16" : "=&v"(a), "=&v"(b) : "v"(off), "s"(base));
}
__device__ __forceinline__ void load_rec16(u32x4& a, uint32_t off, unsigned long long base) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(a) : "v"(off), "s"(base));
}
// (one overload per register count: naming the same variable twice in one asm makes the compiler copy it BEFORE the wait)
#define VS_WAIT_CASES(OPS)                                                                                              \
    switch (n) {                                                                                                        \
        case 1: asm volatile("s_waitcnt vmcnt(1)" : OPS); break;                                                      \
        case 2: asm volatile("s_waitcnt vmcnt(2)" : OPS); break;                                                      \
        case 3: asm volatile("s_waitcnt vmcnt(3)" : OPS); break;                                                      \
        case 4: asm volatile("s_waitcnt vmcnt(4)" : OPS); break;                                                      \
        case 5: asm volatile("s_waitcnt vmcnt(5)" : OPS); break;                                                      \
        case 6: asm volatile("s_waitcnt vmcnt(6)" : OPS); break;                                                      \
        case 7: asm volatile("s_waitcnt vmcnt(7)" : OPS); break;                                                      \
        case 8: asm volatile("s_waitcnt vmcnt(8)" : OPS); break;                                                      \
        case 9: asm volatile("s_waitcnt vmcnt(9)" : OPS); break;                                                      \
        case 10: asm volatile("s_waitcnt vmcnt(10)" : OPS); break;                                                    \
        case 11: asm volatile("s_waitcnt vmcnt(11)" : OPS); break;                                                    \
        case 12: asm volatile("s_waitcnt vmcnt(12)" : OPS); break;                                                    \
        case 13: asm volatile("s_waitcnt vmcnt(13)" : OPS); break;                                                    \
        case 14: asm volatile("s_waitcnt vmcnt(14)" : OPS); break;                                                    \
        case 15: asm volatile("s_waitcnt vmcnt(15)" : OPS); break;                                                    \
        case 16: asm volatile("s_waitcnt vmcnt(16)" : OPS); break;                                                    \
        case 17: asm volatile("s_waitcnt vmcnt(17)" : OPS); break;                                                    \
        case 18: asm volatile("s_waitcnt vmcnt(18)" : OPS); break;                                                    \
        case 19: asm volatile("s_waitcnt vmcnt(19)" : OPS); break;                                                    \
        case 20: asm volatile("s_waitcnt vmcnt(20)" : OPS); break;                                                    \
        case 21: asm volatile("s_waitcnt vmcnt(21)" : OPS); break;                                                    \
        case 22: asm volatile("s_waitcnt vmcnt(22)" : OPS); break;                                                    \
        case 23: asm volatile("s_waitcnt vmcnt(23)" : OPS); break;                                                    \
        case 24: asm volatile("s_waitcnt vmcnt(24)" : OPS); break;                                                    \
        default: asm volatile("s_waitcnt vmcnt(0)" : OPS); break;                                                       \
    }
#define VS_OPS3 "+v"(a), "+v"(b), "+v"(c)
#define VS_OPS2 "+v"(a), "+v"(b)
#define VS_OPS1 "+v"(a)
__device__ __forceinline__ void wait_loads(int n, u32x4& a, u32x4& b, u32x4& c) { VS_WAIT_CASES(VS_OPS3) }      // n is a constant after unrolling
__device__ __forceinline__ void wait_loads(int n, u32x4& a, u32x4& b) { VS_WAIT_CASES(VS_OPS2) }
__device__ __forceinline__ void wait_loads(int n, u32x4& a) { VS_WAIT_CASES(VS_OPS1) }
#undef VS_OPS1
#undef VS_OPS2
#undef VS_OPS3
#undef VS_WAIT_CASES
// accumulate into LDS at a byte address (ds_add_u32 / ds_add_f64, no return)
__device__ __forceinline__ void lds_add(uint32_t addr, int32_t v) {
    __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) int32_t*>(addr), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_add(uint32_t addr, double v) {
    __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) double*>(addr), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int U>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t x) {          // lane U of every quad -> the whole quad: a VALU move, no LDS
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, U | (U << 2) | (U << 4) | (U << 6), 0xF, 0xF, false);
}

// LG = lanes per posting list (8 for the long lists of a valued index, 1 for the short lists of the bag-of-token index),
// RMAX = block capacity in documents
// HD = 1: the index has head columns (dense strips) multiplied in this kernel, one tile at a time; HD = 2: their part of the sums was
// computed by the head pre-pass (bp_head.h) and is added in the epilogue; 0 compiles both out.
template <int VM, int QT, int AM, int LG, int RMAX, int NB = kBpNB, int HD = 0>
__global__ __launch_bounds__(kScanThreads) void bp_walk_topk(BpArgs a) {
    static_assert(!HD || (AM == AM_FIX && VM != VM_BIN && QT == 8), "dense strips: valued filter walk only");
    static_assert(bp_acc_bytes<QT, AM, RMAX>() >= kBpSortBytes, "the accumulator area holds the 8192-slot entry sort");
    static_assert(VM != VM_BIN || RMAX == kBpRowsMaxBin, "pad postings of a binary list carry document id kBpRowsMaxBin");
    static_assert(NB % LG == 0 || LG % NB == 0 || (LG == 8 && NB <= 8), "lane l of a group owns the directory words of lists l, l + LG, ... of a slot");
    using acc_t = typename std::conditional<AM == AM_F64, double, int32_t>::type;
    constexpr int PITCH = QT + 1;                 // accumulator row pitch in elements: a document's row starts an odd number of words
                                                  // after its neighbour's, so the adds of a wave spread over all LDS banks
    constexpr uint32_t PITCHB = PITCH * sizeof(acc_t);
    constexpr int RS = bp_rec_bytes(VM);
    constexpr int OWN = NB >= LG ? NB / LG : 1;   // directory words a lane owns per slot (lanes >= NB of a wide group own none)
    constexpr bool kWide = LG == 8 && NB > 4;     // 8-lane groups, 5 .. 8 lists per slot: lane l owns list l's word (else: every quad holds all NB)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    acc_t* acc = reinterpret_cast<acc_t*>(smem);                                            // [RMAX + 1][PITCH]
    uint64_t* sortbuf = reinterpret_cast<uint64_t*>(smem + bp_acc_bytes<QT, AM, RMAX>());   // [kBpCap]
    unsigned long long* tau = reinterpret_cast<unsigned long long*>(sortbuf + kBpCap);      // [QT]
    unsigned long long* upper_sh = tau + QT;                                                // [QT] exclusive upper bounds ("search after")
    int* scratch = reinterpret_cast<int*>(upper_sh + QT);                                   // [48]
    unsigned int* ccnt = reinterpret_cast<unsigned int*>(scratch + 48);                     // [QT]
    unsigned int* chi = ccnt + 8;                                                           // [QT <= 8] the counters' high halves at the end of the previous block (epilogue)
    uint32_t* cut_hist = reinterpret_cast<uint32_t*>(scratch + 64);                         // [kCutHistWords] the cut's histograms (wg_cut_topk)
    uint2* ent = reinterpret_cast<uint2*>(scratch + 64 + kCutHistWords);                    // [ent_cap]: x = column | slot byte offset << 16, y = weight bits
    _Float16* hw = reinterpret_cast<_Float16*>(ent + a.ent_cap);                            // [16][ldb] the tile's weights on the head columns (bp_head_lds)

    const int tid = threadIdx.x;
    const int gid = tid / LG, gl = tid % LG;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;     // LDS address of the accumulators
    const int K = a.k;
    uint64_t* my_gcand = a.gcand + (size_t)blockIdx.x * QT * kBpCap;
    const int64_t n_blocks = (a.n_rows + a.rows - 1) / a.rows;
    int n_tiles_all = a.n_tiles_dev ? a.n_tiles_dev[0] : a.n_tiles;
    if (a.tile_cnt > 0) n_tiles_all = max(0, min(n_tiles_all - a.tile0, a.tile_cnt));       // a pass over a range of tiles
    const int64_t items = (int64_t)n_tiles_all * a.nchunk;
    bool pace_off = false;                      // the lock-step wait timed out once (pace_wait): this workgroup runs free from then on
    const size_t dir_ld = (size_t)a.n_cols + 1;

    const unsigned long long k_rt0 = a.timing ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // Work items = (tile, chunk), taken round-robin (the host picks nchunk so that an XCD keeps to few chunks, see the launch)
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int tile_rel = (int)(item / a.nchunk), c = (int)(item % a.nchunk);
        const int tile = (a.tile_cnt > 0 ? a.tile0 : 0) + tile_rel;
        const int q0 = a.tiles[tile].x, nq = a.tiles[tile].y;
        long long tm = a.timing ? (long long)__builtin_readcyclecounter() : 0;
        uint32_t tacc[6] = {0u, 0u, 0u, 0u, 0u, 0u};              // (flushed once per item: 6 atomics per wave)
        auto lap = [&](int phase) {
            if (a.timing) {
                const long long now = (long long)__builtin_readcyclecounter();
                tacc[phase] += (uint32_t)(now - tm);
                tm = now;
            }
        };
        const int64_t b0 = (int64_t)c * a.blocks_per_chunk, b1 = min(n_blocks, b0 + a.blocks_per_chunk);
        __syncthreads();
        const int64_t e0 = a.qptr[q0], e1 = a.qptr[q0 + nq];
        int n_ent = (int)(e1 - e0);
        const int n_head = HD ? a.n_head : 0;
        const int ldb = bp_head_pad(n_head) + 8;                 // + 8 halves: the 16 rows a b128 operand read touches fall in different banks
        if (HD == 1 && n_head > 0) {
            uint32_t* z = reinterpret_cast<uint32_t*>(hw);
            for (int i = tid; i < 16 * ldb / 2; i += kScanThreads) z[i] = 0u;
            __syncthreads();
        }
        // Entries sorted by column (the accumulator area doubles as the sort buffer): neighbouring groups then read neighbouring
        // directory words and neighbouring posting lists -- the walk over the block's records is a forward sweep with gaps.
        {
            uint64_t* skey = reinterpret_cast<uint64_t*>(acc);
            for (int i = tid; i < 8192; i += kScanThreads) {
                uint64_t key = 0;
                if (i < n_ent) {
                    const int64_t e = e0 + i;
                    int qs = 0;
                    while (e >= a.qptr[q0 + qs + 1]) ++qs;
                    float w = a.qvals[e];
                    if constexpr (AM == AM_FIX) w *= a.qscale[q0 + qs];          // power of two: exact
                    const uint32_t col = (uint32_t)a.qcols[e];
                    const uint32_t hx = n_head > 0 ? a.hmap[col] : 0xFFFFu;
                    if (hx != 0xFFFFu) {
                        // a head column: no list to walk, its weight joins the dense part as two fp16 numbers hi + lo (22 bits)
                        // (HD = 2: the head pre-pass has multiplied it already)
                        if constexpr (HD == 1) {
                            const float ws = w * a.head_pre;
                            const _Float16 hi = (_Float16)ws;
                            hw[qs * ldb + hx] = hi;
                            hw[(8 + qs) * ldb + hx] = (_Float16)(ws - (float)hi);
                        }
                    }
                    else {
                        // Lists of similar length are dealt together: the sort key leads with a length class (rounds of the group's
                        // LG x 8 postings an average block's list of this column takes: 1, 2, 3-4, 5-8, more), so a chunk's lists
                        // finish their rounds together instead of one long list keeping a lane group's slot alone (skewed
                        // vocabularies; a uniform corpus has one class).  Within a class the entries stay sorted by column.
                        uint32_t cls = 0;
                        if (a.df) {
                            const float per_block = (float)a.df[col] * (float)a.rows / (float)max(a.n_rows, (int64_t)1);
                            const float rounds = per_block * (1.f / (8.f * LG));
                            cls = rounds <= 1.f ? 0u : rounds <= 2.f ? 1u : rounds <= 4.f ? 2u : rounds <= 8.f ? 3u : 4u;
                        }
                        key = ((uint64_t)cls << 56) | ((uint64_t)col << 40) | ((uint64_t)qs << 32) | (uint64_t)__float_as_uint(w);
                    }
                }
                skey[i] = key;
            }
            wg_sort_desc<kScanThreads>(skey, 8192, tid);
            if (n_head > 0) {                                   // entries left after the head columns: the non-zero keys (sorted first)
                __shared__ int n_left;
                if (tid == 0) n_left = 0;
                __syncthreads();
                for (int i = tid; i < n_ent; i += kScanThreads)
                    if (skey[i] != 0ull && (i + 1 == 8192 || skey[i + 1] == 0ull)) n_left = i + 1;
                __syncthreads();
                n_ent = n_left;
            }
            for (int i = tid; i < n_ent; i += kScanThreads) {
                const uint64_t key = skey[i];
                ent[i] = make_uint2(((uint32_t)(key >> 40) & 0xFFFFu) | ((uint32_t)((key >> 32) & 0xFFu) * (uint32_t)sizeof(acc_t) << 16), (uint32_t)key);
            }
            __syncthreads();
        }
        for (int i = tid; i < RMAX * PITCH; i += kScanThreads) acc[i] = (acc_t)0;
        if (tid < QT) { tau[tid] = 0ull; ccnt[tid] = 0u; chi[tid] = 0u; }
        if (tid < QT) upper_sh[tid] = (a.upper && tid < nq) ? a.upper[q0 + tid] : ~0ull;
        if (tid < 2) scratch[40 + tid] = 0;                   // chunk counters of even / odd blocks (below)
        __syncthreads();

        // Entries are dealt to the waves in CHUNKS of CW = (groups per wave) x NB consecutive entries -- one slot of a wave.  A wave's
        // first chunk of a block is its own number; the following ones it takes from a counter in LDS as it goes (one atomic per
        // slot, fetched a slot ahead), so the waves reach the block's barrier within one slot of each other whatever their lists'
        // lengths (static dealing left 11 % of the wave-cycles of the valued walk, 18 % of the binary one, waiting there).
        constexpr int GPW = 64 / LG, CW = GPW * NB, NW = kScanThreads / 64;
        const int wv_id = tid >> 6, gw = gid & (GPW - 1);
        int* chunk_cnt = scratch + 40;
        auto grab = [&](int par) {
            int v = 0;
            if ((tid & 63) == 0) v = atomicAdd(&chunk_cnt[par], 1);
            return NW + __builtin_amdgcn_readfirstlane(v);
        };
        // directory pairs of a block's first slot: fetched while the previous block is finished (they stay in flight across its epilogue)
        uint32_t nd[OWN];
        auto first_pairs = [&](int64_t bb, int li0) {
            const uint32_t* dirn = a.dir + (size_t)bb * dir_ld;
#pragma unroll
            for (int o = 0; o < OWN; ++o) {
                const int lu = kWide ? gl : ((LG >= NB ? (gl & (NB - 1)) : gl) + LG * o);       // LG >= NB, not wide: every quad of the group holds all NB words
                const int e = li0 * CW + lu * GPW + gw;
                nd[o] = 0;
                if (li0 >= 0 && lu < NB && e < n_ent) nd[o] = dirn[ent[e].x & 0xFFFFu];
            }
        };
        // (short binary lists: one or two slots per block, the directory latency would be exposed once per block; with the long
        //  lists of a valued index the pairs are fetched at block start -- holding them across the epilogue costs more than it hides)
        constexpr bool kPairsAhead = LG == 1;
        // ... and for the headless valued kernels too since the waves take chunks dynamically (B = 16 at 21 M docs: -4 %, B = 1024:
        // -0.3 %; before, with static dealing, holding the words across the epilogue cost more than it hid).  Not with dense chunks:
        // there the kind of a wave's first chunk depends on the block.
        const bool pairs_ahead = kPairsAhead || HD != 1;
        if (pairs_ahead && b0 < b1) first_pairs(b0, wv_id);          // (no dense chunks: chunk = list chunk)
        for (int64_t b = b0; b < b1 || b == b0; ++b) {
            const bool have = b < b1;
            const int rows_b = have ? (int)min((int64_t)a.rows, a.n_rows - b * a.rows) : 0;
            if (have) {
                const uint32_t* dirb = a.dir + (size_t)b * dir_ld;
                // block-uniform base pointer in SGPRs + 32-bit record offsets per lane
                // block-uniform base pointer in SGPRs + 32-bit byte offsets per lane
                const unsigned long long pb = (unsigned long long)(a.rec + (size_t)a.base[b] * RS);
                const unsigned long long brec = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32)) << 32) |
                                                (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pb);   // (the builtin returns int)
                // A slot = the group's next NB entries; lane l owns the directory pairs (first, end) of entries l, l + LG, ... of the
                // slot and fetches the next slot's pairs while the current one is walked.  A lane takes one record (8 postings) of a
                // list per round; the NB lists of a slot are loaded before any multiply-add.  Loads are unconditional (a lane past
                // its list's end re-reads the record at the end: always inside the array) so that no loaded register is merged with
                // an older value; only the adds are predicated.
                // Chunks of a block = list chunks (CW entries each) and, with head columns, dense chunks (64 documents x all head
                // columns on the matrix cores) INTERLEAVED in one queue: both add into the same accumulators, no barrier between
                // them, and a wave streaming a strip from L2 runs next to waves whose time goes into LDS adds.
                const int n_lc = (n_ent + CW - 1) / CW;
                constexpr int kDenseDocs = 64;                      // documents of a dense chunk (4 operand rows; 32: re-reads the weights twice as often and streams worse -- 489 vs 370 ms at 21 M docs)
                const int n_dc = (HD == 1 && n_head > 0) ? (rows_b + kDenseDocs - 1) / kDenseDocs : 0;
                const int n_ch = n_lc + n_dc;
                // chunk c is dense iff floor((c + 1) n_dc / n_ch) > floor(c n_dc / n_ch); floor(c n_dc / n_ch) dense chunks precede it
                auto dense_before = [&](int c) { return (int)(((uint32_t)c * (uint32_t)n_dc) / (uint32_t)max(n_ch, 1)); };
                auto list_index = [&](int c) {
                    if constexpr (HD != 1) return c < n_ch ? c : -1;              // (no dense chunks: no division in the headless kernels)
                    else return (c < n_ch && dense_before(c + 1) == dense_before(c)) ? c - dense_before(c) : -1;
                };
                [[maybe_unused]] auto dense_chunk = [&](int dj) {
                    if constexpr (HD == 1) {
                        static_assert(QT == 8 && RMAX == 16 * 128, "dense part: 8 slots x (hi, lo) = the 16 columns of the MFMA");
                        // [64 documents] x [head columns] (fp16 strip, MFMA A-operand order) times [head columns] x [8 slots x (hi, lo)]
                        // (the tile's weights) -- v_mfma_f32_16x16x32_f16, fp32 accumulate; the strip streams from L2 / Infinity
                        // Cache with the next two k-steps (8 KB a wave) in flight.  Each sum is scaled back (power of two), truncated
                        // once and added to the accumulators (hi and lo parts separately).
                        using h8 = __attribute__((ext_vector_type(8))) _Float16;
                        using f4 = __attribute__((ext_vector_type(4))) float;
                        const int ln = tid & 63;
                        const int ks = bp_head_pad(n_head) / 32, mbk = a.rows / 16;
                        const _Float16* bp = hw + (ln & 15) * ldb + 8 * (ln >> 4);
                        const int slot = ln & 7;
                        constexpr int MR = kDenseDocs / 16;                 // operand rows of the chunk
                        const int d0 = dj * kDenseDocs;
                        const uint4* sp = reinterpret_cast<const uint4*>(a.strip) + ((size_t)b * ks * mbk + (size_t)(d0 >> 4)) * 64 + ln;
                        f4 c[MR];
#pragma unroll
                        for (int m = 0; m < MR; ++m) c[m] = f4{0.f, 0.f, 0.f, 0.f};
                        uint4 s0[MR], s1[MR], s2[MR];                       // k-steps j, j + 1, j + 2
#pragma unroll
                        for (int m = 0; m < MR; ++m) { s0[m] = sp[(size_t)m * 64]; s1[m] = s0[m]; s2[m] = s0[m]; }
                        if (ks > 1) {
#pragma unroll
                            for (int m = 0; m < MR; ++m) s1[m] = sp[((size_t)mbk + m) * 64];
                        }
#pragma unroll 1
                        for (int j = 0; j < ks; ++j) {
                            if (j + 2 < ks) {
#pragma unroll
                                for (int m = 0; m < MR; ++m) s2[m] = sp[((size_t)(j + 2) * mbk + m) * 64];
                            }
                            const h8 bf = *reinterpret_cast<const h8*>(bp + 32 * j);
#pragma unroll
                            for (int m = 0; m < MR; ++m) {
                                h8 af;
                                __builtin_memcpy(&af, &s0[m], 16);
                                c[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, c[m], 0, 0, 0);
                            }
#pragma unroll
                            for (int m = 0; m < MR; ++m) { s0[m] = s1[m]; s1[m] = s2[m]; }
                        }
                        // C: lane holds rows 4 * (ln >> 4) + i, column ln & 15 = slot + 8 * (hi | lo)
#pragma unroll
                        for (int m = 0; m < MR; ++m) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int d = d0 + m * 16 + 4 * (ln >> 4) + i;
                                atomicAdd(&acc[d * PITCH + slot], (int32_t)(c[m][i] * a.head_mul));
                            }
                        }
                    }
                };
                const int par = (int)(b & 1);
                int cur = wv_id, nxt = grab(par);
                if (!pairs_ahead) first_pairs(b, list_index(cur));
                while (cur < n_ch) {
                    const int li = list_index(cur), li_n = list_index(nxt);
                    uint32_t cd[OWN];
#pragma unroll
                    for (int o = 0; o < OWN; ++o) {
                        cd[o] = nd[o];
                        nd[o] = 0;
                        const int lu = kWide ? gl : ((LG >= NB ? (gl & (NB - 1)) : gl) + LG * o);
                        const int e = li_n * CW + lu * GPW + gw;
                        if (li_n >= 0 && lu < NB && e < n_ent) nd[o] = dirb[ent[e].x & 0xFFFFu];
                    }
                    if (li < 0) {                                   // a dense chunk (the words fetched above wait in nd for the next list chunk)
                        const long long t_d = a.timing ? (long long)__builtin_readcyclecounter() : 0;
                        dense_chunk(dense_before(cur));
                        if (a.timing) { const uint32_t dt = (uint32_t)((long long)__builtin_readcyclecounter() - t_d); tacc[3] += dt; tacc[1] -= dt; }
                        cur = nxt;
                        nxt = grab(par);
                        continue;
                    }
                    const int cbase = li * CW;
                    cur = nxt;
                    nxt = grab(par);
                    uint32_t rec[NB], end[NB];
                    uint2 en[NB];
                    bool more = false;
                    uint32_t bd[NB];                                // list u of the slot: its directory word sits in lane u of every quad (LG >= 4) / in register u
                    if constexpr (LG == 1) {
#pragma unroll
                        for (int u = 0; u < NB; ++u) bd[u] = cd[u];
                    } else if constexpr (kWide) {
                        // lane l of the 8-lane group holds list l's word: the other quad's four come over with a row shift, then quad broadcasts
                        const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cd[0], 0x104, 0xF, 0xF, false);      // row_shl:4: lane i <- lane i + 4
                        const uint32_t dn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cd[0], 0x114, 0xF, 0xF, false);      // row_shr:4: lane i <- lane i - 4
                        const bool upper = (gl & 4) != 0;
                        const uint32_t oth = upper ? dn : up;
                        const uint32_t own4[4] = {quad_bcast<0>(cd[0]), quad_bcast<1>(cd[0]), quad_bcast<2>(cd[0]), quad_bcast<3>(cd[0])};
                        const uint32_t oth4[4] = {quad_bcast<0>(oth), quad_bcast<1>(oth), quad_bcast<2>(oth), quad_bcast<3>(oth)};
#pragma unroll
                        for (int u = 0; u < NB; ++u) bd[u] = (u < 4) == upper ? oth4[u & 3] : own4[u & 3];
                    } else {
                        static_assert(NB == 4 && OWN == 1, "quad broadcast of four words");
                        bd[0] = quad_bcast<0>(cd[0]); bd[1] = quad_bcast<1>(cd[0]); bd[2] = quad_bcast<2>(cd[0]); bd[3] = quad_bcast<3>(cd[0]);
                    }
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        const uint32_t lo = (bd[u] >> 12) << a.al_shift, hi = lo + (bd[u] & kBpDirRecMask);
                        rec[u] = lo + gl; end[u] = hi;
                        more = more || (rec[u] < end[u]);
                        en[u] = ent[min(cbase + u * GPW + gw, n_ent - 1)];          // (column | slot offset << 16, weight): LDS broadcast per group
                    }
                    // the 8 postings of one record into the accumulators of slot / weight `e`
                    auto add_record = [&](const u32x4& idv, const u32x4& vav, const u32x4& vbv, const uint2 e) {
                        const float wq = __uint_as_float(e.y);
                        uint32_t so = (e.x >> 16) + lds0;                           // LDS byte address of [document 0][slot]
                        uint32_t pitchb = PITCHB;
#ifdef VS_BP_EXPERIMENT
                        // throw-away build #4 (WRONG results on purpose): every 32-lane half of a ds_add hits 32
                        // different banks (lane -> bank, the slot picks the row), same instruction stream
                        if (a.knob & 16) { pitchb = 0u; so = lds0 + (e.x >> 16) * 32u + (uint32_t)(tid & 31) * 4u; }
                        // ... at most 2 lanes of a half per bank
                        if (a.knob & 32) { pitchb = 0u; so = lds0 + (e.x >> 16) * 32u + (uint32_t)(tid & 15) * 4u + (uint32_t)(tid & 16) * 64u; }
#endif
                        const uint32_t dw[4] = {idv.x, idv.y, idv.z, idv.w};
                        float vv[8];
                        if constexpr (VM == VM_F32) {
                            vv[0] = __uint_as_float(vav.x); vv[1] = __uint_as_float(vav.y); vv[2] = __uint_as_float(vav.z);
                            vv[3] = __uint_as_float(vav.w); vv[4] = __uint_as_float(vbv.x); vv[5] = __uint_as_float(vbv.y);
                            vv[6] = __uint_as_float(vbv.z); vv[7] = __uint_as_float(vbv.w);
                        } else if constexpr (VM == VM_F16) {
                            // weight x fp16 value in ONE instruction (v_fma_mix_f32 converts the selected half on the way in; + 0: the
                            // product is rounded once, exactly as convert-then-multiply): the walk is bound by VALU issue (77 % busy)
                            const uint32_t hw2[4] = {vav.x, vav.y, vav.z, vav.w};
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(vv[2 * t]) : "v"(wq), "v"(hw2[t]));
                                asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(vv[2 * t + 1]) : "v"(wq), "v"(hw2[t]));
                            }
                        }
                        [[maybe_unused]] int32_t wi = 0;
                        [[maybe_unused]] double wd = 0.0;
                        if constexpr (VM == VM_BIN && AM == AM_FIX) wi = (int32_t)wq;
                        if constexpr (VM == VM_BIN && AM == AM_F64) wd = (double)wq;
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const uint32_t off = (t & 1) ? acc_off_hi(dw[t >> 1], pitchb, so) : acc_off_lo(dw[t >> 1], pitchb, so);
                            if constexpr (VM == VM_BIN) {
                                // (pad postings of a binary list carry document id RMAX: the scratch row behind the accumulators)
                                if constexpr (AM == AM_FIX) lds_add(off, wi);
                                else lds_add(off, wd);
                            } else {
                                const float prod = VM == VM_F16 ? vv[t] : wq * vv[t];     // (fp16 records: already the product)
                                if constexpr (AM == AM_FIX) lds_add(off, (int32_t)prod);
                                else lds_add(off, (double)prod);
                            }
                        }
                    };
                    // A list longer than the group's LG records per round (64 postings: 1 list in 15 at 53 postings a list) would cost
                    // the whole wave a second round of loads for a handful of lanes -- 9 slots in 10 have such a list somewhere in the
                    // wave, and the walk is bound by exactly that latency.  So a lane also loads, in the SAME round, the second record
                    // it owes to the first of its lists that has one (kTail): second rounds become rare (two long lists in one group's
                    // slot, or a list beyond 128 postings).
                    constexpr bool kTail = LG > 1 && VM != VM_BIN;
                    while (__builtin_amdgcn_ballot_w64(more)) {
                        // The loads of a round are issued back to back -- hand-written: the compiler sinks plain loads into the
                        // predicated blocks below and then waits for each list separately -- and each list waits only for its own.
                        u32x4 ids[NB], va[NB], vb[NB];
                        [[maybe_unused]] u32x4 tid_r, tva_r, tvb_r = u32x4{0u, 0u, 0u, 0u};
                        [[maybe_unused]] uint32_t trec = 0, tend = 0;
                        [[maybe_unused]] uint2 ten = en[0];
                        [[maybe_unused]] int tu = -1;
#pragma unroll
                        for (int u = 0; u < NB; ++u) {
                            const uint32_t off = __umul24(min(rec[u], end[u]), (uint32_t)RS);
                            if constexpr (VM == VM_F32) load_rec48(ids[u], va[u], vb[u], off, brec);
                            else if constexpr (VM == VM_F16) load_rec32(ids[u], va[u], off, brec);
                            else load_rec16(ids[u], off, brec);
                            if constexpr (kTail) {
                                if (tu < 0 && rec[u] + LG < end[u]) { tu = u; trec = rec[u] + LG; tend = end[u]; ten = en[u]; }
                            }
                        }
                        if constexpr (kTail) {
                            const uint32_t off = __umul24(min(trec, tend), (uint32_t)RS);      // (no tail: record 0 of the block, not added)
                            if constexpr (VM == VM_F32) load_rec48(tid_r, tva_r, tvb_r, off, brec);
                            else load_rec32(tid_r, tva_r, off, brec);
                        }
                        constexpr int kPer = VM == VM_F32 ? 3 : (VM == VM_F16 ? 2 : 1);       // loads per record
                        constexpr int kAfter = kTail ? kPer : 0;                                // issued after the lists' own
                        more = false;
#pragma unroll
                        for (int u = 0; u < NB; ++u) {
                            if constexpr (VM == VM_F32) wait_loads((NB - 1 - u) * 3 + kAfter, ids[u], va[u], vb[u]);
                            else if constexpr (VM == VM_F16) wait_loads((NB - 1 - u) * 2 + kAfter, ids[u], va[u]);
                            else wait_loads(NB - 1 - u, ids[u]);
                            if (rec[u] < end[u]) add_record(ids[u], va[u], vb[u], en[u]);
                            rec[u] += (kTail && tu == u) ? 2 * LG : LG;
                            more = more || (rec[u] < end[u]);
                        }
                        if constexpr (kTail) {
                            if constexpr (VM == VM_F32) wait_loads(0, tid_r, tva_r, tvb_r);
                            else wait_loads(0, tid_r, tva_r);
                            if (trec < tend) add_record(tid_r, tva_r, tvb_r, ten);
                        }
                    }
                }
                if (pairs_ahead && b + 1 < b1) first_pairs(b + 1, wv_id);
            }
            // thresholds other items of the same queries have published meanwhile (read before the barrier: the latency hides in it)
            if (a.gtau && tid < nq) { const unsigned long long g = a.gtau[q0 + tid]; if (g > tau[tid]) tau[tid] = g; }
            lap(1);
            // Lock step (BpArgs::pace): the tiles of a chunk of blocks sweep the same blocks; one that falls behind loses the L2 /
            // Infinity-Cache copies the pack left behind and falls further behind (per-workgroup clocks of a 4 M-doc run: 250 of 256
            // at 31 ms, a handful at 41 ms -- and the launch ends with the last).  An item counts its arrival at the end of block j
            // and waits while the slowest item of its chunk has not reached block j - window.  Only when every item is resident.
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
            __syncthreads();
            if (tid == 0) scratch[40 + (int)((b + 1) & 1)] = 0;     // the next block's chunk counter (its last user was block b - 1)
            lap(2);
            lap(3);
            // epilogue: 1024 documents at a time, one per thread: its QT sums -> order keys -> candidates; prune when a buffer could overflow
            [[maybe_unused]] uint32_t nx0 = 0u, nx4 = 0u;                 // (HD = 2: the head sums' lines of the second round, asked for in the first)
            for (int d0 = 0; d0 < rows_b || d0 == 0; d0 += kScanThreads) {
                const int d = d0 + tid;
                const bool more = d0 + kScanThreads < rows_b;         // another round of this block follows
                const uint32_t inc = more ? 1u : 0x10000u;
                // the score halves of the thresholds, once per round: nearly every (document, slot) ends at ONE 32-bit compare
                uint32_t thi[QT];
#pragma unroll
                for (int q = 0; q < QT; ++q) thi[q] = (uint32_t)(tau[q] >> 32);
                if (d < rows_b) {
                    const int64_t row = b * a.rows + d;
                    acc_t* pa = acc + (size_t)d * PITCH;
                    acc_t sums[QT];
                    [[maybe_unused]] int32_t pre[QT];
                    if constexpr (HD == 2) {
                        // the dense part of this document's sums (head pre-pass): 8 loads of a slot's 16-document run, in flight together
                        // (tried and dropped, 21 M docs: the same loads issued BEFORE the block's barrier, or for both of the thread's
                        //  documents at once right behind it -- the walk went from 126 ms to 161 - 186, the waves' wait at the barrier
                        //  from 6 k to 22 - 26 k cycles a block: docs/EXPERIMENTS.md)
                        // (uint16 sums read as the DWORD that holds the document and its neighbour: 2-byte loads took the epilogue from 8.5 k to
                        //  13.9 k cycles a block)
                        const uint32_t* hp = reinterpret_cast<const uint32_t*>(a.head_out + head_out_offset((int64_t)tile_rel, n_blocks, b, a.rows, d & ~1));
                        const uint32_t sh = (uint32_t)(d & 1) * 16u;
                        // (a 16-document group's sums are two cache lines, slots 0 - 3 and 4 - 7: four loads in flight to ONE missing line wait for
                        //  each other in the L1 -- the lines are asked for once each, the other six loads hit behind them)
                        uint32_t hw[QT];
#pragma unroll
                        for (int q = 0; q < QT; ++q) hw[q] = 0u;
                        if (!(a.knob & 256)) {                                  // (VS_BP_KNOB=256: ablation, WRONG results -- what the scratch reads cost)
                            if (d0 == 0) {
                                hw[0] = hp[0];
                                if constexpr (QT > 4) hw[4] = hp[32];
                                // ... and the two lines of the thread's SECOND document (the block's next round), asked for with them: four
                                // different lines, in flight together (one wait); the next round finds them in its registers
                                if (more && d + kScanThreads < rows_b) {
                                    const uint32_t* hn = reinterpret_cast<const uint32_t*>(a.head_out + head_out_offset((int64_t)tile_rel, n_blocks, b, a.rows, (d + kScanThreads) & ~1));
                                    nx0 = hn[0];
                                    if constexpr (QT > 4) nx4 = hn[32];
                                }
                                asm volatile("s_waitcnt vmcnt(0)" : "+v"(hw[0]), "+v"(hw[QT > 4 ? 4 : 0]), "+v"(nx0), "+v"(nx4) :: "memory");
                            } else {
                                hw[0] = nx0;
                                if constexpr (QT > 4) hw[4] = nx4;
                            }
#pragma unroll
                            for (int q = 1; q < QT; ++q)
                                if (q != 4) hw[q] = hp[q * 8];
                        }
#pragma unroll
                        for (int q = 0; q < QT; ++q) pre[q] = q < nq ? (int32_t)(((hw[q] >> sh) & 0xFFFFu) << kHeadOutShift) : 0;
                    }
#pragma unroll
                    for (int q = 0; q < QT; ++q) sums[q] = pa[q];         // independent reads, in flight together
                    if constexpr (HD == 2 && AM == AM_FIX) {
#pragma unroll
                        for (int q = 0; q < QT; ++q) sums[q] += (acc_t)pre[q];
                    }
#pragma unroll
                    for (int q = 0; q < QT; ++q) pa[q] = (acc_t)0;
#pragma unroll
                    for (int q = 0; q < QT; ++q) {                        // (slots >= nq are never written: a ragged tile skips them)
                        uint32_t hi;
                        if constexpr (AM == AM_F64) hi = flip_f32((float)sums[q]);
                        else hi = (uint32_t)sums[q] ^ 0x80000000u;
                        if (q < nq && hi >= thi[q]) {
                            const uint64_t key = ((uint64_t)hi << 32) | (uint32_t)(~(uint32_t)row);
                            if (key > tau[q] && key < upper_sh[q]) {
                                const uint32_t old = atomicAdd(&ccnt[q], inc);                 // (low half: first rounds, high half: last rounds -- below)
                                my_gcand[(size_t)q * kBpCap + (old & 0xFFFFu) + (old >> 16)] = key;
                            }
                        }
                    }
                }
                // Whether a buffer has to be cut is ONE decision of the workgroup (barriers sit behind it), read by every thread from the
                // counters behind the round's barrier: a block's first round counts in the LOW half of a slot's counter, its last round in
                // the HIGH half; behind the first round the waves already in the second change the high halves only, and the count is low
                // half + the high half as it stood at the end of the previous block (chi[]) -- bp_quad.h, docs/EXPERIMENTS.md round 6
                __syncthreads();
                // (VS_BP_KNOB = 128 + 4096 n, tests: one wave reads the counters n x 512 cycles late; a scalar branch -- s_sleep ignores exec)
                if ((a.knob & 128) && __builtin_amdgcn_readfirstlane((tid >> 6)) == 5)
                    for (int i = 0; i < (a.knob >> 12); ++i) __builtin_amdgcn_s_sleep(8);
                const bool last = b + 1 >= b1 && !more;
                // (the counters in one round of reads; every vector instruction of a thread costs a round ~ 16 cycles -- 4 waves a SIMD: low + high
                //  of a counter is one v_dot2_u32_u16, the limits one compare of the maximum)
                uint32_t cmax = 0u;
                {
                    uint32_t w[QT];
#pragma unroll
                    for (int q = 0; q < QT; ++q) w[q] = ccnt[q];
                    if (more) {
                        uint32_t h[QT];
#pragma unroll
                        for (int q = 0; q < QT; ++q) h[q] = chi[q];
#pragma unroll
                        for (int q = 0; q < QT; ++q) cmax = max(cmax, __builtin_amdgcn_udot2(__builtin_bit_cast(walk_us2, w[q]), walk_us2{1, 0}, h[q], false));
                    } else {
#pragma unroll
                        for (int q = 0; q < QT; ++q) cmax = max(cmax, __builtin_amdgcn_udot2(__builtin_bit_cast(walk_us2, w[q]), walk_us2{1, 1}, 0u, false));
                        if (tid < QT) chi[tid] = ccnt[tid] >> 16;             // (the next block's first round reads it two barriers from here)
                    }
                }
                const bool any = last || cmax > (uint32_t)(kBpCap - kScanThreads);
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
                            // (a cut needs the K best as a set and the K-th key: radix select, topk_keys.h, instead of the 66-stage sort)
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
            }
            lap(4);
            tacc[5] += 1u;
            if (b + 1 >= b1) break;
        }
        if (a.timing && (tid & 63) == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) atomicAdd(a.timing + i, (unsigned long long)tacc[i]);
        }
    }
    if (a.timing && threadIdx.x == 0) {        // per workgroup: 100 MHz ticks, shader cycles, where it ran (XCC_ID, HW_ID)
        a.timing[16 + 4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime() - k_rt0;
        a.timing[17 + 4 * blockIdx.x] = k_rt0;                       // (absolute start, 100 MHz ticks)
        a.timing[18 + 4 * blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20);
        a.timing[19 + 4 * blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}


}  // namespace vs
