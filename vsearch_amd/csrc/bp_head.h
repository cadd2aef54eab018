// bp_head.h -- the HEAD PRE-PASS of a corpus with head columns (skewed vocabularies): the dense part of the filter scores as ONE
// matrix product per range of query tiles, before the list walk.
//
// bp_walk_topk<.., HD = 1> multiplies a block's fp16 strip [documents x head columns] with ITS tile's weights [head columns x 8 slots
// x (hi, lo)] on the matrix cores -- and streams the strip (2 MB a block at 512 head columns) from L2 once per TILE: 128 tiles x
// 10 263 blocks x 2 MB = 2.7 TB through the CUs' L1s per 1024 queries, ~ 47 of the 64 B/clk a CU can pull: 95 of the 289 ms of a
// 21 M-doc Zipf step, with the matrix cores a quarter busy (profiles/r04: VERDICT r4 item 1).  The strip's traffic is amortised only
// by more queries per strip load, and a walk workgroup cannot hold more than 8 query slots of accumulators.
//
// So the dense part moves out: head_gemm_kernel reads every strip operand ONCE per pass for TW x 8 tiles (64 tiles = 512 queries at
// TW = 8) and writes the sums -- int32 fixed-point units, exactly what the walk's accumulators hold -- to a scratch array in HBM
// (32 KB per (tile, block): 16-bit, in units of 2^14 -- a sum below 2^30 fits, and the truncation is one more term of the proof's slack:
// round 6 halved the scratch's HBM round trip this way); the walk's epilogue adds them to its list sums (bp_walk_topk<.., HD = 2>).  The strip is no longer
// bound by a CU's L1, so the head can be wider: columns present in >= 1/8 of the documents, up to 1024 of them (a list of density p
// costs ~ 4 800 p^2 clocks per block and tile against ~ 30 per head column here), which halves what is left for the lists.
//
// Arithmetic: v_mfma_f32_16x16x32_f16, A = strip operand (16 documents x 32 head columns), B = the weights of TWO tiles (16 columns =
// 2 x 8 slots), fp32 accumulate; a sum is scaled back by a power of two and truncated.  The in-walk dense part (bp_walk.h dense_chunk)
// splits a weight in two fp16 numbers hi + lo and spends the 16 columns on one tile; here a weight is ONE fp16 number (round to
// nearest: 2^-11 relative) -- half the product, half the weight traffic, and the strip read once per pass instead of twice -- and the
// refine step's bound widens by that relative error (bp_refine.h: RefineArgs::quant = 2; 21 M docs x 1024 Zipf queries: still no query
// falls back).  bp_head_slack (bp_search.hip) bounds the absolute part of the error.
#pragma once
#include "bp_walk.h"
#include "bp_head_asm.h"

namespace vs {

struct HeadArgs {
    const __half* strip;          // [block][k-step][document / 16][64 lanes] x 8 halves (bp_strip_index)
    uint4* wt;                    // [tile - tile0][k-step][64 lanes] x 8 halves: the tiles' weights as MFMA B operands
    uint16_t* out;                // [tile - tile0][block][document / 16][slot 8][16 documents] uint16: the dense sums in units of 2^14 (kHeadOutShift) of the walk's fixed point
    const int2* tiles;            // (first query, queries) of every tile
    const int32_t* n_tiles_dev;   // tile count of the batch (device)
    int32_t tile0, tile_cnt;      // this pass: tiles [tile0, tile0 + tile_cnt)
    const int64_t* qptr;
    const int32_t* qcols;
    const float* qvals;
    const float* qscale;
    const uint16_t* hmap;
    int32_t n_head;
    int32_t rows;                 // documents per block (a multiple of 128)
    int64_t n_rows, n_blocks;
    float head_pre, head_mul;
};

__host__ __device__ inline size_t head_out_index(int64_t tile_rel, int64_t n_blocks, int64_t b, int rows, int d, int slot) {
    return (((size_t)tile_rel * (size_t)n_blocks + (size_t)b) * (size_t)(rows / 16) + (size_t)(d >> 4)) * 128 + (size_t)slot * 16 + (size_t)(d & 15);
}

// the weights of tiles [tile0, tile0 + tile_cnt) on the head columns, in B-operand order, TWO tiles to an operand: lane l of k-step j holds
// MFMA column n = l & 15 = (tile & 1) * 8 + slot, head columns 32 j + 8 (l >> 4) .. + 7, of the tile pair (tile - tile0) / 2.  One fp16
// number per weight.  One workgroup per tile; it zeroes its own lanes first (and the missing partner's, when the pass ends on an even tile).
template <int UNUSED>
__global__ __launch_bounds__(256) void head_weights_kernel(HeadArgs a) {
    const int n_tiles = a.n_tiles_dev[0];
    const int rel = (int)blockIdx.x, tile = a.tile0 + rel;
    if (rel >= a.tile_cnt || tile >= n_tiles) return;
    const int ks = bp_head_pad(a.n_head) / 32;
    const bool alone = !(rel & 1) && (rel + 1 >= a.tile_cnt || tile + 1 >= n_tiles);       // an even tile without its partner
    uint4* w4 = a.wt + (size_t)(rel >> 1) * ks * 64;
    for (int i = threadIdx.x; i < ks * 64; i += 256)
        if (alone || ((i >> 3) & 1) == (rel & 1)) w4[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    _Float16* w = reinterpret_cast<_Float16*>(w4);
    const int q0 = a.tiles[tile].x, nq = a.tiles[tile].y;
    const int64_t e0 = a.qptr[q0], e1 = a.qptr[q0 + nq];
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
        const uint32_t hx = a.hmap[a.qcols[e]];
        if (hx == 0xFFFFu) continue;
        int qs = 0;
        while (e >= a.qptr[q0 + qs + 1]) ++qs;
        const float wsc = a.qvals[e] * a.qscale[q0 + qs] * a.head_pre;           // powers of two: exact
        const size_t at = ((size_t)(hx >> 5) * 64 + (size_t)(((hx & 31u) >> 3) * 16)) * 8 + (size_t)(hx & 7u);
        w[at + (size_t)((rel & 1) * 8 + qs) * 8] = (_Float16)wsc;                 // (round to nearest: 2^-11 relative, bp_refine.h)
    }
}

// The product.  A workgroup = WD x WT waves; a wave = 64 documents (4 strip operands per k-step) x 16 tiles (8 weight operands): 32
// MFMAs per k-step on 12 KB of operands, 128 accumulator registers.  Work items = (block, run of WD x 64 documents, group of WT x 16
// tiles), the tile groups of one document run on consecutive items (the strip run is re-read from L2 / Infinity Cache).  Operands
// come straight from global memory in operand order (one coalesced 16-byte load per lane each), the next k-step's in flight.
// WIDE = 1: a wave takes 128 documents (8 strip operands per k-step), accumulators in AGPRs -- 256-thread workgroups, one wave per SIMD
template <int WD, int WT, int WIDE = 0>
__global__ __launch_bounds__(WD * WT * 64) void head_gemm_kernel(HeadArgs a) {
    constexpr int NT = 8, TW = 2 * NT;                                        // weight operands / tiles of a wave (tools/gen_head_asm.py)
    constexpr int WDOCS = WIDE ? 128 : 64;                                    // documents of a wave
    const int n_tiles = a.n_tiles_dev[0];
    const int nt = max(0, min(n_tiles - a.tile0, a.tile_cnt));
    if (nt <= 0) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wd = wv / WT, wt_ = wv % WT;
    const int ks = bp_head_pad(a.n_head) / 32, mbk = a.rows / 16;
    const int tgroups = (nt + WT * TW - 1) / (WT * TW);
    const int druns = a.rows / (WD * WDOCS);
    const int64_t items = a.n_blocks * druns * tgroups;
    const uint4* strip4 = reinterpret_cast<const uint4*>(a.strip);
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int tg = (int)(item % tgroups);
        const int64_t r = item / tgroups;
        const int dr = (int)(r % druns);
        const int64_t b = r / druns;
        const int d0 = (dr * WD + wd) * WDOCS;                                 // this wave's first document in the block
        const int t0 = (tg * WT + wt_) * TW;                                   // ... first tile (relative to tile0), an even one
        const int rows_b = (int)min((int64_t)a.rows, a.n_rows - b * a.rows);
        if (d0 >= rows_b || t0 >= nt) continue;                                // (no barrier in this kernel: waves are independent)
        // the whole item -- operand loads, 32 MFMAs per k-step, conversion and store of the sums -- is ONE generated asm statement
        // (tools/gen_head_asm.py says why): the strip operands from SGPR base + lane * 16 (+ 1 KB per document group), the weights from
        // SGPR base + a constant VGPR offset per tile
        const unsigned long long abase = (unsigned long long)(strip4 + ((size_t)b * ks * mbk + (size_t)(d0 >> 4)) * 64);
        const unsigned long long bbase = (unsigned long long)(a.wt + (size_t)(t0 >> 1) * ks * 64);
        const uint32_t l16 = (uint32_t)lane * 16u;
        uint32_t boff[NT];                                                      // tiles past the pass's last: clamped reads, nothing stored
#pragma unroll
        for (int t = 0; t < NT; ++t) boff[t] = (uint32_t)min(t, (nt - 1 - t0) >> 1) * (uint32_t)ks * 1024u + l16;
        const unsigned long long obase = (unsigned long long)(a.out + head_out_index(t0, a.n_blocks, b, a.rows, d0, 0));
        const unsigned long long ostride = (unsigned long long)a.n_blocks * (unsigned long long)mbk * 256ull;      // bytes between the tiles of a (block, document group)
        const uint32_t so = (uint32_t)(lane & 7) * 32u + (uint32_t)(lane >> 4) * 8u;                              // slot row + the lane's 4 documents (2 bytes each)
        if constexpr (WIDE != 0) head_item_asm_wide(abase, bbase, (uint32_t)mbk * 1024u, boff, l16, (uint32_t)ks, obase, ostride, so, (uint32_t)min(TW, nt - t0), a.head_mul);
        else head_item_asm(abase, bbase, (uint32_t)mbk * 1024u, boff, l16, (uint32_t)ks, obase, ostride, so, (uint32_t)min(TW, nt - t0), a.head_mul);
    }
}

// The product with BOTH operands through LDS (round 6; tools/gen_head_asm.py says how): a workgroup = 2 x 2 wide waves = 256 documents x
// 32 tiles, its four waves load a stage's 64 KB together -- 32 KB per k-step and CU instead of the 64 KB of four independent wide
// waves -- and keeps four k-steps of loads in flight (LDS-DMA into a ring of five k-step slots: all 160 KB of LDS, from address 0).
template <int UNUSED>
__global__ __launch_bounds__(256) void head_gemm_lds_kernel(HeadArgs a) {
    const int n_tiles = a.n_tiles_dev[0];
    const int nt = max(0, min(n_tiles - a.tile0, a.tile_cnt));
    if (nt <= 0) return;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int wd = wv >> 1, wt_ = wv & 1;
    const int ks = bp_head_pad(a.n_head) / 32, mbk = a.rows / 16;
    const int tgroups = (nt + 31) / 32, npairs = (nt + 1) / 2;
    const int druns = (a.rows + 255) / 256;
    const int64_t items = a.n_blocks * druns * tgroups;
    const uint4* strip4 = reinterpret_cast<const uint4*>(a.strip);
    const uint32_t l16 = (uint32_t)lane * 16u;
    // Work items = (run of 256 documents, tile group).  The tile groups of ONE run go to workgroups of the same XCD (blockIdx % 8) that
    // run side by side: the run's strip operands miss that XCD's L2 once and hit it for the other groups -- the product is bound by the
    // requests that leave the L2 (0.095 per clock and CU against 0.40 that hit; profiles/r06_head_lds.txt).
    const int per_xcd = (int)gridDim.x >> 3, sets = per_xcd / tgroups;
    const bool by_xcd = (gridDim.x & 7) == 0 && sets >= 1;
    const int xcd = (int)blockIdx.x & 7, yy = (int)blockIdx.x >> 3;
    if (by_xcd && yy >= sets * tgroups) return;
    const int64_t n_runs = a.n_blocks * druns;
    const int64_t first = by_xcd ? (int64_t)(xcd * sets + yy / tgroups) : (int64_t)blockIdx.x;
    const int64_t stride = by_xcd ? (int64_t)(8 * sets) : (int64_t)gridDim.x;
    const int64_t last = by_xcd ? n_runs : items;
    for (int64_t item = first; item < last; item += stride) {
        const int tg = by_xcd ? yy % tgroups : (int)(item % tgroups);
        const int64_t r = by_xcd ? item : item / tgroups;
        const int dr = (int)(r % druns);
        const int64_t b = r / druns;
        const int rows_b = (int)min((int64_t)a.rows, a.n_rows - b * a.rows);
        // what this wave loads: strip groups 16 dr + 4 wv + i (clamped into the block), weight operands 16 tg + 4 wv + i (clamped into the pass)
        uint32_t aoff[4], boff[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            aoff[i] = (uint32_t)min(dr * 16 + 4 * wv + i, mbk - 1) * 1024u + l16;
            boff[i] = (uint32_t)min(tg * 16 + 4 * wv + i, npairs - 1) * (uint32_t)ks * 1024u + l16;
        }
        const unsigned long long abase = (unsigned long long)(strip4 + (size_t)b * ks * mbk * 64);
        const unsigned long long bbase = (unsigned long long)a.wt;
        // what it multiplies and stores: documents d0 .., tiles t0 .. (relative to tile0)
        const int d0 = dr * 256 + wd * 128, t0 = tg * 32 + wt_ * 16;
        const int n_docs16 = max(0, min(8, (rows_b - d0 + 15) / 16));
        const int n_store = n_docs16 > 0 ? max(0, min(16, nt - t0)) : 0;
        const unsigned long long obase = (unsigned long long)(a.out + head_out_index(min(t0, nt - 1), a.n_blocks, b, a.rows, min(d0, a.rows - 16), 0));
        const unsigned long long ostride = (unsigned long long)a.n_blocks * (unsigned long long)mbk * 256ull;
        const uint32_t so = (uint32_t)(lane & 7) * 32u + (uint32_t)(lane >> 4) * 8u;
        head_item_asm_lds(abase, bbase, (uint32_t)mbk * 1024u, aoff, boff, (uint32_t)wv * 4096u, (uint32_t)wd * 8192u + l16, (uint32_t)wt_ * 8192u + l16, (uint32_t)ks,
                          obase, ostride, so, (uint32_t)n_store, (uint32_t)n_docs16, a.head_mul);
    }
}

}  // namespace vs
