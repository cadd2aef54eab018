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
// (64 KB per (tile, block)); the walk's epilogue adds them to its list sums (bp_walk_topk<.., HD = 2>).  The strip is no longer
// bound by a CU's L1, so the head can be wider: columns present in >= 1/8 of the documents, up to 1024 of them (a list of density p
// costs ~ 4 800 p^2 clocks per block and tile against ~ 30 per head column here), which halves what is left for the lists.
//
// Arithmetic as the in-walk dense part (bp_walk.h dense_chunk): v_mfma_f32_16x16x32_f16, A = strip operand (16 documents x 32 head
// columns), B = the tile's weights split in two fp16 numbers hi + lo (16 columns = 8 slots x (hi, lo)), fp32 accumulate; each of the
// two sums is scaled back by a power of two and truncated, then added.  bp_head_slack (bp_search.hip) bounds the error.
#pragma once
#include "bp_walk.h"

namespace vs {

struct HeadArgs {
    const __half* strip;          // [block][k-step][document / 16][64 lanes] x 8 halves (bp_strip_index)
    uint4* wt;                    // [tile - tile0][k-step][64 lanes] x 8 halves: the tiles' weights as MFMA B operands
    int32_t* out;                 // [tile - tile0][block][document / 16][slot 8][16 documents] int32
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

// the weights of tiles [tile0, tile0 + tile_cnt) on the head columns, in B-operand order: lane l of k-step j holds column n = l & 15
// (slot | 8 x lo), head columns 32 j + 8 (l >> 4) .. + 7.  One workgroup per tile; the array is zeroed first (same launch).
template <int UNUSED>
__global__ __launch_bounds__(256) void head_weights_kernel(HeadArgs a) {
    const int n_tiles = a.n_tiles_dev[0];
    const int tile = a.tile0 + (int)blockIdx.x;
    if ((int)blockIdx.x >= a.tile_cnt || tile >= n_tiles) return;
    const int ks = bp_head_pad(a.n_head) / 32;
    uint4* w4 = a.wt + (size_t)blockIdx.x * ks * 64;
    for (int i = threadIdx.x; i < ks * 64; i += 256) w4[i] = make_uint4(0u, 0u, 0u, 0u);
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
        const _Float16 hi = (_Float16)wsc;
        const _Float16 lo = (_Float16)(wsc - (float)hi);
        const size_t at = ((size_t)(hx >> 5) * 64 + (size_t)(((hx & 31u) >> 3) * 16)) * 8 + (size_t)(hx & 7u);
        w[at + (size_t)qs * 8] = hi;
        w[at + (size_t)(qs + 8) * 8] = lo;
    }
}

// The product.  A workgroup = WD x WT waves; a wave = 64 documents (4 strip operands per k-step) x 8 tiles (8 weight operands): 32
// MFMAs per k-step on 12 KB of operands, 128 accumulator registers.  Work items = (block, run of WD x 64 documents, group of WT x 8
// tiles), the tile groups of one document run on consecutive items (the strip run is re-read from L2 / Infinity Cache).  Operands
// come straight from global memory in operand order (one coalesced 16-byte load per lane each), the next k-step's in flight.
template <int WD, int WT>
__global__ __launch_bounds__(WD * WT * 64) void head_gemm_kernel(HeadArgs a) {
    using h8 = __attribute__((ext_vector_type(8))) _Float16;
    using f4 = __attribute__((ext_vector_type(4))) float;
    constexpr int MA = 4, NT = 8;
    const int n_tiles = a.n_tiles_dev[0];
    const int nt = max(0, min(n_tiles - a.tile0, a.tile_cnt));
    if (nt <= 0) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wd = wv / WT, wt_ = wv % WT;
    const int ks = bp_head_pad(a.n_head) / 32, mbk = a.rows / 16;
    const int tgroups = (nt + WT * NT - 1) / (WT * NT);
    const int druns = a.rows / (WD * 64);
    const int64_t items = a.n_blocks * druns * tgroups;
    const uint4* strip4 = reinterpret_cast<const uint4*>(a.strip);
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int tg = (int)(item % tgroups);
        const int64_t r = item / tgroups;
        const int dr = (int)(r % druns);
        const int64_t b = r / druns;
        const int d0 = (dr * WD + wd) * 64;                                    // this wave's first document in the block
        const int t0 = (tg * WT + wt_) * NT;                                   // ... first tile (relative to tile0)
        const int rows_b = (int)min((int64_t)a.rows, a.n_rows - b * a.rows);
        if (d0 >= rows_b || t0 >= nt) continue;                                // (no barrier in this kernel: waves are independent)
        const uint4* ap = strip4 + ((size_t)b * ks * mbk + (size_t)(d0 >> 4)) * 64 + lane;
        const uint4* bp = a.wt + (size_t)t0 * ks * 64 + lane;
        // tiles past the pass's last: their operand reads stay inside the array (clamped), their sums are not stored
        int tclamp[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) tclamp[t] = min(t, nt - 1 - t0);
        f4 c[MA][NT];
#pragma unroll
        for (int m = 0; m < MA; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) c[m][t] = f4{0.f, 0.f, 0.f, 0.f};
        uint4 an[MA], bn[NT];
#pragma unroll
        for (int m = 0; m < MA; ++m) an[m] = ap[(size_t)m * 64];
#pragma unroll
        for (int t = 0; t < NT; ++t) bn[t] = bp[(size_t)tclamp[t] * ks * 64];
#pragma unroll 1
        for (int j = 0; j < ks; ++j) {
            uint4 ac[MA], bc[NT];
#pragma unroll
            for (int m = 0; m < MA; ++m) ac[m] = an[m];
#pragma unroll
            for (int t = 0; t < NT; ++t) bc[t] = bn[t];
            if (j + 1 < ks) {
#pragma unroll
                for (int m = 0; m < MA; ++m) an[m] = ap[((size_t)(j + 1) * mbk + m) * 64];
#pragma unroll
                for (int t = 0; t < NT; ++t) bn[t] = bp[((size_t)tclamp[t] * ks + (j + 1)) * 64];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                h8 bf;
                __builtin_memcpy(&bf, &bc[t], 16);
#pragma unroll
                for (int m = 0; m < MA; ++m) {
                    h8 af;
                    __builtin_memcpy(&af, &ac[m], 16);
                    c[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, c[m][t], 0, 0, 0);
                }
            }
        }
        // C: lane holds rows 4 (lane >> 4) + i, column lane & 15 = slot + 8 (hi | lo): both parts truncated, added; the lanes of the hi
        // columns store 4 consecutive documents of their slot (16 bytes)
        const int col = lane & 15, rg = lane >> 4;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t0 + t >= nt) continue;                                        // (wave-uniform)
#pragma unroll
            for (int m = 0; m < MA; ++m) {
                int32_t v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int32_t mine = (int32_t)(c[m][t][i] * a.head_mul);
                    v[i] = mine + __shfl_xor(mine, 8, 64);                      // hi + lo of the same slot and document
                }
                if (col < 8) {
                    int32_t* o = a.out + head_out_index(t0 + t, a.n_blocks, b, a.rows, d0 + m * 16 + 4 * rg, col);
                    *reinterpret_cast<int4*>(o) = make_int4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
}

// The same product with its operands staged through LDS by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction, no VGPRs).
// The strip and the weight array are stored in operand order, so a 1 KB operand lands lane-linear and is read back conflict-free with
// one ds_read_b128 per lane.  A k-step's image = WD x 4 strip operands + WT x 8 weight operands (40 KB at 2 x 4 waves), NBUF images in
// a ring: the pieces of step j + NBUF - 1 are issued right after the barrier of step j (every wave has finished reading that image),
// a wave waits for its own pieces of step j with a counted vmcnt, the barrier is LDS-only (s_waitcnt lgkmcnt(0); s_barrier: a
// __syncthreads() would drain the DMA).  Against the register-staged kernel above: the strip operand is fetched once per WORKGROUP
// and k-step instead of once per wave (4 tile waves share it), the weights once instead of twice, and the loads run NBUF - 1 steps
// ahead without costing a register.
template <int WD, int WT, int NBUF>
__host__ __device__ constexpr size_t head_gemm_lds_bytes() { return (size_t)NBUF * (WD * 4 + WT * 8) * 1024; }

template <int WD, int WT, int NBUF>
__global__ __launch_bounds__(WD * WT * 64) void head_gemm_lds_kernel(HeadArgs a) {
    using h8 = __attribute__((ext_vector_type(8))) _Float16;
    using f4 = __attribute__((ext_vector_type(4))) float;
    constexpr int MA = 4, NT = 8, NW = WD * WT;
    constexpr int A_PIECES = WD * MA, PIECES = A_PIECES + WT * NT;
    static_assert(PIECES % NW == 0, "every wave loads the same number of pieces");
    constexpr int PER = PIECES / NW;
    extern __shared__ __attribute__((aligned(16))) char smem_h[];
    const int n_tiles = a.n_tiles_dev[0];
    const int nt = max(0, min(n_tiles - a.tile0, a.tile_cnt));
    if (nt <= 0) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wd = wv / WT, wt_ = wv % WT;
    const int ks = bp_head_pad(a.n_head) / 32, mbk = a.rows / 16;
    const int tgroups = (nt + WT * NT - 1) / (WT * NT);
    const int druns = a.rows / (WD * 64);
    const int64_t items = a.n_blocks * druns * tgroups;
    const uint4* strip4 = reinterpret_cast<const uint4*>(a.strip);
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int tg = (int)(item % tgroups);
        const int64_t r = item / tgroups;
        const int dr = (int)(r % druns);
        const int64_t b = r / druns;
        const int d0w = dr * WD * 64, t0w = tg * WT * NT;                      // the workgroup's first document / tile
        const int rows_b = (int)min((int64_t)a.rows, a.n_rows - b * a.rows);
        if (d0w >= rows_b) continue;                                           // (workgroup-uniform)
        // this wave's pieces of a k-step: piece p < A_PIECES = strip operand p of the workgroup's document run, else weight operand
        const uint4* src[PER];
        size_t step[PER];
        uint32_t dst[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int p = wv * PER + i;
            if (p < A_PIECES) {
                src[i] = strip4 + ((size_t)b * ks * mbk + (size_t)(d0w >> 4) + p) * 64 + lane;
                step[i] = (size_t)mbk * 64;
            } else {
                const int tt = min(t0w + (p - A_PIECES), nt - 1);               // (tiles past the pass's last: clamped reads, nothing stored)
                src[i] = a.wt + (size_t)tt * ks * 64 + lane;
                step[i] = 64;
            }
            dst[i] = (uint32_t)p * 1024u;
        }
        auto issue = [&](int j) {
            const uint32_t img = (uint32_t)(j % NBUF) * (uint32_t)(PIECES * 1024);
#pragma unroll
            for (int i = 0; i < PER; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)j * step[i]),
                                                 (__attribute__((address_space(3))) void*)(smem_h + img + dst[i]), 16, 0, 0);
        };
        lds_barrier();                                                         // the previous item's last image is read
#pragma unroll
        for (int j = 0; j < NBUF - 1; ++j)
            if (j < ks) issue(j);
        f4 c[MA][NT];
#pragma unroll
        for (int m = 0; m < MA; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) c[m][t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int j = 0; j < ks; ++j) {
            // this wave's pieces of step j have landed: younger than them are the pieces of steps j + 1 .. j + NBUF - 2 (where they exist)
            const int ahead = min(NBUF - 2, ks - 1 - j);
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PER) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();                                                     // ... and everyone's; the image of step j - 1 is free
            if (j + NBUF - 1 < ks) issue(j + NBUF - 1);
            const char* img = smem_h + (size_t)(j % NBUF) * (PIECES * 1024) + (size_t)lane * 16;
            uint4 ac[MA], bc[NT];
#pragma unroll
            for (int m = 0; m < MA; ++m) ac[m] = *reinterpret_cast<const uint4*>(img + (size_t)(wd * MA + m) * 1024);
#pragma unroll
            for (int t = 0; t < NT; ++t) bc[t] = *reinterpret_cast<const uint4*>(img + (size_t)(A_PIECES + wt_ * NT + t) * 1024);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                h8 bf;
                __builtin_memcpy(&bf, &bc[t], 16);
#pragma unroll
                for (int m = 0; m < MA; ++m) {
                    h8 af;
                    __builtin_memcpy(&af, &ac[m], 16);
                    c[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, c[m][t], 0, 0, 0);
                }
            }
        }
        const int d0 = d0w + wd * 64, t0 = t0w + wt_ * NT;
        const int col = lane & 15, rg = lane >> 4;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t0 + t >= nt || d0 >= rows_b) continue;                        // (wave-uniform)
#pragma unroll
            for (int m = 0; m < MA; ++m) {
                int32_t v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int32_t mine = (int32_t)(c[m][t][i] * a.head_mul);
                    v[i] = mine + __shfl_xor(mine, 8, 64);
                }
                if (col < 8) {
                    int32_t* o = a.out + head_out_index(t0 + t, a.n_blocks, b, a.rows, d0 + m * 16 + 4 * rg, col);
                    *reinterpret_cast<int4*>(o) = make_int4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
}

}  // namespace vs
