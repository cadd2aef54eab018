// csr_scan_mq.h -- CSR scoring pass for a TILE of QT sparse queries per index pass (Qt > 1).
//
// The HBM stream (column packets + values) is the same as in csr_scan.h; what changes is the query
// side.  A dense fp32 image of one query fills 118 KB of LDS, so Qt = 1.  Queries on this path are
// sparse (768 + lexical dims out of V = 29 523), so the tile is kept as
//     tab[c]  (uint32, c in [0, V]):  low QT bits = which queries of the tile have weight at column c,
//                                     high bits   = LDS byte address of column c's weights (0 = unused column)
//     qv[]    (fp32): the tile's non-zero weights, grouped by column, ordered by query slot
// One LDS gather per index non-zero answers "does any of the QT queries touch this column?".  A position
// hits with p ~ 0.026 x QT and most hits are single, so packet positions are paired (i, i + 4): the two
// masks of a pair form a 16-bit hit word whose lowest set bit is served by straight-line, exec-masked
// code (one weight fetch + one ds_add_f64 per pair, four independent chains per packet); the remaining
// bits go through short loops, one hit per lane per trip.  Accumulators are doubles in LDS (ds_add_f64,
// a few copies per row to keep lanes off the same address): an fp32 x fp32 product is exact in fp64 and
// the fp64 sum of <= 2^16 of them carries ~29 spare bits, so the fp32 score does not depend on the order
// of the adds -- reproducible, and equal to the oracle's fp64-accumulated score.
//
// Top-k: per (workgroup, query slot) candidate keys go to an L2-resident global buffer (appends
// become rare once the k-th-best threshold tightens); every kMqSuperRows rows the workgroup meets
// at a barrier and prunes over-full buffers with the LDS bitonic sort.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "csr_scan.h"
#include "dense_csr.h"

namespace vs {

constexpr int kMqCap = 1024;          // candidate slots per (workgroup, query slot)
constexpr int kMqSuperRows = 512;     // rows between prune checks; prune when count > kMqCap - kMqSuperRows
constexpr int kMaxKMq = kMqCap - kMqSuperRows;

// x_i for a lane-varying i in [0, 8): select tree on registers.  Scalars are passed by value on purpose:
// indexing an array with a lane-varying index makes the compiler place the array in scratch memory.
template <class T>
__device__ __forceinline__ T sel8(T x0, T x1, T x2, T x3, T x4, T x5, T x6, T x7, int i) {
    const bool b0 = i & 1, b1 = i & 2, b2 = i & 4;
    const T a0 = b0 ? x1 : x0, a1 = b0 ? x3 : x2, a2 = b0 ? x5 : x4, a3 = b0 ? x7 : x6;
    const T c0 = b1 ? a1 : a0, c1 = b1 ? a3 : a2;
    return b2 ? c1 : c0;
}

// LDS byte address of tab[c] for the two uint16 column ids of a packet dword: one SDWA shift each (the
// compiler's and/bfe + shift pair costs two VALU ops per index non-zero; tab sits at LDS offset 0)
__device__ __forceinline__ uint32_t tab_addr_lo(uint32_t cw) {
    uint32_t r;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(2u), "v"(cw));
    return r;
}
__device__ __forceinline__ uint32_t tab_addr_hi(uint32_t cw) {
    uint32_t r;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(2u), "v"(cw));
    return r;
}

// LDS access by byte address (the kernel's dynamic LDS block starts at LDS address 0 -- checked at kernel entry):
// lets table entries carry ready-to-use addresses, with no base add in front of the ds_read
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) float lds_f32_t;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4_t lds_f32x4_t;
__device__ __forceinline__ uint32_t lds_u32(uint32_t addr) { return *reinterpret_cast<lds_u32_t*>(addr); }
__device__ __forceinline__ float lds_f32(uint32_t addr) { return *reinterpret_cast<lds_f32_t*>(addr); }
__device__ __forceinline__ f32x4_t lds_f32x4(uint32_t addr) { return *reinterpret_cast<lds_f32x4_t*>(addr); }

constexpr uint32_t kMqSharedFlag = 0x80000000u;   // tab entry of a shared column (DN variant): flag | LDS address << 8, mask byte 0

template <class T>
__device__ __forceinline__ T sel8b(T x0, T x1, T x2, T x3, T x4, T x5, T x6, T x7, bool b0, bool b1, bool b2) {
    const T a0 = b0 ? x1 : x0, a1 = b0 ? x3 : x2, a2 = b0 ? x5 : x4, a3 = b0 ? x7 : x6;
    const T c0 = b1 ? a1 : a0, c1 = b1 ? a3 : a2;
    return b2 ? c1 : c0;
}

template <class T>
__device__ __forceinline__ T sel4b(T x0, T x1, T x2, T x3, bool b0, bool b1) {
    const T a0 = b0 ? x1 : x0, a1 = b0 ? x3 : x2;
    return b1 ? a1 : a0;
}

struct MqArgs {
    const uint32_t* pk_ptr;
    const uint4* cols;
    const void* vals;
    int64_t n_rows;
    int32_t n_cols;
    int32_t k;
    int32_t nchunk;
    int64_t rows_per_chunk;
    // sparse queries (CSR over the batch) and the tile plan
    const int64_t* qptr;      // [B + 1]
    const int32_t* qcols;     // [qnnz]
    const float* qvals;       // [qnnz]
    const int2* tiles;        // [n_tiles] (first query, count <= QT)
    int32_t n_tiles;
    int32_t vals_cap;         // LDS capacity for tile weights (entries)
    uint64_t* cand;           // [B, nchunk, k] output keys, sorted descending
    uint64_t* gcand;          // [grid, QT, kMqCap] scratch (candidate keys; the counters live in LDS)
    const uint64_t* upper;    // optional [B]: only keys < upper[b] take part ("search after": passes beyond the first when k > kMaxKMq)
};

template <int QT>
__host__ __device__ inline size_t mq_fixed_lds_bytes(int32_t n_cols, int rows_in_flight) {
    const size_t tab = (((size_t)n_cols + 1) * 4 + 15) & ~(size_t)15;
    return tab + (size_t)kMqCap * 8 + (size_t)rows_in_flight * QT * 8 + (size_t)QT * 8 + 64 * 4;
}

// DN = 1: "shared-column" variant for batches whose queries overlap heavily (skewed column popularity: the
// head columns are non-zero in most queries AND most documents).  Columns used by >= T of the tile's 8
// queries (T as low as the LDS slack allows, >= 3) are stored as zero-padded 8-float rows; a hit on such a
// column costs two ds_read_b128 + 8 multiply-adds into per-lane fp64 registers (flushed once per row)
// instead of up to 8 trips through the one-hit-at-a-time remainder loop.
template <int G, int VM, int QT, int U, int DN>
__global__ __launch_bounds__(kScanThreads) void csr_scan_topk_mq(MqArgs a) {
    static_assert(QT == 8 && G >= 8, "the packed hit word assumes 8 query slots per tile; lanes 0..7 of a row group finish them");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RPW = 64 / G;
    constexpr int RPI = kScanWaves * RPW;
    constexpr int SB = kMqSuperRows / RPI;
    const size_t tab_bytes = (((size_t)a.n_cols + 1) * 4 + 15) & ~(size_t)15;
    uint32_t* tab = reinterpret_cast<uint32_t*>(smem);
    uint64_t* sortbuf = reinterpret_cast<uint64_t*>(smem + tab_bytes);          // [kMqCap]
    constexpr int S = G >= 32 ? 4 : (G >= 16 ? 2 : 1);                          // accumulator copies per row
    double* acc = reinterpret_cast<double*>(sortbuf + kMqCap);                  // [RPI][S][QT]
    unsigned long long* tau = reinterpret_cast<unsigned long long*>(acc + RPI * S * QT);   // [QT]
    int* scratch = reinterpret_cast<int*>(tau + QT);                            // [48]
    unsigned int* ccnt = reinterpret_cast<unsigned int*>(scratch + 48);         // [QT] candidate counts (keys live in global memory)
    float* qv = reinterpret_cast<float*>(scratch + 64);                         // [vals_cap]

    const uint32_t qv_addr = (uint32_t)(reinterpret_cast<char*>(qv) - smem);    // LDS byte address of qv (smem starts at 0)
    if ((uint32_t)(size_t)((__attribute__((address_space(3))) char*)smem) != 0u) __builtin_trap();
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane / G, lg = lane % G;
    const int slot = w * RPW + g;
    const int K = a.k;
    uint64_t* my_gcand = a.gcand + (size_t)blockIdx.x * QT * kMqCap;
    const int seg = (a.n_cols + 1 + kScanThreads - 1) / kScanThreads;
    const int64_t items = (int64_t)a.n_tiles * a.nchunk;

    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int tile = (int)(item / a.nchunk), c = (int)(item % a.nchunk);
        const int q0 = a.tiles[tile].x, nq = a.tiles[tile].y;
        const int64_t r0 = (int64_t)c * a.rows_per_chunk;
        const int64_t r1 = min(a.n_rows, r0 + a.rows_per_chunk);
        const uint64_t my_upper = (a.upper && lg < nq) ? a.upper[q0 + lg] : ~0ull;
        __syncthreads();
        // ---- build the tile tables ----
        for (int i = tid; i <= a.n_cols; i += kScanThreads) tab[i] = 0;
        for (int i = tid; i < RPI * S * QT; i += kScanThreads) acc[i] = 0.0;
        if (tid < QT) { tau[tid] = 0ull; ccnt[tid] = 0u; }
        __syncthreads();
        const int64_t e0 = a.qptr[q0], e1 = a.qptr[q0 + nq];
        for (int64_t e = e0 + tid; e < e1; e += kScanThreads) {
            int qs = 0;
            while (e >= a.qptr[q0 + qs + 1]) ++qs;
            atomicOr(&tab[a.qcols[e]], 1u << qs);
        }
        __syncthreads();
        if constexpr (DN == 0) {   // offsets: exclusive scan of popc(mask) in column order (contiguous segment per thread)
            const int i0 = tid * seg, i1 = min(a.n_cols + 1, i0 + seg);
            int mine = 0;
            for (int i = i0; i < i1; ++i) mine += __popc(tab[i]);
            int off = block_excl_scan(mine, scratch, tid, nullptr);
            for (int i = i0; i < i1; ++i) {
                const uint32_t m = tab[i];
                // LDS byte address of the column's weights; unused columns stay 0 so that a lane without a hit
                // fetches LDS address 0 (one broadcast address: no bank conflicts, result unused)
                if (m) tab[i] = m | ((qv_addr + (uint32_t)(off * 4)) << QT);
                off += __popc(m);
            }
            __syncthreads();
            for (int64_t e = e0 + tid; e < e1; e += kScanThreads) {
                int qs = 0;
                while (e >= a.qptr[q0 + qs + 1]) ++qs;
                const uint32_t t = tab[a.qcols[e]];
                *reinterpret_cast<float*>(smem + (t >> QT) + 4 * __popc(t & ((1u << qs) - 1u))) = a.qvals[e];
            }
        } else {
            // shared columns (popc >= T) first, as 32-byte zero-padded rows; tab = 0x100 | byte offset << 8 (mask byte
            // 0: the one-hit paths never see them; bit 8 is free because offsets are multiples of 4); the rest packed
            int* hist = scratch + 56;                                  // [6]: columns with popc 3..8
            if (tid < 6) hist[tid] = 0;
            __syncthreads();
            const int i0 = tid * seg, i1 = min(a.n_cols + 1, i0 + seg);
            for (int i = i0; i < i1; ++i) {
                const int pc = __popc(tab[i]);
                if (pc >= 3) atomicAdd(&hist[pc - 3], 1);
            }
            __syncthreads();
            int T = 9;
            {
                const int slack = a.vals_cap - (int)(e1 - e0);
                int extra = 0;
                for (int pc = 8; pc >= 3; --pc) {
                    extra += hist[pc - 3] * (8 - pc);
                    if (extra > slack) break;
                    T = pc;
                }
            }
            int mine = 0;
            for (int i = i0; i < i1; ++i) {
                const int pc = __popc(tab[i]);
                mine += pc >= T ? (1 << 16) : pc;
            }
            int total = 0;
            const int off = block_excl_scan(mine, scratch, tid, &total);
            const int n_shared = total >> 16;
            int off_d = off >> 16, off_p = (off & 0xFFFF) + 8 * n_shared;
            for (int i = i0; i < i1; ++i) {
                const uint32_t m = tab[i];
                const int pc = __popc(m);
                if (pc >= T) {
                    tab[i] = m | kMqSharedFlag | ((qv_addr + (uint32_t)(off_d * 32)) << QT);   // mask kept for the fill below
                    ++off_d;
                } else {
                    if (m) tab[i] = m | ((qv_addr + (uint32_t)(off_p * 4)) << QT);
                    off_p += pc;
                }
            }
            for (int i = tid; i < 8 * n_shared; i += kScanThreads) qv[i] = 0.f;
            __syncthreads();
            for (int64_t e = e0 + tid; e < e1; e += kScanThreads) {
                int qs = 0;
                while (e >= a.qptr[q0 + qs + 1]) ++qs;
                const uint32_t t = tab[a.qcols[e]];
                const uint32_t addr = (t & ~kMqSharedFlag) >> QT;
                if (t & kMqSharedFlag) *reinterpret_cast<float*>(smem + addr + 4 * qs) = a.qvals[e];
                else *reinterpret_cast<float*>(smem + addr + 4 * __popc(t & ((1u << qs) - 1u))) = a.qvals[e];
            }
            __syncthreads();
            for (int i = i0; i < i1; ++i)
                if (tab[i] & kMqSharedFlag) tab[i] &= ~0xFFu;                   // hide shared columns from the one-hit paths
        }
        __syncthreads();

        // ---- scan ----
        const int64_t iters = (r1 - r0 + RPI - 1) / RPI;
        // row pointers of the NEXT row are fetched while the current row is processed (short rows -- BoT --
        // are otherwise a chain of dependent loads: pointer -> packets -> table)
        uint32_t np0 = 0, np1 = 0;
        if (r0 + slot < r1) { np0 = a.pk_ptr[r0 + slot]; np1 = a.pk_ptr[r0 + slot + 1]; }
        double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0, d4 = 0.0, d5 = 0.0, d6 = 0.0, d7 = 0.0;   // DN: per-lane sums over shared columns
        bool shared_seen = false;                                                                  // wave-uniform
        for (int64_t it0 = 0; it0 < iters || it0 == 0; it0 += SB) {
            const int64_t it1 = min(iters, it0 + SB);
            for (int64_t it = it0; it < it1; ++it) {
                const int64_t row = r0 + it * RPI + slot;
                double* myacc = acc + ((size_t)slot * S + (lg & (S - 1))) * QT;
                const uint32_t p0 = np0, p1 = np1;
                {
                    const int64_t nrow = row + RPI;
                    if (nrow < r1) { np0 = a.pk_ptr[nrow]; np1 = a.pk_ptr[nrow + 1]; }
                }
                if (row < r1) {
                    // U packets per lane per trip: all loads first, then the hit walks
                    const uint32_t padw = (uint32_t)a.n_cols | ((uint32_t)a.n_cols << 16);
                    for (uint32_t pb = p0 + lg; pb < p1; pb += U * G) {
                        uint4 cwu[U];
                        float vu[U][8];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const uint32_t p = pb + u * G;
                            const bool ok = p < p1;
                            const uint32_t pc = ok ? p : pb;
                            cwu[u] = a.cols[pc];
                            if (!ok) cwu[u] = make_uint4(padw, padw, padw, padw);
                            if constexpr (VM == VM_F32) {
                                const float4* vp = reinterpret_cast<const float4*>(a.vals);
                                const float4 v0 = vp[2 * (size_t)pc], v1 = vp[2 * (size_t)pc + 1];
                                vu[u][0] = v0.x; vu[u][1] = v0.y; vu[u][2] = v0.z; vu[u][3] = v0.w;
                                vu[u][4] = v1.x; vu[u][5] = v1.y; vu[u][6] = v1.z; vu[u][7] = v1.w;
                            } else if constexpr (VM == VM_F16) {
                                const uint4 hv = reinterpret_cast<const uint4*>(a.vals)[pc];
                                const __half2* h = reinterpret_cast<const __half2*>(&hv);
#pragma unroll
                                for (int i = 0; i < 4; ++i) { const float2 f = __half22float2(h[i]); vu[u][2 * i] = f.x; vu[u][2 * i + 1] = f.y; }
                            } else {
#pragma unroll
                                for (int i = 0; i < 8; ++i) vu[u][i] = 1.0f;
                            }
                        }
                        // Hits are sparse (a position hits with p ~ 0.026 x QT) and almost always single: per position
                        // a straight-line, exec-masked "first hit" and "second hit" (static registers, no select
                        // trees, independent chains); >= 3 queries sharing one column fall into a rare generic loop.
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const uint32_t cwv[4] = {cwu[u].x, cwu[u].y, cwu[u].z, cwu[u].w};
                            uint32_t t[8];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                t[2 * i] = lds_u32(tab_addr_lo(cwv[i]));
                                t[2 * i + 1] = lds_u32(tab_addr_hi(cwv[i]));
                            }
                            if constexpr (DN) {
                                const uint32_t anyd = (t[0] | t[1] | t[2] | t[3] | t[4] | t[5] | t[6] | t[7]) & kMqSharedFlag;
                                if (__builtin_amdgcn_ballot_w64(anyd != 0)) {
                                    shared_seen = true;
                                    uint32_t dm = 0;
#pragma unroll
                                    for (int i = 0; i < 8; ++i) dm |= (t[i] >> 31) << i;
                                    while (dm) {
                                        const int i = __ffs(dm) - 1;
                                        dm &= dm - 1;
                                        const uint32_t ts = sel8(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], i);
                                        const float vs_ = sel8(vu[u][0], vu[u][1], vu[u][2], vu[u][3], vu[u][4], vu[u][5], vu[u][6], vu[u][7], i);
                                        const uint32_t wa = (ts & ~kMqSharedFlag) >> QT;
                                        const f32x4_t w0 = lds_f32x4(wa), w1 = lds_f32x4(wa + 16);
                                        d0 += (double)(vs_ * w0.x); d1 += (double)(vs_ * w0.y); d2 += (double)(vs_ * w0.z); d3 += (double)(vs_ * w0.w);
                                        d4 += (double)(vs_ * w1.x); d5 += (double)(vs_ * w1.y); d6 += (double)(vs_ * w1.z); d7 += (double)(vs_ * w1.w);
                                    }
                                }
                            }
                            // First hits, two packet positions (i, i+4) per LDS round trip: a position hits with p ~ 0.2, so
                            // "exactly one of the pair" is the common case and is served by ONE weight fetch + ONE ds_add_f64.
                            // hw = mask(i) | mask(i+4) << 8 (one v_perm); its lowest set bit is the first hit (position i
                            // before i+4, lower query slot first); the other bits go to the remainder loop.
                            uint32_t hw[4], hb[4], rm[4];
                            float wv[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {                              // all four weight fetches first (independent)
                                hw[i] = __builtin_amdgcn_perm(t[i + 4], t[i], 0x0c0c0400u);
                                hb[i] = (uint32_t)__builtin_ctz(hw[i] | 0x10000u);     // 16 when there is no hit
                                const bool use_a = hb[i] < 8u;
                                uint32_t ts = use_a ? t[i] : t[i + 4];
                                if constexpr (DN) ts &= ~kMqSharedFlag;                // a shared column's entry carries the flag (mask 0: fetch unused)
                                const float vs_ = use_a ? vu[u][i] : vu[u][i + 4];
                                // lanes without a hit read a zero entry's address: LDS address 0, one broadcast, result unused
                                wv[i] = vs_ * lds_f32(ts >> QT);
                                rm[i] = hw[i] & (hw[i] - 1);
                            }
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (hw[i]) atomicAdd(&myacc[hb[i] & 7u], (double)wv[i]);               // ds_add_f64
                            // second hits of a pair and columns shared by >= 2 queries of the tile: one loop over the packed
                            // remainder words (16 bits per pair) -- few iterations, one hit per lane per iteration
                            // (two loops, one per pair of pairs: a 4-way select per hit instead of an 8-way one)
                            uint32_t mlo = rm[0] | (rm[1] << 16);
                            while (mlo) {
                                const int bit = __builtin_ctz(mlo);
                                mlo &= mlo - 1;
                                const bool b0 = bit & 16, b2 = bit & 8;                // position = (bit / 16) + 4 * ((bit / 8) & 1)
                                const int qs = bit & 7;
                                const uint32_t ts = sel4b(t[0], t[1], t[4], t[5], b0, b2);
                                const float vs_ = sel4b(vu[u][0], vu[u][1], vu[u][4], vu[u][5], b0, b2);
                                const uint32_t addr = (ts >> QT) + 4u * __popc(ts & ((1u << qs) - 1u));
                                atomicAdd(&myacc[qs], (double)(vs_ * lds_f32(addr)));
                            }
                            uint32_t mhi = rm[2] | (rm[3] << 16);
                            while (mhi) {
                                const int bit = __builtin_ctz(mhi);
                                mhi &= mhi - 1;
                                const bool b0 = bit & 16, b2 = bit & 8;                // position = 2 + (bit / 16) + 4 * ((bit / 8) & 1)
                                const int qs = bit & 7;
                                const uint32_t ts = sel4b(t[2], t[3], t[6], t[7], b0, b2);
                                const float vs_ = sel4b(vu[u][2], vu[u][3], vu[u][6], vu[u][7], b0, b2);
                                const uint32_t addr = (ts >> QT) + 4u * __popc(ts & ((1u << qs) - 1u));
                                atomicAdd(&myacc[qs], (double)(vs_ * lds_f32(addr)));
                            }
                        }
                    }
                }
                if constexpr (DN) {
                    if (shared_seen) {
                        shared_seen = false;
                        if (d0 != 0.0) { atomicAdd(&myacc[0], d0); d0 = 0.0; }
                        if (d1 != 0.0) { atomicAdd(&myacc[1], d1); d1 = 0.0; }
                        if (d2 != 0.0) { atomicAdd(&myacc[2], d2); d2 = 0.0; }
                        if (d3 != 0.0) { atomicAdd(&myacc[3], d3); d3 = 0.0; }
                        if (d4 != 0.0) { atomicAdd(&myacc[4], d4); d4 = 0.0; }
                        if (d5 != 0.0) { atomicAdd(&myacc[5], d5); d5 = 0.0; }
                        if (d6 != 0.0) { atomicAdd(&myacc[6], d6); d6 = 0.0; }
                        if (d7 != 0.0) { atomicAdd(&myacc[7], d7); d7 = 0.0; }
                    }
                }
                __builtin_amdgcn_wave_barrier();     // this wave's adds precede its reads (LDS executes a wave's ops in order)
                if (row < r1 && lg < nq) {
                    double sum = 0.0;
#pragma unroll
                    for (int c2 = 0; c2 < S; ++c2) {
                        double* pa = acc + ((size_t)slot * S + c2) * QT + lg;
                        sum += *pa;
                        *pa = 0.0;
                    }
                    const uint64_t key = make_key((float)sum, (uint32_t)row);
                    if (key > tau[lg] && key < my_upper) {
                        const uint32_t pos = atomicAdd(&ccnt[lg], 1u);
                        my_gcand[(size_t)lg * kMqCap + pos] = key;
                    }
                }
            }
            __syncthreads();
            const bool last = it1 >= iters;
            for (int qs = 0; qs < nq; ++qs) {
                const uint32_t cnt = ccnt[qs];
                if (last || cnt > (uint32_t)(kMqCap - kMqSuperRows)) {
                    for (int i = tid; i < kMqCap; i += kScanThreads) sortbuf[i] = (uint32_t)i < cnt ? my_gcand[(size_t)qs * kMqCap + i] : 0ull;
                    wg_sort_desc<kScanThreads>(sortbuf, kMqCap, tid);
                    if (last) {
                        uint64_t* out = a.cand + ((size_t)(q0 + qs) * a.nchunk + c) * (size_t)K;
                        for (int i = tid; i < K; i += kScanThreads) out[i] = sortbuf[i];
                    } else if (cnt > (uint32_t)K) {
                        for (int i = tid; i < K; i += kScanThreads) my_gcand[(size_t)qs * kMqCap + i] = sortbuf[i];
                        if (tid == 0) {
                            tau[qs] = sortbuf[K - 1];
                            ccnt[qs] = (uint32_t)K;
                        }
                    }
                    __syncthreads();
                }
            }
            __syncthreads();
            if (last) break;
        }
    }
}

// column use counts of a dense [B, V] query batch (one workgroup per query)
// colfreq[V] = sum over columns of f (f - 1): a query joining a column already used by f others adds 2 f
template <int UNUSED>
__global__ __launch_bounds__(kSpThreads) void mq_colfreq_kernel(const float* x, int64_t ld, int32_t B, int32_t V, uint32_t* colfreq) {
    unsigned long long* share = reinterpret_cast<unsigned long long*>(colfreq + ((V + 2) & ~1));
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        unsigned long long mine = 0;
        for (int i = threadIdx.x; i < V; i += kSpThreads)
            if (x[(size_t)b * ld + i] != 0.f) mine += 2ull * atomicAdd(&colfreq[i], 1u);
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
        if ((threadIdx.x & 63) == 0 && mine) atomicAdd(share, mine);
    }
}

// Single-thread planner: row pointers of the sparse query batch + greedy tiling (<= QT queries and
// <= vals_cap non-zeros per tile).  plan[0] = n_tiles, plan[1] = max nnz of one query, plan[2] = total nnz.
// plan[3] = sum over columns of f (f - 1), f = number of queries of the batch using the column: divided by
// B (B - 1) it is the expected number of columns two queries share (uniform 776-nnz queries: ~20; skewed
// column popularity: hundreds) -- the host picks the shared-column kernel variant from it.
template <int UNUSED>
__global__ void mq_plan_kernel(const int64_t* counts, int32_t B, int32_t qt, int32_t vals_cap, int64_t* qptr, int2* tiles, int64_t* plan,
                               const uint32_t* colfreq, int32_t n_cols) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    plan[3] = (int64_t)*reinterpret_cast<const unsigned long long*>(colfreq + ((n_cols + 2) & ~1));
    int64_t acc = 0, mx = 0;
    qptr[0] = 0;
    for (int b = 0; b < B; ++b) {
        acc += counts[b];
        qptr[b + 1] = acc;
        mx = counts[b] > mx ? counts[b] : mx;
    }
    int nt = 0, start = 0;
    while (start < B) {
        int cnt = 0;
        int64_t nz = 0;
        while (cnt < qt && start + cnt < B && nz + counts[start + cnt] <= vals_cap) { nz += counts[start + cnt]; ++cnt; }
        if (cnt == 0) cnt = 1;            // a query that does not fit: the host sees plan[1] > vals_cap and takes the dense path
        tiles[nt++] = make_int2(start, cnt);
        start += cnt;
    }
    plan[0] = nt;
    plan[1] = mx;
    plan[2] = acc;
}

}  // namespace vs
