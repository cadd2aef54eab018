// bp_duo.h -- the list walk (bp_walk.h) on TWO accumulator sets: no workgroup barrier between blocks, the epilogue done by all waves.
//
// Phase clocks of bp_walk_topk at 21 M docs (profiles/r03_phase_clocks.txt): of a block's 65 k cycles, 5.2 k pass at the block
// barrier and 4.5 k in the epilogue -- 15 % in which the LDS, the resource the walk is bound by, idles.  A first version (round 3's
// bp_pipe.h, replaced by this file) removed the barrier but walked flat worklists (a slower inner loop) and let ONE wave finish a
// block (24 k cycles); a second one had 4-query tiles (every block's records pass twice as often through the L2: 13 % slower per
// entry).  Here:
//
//  * the inner loop is bp_walk_topk's (a list per 8-lane group, NB lists in flight, second records in the same round);
//  * a tile has 8 queries like the list walk's, cut in two HALVES of 4 query slots, and the LDS of the 8-slot accumulators holds two
//    sets [2][2049][5]: step j = (block j / 2, half j & 1) adds into set j & 1 -- block b is walked for slots 0 .. 3, then for slots
//    4 .. 7, then block b + 1 ...;
//  * a wave that finds step j's chunk queue empty counts itself out (done[j]) and goes on to step j + 1.  Between two chunks of
//    step j it looks whether every wave has left step j - 1 (done[j - 1]) and then finishes ITS SHARE of that step -- 128
//    documents x 4 slots -- and counts that (epi[j - 1]); step j + 1 may be entered when epi[j - 1] is complete (the set is free
//    again), which by then it long is.  Nobody waits unless it is a step ahead of the slowest wave;
//  * candidate buffers (4096 per query slot, in global memory) are pruned -- workgroup sort, the only barrier -- when a slot could
//    overflow in the next epilogue: a share that sees its slot's count beyond the mark raises flag[j]; every wave reads the flag at
//    the same point -- entering step j + 2, after epi[j] is complete and before it can owe the epilogue of step j + 1 -- so all of
//    them take the same branch.  The sort buffer holds 2048 keys (the entry table needs the rest): two halves, then their winners.
//
// Counters are cumulative and live in a ring of 4 (no wave is two steps ahead of another); the LDS executes a wave's
// instructions in program order and everybody's in one order, so relaxed atomics between wavefront-scope fences (compiler ordering
// only) are enough -- a workgroup-scope acquire / release would also drain the global loads in flight.
// Valued records, fixed-point sums, no dense strips, no exclusive upper bounds.
//
// Measured (4 M docs, 1024 queries; the list walk: 29.7 ms): the boundary cost is gone -- 1.6 k cycles of waiting and 1.2 k of epilogue
// per step instead of 5.2 k + 4.5 k per block -- and the walk is SLOWER: 56.9 ms free running, 37.9 ms with a lock-step window of two
// blocks between the workgroups (BpArgs::pace, the default for this walk).  What the block barrier buys is the common pace: every
// workgroup of an XCD sweeps a block's lists in the same column order at the same time, and an XCD's L2 (4 MB) holds the window
// they share of the block's 6.4 MB of records; waves that run ahead into the next step, and workgroups that drift, widen that window
// until it no longer fits (the second half walks its columns in the opposite order to narrow it: 38.9 -> 37.9 ms).  DESIGN 8.1.
#pragma once
#include "bp_flat.h"

namespace vs {

constexpr int kDuoQT = 8;                        // queries of a tile
constexpr int kDuoHQ = 4;                        // query slots of an accumulator set (half a tile)
constexpr int kDuoEntCap = kBpEntCap;            // (query, column) entries of a tile
constexpr int kDuoCap = kFlCap;                  // candidate slots per query (global memory): K' kept + a block's documents
constexpr int kDuoSort = 2048;                   // keys the LDS sort buffer holds
constexpr int kDuoMaxK = kDuoSort / 2;           // the winners of the two halves are sorted together

template <int RMAX>
__host__ __device__ inline size_t bp_duo_lds_bytes(int ent_cap) {
    return 2 * bp_acc_bytes<kDuoHQ, AM_FIX, RMAX>() + (size_t)kDuoSort * 8 + 8 * 16 + 32 * 4 + (size_t)ent_cap * 8;
}

template <int VM, int NB, int RMAX>
__global__ __launch_bounds__(kScanThreads) void bp_duo_topk(BpArgs a) {
    static_assert(VM == VM_F16 || VM == VM_F32, "valued records");
    static_assert(RMAX == 2048 && kScanThreads == 1024, "a wave finishes 128 documents of a block");
    constexpr int QT = kDuoQT, HQ = kDuoHQ, LG = 8, PITCH = HQ + 1;
    constexpr uint32_t PITCHB = PITCH * 4u;
    constexpr uint32_t SETB = (uint32_t)bp_acc_bytes<HQ, AM_FIX, RMAX>();
    static_assert((size_t)2 * SETB >= kBpSortBytes, "the accumulator area holds the 8192-slot entry sort");
    static_assert(kDuoCap - RMAX <= kDuoSort && 2 * kDuoMaxK <= kDuoSort, "prune in two halves");
    constexpr int RS = bp_rec_bytes(VM);
    constexpr bool kWide = NB > 4;                // 8-lane groups, 5 .. 8 lists per slot: lane l owns list l's word (else: every quad holds all NB)
    static_assert(NB == 4 || (NB > 4 && NB <= 8), "lists per lane group");
    constexpr int GPW = 64 / LG, CW = GPW * NB, NW = kScanThreads / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* acc = reinterpret_cast<int32_t*>(smem);                                        // [2][RMAX + 1][PITCH]
    uint64_t* sortbuf = reinterpret_cast<uint64_t*>(smem + 2 * SETB);                       // [kDuoSort]
    unsigned long long* tau = reinterpret_cast<unsigned long long*>(sortbuf + kDuoSort);    // [8]
    int* sync = reinterpret_cast<int*>(tau + 16);                                       // chunk counters [0..3], done [4..7], epi [8..11], flag [12..15]
    unsigned int* ccnt = reinterpret_cast<unsigned int*>(sync + 16);                        // [16]
    uint2* ent = reinterpret_cast<uint2*>(ccnt + 16);                                       // [ent_cap]: x = column | slot byte offset << 16, y = weight bits

    const int tid = threadIdx.x, lane = tid & 63, wv_id = tid >> 6;
    const int gid = tid / LG, gl = tid % LG, gw = gid & (GPW - 1);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int K = a.k;
    uint64_t* my_gcand = a.gcand + (size_t)blockIdx.x * QT * kDuoCap;
    const int64_t n_blocks = (a.n_rows + a.rows - 1) / a.rows;
    const int64_t items = (int64_t)(a.n_tiles_dev ? a.n_tiles_dev[0] : a.n_tiles) * a.nchunk;
    bool pace_off = false;                      // the lock-step wait timed out once (pace_wait): this workgroup runs free from then on
    const size_t dir_ld = (size_t)a.n_cols + 1;
    const unsigned long long k_rt0 = a.timing ? __builtin_amdgcn_s_memrealtime() : 0ull;

    auto ld_acq = [&](const int* p) {
        const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        return v;
    };
    // (bounded: a wave that is never released -- a bug -- gives up after ~0.1 s and reports through BpArgs::debug instead of hanging the GPU)
    auto spin_ge = [&](const int* p, int target) {
        for (int it = 0; ld_acq(p) < target && it < (1 << 22); ++it) __builtin_amdgcn_s_sleep(1);
    };
    auto count_up = [&](int* p) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (lane == 0) __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int tile = (int)(item / a.nchunk), c = (int)(item % a.nchunk);
        const int q0 = a.tiles[tile].x, nq = a.tiles[tile].y;
#ifdef VS_DUO_TIMING            // phase clocks (developer build: the counters cost registers the walk has none to spare of)
        long long tm = a.timing ? (long long)__builtin_readcyclecounter() : 0;
        uint32_t tacc[6] = {0u, 0u, 0u, 0u, 0u, 0u};
        auto lap = [&](int phase) {
            if (a.timing) {
                const long long now = (long long)__builtin_readcyclecounter();
                tacc[phase] += (uint32_t)(now - tm);
                tm = now;
            }
        };
#else
        auto lap = [&](int) {};
#endif
        const int64_t b0 = (int64_t)c * a.blocks_per_chunk, b1 = min(n_blocks, b0 + a.blocks_per_chunk);
        const int nb = (int)max((int64_t)0, b1 - b0);
        __syncthreads();
        const int64_t e0 = a.qptr[q0], e1 = a.qptr[q0 + nq];
        const int n_ent = (int)(e1 - e0);
        const int n_ent0 = (int)(a.qptr[q0 + min(nq, HQ)] - e0);        // entries of the first half (query slots 0 .. 3)
        {   // entries sorted by half, length class, column (bp_walk.h); the accumulator area doubles as the sort buffer
            uint64_t* skey = reinterpret_cast<uint64_t*>(acc);
            for (int i = tid; i < 8192; i += kScanThreads) {
                uint64_t key = 0;
                if (i < n_ent) {
                    const int64_t e = e0 + i;
                    int qs = 0;
                    while (e >= a.qptr[q0 + qs + 1]) ++qs;
                    const float w = a.qvals[e] * a.qscale[q0 + qs];               // power of two: exact
                    const uint32_t col = (uint32_t)a.qcols[e];
                    uint32_t cls = 0;
                    if (a.df) {
                        const float per_block = (float)a.df[col] * (float)a.rows / (float)max(a.n_rows, (int64_t)1);
                        const float rounds = per_block * (1.f / (8.f * LG));
                        cls = rounds <= 1.f ? 0u : rounds <= 2.f ? 1u : rounds <= 4.f ? 2u : rounds <= 8.f ? 3u : 4u;
                    }
                    key = ((uint64_t)(qs < HQ ? 1 : 0) << 60) | (1ull << 59) | ((uint64_t)cls << 56) | ((uint64_t)col << 40) | ((uint64_t)qs << 32) |
                          (uint64_t)__float_as_uint(w);
                }
                skey[i] = key;
            }
            wg_sort_desc<kScanThreads>(skey, 8192, tid);
            // (the second half is walked in the OPPOSITE column order: the waves finishing the first half and the waves starting the second
            //  touch the same end of the block's records -- the workgroup's, and the XCD's, window over a block stays narrow)
            for (int i = tid; i < n_ent; i += kScanThreads) {
                const uint64_t key = skey[i < n_ent0 ? i : n_ent0 + (n_ent - 1 - i)];
                ent[i] = make_uint2(((uint32_t)(key >> 40) & 0xFFFFu) | ((uint32_t)((key >> 32) & 3u) * 4u) << 16, (uint32_t)key);
            }
            __syncthreads();
        }
        for (int i = tid; i < (int)(2 * SETB / 4); i += kScanThreads) acc[i] = 0;
        if (tid < 8) { tau[tid] = 0ull; ccnt[tid] = 0u; }
        if (tid < 16) sync[tid] = 0;
        __syncthreads();

        // a step = (block, half): its entries, chunks and grabs (every wave grabs once up front and once per chunk it walks)
        const int nsteps = 2 * nb;

        // directory words of a wave's first chunk of a step: fetched before the previous step's tail, they stay in flight across it
        uint32_t nd = 0;
        auto first_pairs = [&](int js, int li0) {
            const int ebh = (js & 1) ? n_ent0 : 0, neh = (js & 1) ? n_ent - n_ent0 : n_ent0;
            const uint32_t* dirn = a.dir + (size_t)(b0 + (js >> 1)) * dir_ld;
            const int lu = kWide ? gl : (gl & (NB - 1));
            const int off = li0 * CW + lu * GPW + gw;
            nd = 0;
            if (li0 >= 0 && lu < NB && off < neh) nd = dirn[ent[ebh + off].x & 0xFFFFu];
        };

        // ---- a wave's share of step js's epilogue: documents 128 w .. 128 w + 127, 4 slots -> order keys -> candidates -----------------------------
        auto epilogue_share = [&](const int js) -> bool {
            const int px = js & 1;
            const int64_t x = b0 + (js >> 1);
            const int rows_x = (int)min((int64_t)a.rows, a.n_rows - x * a.rows);
            int32_t* base = acc + px * (int)(SETB / 4);
            uint32_t thi[HQ];
#pragma unroll
            for (int s4 = 0; s4 < HQ; ++s4) thi[s4] = (uint32_t)(tau[px * HQ + s4] >> 32);
            bool full = false;
            int32_t sums[2][HQ];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int d = wv_id * 128 + h * 64 + lane;
#pragma unroll
                for (int s4 = 0; s4 < HQ; ++s4) sums[h][s4] = base[d * PITCH + s4];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int d = wv_id * 128 + h * 64 + lane;
#pragma unroll
                for (int s4 = 0; s4 < HQ; ++s4) base[d * PITCH + s4] = 0;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int d = wv_id * 128 + h * 64 + lane;
                const int64_t row = x * a.rows + d;
#pragma unroll
                for (int s4 = 0; s4 < HQ; ++s4) {
                    const int q = px * HQ + s4;
                    const uint32_t hi = (uint32_t)sums[h][s4] ^ 0x80000000u;
                    if (q < nq && d < rows_x && hi >= thi[s4]) {
                        const uint64_t key = ((uint64_t)hi << 32) | (uint32_t)(~(uint32_t)row);
                        if (key > tau[q]) {
                            const uint32_t pos = atomicAdd(&ccnt[q], 1u);
                            my_gcand[(size_t)q * kDuoCap + pos] = key;
                            full = full || pos + 1u > (uint32_t)(kDuoCap - RMAX);
                        }
                    }
                }
            }
            return full;
        };
        // candidate buffers beyond the mark (or all, at the end) -> keep the K best, raise the threshold.  Whole workgroup.  A buffer holds
        // up to 4096 keys, the sort buffer 2048: each half is sorted and its K best written back, then the 2 K winners together.
        auto sort_part = [&](const uint64_t* src, const uint32_t n) {            // sortbuf <- src[0 .. n) padded with zeros, sorted descending
            for (int i = tid; i < kDuoSort; i += kScanThreads) sortbuf[i] = (uint32_t)i < n ? src[i] : 0ull;
            wg_sort_desc<kScanThreads>(sortbuf, kDuoSort, tid);
        };
        auto prune = [&](const bool last) {
            for (int qs = 0; qs < nq; ++qs) {
                const uint32_t cnt = ccnt[qs];
                if (last || cnt > (uint32_t)(kDuoCap - RMAX)) {
                    uint64_t* buf = my_gcand + (size_t)qs * kDuoCap;
                    if (cnt > (uint32_t)kDuoSort) {
                        sort_part(buf + kDuoSort, cnt - (uint32_t)kDuoSort);          // the upper half first: its winners go where it began
                        __syncthreads();                                                 // (all of it is read before any of it is overwritten)
                        for (int i = tid; i < K; i += kScanThreads) buf[kDuoSort + i] = sortbuf[i];
                        __syncthreads();
                        sort_part(buf, (uint32_t)kDuoSort);
                        __syncthreads();
                        for (int i = tid; i < K; i += kScanThreads) buf[i] = sortbuf[i];
                        __syncthreads();
                        for (int i = tid; i < kDuoSort; i += kScanThreads)
                            sortbuf[i] = i < K ? buf[i] : (i < 2 * K ? buf[kDuoSort + (i - K)] : 0ull);
                        wg_sort_desc<kScanThreads>(sortbuf, kDuoSort, tid);
                    } else {
                        sort_part(buf, cnt);
                    }
                    if (last) {
                        uint64_t* out = a.cand + ((size_t)(q0 + qs) * a.nchunk + c) * (size_t)K;
                        for (int i = tid; i < K; i += kScanThreads) out[i] = sortbuf[i];
                    } else if (cnt > (uint32_t)K) {
                        for (int i = tid; i < K; i += kScanThreads) buf[i] = sortbuf[i];
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
        };

        if (nb > 0) first_pairs(0, wv_id);
        lap(0);
        for (int j = 0; j < nsteps; ++j) {
            const int64_t b = b0 + (j >> 1);
            const int p = j & 1, r4 = j & 3, gen = j >> 2;
            const int eb = p ? n_ent0 : 0, ne = p ? n_ent - n_ent0 : n_ent0, n_lc = (ne + CW - 1) / CW;
            // step j - 2 (same accumulator set) is finished by every wave; its overflow flag is final
            if (j >= 2) {
                const int r2 = (j - 2) & 3;
                spin_ge(&sync[8 + r2], NW * (((j - 2) >> 2) + 1));
                lap(2);
                if (ld_acq(&sync[12 + r2]) != 0) {              // uniform over the workgroup (see the header)
                    __syncthreads();
                    prune(false);
                    if (tid == 0) sync[12 + r2] = 0;
                    __syncthreads();
                    lap(4);
                }
            }
            // lock step with the chunk's other items (BpArgs::pace, bp_walk.h): without a block barrier the workgroups lose their common
            // pace, and with it the L2 copies the pack leaves behind
            if (a.pace && p == 0 && wv_id == 0 && items <= (int64_t)gridDim.x && !pace_off) {
                if (lane == 0) {
                    uint32_t* pc = a.pace + (size_t)c * a.blocks_per_chunk;
                    const int rel = j >> 1;
                    __hip_atomic_fetch_add(pc + rel, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (rel >= a.pace_window) {
                        const uint32_t need = (uint32_t)(items / a.nchunk);
                        if (!pace_wait(pc + rel - a.pace_window, need)) pace_off = true;
                    }
                }
            }
            const uint32_t* dirb = a.dir + (size_t)b * dir_ld;
            const unsigned long long pb = (unsigned long long)(a.rec + (size_t)a.base[b] * RS);
            const unsigned long long brec = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32)) << 32) |
                                            (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pb);
            const uint32_t accb = lds0 + (p ? SETB : 0u);         // LDS byte address of this block's accumulator set
            const int cbase = gen * (NW + n_lc);
            auto grab = [&]() {
                int v = 0;
                if (lane == 0) v = atomicAdd(&sync[r4], 1);
                return NW + (__builtin_amdgcn_readfirstlane(v) - cbase);
            };
            // this wave's share of block j - 1 is due as soon as every wave has left that block: looked at after every chunk of THIS
            // block, so that the set is free long before anybody wants it for block j + 1
            bool owe = j >= 1;
            const int r1 = (j - 1) & 3, done1 = NW * (((j - 1) >> 2) + 1);
            auto pay = [&]() {
                lap(1);
                const bool full = epilogue_share(j - 1);
                if (full) sync[12 + r1] = 1;                    // prune before the next epilogue
                count_up(&sync[8 + r1]);
                owe = false;
                lap(4);
            };
            int cur = wv_id, nxt = grab();
            while (cur < n_lc) {
                const int li = cur, li_n = nxt < n_lc ? nxt : -1;
                const uint32_t cd = nd;
                nd = 0;
                {
                    const int lu = kWide ? gl : (gl & (NB - 1));
                    const int off = li_n * CW + lu * GPW + gw;
                    if (li_n >= 0 && lu < NB && off < ne) nd = dirb[ent[eb + off].x & 0xFFFFu];
                }
                const int cb = li * CW;
                cur = nxt;
                nxt = grab();
                uint32_t rec[NB], end[NB];
                uint2 en[NB];
                bool more = false;
                uint32_t bd[NB];
                if constexpr (kWide) {
                    const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cd, 0x104, 0xF, 0xF, false);      // row_shl:4: lane i <- lane i + 4
                    const uint32_t dn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cd, 0x114, 0xF, 0xF, false);      // row_shr:4: lane i <- lane i - 4
                    const bool upper = (gl & 4) != 0;
                    const uint32_t oth = upper ? dn : up;
                    const uint32_t own4[4] = {quad_bcast<0>(cd), quad_bcast<1>(cd), quad_bcast<2>(cd), quad_bcast<3>(cd)};
                    const uint32_t oth4[4] = {quad_bcast<0>(oth), quad_bcast<1>(oth), quad_bcast<2>(oth), quad_bcast<3>(oth)};
#pragma unroll
                    for (int u = 0; u < NB; ++u) bd[u] = (u < 4) == upper ? oth4[u & 3] : own4[u & 3];
                } else {
                    bd[0] = quad_bcast<0>(cd); bd[1] = quad_bcast<1>(cd); bd[2] = quad_bcast<2>(cd); bd[3] = quad_bcast<3>(cd);
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const uint32_t lo = (bd[u] >> 12) << a.al_shift, hi = lo + (bd[u] & kBpDirRecMask);
                    rec[u] = lo + gl; end[u] = hi;
                    more = more || (rec[u] < end[u]);
                    en[u] = ent[min(eb + cb + u * GPW + gw, max(n_ent - 1, 0))];
                }
                auto add_record = [&](const u32x4& idv, const u32x4& vav, const u32x4& vbv, const uint2 e) {
                    const float wq = __uint_as_float(e.y);
                    const uint32_t so = (e.x >> 16) + accb;
                    const uint32_t dw[4] = {idv.x, idv.y, idv.z, idv.w};
                    float vv[8];
                    if constexpr (VM == VM_F32) {
                        vv[0] = wq * __uint_as_float(vav.x); vv[1] = wq * __uint_as_float(vav.y); vv[2] = wq * __uint_as_float(vav.z);
                        vv[3] = wq * __uint_as_float(vav.w); vv[4] = wq * __uint_as_float(vbv.x); vv[5] = wq * __uint_as_float(vbv.y);
                        vv[6] = wq * __uint_as_float(vbv.z); vv[7] = wq * __uint_as_float(vbv.w);
                    } else {
                        const uint32_t hw2[4] = {vav.x, vav.y, vav.z, vav.w};
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(vv[2 * t]) : "v"(wq), "v"(hw2[t]));
                            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(vv[2 * t + 1]) : "v"(wq), "v"(hw2[t]));
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const uint32_t off = (t & 1) ? acc_off_hi(dw[t >> 1], PITCHB, so) : acc_off_lo(dw[t >> 1], PITCHB, so);
                        lds_add(off, (int32_t)vv[t]);
                    }
                };
                while (__builtin_amdgcn_ballot_w64(more)) {
                    u32x4 ids[NB], va[NB], vb[NB];
                    u32x4 tid_r, tva_r, tvb_r = u32x4{0u, 0u, 0u, 0u};
                    uint32_t trec = 0, tend = 0;
                    uint2 ten = en[0];
                    int tu = -1;
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        const uint32_t off = __umul24(min(rec[u], end[u]), (uint32_t)RS);
                        if constexpr (VM == VM_F32) load_rec48(ids[u], va[u], vb[u], off, brec);
                        else load_rec32(ids[u], va[u], off, brec);
                        if (tu < 0 && rec[u] + LG < end[u]) { tu = u; trec = rec[u] + LG; tend = end[u]; ten = en[u]; }
                    }
                    {
                        const uint32_t off = __umul24(min(trec, tend), (uint32_t)RS);      // (no tail: record 0 of the block, not added)
                        if constexpr (VM == VM_F32) load_rec48(tid_r, tva_r, tvb_r, off, brec);
                        else load_rec32(tid_r, tva_r, off, brec);
                    }
                    constexpr int kPer = VM == VM_F32 ? 3 : 2;
                    more = false;
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        if constexpr (VM == VM_F32) wait_loads((NB - 1 - u) * 3 + kPer, ids[u], va[u], vb[u]);
                        else wait_loads((NB - 1 - u) * 2 + kPer, ids[u], va[u]);
                        if (rec[u] < end[u]) add_record(ids[u], va[u], vb[u], en[u]);
                        rec[u] += (tu == u) ? 2 * LG : LG;
                        more = more || (rec[u] < end[u]);
                    }
                    if constexpr (VM == VM_F32) wait_loads(0, tid_r, tva_r, tvb_r);
                    else wait_loads(0, tid_r, tva_r);
                    if (trec < tend) add_record(tid_r, tva_r, tvb_r, ten);
                }
                if (owe && ld_acq(&sync[4 + r1]) >= done1) pay();
            }
            if (j + 1 < nsteps) first_pairs(j + 1, wv_id);
            // thresholds other items of the same queries have published meanwhile
            if (wv_id == 0 && a.gtau && lane < nq) { const unsigned long long g = a.gtau[q0 + lane]; if (g > tau[lane]) tau[lane] = g; }
            lap(1);
            count_up(&sync[4 + r4]);                          // this wave has left block j (its adds are queued in front of the count)
            if (owe) {                                          // (a block of few chunks: nobody had left block j - 1 yet)
                spin_ge(&sync[4 + r1], done1);
                lap(2);
                pay();
            }
#ifdef VS_DUO_TIMING
            tacc[5] += 1u;
#endif
        }
        // the item's last step: every wave has left it behind this barrier; the overflow flag of the step before it first
        __syncthreads();
        lap(2);
        if (nsteps >= 2 && sync[12 + ((nsteps - 2) & 3)] != 0) {
            __syncthreads();
            prune(false);
        }
        if (nsteps > 0) (void)epilogue_share(nsteps - 1);
        __syncthreads();
        prune(true);
        lap(4);
#ifdef VS_DUO_TIMING
        if (a.timing && lane == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) atomicAdd(a.timing + i, (unsigned long long)tacc[i]);
        }
#endif
    }
    if (a.timing && threadIdx.x == 0) {        // per workgroup: 100 MHz ticks, absolute start, where it ran (XCC_ID, HW_ID)
        a.timing[16 + 4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime() - k_rt0;
        a.timing[17 + 4 * blockIdx.x] = k_rt0;
        a.timing[18 + 4 * blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20);
        a.timing[19 + 4 * blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}

}  // namespace vs
