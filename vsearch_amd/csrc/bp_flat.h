// bp_flat.h -- the FLAT walk over the blocked postings (bp_walk.h holds the layout, the builder and the list-per-lane-group walk).
//
// bp_walk_topk gives a posting list to a group of 8 lanes: a 50-posting list fills 7 of the group's 8 record slots, lists of
// different lengths leave lanes idle until the longest of the round is done, and a third of the kernel's VALU work is list
// bookkeeping (rocprofv3, 21 M docs: 31 % of the ds_add lane slots carry no posting, 7.5 VALU instructions per LDS atomic
// where the posting's arithmetic needs 3).  Here the lists of a chunk of entries are EXPANDED first: a lane owns one entry,
// reads its directory word, a wave prefix sum of the record counts places every list in the wave's worklist (LDS), and each
// lane writes one 32-bit item per record of its list -- (entry << 19 | record).  The add loop then knows nothing about lists:
// round r of a batch is item r * 64 + lane, whatever list it belongs to -- one item read, one entry read (weight, slot plane),
// one 32-byte record load, 8 x (multiply-convert, truncate, address, ds_add_u32).  Every lane of every round carries a record;
// what is left over after the last full round of a chunk (< 64 items) moves to the front and joins the next chunk's items, so a
// wave issues ONE partial round per block.  Lists of any length (skewed vocabularies) take as many batches as they need: a lane
// keeps (next record, records left) and writes what fits.
//
// Accumulators are SLOT-MAJOR here, acc[slot][document]: the address of a posting is slot plane + 4 * document (the same
// v_mad_u32_u16 on the packed id word), a list's adds spread over all banks without a padded pitch, and the epilogue reads
// TWO documents' sums per ds_read_b64, conflict-free -- a thread finishes documents 2 t and 2 t + 1 in one round.
// Pad postings (a list's last record) carry value 0 and a spread-out document id (bp_fill_kernel): they add nothing, and do not
// pile up on one bank.
//
// Valued records without dense head strips; the fixed-point filter only (AM_FIX).  Numerics, candidate keys, thresholds and
// output are bp_walk_topk's: the two kernels return the same candidate sets.
#pragma once
#include "bp_walk.h"

namespace vs {

constexpr int kFlCap = 4096;          // candidate slots per (workgroup, query slot): K' kept + 2048 new per epilogue round
constexpr int kFlRecBits = 19;        // worklist item = entry << 19 | record of the block: a block holds < 2^19 records (checked at build)
constexpr uint32_t kFlRecMask = (1u << kFlRecBits) - 1u;

template <int NR>
__host__ __device__ constexpr size_t bp_flat_work_bytes() {
    const size_t w = (size_t)kScanWaves * (64 * NR + 64) * 4, s = (size_t)kFlCap * 8;      // worklists; the candidate sort borrows the area
    return w > s ? w : s;
}
template <int NR, int RMAX>
__host__ __device__ inline size_t bp_flat_lds_bytes(int ent_cap) {
    return (size_t)8 * RMAX * 4 + bp_flat_work_bytes<NR>() + 8 * 16 + 64 * 4 + 64 + (size_t)ent_cap * 8;
}

// inclusive prefix sum over the 64 lanes of a wave: four row shifts, two row broadcasts (DPP; no LDS)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    uint32_t x = v;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);      // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);      // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);      // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);      // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);      // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);      // row_bcast:31 -> rows 2, 3
    return x;
}

// NR = record loads in flight per lane (rounds of a batch), RMAX = block capacity in documents (slot plane pitch)
template <int VM, int NR, int RMAX>
__global__ __launch_bounds__(kScanThreads) void bp_flat_topk(BpArgs a) {
    static_assert(VM == VM_F16 || VM == VM_F32, "valued records");
    static_assert((size_t)8 * RMAX * 4 >= kBpSortBytes, "the accumulator area holds the 8192-slot entry sort");
    static_assert(RMAX == 2 * kScanThreads, "a thread finishes documents 2 t and 2 t + 1");
    constexpr int QT = 8;
    constexpr int RS = bp_rec_bytes(VM);
    constexpr int kWork = 64 * NR + 64;                  // items of a wave's worklist: < 64 carried + NR full rounds
    constexpr uint32_t PLANE = (uint32_t)RMAX * 4u;     // bytes of a slot plane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* acc = reinterpret_cast<int32_t*>(smem);                                        // [QT][RMAX]
    uint32_t* work = reinterpret_cast<uint32_t*>(smem + (size_t)QT * PLANE);                // [waves][kWork]
    uint64_t* sortbuf = reinterpret_cast<uint64_t*>(work);                                  // [kFlCap] (between blocks: the worklists are empty)
    unsigned long long* tau = reinterpret_cast<unsigned long long*>(smem + (size_t)QT * PLANE + bp_flat_work_bytes<NR>());    // [QT]
    unsigned long long* upper_sh = tau + QT;                                                // [QT]
    int* scratch = reinterpret_cast<int*>(upper_sh + QT);                                   // [64]
    unsigned int* ccnt = reinterpret_cast<unsigned int*>(scratch + 64);                     // [QT] (+ 8 spare)
    uint2* ent = reinterpret_cast<uint2*>(ccnt + 16);                                       // [ent_cap]: x = column | slot plane offset << 16, y = weight bits

    const int tid = threadIdx.x, lane = tid & 63, wv_id = tid >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    uint32_t* workw = work + wv_id * kWork;
    const int K = a.k;
    uint64_t* my_gcand = a.gcand + (size_t)blockIdx.x * QT * kFlCap;
    const int64_t n_blocks = (a.n_rows + a.rows - 1) / a.rows;
    const int64_t items = (int64_t)(a.n_tiles_dev ? a.n_tiles_dev[0] : a.n_tiles) * a.nchunk;
    bool pace_off = false;                      // the lock-step wait timed out once (pace_wait): this workgroup runs free from then on
    const size_t dir_ld = (size_t)a.n_cols + 1;
    constexpr int NW = kScanThreads / 64;
    const unsigned long long k_rt0 = a.timing ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const long long k_c0 = a.timing ? (long long)__builtin_readcyclecounter() : 0;

    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int tile = (int)(item / a.nchunk), c = (int)(item % a.nchunk);
        const int q0 = a.tiles[tile].x, nq = a.tiles[tile].y;
        long long tm = a.timing ? (long long)__builtin_readcyclecounter() : 0;
        const unsigned long long rt0 = a.timing ? __builtin_amdgcn_s_memrealtime() : 0ull;      // 100 MHz: tells the shader clock of the run
        uint32_t tacc[6] = {0u, 0u, 0u, 0u, 0u, 0u};
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
        const int n_ent = (int)(e1 - e0);
        // entries sorted by column (then slot): the items of a wave sweep the block's records forward, and the entries of two
        // queries on one column follow each other (the second walk of the list hits the L1)
        {
            uint64_t* skey = reinterpret_cast<uint64_t*>(acc);
            for (int i = tid; i < 8192; i += kScanThreads) {
                uint64_t key = 0;
                if (i < n_ent) {
                    const int64_t e = e0 + i;
                    int qs = 0;
                    while (e >= a.qptr[q0 + qs + 1]) ++qs;
                    const float w = a.qvals[e] * a.qscale[q0 + qs];               // power of two: exact
                    // (the column leads, complemented: the sort is descending, the walk ascending)
                    key = ((uint64_t)(0xFFFFu - (uint32_t)a.qcols[e]) << 40) | ((uint64_t)qs << 32) | (uint64_t)__float_as_uint(w);
                    key |= 1ull << 63;                                            // a real entry never sorts as 0
                }
                skey[i] = key;
            }
            wg_sort_desc<kScanThreads>(skey, 8192, tid);
            for (int i = tid; i < n_ent; i += kScanThreads) {
                const uint64_t key = skey[i];
                const uint32_t col = 0xFFFFu - ((uint32_t)(key >> 40) & 0xFFFFu);
                ent[i] = make_uint2(col | (((uint32_t)(key >> 32) & 0xFFu) * PLANE) << 16, (uint32_t)key);
            }
            __syncthreads();
        }
        for (int i = tid; i < QT * RMAX; i += kScanThreads) acc[i] = 0;
        if (tid < QT) { tau[tid] = 0ull; ccnt[tid] = 0u; }
        if (tid < QT) upper_sh[tid] = (a.upper && tid < nq) ? a.upper[q0 + tid] : ~0ull;
        if (tid < 2) scratch[40 + tid] = 0;                   // chunk counters of even / odd blocks
        __syncthreads();

        // A chunk = CE consecutive entries, one per lane (a small tile takes fewer per chunk so that every wave gets some);
        // a wave's first chunk of a block is its own number, the following ones come from a counter in LDS.
        const int CE = min(64, max(8, (n_ent + 2 * NW - 1) / (2 * NW)));
        const int n_ch = (n_ent + CE - 1) / CE;
        int* chunk_cnt = scratch + 40;
        auto grab = [&](int par) {
            int v = 0;
            if (lane == 0) v = atomicAdd(&chunk_cnt[par], 1);
            return NW + __builtin_amdgcn_readfirstlane(v);
        };
        // the directory word of this lane's entry in chunk ch of the block whose directory row is dirp (lanes without an entry
        // read the word of the pad column n_cols: an empty list)
        auto dir_word = [&](const uint32_t* dirp, int ch) -> uint32_t {
            const int e = ch * CE + lane;
            const uint32_t col = (ch < n_ch && lane < CE && e < n_ent) ? (ent[e].x & 0xFFFFu) : (uint32_t)a.n_cols;
            return dirp[col];
        };
        uint32_t nd = 0;
        if (b0 < b1) nd = dir_word(a.dir + (size_t)b0 * dir_ld, wv_id);
        for (int64_t b = b0; b < b1 || b == b0; ++b) {
            const bool have_b = b < b1;
            const int rows_b = have_b ? (int)min((int64_t)a.rows, a.n_rows - b * a.rows) : 0;
            if (have_b) {
                const uint32_t* dirb = a.dir + (size_t)b * dir_ld;
                const unsigned long long pb = (unsigned long long)(a.rec + (size_t)a.base[b] * RS);
                const unsigned long long brec = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32)) << 32) |
                                                (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pb);

                // the 8 postings of one record into slot plane `so` (LDS byte address) with weight wq
                auto add_record = [&](const u32x4& idv, const u32x4& vav, const u32x4& vbv, const float wq, const uint32_t so) {
                    const uint32_t dw[4] = {idv.x, idv.y, idv.z, idv.w};
                    float vv[8];
                    if constexpr (VM == VM_F32) {
                        vv[0] = wq * __uint_as_float(vav.x); vv[1] = wq * __uint_as_float(vav.y); vv[2] = wq * __uint_as_float(vav.z);
                        vv[3] = wq * __uint_as_float(vav.w); vv[4] = wq * __uint_as_float(vbv.x); vv[5] = wq * __uint_as_float(vbv.y);
                        vv[6] = wq * __uint_as_float(vbv.z); vv[7] = wq * __uint_as_float(vbv.w);
                    } else {
                        // weight x fp16 value in ONE instruction (v_fma_mix_f32 converts the selected half on the way in; + 0: the
                        // product is rounded once, exactly as convert-then-multiply)
                        const uint32_t hw2[4] = {vav.x, vav.y, vav.z, vav.w};
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(vv[2 * t]) : "v"(wq), "v"(hw2[t]));
                            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(vv[2 * t + 1]) : "v"(wq), "v"(hw2[t]));
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const uint32_t off = (t & 1) ? acc_off_hi(dw[t >> 1], 4u, so) : acc_off_lo(dw[t >> 1], 4u, so);
                        lds_add(off, (int32_t)vv[t]);
                    }
                };
                // One batch: items [0, have) of the wave's worklist, NR rounds of 64; the loads of all rounds are issued back to
                // back (asm: the compiler would sink them into the predicated adds), each round waits for its own.  Rounds past
                // `have` load record 0 of the block and add nothing.
                auto consume = [&](const int have) {
                    uint32_t it[NR];
#pragma unroll
                    for (int r = 0; r < NR; ++r) it[r] = workw[r * 64 + lane];
                    uint2 en[NR];
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        it[r] = (r * 64 + lane < have) ? it[r] : 0u;
                        en[r] = ent[it[r] >> kFlRecBits];
                    }
                    u32x4 ids[NR], va[NR];
                    [[maybe_unused]] u32x4 vb[NR];
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const uint32_t off = __umul24(it[r] & kFlRecMask, (uint32_t)RS);
                        if constexpr (VM == VM_F32) load_rec48(ids[r], va[r], vb[r], off, brec);
                        else load_rec32(ids[r], va[r], off, brec);
                    }
                    constexpr int kPer = VM == VM_F32 ? 3 : 2;
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        if constexpr (VM == VM_F32) wait_loads((NR - 1 - r) * kPer, ids[r], va[r], vb[r]);
                        else wait_loads((NR - 1 - r) * kPer, ids[r], va[r]);
                        if (r * 64 + lane < have) {
                            if constexpr (VM == VM_F32) add_record(ids[r], va[r], vb[r], __uint_as_float(en[r].y), (en[r].x >> 16) + lds0);
                            else add_record(ids[r], va[r], va[r], __uint_as_float(en[r].y), (en[r].x >> 16) + lds0);
                        }
                    }
                };

                const int par = (int)(b & 1);
                int cur = wv_id, nxt = grab(par);
                int carried = 0;                                // items at the front of the worklist (< 64), wave-uniform
                while (cur < n_ch) {
                    const uint32_t cd = nd;
                    nd = dir_word(dirb, nxt);                   // the next chunk's words fly while this one is walked
                    uint32_t first = (cd >> 12) << a.al_shift;  // next record of this lane's list (relative to the block)
                    uint32_t rem = cd & kBpDirRecMask;          // records left
                    const uint32_t etag = (uint32_t)(cur * CE + lane) << kFlRecBits;
                    cur = nxt;
                    nxt = grab(par);
                    do {
                        const uint32_t incl = wave_incl_scan(rem);
                        const int total = (int)__builtin_amdgcn_readlane((int)incl, 63);
                        const int space = kWork - 1 - carried;              // so that carried + new <= 64 NR + 63: at most NR full rounds
                        const int excl = (int)(incl - rem);
                        const int take = min((int)rem, max(space - excl, 0));
                        uint32_t* wp = workw + carried + excl;
                        const uint32_t itv = etag | first;
                        for (int j = 0; __builtin_amdgcn_ballot_w64(j < take) != 0ull; ++j)
                            if (j < take) wp[j] = itv + (uint32_t)j;
                        first += (uint32_t)take;
                        rem -= (uint32_t)take;
                        const int have = carried + min(total, space);
                        const int full = have >> 6;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        if (full > 0) {
                            consume(full * 64);
                            const int left = have & 63;
                            if (left > 0) {                                 // the incomplete round joins the next batch
                                const uint32_t v = workw[full * 64 + lane];
                                __builtin_amdgcn_wave_barrier();
                                if (lane < left) workw[lane] = v;
                            }
                            carried = left;
                        } else {
                            carried = have;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    } while (__builtin_amdgcn_ballot_w64(rem != 0u) != 0ull);
                }
                if (carried > 0) consume(carried);              // the block's one partial round
                if (b + 1 < b1) nd = dir_word(a.dir + (size_t)(b + 1) * dir_ld, wv_id);
            }
            // thresholds other items of the same queries have published meanwhile (read before the barrier: the latency hides in it)
            if (a.gtau && tid < nq) { const unsigned long long g = a.gtau[q0 + tid]; if (g > tau[tid]) tau[tid] = g; }
            lap(1);
            // Lock step: every tile of this chunk sweeps the same blocks; held within `pace_window` blocks of the slowest, a block's
            // records are fetched from HBM once for all of them (L2 / Infinity Cache).  Only when all items are resident (one per
            // workgroup): a waiting item would otherwise wait for one that has not started.
            if (a.pace && items <= (int64_t)gridDim.x && tid == 0 && have_b && !pace_off) {
                uint32_t* pc = a.pace + (size_t)c * a.blocks_per_chunk;
                const int64_t rel = b - b0;
                __hip_atomic_fetch_add(pc + rel, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (rel >= a.pace_window) {
                    const uint32_t need = (uint32_t)(items / a.nchunk);
                    if (!pace_wait(pc + rel - a.pace_window, need)) pace_off = true;
                }
            }
            __syncthreads();
            if (tid == 0) scratch[40 + (int)((b + 1) & 1)] = 0;     // the next block's chunk counter (its last user was block b - 1)
            lap(2);
            // epilogue: a thread finishes documents 2 t and 2 t + 1: their QT sums (one ds_read_b64 per slot) -> order keys -> candidates
            {
                const int d = 2 * tid;
                uint32_t thi[QT];
#pragma unroll
                for (int q = 0; q < QT; ++q) thi[q] = (uint32_t)(tau[q] >> 32);
                if (d < rows_b) {
                    const int64_t row = b * a.rows + d;
                    uint2 sums[QT];
#pragma unroll
                    for (int q = 0; q < QT; ++q) sums[q] = *reinterpret_cast<const uint2*>(acc + q * RMAX + d);
#pragma unroll
                    for (int q = 0; q < QT; ++q) *reinterpret_cast<uint2*>(acc + q * RMAX + d) = make_uint2(0u, 0u);
#pragma unroll
                    for (int q = 0; q < QT; ++q) {                        // (slots >= nq are never written: a ragged tile skips them)
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
                __syncthreads();
                const bool last = b + 1 >= b1;
                uint32_t cnts[QT];
#pragma unroll
                for (int q = 0; q < QT; ++q) cnts[q] = ccnt[q];
                bool any = last;
#pragma unroll
                for (int q = 0; q < QT; ++q) any = any || cnts[q] > (uint32_t)(kFlCap - RMAX);
                if (any)
                for (int qs = 0; qs < nq; ++qs) {
                    const uint32_t cnt = ccnt[qs];
                    if (last || cnt > (uint32_t)(kFlCap - RMAX)) {
                        for (int i = tid; i < kFlCap; i += kScanThreads) sortbuf[i] = (uint32_t)i < cnt ? my_gcand[(size_t)qs * kFlCap + i] : 0ull;
                        wg_sort_desc<kScanThreads>(sortbuf, kFlCap, tid);
                        if (last) {
                            uint64_t* out = a.cand + ((size_t)(q0 + qs) * a.nchunk + c) * (size_t)K;
                            for (int i = tid; i < K; i += kScanThreads) out[i] = sortbuf[i];
                        } else if (cnt > (uint32_t)K) {
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
            tacc[5] += 1u;
            if (b + 1 >= b1) break;
        }
        if (a.timing) tacc[3] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - rt0);
        if (a.timing && lane == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) atomicAdd(a.timing + i, (unsigned long long)tacc[i]);
        }
    }
    if (a.timing && tid == 0) {        // per workgroup: 100 MHz ticks, shader cycles, where it ran (XCC_ID, HW_ID)
        a.timing[16 + 4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime() - k_rt0;
        a.timing[17 + 4 * blockIdx.x] = (unsigned long long)((long long)__builtin_readcyclecounter() - k_c0);
        a.timing[18 + 4 * blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20);
        a.timing[19 + 4 * blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}

}  // namespace vs
