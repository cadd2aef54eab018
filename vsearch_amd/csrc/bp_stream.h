// bp_stream.h -- the STREAMED walk over the blocked postings: the flat walk (bp_flat.h) with its record loads software-pipelined.
//
// Measured on the flat walk (VS_BP_TIMING chunk anatomy, 4 M docs): of the ~10 k cycles a wave spends on a chunk of 64 lists, 5.6 k
// pass between "record loads issued" and "first record landed" -- every batch exposes one full memory latency, and with 16 waves
// per CU that is exactly what keeps the LDS (the scatter-add, the real bound) at ~85 % while a block is walked.  Here a wave keeps
// TWO batches of NR rounds in registers: the loads of batch k + 1 are issued BEFORE the adds of batch k, so a batch's latency passes
// under the previous batch's adds -- also across the block barrier: the first batch of block b + 1 is in flight while block b is
// finished.  What changes against bp_flat.h:
//
//  * The worklist is a ring of 512 items per wave, filled by a producer that runs ahead of the consumer: `cur` is the chunk being
//    expanded (lane = entry: next record, records left), p1 and p2 the two chunks after it, their directory words in flight, and
//    one more chunk number on its way from the LDS counter.  At most ONE advance per iteration: every iteration issues exactly one
//    directory load (a repeat when nothing advanced) and one batch of record loads, so the s_waitcnt counts are constants.
//  * Chunks of consecutive blocks are dealt from EIGHT counters (block mod 8) that only grow: producers run up to four blocks
//    ahead of the slowest wave's grabs.  The successor of a chunk of block x is the next number of x's counter, or -- the counter has
//    run out: exactly one failing grab per wave and block -- the wave's own chunk (its number) of block x + 1.
//  * A batch never mixes blocks; the batch that empties the ring after the producer has left the block is the wave's LAST of that
//    block (possibly empty): after its adds the wave goes to the block barrier and the epilogue.  The candidate sort borrows the
//    accumulator area (all sums are zero between the epilogue and the next block's adds), not the rings, which hold live items.
#pragma once
#include "bp_flat.h"

namespace vs {

constexpr int kStRing = 512;          // worklist items per wave

template <int RMAX>
__host__ __device__ inline size_t bp_stream_lds_bytes(int ent_cap) {
    return (size_t)8 * RMAX * 4 + (size_t)kScanWaves * kStRing * 4 + 8 * 16 + 32 * 4 + (size_t)ent_cap * 8;
}

// NR = rounds (records per lane) of a batch; two batches are in registers
template <int VM, int NR, int RMAX>
__global__ __launch_bounds__(kScanThreads) void bp_stream_topk(BpArgs a) {
    static_assert(VM == VM_F16 || VM == VM_F32, "valued records");
    static_assert((size_t)8 * RMAX * 4 >= kBpSortBytes && (size_t)8 * RMAX * 4 >= (size_t)kFlCap * 8, "the accumulator area holds the entry sort and the candidate sort");
    static_assert(RMAX == 2 * kScanThreads, "a thread finishes documents 2 t and 2 t + 1");
    static_assert(NR * 64 * 2 <= kStRing, "ring");
    constexpr int QT = 8;
    constexpr int RS = bp_rec_bytes(VM);
    constexpr int kPer = VM == VM_F32 ? 3 : 2;          // loads per record
    constexpr int LPB = NR * kPer;                      // loads per batch
    constexpr uint32_t PLANE = (uint32_t)RMAX * 4u;
    constexpr uint32_t RM = kStRing - 1;
    constexpr int NW = kScanThreads / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* acc = reinterpret_cast<int32_t*>(smem);                                        // [QT][RMAX]
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem + (size_t)QT * PLANE);                // [waves][kStRing]
    unsigned long long* tau = reinterpret_cast<unsigned long long*>(smem + (size_t)QT * PLANE + (size_t)kScanWaves * kStRing * 4);    // [8]
    unsigned long long* upper_sh = tau + 8;                                                 // [8]
    int* sync = reinterpret_cast<int*>(upper_sh + 8);                                       // [16]: chunk counters of block mod 8
    unsigned int* ccnt = reinterpret_cast<unsigned int*>(sync + 16);                        // [16]
    uint2* ent = reinterpret_cast<uint2*>(ccnt + 16);                                       // [ent_cap]: x = column | slot plane offset << 16, y = weight bits
    uint64_t* sortbuf = reinterpret_cast<uint64_t*>(acc);                                   // [kFlCap]: between the epilogue and the next block's adds

    const int tid = threadIdx.x, lane = tid & 63, wv_id = tid >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    uint32_t* ringw = ring + wv_id * kStRing;
    const int K = a.k;
    uint64_t* my_gcand = a.gcand + (size_t)blockIdx.x * QT * kFlCap;
    const int64_t n_blocks = (a.n_rows + a.rows - 1) / a.rows;
    const int64_t items = (int64_t)(a.n_tiles_dev ? a.n_tiles_dev[0] : a.n_tiles) * a.nchunk;
    bool pace_off = false;                      // the lock-step wait timed out once (pace_wait): this workgroup runs free from then on
    const size_t dir_ld = (size_t)a.n_cols + 1;
    const unsigned long long k_rt0 = a.timing ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const long long k_c0 = a.timing ? (long long)__builtin_readcyclecounter() : 0;

    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int tile = (int)(item / a.nchunk), c = (int)(item % a.nchunk);
        const int q0 = a.tiles[tile].x, nq = a.tiles[tile].y;
        long long tm = a.timing ? (long long)__builtin_readcyclecounter() : 0;
        const unsigned long long rt0 = a.timing ? __builtin_amdgcn_s_memrealtime() : 0ull;
        uint32_t tacc[6] = {0u, 0u, 0u, 0u, 0u, 0u};
        auto lap = [&](int phase) {
            if (a.timing) {
                const long long now = (long long)__builtin_readcyclecounter();
                tacc[phase] += (uint32_t)(now - tm);
                tm = now;
            }
        };
        const int b0 = (int)((int64_t)c * a.blocks_per_chunk), b1 = (int)min(n_blocks, (int64_t)b0 + a.blocks_per_chunk);
        __syncthreads();
        const int64_t e0 = a.qptr[q0], e1 = a.qptr[q0 + nq];
        const int n_ent = (int)(e1 - e0);
        {   // entries sorted by column (the accumulator area doubles as the sort buffer)
            uint64_t* skey = reinterpret_cast<uint64_t*>(acc);
            for (int i = tid; i < 8192; i += kScanThreads) {
                uint64_t key = 0;
                if (i < n_ent) {
                    const int64_t e = e0 + i;
                    int qs = 0;
                    while (e >= a.qptr[q0 + qs + 1]) ++qs;
                    const float w = a.qvals[e] * a.qscale[q0 + qs];               // power of two: exact
                    key = (1ull << 63) | ((uint64_t)(0xFFFFu - (uint32_t)a.qcols[e]) << 40) | ((uint64_t)qs << 32) | (uint64_t)__float_as_uint(w);
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
        if (tid < 8) { tau[tid] = 0ull; ccnt[tid] = 0u; upper_sh[tid] = (a.upper && tid < nq) ? a.upper[q0 + tid] : ~0ull; }
        if (tid < 16) sync[tid] = 0;
        __syncthreads();

        const int CE = min(64, max(8, (n_ent + 2 * NW - 1) / (2 * NW)));
        const int n_ch = (n_ent + CE - 1) / CE;
        const int grabs_per_block = max(n_ch, NW);          // a counter's growth per block: n_ch - 16 chunks dealt + one failing grab per wave

        // ---- producer ------------------------------------------------------------------------------------------------------------------
        // the column of this lane's entry in chunk (blk, id); a lane without one gets the pad column (an empty list)
        auto chunk_col = [&](int blk, int id) -> uint32_t {
            const int e = id * CE + lane;
            uint32_t col = ent[min(e, max(n_ent - 1, 0))].x & 0xFFFFu;
            if (!(blk < b1 && id < n_ch && lane < CE && e < n_ent)) col = (uint32_t)a.n_cols;
            return col;
        };
        auto dir_row = [&](int blk) -> const uint32_t* { return a.dir + (size_t)min(blk, max(b1 - 1, b0)) * dir_ld; };
        // the next number of block blk's counter (lane 0's value: read with readfirstlane when it is needed); no blocks past the item's
        auto grab = [&](int blk) -> int {
            int v = 0x3FFFFFFF;
            if (blk < b1 && lane == 0) v = atomicAdd(&sync[(blk - b0) & 7], 1);
            return v;
        };
        // successor of a chunk of block blk, given the number its counter returned
        auto succ = [&](int blk, int gv, int& nblk, int& nid) {
            const int v = __builtin_amdgcn_readfirstlane(gv);
            const int id = (v == 0x3FFFFFFF) ? n_ch : NW + (v - (((blk - b0) >> 3) * grabs_per_block));
            if (id < n_ch) { nblk = blk; nid = id; }
            else { nblk = blk + 1; nid = wv_id; }
        };
        auto load_dir = [&](uint32_t& dst, int blk, uint32_t col) {
            const unsigned long long dp = (unsigned long long)dir_row(blk);
            const unsigned long long dps = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(dp >> 32)) << 32) |
                                           (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)dp);
            const uint32_t doff = col * 4u;
            asm volatile("global_load_dword %0, %1, %2" : "=&v"(dst) : "v"(doff), "s"(dps));
        };
        // cur: the chunk being expanded
        int cblk = b0;
        uint32_t first = 0, rem = 0, etag = 0;
        // p1, p2: the chunks after it (directory words landed / in flight), and the number on its way for the one after p2
        int blk1 = b0, id1 = 0, blk2 = b0, id2 = 0;
        uint32_t nd1 = 0, nd2 = 0, col2 = (uint32_t)a.n_cols;
        int gpend = 0x3FFFFFFF;
        auto set_cur = [&](int blk, int id, uint32_t cd) {
            cblk = blk;
            first = (cd >> 12) << a.al_shift;
            rem = blk < b1 ? (cd & kBpDirRecMask) : 0u;
            etag = (uint32_t)(id * CE + lane) << kFlRecBits;
        };
        if (b0 < b1) {
            // fill the pipeline (once per item: the loads are waited for where they are used)
            set_cur(b0, wv_id, dir_row(b0)[chunk_col(b0, wv_id)]);
            succ(b0, grab(b0), blk1, id1);
            nd1 = dir_row(blk1)[chunk_col(blk1, id1)];
            succ(blk1, grab(blk1), blk2, id2);
            col2 = chunk_col(blk2, id2);
            nd2 = dir_row(blk2)[col2];
            gpend = grab(blk2);
            // (these were compiler-issued loads: make it wait for them HERE -- a load it still believes pending on one of these
            //  registers makes it put s_waitcnt vmcnt(0) in front of every hand-issued load of the loop below)
            asm volatile("" : "+v"(nd1), "+v"(nd2), "+v"(first), "+v"(rem));
        } else {
            cblk = b1;
        }
        // one step forward: cur <- p1 <- p2 <- the counter's next chunk; issues this iteration's directory load
        auto advance = [&]() {
            if (a.knob & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // Everything but the latest batch of record loads has landed: nd1 and nd2 among it (see the header).  The wait and the two
            // copies are ONE asm statement: given a wait with the registers merely tied ("+v"), the compiler satisfied the ties by
            // copying nd1 and nd2 into other registers IN FRONT of the wait -- stale words.
            uint32_t cd, shifted;
            if constexpr (LPB == 8) asm volatile("s_waitcnt vmcnt(8)\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(cd), "=&v"(shifted) : "v"(nd1), "v"(nd2));
            else asm volatile("s_waitcnt vmcnt(9)\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(cd), "=&v"(shifted) : "v"(nd1), "v"(nd2));
            set_cur(blk1, id1, cd);
            blk1 = blk2; id1 = id2; nd1 = shifted;
            const int ob = blk2;
            succ(ob, gpend, blk2, id2);
            col2 = chunk_col(blk2, id2);
            load_dir(nd2, blk2, col2);
            gpend = grab(blk2);
        };
        static_assert(LPB == 8 || LPB == 9, "advance() waits for all but one batch of loads");

        uint32_t head = 0;                      // ring: items [head, head + c0) belong to block nb_blk, the block of the next batch
        int c0 = 0;
        int nb_blk = b0;
        // items of block nb_blk until a batch is full or the producer has left the block; at most one advance
        auto ensure = [&]() -> bool {
            bool advanced = false;
            for (;;) {
                if (cblk != nb_blk) break;
                if (__builtin_amdgcn_ballot_w64(rem != 0u) == 0ull) {          // the chunk is spent: the next one (so that the block's end shows at once)
                    if (advanced) break;
                    advance();
                    advanced = true;
                    continue;
                }
                if (c0 >= NR * 64) break;
                const uint32_t incl = wave_incl_scan(rem);
                const int total = (int)__builtin_amdgcn_readlane((int)incl, 63);
                const int space = (kStRing - 1) - c0;
                const int excl = (int)(incl - rem);
                const int take = min((int)rem, max(space - excl, 0));
                const uint32_t pos = head + (uint32_t)(c0 + excl);
                const uint32_t itv = etag | first;
                for (int i = 0; __builtin_amdgcn_ballot_w64(i < take) != 0ull; ++i)
                    if (i < take) ringw[(pos + (uint32_t)i) & RM] = itv + (uint32_t)i;
                first += (uint32_t)take;
                rem -= (uint32_t)take;
                c0 += min(total, space);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            return advanced;
        };

        // ---- consumer ------------------------------------------------------------------------------------------------------------------
        auto add_record = [&](const u32x4& idv, const u32x4& vav, const u32x4& vbv, const float wq, const uint32_t so) {
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
                const uint32_t off = (t & 1) ? acc_off_hi(dw[t >> 1], 4u, so) : acc_off_lo(dw[t >> 1], 4u, so);
                lds_add(off, (int32_t)vv[t]);
            }
        };
        struct Batch {
            u32x4 ids[NR], va[NR], vb[VM == VM_F32 ? NR : 1];
            uint32_t w[NR], so[NR];
            int n, blk;
            bool last;
        };
        // the next batch: items -> entries -> record loads (always NR rounds of loads: rounds past n read record 0 and add nothing)
        auto assemble = [&](Batch& x) {
            bool advanced = false;
            if (nb_blk < b1) advanced = ensure();
            if (!advanced) load_dir(nd2, blk2, col2);               // (this iteration's directory load: a repeat)
            const int n = nb_blk < b1 ? min(NR * 64, c0) : 0;
            x.n = n;
            x.blk = nb_blk;
            x.last = nb_blk < b1 && c0 - n == 0 && cblk > nb_blk;
            const int bb = min(nb_blk, max(b1 - 1, 0));
            // (a SCALAR load, spelled out: a compiler-issued vector load here would count in vmcnt between the hand-issued loads)
            unsigned long long base_bb;
            {
                const unsigned long long bp = (unsigned long long)(a.base + bb);
                const unsigned long long bps = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(bp >> 32)) << 32) |
                                               (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)bp);
                asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(base_bb) : "s"(bps) : "memory");
            }
            const unsigned long long brec = (unsigned long long)a.rec + base_bb * (unsigned long long)RS;
            uint32_t it[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) it[r] = ringw[(head + (uint32_t)(r * 64 + lane)) & RM];
            if (a.debug) {                                          // debug: items must address records of their block and entries of the tile
                const uint32_t nrec = (uint32_t)(a.base[bb + 1] - a.base[bb]);
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    if (r * 64 + lane < n && ((it[r] & kFlRecMask) >= nrec || (int)(it[r] >> kFlRecBits) >= max(n_ent, 1))) {
                        atomicAdd(a.debug + 0, 1ull);
                        a.debug[1] = ((unsigned long long)it[r] << 32) | (unsigned long long)nrec;
                        a.debug[2] = ((unsigned long long)(uint32_t)nb_blk << 32) | (unsigned long long)(uint32_t)n;
                        it[r] = 0u;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                it[r] = (r * 64 + lane < n) ? it[r] : 0u;
                const uint2 en = ent[it[r] >> kFlRecBits];
                x.w[r] = en.y;
                x.so[r] = (en.x >> 16) + lds0;
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const uint32_t off = __umul24(it[r] & kFlRecMask, (uint32_t)RS);
                if constexpr (VM == VM_F32) load_rec48(x.ids[r], x.va[r], x.vb[r], off, brec);
                else load_rec32(x.ids[r], x.va[r], off, brec);
            }
            head = (head + (uint32_t)n) & RM;
            c0 -= n;
            if (x.last) ++nb_blk;
        };
        // the adds of a batch whose loads were issued one iteration ago: younger than them are this iteration's directory load and
        // record loads
        // (the waits run even for an empty batch: they keep the batch's registers reserved until its loads -- of record 0 -- have
        //  landed; a register the compiler considers dead would be handed to something else and then overwritten by the late load)
        auto process = [&](Batch& x) {
            if (a.knob & 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            if (a.knob & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (a.knob & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if constexpr (VM == VM_F32) wait_loads((NR - 1 - r) * kPer + 1 + LPB, x.ids[r], x.va[r], x.vb[r]);
                else wait_loads((NR - 1 - r) * kPer + 1 + LPB, x.ids[r], x.va[r]);
                if (r * 64 + lane < x.n) {
                    if constexpr (VM == VM_F32) add_record(x.ids[r], x.va[r], x.vb[r], __uint_as_float(x.w[r]), x.so[r]);
                    else add_record(x.ids[r], x.va[r], x.va[r], __uint_as_float(x.w[r]), x.so[r]);
                }
            }
        };
        // the end of a block: barrier, sums -> order keys -> candidates, prune when a buffer could overflow (or at the end)
        auto finish_block = [&](const int b, const bool real) {
            if (a.gtau && tid < nq) { const unsigned long long g = a.gtau[q0 + tid]; if (g > tau[tid]) tau[tid] = g; }
            lap(1);
            if (real && a.pace && items <= (int64_t)gridDim.x && tid == 0 && !pace_off) {
                uint32_t* pc = a.pace + (size_t)c * a.blocks_per_chunk;
                const int rel = b - b0;
                __hip_atomic_fetch_add(pc + rel, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (rel >= a.pace_window) {
                    const uint32_t need = (uint32_t)(items / a.nchunk);
                    if (!pace_wait(pc + rel - a.pace_window, need)) pace_off = true;
                }
            }
            __syncthreads();
            lap(2);
            const int rows_b = (int)min((int64_t)a.rows, a.n_rows - (int64_t)b * a.rows);
            const int d = 2 * tid;
            uint32_t thi[QT];
#pragma unroll
            for (int q = 0; q < QT; ++q) thi[q] = (uint32_t)(tau[q] >> 32);
            if (d < rows_b) {
                const int64_t row = (int64_t)b * a.rows + d;
                uint2 sums[QT];
#pragma unroll
                for (int q = 0; q < QT; ++q) sums[q] = *reinterpret_cast<const uint2*>(acc + q * RMAX + d);
#pragma unroll
                for (int q = 0; q < QT; ++q) *reinterpret_cast<uint2*>(acc + q * RMAX + d) = make_uint2(0u, 0u);
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
            __syncthreads();
            const bool last = b + 1 >= b1;
            uint32_t cnts[QT];
#pragma unroll
            for (int q = 0; q < QT; ++q) cnts[q] = ccnt[q];
            bool any = last;
#pragma unroll
            for (int q = 0; q < QT; ++q) any = any || cnts[q] > (uint32_t)(kFlCap - RMAX);
            if (any) {
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
                // (the sort ran in the accumulator area: zero what it left before the next block's adds)
                for (int i = tid; i < kFlCap * 2; i += kScanThreads) acc[i] = 0;
                __syncthreads();
            }
            lap(4);
            tacc[5] += 1u;
        };

        lap(0);
        if (b0 >= b1) {
            finish_block(b1, false);                                // an empty item: sentinels out (no rows, last)
        } else {
            // the loads of a batch that will never be processed (assembled past the item's last block) and the last directory load: wait
            // for them with their registers tied, for the same reason
            auto drain = [&](Batch& x) {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    if constexpr (VM == VM_F32) wait_loads(0, x.ids[r], x.va[r], x.vb[r]);
                    else wait_loads(0, x.ids[r], x.va[r]);
                }
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(nd1), "+v"(nd2));
            };
            Batch A, B;
            assemble(A);
            for (;;) {
                assemble(B);
                process(A);
                if (A.last) {
                    const bool end = A.blk + 1 >= b1;
                    if (end) drain(B);
                    finish_block(A.blk, true);
                    if (end) break;
                }
                assemble(A);
                process(B);
                if (B.last) {
                    const bool end = B.blk + 1 >= b1;
                    if (end) drain(A);
                    finish_block(B.blk, true);
                    if (end) break;
                }
            }
        }
        if (a.timing) tacc[3] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - rt0);
        if (a.timing && lane == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) atomicAdd(a.timing + i, (unsigned long long)tacc[i]);
        }
    }
    if (a.timing && tid == 0) {
        a.timing[16 + 4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime() - k_rt0;
        a.timing[17 + 4 * blockIdx.x] = (unsigned long long)((long long)__builtin_readcyclecounter() - k_c0);
        a.timing[18 + 4 * blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20);
        a.timing[19 + 4 * blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}

}  // namespace vs
