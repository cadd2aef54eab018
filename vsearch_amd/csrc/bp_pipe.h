// bp_pipe.h -- the PIPELINED walk over the blocked postings: flat worklists on TWO accumulator sets, no block barrier.
//
// rocprofv3 on the flat walk (bp_flat.h; profiles/r03_flat_utilisation.txt): the scatter-add itself binds -- ds_add_u32 on random
// document slots costs ~8 LDS cycles per wave instruction (bank conflicts), the LDS is 88 % busy while a block is walked, as in
// bp_walk_topk -- but a fifth of the time passes at the block barrier and in the epilogue, where the LDS idles, and a wave's chunk
// is a chain of dependent LDS round trips (chunk counter, entry columns, items, entry weights, carry), each queued behind the
// other waves' atomics.  This kernel removes both:
//
//  * A tile has QT = 4 query slots and two accumulator sets [2][QT][2048] (the LDS of one 8-slot set).  A wave that runs out of
//    chunks in block b goes straight on to block b + 1 (other set); the wave that leaves block b LAST finishes it alone (sums ->
//    keys -> candidates, 16 document pairs per lane in four batches of reads) before it takes chunks of block b + 1 -- the dealing
//    is dynamic, it simply takes fewer.  add(b) may start when the epilogue of block b - 2 is done (done_epi[parity], grows by one
//    per block); nobody waits unless it is two blocks ahead.  Candidate buffers are pruned (workgroup sort) only when a slot could
//    overflow in the next epilogue: epilogue(b) raises flag[parity of b]; every wave reads it at ITS start of add(b + 2) -- after
//    done_epi says it is final, and BEFORE the epilogue of block b + 1 it may owe -- so all waves take the same branch into the
//    barrier-and-sort, and no epilogue runs between a raised flag and its prune.  Steady state: no workgroup barrier at all.
//  * Worklist items are 64-bit -- (record | slot << 19, weight) -- in a ring of 512 per wave: the consumer needs ONE read per
//    record, the round left incomplete by a chunk stays where it is.  The next chunk's number (LDS counter), its entries (LDS) and
//    its directory words (global) are fetched inside the current chunk's batch, under the record loads' latency; at the end of a
//    block the "next chunk" is the wave's own first chunk of the next block.
//
// Lock step between work items (BpArgs::pace): the tiles of a chunk of blocks sweep the same blocks; a tile that falls behind
// loses the L2 / Infinity-Cache copies the pack left behind and falls further behind.  Wave 0 of an item counts its arrival at block
// j and waits while the slowest item has not reached block j - window.  Only when every item is resident (items <= workgroups).
#pragma once
#include "bp_flat.h"

namespace vs {

constexpr int kPipeQT = 4;
constexpr int kPipeRing = 512;        // 64-bit items per wave
constexpr int kPipeEntCap = 4032;     // (query, column) entries of a tile: what 160 KB leave beside 2 x 32 KB of sums and 64 KB of rings

template <int RMAX>
__host__ __device__ inline size_t bp_pipe_lds_bytes(int ent_cap) {
    return (size_t)2 * kPipeQT * RMAX * 4 + (size_t)kScanWaves * kPipeRing * 8 + 8 * 16 + 32 * 4 + (size_t)ent_cap * 8;
}

// NR = record loads in flight per lane (<= 7: a ring holds 7 full rounds + an incomplete one)
template <int VM, int NR, int RMAX>
__global__ __launch_bounds__(kScanThreads) void bp_pipe_topk(BpArgs a) {
    static_assert(VM == VM_F16 || VM == VM_F32, "valued records");
    constexpr int QT = kPipeQT;
    static_assert((size_t)2 * QT * RMAX * 4 >= kBpSortBytes, "the accumulator area holds the 8192-slot entry sort");
    static_assert(RMAX == 2048 && kScanThreads == 1024, "plane shift, document pairs per lane");
    static_assert(NR * 64 + 64 <= kPipeRing && (size_t)kScanWaves * kPipeRing * 8 >= (size_t)kFlCap * 8, "ring / candidate sort area");
    constexpr int RS = bp_rec_bytes(VM);
    constexpr uint32_t PLANE = (uint32_t)RMAX * 4u;     // bytes of a slot plane (2^13)
    constexpr uint32_t SET = (uint32_t)QT * PLANE;      // bytes of an accumulator set
    constexpr int NW = kScanThreads / 64;
    constexpr uint32_t RM = kPipeRing - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* acc = reinterpret_cast<int32_t*>(smem);                                        // [2][QT][RMAX]
    uint2* ring = reinterpret_cast<uint2*>(smem + 2 * SET);                                 // [waves][kPipeRing]
    uint64_t* sortbuf = reinterpret_cast<uint64_t*>(ring);                                  // [kFlCap] (all waves between blocks: the rings are empty)
    unsigned long long* tau = reinterpret_cast<unsigned long long*>(smem + 2 * SET + (size_t)kScanWaves * kPipeRing * 8);    // [8]
    unsigned long long* upper_sh = tau + 8;                                                 // [8]
    int* sync = reinterpret_cast<int*>(upper_sh + 8);                                       // [16]: chunk counters [0..1], done_add [2..3], done_epi [4..5], flag [6..7]
    unsigned int* ccnt = reinterpret_cast<unsigned int*>(sync + 16);                        // [16]
    uint2* ent = reinterpret_cast<uint2*>(ccnt + 16);                                       // [ent_cap]: x = column | slot << 16, y = weight bits

    const int tid = threadIdx.x, lane = tid & 63, wv_id = tid >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    uint2* ringw = ring + wv_id * kPipeRing;
    const int K = a.k;
    uint64_t* my_gcand = a.gcand + (size_t)blockIdx.x * QT * kFlCap;
    const int64_t n_blocks = (a.n_rows + a.rows - 1) / a.rows;
    const int64_t items = (int64_t)(a.n_tiles_dev ? a.n_tiles_dev[0] : a.n_tiles) * a.nchunk;
    const size_t dir_ld = (size_t)a.n_cols + 1;
    const bool paced = a.pace != nullptr && items <= (int64_t)gridDim.x;
    const unsigned long long k_rt0 = a.timing ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const long long k_c0 = a.timing ? (long long)__builtin_readcyclecounter() : 0;

    // The counters live in LDS, which executes a CU's instructions in the order it receives them, a wave's own in program order:
    // a counter update issued after a wave's adds is performed after them, and reads issued after a counter read see what
    // preceded the update.  So the synchronising accesses are RELAXED atomics between wavefront-scope fences (compiler ordering
    // only): a workgroup-scope acquire / release would also drain the global loads in flight -- the next chunk's directory words.
    auto ld_acq = [&](const int* p) {
        const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        return v;
    };
    auto spin_ge = [&](const int* p, int target) { while (ld_acq(p) < target) __builtin_amdgcn_s_sleep(1); };

    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int tile = (int)(item / a.nchunk), c = (int)(item % a.nchunk);
        const int q0 = a.tiles[tile].x, nq = a.tiles[tile].y;
        long long tm = a.timing ? (long long)__builtin_readcyclecounter() : 0;
        const unsigned long long rt0 = a.timing ? __builtin_amdgcn_s_memrealtime() : 0ull;
        uint32_t tacc[6] = {0u, 0u, 0u, 0u, 0u, 0u};
        uint32_t tch[4] = {0u, 0u, 0u, 0u};       // chunk anatomy (VS_BP_TIMING): produce, item round trip, first record landed, adds drained
        long long tc = 0;
        auto clap = [&](int ph) { if (a.timing) { const long long now = (long long)__builtin_readcyclecounter(); tch[ph] += (uint32_t)(now - tc); tc = now; } };
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
                ent[i] = make_uint2(col | ((uint32_t)(key >> 32) & 0xFFu) << 16, (uint32_t)key);
            }
            __syncthreads();
        }
        for (int i = tid; i < 2 * QT * RMAX; i += kScanThreads) acc[i] = 0;
        if (tid < 8) { tau[tid] = 0ull; ccnt[tid] = 0u; upper_sh[tid] = (a.upper && tid < nq) ? a.upper[q0 + tid] : ~0ull; }
        if (tid < 16) sync[tid] = 0;
        __syncthreads();

        // A chunk = CE consecutive entries, one per lane; a wave's first chunk of a block is its own number, the following ones come
        // from a counter in LDS that only grows: max(n_ch, 16) per block (every processed chunk grabs once; the last 16 grabs fail).
        const int CE = min(64, max(8, (n_ent + 2 * NW - 1) / (2 * NW)));
        const int n_ch = (n_ent + CE - 1) / CE;
        const int grabs_per_block = max(n_ch, NW);
        // this lane's entry in chunk ch (a lane without one gets the pad column: an empty list)
        auto entry_of = [&](int ch) -> uint2 {
            const int e = ch * CE + lane;
            uint2 v = ent[min(e, max(n_ent - 1, 0))];
            if (!(ch < n_ch && lane < CE && e < n_ent)) v = make_uint2((uint32_t)a.n_cols, 0u);
            return v;
        };

        // ---- epilogue: sums -> order keys -> candidates ------------------------------------------------------------------------------
        // one document pair (d, d + 1) of block x from its sums; returns "a slot could overflow in the next epilogue"
        auto finish_pair = [&](const int64_t x, const int d, const int rows_x, const uint2 (&sums)[QT], const uint32_t (&thi)[QT]) -> bool {
            bool full = false;
            const int64_t row = x * a.rows + d;
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                const uint32_t h0 = sums[q].x ^ 0x80000000u, h1 = sums[q].y ^ 0x80000000u;
                if (q < nq && d < rows_x && (h0 >= thi[q] || h1 >= thi[q])) {
                    const uint64_t k0 = ((uint64_t)h0 << 32) | (uint32_t)(~(uint32_t)row);
                    const uint64_t k1 = ((uint64_t)h1 << 32) | (uint32_t)(~(uint32_t)(row + 1));
                    const unsigned long long tq = tau[q], uq = upper_sh[q];
                    if (k0 > tq && k0 < uq) {
                        const uint32_t pos = atomicAdd(&ccnt[q], 1u);
                        my_gcand[(size_t)q * kFlCap + pos] = k0;
                        full = full || pos + 1u > (uint32_t)(kFlCap - RMAX);
                    }
                    if (d + 1 < rows_x && k1 > tq && k1 < uq) {
                        const uint32_t pos = atomicAdd(&ccnt[q], 1u);
                        my_gcand[(size_t)q * kFlCap + pos] = k1;
                        full = full || pos + 1u > (uint32_t)(kFlCap - RMAX);
                    }
                }
            }
            return full;
        };
        // ONE wave finishes block x: lane l takes pairs 2 (64 i + l), i = 0 .. 15, four at a time (16 reads in flight: a read waits
        // behind the other waves' queued atomics)
        auto wave_epilogue = [&](const int64_t x) {
            const int px = (int)((x - b0) & 1);
            const int rows_x = (int)min((int64_t)a.rows, a.n_rows - x * a.rows);
            int32_t* base = acc + px * QT * RMAX;
            uint32_t thi[QT];
#pragma unroll
            for (int q = 0; q < QT; ++q) thi[q] = (uint32_t)(tau[q] >> 32);
            bool full = false;
#pragma unroll 1
            for (int g = 0; g < RMAX / 128 / 4; ++g) {
                uint2 sums[4][QT];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < QT; ++q) sums[u][q] = *reinterpret_cast<const uint2*>(base + q * RMAX + 2 * ((g * 4 + u) * 64 + lane));
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < QT; ++q) *reinterpret_cast<uint2*>(base + q * RMAX + 2 * ((g * 4 + u) * 64 + lane)) = make_uint2(0u, 0u);
#pragma unroll
                for (int u = 0; u < 4; ++u) full = finish_pair(x, 2 * ((g * 4 + u) * 64 + lane), rows_x, sums[u], thi) || full;
            }
            if (full) sync[6 + px] = 1;                         // prune before the next epilogue
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (lane == 0) __hip_atomic_fetch_add(&sync[4 + px], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        // candidate buffers beyond kFlCap - RMAX (or all, at the end) -> sort, keep the K best, raise the threshold.  Whole workgroup.
        auto prune = [&](const bool last) {
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
        };

        // ---- the walk ----------------------------------------------------------------------------------------------------------------
        bool owe = false;                        // this wave left the previous block last: it owes that block's epilogue
        // the chunk about to be walked: its entry (column | slot << 16, weight) and its directory word in flight
        uint2 en = (b0 < b1) ? entry_of(wv_id) : make_uint2((uint32_t)a.n_cols, 0u);
        uint32_t nd = 0;
        if (b0 < b1) nd = (a.dir + (size_t)b0 * dir_ld)[en.x & 0xFFFFu];
        lap(0);
        for (int64_t b = b0; b < b1; ++b) {
            const int j = (int)(b - b0), p = j & 1, jj = j >> 1;
            // the epilogue of block b - 2 done (by the wave that left it last): its accumulator set is free, its overflow flag final
            if (j >= 2) {
                spin_ge(&sync[4 + p], jj);
                lap(2);
                if (ld_acq(&sync[6 + p]) != 0) {                // uniform over the workgroup (see the header)
                    __syncthreads();
                    prune(false);
                    if (tid == 0) sync[6 + p] = 0;
                    __syncthreads();
                    lap(4);
                }
            }
            // (after the overflow check: a raised flag means the buffers cannot take another epilogue before the prune)
            if (owe) { wave_epilogue(b - 1); owe = false; lap(4); }
            // thresholds other items of the same queries have published meanwhile; lock step with the chunk's other items
            if (wv_id == 0) {
                if (a.gtau && lane < nq) { const unsigned long long g = a.gtau[q0 + lane]; if (g > tau[lane]) tau[lane] = g; }
                if (paced && lane == 0) {
                    uint32_t* pc = a.pace + (size_t)c * a.blocks_per_chunk;
                    __hip_atomic_fetch_add(pc + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (j >= a.pace_window) {
                        const uint32_t need = (uint32_t)(items / a.nchunk);
                        while (__hip_atomic_load(pc + j - a.pace_window, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(8);
                    }
                }
                lap(2);
            }
            const uint32_t* dirb = a.dir + (size_t)b * dir_ld;
            const uint32_t* dirn = a.dir + (size_t)min(b + 1, b1 - 1) * dir_ld;
            const unsigned long long pb = (unsigned long long)(a.rec + (size_t)a.base[b] * RS);
            const unsigned long long brec = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32)) << 32) |
                                            (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pb);
            const uint32_t accb = lds0 + (p ? SET : 0u);           // LDS byte address of this block's accumulator set

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
            // the directory word of column `col` in directory row `dirp`: asm, so that it takes ITS place among the record loads
            auto load_dir = [&](const uint32_t* dirp, const uint32_t col) {
                const unsigned long long dp = (unsigned long long)dirp;
                const unsigned long long dps = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(dp >> 32)) << 32) |
                                               (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)dp);
                const uint32_t doff = col * 4u;
                asm volatile("global_load_dword %0, %1, %2" : "=&v"(nd) : "v"(doff), "s"(dps));
            };

            // chunk bookkeeping: `cur` is walked now; the next one is grabbed at its top and resolved inside its first batch
            const int cbase = jj * grabs_per_block;
            auto grab = [&]() {
                int v = 0;
                if (lane == 0) v = atomicAdd(&sync[p], 1);
                return v;                                       // (lane 0's value: read with readfirstlane where it is needed)
            };
            int cur = wv_id;
            if (cur >= n_ch) (void)grab();                      // (a wave without a chunk of its own still counts its one failing grab)
            uint32_t head = 0;                                  // ring: items [head, head + carried) wait for their round
            int carried = 0;
            while (cur < n_ch) {
                // the chunk's directory words have landed (they were fetched during the previous chunk / block)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(nd));
                const uint32_t cd = nd;
                uint32_t first = (cd >> 12) << a.al_shift;     // next record of this lane's list (relative to the block)
                uint32_t rem = cd & kBpDirRecMask;              // records left
                const uint32_t itx = ((en.x >> 16) & 3u) << kFlRecBits;
                const uint32_t ity = en.y;
                if (a.timing) tc = (long long)__builtin_readcyclecounter();
                const int gv = grab();                          // the next chunk: resolved below, inside the batch
                bool fetched = false;
                int nch = n_ch;                                 // next chunk's number, or >= n_ch: none left in this block
                uint2 nen = make_uint2((uint32_t)a.n_cols, 0u);
                const uint32_t* ndir = dirb;
                // the next chunk's entry: this block's chunk `nch`, or -- the counter has run out -- the wave's own first chunk of
                // the next block
                auto fetch_next = [&]() {
                    nch = NW + (__builtin_amdgcn_readfirstlane(gv) - cbase);
                    const bool here = nch < n_ch;
                    nen = entry_of(here ? nch : (b + 1 < b1 ? wv_id : n_ch));
                    ndir = here ? dirb : dirn;
                    fetched = true;
                };
                do {
                    const uint32_t incl = wave_incl_scan(rem);
                    const int total = (int)__builtin_amdgcn_readlane((int)incl, 63);
                    const int space = (kPipeRing - 1) - carried;       // carried + new < 512
                    const int excl = (int)(incl - rem);
                    const int take = min((int)rem, max(space - excl, 0));
                    const uint32_t pos = head + (uint32_t)(carried + excl);
                    for (int i = 0; __builtin_amdgcn_ballot_w64(i < take) != 0ull; ++i)
                        if (i < take) ringw[(pos + (uint32_t)i) & RM] = make_uint2(itx | (first + (uint32_t)i), ity);
                    first += (uint32_t)take;
                    rem -= (uint32_t)take;
                    const int have = carried + min(total, space);
                    const int full = min(have >> 6, NR);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    clap(0);
                    if (full > 0) {
                        // One batch: `full` rounds of 64 items; all record loads are issued back to back (asm: the compiler would sink
                        // them into the predicated adds), each round waits for its own.  Rounds past `full` load record 0, add nothing.
                        uint2 it[NR];
#pragma unroll
                        for (int r = 0; r < NR; ++r) it[r] = ringw[(head + (uint32_t)(r * 64 + lane)) & RM];
                        if (a.timing) { asm volatile("" :: "v"(it[NR - 1].x)); clap(1); }
                        if (!fetched) fetch_next();             // (its LDS reads return while the record loads fly)
                        u32x4 ids[NR], va[NR];
                        [[maybe_unused]] u32x4 vb[NR];
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            if (r >= full) it[r] = make_uint2(0u, 0u);
                            const uint32_t off = __umul24(it[r].x & kFlRecMask, (uint32_t)RS);
                            if constexpr (VM == VM_F32) load_rec48(ids[r], va[r], vb[r], off, brec);
                            else load_rec32(ids[r], va[r], off, brec);
                        }
                        // the next chunk's directory words, behind the record loads (every batch issues the load: the waits count it)
                        load_dir(ndir, nen.x & 0xFFFFu);
                        constexpr int kPer = VM == VM_F32 ? 3 : 2;
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            if constexpr (VM == VM_F32) wait_loads((NR - 1 - r) * kPer + 1, ids[r], va[r], vb[r]);
                            else wait_loads((NR - 1 - r) * kPer + 1, ids[r], va[r]);
                            if (r == 0) clap(2);
                            if (r < full) {
                                const uint32_t so = accb + ((it[r].x >> kFlRecBits) << 13);
                                if constexpr (VM == VM_F32) add_record(ids[r], va[r], vb[r], __uint_as_float(it[r].y), so);
                                else add_record(ids[r], va[r], va[r], __uint_as_float(it[r].y), so);
                            }
                        }
                        head = (head + (uint32_t)(full * 64)) & RM;
                        clap(3);
                    }
                    carried = have - full * 64;
                    __builtin_amdgcn_wave_barrier();
                } while (__builtin_amdgcn_ballot_w64(rem != 0u) != 0ull || carried >= 64);
                if (!fetched) {                                 // a chunk without a full round: fetch the next one's words here
                    fetch_next();
                    load_dir(ndir, nen.x & 0xFFFFu);
                }
                cur = nch;
                en = nen;
            }
            // the block's incomplete round: lanes < carried
            if (carried > 0) {
                uint2 it = ringw[(head + (uint32_t)lane) & RM];
                if (lane >= carried) it = make_uint2(0u, 0u);
                u32x4 ids, va;
                [[maybe_unused]] u32x4 vb;
                const uint32_t off = __umul24(it.x & kFlRecMask, (uint32_t)RS);
                // (the wait also drains the next block's directory words: harmless)
                if constexpr (VM == VM_F32) { load_rec48(ids, va, vb, off, brec); wait_loads(0, ids, va, vb); }
                else { load_rec32(ids, va, off, brec); wait_loads(0, ids, va); }
                if (lane < carried) {
                    const uint32_t so = accb + ((it.x >> kFlRecBits) << 13);
                    if constexpr (VM == VM_F32) add_record(ids, va, vb, __uint_as_float(it.y), so);
                    else add_record(ids, va, va, __uint_as_float(it.y), so);
                }
            }
            {   // the last of the 16 waves to leave the block finishes it -- except the item's last block (all waves, below)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                int old = 0;
                if (lane == 0) old = __hip_atomic_fetch_add(&sync[2 + p], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                old = __builtin_amdgcn_readfirstlane(old);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                owe = (old + 1 == NW * (jj + 1)) && (b + 1 < b1);
            }
            tacc[5] += 1u;
            lap(1);
        }
        // (the directory load issued for a block past the item's last is still in flight: wait with its register tied, or the register is
        //  handed to something else and then overwritten by the late load)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(nd));
        // the end of the item: every epilogue but the last block's is done (a wave finishes the block it owes before it gets here);
        // the overflow check of block b1 - 2, then the last block's epilogue by all waves
        __syncthreads();
        lap(2);
        if (b1 - b0 >= 2 && sync[6 + (int)((b1 - 2 - b0) & 1)] != 0) {
            __syncthreads();
            prune(false);
            __syncthreads();
        }
        if (b0 < b1) {
            const int64_t x = b1 - 1;
            const int rows_x = (int)min((int64_t)a.rows, a.n_rows - x * a.rows);
            int32_t* base = acc + (int)((x - b0) & 1) * QT * RMAX;
            uint32_t thi[QT];
            uint2 sums[QT];
#pragma unroll
            for (int q = 0; q < QT; ++q) { thi[q] = (uint32_t)(tau[q] >> 32); sums[q] = *reinterpret_cast<const uint2*>(base + q * RMAX + 2 * tid); }
            (void)finish_pair(x, 2 * tid, rows_x, sums, thi);
        }
        __syncthreads();
        prune(true);
        lap(4);
        if (a.timing) tacc[3] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - rt0);
        if (a.timing && lane == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) atomicAdd(a.timing + i, (unsigned long long)tacc[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(a.timing + 8 + i, (unsigned long long)tch[i]);
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
