// bp_bin.h -- the walk over the blocked postings of a BINARY (bag-of-token) index, record loads prefetched across the block barrier.
//
// A bag-of-token block is all latency: ~6 postings a list (one 16-byte record, a second one for 1 list in 6), 37 k adds per
// (8-query tile, 2048-document block) -- 5 k LDS cycles -- against 28 k cycles the list walk spends on it (bp_walk_topk, phase
// clocks: walk 16.2 k, barrier wait 7.5 k, epilogue 4.7 k).  A wave has ONE chunk per block (6208 entries = 14 chunks of 448 for
// 16 waves), so a block is: directory words -> record loads -> adds -> barrier -> epilogue -> barrier, each step waiting for
// the one before.  Here the chain is cut twice:
//   * the RECORD loads of block b + 1 (the first record of every list) are issued before block b's barrier and land during its
//     epilogue; the second records (1 list in 6 has one) are issued right after the barrier and land while the first ones are added;
//   * the DIRECTORY words of block b + 2 are issued right after them and have a whole block to land.
// After the barrier a wave finds its records in registers and only has to add.  A wave's chunk is its own number (static): chunk c
// holds entries c C .. c C + C - 1, C = ceil(entries / 16), lane l takes entries c C + 64 u + l, u = 0 .. 7 (neighbouring lanes
// read neighbouring directory words and lists).
// Accumulators slot-major [8][2048 + 66]: pad postings of a binary list carry document id 2048 (bp_fill_kernel), the spare
// documents behind each plane absorb them.  Epilogue, candidate keys, thresholds, output: bp_flat_topk's.
//
// Hand-issued loads (rules learnt in bp_stream.h): every one gets a wait with its registers tied; values are copied out of a
// register in the SAME asm statement that waits for it; no compiler-issued vector load between them.
#pragma once
#include "bp_flat.h"

namespace vs {

constexpr int kBinSpare = 66;         // spare documents behind each slot plane: pad postings carry document id kBpRowsMaxBin, and with a plane
                                      // pitch of 2114 = 2 (mod 32) the 8 slots' pad words fall on 8 different LDS banks (a pitch of 2112 put all 16 pad
                                      // lanes of a ds_add on bank 0); 2114 * 4 bytes keeps the planes 8-byte aligned for the epilogue's ds_read_b64

template <int RMAX>
__host__ __device__ inline size_t bp_bin_lds_bytes(int ent_cap) {
    return (size_t)8 * (RMAX + kBinSpare) * 4 + (size_t)kFlCap * 8 + 8 * 16 + 32 * 4 + (size_t)ent_cap * 8;
}

template <int RMAX>
__global__ __launch_bounds__(kScanThreads) void bp_bin_topk(BpArgs a) {
    static_assert(RMAX == kBpRowsMaxBin && RMAX == 2 * kScanThreads, "pad postings carry document id kBpRowsMaxBin; a thread finishes documents 2 t and 2 t + 1");
    constexpr int QT = 8, NB = 8;
    constexpr int RS = 16;
    constexpr uint32_t PLANE = (uint32_t)(RMAX + kBinSpare) * 4u;
    static_assert((size_t)QT * PLANE >= kBpSortBytes && 7u * PLANE < 65536u, "entry sort area; slot offsets fit 16 bits");
    constexpr int NW = kScanThreads / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* acc = reinterpret_cast<int32_t*>(smem);                                        // [QT][RMAX + spare]
    uint64_t* sortbuf = reinterpret_cast<uint64_t*>(smem + (size_t)QT * PLANE);             // [kFlCap]
    unsigned long long* tau = reinterpret_cast<unsigned long long*>(sortbuf + kFlCap);      // [8]
    unsigned long long* upper_sh = tau + 8;                                                 // [8]
    int* scratch = reinterpret_cast<int*>(upper_sh + 8);                                    // [16]
    unsigned int* ccnt = reinterpret_cast<unsigned int*>(scratch + 16);                     // [16]
    uint2* ent = reinterpret_cast<uint2*>(ccnt + 16);                                       // [ent_cap]: x = column | slot plane offset << 16, y = weight bits

    const int tid = threadIdx.x, lane = tid & 63, wv_id = tid >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int K = a.k;
    uint64_t* my_gcand = a.gcand + (size_t)blockIdx.x * QT * kFlCap;
    const int64_t n_blocks = (a.n_rows + a.rows - 1) / a.rows;
    const int64_t items = (int64_t)(a.n_tiles_dev ? a.n_tiles_dev[0] : a.n_tiles) * a.nchunk;
    bool pace_off = false;                      // the lock-step wait timed out once (pace_wait): this workgroup runs free from then on
    const size_t dir_ld = (size_t)a.n_cols + 1;

    // Block barrier that does NOT drain the vector-memory counter: __syncthreads() comes with s_waitcnt vmcnt(0), which would make
    // every wave wait for the next block's records right here.  What the barrier orders is LDS (the sums, the candidate
    // counters): this wave's LDS operations are complete (lgkmcnt) before it arrives.  Global stores to the candidate buffers
    // are read back only inside the prune, behind a full __syncthreads().
    auto lds_barrier = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    const unsigned long long k_rt0 = a.timing ? __builtin_amdgcn_s_memrealtime() : 0ull;
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int tile = (int)(item / a.nchunk), c = (int)(item % a.nchunk);
        const int q0 = a.tiles[tile].x, nq = a.tiles[tile].y;
        long long tm = a.timing ? (long long)__builtin_readcyclecounter() : 0;
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
        for (int i = tid; i < (int)(QT * PLANE / 4); i += kScanThreads) acc[i] = 0;
        if (tid < 8) { tau[tid] = 0ull; ccnt[tid] = 0u; upper_sh[tid] = (a.upper && tid < nq) ? a.upper[q0 + tid] : ~0ull; }
        __syncthreads();

        // this wave's chunk: entries wv_id * C + 64 u + lane.  Columns, integer weights and slot plane addresses stay in registers
        // for the whole item (they do not depend on the block).
        const int C = max(1, (n_ent + NW - 1) / NW);                   // an even share for every wave (<= 448 = 7 slots of 64 lanes at the entry capacity)
        // (the entries are re-read from LDS where they are used -- 16 reads a block -- to leave the registers to the 16 records in flight)
        auto entry = [&](int u) -> uint2 {                               // -> (column | slot plane offset << 16, integer weight); no entry: the pad column, weight 0
            const int e = wv_id * C + 64 * u + lane;
            const bool ok = 64 * u + lane < C && e < n_ent;
            uint2 en = ent[min(e, max(n_ent - 1, 0))];
            en.y = ok ? (uint32_t)(int32_t)__uint_as_float(en.y) : 0u;
            if (!ok) en.x = (en.x & 0xFFFF0000u) | (uint32_t)a.n_cols;
            return en;
        };
        auto add_record = [&](const u32x4& idv, const int32_t w, const uint32_t s) {
            const uint32_t dw[4] = {idv.x, idv.y, idv.z, idv.w};
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint32_t off = (t & 1) ? acc_off_hi(dw[t >> 1], 4u, s) : acc_off_lo(dw[t >> 1], 4u, s);
                lds_add(off, w);
            }
        };
        auto sgpr_ptr = [&](const void* p) -> unsigned long long {
            const unsigned long long v = (unsigned long long)p;
            return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
        };
        auto block_base = [&](int b) -> unsigned long long {                 // first record of block b: a scalar load, spelled out
            unsigned long long v;
            const unsigned long long bp = sgpr_ptr(a.base + min(b, max(b1 - 1, 0)));
            asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(bp) : "memory");
            return (unsigned long long)a.rec + v * (unsigned long long)RS;
        };
        // directory words of this wave's chunk in block b -> nd[] (8 loads)
        uint32_t nd[NB];
        auto load_dirs = [&](int b, const uint2 (&en)[NB]) {
            const unsigned long long dps = sgpr_ptr(a.dir + (size_t)min(b, max(b1 - 1, 0)) * dir_ld);
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const uint32_t doff = (en[u].x & 0xFFFFu) * 4u;
                asm volatile("global_load_dword %0, %1, %2" : "=&v"(nd[u]) : "v"(doff), "s"(dps));
            }
        };
        // the chunk's records of one block, two per list: r1[u] = record `first`, r2[u] = record `first + 1`; fst / cnt describe the lists
        // the chunk's first records of one block: r1[u] = record fst[u] of list u; cpack: records of the 8 lists, 4 bits each (15 = 15 or more)
        u32x4 r1[NB];
        uint32_t fst[NB];
        uint32_t cpack = 0u;
        auto load_records = [&](int b, const unsigned long long brec) {
            // (the directory words were issued a block ago; wait and unpack in ONE statement per word)
            cpack = 0u;
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                uint32_t w;
                asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, %1" : "=&v"(w) : "v"(nd[u]));
                fst[u] = (w >> 12) << a.al_shift;
                cpack |= (b < b1 ? min(w & kBpDirRecMask, 15u) : 0u) << (4 * u);
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) load_rec16(r1[u], __umul24(fst[u], (uint32_t)RS), brec);
        };
        if (b0 < b1) {
            uint2 en[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) en[u] = entry(u);
            load_dirs(b0, en);
            load_records(b0, block_base(b0));
            load_dirs(b0 + 1, en);
        }
        lap(0);
        for (int b = b0; b < b1 || b == b0; ++b) {
            const bool have_b = b < b1;
            const int rows_b = have_b ? (int)min((int64_t)a.rows, a.n_rows - (int64_t)b * a.rows) : 0;
            if (have_b) {
                // The first records are in flight since before the previous block's barrier.  The second records (past a list's end:
                // the next list's record, not added; the array ends with two spare records) go out now and land under the first ones'
                // adds.  Younger than the first records: the 8 directory loads of block b + 1 and these 8.
                // (the base pointers of this and the next block and the wave's entries FIRST: a scalar-load wait or an LDS read between the
                //  adds would wait for every add issued before it)
                const unsigned long long brec_b = block_base(b), brec_n = block_base(b + 1);
                u32x4 r2[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) load_rec16(r2[u], __umul24(fst[u] + 1u, (uint32_t)RS), brec_b);
                uint2 en[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) en[u] = entry(u);
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    wait_loads((NB - 1 - u) + 2 * NB, r1[u]);
                    if (((cpack >> (4 * u)) & 15u) > 0u) add_record(r1[u], (int32_t)en[u].y, (en[u].x >> 16) + lds0);
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    wait_loads(NB - 1 - u, r2[u]);
                    if (((cpack >> (4 * u)) & 15u) > 1u) add_record(r2[u], (int32_t)en[u].y, (en[u].x >> 16) + lds0);
                }
                // lists beyond 16 postings (2 in 10 000 at 6 postings a list): their directory word again, further records one by one
                bool more = false;
#pragma unroll
                for (int u = 0; u < NB; ++u) more = more || ((cpack >> (4 * u)) & 15u) > 2u;
                if (__builtin_amdgcn_ballot_w64(more) != 0ull) {
                    const unsigned long long brec = brec_b;
                    const uint32_t* dirb = a.dir + (size_t)b * dir_ld;
#pragma unroll 1
                    for (int u = 0; u < NB; ++u) {
                        const uint32_t cn = (cpack >> (4 * u)) & 15u;
                        if (__builtin_amdgcn_ballot_w64(cn > 2u) == 0ull) continue;
                        const uint2 eu = entry(u);
                        const uint32_t w = cn > 2u ? dirb[eu.x & 0xFFFFu] : 0u;       // (a compiler-issued load: it drains the counter, which is fine here)
                        const uint32_t first = (w >> 12) << a.al_shift, total = w & kBpDirRecMask;
                        for (uint32_t j = 2; __builtin_amdgcn_ballot_w64(j < total) != 0ull; ++j) {
                            u32x4 x;
                            const uint32_t off = __umul24(j < total ? first + j : 0u, (uint32_t)RS);
                            load_rec16(x, off, brec);
                            wait_loads(0, x);
                            if (j < total) add_record(x, (int32_t)eu.y, (eu.x >> 16) + lds0);
                        }
                    }
                }
                // thresholds other items of the same queries have published meanwhile (a compiler-issued load: it is waited for with
                // vmcnt(0) -- BEFORE the prefetch below is issued, not after)
                if (a.gtau && tid < nq) { const unsigned long long g = a.gtau[q0 + tid]; if (g > tau[tid]) tau[tid] = g; }
                // block b + 1's records (its directory words were issued a block ago), then block b + 2's directory words
                load_records(b + 1, brec_n);
                load_dirs(b + 2, en);
            }
            lap(1);
            // lock step with the chunk's other items (BpArgs::pace, bp_walk.h): this walk runs ahead of its memory (prefetches), and an
            // item that falls out of the Infinity-Cache window of the pack never catches up (free running: 55 ms for most workgroups,
            // 83 ms for the last)
            if (a.pace && items <= (int64_t)gridDim.x && tid == 0 && have_b && !pace_off) {
                uint32_t* pc = a.pace + (size_t)c * a.blocks_per_chunk;
                const int rel = b - b0;
                if (!((a.knob & 64) && blockIdx.x == 0))              // (VS_BP_KNOB=64, tests: workgroup 0 never reports -- every peer's wait must time out, not hang)
                    __hip_atomic_fetch_add(pc + rel, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (rel >= a.pace_window) {
                    const uint32_t need = (uint32_t)(items / a.nchunk);
                    if (!pace_wait(pc + rel - a.pace_window, need)) pace_off = true;
                }
            }
            lds_barrier();
            lap(2);
            {
                const int d = 2 * tid;
                uint32_t thi[QT];
#pragma unroll
                for (int q = 0; q < QT; ++q) thi[q] = (uint32_t)(tau[q] >> 32);
                if (d < rows_b) {
                    const int64_t row = (int64_t)b * a.rows + d;
                    uint2 sums[QT];
#pragma unroll
                    for (int q = 0; q < QT; ++q) sums[q] = *reinterpret_cast<const uint2*>(acc + q * (RMAX + kBinSpare) + d);
#pragma unroll
                    for (int q = 0; q < QT; ++q) *reinterpret_cast<uint2*>(acc + q * (RMAX + kBinSpare) + d) = make_uint2(0u, 0u);
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
                lds_barrier();
                const bool last = b + 1 >= b1;
                uint32_t cnts[QT];
#pragma unroll
                for (int q = 0; q < QT; ++q) cnts[q] = ccnt[q];
                bool any = last;
#pragma unroll
                for (int q = 0; q < QT; ++q) any = any || cnts[q] > (uint32_t)(kFlCap - RMAX);
                if (any) __syncthreads();                        // (the candidates pushed above are read back: a full barrier)
                if (any)
                for (int qs = 0; qs < nq; ++qs) {
                    const uint32_t cn = ccnt[qs];
                    if (last || cn > (uint32_t)(kFlCap - RMAX)) {
                        for (int i = tid; i < kFlCap; i += kScanThreads) sortbuf[i] = (uint32_t)i < cn ? my_gcand[(size_t)qs * kFlCap + i] : 0ull;
                        wg_sort_desc<kScanThreads>(sortbuf, kFlCap, tid);
                        if (last) {
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
                        }
                        __syncthreads();
                    }
                }
            }
            lap(4);
            tacc[5] += 1u;
            if (b + 1 >= b1) break;
        }
        // the loads issued for the blocks past the item's last: wait for them with their registers tied (a register the compiler
        // considers dead is handed to something else and then overwritten by the late load)
        if (b0 < b1) {
#pragma unroll
            for (int u = 0; u < NB; ++u) wait_loads(0, r1[u]);
#pragma unroll
            for (int u = 0; u < NB; ++u) asm volatile("s_waitcnt vmcnt(0)" : "+v"(nd[u]));
        }
        if (a.timing && lane == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) atomicAdd(a.timing + i, (unsigned long long)tacc[i]);
            // wave 0 (thresholds, lock step) against a plain wave: walk and barrier wait
            if (wv_id == 0) { atomicAdd(a.timing + 12, (unsigned long long)tacc[1]); atomicAdd(a.timing + 13, (unsigned long long)tacc[2]); }
            if (wv_id == 8) { atomicAdd(a.timing + 14, (unsigned long long)tacc[1]); atomicAdd(a.timing + 15, (unsigned long long)tacc[2]); }
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
