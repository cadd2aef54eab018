// mask_rows_fast.h -- the encoder's mask stage (vdr.py:152-169: emb *= topk_mask | lexical_mask) for V <= 32 Ki columns.
//   reference: src/ir/utils/sparse.py:8-29 (build_topk_mask), src/ir/encoder/vdr.py:152-169
// (included by sparsify.hip inside its anonymous namespace, after MaskArgs)
//
// mask_rows_kernel (sparsify.hip) reads a row into LDS, selects on the LDS copy and writes the row back, one row after the other:
// measured without any select at all that sequence moves 2.6 TB/s (a CU has nothing in flight while it works), with it 1.5.  Here a
// workgroup is persistent and the NEXT row's 16-byte loads (and its token ids) are issued before the current row's select starts;
// nothing else reads global memory, so they stay in flight through the whole select, and the row's stores drain behind the next.
//   0. a thread owns columns g * 4096 + 4 tid + 0..3 (one 16-byte load per g).  It keeps the TOP halves of their order keys in
//      registers (two to a VGPR) and parks the low halves in LDS.
//   1. two 8-bit histogram passes over the top halves find P, the top half of the k-th largest key.  A bin is split 32 ways by lane
//      (the top byte is the fp32 exponent, the same for most of a row: 64 lanes on ONE LDS address would serialise); the two passes
//      use two histograms, so no barrier is spent on clearing.
//   2. only the n keys whose top half equals P (~ V / 2^8 of a real row) need their low halves.  n <= kMrCand: they go to a list in LDS
//      as (low half, column) words and every one is ranked by counting the larger words -- exact, ties to the lowest columns, one
//      barrier.  A row of few distinct values (n > kMrCand): two more histogram passes over the low halves, then bitmap + prefix
//      popcounts for the ties, every thread sweeping its own keys.
//   3. the row is written back whole, 16 bytes a lane: unselected elements as x * 0 (a signed zero from the key's sign bit; NaN / inf
//      rebuilt from the two halves and multiplied, so the result is torch's), selected elements with the bits they had.
#pragma once

constexpr int kMrSub = 32;                 // copies of a histogram bin
constexpr int kMrCand = 2048;              // list capacity: keys sharing the top half of the k-th key
constexpr int kMrThreads = 512;            // 8 waves, up to 256 VGPRs each: the row in flight (64) + the packed keys (32) + the rest, no spill --
                                           // with 1024 threads (128 VGPRs) the compiler spills, and a scratch reload waits for the prefetch
constexpr int kMrStep = kMrThreads * 4;    // columns per g
constexpr int kMrCols = 32 * 1024;         // columns the tables cover, whatever V: no bound checks on them
constexpr int kMrWords = kMrCols / 32;     // words of a column bitmap

typedef uint32_t mr_u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr size_t mask_fast_lds_bytes() {
    return (size_t)2 * 256 * kMrSub * 4 + (size_t)kMrCols * 2 + 4 * (size_t)kMrWords * 4 + (size_t)kMrCand * 4 + (32 + 8) * 4;
}

// LDS-only barrier: __syncthreads() also waits for every global access of the wave (vmcnt(0)) -- here that would be the next row's
// prefetch and the previous row's stores, i.e. the overlap this kernel exists for.  Nothing is exchanged through global memory.
__device__ __forceinline__ void mr_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// inclusive scan across a wave in 6 DPP adds (rows of 16: row_shr 1, 2, 4, 8; then row_bcast 15 into rows 1 and 3, row_bcast 31 into rows 2, 3)
__device__ __forceinline__ int mr_wave_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);
    return v;
}
// workgroup-wide exclusive scan of one int per thread (kMrThreads threads); scratch: 8 ints in LDS, free again after the call's barrier
__device__ __forceinline__ int mr_scan(int v, int* scratch, int tid) {
    const int lane = tid & 63, w = tid >> 6;
    const int incl = mr_wave_scan(v);
    if (lane == 63) scratch[w] = incl;
    mr_barrier();
    int base = 0;
#pragma unroll
    for (int i = 0; i < kMrThreads / 64; ++i) base += i < w ? scratch[i] : 0;
    return base + incl - v;
}
template <int N>            // N int4 per thread
__device__ __forceinline__ void mr_clear(int* p, int tid) {
    int4* p4 = reinterpret_cast<int4*>(p);
#pragma unroll
    for (int i = 0; i < N; ++i) p4[tid + i * kMrThreads] = make_int4(0, 0, 0, 0);
}
constexpr int kMrHistClr = 256 * kMrSub / 4 / kMrThreads;

// after a pass's atomics: the bin (descending) that holds the `remaining`-th key; two threads per bin.  The OTHER histogram
// is cleared on the way (it serves the next pass).
__device__ __forceinline__ uint32_t mr_pick(const int* hist, int* other, int* scratch, int* sel_sh, int tid, int& remaining, int& n_eq) {
    mr_barrier();
    mr_clear<kMrHistClr>(other, tid);
    // thread t: bin 255 - t / 2, half t % 2 of its 32 copies (bank-staggered); the pair's sum lands in both lanes, counted once
    const int bin = 255 - (tid >> 1);
    const int4* hb = reinterpret_cast<const int4*>(hist + bin * kMrSub + (tid & 1) * (kMrSub / 2));
    int hh = 0;
#pragma unroll
    for (int c = 0; c < kMrSub / 8; ++c) {
        const int4 q = hb[(c + (tid >> 1)) & (kMrSub / 8 - 1)];
        hh += q.x + q.y + q.z + q.w;
    }
    const int h = hh + __builtin_amdgcn_update_dpp(0, hh, 0xB1, 0xF, 0xF, true);       // quad_perm [1, 0, 3, 2]: the neighbour's half
    const int above = mr_scan((tid & 1) ? 0 : h, scratch, tid);               // keys in strictly higher bins
    if (!(tid & 1) && above < remaining && remaining <= above + h) { sel_sh[0] = bin; sel_sh[1] = above; sel_sh[2] = h; }
    mr_barrier();
    remaining -= sel_sh[1];
    n_eq = sel_sh[2];
    return (uint32_t)sel_sh[0];
}

// (developer builds, -DMR_TIMING: s_memtime at the phase boundaries, wave 0 of every workgroup, summed behind a.flags)
#ifdef MR_TIMING
#define MR_T(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define MR_T(i) do { } while (0)
#endif

// CSR = 1: also emits the kept non-zero elements as CSR slot runs (step 4 below; vs_embed_mask_to_csr)
template <int G, int CSR = 0>
__global__ __launch_bounds__(kMrThreads) void mask_rows_fast_kernel(MaskArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* hist0 = reinterpret_cast<int*>(smem);                                 // [256 * kMrSub] x 2
    int* hist1 = hist0 + 256 * kMrSub;
    uint16_t* lo16 = reinterpret_cast<uint16_t*>(hist1 + 256 * kMrSub);        // [kMrCols] low halves of the order keys
    uint32_t* lex = reinterpret_cast<uint32_t*>(lo16 + kMrCols);               // [1024] lexical bitmap
    uint32_t* eqb = lex + kMrWords;                                            // [1024] columns whose key equals the k-th key
    uint32_t* selb = eqb + kMrWords;                                           // [1024] selected columns among those sharing its top half
    int* eqp = reinterpret_cast<int*>(selb + kMrWords);                        // [1024] eqb's bits in the words below
    uint32_t* cand = reinterpret_cast<uint32_t*>(eqp + kMrWords);              // [kMrCand] (low half << 15) | (32767 - column)
    int* scratch = reinterpret_cast<int*>(cand + kMrCand);                     // [32]
    int* sel_sh = scratch + 32;                                                // [4]
    int* cnt = sel_sh + 4;                                                     // [1]
    const int tid = threadIdx.x, sub = tid & (kMrSub - 1);
    const bool lexical = a.ids && a.activate_lexical;
    // 4 * tid, opaque to the optimiser: a phase computes its column-derived values (32 per thread) where it uses them; left to itself the
    // compiler hoists them all out of the row loop and keeps them alive (> 100 VGPRs, spilled)
    auto lane4 = [&]() -> int { int t; asm volatile("v_lshlrev_b32 %0, 2, %1" : "=v"(t) : "v"(tid)); return t; };

    int b = blockIdx.x;
    if (b >= a.B) return;
    // the row in flight: raw elements + this thread's token id (tokens beyond the first 512 of a row are read in place)
    mr_u32x4 raw[G];
    int64_t tok = -1;
    // (the loads of a row are issued one by one from inside pass A of the row before: the L1 takes ~ 16 cycles per 1 KB instruction,
    //  bunched together they -- and the stores queued before them -- held every wave for 9 k cycles a row)
    auto row_rsrc = [&](int row) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (size_t)row * a.ld), 0, a.V * 4, 0x00020000); };
    auto issue_tok = [&](int row) { if (lexical && tid < a.L) tok = a.ids[(size_t)row * a.L + tid]; };
    // buffer addressing: ONE lane offset (tid * 16) + a scalar offset per g; reads past the row's end return 0, stores there are dropped
    auto issue_g = [&](const __amdgpu_buffer_rsrc_t& rx, int g) { raw[g] = __builtin_amdgcn_raw_buffer_load_b128(rx, tid * 16, g * kMrStep * 4, 0); };
    {
        const __amdgpu_buffer_rsrc_t rx0 = row_rsrc(b);
        issue_tok(b);
#pragma unroll
        for (int g = 0; g < G; ++g) issue_g(rx0, g);
    }
#ifdef MR_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
    for (; b < a.B; b += gridDim.x) {
        mr_barrier();                                                       // (the previous row's readers of the tables are done)
        MR_T(0);
        // ---- 0. order keys: top halves -> kk (columns past V: 0, the lowest), low halves -> LDS ----------------------------------
        uint32_t kk[G][2];
        const int64_t my_tok = tok;
        const int t4p = lane4();
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int left = a.V - (g * kMrStep + t4p);
            const uint32_t f0 = flip_f32(__uint_as_float(raw[g].x)), f1 = flip_f32(__uint_as_float(raw[g].y));
            const uint32_t f2 = flip_f32(__uint_as_float(raw[g].z)), f3 = flip_f32(__uint_as_float(raw[g].w));
            const uint32_t k0 = left > 0 ? f0 >> 16 : 0u, k1 = left > 1 ? f1 >> 16 : 0u;
            const uint32_t k2 = left > 2 ? f2 >> 16 : 0u, k3 = left > 3 ? f3 >> 16 : 0u;
            kk[g][0] = k0 | (k1 << 16);
            kk[g][1] = k2 | (k3 << 16);
            *reinterpret_cast<uint2*>(lo16 + g * kMrStep + t4p) = make_uint2((f0 & 0xFFFFu) | (f1 << 16), (f2 & 0xFFFFu) | (f3 << 16));
            __builtin_amdgcn_sched_barrier(0);          // (one g at a time: bounds the live ranges, the row's prefetch needs the registers)
        }
        MR_T(1);
        const bool has_next = b + (int)gridDim.x < a.B;
        const __amdgpu_buffer_rsrc_t rxn = row_rsrc(has_next ? b + (int)gridDim.x : b);
        if (has_next) issue_tok(b + gridDim.x);                                // in flight until the next iteration
        auto k16 = [&](int g, int r) -> uint32_t { return (r & 1) ? kk[g][r >> 1] >> 16 : kk[g][r >> 1] & 0xFFFFu; };
        mr_clear<3 * kMrWords / 4 / kMrThreads + 1>(reinterpret_cast<int*>(lex), tid);       // lex, eqb, selb (and the head of eqp)
        mr_clear<kMrHistClr>(hist0, tid);
        if (tid == 0) *cnt = 0;
        mr_barrier();
        if (lexical) {
            int bad = 0;
            if (tid < a.L) {
                if (my_tok < 0 || my_tok >= a.vocab) bad = 1;
                else if (my_tok >= a.shift) atomicOr(&lex[(my_tok - a.shift) >> 5], 1u << ((my_tok - a.shift) & 31));
            }
            for (int l = tid + kMrThreads; l < a.L; l += kMrThreads) {
                const int64_t t = a.ids[(size_t)b * a.L + l];
                if (t < 0 || t >= a.vocab) bad = 1;
                else if (t >= a.shift) atomicOr(&lex[(t - a.shift) >> 5], 1u << ((t - a.shift) & 31));
            }
            if (bad) atomicOr(a.flags, 1);
        }
        MR_T(2);
        // ---- 1. the top half of the k-th largest key -------------------------------------------------------------------------
        int remaining = a.topk, n_eq = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(&hist0[(k16(g, r) >> 8) * kMrSub + sub], 1);
            if (has_next) issue_g(rxn, g);                                      // the next row: in flight until the next iteration
            __builtin_amdgcn_sched_barrier(0);
        }
        MR_T(3);
        const uint32_t dA = mr_pick(hist0, hist1, scratch, sel_sh, tid, remaining, n_eq);
        MR_T(4);
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t k = k16(g, r);
                const bool in = (k >> 8) == dA;
                if (__any(in)) {
                    if (in) atomicAdd(&hist1[(k & 255u) * kMrSub + sub], 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const uint32_t P = (dA << 8) | mr_pick(hist1, hist0, scratch, sel_sh, tid, remaining, n_eq);
        // ---- 2. among the n_eq keys whose top half is P, the `remaining` largest (ties: lowest columns) -> selb ------------------
        MR_T(5);
        // (P = 0 -- the k-th key is a negative NaN of the highest payload -- is the one top half the columns past V share: not candidates)
        if (P == 0) n_eq -= G * kMrStep - a.V;
        const bool take_all = n_eq == remaining;
        if (!take_all && n_eq <= kMrCand) {
            const int t4c = lane4();
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k16(g, r) == P && (P != 0 || g * kMrStep + t4c + r < a.V)) {
                        const uint32_t col = (uint32_t)(g * kMrStep + t4c + r);
                        cand[atomicAdd(cnt, 1)] = ((uint32_t)lo16[col] << 15) | (32767u - col);
                    }
            mr_barrier();
            // rank by counting: the words are distinct, a larger word = a larger key or the same key in a lower column
            for (int c = tid; c < n_eq; c += kMrThreads) {
                const uint32_t mine = cand[c];
                int larger = 0;
                int j = 0;
                for (; j + 4 <= n_eq; j += 4) {
                    const uint4 q = *reinterpret_cast<const uint4*>(cand + j);
                    larger += (q.x > mine) + (q.y > mine) + (q.z > mine) + (q.w > mine);
                }
                for (; j < n_eq; ++j) larger += cand[j] > mine;
                if (larger < remaining) {
                    const uint32_t col = 32767u - (mine & 32767u);
                    atomicOr(&selb[col >> 5], 1u << (col & 31));
                }
            }
            mr_barrier();
        } else if (!take_all) {
            // f(column, low half) for every key of this thread whose top half is P
            auto each_cand = [&](auto&& f) {
                const int t4c = lane4();
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (k16(g, r) == P && (P != 0 || g * kMrStep + t4c + r < a.V)) {
                            const uint32_t col = (uint32_t)(g * kMrStep + t4c + r);
                            f(col, (uint32_t)lo16[col]);
                        }
            };
            each_cand([&](uint32_t, uint32_t lo) { atomicAdd(&hist0[(lo >> 8) * kMrSub + sub], 1); });
            const uint32_t dC = mr_pick(hist0, hist1, scratch, sel_sh, tid, remaining, n_eq);
            each_cand([&](uint32_t, uint32_t lo) { if ((lo >> 8) == dC) atomicAdd(&hist1[(lo & 255u) * kMrSub + sub], 1); });
            const uint32_t T = (dC << 8) | mr_pick(hist1, hist0, scratch, sel_sh, tid, remaining, n_eq);       // low half of the k-th key
            const int r_eq = remaining;                                         // >= 1 keys equal to it are selected
            const bool all_eq = n_eq == r_eq;
            if (!all_eq) {          // rank the equal keys by column
                each_cand([&](uint32_t col, uint32_t lo) { if (lo == T) atomicOr(&eqb[col >> 5], 1u << (col & 31)); });
                mr_barrier();
                const int p0 = __popc(eqb[2 * tid]), p1 = __popc(eqb[2 * tid + 1]);       // (two bitmap words per thread)
                const int below = mr_scan(p0 + p1, scratch, tid);
                eqp[2 * tid] = below;
                eqp[2 * tid + 1] = below + p0;
                mr_barrier();
            }
            each_cand([&](uint32_t col, uint32_t lo) {
                bool sel = lo > T;
                if (lo == T) sel = all_eq || eqp[col >> 5] + __popc(eqb[col >> 5] & ((1u << (col & 31)) - 1u)) < r_eq;
                if (sel) atomicOr(&selb[col >> 5], 1u << (col & 31));
            });
            mr_barrier();
        }
        MR_T(6);
        // ---- 3. write: unselected elements <- x * 0 ---------------------------------------------------------------------------
        const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(a.emb ? a.emb + (size_t)b * a.ld : nullptr, 0, a.emb ? a.V * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(a.mask ? a.mask + (size_t)b * a.V : nullptr, 0, a.mask ? a.V : 0, 0x00020000);
        const int t4o = lane4();
        // branch-free: every element is rebuilt from its two key halves, multiplied by zero (torch's x * 0: signed zeros, NaN for NaN
        // and inf) and the product or the element itself is picked by the select bit -- a version with the obvious early-outs compiled
        // to ~ 40 scalar branches per 4 elements and took half the kernel's time.  The row goes back whole, 16 bytes a lane.
        const uint32_t lex_on = a.activate_lexical ? 0xFu : 0u, sel_all = take_all ? 0xFu : 0u;
        [[maybe_unused]] uint32_t nzw[2] = {0u, 0u};                            // 4 bits per g: this thread's selected elements (CSR emission below)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int i = g * kMrStep + t4o;                                    // (4 columns of one bitmap word)
            const uint32_t lw = (lex[i >> 5] >> (i & 31)) & lex_on;
            const uint32_t sw = (selb[i >> 5] >> (i & 31)) | sel_all;
            const uint2 lo = *reinterpret_cast<const uint2*>(lo16 + i);
            const uint32_t w0 = kk[g][0], w1 = kk[g][1];
            const uint32_t key[4] = {(w0 << 16) | (lo.x & 0xFFFFu), (w0 & 0xFFFF0000u) | (lo.x >> 16), (w1 << 16) | (lo.y & 0xFFFFu), (w1 & 0xFFFF0000u) | (lo.y >> 16)};
            uint32_t o[4], selm = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t k = key[r] >> 16;
                const uint32_t sel = ((uint32_t)(k > P) | ((uint32_t)(k == P) & (sw >> r)) | (lw >> r)) & 1u;
                const uint32_t xb = key[r] ^ (~(uint32_t)((int32_t)key[r] >> 31) | 0x80000000u);       // unflip_f32
                const uint32_t zb = __float_as_uint(__uint_as_float(xb) * 0.f);
                o[r] = sel ? xb : zb;
                selm |= sel << r;
            }
            if constexpr (CSR != 0) nzw[g >> 3] |= selm << (4 * (g & 7));
            if (a.mask) {
#pragma unroll
                for (int r = 0; r < 4; ++r) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)((selm >> r) & 1u), rm, t4o + r, g * kMrStep, 0);
            }
            const mr_u32x4 ov = {o[0], o[1], o[2], o[3]};
            if (a.emb) __builtin_amdgcn_raw_buffer_store_b128(ov, re, t4o * 4, g * kMrStep * 4, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        MR_T(7);
        // ---- 4. CSR emission ("write CSR rows directly"): the kept non-zero elements as (column, value) pairs in column
        // order into the row's slot run -- what Tensor.to_sparse_csr() of the masked row holds (retriever.py:304), without a second and
        // third pass over the dense row.  Rank of a kept column = kept columns below it: a bitmap of the kept columns (the tie bitmaps
        // of step 2 are free again) + prefix popcounts of its words, as step 2 ranks ties.
        if constexpr (CSR != 0) {
            static_assert(G <= 16, "4 bits per g in two words");
            uint32_t* nzb = eqb;                                                // [1024] kept non-zero columns
            int* nzp = eqp;                                                     // [1024] ... in the words below
            mr_barrier();                                                       // (step 3's readers of selb / lex are done; eqb / eqp idle since step 2)
            mr_clear<kMrWords / 4 / kMrThreads + 1>(reinterpret_cast<int*>(nzb), tid);           // (+ the head of selb: not read again this row)
            mr_barrier();
            // (what to_sparse_csr() keeps: selected, inside the row, and not +-0 -- tested here, on the few selected elements, not in the
            //  write loop: there the test cost 50 VGPRs and with them the spill-free row prefetch)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t s4 = (nzw[g >> 3] >> (4 * (g & 7))) & 15u;
                if (s4) {
                    const int i = g * kMrStep + lane4();                        // (a multiple of 4: the four bits fall into one word)
                    const uint2 lo = *reinterpret_cast<const uint2*>(lo16 + i);
                    // (opaque copies: else the compiler keeps the write loop's 64 key halves alive for this loop instead of rebuilding the
                    //  few it needs -- 80 spilled VGPRs, and a scratch reload waits for the next row's prefetch)
                    uint32_t w0 = kk[g][0], w1 = kk[g][1];
                    asm volatile("" : "+v"(w0), "+v"(w1));
                    const uint32_t key[4] = {(w0 << 16) | (lo.x & 0xFFFFu), (w0 & 0xFFFF0000u) | (lo.x >> 16), (w1 << 16) | (lo.y & 0xFFFFu), (w1 & 0xFFFF0000u) | (lo.y >> 16)};
                    uint32_t n4 = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool zero = key[r] == 0x80000000u || key[r] == 0x7FFFFFFFu;       // flip_f32(+0), flip_f32(-0)
                        n4 |= (uint32_t)(!zero && i + r < a.V) << r;
                    }
                    n4 &= s4;
                    if (n4) atomicOr(&nzb[i >> 5], n4 << (i & 31));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            mr_barrier();
            {
                const int p0 = __popc(nzb[2 * tid]), p1 = __popc(nzb[2 * tid + 1]);       // (two bitmap words per thread)
                const int below = mr_scan(p0 + p1, scratch, tid);
                nzp[2 * tid] = below;
                nzp[2 * tid + 1] = below + p0;
                if (tid == kMrThreads - 1) a.row_nnz[b] = (int64_t)(below + p0 + p1);
            }
            mr_barrier();
            int32_t* oc = a.slot_cols + (size_t)b * a.slot_cap;
            float* ov = a.slot_vals + (size_t)b * a.slot_cap;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                __builtin_amdgcn_sched_barrier(0);          // (one g at a time: the next row's prefetch needs the registers)
                if ((nzw[g >> 3] >> (4 * (g & 7))) & 15u) {
                    const int i = g * kMrStep + lane4();
                    const uint32_t word = nzb[i >> 5];
                    const uint32_t n4 = (word >> (i & 31)) & 15u;                // this thread's kept non-zero elements of the group
                    uint32_t pos = (uint32_t)nzp[i >> 5] + (uint32_t)__popc(word & ((1u << (i & 31)) - 1u));
                    const uint2 lo = *reinterpret_cast<const uint2*>(lo16 + i);
                    uint32_t w0 = kk[g][0], w1 = kk[g][1];
                    asm volatile("" : "+v"(w0), "+v"(w1));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if ((n4 >> r) & 1u) {
                            const uint32_t key = r == 0 ? (w0 << 16) | (lo.x & 0xFFFFu) : r == 1 ? (w0 & 0xFFFF0000u) | (lo.x >> 16) : r == 2 ? (w1 << 16) | (lo.y & 0xFFFFu) : (w1 & 0xFFFF0000u) | (lo.y >> 16);
                            if (pos < (uint32_t)a.slot_cap) {
                                oc[pos] = i + r;
                                ov[pos] = __uint_as_float(key ^ (~(uint32_t)((int32_t)key >> 31) | 0x80000000u));
                            } else {
                                atomicOr(a.flags, 2);                           // more kept elements than the slot run holds (the host sized it from topk + L)
                            }
                            ++pos;
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#ifdef MR_TIMING
    if (tid == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(a.flags) + 1 + i, tacc[i]);
#endif
}
