// mask_rows_fast.h -- the encoder's mask stage (vdr.py:152-169: emb *= topk_mask | lexical_mask) for V <= 32 Ki columns.
//   reference: src/ir/utils/sparse.py:8-29 (build_topk_mask), src/ir/encoder/vdr.py:152-169
// (included by sparsify.hip inside its anonymous namespace, after MaskArgs)
//
// mask_rows_kernel (sparsify.hip) reads a row into LDS, selects on the LDS copy and writes the row back, one row after the other:
// measured without any select at all that sequence moves 2.6 TB/s (a CU has nothing in flight while it works), with it 1.5.  Here a
// workgroup is persistent and the NEXT row's 16-byte loads (and its token ids) are issued before the current row's select starts;
// nothing else reads global memory, so they stay in flight through the whole select, and the row's stores drain behind the next.
//   0. a WAVE owns G * 256 consecutive columns, a lane columns wave * G * 256 + g * 256 + 4 lane + 0..3 (one 16-byte load per g; column
//      order = (wave, g, lane): the CSR emission below ranks a wave's kept elements with wave scans, no bitmaps).  A thread keeps the
//      TOP halves of its order keys in registers (two to a VGPR) and parks the low halves in LDS.
//   1. two 8-bit histogram passes over the top halves find P, the top half of the k-th largest key.  A bin is split 32 ways by lane
//      (the top byte is the fp32 exponent, the same for most of a row: 64 lanes on ONE LDS address would serialise); the two passes
//      use two histograms, so no barrier is spent on clearing.
//   2. every thread reduces its 4 G top halves to two 64-bit masks with packed 16-bit arithmetic (v_pk_sub_u16 clamp, v_pk_min_u16):
//      keys above P (selected) and keys equal to P (candidates: ~ V / 2^8 of a real row).  Everything after that walks the SET BITS of a
//      mask in a rolled loop -- column and low half from the bit's index, nothing from the register arrays: an earlier version tested
//      its 4 G keys one by one in every phase, 57 KB of unrolled branches whose taken blocks the compiler moved out of line: the
//      instruction cache (64 KB for two CUs) missed on each, 13 k cycles a row in the candidate phase alone (MR_TIMING clocks, round 6).
//      n <= kMrCand candidates: they go to a list in LDS as (low half, column) words and every one is ranked by counting the larger
//      words -- exact, ties to the lowest columns, one barrier; a selected candidate's bit goes to its OWNER's mask words in LDS (where
//      the lexical tokens' bits are too).  A row of few distinct values (n > kMrCand): two more histogram passes over the candidates'
//      low halves, then bitmap + prefix popcounts for the ties, every thread sweeping its own candidates.
//   3. kept = above | selected candidates | lexical.  Mask stage: the row is written back whole, 16 bytes a lane: unselected elements
//      as x * 0 (a signed zero from the key's sign bit; NaN / inf rebuilt from the two halves and multiplied, so the result is torch's),
//      selected elements with the bits they had.  Mask -> CSR: see 3' in the kernel.
#pragma once

constexpr int kMrSub = 32;                 // copies of a histogram bin
constexpr int kMrCand = 2048;              // list capacity: keys sharing the top half of the k-th key
constexpr int kMrThreads = 512;            // 8 waves, up to 256 VGPRs each: the row in flight (64) + the packed keys (32) + the rest, no spill --
                                           // with 1024 threads (128 VGPRs) the compiler spills, and a scratch reload waits for the prefetch
constexpr int kMrStep = kMrThreads * 4;    // columns per g
constexpr int kMrCols = 32 * 1024;         // columns the tables cover, whatever V: no bound checks on them
constexpr int kMrWords = kMrCols / 32;     // words of a column bitmap
constexpr int kMrStage = 2 * 256 * kMrSub / 2;       // (column, value) pairs the two histograms hold between a row's select and the next row's: the CSR staging area
                                                     // (vs_embed_mask_to_csr serves topk + L <= kMrStage)
constexpr unsigned long long kMrDone = 1ull << 62;   // a workgroup's total is published (mask -> CSR: MaskArgs::wg_tot)

typedef uint32_t mr_u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr size_t mask_fast_lds_bytes() {
    return (size_t)2 * 256 * kMrSub * 4 + (size_t)kMrCols * 2 + (size_t)kMrThreads * 8 + 2 * (size_t)kMrWords * 4 + (size_t)(kMrCand + 64) * 4 + (32 + 8) * 4;
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

// CSR = 1: emits the kept non-zero elements as CSR rows instead of writing the masked row back (step 3' below; vs_embed_mask_to_csr)
template <int G, int CSR = 0>
__global__ __launch_bounds__(kMrThreads) void mask_rows_fast_kernel(MaskArgs a) {
    static_assert(G <= 16, "4 bits per g in two mask words");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* hist0 = reinterpret_cast<int*>(smem);                                 // [256 * kMrSub] x 2
    int* hist1 = hist0 + 256 * kMrSub;
    uint16_t* lo16 = reinterpret_cast<uint16_t*>(hist1 + 256 * kMrSub);        // [kMrCols] low halves of the order keys
    uint32_t* lm = reinterpret_cast<uint32_t*>(lo16 + kMrCols);                // [kMrThreads][2] a thread's mask words: lexical columns, selected candidates
    uint32_t* eqb = lm + 2 * kMrThreads;                                       // [1024] columns whose key equals the k-th key (rows of few distinct values)
    int* eqp = reinterpret_cast<int*>(eqb + kMrWords);                         // [1024] eqb's bits in the words below
    uint32_t* cand = reinterpret_cast<uint32_t*>(eqp + kMrWords);              // [kMrCand + 64] (low half << 15) | (32767 - column); mask -> CSR: the lanes' dummy pairs
    int* scratch = reinterpret_cast<int*>(cand + kMrCand + 64);                // [32]
    int* sel_sh = scratch + 32;                                                // [4]
    int* cnt = sel_sh + 4;                                                     // [1]
    const int tid = threadIdx.x, sub = tid & (kMrSub - 1), lane = tid & 63;
    const int wbase = __builtin_amdgcn_readfirstlane(tid >> 6) * (G * 256);     // the wave's first column (scalar)
    const bool lexical = a.ids && a.activate_lexical;
    // 4 * lane, opaque to the optimiser: a phase computes its column-derived values where it uses them; left to itself the compiler hoists
    // them all out of the row loop and keeps them alive (> 100 VGPRs, spilled)
    auto lane4 = [&]() -> int { int t; asm volatile("v_lshlrev_b32 %0, 2, %1" : "=v"(t) : "v"(lane)); return t; };
    // bit e = 4 g + r of a thread's masks <-> column wbase + 256 g + 4 lane + r
    auto col_of_bit = [&](int e, int l4) -> uint32_t { return (uint32_t)(wbase + (e >> 2) * 256 + l4 + (e & 3)); };
    // ... and back: the mask word (of lm) and bit that stand for a column
    auto owner_or = [&](uint32_t col) {
        const uint32_t w = col / (uint32_t)(G * 256), rem = col - w * (uint32_t)(G * 256);
        const uint32_t bit = (rem >> 8) * 4u + (rem & 3u), own = w * 64u + ((rem >> 2) & 63u);
        atomicOr(&lm[own * 2u + (bit >> 5)], 1u << (bit & 31u));
    };

    // rows of this workgroup: b_first, b_first + b_step, ... < b_end.  Mask only: rows blockIdx.x + i * gridDim.x.  Mask -> CSR: a run of
    // consecutive rows per workgroup, workgroups numbered by a ticket (in the order they START: workgroup v waits at the end for the
    // totals of the workgroups < v, which are running by then whatever the dispatcher does)
    int b_first = blockIdx.x, b_end = a.B, b_step = gridDim.x;
    [[maybe_unused]] int vwg = blockIdx.x;
    if constexpr (CSR != 0) {
        if (tid == 0) sel_sh[3] = (int)atomicAdd(reinterpret_cast<unsigned int*>(a.flags) + 1, 1u);
        __syncthreads();
        vwg = sel_sh[3];
        const int per = a.B / (int)gridDim.x, rem = a.B % (int)gridDim.x;
        b_first = vwg * per + min(vwg, rem);
        b_end = b_first + per + (vwg < rem ? 1 : 0);
        b_step = 1;
    }
    [[maybe_unused]] int run_rows = 0;                                         // mask -> CSR: kept elements of this workgroup's rows so far
    int b = b_first;
    if constexpr (CSR == 0) { if (b >= b_end) return; }
    // the row in flight: raw elements + this thread's token id (tokens beyond the first 512 of a row are read in place)
    mr_u32x4 raw[G];
    int64_t tok = -1;
    // (the loads of a row are issued one by one from inside pass A of the row before: the L1 takes ~ 16 cycles per 1 KB instruction,
    //  bunched together they -- and the stores queued before them -- held every wave for 9 k cycles a row)
    auto row_rsrc = [&](int row) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (size_t)row * a.ld), 0, a.V * 4, 0x00020000); };
    auto issue_tok = [&](int row) { if (lexical && tid < a.L) tok = a.ids[(size_t)row * a.L + tid]; };
    // buffer addressing: ONE lane offset (lane * 16) + a scalar offset per (wave, g); reads past the row's end return 0, stores there are dropped
    auto issue_g = [&](const __amdgpu_buffer_rsrc_t& rx, int g) { raw[g] = __builtin_amdgcn_raw_buffer_load_b128(rx, lane * 16, (wbase + g * 256) * 4, 0); };
    if (b < b_end) {
        const __amdgpu_buffer_rsrc_t rx0 = row_rsrc(b);
        issue_tok(b);
#pragma unroll
        for (int g = 0; g < G; ++g) issue_g(rx0, g);
    }
    // this thread's columns inside the row (the same for every row)
    uint32_t inr[2] = {0u, 0u};
    {
        const int t4i = lane4();
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int left = min(max(a.V - (wbase + g * 256 + t4i), 0), 4);
            inr[g >> 3] |= ((1u << left) - 1u) << (4 * (g & 7));
        }
    }
#ifdef MR_TIMING
    unsigned long long tacc[17] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tlast;
    const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();
#endif
    for (; b < b_end; b += b_step) {
        mr_barrier();                                                       // (the previous row's readers of the tables are done)
        MR_T(0);
        // ---- 0. order keys: top halves -> kk (columns past V: 0, the lowest), low halves -> LDS ----------------------------------
        uint32_t kk[G][2];
        [[maybe_unused]] uint32_t zm[2] = {0u, 0u};                           // mask -> CSR: this thread's +-0 elements (to_sparse_csr() drops them)
        [[maybe_unused]] float zmin = 1.f;                                     // ... the smallest |x| of its columns: 0 = it has one (two min3 per g; the bits only then)
        const int64_t my_tok = tok;
        const int t4p = lane4();
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int left = a.V - (wbase + g * 256 + t4p);
            if constexpr (CSR != 0) {
                const float x0 = fabsf(__uint_as_float(raw[g].x)), x1 = fabsf(__uint_as_float(raw[g].y)), x2 = fabsf(__uint_as_float(raw[g].z)), x3 = fabsf(__uint_as_float(raw[g].w));
                const int gleft = a.V - (wbase + g * 256);                      // (scalar: columns of the row from this g's first on)
                if (gleft >= 256) zmin = __builtin_fminf(__builtin_fminf(zmin, __builtin_fminf(x0, x1)), __builtin_fminf(x2, x3));
                else if (gleft > 0) zmin = __builtin_fminf(__builtin_fminf(zmin, __builtin_fminf(left > 0 ? x0 : 1.f, left > 1 ? x1 : 1.f)), __builtin_fminf(left > 2 ? x2 : 1.f, left > 3 ? x3 : 1.f));
            }
            const uint32_t f0 = flip_f32(__uint_as_float(raw[g].x)), f1 = flip_f32(__uint_as_float(raw[g].y));
            const uint32_t f2 = flip_f32(__uint_as_float(raw[g].z)), f3 = flip_f32(__uint_as_float(raw[g].w));
            if (a.V - (wbase + g * 256) >= 256) {                               // (scalar) every lane's four columns are inside the row: two v_perm_b32 per pair
                kk[g][0] = __builtin_amdgcn_perm(f1, f0, 0x07060302u);
                kk[g][1] = __builtin_amdgcn_perm(f3, f2, 0x07060302u);
            } else {
                const uint32_t k0 = left > 0 ? f0 >> 16 : 0u, k1 = left > 1 ? f1 >> 16 : 0u;
                const uint32_t k2 = left > 2 ? f2 >> 16 : 0u, k3 = left > 3 ? f3 >> 16 : 0u;
                kk[g][0] = k0 | (k1 << 16);
                kk[g][1] = k2 | (k3 << 16);
            }
            *reinterpret_cast<uint2*>(lo16 + wbase + g * 256 + t4p) = make_uint2(__builtin_amdgcn_perm(f1, f0, 0x05040100u), __builtin_amdgcn_perm(f3, f2, 0x05040100u));
            __builtin_amdgcn_sched_barrier(0);          // (one g at a time: bounds the live ranges, the row's prefetch needs the registers)
        }
        if constexpr (CSR != 0) {
            if (__builtin_amdgcn_ballot_w64(zmin == 0.f) != 0) {                // (rows of real activations: never)
                const int t4z = lane4();
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int i = wbase + g * 256 + t4z;
                    const uint2 lo = *reinterpret_cast<const uint2*>(lo16 + i);
                    const uint32_t w0 = kk[g][0], w1 = kk[g][1];
                    const uint32_t key[4] = {(w0 << 16) | (lo.x & 0xFFFFu), (w0 & 0xFFFF0000u) | (lo.x >> 16), (w1 << 16) | (lo.y & 0xFFFFu), (w1 & 0xFFFF0000u) | (lo.y >> 16)};
                    uint32_t z4 = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) z4 |= (uint32_t)(key[r] == 0x80000000u || key[r] == 0x7FFFFFFFu) << r;       // flip_f32(+0), flip_f32(-0)
                    zm[g >> 3] |= z4 << (4 * (g & 7));
                }
            }
        }
        MR_T(1);
        const bool has_next = b + b_step < b_end;
        const __amdgpu_buffer_rsrc_t rxn = row_rsrc(has_next ? b + b_step : b);
        if (has_next) issue_tok(b + b_step);                                   // in flight until the next iteration
        auto k16 = [&](int g, int r) -> uint32_t { return (r & 1) ? kk[g][r >> 1] >> 16 : kk[g][r >> 1] & 0xFFFFu; };
        *reinterpret_cast<uint2*>(lm + 2 * tid) = make_uint2(0u, 0u);
        mr_clear<kMrHistClr>(hist0, tid);
        if (tid == 0) *cnt = 0;
        mr_barrier();
        if (lexical) {
            int bad = 0;
            if (tid < a.L) {
                if (my_tok < 0 || my_tok >= a.vocab) bad = 1;
                else if (my_tok >= a.shift) owner_or((uint32_t)(my_tok - a.shift));
            }
            for (int l = tid + kMrThreads; l < a.L; l += kMrThreads) {
                const int64_t t = a.ids[(size_t)b * a.L + l];
                if (t < 0 || t >= a.vocab) bad = 1;
                else if (t >= a.shift) owner_or((uint32_t)(t - a.shift));
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
        const uint32_t dA2 = (dA << 8) | (dA << 24);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            // (a key's top byte is dA <=> its half of  w ^ (dA in both top bytes)  is below 256 -- and is the key's second byte, the bin)
            const uint32_t x01 = kk[g][0] ^ dA2, x23 = kk[g][1] ^ dA2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t h = (r & 1) ? ((r & 2) ? x23 : x01) >> 16 : ((r & 2) ? x23 : x01) & 0xFFFFu;
                const bool in = h < 256u;
                if (__any(in)) {
                    if (in) atomicAdd(&hist1[h * kMrSub + sub], 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const uint32_t P = (dA << 8) | mr_pick(hist1, hist0, scratch, sel_sh, tid, remaining, n_eq);
        MR_T(5);
        // ---- 2. masks: keys above P, keys whose top half is P (4 bits per g, packed 16-bit arithmetic: two keys an instruction) ----
        uint32_t gtm[2] = {0u, 0u}, nem[2] = {0u, 0u};
        {
            const uint32_t pp = P | (P << 16), one2 = 0x00010001u;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                uint32_t d0, d1, x0, x1;
                asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d0) : "v"(kk[g][0]), "v"(pp));       // > 0 where the key is above P
                asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d1) : "v"(kk[g][1]), "v"(pp));
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(d0) : "v"(d0), "v"(one2));
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(d1) : "v"(d1), "v"(one2));
                x0 = kk[g][0] ^ pp;                                                              // != 0 where it differs from P
                x1 = kk[g][1] ^ pp;
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(x0) : "v"(x0), "v"(one2));
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(x1) : "v"(x1), "v"(one2));
                // bits 0 / 16 of the four words -> two nibbles: above at 0 .. 3, different at 4 .. 7 (keys 0 .. 3 of the g in this order)
                uint32_t y = d0 | (d1 << 2) | (x0 << 4) | (x1 << 6);
                y |= y >> 15;
                gtm[g >> 3] |= (y & 15u) << (4 * (g & 7));
                nem[g >> 3] |= ((y >> 4) & 15u) << (4 * (g & 7));
            }
        }
        // (P = 0 -- the k-th key is a negative NaN of the highest payload -- is the one top half the columns past V share: not candidates)
        uint32_t eqm[2] = {~nem[0] & inr[0], ~nem[1] & inr[1]};
        if (P == 0) n_eq -= G * kMrStep - a.V;
        MR_T(6);
        // ---- among the n_eq keys whose top half is P, the `remaining` largest (ties: lowest columns) ------------------------------
        // f(bit, column) for every candidate of this thread
        auto each_cand = [&](auto&& f) {
            const int t4c = lane4();
            uint32_t m0 = eqm[0], m1 = eqm[1];
            while (m0 | m1) {
                const int e = m0 ? __ffs(m0) - 1 : 31 + __ffs(m1);
                if (m0) m0 &= m0 - 1u; else m1 &= m1 - 1u;
                f(e, col_of_bit(e, t4c));
            }
        };
        const bool take_all = n_eq == remaining;
        uint32_t own[2] = {0u, 0u};                                            // candidates this thread selected itself (rows of few distinct values)
        if (!take_all && n_eq <= kMrCand) {
            // the list: one atomic per wave (its candidates, counted from the masks), a lane's run from a wave scan
            {
                const int mine = __popc(eqm[0]) + __popc(eqm[1]);
                const int incl = mr_wave_scan(mine);
                int base = 0;
                if (lane == 63) base = atomicAdd(cnt, incl);
                int at = __builtin_amdgcn_readlane(base, 63) + incl - mine;
                each_cand([&](int, uint32_t col) { cand[at++] = ((uint32_t)lo16[col] << 15) | (32767u - col); });
            }
            const int n_pad = (n_eq + 63) & ~63;                                // zero words (no candidate's: larger than none) up to a multiple of 64
            if (tid < n_pad - n_eq) cand[n_eq + tid] = 0u;
            MR_T(11);
            mr_barrier();
            MR_T(12);
            // rank by counting: the words are distinct, a larger word = a larger key or the same key in a lower column.  2^s threads (one
            // quad's) share a candidate, a slice of the list each, 16 words in flight
            const int sh = n_eq <= kMrThreads / 4 ? 2 : n_eq <= kMrThreads / 2 ? 1 : 0;
            const int slice = n_pad >> sh, part = tid & ((1 << sh) - 1);
            for (int c0 = 0; c0 < n_eq; c0 += kMrThreads >> sh) {
                const int c = c0 + (tid >> sh);
                const uint32_t mine = c < n_eq ? cand[c] : 0xFFFFFFFFu;
                int larger = 0;
                const uint4* lst = reinterpret_cast<const uint4*>(cand + part * slice);
                for (int j = 0; j < slice / 4; j += 4) {
                    const uint4 q0 = lst[j], q1 = lst[j + 1], q2 = lst[j + 2], q3 = lst[j + 3];
                    larger += (q0.x > mine) + (q0.y > mine) + (q0.z > mine) + (q0.w > mine) + (q1.x > mine) + (q1.y > mine) + (q1.z > mine) + (q1.w > mine);
                    larger += (q2.x > mine) + (q2.y > mine) + (q2.z > mine) + (q2.w > mine) + (q3.x > mine) + (q3.y > mine) + (q3.z > mine) + (q3.w > mine);
                }
                if (sh >= 1) larger += __builtin_amdgcn_update_dpp(0, larger, 0xB1, 0xF, 0xF, true);       // quad_perm [1, 0, 3, 2]
                if (sh == 2) larger += __builtin_amdgcn_update_dpp(0, larger, 0x4E, 0xF, 0xF, true);       // quad_perm [2, 3, 0, 1]
                if (part == 0 && c < n_eq && larger < remaining) owner_or(32767u - (mine & 32767u));
            }
            MR_T(13);
            mr_barrier();
            MR_T(14);
        } else if (!take_all) {
            auto add_own = [&](int e) { if (e < 32) own[0] |= 1u << e; else own[1] |= 1u << (e - 32); };
            each_cand([&](int, uint32_t col) { atomicAdd(&hist0[((uint32_t)lo16[col] >> 8) * kMrSub + sub], 1); });
            const uint32_t dC = mr_pick(hist0, hist1, scratch, sel_sh, tid, remaining, n_eq);
            each_cand([&](int, uint32_t col) { const uint32_t lo = lo16[col]; if ((lo >> 8) == dC) atomicAdd(&hist1[(lo & 255u) * kMrSub + sub], 1); });
            const uint32_t T = (dC << 8) | mr_pick(hist1, hist0, scratch, sel_sh, tid, remaining, n_eq);       // low half of the k-th key
            const int r_eq = remaining;                                         // >= 1 keys equal to it are selected
            const bool all_eq = n_eq == r_eq;
            if (!all_eq) {          // rank the equal keys by column
                eqb[2 * tid] = 0u;
                eqb[2 * tid + 1] = 0u;
                mr_barrier();
                each_cand([&](int, uint32_t col) { if ((uint32_t)lo16[col] == T) atomicOr(&eqb[col >> 5], 1u << (col & 31)); });
                mr_barrier();
                const int p0 = __popc(eqb[2 * tid]), p1 = __popc(eqb[2 * tid + 1]);       // (two bitmap words per thread)
                const int below = mr_scan(p0 + p1, scratch, tid);
                eqp[2 * tid] = below;
                eqp[2 * tid + 1] = below + p0;
                mr_barrier();
            }
            each_cand([&](int e, uint32_t col) {
                const uint32_t lo = lo16[col];
                bool sel = lo > T;
                if (lo == T) sel = all_eq || eqp[col >> 5] + __popc(eqb[col >> 5] & ((1u << (col & 31)) - 1u)) < r_eq;
                if (sel) add_own(e);
            });
        }
        // ---- 3. kept = above P | selected candidates | lexical columns ---------------------------------------------------------------
        uint32_t kept[2];
        {
            const uint2 mine = *reinterpret_cast<const uint2*>(lm + 2 * tid);
            const uint32_t all_on = take_all ? ~0u : 0u;
            kept[0] = (gtm[0] | mine.x | own[0] | (eqm[0] & all_on)) & inr[0];
            kept[1] = (gtm[1] | mine.y | own[1] | (eqm[1] & all_on)) & inr[1];
        }
        MR_T(7);
        if constexpr (CSR == 0) {
            // write: unselected elements <- x * 0.  Branch-free: every element is rebuilt from its two key halves, multiplied by zero (torch's
            // x * 0: signed zeros, NaN for NaN and inf) and the product or the element itself is picked by its kept bit -- a version with
            // the obvious early-outs compiled to ~ 40 scalar branches per 4 elements and took half the kernel's time.  The row goes
            // back whole, 16 bytes a lane.
            const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(a.emb ? a.emb + (size_t)b * a.ld : nullptr, 0, a.emb ? a.V * 4 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(a.mask ? a.mask + (size_t)b * a.V : nullptr, 0, a.mask ? a.V : 0, 0x00020000);
            const int t4o = lane4();
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int i = wbase + g * 256 + t4o;
                const uint32_t selm = (kept[g >> 3] >> (4 * (g & 7))) & 15u;
                const uint2 lo = *reinterpret_cast<const uint2*>(lo16 + i);
                const uint32_t w0 = kk[g][0], w1 = kk[g][1];
                const uint32_t key[4] = {(w0 << 16) | (lo.x & 0xFFFFu), (w0 & 0xFFFF0000u) | (lo.x >> 16), (w1 << 16) | (lo.y & 0xFFFFu), (w1 & 0xFFFF0000u) | (lo.y >> 16)};
                uint32_t o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t xb = key[r] ^ (~(uint32_t)((int32_t)key[r] >> 31) | 0x80000000u);       // unflip_f32
                    const uint32_t zb = __float_as_uint(__uint_as_float(xb) * 0.f);
                    o[r] = ((selm >> r) & 1u) ? xb : zb;
                }
                if (a.mask) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)((selm >> r) & 1u), rm, t4o + r, wbase + g * 256, 0);
                }
                const mr_u32x4 ov = {o[0], o[1], o[2], o[3]};
                if (a.emb) __builtin_amdgcn_raw_buffer_store_b128(ov, re, t4o * 4, (wbase + g * 256) * 4, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            MR_T(8);
        } else {
            // ---- 3'. mask -> CSR ("write CSR rows directly"): the kept non-zero elements as (column, value) pairs in column
            // order -- what Tensor.to_sparse_csr() of the masked row holds (retriever.py:304); the dense masked row is never written.
            //   rank of a kept element = kept elements in the waves below (8 totals through LDS) + in this wave's lower g (a scalar
            //   running sum) + in the lower lanes of its g (one wave scan per g) + in its own lower bits.  The pairs go to a staging area
            //   in LDS (the two histograms are idle until the next row's passes; a lane without a kept element in the g writes a dummy
            //   pair of its own: no branch) and from there to the workgroup's run in the scratch arrays, coalesced.
            kept[0] &= ~zm[0];
            kept[1] &= ~zm[1];
            const int mine = __popc(kept[0]) + __popc(kept[1]);
            const int wave_incl = mr_wave_scan(mine);
            if (lane == 63) scratch[8 + (tid >> 6)] = wave_incl;
            mr_barrier();
            int below = 0, row_tot = 0;                                         // (the same for the wave)
#pragma unroll
            for (int w = 0; w < kMrThreads / 64; ++w) {
                const int t = scratch[8 + w];
                below += w < (tid >> 6) ? t : 0;
                row_tot += t;
            }
            int run = __builtin_amdgcn_readfirstlane(below);
            row_tot = __builtin_amdgcn_readfirstlane(row_tot);
            MR_T(8);
            uint2* stage = reinterpret_cast<uint2*>(hist0);                      // [kMrStage] (column, value bits)
            const uint32_t dummy = (uint32_t)((reinterpret_cast<uint2*>(cand) + tid) - stage);       // this lane's dummy pair, as an index of `stage`
            if (row_tot > a.slot_cap) { if (tid == 0) atomicOr(a.flags, 2); }    // more kept elements than the slot run holds (the host sized it from topk + L)
            const int t4e = lane4();
#pragma unroll
            for (int g = 0; g < G; ++g) {
                __builtin_amdgcn_sched_barrier(0);          // (one g at a time: the next row's prefetch needs the registers)
                uint32_t n4 = (kept[g >> 3] >> (4 * (g & 7))) & 15u;
                const int c = __popc(n4);
                const int incl = mr_wave_scan(c);
                uint32_t pos = (uint32_t)(run + incl - c);
                run += __builtin_amdgcn_readlane(incl, 63);
                const int i = wbase + g * 256 + t4e;
                const uint32_t w0 = kk[g][0], w1 = kk[g][1];
                do {                                                            // (a second round where some lane keeps two elements of a g: one g in four)
                    const uint32_t r = (uint32_t)(__ffs(n4) - 1) & 3u;           // (a lane without one: element 3, into its dummy pair)
                    const uint32_t top = ((r & 2u) ? w1 : w0) >> ((r & 1u) * 16u);
                    const uint32_t key = (top << 16) | (uint32_t)lo16[i + r];
                    const uint32_t vb = key ^ (~(uint32_t)((int32_t)key >> 31) | 0x80000000u);
                    stage[(n4 != 0u && pos < (uint32_t)kMrStage) ? pos : dummy] = make_uint2((uint32_t)i + r, vb);
                    ++pos;
                    n4 &= n4 - 1u;
                } while (__builtin_amdgcn_ballot_w64(n4 != 0u) != 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            MR_T(9);
            const int n_out = min(row_tot, a.slot_cap);
            const size_t run0 = (size_t)b_first * a.slot_cap + (size_t)run_rows;     // the row's first pair in the scratch arrays
            mr_barrier();
            for (int e = tid; e < n_out; e += kMrThreads) {
                const uint2 pr = stage[e];
                a.slot_cols[run0 + e] = (int32_t)pr.x;
                a.slot_vals[run0 + e] = __uint_as_float(pr.y);
            }
            if (tid == 0) a.row_nnz[b] = (int64_t)run_rows;                      // the row's offset in the workgroup's run; step 5 adds the workgroup's
            run_rows += n_out;
            MR_T(10);
        }
    }
#ifdef MR_TIMING
    const unsigned long long rt_rows = __builtin_amdgcn_s_memrealtime();
    unsigned long long rt_prefix = 0;
#endif
    // ---- 5. mask -> CSR: the workgroup's run -> its place in the CSR arrays (what a scan + a compaction kernel did before) ----------------
    if constexpr (CSR != 0) {
        __syncthreads();                                                        // (vmcnt(0) + barrier: the run and the row offsets are written)
        unsigned long long* wg_tot = reinterpret_cast<unsigned long long*>(a.flags) + 32;
        // (total and flag are ONE word and nothing else passes between workgroups: relaxed agent-scope accesses -- sc1: past the XCD's own
        //  L2 -- and no fence.  A release / acquire pair at agent scope writes back and invalidates the XCD's L2: measured, 10 - 30 us per
        //  workgroup behind a kernel that has just written its rows)
        if (tid == 0) __hip_atomic_store(&wg_tot[vwg], (unsigned long long)run_rows | kMrDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 64) {                                                         // wave 0: the totals of the workgroups before this one
            long long part = 0;
            for (int c0 = 0; c0 < vwg; c0 += 64) {
                const int j = c0 + lane;
                unsigned long long t = kMrDone;
                if (j < vwg) {
                    t = __hip_atomic_load(&wg_tot[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while (!(t & kMrDone)) {
                        __builtin_amdgcn_s_sleep(8);
                        t = __hip_atomic_load(&wg_tot[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                part += (long long)(t & (kMrDone - 1));
            }
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
            if (lane == 0) { scratch[16] = (int)(part & 0xFFFFFFFFll); scratch[17] = (int)(part >> 32); }
        }
        __syncthreads();
        const long long p0 = ((long long)(uint32_t)scratch[16]) | ((long long)scratch[17] << 32);
#ifdef MR_TIMING
        rt_prefix = __builtin_amdgcn_s_memrealtime();
#endif
        const int n_rows_wg = b_end - b_first;
        for (int t = tid; t < n_rows_wg; t += kMrThreads)
            a.csr_rowptr[b_first + t] = p0 + (long long)a.row_nnz[b_first + t];
        if (b_end == a.B && tid == 0) a.csr_rowptr[a.B] = p0 + run_rows;
        if (p0 + run_rows > a.csr_cap) { if (tid == 0) atomicOr(a.flags, 4); }
        else {
            const int32_t* sc = a.slot_cols + (size_t)b_first * a.slot_cap;
            const int32_t* sv = reinterpret_cast<const int32_t*>(a.slot_vals) + (size_t)b_first * a.slot_cap;
            int32_t* dc = a.csr_cols + p0;
            int32_t* dv = reinterpret_cast<int32_t*>(a.csr_vals) + p0;
            // (plain loads: the run is this workgroup's own -- written by this CU after the launch, by nobody else, and not read before: its
            //  L1 holds no older copy; the stores are complete (vmcnt(0) above))
            // (eight elements' loads in flight per thread before the first store: 4 rows of 900 pairs in one round trip)
            for (int e0 = 0; e0 < run_rows; e0 += 8 * kMrThreads) {
                int32_t c[8], v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { const int e = e0 + i * kMrThreads + tid; c[i] = e < run_rows ? sc[e] : 0; v[i] = e < run_rows ? sv[e] : 0; }
#pragma unroll
                for (int i = 0; i < 8; ++i) { const int e = e0 + i * kMrThreads + tid; if (e < run_rows) { dc[e] = c[i]; dv[e] = v[i]; } }
            }
        }
    }
#ifdef MR_TIMING
    MR_T(15);
    tacc[16] = tlast - tstart;
    if constexpr (CSR != 0) {
        if (tid == 0) {       // (100 MHz clock, the same for every CU: start, rows done, end of this workgroup -- behind the totals)
            unsigned long long* rt = reinterpret_cast<unsigned long long*>(a.flags) + 32 + gridDim.x + 4 * vwg;
            rt[0] = rt_start; rt[1] = rt_rows; rt[2] = __builtin_amdgcn_s_memrealtime(); rt[3] = rt_prefix;
        }
    }
    if (tid == 0)
        for (int i = 0; i < 17; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(a.flags) + 1 + i, tacc[i]);
#endif
}
