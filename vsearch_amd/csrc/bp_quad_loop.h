// bp_quad_loop.h -- the inner loop of the quad walk (bp_quad.h): what a wave does with its share of a (block, tile)'s descriptors.
//
// A list is stored in CHUNKS of 256 bytes = 16 lanes x 4 postings; a posting is one dword: accumulator index of the document
// (low half) | fp16 value (high half); unused cells hold value 0.  A DESCRIPTOR (8 bytes, in LDS) = one chunk of one list for one
// query slot:  x = byte offset of the chunk from the block's first chunk | slot * 16 (the chunk offset's low byte is free),
//              y = the query's weight on the column (fp32 bits, pre-scaled).
// A wave STEP = 4 descriptors, one per 16-lane group: one ds_read_b64 (the group's descriptor), ONE global_load_dwordx4 (a
// lane's 4 postings), then 4 x (v_fma_mix_f32, v_cvt_i32_f32, v_mad_u32_u16, ds_add_u32).  No list lengths, no tails, no
// predication: 3.75 VALU instructions per list where the list walk of bp_walk.h spends ~ 5 per ds_add (7.5 all in) and is bound by
// VALU issue (profiles/r04_conflicts.txt).  The 16 postings a list's lanes add in one instruction sit in 16 different LDS banks
// (the builder deals them so), so a 32-lane half of a ds_add -- two lists -- puts at most 2 lanes on a bank: the cost of a
// conflict-free atomic (tools/microbench/lds_conflicts.hip: 5.0 cycles for <= 2 lanes per bank, 7.3 for 3).
// Accumulator index of (document d, slot q) = (d / 16) * 144 + q * 16 + d % 16 (dwords from LDS address 0): a slot's 16 documents are
// 64 contiguous bytes, the 9th row of a 16-document group is padding that staggers the banks of neighbouring groups.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vs {

constexpr int kQuadChunkBytes = 256, kQuadCells = 64, kQuadGroupDw = 144;
__host__ __device__ constexpr uint32_t quad_acc_index(uint32_t doc) { return (doc >> 4) * (uint32_t)kQuadGroupDw + (doc & 15u); }

typedef uint32_t quad_u32x4 __attribute__((ext_vector_type(4)));

// one posting into the accumulators: (weight x fp16 value) truncated, added at index(document) + slot row
__device__ __forceinline__ void quad_add(uint32_t p, float w, uint32_t so_bytes) {
    float prod;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(prod) : "v"(w), "v"(p));
    uint32_t addr;
    asm("v_mad_u32_u16 %0, %1, 4, %2" : "=v"(addr) : "v"(p), "v"(so_bytes));
    __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) int32_t*>(addr), (int32_t)prod, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// `desc`: LDS, `n_steps` wave steps in all (the table holds 4 * n_steps descriptors, a multiple of 64: null descriptors fill up);
// wave `wv` of `NW` takes steps wv, wv + NW, ...; `base`: the block's first chunk (wave-uniform); D steps' loads in flight.
template <int D, int NW>
__device__ __forceinline__ void quad_walk(const uint2* desc, int n_steps, int wv, int lane, const char* base) {
    const int g = lane >> 4;
    const uint32_t s16 = (uint32_t)(lane & 15) * 16u;
    const int trips = n_steps / NW;                      // (n_steps is a multiple of NW)
    uint2 d[D];
    quad_u32x4 p[D];
    const uint2* dp = desc + (size_t)wv * 4 + g;
    auto fetch = [&](int i) {
        d[i] = *dp;
        dp += NW * 4;
        const uint32_t off = (d[i].x & 0xFFFFFF00u) | s16;
        p[i] = *reinterpret_cast<const quad_u32x4*>(base + off);
    };
#pragma unroll
    for (int i = 0; i < D; ++i)
        if (i < trips) fetch(i);
    for (int t = 0; t < trips; t += D) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            if (t + i < trips) {
                const float w = __uint_as_float(d[i].y);
                const uint32_t so = (d[i].x & 0xFFu) << 2;              // slot * 64 bytes
                const quad_u32x4 q = p[i];
                if (t + i + D < trips) fetch(i);
                quad_add(q.x, w, so);
                quad_add(q.y, w, so);
                quad_add(q.z, w, so);
                quad_add(q.w, w, so);
            }
        }
    }
}

}  // namespace vs
