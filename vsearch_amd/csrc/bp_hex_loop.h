// bp_hex_loop.h -- the data layout the 16-slot ("hex") walk's inner loop works on (bp_hex.h: the kernel; tools/gen_hex_asm.py: the loop).
//
// A list is stored in CHUNKS of 128 bytes = ONE cache line = 16 lanes x 2 postings; a posting is one dword: accumulator index of the
// document (low half) | fp16 value (high half); unused cells hold value 0.  A wave STEP = 4 lists, one per 16-lane group: one
// ds_read_b32 (the group's descriptor), one global_load_dwordx2 (a lane's 2 postings), then 2 x (v_fma_mix_f32, v_cvt_i32_f32,
// v_mad_u32_u16, ds_add_u32).  The 16 postings a list's lanes add in one instruction sit in 16 different LDS banks (the builder deals
// them so), so a 32-lane half of a ds_add -- two lists -- puts at most 2 lanes on a bank.
// A TABLE DESCRIPTOR (4 bytes, in LDS) = the main chunk of one list for one query: column (low half) | fp16 weight (high half); the
// query slot follows from the step's position (the table is the concatenation of the tile's queries, bp_hex.h).
// Accumulator index of (document d, slot q) = (d / 16) * 272 + q * 16 + d % 16 (dwords from LDS address 0): a slot's 16 documents are 64
// contiguous bytes, the 17th row of a 16-document group is padding that staggers the banks of neighbouring groups.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vs {

constexpr int kHexChunkBytes = 128, kHexCells = 32, kHexLinked = kHexCells - 2, kHexGroupDw = 272, kHexQT = 16;
__host__ __device__ constexpr uint32_t hex_acc_index(uint32_t doc) { return (doc >> 4) * (uint32_t)kHexGroupDw + (doc & 15u); }

}  // namespace vs
