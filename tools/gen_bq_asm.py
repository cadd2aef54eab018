#!/usr/bin/env python3
"""Generates vsearch_amd/csrc/bp_bq_asm.h: the inner loop of the bag-of-token chunk walk (bp_bq.h) as ONE inline-asm statement --
the quad walk's loop (tools/gen_quad_asm.py) re-cut for a BINARY index: lists of ~6 postings, no values.

Data (bp_bq.h): a list of a (block, column) is a CHUNK of 32 bytes = 16 cells of uint16 (document in the block; pad cells point at the
spare documents behind a slot's plane); the main chunk of column c is chunk c of its block.  A DESCRIPTOR (8 bytes, LDS) = one chunk
for one query slot: x = chunk index | LDS byte address of the slot's plane << 16, y = the query's integer weight.
A wave STEP = 8 descriptors, one per 8-lane group: lane i of a group loads dword i of the chunk (cells 2 i, 2 i + 1) with ONE
global_load_dword and adds the weight at both documents:
    1 ds_read_b64 + 1 global_load_dword + 2 x (v_mad_u32_u16, ds_add_u32) + 3 address instructions      per 8 lists
against ~ 20 instructions per list of the record walk (bp_bin.h), which is bound by exactly that (1 700 instructions per
wave and block).  The loop keeps S - 1 loads and one descriptor read of a wave in flight with counted vmcnt / lgkmcnt, as the quad
loop does (register sets {descriptor 2 VGPRs, postings 1 VGPR}).

LINKS.  A list of more than 16 postings (1 in 7 000 at 6 postings a list) keeps 15 in its chunk; cell 15 -- the high half of the
group's last dword -- has bit 15 set and carries the index of the block's overflow chunk that continues the list.  After a step's
postings have landed one v_cmp + s_cbranch asks whether any lane holds a link; the lanes that do append a descriptor {overflow chunk |
plane, weight} to the WAVE's list (SGPR count + v_mbcnt rank, no atomic) and sit out the second add of that step (their high cell is
not a document).  Three statements as for the quad walk: bq_walk_asm (the workgroup's table: a wave takes steps w, w + 16, ...),
bq_list_asm (the wave's own list), bq_collect_asm (links only: the segment mode when a list overflows).
"""
import sys

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
OUT = sys.argv[2] if len(sys.argv) > 2 else "vsearch_amd/csrc/bp_bq_asm.h"
# experiment ("mask"): pad cells (document >= 2048) sit out their ds_add (v_cmpx_gt_u16_sdwa) -- fewer lanes per bank, 3 more instructions a
# step: 51.9 ms against 51.0 at 21 M docs (profiles/r05_bq_mask.txt): the LDS conflicts are not what the walk waits for.  Off.
MASK = (sys.argv[3] if len(sys.argv) > 3 else "nomask") == "mask"
NW = 16
STEP = NW * 8 * 8            # bytes between a wave's consecutive steps of the table (8 descriptors of 8 bytes per step); own list: 8 * 8
V0 = 64
out = []
def emit(x): out.append(x)
COLLECT = False

# (64-bit VGPR tuples must start on an even register: the descriptor pairs first, then the postings, then an even-aligned pair of
#  temporaries that doubles as the link descriptor a ds_write_b64 stores)
def d(i, j): return f"v{V0 + 2 * i + j}"
def p(i): return f"v{V0 + 2 * S + i}"
def drange(i): return f"v[{V0 + 2 * i}:{V0 + 2 * i + 1}]"
T0 = (V0 + 3 * S + 1) & ~1
A0, A1, SO, VOFF, T2, LIM = (f"v{T0 + k}" for k in range(6))

def load(j):
    emit(f"v_and_b32 {VOFF}, 0xffff, {d(j, 0)}")
    emit(f"v_lshl_or_b32 {VOFF}, {VOFF}, 5, %[l4]")
    emit(f"global_load_dword {p(j)}, {VOFF}, %[base]")

def append_links(i):
    """lanes of set i whose high cell is a link (vcc) append {overflow chunk | plane, weight} to the wave's list"""
    emit("s_bcnt1_i32_b64 %[st], vcc")
    emit(f"v_mbcnt_lo_u32_b32 {T2}, vcc_lo, 0")
    emit(f"v_mbcnt_hi_u32_b32 {T2}, vcc_hi, {T2}")
    emit("s_and_saveexec_b64 %[sv], vcc")
    emit(f"v_add_u32 {T2}, %[cnt], {T2}")                          # number in the wave's list
    emit(f"v_bfe_u32 {A0}, {p(i)}, 16, 15")                        # overflow chunk of the block (the link's payload)
    emit(f"v_add_u32 {A0}, %[ncols], {A0}")                        # ... behind the main chunks
    emit(f"v_and_b32 {A1}, 0xffff0000, {d(i, 0)}")                 # plane address
    emit(f"v_or_b32 {A0}, {A0}, {A1}")
    emit(f"v_mov_b32 {A1}, {d(i, 1)}")                             # weight
    emit(f"v_cmp_gt_u32 vcc, %[cap], {T2}")
    emit(f"v_lshl_add_u32 {T2}, {T2}, 3, %[lbase]")
    emit("s_and_b64 exec, exec, vcc")                              # lanes with room in the list (the caller sees cnt > cap otherwise)
    emit(f"ds_write_b64 {T2}, v[{T0}:{T0 + 1}]")                   # (A0, A1: an even-aligned pair)
    emit("s_mov_b64 exec, %[sv]")
    emit("s_add_u32 %[cnt], %[cnt], %[st]")

def build():
    global out
    out = []
    emit(f"v_mov_b32 {LIM}, 0x800")                                 # first document id that is not one (pads 2048 .. 2111, links >= 0x8000)
    for k in range(S):
        emit(f"ds_read_b64 {drange(k)}, %[dptr] offset:{k * STEP}")
    emit("s_waitcnt lgkmcnt(0)")
    for k in range(S - 1):
        load(k)
    emit(f"v_add_u32 %[dptr], {S * STEP}, %[dptr]")
    emit("1:")
    for i in range(S):
        j = (i - 1) % S
        emit(f"s_waitcnt vmcnt({S - 2})")
        emit(f"v_cmp_gt_i32 vcc, 0, {p(i)}")
        if COLLECT:
            emit(f"s_cbranch_vccz 7{i}f")
            append_links(i)
            emit(f"7{i}:")
            emit(f"ds_read_b64 {drange(i)}, %[dptr] offset:{i * STEP}")
            emit("s_waitcnt lgkmcnt(1)" )                           # the previous trip's read (set j) is back -- (a link write drains in order before it)
            load(j)
        else:
            emit(f"s_cbranch_vccnz 7{i}f")
            # common case: no link in this step
            emit(f"v_lshrrev_b32 {SO}, 16, {d(i, 0)}")
            emit(f"v_mad_u32_u16 {A0}, {p(i)}, 4, {SO}")
            emit(f"v_mad_u32_u16 {A1}, {p(i)}, 4, {SO} op_sel:[1,0,0,0]")
            if MASK:
                # 10 of a list's 16 cells are pads: masked out, a ds_add's active lanes rarely share a bank (a link cell, bit 15, sits out too)
                emit(f"v_cmpx_gt_u16_sdwa vcc, {LIM}, {p(i)} src0_sel:DWORD src1_sel:WORD_0")
                emit(f"ds_add_u32 {A0}, {d(i, 1)}")
                emit(f"v_cmpx_gt_u16_sdwa vcc, {LIM}, {p(i)} src0_sel:DWORD src1_sel:WORD_1")
                emit(f"ds_add_u32 {A1}, {d(i, 1)}")
                emit("s_mov_b64 exec, -1")
            else:
                emit(f"ds_add_u32 {A0}, {d(i, 1)}")
                emit(f"ds_add_u32 {A1}, {d(i, 1)}")
            emit(f"6{i}:")
            # set i is consumed: its descriptor registers take the descriptor of step t + S
            emit(f"ds_read_b64 {drange(i)}, %[dptr] offset:{i * STEP}")
            # the descriptor read of the previous trip (set j) is back: behind it were issued this trip's 2 ds_add + ds_read
            emit("s_waitcnt lgkmcnt(3)")
            load(j)
        if i == S - 1:
            emit(f"v_add_u32 %[dptr], {S * STEP}, %[dptr]")
        emit("s_sub_u32 %[n], %[n], 1")
        emit("s_cmp_eq_u32 %[n], 0")
        emit("s_cbranch_scc1 8f")
    emit("s_branch 1b")
    if not COLLECT:
        # rare: a step with links -- the descriptors go to the wave's list, the link lanes sit out the high cells' add; every LDS
        # operation issued so far is waited for, so that the counted wait behind the next descriptor read holds again
        for i in range(S):
            emit(f"7{i}:")
            append_links(i)
            emit(f"v_lshrrev_b32 {SO}, 16, {d(i, 0)}")
            emit(f"v_mad_u32_u16 {A0}, {p(i)}, 4, {SO}")
            emit(f"v_mad_u32_u16 {A1}, {p(i)}, 4, {SO} op_sel:[1,0,0,0]")
            emit(f"v_cmp_le_i32 vcc, 0, {p(i)}")                   # lanes WITHOUT a link
            emit(f"ds_add_u32 {A0}, {d(i, 1)}")
            emit("s_and_saveexec_b64 %[sv], vcc")
            emit(f"ds_add_u32 {A1}, {d(i, 1)}")
            emit("s_mov_b64 exec, %[sv]")
            emit(f"s_branch 6{i}b")
    emit("8:")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return out

COLLECT = False
body_add = build()
STEP = 8 * 8
body_list = build()
STEP = NW * 8 * 8
COLLECT = True
body_collect = build()
vregs = [f"v{r}" for r in range(V0, T0 + 6)]
def stmt(lines): return "\\n\\t\"\n        \"".join(lines)
clob = ", ".join(f'"{r}"' for r in vregs)
def fn(name, lines, what):
    return f'''// {what}
// dptr: LDS byte address of this 8-lane group's descriptor of the wave's first step; trips >= 1: steps of this wave; base: the block's
// first chunk (wave-uniform); l4: 4 x (lane & 7); ncols: chunks before the block's overflow chunks.  Slot planes from LDS address 0.
// Links go to the wave's list at LDS byte address lbase (capacity cap descriptors); returns how many there were (> cap: not all stored).
__device__ __forceinline__ uint32_t {name}(uint32_t dptr, uint32_t trips, const char* base, uint32_t l4, uint32_t ncols, uint32_t lbase, uint32_t cap) {{
    uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane(trips);
    const unsigned long long pb = (unsigned long long)base;
    const unsigned long long ub = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32)) << 32) |
                                  (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pb);
    const uint32_t lb = (uint32_t)__builtin_amdgcn_readfirstlane(lbase), cp = (uint32_t)__builtin_amdgcn_readfirstlane(cap);
    const uint32_t nc = (uint32_t)__builtin_amdgcn_readfirstlane(ncols);
    uint32_t cnt = 0, st;
    unsigned long long sv;
    asm volatile(
        "{stmt(lines)}\\n\\t"
        : [n] "+s"(n), [dptr] "+v"(dptr), [cnt] "+s"(cnt), [st] "=&s"(st), [sv] "=&s"(sv)
        : [base] "s"(ub), [l4] "v"(l4), [ncols] "s"(nc), [lbase] "s"(lb), [cap] "s"(cp)
        : "memory", "scc", "vcc", {clob});
    return cnt;
}}
'''
hdr = f'''// GENERATED by tools/gen_bq_asm.py {S} -- do not edit; the generator says what the statements do and why they are asm.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vs {{

constexpr int kBqSets = {S};                  // register sets: kBqSets - 1 chunk loads of a wave in flight
constexpr int kBqOverRead = {S};              // steps a wave reads descriptors of beyond its last one: a table ends with 16 x that many null steps

''' + fn("bq_walk_asm", body_add, "walk the workgroup's descriptor table (a wave takes steps w, w + 16, ...): add its chunks' postings, collect their links") + "\n" + \
      fn("bq_list_asm", body_list, "the same over the wave's OWN list (consecutive steps)") + "\n" + \
      fn("bq_collect_asm", body_collect, "collect the links of the wave's steps of the workgroup's table, add nothing") + '''
}  // namespace vs
'''
open(OUT, "w").write(hdr)
print(f"{OUT}: S {S}, VGPRs v{V0}..v{T0 + 5}, {len(body_add)} + {len(body_list)} + {len(body_collect)} instructions")
