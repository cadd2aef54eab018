#!/usr/bin/env python3
"""Generates vsearch_amd/csrc/bp_hex_asm.h: the inner loops of the 16-slot ("hex") walk (bp_hex.h states the data layout), each ONE
inline-asm statement -- the recipe of tools/gen_quad_asm.py (named VGPR sets, counted s_waitcnt vmcnt / lgkmcnt, loads, waits and
consumers in one statement) re-cut for tiles of 16 queries over blocks of <= 1024 documents:

  * a list's chunk is 128 bytes = ONE cache line = 16 lanes x 2 postings (one global_load_dwordx2 per lane); a wave step serves 4 lists,
    one per 16-lane group, so a 32-lane half of a ds_add_u32 still holds two lists of 16 bank-distinct postings each (<= 2 lanes a bank:
    the cost of a conflict-free atomic).  (8 lanes x dwordx4 would put four independent lists into a half: 3 - 4 lanes on a bank.)
  * a table descriptor is 4 bytes: column (low half: the main chunk of column c is chunk c of every block) | the weight as ONE fp16
    number (high half).  The query slot is not in it: the table is the concatenation of the tile's queries (each padded to whole steps),
    and the statement tracks the slot of the step it consumes in SGPRs -- the step number against the next slot's first step, taken
    from a VGPR of boundaries by v_readlane_b32 when it is passed (at most 15 times a block).  So the accumulator address is
    v_mad_u32_u16(posting, 4, slot offset SGPR) and nothing per lane decodes a slot.
  * the weight w * S * 2^(ve - 16) (< 2^15) was rounded to fp16 by the table's builder; one v_fma_mix_f32 per step turns it back into
    fp32 times 2^(16 - ve) (an SGPR), the per-posting product is the quad walk's v_fma_mix_f32 (fp32 weight x fp16 value), exact.

Register SETS i = 0 .. S-1, each {descriptor d (1 VGPR), postings p (2 VGPRs)}.  Trip t works on set i = t mod S:
    s_waitcnt vmcnt(S-2)                      the postings of step t have landed
    [slot boundary check: s_cmp + s_cbranch]  scalar
    1 VALU                                    weight -> fp32
    [link check: v_cmp + s_cbranch]
    4 VALU                                    2 x v_fma_mix_f32, 2 x v_mad_u32_u16
    v_cvt_i32_f32, ds_add_u32                 posting 0
    ds_read_b32 d[i] <- descriptor of step t+S
    s_waitcnt lgkmcnt(3)                      the descriptor read of the PREVIOUS trip (set j = i-1) is back: behind it were issued
                                              that trip's second ds_add, this trip's first and this trip's ds_read
    1 VALU + global_load_dwordx2 p[j]         the postings of step t+S-1
    v_cvt_i32_f32, ds_add_u32                 posting 1
= 9 VALU, 1 VMEM, 3 DS per step of 4 lists x 32 cells.

LINKS as in the quad walk: a list longer than a chunk continues in an overflow chunk of its block; the chunk's last cell (lane 15 of the
group, posting 1) has the sign bit set and, with the cell before it, carries the overflow chunk's index in 2 x 14 bits.  Linking lanes
append an 8-byte descriptor {chunk byte offset | slot * 4, fp32 weight} to the WAVE's list in LDS; the wave walks that list right after
its share of the table (hex_list_asm: ds_read_b64 descriptors, the slot decoded per lane -- the rare path).

Three statements: hex_walk_asm (table: adds + links), hex_list_asm (the wave's own list: consecutive steps, 8-byte descriptors),
hex_collect_asm (table, links only: the segment mode's first pass).

usage: gen_hex_asm.py [S] [out] [variant]      variant (microbenchmarks): nolds / noload / nolink
"""
import sys

S = int(sys.argv[1]) if len(sys.argv) > 1 else 6
OUT = sys.argv[2] if len(sys.argv) > 2 else "vsearch_amd/csrc/bp_hex_asm.h"
VARIANT = sys.argv[3] if len(sys.argv) > 3 else ""
NW = 16                      # waves of the workgroup
V0 = 64
out = []
def emit(x): out.append(x)

# registers: postings pairs p[i] at v[V0 + 2 i] (gfx950 wants 64-bit tuples on even registers), descriptors behind them (table: 1 register
# a set; list: an even-aligned pair), 8 temporaries behind those (even base: DX:DY is a ds_write_b64 operand)
PB = V0
DBASE = V0 + 2 * S
T0 = DBASE + 2 * S           # (the table statements leave S registers unused: one base for all three)
T = [f"v{T0 + k}" for k in range(8)]
t0, t1, a0, a1, WF, VOFF, DX, DY = T          # DX:DY = a list descriptor being written (consecutive registers)
NTMP = 8

# MODE: "table" (4-byte descriptors, slot by boundary), "list" (8-byte descriptors), "collect" (table, links only)

def gen(MODE):
    global out
    out = []
    table = MODE in ("table", "collect")
    collect = MODE == "collect"
    DB = 4 if table else 8                               # descriptor bytes
    STEP = (NW if table else 1) * 4 * DB                 # bytes between a wave's consecutive steps
    # list mode: set i = {dx, dy, p0, p1}: 4 registers
    def pp(i, k): return f"v{PB + 2 * i + k}"
    def ppr(i): return f"v[{PB + 2 * i}:{PB + 2 * i + 1}]"
    if table:
        def dx(i): return f"v{DBASE + i}"
        def dread(i, off): emit(f"ds_read_b32 {dx(i)}, %[dptr] offset:{off}")
    else:
        def dx(i): return f"v{DBASE + 2 * i}"
        def dy(i): return f"v{DBASE + 2 * i + 1}"
        def dread(i, off): emit(f"ds_read_b64 v[{DBASE + 2 * i}:{DBASE + 2 * i + 1}], %[dptr] offset:{off}")

    def load(j):
        if table: emit(f"v_mad_u32_u16 {VOFF}, {dx(j)}, %[c128], %[s8]")            # column * 128 + 8 * (lane & 15)
        else: emit(f"v_and_or_b32 {VOFF}, {dx(j)}, %[m128], %[s8]")                 # chunk byte offset | 8 * (lane & 15)
        if "noload" in VARIANT: emit(f"v_mov_b32 {pp(j, 0)}, {VOFF}"); emit(f"v_mov_b32 {pp(j, 1)}, 0")
        else: emit(f"global_load_dwordx2 {ppr(j)}, {VOFF}, %[base]")

    def slotfix(i):
        """table modes: the step consumed now is step tcur; passing the next slot's first step moves the slot registers on"""
        emit("s_cmp_ge_u32 %[tcur], %[nb]")
        emit(f"s_cbranch_scc1 5{i}f")
        emit(f"6{i}:")

    def slotfix_tail(i):
        emit(f"5{i}:")
        emit("s_add_u32 %[sidx], %[sidx], 1")
        emit("s_add_u32 %[so], %[so], 64")
        emit("s_nop 0")
        emit("v_readlane_b32 %[nb], %[vbnd], %[sidx]")
        emit("s_nop 3")
        emit("s_cmp_ge_u32 %[tcur], %[nb]")
        emit(f"s_cbranch_scc1 5{i}b")
        emit(f"s_branch 6{i}b")

    def links(i):
        """lanes of set i whose last cell is a link append {chunk byte offset | slot * 4, fp32 weight} to the WAVE's list: index = the
        wave's running count (an SGPR) + the lane's rank among the linking lanes -- no atomic, nothing to wait for"""
        emit(f"v_cmp_gt_i32 vcc, 0, {pp(i, 1)}")
        emit(f"s_cbranch_vccz 7{i}f")
        emit("s_bcnt1_i32_b64 %[st], vcc")
        emit(f"v_mbcnt_lo_u32_b32 {a1}, vcc_lo, 0")
        emit(f"v_mbcnt_hi_u32_b32 {a1}, vcc_hi, {a1}")
        emit("s_and_saveexec_b64 %[sv], vcc")
        emit(f"v_add_u32 {a1}, %[cnt], {a1}")                           # number in the wave's list
        emit(f"v_and_b32 {a0}, 0x3fff, {pp(i, 1)}")
        emit(f"v_lshl_or_b32 {a0}, {pp(i, 0)}, 14, {a0}")               # overflow chunk index
        if table:
            emit("s_lshr_b32 %[st2], %[so], 4")                         # slot * 4
            emit(f"v_lshl_or_b32 {DX}, {a0}, 7, %[st2]")                # descriptor: chunk byte offset | slot * 4
            emit(f"v_mov_b32 {DY}, {WF}")                               #             fp32 weight
        else:
            emit(f"v_and_b32 {DX}, 0x7c, {dx(i)}")
            emit(f"v_lshl_or_b32 {DX}, {a0}, 7, {DX}")
            emit(f"v_mov_b32 {DY}, {dy(i)}")
        emit(f"v_cmp_gt_u32 vcc, %[cap], {a1}")
        emit(f"v_lshl_add_u32 {a1}, {a1}, 3, %[lbase]")
        emit("s_and_b64 exec, exec, vcc")                               # lanes with room in the list (the caller sees cnt > cap otherwise)
        emit(f"ds_write_b64 {a1}, v[{T0 + 6}:{T0 + 7}]")
        emit("s_mov_b64 exec, %[sv]")
        emit("s_add_u32 %[cnt], %[cnt], %[st]")
        emit(f"7{i}:")

    # prologue: steps 0 .. S-2 loaded, descriptor of step S-1 read
    if table:
        emit("v_readlane_b32 %[nb], %[vbnd], 0")
    for k in range(S):
        dread(k, k * STEP)
    emit("s_waitcnt lgkmcnt(0)")
    for k in range(S - 1):
        load(k)
    emit(f"v_add_u32 %[dptr], {S * STEP}, %[dptr]")
    emit("1:")
    for i in range(S):
        j = (i - 1) % S
        if "noload" not in VARIANT: emit(f"s_waitcnt vmcnt({S - 2})")
        if table:
            slotfix(i)
            # the weight: fp16 (high half of the descriptor) -> fp32, times 2^(16 - ve)
            emit(f"v_fma_mix_f32 {WF}, {dx(i)}, %[hmul], 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]")
            wsrc = WF
        else:
            wsrc = dy(i)
        if "nolink" not in VARIANT: links(i)
        if collect:
            dread(i, i * STEP)
            emit("s_waitcnt lgkmcnt(1)")                                   # the previous trip's read (set j) is back
            load(j)
        else:
            if table:
                so0 = so1 = "%[so]"
            else:
                emit(f"v_and_b32 {VOFF}, 0x7c, {dx(i)}")
                emit(f"v_lshlrev_b32 {VOFF}, 4, {VOFF}")                   # slot * 64
                so0 = so1 = VOFF
            emit(f"v_fma_mix_f32 {t0}, {wsrc}, {pp(i, 0)}, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]")
            emit(f"v_mad_u32_u16 {a0}, {pp(i, 0)}, 4, {so0}")
            emit(f"v_fma_mix_f32 {t1}, {wsrc}, {pp(i, 1)}, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]")
            emit(f"v_mad_u32_u16 {a1}, {pp(i, 1)}, 4, {so1}")
            emit(f"v_cvt_i32_f32 {t0}, {t0}")
            if "nolds" not in VARIANT: emit(f"ds_add_u32 {a0}, {t0}")
            # set i is consumed (its last readers have issued): its descriptor register takes the descriptor of step t + S
            dread(i, i * STEP)
            # the descriptor read of the previous trip (set j) is back: behind it were issued 1 ds_add, this trip's ds_add + ds_read
            emit("s_waitcnt lgkmcnt(3)" if "nolds" not in VARIANT else "s_waitcnt lgkmcnt(1)")
            load(j)
            emit(f"v_cvt_i32_f32 {t1}, {t1}")
            if "nolds" not in VARIANT: emit(f"ds_add_u32 {a1}, {t1}")
        if i == S - 1:
            emit(f"v_add_u32 %[dptr], {S * STEP}, %[dptr]")
        if table: emit(f"s_add_u32 %[tcur], %[tcur], {NW}")
        emit("s_sub_u32 %[n], %[n], 1")
        emit("s_cmp_eq_u32 %[n], 0")
        emit("s_cbranch_scc1 8f")
    emit("s_branch 1b")
    if table:
        for i in range(S): slotfix_tail(i)
    emit("8:")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return out

body_table = gen("table")
body_list = gen("list")
body_collect = gen("collect")
n_vregs = 4 * S + NTMP
vregs = [f"v{V0 + i}" for i in range(n_vregs)]

def stmt(lines): return "\\n\\t\"\n        \"".join(lines)
clob = ", ".join(f'"{r}"' for r in vregs)

def fn_table(name, lines, what):
    return f'''// {what}
// dptr: LDS byte address of this lane group's descriptor of the wave's first step; trips >= 1: steps of this wave; step0: number of that
// first step in the table; vbnd: lane s = first step of slot s + 1 (lanes >= 15: 0xFFFFFFFF); hmul: 2^(16 - ve) (hex_weight_scale);
// base: the block's first chunk (wave-uniform); s8: 8 x (lane & 15).  Accumulators at LDS address 0.
// Links go to the wave's list at LDS byte address lbase (capacity cap descriptors); returns how many there were (> cap: not all stored).
__device__ __forceinline__ uint32_t {name}(uint32_t dptr, uint32_t trips, uint32_t step0, uint32_t vbnd, float hmul, const char* base, uint32_t s8, uint32_t lbase, uint32_t cap) {{
    uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane(trips);
    uint32_t tcur = (uint32_t)__builtin_amdgcn_readfirstlane(step0);
    const unsigned long long pb = (unsigned long long)base;
    const unsigned long long ub = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32)) << 32) |
                                  (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pb);
    const uint32_t c128 = 128u;
    const uint32_t hm = (uint32_t)__builtin_amdgcn_readfirstlane(__float_as_uint(hmul));
    const uint32_t lb = (uint32_t)__builtin_amdgcn_readfirstlane(lbase), cp = (uint32_t)__builtin_amdgcn_readfirstlane(cap);
    uint32_t cnt = 0, st, st2, nb, sidx = 0, so = 0;
    unsigned long long sv;
    asm volatile(
        "{stmt(lines)}\\n\\t"
        : [n] "+s"(n), [dptr] "+v"(dptr), [cnt] "+s"(cnt), [tcur] "+s"(tcur), [sidx] "+s"(sidx), [so] "+s"(so), [st] "=&s"(st), [st2] "=&s"(st2), [nb] "=&s"(nb), [sv] "=&s"(sv)
        : [base] "s"(ub), [s8] "v"(s8), [c128] "s"(c128), [hmul] "s"(hm), [vbnd] "v"(vbnd), [lbase] "s"(lb), [cap] "s"(cp)
        : "memory", "scc", "vcc", {clob});
    return cnt;
}}
'''

def fn_list(name, lines, what):
    return f'''// {what}
// dptr: LDS byte address of this lane group's first descriptor (8 bytes: chunk byte offset | slot * 4, fp32 weight); trips >= 1: steps.
__device__ __forceinline__ uint32_t {name}(uint32_t dptr, uint32_t trips, const char* base, uint32_t s8, uint32_t lbase, uint32_t cap) {{
    uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane(trips);
    const unsigned long long pb = (unsigned long long)base;
    const unsigned long long ub = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32)) << 32) |
                                  (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pb);
    const uint32_t m128 = 0xFFFFFF80u;
    const uint32_t lb = (uint32_t)__builtin_amdgcn_readfirstlane(lbase), cp = (uint32_t)__builtin_amdgcn_readfirstlane(cap);
    uint32_t cnt = 0, st;
    unsigned long long sv;
    asm volatile(
        "{stmt(lines)}\\n\\t"
        : [n] "+s"(n), [dptr] "+v"(dptr), [cnt] "+s"(cnt), [st] "=&s"(st), [sv] "=&s"(sv)
        : [base] "s"(ub), [s8] "v"(s8), [m128] "s"(m128), [lbase] "s"(lb), [cap] "s"(cp)
        : "memory", "scc", "vcc", {clob});
    return cnt;
}}
'''

hdr = f'''// GENERATED by tools/gen_hex_asm.py {S} -- do not edit; the generator says what the statements do and why they are asm.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vs {{

constexpr int kHexSets = {S};                 // register sets: kHexSets - 1 posting loads of a wave in flight
constexpr int kHexOverRead = {S};             // steps a wave reads descriptors of beyond its last one: a table ends with 16 x that many null steps

''' + fn_table("hex_walk_asm", body_table, "walk the workgroup's descriptor table (a wave takes steps w, w + 16, ...): add its chunks' postings, collect their links") + "\n" + \
      fn_list("hex_list_asm", body_list, "the same over the wave's OWN list (consecutive steps, 8-byte descriptors)") + "\n" + \
      fn_table("hex_collect_asm", body_collect, "collect the links of the wave's steps of the workgroup's table, add nothing") + '''
}  // namespace vs
'''
open(OUT, "w").write(hdr)
print(f"{OUT}: S {S}, VGPRs v{V0}..v{V0 + n_vregs - 1}, {len(body_table)} + {len(body_list)} + {len(body_collect)} instructions")
