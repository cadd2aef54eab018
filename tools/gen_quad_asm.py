#!/usr/bin/env python3
"""Generates vsearch_amd/csrc/bp_quad_asm.h: the inner loop of the quad walk (bp_quad_loop.h states the data layout) as ONE
inline-asm statement.

Why asm: a wave keeps S - 1 posting loads (global_load_dwordx4) and one descriptor read (ds_read_b64) in flight across loop
trips and counts them with s_waitcnt vmcnt(N) / lgkmcnt(N).  hipcc's own version of this loop waits vmcnt(0) lgkmcnt(0) in every
trip (the descriptor read feeds the load's address), which serialises the pipeline; and hipcc neither counts asm loads nor keeps
its own code out of their destination registers, so loads, waits and consumers live in one statement
(cdna_hip_programming.md 5, 'What hipcc does not do').  The statement names its VGPRs and lists them as clobbers.

Register SETS i = 0 .. S-1, each {descriptor d (2 VGPRs), postings p (4 VGPRs)}.  Trip t works on set i = t mod S:
    s_waitcnt vmcnt(S-2)                      the postings of step t have landed (S-1 loads were outstanding)
    2 VALU                                    slot row offset from d[i].x
    8 VALU                                    4 x v_fma_mix_f32 (weight x fp16 value), 4 x v_mad_u32_u16 (accumulator address)
    ds_read_b64 d[i] <- descriptor of step t+S    (set i is consumed: VALU operands are read at issue)
    s_waitcnt lgkmcnt(5)                      the descriptor read of the PREVIOUS trip (set j = i-1, step t+S-1) is back:
                                              behind it were issued 4 ds_add and this trip's ds_read
    1 VALU + global_load_dwordx4 p[j]         the postings of step t+S-1
    4 x (v_cvt_i32_f32, ds_add_u32)
= 15 VALU, 1 VMEM, 5 DS per step of 4 lists.  LDS operations complete in order (no scalar memory operations inside), which is
what makes the lgkmcnt arithmetic valid.

LINKS.  A list longer than a chunk continues in an overflow chunk of its block; the chunk says so itself: its last cell (lane 15 of
the group, posting 3) has the sign bit set (postings have non-negative values) and, with the cell before it, carries the overflow
chunk's index in 2 x 14 payload bits (both cells have value 0: as postings they add nothing, to a valid accumulator).  Right after a
step's postings have landed, one v_cmp + s_cbranch asks whether any lane holds a link; the (rare) lanes that do append a descriptor
{overflow chunk | slot, weight} to the WAVE's list in LDS (index = the wave's running count in an SGPR + v_mbcnt rank: no atomic, no
wait), which the wave walks itself after its share of the table.  The statement returns the count; beyond the list's capacity the
descriptors are dropped and the caller redoes the wave's overflow in segments.  No directory is read at search time.
Three statements: quad_walk_asm (table: adds + links), quad_list_asm (the wave's own list: consecutive steps), quad_collect_asm (links
only: the segment mode's first pass).
"""
import sys

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
OUT = sys.argv[2] if len(sys.argv) > 2 else "vsearch_amd/csrc/bp_quad_asm.h"
VARIANT = sys.argv[3] if len(sys.argv) > 3 else ""        # microbenchmark variants: nolds / noload / novalu
NW = 16                      # waves of the workgroup
STEP = NW * 4 * 8            # bytes between a wave's consecutive steps in the descriptor table (the wave's own list: 4 * 8)
V0 = 64
out = []
def emit(x): out.append(x)
COLLECT = False

def d(i, j): return f"v{V0 + 6 * i + j}"
def p(i, k): return f"v{V0 + 6 * i + 2 + k}"
def prange(i): return f"v[{V0 + 6 * i + 2}:{V0 + 6 * i + 5}]"
def drange(i): return f"v[{V0 + 6 * i}:{V0 + 6 * i + 1}]"
T0 = V0 + 6 * S              # temporaries: t0..1, a0..1 (two postings at a time), so, voff
def tk(k): return f"v{T0 + (k & 1)}"
def ak(k): return f"v{T0 + 2 + (k & 1)}"
SO, VOFF = f"v{T0 + 4}", f"v{T0 + 5}"

def load(j):
    emit(f"v_and_or_b32 {VOFF}, {d(j, 0)}, %[m256], %[s16]")
    if "noload" in VARIANT: emit(f"v_mov_b32 {p(j, 0)}, {VOFF}")
    else: emit(f"global_load_dwordx4 {prange(j)}, {VOFF}, %[base]")

def links(i, tag):
    """lanes of set i whose last cell is a link append a descriptor to the WAVE's list: the list index is the wave's running count
    (an SGPR) + the lane's rank among the linking lanes -- no atomic, nothing to wait for"""
    t0, t1, a0, a1 = f"v{T0}", f"v{T0 + 1}", f"v{T0 + 2}", f"v{T0 + 3}"
    emit(f"v_cmp_gt_i32 vcc, 0, {p(i, 3)}")
    emit(f"s_cbranch_vccz 7{tag}f")
    emit("s_bcnt1_i32_b64 %[st], vcc")
    emit(f"v_mbcnt_lo_u32_b32 {a1}, vcc_lo, 0")
    emit(f"v_mbcnt_hi_u32_b32 {a1}, vcc_hi, {a1}")
    emit("s_and_saveexec_b64 %[sv], vcc")
    emit(f"v_add_u32 {a1}, %[cnt], {a1}")                          # number in the wave's list
    emit(f"v_and_b32 {a0}, 0x3fff, {p(i, 3)}")
    emit(f"v_lshl_or_b32 {a0}, {p(i, 2)}, 14, {a0}")               # overflow chunk index
    emit(f"v_and_b32 {t0}, 0xff, {d(i, 0)}")                       # slot row
    emit(f"v_lshl_or_b32 {t0}, {a0}, 8, {t0}")                     # descriptor: chunk offset | slot row
    emit(f"v_mov_b32 {t1}, {d(i, 1)}")                             #             weight
    emit(f"v_cmp_gt_u32 vcc, %[cap], {a1}")
    emit(f"v_lshl_add_u32 {a1}, {a1}, 3, %[lbase]")
    emit("s_and_b64 exec, exec, vcc")                              # lanes with room in the list (the caller sees cnt > cap otherwise)
    emit(f"ds_write_b64 {a1}, v[{T0}:{T0 + 1}]")
    emit("s_mov_b64 exec, %[sv]")
    emit("s_add_u32 %[cnt], %[cnt], %[st]")
    emit(f"7{tag}:")

def build():
    global out
    out = []
    # prologue: steps 0 .. S-2 loaded, descriptor of step S-1 read
    for k in range(S):
        emit(f"ds_read_b64 {drange(k)}, %[dptr] offset:{k * STEP}")
    emit("s_waitcnt lgkmcnt(0)")
    for k in range(S - 1):
        load(k)
    emit(f"v_add_u32 %[dptr], {S * STEP}, %[dptr]")
    emit("1:")
    for i in range(S):
        j = (i - 1) % S
        if "noload" not in VARIANT: emit(f"s_waitcnt vmcnt({S - 2})")
        if "nolink" not in VARIANT: links(i, i)
        if COLLECT:
            emit(f"ds_read_b64 {drange(i)}, %[dptr] offset:{i * STEP}")
            emit("s_waitcnt lgkmcnt(1)")                                   # the previous trip's read (set j) is back
            load(j)
            if i == S - 1:
                emit(f"v_add_u32 %[dptr], {S * STEP}, %[dptr]")
            emit("s_sub_u32 %[n], %[n], 1")
            emit("s_cmp_eq_u32 %[n], 0")
            emit("s_cbranch_scc1 8f")
            continue
        emit(f"v_lshlrev_b32 {SO}, 2, {d(i, 0)}")
        emit(f"v_and_b32 {SO}, 0x3fc, {SO}")
        # postings 0, 1: products and addresses, truncation, adds -- the temporaries are free again once the ds_add has issued
        def mul(k):
            emit(f"v_fma_mix_f32 {tk(k)}, {d(i, 1)}, {p(i, k)}, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]")
            emit(f"v_mad_u32_u16 {ak(k)}, {p(i, k)}, 4, {SO}")
        def add(k):
            emit(f"v_cvt_i32_f32 {tk(k)}, {tk(k)}")
            if "nolds" not in VARIANT: emit(f"ds_add_u32 {ak(k)}, {tk(k)}")
        mul(0); mul(1); add(0); add(1)
        mul(2); mul(3)
        # set i is consumed (its last readers have issued): its descriptor registers take the descriptor of step t + S
        emit(f"ds_read_b64 {drange(i)}, %[dptr] offset:{i * STEP}")
        # the descriptor read of the previous trip (set j) is back: behind it were issued 2 + 2 ds_add and this trip's 2 ds_add + ds_read
        emit("s_waitcnt lgkmcnt(5)" if "nolds" not in VARIANT else "s_waitcnt lgkmcnt(1)")
        load(j)
        add(2); add(3)
        if i == S - 1:
            emit(f"v_add_u32 %[dptr], {S * STEP}, %[dptr]")
        emit("s_sub_u32 %[n], %[n], 1")
        emit("s_cmp_eq_u32 %[n], 0")
        emit("s_cbranch_scc1 8f")
    emit("s_branch 1b")
    emit("8:")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")

    return out

COLLECT = False
body_add = build()
STEP = 4 * 8
body_list = build()
STEP = NW * 4 * 8
COLLECT = True
body_collect = build()
out = body_add
vregs = [f"v{V0 + i}" for i in range(6 * S + 6)]
def stmt(lines): return "\\n\\t\"\n        \"".join(lines)
clob = ", ".join(f'"{r}"' for r in vregs)
def fn(name, lines, what):
    return f'''// {what}
// dptr: LDS byte address of this lane group's descriptor of the wave's first step; trips >= 1: steps of this wave;
// base: the block's first chunk (wave-uniform); s16: 16 x (lane & 15).  Accumulators at LDS address 0.
// Links go to the wave's list at LDS byte address lbase (capacity cap descriptors); returns how many there were (> cap: not all stored).
__device__ __forceinline__ uint32_t {name}(uint32_t dptr, uint32_t trips, const char* base, uint32_t s16, uint32_t lbase, uint32_t cap) {{{{
    uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane(trips);
    const unsigned long long pb = (unsigned long long)base;
    const unsigned long long ub = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32)) << 32) |
                                  (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pb);
    const uint32_t m256 = 0xFFFFFF00u;
    const uint32_t lb = (uint32_t)__builtin_amdgcn_readfirstlane(lbase), cp = (uint32_t)__builtin_amdgcn_readfirstlane(cap);
    uint32_t cnt = 0, st;
    unsigned long long sv;
    asm volatile(
        "{stmt(lines)}\\n\\t"
        : [n] "+s"(n), [dptr] "+v"(dptr), [cnt] "+s"(cnt), [st] "=&s"(st), [sv] "=&s"(sv)
        : [base] "s"(ub), [s16] "v"(s16), [m256] "s"(m256), [lbase] "s"(lb), [cap] "s"(cp)
        : "memory", "scc", "vcc", {clob});
    return cnt;
}}}}
'''
hdr = f'''// GENERATED by tools/gen_quad_asm.py {S} -- do not edit; the generator says what the statements do and why they are asm.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vs {{

constexpr int kQuadSets = {S};                // register sets: kQuadSets - 1 posting loads of a wave in flight
constexpr int kQuadOverRead = {S};            // steps a wave reads descriptors of beyond its last one: a table ends with 16 x that many null steps

''' + fn("quad_walk_asm", body_add, "walk the workgroup's descriptor table (a wave takes steps w, w + 16, ...): add its chunks' postings, collect their links").replace("{{", "{").replace("}}", "}") + "\n" + \
      fn("quad_list_asm", body_list, "the same over the wave's OWN list (consecutive steps)").replace("{{", "{").replace("}}", "}") + "\n" + \
      fn("quad_collect_asm", body_collect, "collect the links of the wave's steps of the workgroup's table, add nothing").replace("{{", "{").replace("}}", "}") + '''
}  // namespace vs
'''
open(OUT, "w").write(hdr)
print(f"{OUT}: S {S}, VGPRs v{V0}..v{V0 + 6 * S + 5}, {len(body_add)} + {len(body_list)} + {len(body_collect)} instructions")
