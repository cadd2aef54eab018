#!/usr/bin/env python3
"""Generates vsearch_amd/csrc/bp_head_asm.h: one work item of a wave of the head pre-pass (bp_head.h) as ONE inline-asm statement --
64 documents (4 strip operands per k-step) x 16 query tiles (8 weight operands of two tiles each), all k-steps, and the conversion + store of the sums.

Why asm: the product is 32 v_mfma_f32_16x16x32_f16 per k-step on 12 operands of 16 bytes per lane, 128 accumulator registers; with
two operand sets (the next step's loads in flight across this step's MFMAs) that is 224 of the 256 registers a wave has at two waves
per SIMD.  hipcc's own version of the loop (profiles/r05_head_gemm_v1_direct_utilisation.txt) ran at 31 % of the MFMA rate: 2.2 vector
instructions per MFMA of 64-bit address arithmetic and operand copies between the MFMAs, half of the prefetches sunk down to their
uses, every step ending on vmcnt(0); with hand-issued loads in C++ the register allocator spills the accumulators (500+ spills).
Here the registers are named:

    v[32:47], v[48:63]     strip operand sets 0 / 1 (4 x 16 bytes per lane)
    v[64:95], v[96:127]    weight operand sets 0 / 1 (8 x 16 bytes)
    v[128:255]             accumulators: tile t, operand row m -> v[128 + 16 t + 4 m .. + 3]

A k-step: [advance the two scalar bases] s_waitcnt vmcnt(0) (this step's operands, issued during the previous step, have landed), then
32 MFMAs with the 12 global_load_dwordx4 of the NEXT step -- into the other set -- spread between them, one behind every second or
third MFMA (weights first; the strip operands' 1 KB steps are immediates, the weights' tile offsets eight constant VGPRs).  No vector
instruction besides loads and MFMAs inside the loop.
A weight operand's 16 MFMA columns are TWO tiles' 8 slots (weights as ONE fp16 number each: the refine step's bound carries the 2^-11
relative rounding, bp_refine.h).  Epilogue per operand: the sums are scaled back and truncated; the lanes of columns 0..7 store 4
consecutive documents of their slot (16 bytes) to the even tile's array, the lanes of columns 8..15 to the odd tile's.
"""
import sys

OUT = sys.argv[1] if len(sys.argv) > 1 else "vsearch_amd/csrc/bp_head_asm.h"
VARIANT = sys.argv[2] if len(sys.argv) > 2 else ""      # experiments (WRONG results): noa / nob (one operand kind is not loaded), nomfma, nostore
A = [32, 48]
B = [64, 96]
ACC = 128
MA, NT = 4, 8
out = []
def emit(x): out.append(x)

def acc(t, m): return ACC + 16 * t + 4 * m
def r4(b): return f"v[{b}:{b + 3}]"

NLOADS = (0 if "nob" in VARIANT else NT) + (0 if "noa" in VARIANT else MA)
def loads(s):
    for t in range(NT):
        if "nob" not in VARIANT: emit(f"global_load_dwordx4 {r4(B[s] + 4 * t)}, %[bo{t}], %[bsg]")
    for m in range(MA):
        if "noa" not in VARIANT: emit(f"global_load_dwordx4 {r4(A[s] + 4 * m)}, %[l16], %[asg]" + (f" offset:{1024 * m}" if m else ""))

def mfmas(s):
    for t in range(NT):
        for m in range(MA):
            emit(f"v_mfma_f32_16x16x32_f16 {r4(acc(t, m))}, {r4(A[s] + 4 * m)}, {r4(B[s] + 4 * t)}, {r4(acc(t, m))}")

def advance():
    # the bases move on to the next k-step unless this is the last one (the loads then read the last operands again; never used)
    emit("s_cmp_gt_u32 %[n], 1")
    emit("s_cselect_b32 %[t0], %[astep], 0")
    emit("s_cselect_b32 %[t1], 0x400, 0")
    emit("s_add_u32 %[asg0], %[asg0], %[t0]")
    emit("s_addc_u32 %[asg1], %[asg1], 0")
    emit("s_add_u32 %[bsg0], %[bsg0], %[t1]")
    emit("s_addc_u32 %[bsg1], %[bsg1], 0")

# accumulators <- 0
for r in range(ACC, ACC + 16 * NT):
    emit(f"v_mov_b32 v{r}, 0")
loads(0)
emit("1:")
def step(s):
    """this step's MFMAs with the next step's loads spread between them (a burst of 12 loads, then 32 MFMAs, measured 15 % slower than
    the compiler's interleaved schedule: the requests of all waves then arrive at the L1 together)"""
    global out
    keep = out
    out = []; loads(1 - s); ld = out
    out = []
    if "nomfma" not in VARIANT: mfmas(s)
    mf = out
    out = keep
    if "burst" in VARIANT or not mf:
        for x in ld: emit(x)
        emit(f"s_waitcnt vmcnt({NLOADS})")
        for x in mf: emit(x)
        return
    # the wait comes first: this step's operands were issued during the previous step's MFMAs, nothing younger is in flight yet
    gap = max(1, len(mf) // max(1, len(ld)))
    emit("s_waitcnt vmcnt(0)")
    li = 0
    for i, x in enumerate(mf):
        emit(x)
        if i % gap == gap - 1 and li < len(ld):
            emit(ld[li]); li += 1
    for x in ld[li:]: emit(x)

for s in (0, 1):
    advance()
    step(s)
    emit("s_sub_u32 %[n], %[n], 1")
    emit("s_cmp_eq_u32 %[n], 0")
    emit("s_cbranch_scc1 2f")
emit("s_branch 1b")
emit("2:")
emit("s_waitcnt vmcnt(0)")
# (the matrix core's last results are read by vector instructions next: the hardware does not interlock that -- 16 passes' worth of
#  wait states, once per item)
emit("s_nop 15")
emit("s_nop 15")
# epilogue: one weight operand = TWO tiles (MFMA columns 0..7: the even tile's 8 slots, 8..15: the odd tile's): scale back, truncate;
# the lanes of columns 0..7 store 4 consecutive documents of their slot to the even tile, then the others to the odd tile.
# Tiles past the pass's last (ns of them are stored) end it.
for t in range(NT):
    emit(f"s_cmp_le_u32 %[ns], {2 * t}")
    emit("s_cbranch_scc1 3f")
    for m in range(MA):
        for i in range(4):
            r = acc(t, m) + i
            emit(f"v_mul_f32 v{r}, %[mul], v{r}")
            emit(f"v_cvt_i32_f32 v{r}, v{r}")
    emit("s_mov_b64 %[sv], exec")
    emit("s_mov_b32 exec_lo, 0x00ff00ff")
    emit("s_mov_b32 exec_hi, 0x00ff00ff")
    for m in range(MA):
        if "nostore" not in VARIANT: emit(f"global_store_dwordx4 %[so], {r4(acc(t, m))}, %[osg]" + (f" offset:{512 * m}" if m else ""))
    emit("s_mov_b64 exec, %[sv]")
    emit("s_add_u32 %[osg0], %[osg0], %[os0]")
    emit("s_addc_u32 %[osg1], %[osg1], %[os1]")
    emit(f"s_cmp_le_u32 %[ns], {2 * t + 1}")
    emit("s_cbranch_scc1 3f")
    emit("s_mov_b32 exec_lo, 0xff00ff00")
    emit("s_mov_b32 exec_hi, 0xff00ff00")
    for m in range(MA):
        if "nostore" not in VARIANT: emit(f"global_store_dwordx4 %[so], {r4(acc(t, m))}, %[osg]" + (f" offset:{512 * m}" if m else ""))
    emit("s_mov_b64 exec, %[sv]")
    emit("s_add_u32 %[osg0], %[osg0], %[os0]")
    emit("s_addc_u32 %[osg1], %[osg1], %[os1]")
emit("3:")

# the three 64-bit bases live in NAMED SGPR pairs (inline asm cannot name the halves of a 64-bit operand): copied in at the start
SG = {"asg": 92, "bsg": 94, "osg": 96}
def fix(line):
    for k, r in SG.items():
        line = line.replace(f"%[{k}0]", f"s{r}").replace(f"%[{k}1]", f"s{r + 1}").replace(f"%[{k}]", f"s[{r}:{r + 1}]")
    return line
out = [f"s_mov_b64 s[{r}:{r + 1}], %[{k}_in]" for k, r in SG.items()] + [fix(x) for x in out]

body = "\\n\\t\"\n        \"".join(out)
clob = ", ".join(f'"v{r}"' for r in range(32, 256))
hdr = f'''// GENERATED by tools/gen_head_asm.py -- do not edit; the generator says what the statement does and why it is asm.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vs {{

// One work item of a wave of the head pre-pass: strip operands from `abase` (+ lane * 16; k-steps `astep` bytes apart), weight operands
// from `bbase` + boff[t] (k-steps 1 KB apart; operand t = tiles 2 t, 2 t + 1), `ks` k-steps; the sums of the first `n_store` tiles go to `obase` + `so` (per lane),
// tiles `ostride` bytes apart, the 4 document groups 512 bytes apart.  All addresses wave-uniform except boff / l16 / so.
__device__ __forceinline__ void head_item_asm(unsigned long long abase, unsigned long long bbase, uint32_t astep, const uint32_t (&boff)[8], uint32_t l16, uint32_t ks,
                                              unsigned long long obase, unsigned long long ostride, uint32_t so, uint32_t n_store, float head_mul) {{
    // (the 64-bit bases are copied into named SGPR pairs inside the statement: s[92:93], s[94:95], s[96:97])
    uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane(ks), ns = (uint32_t)__builtin_amdgcn_readfirstlane(n_store);
    uint32_t t0, t1;
    unsigned long long sv;
    auto sg64 = [](unsigned long long v) {{
        return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
    }};
    const unsigned long long asg = sg64(abase), bsg = sg64(bbase), osg = sg64(obase);
    const uint32_t os0 = (uint32_t)ostride, os1 = (uint32_t)(ostride >> 32);
    asm volatile(
        "{body}\\n\\t"
        : [n] "+s"(n), [t0] "=&s"(t0), [t1] "=&s"(t1), [sv] "=&s"(sv)
        : [asg_in] "s"(asg), [bsg_in] "s"(bsg), [osg_in] "s"(osg), [astep] "s"(astep), [bo0] "v"(boff[0]), [bo1] "v"(boff[1]), [bo2] "v"(boff[2]), [bo3] "v"(boff[3]), [bo4] "v"(boff[4]), [bo5] "v"(boff[5]),
          [bo6] "v"(boff[6]), [bo7] "v"(boff[7]), [l16] "v"(l16), [so] "v"(so), [ns] "s"(ns), [mul] "s"(head_mul), [os0] "s"(os0), [os1] "s"(os1)
        : "memory", "scc", "s92", "s93", "s94", "s95", "s96", "s97", {clob});
}}

}}  // namespace vs
'''
open(OUT, "w").write(hdr)
print(f"{OUT}: {len(out)} instructions")
