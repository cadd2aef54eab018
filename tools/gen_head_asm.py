#!/usr/bin/env python3
"""Generates vsearch_amd/csrc/bp_head_asm.h: one work item of a wave of the head pre-pass (bp_head.h) as ONE inline-asm statement --
MA strip operands (16 documents each) x 8 weight operands (two query tiles each) per k-step, all k-steps, and the conversion + store
of the sums (16-bit: units of 2^14 of the walk's fixed point).  Two shapes:

    head_item_asm       MA = 4:  64 documents x 16 tiles, 128 accumulators in VGPRs, two waves per SIMD (512-thread workgroups)
    head_item_asm_wide  MA = 8: 128 documents x 16 tiles, 256 accumulators in AGPRs, one wave per SIMD (256-thread workgroups):
                                 16 KB of operands per 64 MFMAs instead of 12 KB per 32 -- the product is bound by operand traffic
                                 (profiles/r05_head_gemm_ablation.txt: with the MFMAs REMOVED the kernel takes the same time)

Why asm: with two operand sets (the next step's loads in flight across this step's MFMAs) a wave uses 224 - 400 named registers.
hipcc's own version of the loop (profiles/r05_head_gemm_v1_direct_utilisation.txt) ran at 31 % of the MFMA rate: 2.2 vector
instructions per MFMA of 64-bit address arithmetic and operand copies between the MFMAs, half of the prefetches sunk down to their
uses, every step ending on vmcnt(0); with hand-issued loads in C++ the register allocator spills the accumulators (500+ spills).
Registers (MA = 4 / 8):

    v[32:47], v[48:63]   / v[32:63], v[64:95]     strip operand sets 0 / 1 (MA x 16 bytes per lane)
    v[64:95], v[96:127]  / v[96:127], v[128:159]  weight operand sets 0 / 1 (8 x 16 bytes)
    v[128:255]           / a[0:255]               accumulators: operand t, strip row m -> base + 4 (MA t + m) .. + 3

A k-step: [advance the two scalar bases] s_waitcnt vmcnt(0) (this step's operands, issued during the previous step, have landed), then
the MFMAs with the loads of the NEXT step -- into the other set -- spread between them (weights first; the strip operands' 1 KB steps
are immediates, the weights' tile offsets eight constant VGPRs; a burst of all loads ahead of the MFMAs measured 15 % slower: the
requests of all waves then arrive at the L1 together).  No vector instruction besides loads and MFMAs inside the loop.
A weight operand's 16 MFMA columns are TWO tiles' 8 slots (weights as ONE fp16 number each: the refine step's bound carries the 2^-11
relative rounding, bp_refine.h).  Epilogue per operand: the sums are scaled back and truncated; the lanes of columns 0..7 store 4
consecutive documents of their slot (16 bytes) to the even tile's array, the lanes of columns 8..15 to the odd tile's.
"""
import sys

OUT = sys.argv[1] if len(sys.argv) > 1 else "vsearch_amd/csrc/bp_head_asm.h"
VARIANT = sys.argv[2] if len(sys.argv) > 2 else ""      # experiments (WRONG results): noa / nob (one operand kind is not loaded), nomfma, nostore, burst
NT = 8
SG = {"asg": 92, "bsg": 94, "osg": 96}                   # the three 64-bit bases live in NAMED SGPR pairs (inline asm cannot name the halves of a 64-bit operand)


def r4(b): return f"v[{b}:{b + 3}]"


def gen(MA):
    wide = MA == 8
    A = [32, 32 + 4 * MA]
    B = [32 + 8 * MA, 32 + 8 * MA + 4 * NT]
    VEND = B[1] + 4 * NT                                  # first vector register behind the operand sets
    ACC = 0 if wide else VEND
    out = []
    emit = out.append

    def acc(t, m): return ACC + 4 * (MA * t + m)
    def accr(t, m): return (f"a[{acc(t, m)}:{acc(t, m) + 3}]" if wide else r4(acc(t, m)))

    nloads = (0 if "nob" in VARIANT else NT) + (0 if "noa" in VARIANT else MA)

    def loads(s):
        l = []
        for t in range(NT):
            if "nob" not in VARIANT: l.append(f"global_load_dwordx4 {r4(B[s] + 4 * t)}, %[bo{t}], %[bsg]")
        for m in range(MA):
            # (the immediate offset of a global load ends at 4095: the second four document groups use the lane offset + 4096)
            off, lane = (1024 * (m & 3), "%[l16]" if m < 4 else "%[l16b]")
            if "noa" not in VARIANT: l.append(f"global_load_dwordx4 {r4(A[s] + 4 * m)}, {lane}, %[asg]" + (f" offset:{off}" if off else ""))
        return l

    def mfmas(s):
        return [f"v_mfma_f32_16x16x32_f16 {accr(t, m)}, {r4(A[s] + 4 * m)}, {r4(B[s] + 4 * t)}, {accr(t, m)}" for t in range(NT) for m in range(MA)]

    def advance():
        # the bases move on to the next k-step unless this is the last one (the loads then read the last operands again; never used)
        emit("s_cmp_gt_u32 %[n], 1")
        emit("s_cselect_b32 %[t0], %[astep], 0")
        emit("s_cselect_b32 %[t1], 0x400, 0")
        emit("s_add_u32 %[asg0], %[asg0], %[t0]")
        emit("s_addc_u32 %[asg1], %[asg1], 0")
        emit("s_add_u32 %[bsg0], %[bsg0], %[t1]")
        emit("s_addc_u32 %[bsg1], %[bsg1], 0")

    def step(s):
        ld = loads(1 - s)
        mf = [] if "nomfma" in VARIANT else mfmas(s)
        if "burst" in VARIANT or not mf:
            out.extend(ld)
            emit(f"s_waitcnt vmcnt({nloads})")
            out.extend(mf)
            return
        # the wait comes first: this step's operands were issued during the previous step's MFMAs, nothing younger is in flight yet
        gap = max(1, len(mf) // max(1, len(ld)))
        emit("s_waitcnt vmcnt(0)")
        li = 0
        for i, x in enumerate(mf):
            emit(x)
            if i % gap == gap - 1 and li < len(ld):
                emit(ld[li]); li += 1
        out.extend(ld[li:])

    for k, r in SG.items():
        emit(f"s_mov_b64 s[{r}:{r + 1}], %[{k}_in]")
    for r in range(4 * MA * NT):
        emit(f"v_accvgpr_write_b32 a{r}, 0" if wide else f"v_mov_b32 v{ACC + r}, 0")
    out.extend(loads(0))
    emit("1:")
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
    # epilogue: operand by operand; tiles past the pass's last (ns of them are stored) end it.  (wide: the sums leave the AGPRs through
    # the strip operand registers, free now)
    for t in range(NT):
        emit(f"s_cmp_le_u32 %[ns], {2 * t}")
        emit("s_cbranch_scc1 3f")
        regs = []
        for m in range(MA):
            for i in range(4):
                if wide:
                    v = 32 + 4 * m + i
                    emit(f"v_accvgpr_read_b32 v{v}, a{acc(t, m) + i}")
                else:
                    v = acc(t, m) + i
                regs.append(v)
        if wide:
            emit("s_nop 1")
        for v in regs:
            emit(f"v_mul_f32 v{v}, %[mul], v{v}")
            emit(f"v_cvt_i32_f32 v{v}, v{v}")
        # 16-bit sums (round 6): `mul` carries the 2^-14 (a sum < 2^30 leaves below 2^16), a lane's 4 documents pack into 2 dwords
        for m in range(MA):
            emit(f"v_cvt_pk_u16_u32 v{regs[4 * m]}, v{regs[4 * m]}, v{regs[4 * m + 1]}")
            emit(f"v_cvt_pk_u16_u32 v{regs[4 * m + 1]}, v{regs[4 * m + 2]}, v{regs[4 * m + 3]}")
        emit("s_mov_b64 %[sv], exec")
        for half in (0, 1):
            if half:
                emit(f"s_cmp_le_u32 %[ns], {2 * t + 1}")
                emit("s_cbranch_scc1 3f")
            mask = "0xff00ff00" if half else "0x00ff00ff"
            emit(f"s_mov_b32 exec_lo, {mask}")
            emit(f"s_mov_b32 exec_hi, {mask}")
            for m in range(MA):
                if "nostore" not in VARIANT:
                    emit(f"global_store_dwordx2 %[so], v[{regs[4 * m]}:{regs[4 * m] + 1}], %[osg]" + (f" offset:{256 * m}" if m else ""))
            emit("s_mov_b64 exec, %[sv]")                 # (restored before every branch out of the epilogue)
            emit("s_add_u32 %[osg0], %[osg0], %[os0]")
            emit("s_addc_u32 %[osg1], %[osg1], %[os1]")
    emit("3:")

    def fix(line):
        for k, r in SG.items():
            line = line.replace(f"%[{k}0]", f"s{r}").replace(f"%[{k}1]", f"s{r + 1}").replace(f"%[{k}]", f"s[{r}:{r + 1}]")
        return line
    out = [fix(x) for x in out]
    clob = [f'"v{r}"' for r in range(32, VEND if wide else 256)] + ([f'"a{r}"' for r in range(256)] if wide else [])
    return out, clob


def func(name, MA, what):
    lines, clob = gen(MA)
    body = "\\n\\t\"\n        \"".join(lines)
    l16b = ", [l16b] \"v\"(l16b)" if MA == 8 else ""
    l16b_decl = "    const uint32_t l16b = l16 + 4096u;                     // document groups 4..7 (a global load's immediate offset ends at 4095)\n" if MA == 8 else ""
    return f'''// {what}
// strip operands from `abase` (+ lane * 16; k-steps `astep` bytes apart), weight operands from `bbase` + boff[t] (k-steps 1 KB apart;
// operand t = tiles 2 t, 2 t + 1), `ks` k-steps; the sums of the first `n_store` tiles go to `obase` + `so` (per lane), tiles
// `ostride` bytes apart, the document groups 256 bytes apart (uint16 sums: head_mul carries the 2^-14).  All addresses wave-uniform except boff / l16 / so.
__device__ __forceinline__ void {name}(unsigned long long abase, unsigned long long bbase, uint32_t astep, const uint32_t (&boff)[8], uint32_t l16, uint32_t ks,
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
{l16b_decl}    asm volatile(
        "{body}\\n\\t"
        : [n] "+s"(n), [t0] "=&s"(t0), [t1] "=&s"(t1), [sv] "=&s"(sv)
        : [asg_in] "s"(asg), [bsg_in] "s"(bsg), [osg_in] "s"(osg), [astep] "s"(astep), [bo0] "v"(boff[0]), [bo1] "v"(boff[1]), [bo2] "v"(boff[2]), [bo3] "v"(boff[3]), [bo4] "v"(boff[4]), [bo5] "v"(boff[5]),
          [bo6] "v"(boff[6]), [bo7] "v"(boff[7]), [l16] "v"(l16){l16b}, [so] "v"(so), [ns] "s"(ns), [mul] "s"(head_mul), [os0] "s"(os0), [os1] "s"(os1)
        : "memory", "scc", "s92", "s93", "s94", "s95", "s96", "s97", {", ".join(clob)});
}}
'''


hdr = '''// GENERATED by tools/gen_head_asm.py -- do not edit; the generator says what the statements do and why they are asm.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vs {

''' + func("head_item_asm", 4, "One work item of a wave of the head pre-pass: 64 documents x 16 tiles, accumulators in VGPRs (two waves per SIMD)") + "\n" + \
      func("head_item_asm_wide", 8, "The same for 128 documents x 16 tiles, accumulators in AGPRs (one wave per SIMD: 512 registers)") + '''
}  // namespace vs
'''
open(OUT, "w").write(hdr)
print(f"{OUT}: {len(gen(4)[0])} + {len(gen(8)[0])} instructions")
