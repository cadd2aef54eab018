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



# ---- the product with both operands through LDS (round 6) -----------------------------------------------------------------------------
# A workgroup = 2 x 2 wide waves = 256 documents x 32 tiles; a wave (wd, wt) multiplies document groups 8 wd .. 8 wd + 7 with weight
# operands 8 wt .. 8 wt + 7.  Per k-step the WORKGROUP loads 16 strip operands + 16 weight operands (32 KB) instead of every wave its
# own 16 (64 KB) -- and, what counts (measured: a k-step of the direct kernel takes ~ 3 000 cycles whatever it loads, the latency of a
# load from HBM under this load, against 1 024 cycles of MFMAs: ONE k-step of loads in flight is the bound) -- FOUR k-steps are in flight:
#   LDS = a ring of five k-step slots of 32 KB = [A 16 KB | B 16 KB] (all 160 KB); wave w brings document groups 4 w .. 4 w + 3 and weight
#   operands 4 w .. 4 w + 3 of a k-step by LDS-DMA (global_load_lds_dwordx4: M0 = the 1 KB piece's LDS address, written right before;
#   no staging registers, no ds_write).
#   k-step c (operand set c % 2):  lgkmcnt(0) | MFMAs(c) with, spread between them: vmcnt(16) + barrier (every wave's pieces of k-step
#   c + 1 have landed: the two younger k-steps' 16 loads may be in flight) -> ds_read set (c + 1) % 2 <- slot (c + 1) % 5 -> the
#   LDS-DMA of k-step c + 4 into slot (c + 4) % 5 = (c - 1) % 5 (its last reads were waited for before k-step c - 1's MFMAs; every wave
#   has passed this k-step's barrier since).
# The loop is unrolled over 10 k-steps (2 operand sets x 5 slots); the k-step counter is tested after each.
LSG = {"asg": 88, "bsg": 90, "osg": 92}
LA, LB = [32, 96], [64, 128]      # operand sets: A v[32:63] / v[96:127], B v[64:95] / v[128:159]
LEAD = 3 if "lead3" in VARIANT else 4      # k-steps of DMA in flight (experiment: lead3)
NSLOT = LEAD + 1


def gen_lds():
    out = []
    emit = out.append
    MA = 8

    def acc(t, m): return 4 * (MA * t + m)
    def accr(t, m): return f"a[{acc(t, m)}:{acc(t, m) + 3}]"

    def dma(slot):
        # the wave's 8 pieces of the k-step the bases point at -> ring slot `slot`; then the bases move on (unless that was the last k-step:
        # the loads then read it again, into a slot nobody reads any more)
        l = []
        for ab in (0, 1):
            for i in range(4):
                l.append(f"s_add_u32 m0, %[lws], {slot * 32768 + ab * 16384 + i * 1024}")
                l.append("s_nop 0")
                if "nodma" not in VARIANT: l.append(f"global_load_lds_dwordx4 %[{'ab'[ab]}o{i}], " + ("%[asg]" if ab == 0 else "%[bsg]"))
        l += ["s_cmp_gt_u32 %[nl], 1", "s_cselect_b32 %[t0], %[astep], 0", "s_cselect_b32 %[t1], 0x400, 0",
              "s_add_u32 %[asg0], %[asg0], %[t0]", "s_addc_u32 %[asg1], %[asg1], 0", "s_add_u32 %[bsg0], %[bsg0], %[t1]", "s_addc_u32 %[bsg1], %[bsg1], 0",
              "s_cmp_gt_u32 %[nl], 1", "s_cselect_b32 %[t0], 1, 0", "s_sub_u32 %[nl], %[nl], %[t0]"]
        return l

    def lreads(slot, s):
        # operand set s <- ring slot: A groups 8 wd + m (%[lra*] = wd * 8192 + lane * 16 [+ 64 KB, + 128 KB]), B operands 8 wt + t (%[lrb*])
        l = []
        for m in range(MA):
            off = slot * 32768 + m * 1024
            l.append(f"ds_read_b128 {r4(LA[s] + 4 * m)}, %[lra{off >> 16}] offset:{off & 65535}")
        for t in range(NT):
            off = slot * 32768 + 16384 + t * 1024
            l.append(f"ds_read_b128 {r4(LB[s] + 4 * t)}, %[lrb{off >> 16}] offset:{off & 65535}")
        if "noread" in VARIANT: l = ["s_nop 0"] * len(l)
        return l

    def mfmas(s):
        if "nomfma" in VARIANT: return ["s_nop 0"] * (NT * MA)
        return [f"v_mfma_f32_16x16x32_f16 {accr(t, m)}, {r4(LA[s] + 4 * m)}, {r4(LB[s] + 4 * t)}, {accr(t, m)}" for t in range(NT) for m in range(MA)]

    emit("s_mov_b32 %[m0s], m0")
    for k, r in LSG.items():
        emit(f"s_mov_b64 s[{r}:{r + 1}], %[{k}_in]")
    for r in range(256):
        emit(f"v_accvgpr_write_b32 a{r}, 0")
    # prologue: k-steps 0 .. 3 on their way, operand set 0 <- slot 0
    for c in range(LEAD):
        out.extend(dma(c))
    emit(f"s_waitcnt vmcnt({8 * (LEAD - 1)})")
    emit("s_barrier")
    out.extend(lreads(0, 0))
    emit("1:")
    for c in range(2 * NSLOT if NSLOT % 2 else NSLOT):
        emit("s_waitcnt lgkmcnt(0)")
        mf = mfmas(c % 2)
        others = [f"s_waitcnt vmcnt({8 * (LEAD - 2)})" + ("" if "nobar" in VARIANT else "\n\ts_barrier")] + lreads((c + 1) % NSLOT, (c + 1) % 2)
        d = dma((c + LEAD) % NSLOT)
        # 4 MFMAs, the wait + barrier, then a read every 2 MFMAs (16 reads), then the DMA pieces (3 instructions each) every 3 MFMAs
        seq = []
        mi = 0
        def take(n):
            nonlocal mi
            seq.extend(mf[mi:mi + n]); mi += n
        take(4)
        seq.append(others[0])
        for x in others[1:]:
            take(2); seq.append(x)
        dgroups = [d[3 * i:3 * i + 3] for i in range(8)] + [d[24:]]
        for g in dgroups:
            take(3); seq.extend(g)
        take(len(mf) - mi)
        out.extend(seq)
        emit("s_sub_u32 %[n], %[n], 1")
        emit("s_cmp_eq_u32 %[n], 0")
        emit("s_cbranch_scc1 2f")
    emit("s_branch 1b")
    emit("2:")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    emit("s_barrier")                                         # (no DMA of this item lands in the next item's slots)
    emit("s_nop 15")
    emit("s_nop 15")
    # epilogue: as head_item_asm_wide (the sums leave the AGPRs through operand set 0's registers)
    for t in range(NT):
        emit(f"s_cmp_le_u32 %[ns], {2 * t}")
        emit("s_cbranch_scc1 3f")
        regs = []
        for m in range(MA):
            for i in range(4):
                v = LA[0] + 4 * m + i
                emit(f"v_accvgpr_read_b32 v{v}, a{acc(t, m) + i}")
                regs.append(v)
        emit("s_nop 1")
        for v in regs:
            emit(f"v_mul_f32 v{v}, %[mul], v{v}")
            emit(f"v_cvt_i32_f32 v{v}, v{v}")
        for m in range(MA):
            emit(f"v_cvt_pk_u16_u32 v{regs[4 * m]}, v{regs[4 * m]}, v{regs[4 * m + 1]}")
            emit(f"v_cvt_pk_u16_u32 v{regs[4 * m + 1]}, v{regs[4 * m + 2]}, v{regs[4 * m + 3]}")
        emit("s_mov_b64 %[sv], exec")
        for half_ in (0, 1):
            if half_:
                emit(f"s_cmp_le_u32 %[ns], {2 * t + 1}")
                emit("s_cbranch_scc1 3f")
            mask = "0xff00ff00" if half_ else "0x00ff00ff"
            emit(f"s_mov_b32 exec_lo, {mask}")
            emit(f"s_mov_b32 exec_hi, {mask}")
            for m in range(MA):
                # (document groups past the block's last are not stored: md = groups of this wave inside the block)
                emit(f"s_cmp_le_u32 %[md], {m}")
                emit(f"s_cbranch_scc1 4{t}{half_}f")
                emit(f"global_store_dwordx2 %[so], v[{regs[4 * m]}:{regs[4 * m] + 1}], %[osg]" + (f" offset:{256 * m}" if m else ""))
            emit(f"4{t}{half_}:")
            emit("s_mov_b64 exec, %[sv]")
            emit("s_add_u32 %[osg0], %[osg0], %[os0]")
            emit("s_addc_u32 %[osg1], %[osg1], %[os1]")
    emit("3:")
    emit("s_mov_b32 m0, %[m0s]")

    def fix(line):
        for k, r in LSG.items():
            line = line.replace(f"%[{k}0]", f"s{r}").replace(f"%[{k}1]", f"s{r + 1}").replace(f"%[{k}]", f"s[{r}:{r + 1}]")
        return line
    out = [fix(x) for x in out]
    clob = [f'"v{r}"' for r in range(32, 160)] + [f'"a{r}"' for r in range(256)]
    return out, clob


def func_lds():
    lines, clob = gen_lds()
    body = "\\n\\t\"\n        \"".join(x.replace("\n\t", "\\n\\t") for x in lines)
    sregs = ", ".join(f'"s{r}"' for r in range(88, 94))
    return f"""// One work item of a WAVE of the LDS-tiled head product (tools/gen_head_asm.py, "both operands through LDS"): the four waves of a
// workgroup run it together (barriers inside: the same `ks` for all; LDS addresses 0 .. 160 KB - 1 are the ring).  abase / bbase / astep
// as in head_item_asm (abase = the strip of the block's first k-step, document group 0); aoff[i] / boff[i] = byte offsets (+ lane * 16)
// of the strip groups / weight operands this wave BRINGS (4 each), lws = wave * 4096 (where: scalar), lra / lrb = wd * 8192 + lane * 16 /
// wt * 8192 + lane * 16 (the operands it multiplies); n_store tiles x n_docs16 document groups of its sums are stored (obase / ostride /
// so as in head_item_asm).
__device__ __forceinline__ void head_item_asm_lds(unsigned long long abase, unsigned long long bbase, uint32_t astep, const uint32_t (&aoff)[4], const uint32_t (&boff)[4],
                                                  uint32_t lws, uint32_t lra, uint32_t lrb, uint32_t ks, unsigned long long obase, unsigned long long ostride, uint32_t so,
                                                  uint32_t n_store, uint32_t n_docs16, float head_mul) {{
    uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane(ks), nl = n, ns = (uint32_t)__builtin_amdgcn_readfirstlane(n_store), md = (uint32_t)__builtin_amdgcn_readfirstlane(n_docs16);
    uint32_t t0, t1, m0s;
    unsigned long long sv;
    auto sg64 = [](unsigned long long v) {{
        return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
    }};
    const unsigned long long asg = sg64(abase), bsg = sg64(bbase), osg = sg64(obase);
    const uint32_t os0 = (uint32_t)ostride, os1 = (uint32_t)(ostride >> 32), lws_s = (uint32_t)__builtin_amdgcn_readfirstlane(lws);
    const uint32_t lra0 = lra, lra1 = lra + 65536u, lra2 = lra + 131072u, lrb0 = lrb, lrb1 = lrb + 65536u, lrb2 = lrb + 131072u;      // (a DS offset ends at 65535)
    asm volatile(
        "{body}\\n\\t"
        : [n] "+s"(n), [nl] "+s"(nl), [t0] "=&s"(t0), [t1] "=&s"(t1), [sv] "=&s"(sv), [m0s] "=&s"(m0s)
        : [asg_in] "s"(asg), [bsg_in] "s"(bsg), [osg_in] "s"(osg), [astep] "s"(astep), [ao0] "v"(aoff[0]), [ao1] "v"(aoff[1]), [ao2] "v"(aoff[2]), [ao3] "v"(aoff[3]),
          [bo0] "v"(boff[0]), [bo1] "v"(boff[1]), [bo2] "v"(boff[2]), [bo3] "v"(boff[3]), [lws] "s"(lws_s), [lra0] "v"(lra0), [lra1] "v"(lra1), [lra2] "v"(lra2),
          [lrb0] "v"(lrb0), [lrb1] "v"(lrb1), [lrb2] "v"(lrb2), [so] "v"(so), [ns] "s"(ns), [md] "s"(md), [mul] "s"(head_mul), [os0] "s"(os0), [os1] "s"(os1)
        : "memory", "scc", SREGS, CLOB);
}}
""".replace("SREGS", sregs).replace("CLOB", ", ".join(clob))


hdr = '''// GENERATED by tools/gen_head_asm.py -- do not edit; the generator says what the statements do and why they are asm.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vs {

''' + func("head_item_asm", 4, "One work item of a wave of the head pre-pass: 64 documents x 16 tiles, accumulators in VGPRs (two waves per SIMD)") + "\n" + \
      func("head_item_asm_wide", 8, "The same for 128 documents x 16 tiles, accumulators in AGPRs (one wave per SIMD: 512 registers)") + "\n" + \
      func_lds() + '''
}  // namespace vs
'''
open(OUT, "w").write(hdr)
print(f"{OUT}: {len(gen(4)[0])} + {len(gen(8)[0])} + {len(gen_lds()[0])} instructions")
