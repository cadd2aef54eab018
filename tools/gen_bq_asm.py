#!/usr/bin/env python3
"""Generates vsearch_amd/csrc/bp_bq_asm.h: the inner loop of the bag-of-token chunk walk (bp_bq.h) as ONE inline-asm statement --
the quad walk's loop (tools/gen_quad_asm.py) re-cut for a BINARY index: short lists, no values.

Data (bp_bq.h): a list of a (block, column) is a CHUNK of GL x DW dwords = 2 GL DW cells of uint16 (document in the block; pad cells point
at the spare documents behind a slot's plane); the main chunk of column c is chunk c of its block.  A DESCRIPTOR (8 bytes, LDS) = one
chunk for one query slot: x = chunk index | LDS byte address of the slot's plane << 16, y = the query's integer weight.
A wave STEP = 64 / GL descriptors, one per GL-lane group: lane i of a group loads DW dwords of the chunk (cells 2 DW i ...) with ONE
global_load and adds the weight at every document: per step 1 ds_read_b64 + 1 global_load + 2 DW x (v_mad_u32_u16, ds_add_u32) + 3
address instructions.  Shapes (argv[3] = lanes x dwords), 21 M docs x 1024 queries of 776 tokens, 86 tokens a document (profiles/r05_bq_shapes.txt):
    8x1   16 cells = 32-byte chunks of 2048-document blocks, 8 query slots, S = 4: round 5's first form, 50.9 ms.  What it waits for is
          the L1's requests to L2: ~ 0.2 requests of 64 bytes a clock and CU (profiles/r05_bq_utilisation.txt: 0.182) whatever they
          carry -- the rate the quad walk and the head pre-pass product sit at too -- and a 32-byte chunk of ~6 postings is a request.
    16x1  32 cells = 64-byte chunks, blocks of 6144 documents (18 postings a list), 2 query slots, S = 8: 36.3 - 40 ms (THE DEFAULT).
          A request now carries three times the postings; S = 4: 41.8, 6: 41.7, 12: 38.6, 16: 41.1.
    8x2   the same chunks, 8 lists a step with global_load_dwordx2: 40.3 ms (S = 7; 4: 43.3, 9: 40.8, 12: 41.7).
    4x4   the same chunks, 16 lists a step with global_load_dwordx4: 55.3 ms -- the wider the load the slower (a wave's step is 16
          requests either way; the texture path takes a dwordx4 of 4-lane groups apart lane by lane).
    32x1  64 cells = 128-byte chunks of 8192-document blocks: 48.8 ms (twice the bytes and the LDS adds of 16x1 for 4/3 the postings).
    Blocks of 5120 / 7168 documents at 16x1: 41.5 / 41.3 ms (7168: 21 postings a list, 1 list in 60 links to an overflow chunk).
The loop keeps S - 1 loads and one descriptor read of a wave in flight with counted vmcnt / lgkmcnt, as the quad loop does (register
sets {descriptor 2 VGPRs, postings DW VGPRs}).

LINKS.  A list longer than its chunk keeps all cells but the last for postings; the last cell -- the high half of the group's last
dword -- has bit 15 set and carries the index of the block's overflow chunk that continues the list.  After a step's postings have
landed one v_cmp + s_cbranch asks whether any lane holds a link; the lanes that do append a descriptor {overflow chunk | plane, weight}
to the WAVE's list (SGPR count + v_mbcnt rank, no atomic) and sit out the add of that cell.  Three statements as for the quad walk:
bq_walk_asm (the workgroup's table: a wave takes steps w, w + 16, ...), bq_list_asm (the wave's own list), bq_collect_asm (links only:
the segment mode when a list overflows).

(Tried on the 8x1 shape and dropped: pad cells sitting out their ds_add behind a v_cmpx_gt_u16_sdwa -- 51.9 ms against 51.0,
profiles/r05_bq_mask.txt: the LDS conflicts are not what the walk waits for.)
"""
import sys

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
OUT = sys.argv[2] if len(sys.argv) > 2 else "vsearch_amd/csrc/bp_bq_asm.h"
SHAPE = sys.argv[3] if len(sys.argv) > 3 else "16x1"
GL, DW = (int(x) for x in SHAPE.split("x"))                 # lanes per list, dwords per lane
LPS = 64 // GL                                               # lists (descriptors) of a wave step
CHUNK = GL * DW * 4                                          # bytes of a chunk
SHIFT = CHUNK.bit_length() - 1
NW = 16
STEP = NW * LPS * 8          # bytes between a wave's consecutive steps of the table; own list: LPS * 8
V0 = 64
out = []
def emit(x): out.append(x)
COLLECT = False

# (64-bit VGPR tuples must start on an even register, 128-bit ones on a multiple of... the assembler wants even: the descriptor pairs
#  first, then the postings (DW registers a set, sets on multiples of 4 when DW = 4), then an even-aligned pair of temporaries that
#  doubles as the link descriptor a ds_write_b64 stores)
def d(i, j): return f"v{V0 + 2 * i + j}"
P0 = (V0 + 2 * S + 3) & ~3
def p(i, k=0): return f"v{P0 + DW * i + k}"
def prange(i): return f"v[{P0 + DW * i}:{P0 + DW * i + DW - 1}]" if DW > 1 else p(i)
def plast(i): return p(i, DW - 1)                            # the dword whose high half is the chunk's last cell
def drange(i): return f"v[{V0 + 2 * i}:{V0 + 2 * i + 1}]"
T0 = (P0 + DW * S + 1) & ~1
A0, A1, SO, VOFF, T2, VATOM, VCADDR, VONE = (f"v{T0 + k}" for k in range(8))
VEND = T0 + 8
NADD = 2 * DW                                                # ds_add per step

def load(j):
    emit(f"v_and_b32 {VOFF}, 0xffff, {d(j, 0)}")
    emit(f"v_lshl_or_b32 {VOFF}, {VOFF}, {SHIFT}, %[l4]")
    emit(f"global_load_dword{'x' + str(DW) if DW > 1 else ''} {prange(j)}, {VOFF}, %[base]")

def adds(i, skip_last_for_links=False):
    """the 2 DW cells of set i: address = plane + 4 * document, data = the weight.  (the address registers A0 / A1 are free again once
    their ds_add has issued)"""
    for k in range(DW):
        emit(f"v_mad_u32_u16 {A0}, {p(i, k)}, 4, {SO}")
        emit(f"v_mad_u32_u16 {A1}, {p(i, k)}, 4, {SO} op_sel:[1,0,0,0]")
        emit(f"ds_add_u32 {A0}, {d(i, 1)}")
        if skip_last_for_links and k == DW - 1:
            emit(f"v_cmp_le_i32 vcc, 0, {plast(i)}")               # lanes WITHOUT a link
            emit("s_and_saveexec_b64 %[sv], vcc")
            emit(f"ds_add_u32 {A1}, {d(i, 1)}")
            emit("s_mov_b64 exec, %[sv]")
        else:
            emit(f"ds_add_u32 {A1}, {d(i, 1)}")

def append_links(i):
    """lanes of set i whose high cell is a link (vcc) append {overflow chunk | plane, weight} to the wave's list"""
    emit("s_bcnt1_i32_b64 %[st], vcc")
    emit(f"v_mbcnt_lo_u32_b32 {T2}, vcc_lo, 0")
    emit(f"v_mbcnt_hi_u32_b32 {T2}, vcc_hi, {T2}")
    emit("s_and_saveexec_b64 %[sv], vcc")
    emit(f"v_add_u32 {T2}, %[cnt], {T2}")                          # number in the wave's list
    emit(f"v_bfe_u32 {A0}, {plast(i)}, 16, 15")                        # overflow chunk of the block (the link's payload)
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
        emit(f"v_cmp_gt_i32 vcc, 0, {plast(i)}")
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
            adds(i)
            emit(f"6{i}:")
            # set i is consumed: its descriptor registers take the descriptor of step t + S
            emit(f"ds_read_b64 {drange(i)}, %[dptr] offset:{i * STEP}")
            # the descriptor read of the previous trip (set j) is back: behind it were issued this trip's ds_adds + ds_read
            emit(f"s_waitcnt lgkmcnt({NADD + 1})")
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
            adds(i, skip_last_for_links=True)
            emit(f"s_branch 6{i}b")
    emit("8:")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return out

def build_dyn():
    """The table walk with the steps dealt DYNAMICALLY (round 6): a batch = S consecutive steps (S x LPS consecutive table entries); a wave
    starts with batches w and w + 16 and takes every further one from a counter in LDS -- one ds_add_rtn_u32 per batch by lane 0, issued
    a whole pass before its value is read (LDS operations complete in order: by then it is back), so nothing waits for it.  What the
    static deal (steps w, w + 16, ...) cost: the waves' memory luck differs, and with ~ 100 steps a wave and block the slowest wave kept
    the other fifteen at the block's barrier for a quarter of the walk (3.5 k of 13 k cycles on the packed bag-of-token walk).
    The pipeline is the static loop's: during a pass over batch cur the descriptors of batch nxt are read and its first S - 1 loads
    issued; at a pass's end nxt becomes cur, the counter's answer becomes nxt, the next request goes out.  The statement leaves when cur
    is beyond the table -- or when the wave's link list has no room for another batch's worst case (the caller walks the list and comes
    back with cur / nxt / the pending answer: the segment mode of the static walk is not needed).  Batches beyond the table read the
    null descriptors behind it."""
    global out
    out = []
    BATCH = S * LPS * 8
    def dptr_of(breg):
        emit(f"s_min_u32 %[t], {breg}, %[nb]")
        emit(f"s_mul_i32 %[t], %[t], {BATCH}")
        emit(f"v_add_u32 %[dptr], %[t], %[tbv]")
    def request():
        emit("s_mov_b64 %[sv], exec")
        emit("s_mov_b64 exec, 1")
        emit(f"ds_add_rtn_u32 {VATOM}, {VCADDR}, {VONE}")
        emit("s_mov_b64 exec, %[sv]")
    emit(f"v_mov_b32 {VCADDR}, %[caddr]")
    emit(f"v_mov_b32 {VONE}, 1")
    dptr_of("%[cur]")
    for k in range(S):
        emit(f"ds_read_b64 {drange(k)}, %[dptr] offset:{k * LPS * 8}")
    dptr_of("%[nxt]")
    emit("s_cmp_eq_u32 %[pend], -1")
    emit("s_cbranch_scc0 3f")
    request()
    emit("s_branch 4f")
    emit("3:")
    emit(f"v_mov_b32 {VATOM}, %[pend]")
    emit("4:")
    emit("s_waitcnt lgkmcnt(0)")
    for k in range(S - 1):
        load(k)
    emit("1:")
    for i in range(S):
        j = (i - 1) % S
        emit(f"s_waitcnt vmcnt({S - 2})")
        emit(f"v_cmp_gt_i32 vcc, 0, {plast(i)}")
        emit(f"s_cbranch_vccnz 7{i}f")
        emit(f"v_lshrrev_b32 {SO}, 16, {d(i, 0)}")
        adds(i)
        emit(f"6{i}:")
        emit(f"ds_read_b64 {drange(i)}, %[dptr] offset:{i * LPS * 8}")
        emit(f"s_waitcnt lgkmcnt({NADD + 1})")
        load(j)
    # the pass's end
    emit(f"v_readfirstlane_b32 %[t], {VATOM}")
    emit("s_mov_b32 %[cur], %[nxt]")
    emit("s_mov_b32 %[nxt], %[t]")
    dptr_of("%[nxt]")
    request()
    emit("s_cmp_ge_u32 %[cur], %[nb]")
    emit("s_cbranch_scc1 8f")
    emit("s_cmp_gt_u32 %[cnt], %[room]")
    emit("s_cbranch_scc1 8f")
    emit("s_branch 1b")
    for i in range(S):
        emit(f"7{i}:")
        append_links(i)
        emit(f"v_lshrrev_b32 {SO}, 16, {d(i, 0)}")
        adds(i, skip_last_for_links=True)
        emit(f"s_branch 6{i}b")
    emit("8:")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    emit(f"v_readfirstlane_b32 %[pend], {VATOM}")
    return out

COLLECT = False
body_add = build()
STEP = LPS * 8
body_list = build()
STEP = NW * LPS * 8
COLLECT = True
body_collect = build()
COLLECT = False
body_dyn = build_dyn()
vregs = [f"v{r}" for r in range(V0, VEND)]
def stmt(lines): return "\\n\\t\"\n        \"".join(lines)
clob = ", ".join(f'"{r}"' for r in vregs)
def fn(name, lines, what):
    return f'''// {what}
// dptr: LDS byte address of this lane group's descriptor of the wave's first step; trips >= 1: steps of this wave; base: the block's
// first chunk (wave-uniform); l4: byte offset of the lane's dwords inside a chunk; ncols: chunks before the block's overflow chunks.
// Slot planes from LDS address 0.
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

def fn_dyn(name, lines):
    return f'''// walk the workgroup's descriptor table with the batches of {S} steps dealt from a counter in LDS (build_dyn in the generator says how and why)
// tbv: LDS byte address of the table + this lane group's 8 x (lane / {GL}); nb: batches of the table ({S * LPS} entries each; the table is
// followed by a batch of null descriptors); cur / nxt: the wave's current and next batch (first call of a block: w, w + 16), pend: the
// counter's pending answer (-1: none yet); caddr: LDS byte address of the counter (32 at a block's start).  Links go to the wave's list at
// lbase (capacity cap >= {LPS * S}); returns their number.  Done when cur >= nb on return; else the list wants walking: call again with the same cur / nxt / pend.
__device__ __forceinline__ uint32_t {name}(uint32_t tbv, uint32_t nb, uint32_t& cur, uint32_t& nxt, uint32_t& pend, uint32_t caddr, const char* base, uint32_t l4, uint32_t ncols,
                                           uint32_t lbase, uint32_t cap) {{
    const unsigned long long pb = (unsigned long long)base;
    const unsigned long long ub = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32)) << 32) |
                                  (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pb);
    const uint32_t lb = (uint32_t)__builtin_amdgcn_readfirstlane(lbase), cp = (uint32_t)__builtin_amdgcn_readfirstlane(cap);
    const uint32_t nc = (uint32_t)__builtin_amdgcn_readfirstlane(ncols), nbs = (uint32_t)__builtin_amdgcn_readfirstlane(nb);
    const uint32_t ca = (uint32_t)__builtin_amdgcn_readfirstlane(caddr), room = cp - {LPS * S}u;
    uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane(cur), x = (uint32_t)__builtin_amdgcn_readfirstlane(nxt), pd = (uint32_t)__builtin_amdgcn_readfirstlane(pend);
    uint32_t cnt = 0, st, t, dptr;
    unsigned long long sv;
    asm volatile(
        "{stmt(lines)}\\n\\t"
        : [cur] "+s"(c), [nxt] "+s"(x), [pend] "+s"(pd), [cnt] "+s"(cnt), [st] "=&s"(st), [t] "=&s"(t), [sv] "=&s"(sv), [dptr] "=&v"(dptr)
        : [base] "s"(ub), [l4] "v"(l4), [tbv] "v"(tbv), [ncols] "s"(nc), [lbase] "s"(lb), [cap] "s"(cp), [nb] "s"(nbs), [caddr] "s"(ca), [room] "s"(room)
        : "memory", "scc", "vcc", {clob});
    cur = c; nxt = x; pend = pd;
    return cnt;
}}
'''

hdr = f'''// GENERATED by tools/gen_bq_asm.py {S} {OUT} {SHAPE} -- do not edit; the generator says what the statements do and why they are asm.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vs {{

constexpr int kBqSets = {S};                  // register sets: kBqSets - 1 chunk loads of a wave in flight
constexpr int kBqOverRead = {S};              // steps a wave reads descriptors of beyond its last one: a table ends with 16 x that many null steps
constexpr int kBqGroupLanes = {GL}, kBqLaneDwords = {DW};       // lanes per list, dwords a lane loads: a chunk = {CHUNK} bytes = {2 * GL * DW} cells

''' + fn("bq_walk_asm", body_add, "walk the workgroup's descriptor table (a wave takes steps w, w + 16, ...): add its chunks' postings, collect their links") + "\n" + \
      fn("bq_list_asm", body_list, "the same over the wave's OWN list (consecutive steps)") + "\n" + \
      fn("bq_collect_asm", body_collect, "collect the links of the wave's steps of the workgroup's table, add nothing") + "\n" + fn_dyn("bq_dyn_asm", body_dyn) + '''
}  // namespace vs
'''
open(OUT, "w").write(hdr)
print(f"{OUT}: S {S} shape {SHAPE}, VGPRs v{V0}..v{VEND - 1}, {len(body_add)} + {len(body_list)} + {len(body_collect)} instructions")
