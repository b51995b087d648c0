#!/usr/bin/env python3
"""Writes kx_sweeps.inc: the two fully unrolled, hand-scheduled gfx950 instruction sequences of the
engine's per-piece work, as inline-asm functions.

Both run TWO independent dependency chains per lane (the two 32-byte halves of a 64-byte piece), so
that one chain's LDS round trip hides behind the other's, and both issue the next dependent LDS read
before doing the bookkeeping of the current step.

  piece_forward2(w, hA, hB, himask) -> bo[32]
      forward re-derivation of the back rows: chain A = bytes 0..31 from state handle hA,
      chain B = bytes 32..63 from hB.  Per byte and chain: 1 SDWA byte extract (class address), 1 ds_read_u8
      (class*4), 1 SDWA add on the chain (next fwd address = e.lo16 + class*4), 1 ds_read_b32, 1 op
      packing the row offset (e.hi16) into bo.  The class table sits at LDS address 0.

  piece_sweep2(bo, w, leafA, oA, leafB, oB, jb, jlim)
      backward sweep that places the output: chain A = steps 63..32 from (leafA, oA), chain B = steps
      31..0 from (leafB, oB); see the step description in kx_engine.hip (k_emit).  jb (SGPR, in/out) is
      the LDS address of the wave's next free job slot, jlim that of the last slot.

  piece_sweep2i(...)
      the same sweep for programs in the INLINE-CONSTANT entry layout (DevTables::inl: a one-byte constant rides in byte 2
      of its entry): per step and chain one more address op and one more byte store, no job.  The cursors are kept one
      byte low (o' = o - 1) so that both stores reach their byte with an offset field.

  piece_sweep2j(bo, w, leafA, oA, leafB, oB, pA, pB)
      the same sweep for programs in the JOB-STRIDE entry layout (DevTables::jl; round 4): byte 2 of an entry is 4 where a
      constant follows and 0 elsewhere.  Every step stores its job word (cursor<<16 | entry address) at the lane's own
      list pointer — unconditionally, no vote, no exec games, no scalar bookkeeping — and then advances the pointer by
      byte 2: a step without a constant leaves garbage that the next step overwrites.  Chain A's list grows upward from
      pA, chain B's downward from pB (both in the lane's private slot region).  Per step and chain 6 VALU + 1 LDS read
      + 2 LDS writes against the vote-and-rank form's 6 VALU + ~5 VALU + ~6 SALU (measured: an instruction of ANY
      kind costs a SIMD ≈ 4.1 cycles in these kernels).

  piece_forward1 / piece_run1 / piece_walk1
      one-chain forms for k_backlen (whose two input pieces per trip leave no room for two chains) and
      k_forward: forward with / without recording the back rows, and the measuring backward walk.

The file is generated (python gen_sweeps.py > kx_sweeps.inc) and committed; build.py regenerates it
when this script is newer.
"""
import sys

SD = "dst_sel:DWORD dst_unused:UNUSED_PAD"


def fwd2():
    L = []
    ap = L.append
    # rotating registers: e{A,B}{0,1} fwd words, c{A,B}{0,1,2} classes (prefetch distance 2), x{A,B} class addresses
    def cls_issue(ch, t, slot):
        wi, by = t >> 2, t & 3
        ap("v_lshlrev_b32_sdwa %%[x%s], 0, %%[w%d] %s src0_sel:DWORD src1_sel:BYTE_%d" % (ch, wi, SD, by))
        ap("ds_read_u8 %%[c%s%d], %%[x%s]" % (ch, slot, ch))
    base = {"A": 0, "B": 32}
    for d in (0, 1):
        for ch in "AB":
            cls_issue(ch, base[ch] + d, d)
    ap("s_waitcnt lgkmcnt(2)")          # classes of step 0 (both chains) have arrived
    for ch in "AB":
        ap("v_add_u32 %%[a%s], %%[h%s], %%[c%s0]" % (ch, ch, ch))
        ap("ds_read_b32 %%[e%s0], %%[a%s]" % (ch, ch))
    for j in range(32):
        cur, nxt = j & 1, (j + 1) & 1
        if j + 2 < 32:
            for ch in "AB":
                cls_issue(ch, base[ch] + j + 2, (j + 2) % 3)
        ap("s_waitcnt lgkmcnt(%d)" % (2 if j + 2 < 32 else 0))   # e_j of both chains (and class j+1) are in
        if j + 1 < 32:
            for ch in "AB":
                ap("v_add_u32_sdwa %%[a%s], %%[e%s%d], %%[c%s%d] %s src0_sel:WORD_0 src1_sel:DWORD" % (ch, ch, cur, ch, (j + 1) % 3, SD))
                ap("ds_read_b32 %%[e%s%d], %%[a%s]" % (ch, nxt, ch))
        for ch in "AB":
            t = base[ch] + j
            if t & 1:
                ap("v_and_or_b32 %%[bo%d], %%[e%s%d], %%[hm], %%[bo%d]" % (t >> 1, ch, cur, t >> 1))
            else:
                ap("v_lshrrev_b32 %%[bo%d], 16, %%[e%s%d]" % (t >> 1, ch, cur))
    return L


def sweep2(inl=False):
    L = []
    ap = L.append
    row = lambda t: "%%[bo%d]" % (t >> 1)
    wsel = lambda t: "WORD_%d" % (t & 1)
    base = {"A": 32, "B": 0}
    # the sweep is the longest dependent stretch of a k_emit iteration: waves inside it get issue priority over
    # waves in the phases around it (measured: 9.87 -> 9.60 ms; the same around the forward re-derivation loses)
    ap("s_setprio 3")
    for ch in "AB":
        t = base[ch] + 31
        ap("v_add_u32_sdwa %%[a%s0], %s, %%[leaf%s] %s src0_sel:%s src1_sel:DWORD" % (ch, row(t), ch, SD, wsel(t)))
        ap("ds_read_b32 %%[e%s0], %%[a%s0]" % (ch, ch))
    for j in range(31, -1, -1):
        cur, nxt = (31 - j) & 1, (32 - j) & 1
        # the two entry reads are the oldest operations in flight; behind them sit the previous step's two byte
        # stores (always) and job stores (sometimes): waiting for "all but two" never waits for less than the reads
        # (inl: four byte stores per step, so "all but four" is still no less than the reads)
        ap("s_waitcnt lgkmcnt(%d)" % (0 if j == 31 else 4 if inl else 2))
        for ch in "AB":
            ap("v_and_b32 %%[leaf%s], 0x3fc, %%[e%s%d]" % (ch, ch, cur))
            if j > 0:
                t = base[ch] + j - 1
                ap("v_add_u32_sdwa %%[a%s%d], %s, %%[leaf%s] %s src0_sel:%s src1_sel:DWORD" % (ch, nxt, row(t), ch, SD, wsel(t)))
                ap("ds_read_b32 %%[e%s%d], %%[a%s%d]" % (ch, nxt, ch, nxt))
        for ch in "AB":
            t = base[ch] + j
            e, o = "%%[e%s%d]" % (ch, cur), "%%[o%s]" % ch
            by = t & 3
            if by == 3:
                ap("v_lshrrev_b32 %%[tw%s], 8, %%[w%d]" % (ch, t >> 2))
            src = "%%[tw%s]" % ch if by in (1, 3) else "%%[w%d]" % (t >> 2)
            wr = "ds_write_b8_d16_hi" if by >= 2 else "ds_write_b8"
            if inl:
                # the inline constant is the LAST byte the step appends: address o' (cursor before the subtraction); byte 1 of
                # an entry without one is negative (bit 15), which sign-extended into the address sends the store out of range
                ap("v_or_b32_sdwa %%[ti%s], %s, sext(%s) %s src0_sel:DWORD src1_sel:BYTE_1" % (ch, o, e, SD))
                ap("ds_write_b8_d16_hi %%[ti%s], %s" % (ch, e))
            ap("v_sub_u32_sdwa %s, %s, %s %s src0_sel:DWORD src1_sel:BYTE_3" % (o, o, e, SD))
            ap("v_cmp_gt_i32_sdwa %s, 0, sext(%s) src0_sel:DWORD src1_sel:BYTE_2" % ("%[mA]" if ch == "A" else "vcc", e))
            ap("v_lshl_or_b32 %s, %s, 31, %s" % (e, e, o))
            ap("%s %s, %s%s" % (wr, e, src, " offset:1" if inl else ""))
        # constants of both chains in one store: lanes of the union take a slot each (wave-wide running count +
        # rank), chain A's job where there is one, else chain B's; the rare lane with both stores B's separately
        eA, aA, aB = "%%[eA%d]" % cur, "%%[aA%d]" % cur, "%%[aB%d]" % cur
        ap("s_or_b64 %[mU], %[mA], vcc")
        ap("s_and_saveexec_b64 %[sv], %[mU]")
        ap("s_cbranch_execz 1f")
        ap("v_lshl_or_b32 %s, %%[oA], 16, %s" % (aA, aA))
        ap("v_lshl_or_b32 %s, %%[oB], 16, %s" % (aB, aB))
        ap("v_cndmask_b32_e64 %s, %s, %s, %%[mA]" % (aA, aB, aA))
        ap("v_mbcnt_lo_u32_b32 %s, exec_lo, 0" % eA)
        ap("v_mbcnt_hi_u32_b32 %s, exec_hi, %s" % (eA, eA))
        ap("v_lshl_add_u32 %s, %s, 2, %%[jb]" % (eA, eA))
        ap("v_min_u32 %s, %%[jlim], %s" % (eA, eA))
        ap("ds_write_b32 %s, %s" % (eA, aA))
        ap("s_and_b64 %[mA], %[mA], vcc")
        ap("s_cbranch_scc0 1f")
        ap("s_bcnt1_i32_b64 %[st], %[mU]")
        ap("s_mov_b64 exec, %[mA]")
        ap("v_mbcnt_lo_u32_b32 %s, exec_lo, 0" % eA)
        ap("v_mbcnt_hi_u32_b32 %s, exec_hi, %s" % (eA, eA))
        ap("v_add_u32 %s, %%[st], %s" % (eA, eA))
        ap("v_lshl_add_u32 %s, %s, 2, %%[jb]" % (eA, eA))
        ap("v_min_u32 %s, %%[jlim], %s" % (eA, eA))
        ap("ds_write_b32 %s, %s" % (eA, aB))
        ap("1:")
        ap("s_mov_b64 exec, %[sv]")
        ap("s_bcnt1_i32_b64 %[st], %[mU]")
        ap("s_lshl2_add_u32 %[jb], %[st], %[jb]")
        ap("s_and_b64 %[mA], %[mA], vcc")
        ap("s_bcnt1_i32_b64 %[st], %[mA]")
        ap("s_lshl2_add_u32 %[jb], %[st], %[jb]")
    ap("s_setprio 0")
    return L


def sweep2b():
    """piece_sweep2 with BRANCH-FREE job noting (round 4).  The vote-and-rank form of sweep2 handles the jobs of a step pair under
    `s_and_saveexec … s_cbranch_execz`, with a second branch for lanes that have a constant on both chains: two branches and a
    dozen dependent scalar operations per step pair, 265 cycles per step pair on the wave's clock (KX_DEBUG_FLAGS=64 timeline)
    against the ~120 of the dependent LDS read the sweep is built around.  Here every step of each chain runs the same straight
    code: rank among the lanes with a constant (v_mbcnt over the compare's mask), slot address, -1 (out of range: the hardware
    drops the store) for lanes without one, one store; two scalar ops advance the wave's slot counter."""
    L = []
    ap = L.append
    row = lambda t: "%%[bo%d]" % (t >> 1)
    wsel = lambda t: "WORD_%d" % (t & 1)
    base = {"A": 32, "B": 0}
    ap("s_setprio 3")
    for ch in "AB":
        t = base[ch] + 31
        ap("v_add_u32_sdwa %%[a%s0], %s, %%[leaf%s] %s src0_sel:%s src1_sel:DWORD" % (ch, row(t), ch, SD, wsel(t)))
        ap("ds_read_b32 %%[e%s0], %%[a%s0]" % (ch, ch))
    for j in range(31, -1, -1):
        cur, nxt = (31 - j) & 1, (32 - j) & 1
        # in flight, oldest first: the two entry reads, then the previous step's two byte stores and two job stores
        ap("s_waitcnt lgkmcnt(%d)" % (0 if j == 31 else 4))
        for ch in "AB":
            ap("v_and_b32 %%[leaf%s], 0x3fc, %%[e%s%d]" % (ch, ch, cur))
            if j > 0:
                t = base[ch] + j - 1
                ap("v_add_u32_sdwa %%[a%s%d], %s, %%[leaf%s] %s src0_sel:%s src1_sel:DWORD" % (ch, nxt, row(t), ch, SD, wsel(t)))
                ap("ds_read_b32 %%[e%s%d], %%[a%s%d]" % (ch, nxt, ch, nxt))
        for ch in "AB":
            t = base[ch] + j
            e, o, a = "%%[e%s%d]" % (ch, cur), "%%[o%s]" % ch, "%%[a%s%d]" % (ch, cur)
            by = t & 3
            if by == 3:
                ap("v_lshrrev_b32 %%[tw%s], 8, %%[w%d]" % (ch, t >> 2))
            src = "%%[tw%s]" % ch if by in (1, 3) else "%%[w%d]" % (t >> 2)
            wr = "ds_write_b8_d16_hi" if by >= 2 else "ds_write_b8"
            ap("v_sub_u32_sdwa %s, %s, %s %s src0_sel:DWORD src1_sel:BYTE_3" % (o, o, e, SD))
            ap("v_cmp_gt_i32_sdwa vcc, 0, sext(%s) src0_sel:DWORD src1_sel:BYTE_2" % e)
            ap("v_lshl_or_b32 %s, %s, 31, %s" % (e, e, o))
            ap("%s %s, %s" % (wr, e, src))
            # the job (e is free now: it becomes the slot address)
            ap("v_mbcnt_lo_u32_b32 %s, vcc_lo, 0" % e)
            ap("v_mbcnt_hi_u32_b32 %s, vcc_hi, %s" % (e, e))
            ap("v_lshl_add_u32 %s, %s, 2, %%[jb]" % (e, e))
            ap("v_min_u32 %s, %%[jlim], %s" % (e, e))
            ap("v_cndmask_b32_e32 %s, -1, %s, vcc" % (e, e))
            ap("v_lshl_or_b32 %s, %s, 16, %s" % (a, o, a))
            ap("ds_write_b32 %s, %s" % (e, a))
            ap("s_bcnt1_i32_b64 %[st], vcc")
            ap("s_lshl2_add_u32 %[jb], %[st], %[jb]")
    ap("s_setprio 0")
    return L


def sweep2j():
    """job-stride layout: lane-private job lists, unconditional job stores (see the module docstring)"""
    L = []
    ap = L.append
    row = lambda t: "%%[bo%d]" % (t >> 1)
    wsel = lambda t: "WORD_%d" % (t & 1)
    base = {"A": 32, "B": 0}
    ap("s_setprio 3")
    for ch in "AB":
        t = base[ch] + 31
        ap("v_add_u32_sdwa %%[a%s0], %s, %%[leaf%s] %s src0_sel:%s src1_sel:DWORD" % (ch, row(t), ch, SD, wsel(t)))
        ap("ds_read_b32 %%[e%s0], %%[a%s0]" % (ch, ch))
    for j in range(31, -1, -1):
        cur, nxt = (31 - j) & 1, (32 - j) & 1
        # in flight, oldest first: the two entry reads, then the previous step's four stores (job + byte, per chain)
        ap("s_waitcnt lgkmcnt(%d)" % (0 if j == 31 else 4))
        for ch in "AB":
            ap("v_and_b32 %%[leaf%s], 0x3fc, %%[e%s%d]" % (ch, ch, cur))
            if j > 0:
                t = base[ch] + j - 1
                ap("v_add_u32_sdwa %%[a%s%d], %s, %%[leaf%s] %s src0_sel:%s src1_sel:DWORD" % (ch, nxt, row(t), ch, SD, wsel(t)))
                ap("ds_read_b32 %%[e%s%d], %%[a%s%d]" % (ch, nxt, ch, nxt))
        for ch in "AB":
            t = base[ch] + j
            e, o, a, p_ = "%%[e%s%d]" % (ch, cur), "%%[o%s]" % ch, "%%[a%s%d]" % (ch, cur), "%%[p%s]" % ch
            by = t & 3
            if by == 3:
                ap("v_lshrrev_b32 %%[tw%s], 8, %%[w%d]" % (ch, t >> 2))
            src = "%%[tw%s]" % ch if by in (1, 3) else "%%[w%d]" % (t >> 2)
            wr = "ds_write_b8_d16_hi" if by >= 2 else "ds_write_b8"
            ap("v_sub_u32_sdwa %s, %s, %s %s src0_sel:DWORD src1_sel:BYTE_3" % (o, o, e, SD))
            ap("v_lshl_or_b32 %s, %s, 16, %s" % (a, o, a))
            ap("ds_write_b32 %s, %s" % (p_, a))
            ap("%s %s, %s, %s %s src0_sel:DWORD src1_sel:BYTE_2" % ("v_add_u32_sdwa" if ch == "A" else "v_sub_u32_sdwa", p_, p_, e, SD))
            ap("v_lshl_or_b32 %s, %s, 31, %s" % (e, e, o))
            ap("%s %s, %s" % (wr, e, src))
    ap("s_setprio 0")
    return L


def fwd1(pack):
    """one chain over the 64 bytes of a piece from handle h; pack=True also records the back rows (bo)"""
    L = []
    ap = L.append
    def cls_issue(tt):
        ap("v_lshlrev_b32_sdwa %%[x], 0, %%[w%d] %s src0_sel:DWORD src1_sel:BYTE_%d" % (tt >> 2, SD, tt & 3))
        ap("ds_read_u8 %%[c%d], %%[x]" % (tt % 3))
    for tt in (0, 1, 2):
        cls_issue(tt)
    ap("s_waitcnt lgkmcnt(2)")
    ap("v_add_u32 %[x], %[h], %[c0]")
    ap("ds_read_b32 %[e0], %[x]")
    for j in range(64):
        cur, nxt = j & 1, (j + 1) & 1
        if j + 3 < 64:
            cls_issue(j + 3)
        ap("s_waitcnt lgkmcnt(%d)" % (1 if j + 3 < 64 else 0))
        if j + 1 < 64:
            ap("v_add_u32_sdwa %%[x], %%[e%d], %%[c%d] %s src0_sel:WORD_0 src1_sel:DWORD" % (cur, (j + 1) % 3, SD))
            ap("ds_read_b32 %%[e%d], %%[x]" % nxt)
        if pack:
            if j & 1:
                ap("v_and_or_b32 %%[bo%d], %%[e%d], %%[hm], %%[bo%d]" % (j >> 1, cur, j >> 1))
            else:
                ap("v_lshrrev_b32 %%[bo%d], 16, %%[e%d]" % (j >> 1, cur))
        else:
            if j == 31:
                ap("v_and_b32 %%[mid], 0xffff, %%[e%d]" % cur)
            if j == 63:
                ap("v_and_b32 %%[h], 0xffff, %%[e%d]" % cur)
    return L


def walk1(count=False):
    """backward walk that only measures: leaf chain + appended-byte sum, with the state in the middle of the piece.
    count (job-stride layout, byte 2 of an entry = 4 where a constant follows): the SAME add takes the entry's upper half-word,
    so the sum's low byte counts the constants (x 4) while bits 8.. add up the appended bytes — the count costs no instruction.
    The low byte is cleared in the middle (32 steps x 4 fit a byte, 64 would not)."""
    L = []
    ap = L.append
    ap("s_setprio 2")
    for t in range(63, -1, -1):
        ap("v_add_u32_sdwa %%[a], %%[bo%d], %%[leaf] %s src0_sel:WORD_%d src1_sel:DWORD" % (t >> 1, SD, t & 1))
        ap("ds_read_b32 %[e], %[a]")
        ap("s_waitcnt lgkmcnt(0)")
        ap("v_and_b32 %[leaf], 0x3fc, %[e]")
        ap("v_add_u32_sdwa %%[sum], %%[sum], %%[e] %s src0_sel:DWORD src1_sel:%s" % (SD, "WORD_1" if count else "BYTE_3"))
        if t == 32:
            ap("v_mov_b32 %[lmid], %[leaf]")
            ap("v_mov_b32 %[shi], %[sum]")
            if count:
                ap("v_and_b32 %[sum], 0xffffff00, %[sum]")
    ap("s_setprio 0")
    return L


PAIR_CLS2 = 16128     # k_forward's two-symbol stride: LDS address of the class table that holds class*2 (one byte per symbol)
PAIR_TAB = 16384      # ... and of the pair table (a 16-bit DS offset field reaches it; the image's [cls4 | fwd] stays at 0)


def run_pair():
    """k_forward's piece for programs whose PAIR TABLE fits LDS: 32 chained lookups instead of 64.
      pair[(q*C + c1)*C + c2] = C*C * (state after reading two symbols of classes c1, c2 in state q)     (u16)
    so the chain value e IS the index of the next state's row:  e' = pair[e + c1*C + c2].  Off the chain, per pair:
    two class reads (class*2) and pre = c1_2*C + c2_2; on the chain one v_lshl_add (address = 2e + pre, the table base
    rides in the DS offset field) and one ds_read_u16.  mid = chain value after 32 symbols."""
    L = []
    ap = L.append
    def cls_issue(j):
        for k in (0, 1):
            t = 2 * j + k
            ap("v_lshlrev_b32_sdwa %%[x], 0, %%[w%d] %s src0_sel:DWORD src1_sel:BYTE_%d" % (t >> 2, SD, t & 3))
            ap("ds_read_u8 %%[c%d%s], %%[x] offset:%d" % (j % 3, "ab"[k], PAIR_CLS2))
    cls_issue(0)
    cls_issue(1)
    ap("s_waitcnt lgkmcnt(2)")
    ap("v_mad_u32_u24 %[p0], %[c0a], %[C], %[c0b]")
    ap("v_lshl_add_u32 %[x], %[e], 1, %[p0]")
    ap("ds_read_u16 %%[e0], %%[x] offset:%d" % PAIR_TAB)
    for j in range(32):
        cur, nxt = j & 1, (j + 1) & 1
        if j + 2 < 32:
            cls_issue(j + 2)
        if j + 1 < 32:
            # in flight, oldest first: classes of pair j+1 (2), e_j, classes of pair j+2 (2, if issued)
            ap("s_waitcnt lgkmcnt(%d)" % (3 if j + 2 < 32 else 1))
            ap("v_mad_u32_u24 %%[p%d], %%[c%da], %%[C], %%[c%db]" % (nxt, (j + 1) % 3, (j + 1) % 3))
            ap("s_waitcnt lgkmcnt(%d)" % (2 if j + 2 < 32 else 0))
            ap("v_lshl_add_u32 %%[x], %%[e%d], 1, %%[p%d]" % (cur, nxt))
            ap("ds_read_u16 %%[e%d], %%[x] offset:%d" % (nxt, PAIR_TAB))
        else:
            ap("s_waitcnt lgkmcnt(0)")
        if j == 15:
            ap("v_mov_b32 %%[mid], %%[e%d]" % cur)
        if j == 31:
            ap("v_mov_b32 %%[e], %%[e%d]" % cur)
    return L


def main6(out):
    out.write("constexpr uint32_t PAIR_CLS2 = %d, PAIR_TAB = %d;   // LDS layout of k_forward's pair mode (gen_sweeps.py)\n\n" % (PAIR_CLS2, PAIR_TAB))
    tmp = ["e0", "e1", "c0a", "c0b", "c1a", "c1b", "c2a", "c2b", "p0", "p1", "x"]
    emit_fn(out, "piece_run_pair",
            "const uint32_t (&w)[16], uint32_t& e, uint32_t& mid, uint32_t C",
            "uint32_t " + ", ".join(tmp) + ";",
            run_pair(),
            ['[%s] "=&v"(%s)' % (t, t) for t in tmp] + ['[mid] "=&v"(mid)', '[e] "+v"(e)'],
            ['[w%d] "v"(w[%d])' % (i, i) for i in range(16)] + ['[C] "s"(C)'],
            '"memory"')


def emit_fn(out, name, sig, decl, lines, outs, ins, clob):
    out.write("__device__ __forceinline__ void %s(%s) {\n" % (name, sig))
    if decl:
        out.write("  %s\n" % decl)
    out.write("  asm volatile(\n")
    for ln in lines:
        out.write('      "%s\\n"\n' % ln)
    out.write("      : %s\n      : %s\n      : %s);\n}\n\n" % (", ".join(outs), ", ".join(ins), clob))


def main():
    out = sys.stdout
    out.write("// kx_sweeps.inc — GENERATED by gen_sweeps.py; do not edit.  See that script for the schedule.\n\n")
    tmp = ["eA0", "eA1", "eB0", "eB1", "cA0", "cA1", "cA2", "cB0", "cB1", "cB2", "xA", "xB", "aA", "aB"]
    emit_fn(out, "piece_forward2",
            "const uint32_t (&w)[16], uint32_t hA, uint32_t hB, uint32_t himask, uint32_t (&bo)[32]",
            "uint32_t " + ", ".join(tmp) + ";",
            fwd2(),
            ['[bo%d] "=&v"(bo[%d])' % (i, i) for i in range(32)] + ['[%s] "=&v"(%s)' % (t, t) for t in tmp],
            ['[w%d] "v"(w[%d])' % (i, i) for i in range(16)] + ['[hA] "v"(hA)', '[hB] "v"(hB)', '[hm] "s"(himask)'],
            '"memory"')
    tmp = ["eA0", "eA1", "eB0", "eB1", "aA0", "aA1", "aB0", "aB1", "twA", "twB"]
    emit_fn(out, "piece_sweep2",
            "const uint32_t (&bo)[32], const uint32_t (&w)[16], uint32_t leafA, uint32_t oA, uint32_t leafB, uint32_t oB, "
            "uint32_t& jb, uint32_t jlim",
            "uint32_t " + ", ".join(tmp) + ", st; unsigned long long sv, mA, mU;",
            sweep2(),
            ['[%s] "=&v"(%s)' % (t, t) for t in tmp] + ['[sv] "=&s"(sv)', '[mA] "=&s"(mA)', '[mU] "=&s"(mU)', '[st] "=&s"(st)', '[leafA] "+v"(leafA)', '[oA] "+v"(oA)',
                                                       '[leafB] "+v"(leafB)', '[oB] "+v"(oB)', '[jb] "+s"(jb)'],
            ['[bo%d] "v"(bo[%d])' % (i, i) for i in range(32)] + ['[w%d] "v"(w[%d])' % (i, i) for i in range(16)] +
            ['[jlim] "s"(jlim)'],
            '"vcc", "scc", "memory"')
    tmpb = ["eA0", "eA1", "eB0", "eB1", "aA0", "aA1", "aB0", "aB1", "twA", "twB"]
    emit_fn(out, "piece_sweep2b",
            "const uint32_t (&bo)[32], const uint32_t (&w)[16], uint32_t leafA, uint32_t oA, uint32_t leafB, uint32_t oB, "
            "uint32_t& jb, uint32_t jlim",
            "uint32_t " + ", ".join(tmpb) + ", st;",
            sweep2b(),
            ['[%s] "=&v"(%s)' % (t, t) for t in tmpb] + ['[st] "=&s"(st)', '[leafA] "+v"(leafA)', '[oA] "+v"(oA)',
                                                        '[leafB] "+v"(leafB)', '[oB] "+v"(oB)', '[jb] "+s"(jb)'],
            ['[bo%d] "v"(bo[%d])' % (i, i) for i in range(32)] + ['[w%d] "v"(w[%d])' % (i, i) for i in range(16)] +
            ['[jlim] "s"(jlim)'],
            '"vcc", "scc", "memory"')
    tmpj = ["eA0", "eA1", "eB0", "eB1", "aA0", "aA1", "aB0", "aB1", "twA", "twB"]
    emit_fn(out, "piece_sweep2j",
            "const uint32_t (&bo)[32], const uint32_t (&w)[16], uint32_t leafA, uint32_t oA, uint32_t leafB, uint32_t oB, "
            "uint32_t& pA, uint32_t& pB",
            "uint32_t " + ", ".join(tmpj) + ";",
            sweep2j(),
            ['[%s] "=&v"(%s)' % (t, t) for t in tmpj] + ['[leafA] "+v"(leafA)', '[oA] "+v"(oA)', '[leafB] "+v"(leafB)', '[oB] "+v"(oB)',
                                                        '[pA] "+v"(pA)', '[pB] "+v"(pB)'],
            ['[bo%d] "v"(bo[%d])' % (i, i) for i in range(32)] + ['[w%d] "v"(w[%d])' % (i, i) for i in range(16)],
            '"memory"')
    tmp = tmp + ["tiA", "tiB"]
    emit_fn(out, "piece_sweep2i",
            "const uint32_t (&bo)[32], const uint32_t (&w)[16], uint32_t leafA, uint32_t oA, uint32_t leafB, uint32_t oB, "
            "uint32_t& jb, uint32_t jlim",
            "uint32_t " + ", ".join(tmp) + ", st; unsigned long long sv, mA, mU;",
            sweep2(True),
            ['[%s] "=&v"(%s)' % (t, t) for t in tmp] + ['[sv] "=&s"(sv)', '[mA] "=&s"(mA)', '[mU] "=&s"(mU)', '[st] "=&s"(st)', '[leafA] "+v"(leafA)', '[oA] "+v"(oA)',
                                                       '[leafB] "+v"(leafB)', '[oB] "+v"(oB)', '[jb] "+s"(jb)'],
            ['[bo%d] "v"(bo[%d])' % (i, i) for i in range(32)] + ['[w%d] "v"(w[%d])' % (i, i) for i in range(16)] +
            ['[jlim] "s"(jlim)'],
            '"vcc", "scc", "memory"')


def main2(out):
    tmp = ["e0", "e1", "c0", "c1", "c2", "x"]
    emit_fn(out, "piece_forward1",
            "const uint32_t (&w)[16], uint32_t h, uint32_t himask, uint32_t (&bo)[32]",
            "uint32_t " + ", ".join(tmp) + ";",
            fwd1(True),
            ['[bo%d] "=&v"(bo[%d])' % (i, i) for i in range(32)] + ['[%s] "=&v"(%s)' % (t, t) for t in tmp],
            ['[w%d] "v"(w[%d])' % (i, i) for i in range(16)] + ['[h] "v"(h)', '[hm] "s"(himask)'],
            '"memory"')
    emit_fn(out, "piece_run1",
            "const uint32_t (&w)[16], uint32_t& h, uint32_t& mid",
            "uint32_t " + ", ".join(tmp) + ";",
            fwd1(False),
            ['[%s] "=&v"(%s)' % (t, t) for t in tmp] + ['[mid] "=&v"(mid)', '[h] "+v"(h)'],
            ['[w%d] "v"(w[%d])' % (i, i) for i in range(16)],
            '"memory"')
    emit_fn(out, "piece_walk1j",
            "const uint32_t (&bo)[32], uint32_t& leaf, uint32_t& sum, uint32_t& lmid, uint32_t& shi",
            "uint32_t a, e;",
            walk1(True),
            ['[a] "=&v"(a)', '[e] "=&v"(e)', '[lmid] "=&v"(lmid)', '[shi] "=&v"(shi)', '[leaf] "+v"(leaf)', '[sum] "+v"(sum)'],
            ['[bo%d] "v"(bo[%d])' % (i, i) for i in range(32)],
            '"memory"')
    emit_fn(out, "piece_walk1",
            "const uint32_t (&bo)[32], uint32_t& leaf, uint32_t& sum, uint32_t& lmid, uint32_t& shi",
            "uint32_t a, e;",
            walk1(),
            ['[a] "=&v"(a)', '[e] "=&v"(e)', '[lmid] "=&v"(lmid)', '[shi] "=&v"(shi)', '[leaf] "+v"(leaf)', '[sum] "+v"(sum)'],
            ['[bo%d] "v"(bo[%d])' % (i, i) for i in range(32)],
            '"memory"')


# ------------------------------------------------------------------------------------------------ delayed form (round 5)
# Entries of the delayed form's table are 8 bytes {lo, hi} (kx_dfkernels.inc), read with one ds_read_b64 into a register PAIR;
# inline-asm operands cannot name the halves of a pair, so the sequences below use fixed registers for them (listed as clobbers).
DF_WALK_E = {"A": (("v116", "v117"), ("v118", "v119")), "B": (("v120", "v121"), ("v122", "v123"))}


def dfrun1(wide=False):
    """k_dforward's piece: one chain of 64 product transitions from handle h; sum = bytes the steps write (entry hi, byte 3).
    mid / lenA = state and sum after 32 steps.  Per byte: 1 SDWA byte extract, 1 ds_read_u8 (class*8), 1 SDWA add on the chain,
    1 ds_read_b32 (the entry's LOW word: next handle, bytes appended, "a constant follows" — the high word is the walks'; reading
    it too, as the first version did, doubled the data this kernel pulls out of LDS), 1 SDWA add for the length."""
    L = []
    ap = L.append
    E = (("%[e0]",), ("%[e1]",))
    def cls_issue(tt):
        ap("v_lshlrev_b32_sdwa %%[x], 0, %%[w%d] %s src0_sel:DWORD src1_sel:BYTE_%d" % (tt >> 2, SD, tt & 3))
        ap("ds_read_u8 %%[c%d], %%[x]" % (tt % 3))
    for tt in (0, 1, 2):
        cls_issue(tt)
    ap("s_waitcnt lgkmcnt(2)")
    if wide:   # (more than 31 byte classes: the class table holds the class index, a row's entries are 8 bytes apart)
        ap("v_lshlrev_b32 %[c0], 3, %[c0]")
    ap("v_add_u32 %[x], %[h], %[c0]")
    ap("ds_read_b32 %s, %%[x]" % E[0][0])
    for j in range(64):
        cur, nxt = j & 1, (j + 1) & 1
        if j + 3 < 64:
            cls_issue(j + 3)
        ap("s_waitcnt lgkmcnt(%d)" % (1 if j + 3 < 64 else 0))
        if j + 1 < 64:
            if wide:
                ap("v_lshlrev_b32 %%[c%d], 3, %%[c%d]" % ((j + 1) % 3, (j + 1) % 3))
            ap("v_add_u32_sdwa %%[x], %s, %%[c%d] %s src0_sel:WORD_0 src1_sel:DWORD" % (E[cur][0], (j + 1) % 3, SD))
            ap("ds_read_b32 %s, %%[x]" % E[nxt][0])
        # (lo's upper half = bytes appended << 8 | 4 x "a constant follows": the sum's low byte counts the constants, its bits 8.. the bytes)
        ap("v_add_u32_sdwa %%[sum], %%[sum], %s %s src0_sel:DWORD src1_sel:WORD_1" % (E[cur][0], SD))
        if j == 31:
            ap("v_and_b32 %%[mid], 0xffff, %s" % E[cur][0])
            ap("v_mov_b32 %[lenA], %[sum]")
            ap("v_and_b32 %[sum], 0xffffff00, %[sum]")   # (32 steps x 4 fit a byte, 64 would not)
        if j == 63:
            ap("v_and_b32 %%[h], 0xffff, %s" % E[cur][0])
    return L


def dfwalk2(K, counted=True, wide=False):
    """k_demit's fused walk for delay K: chain A = steps 0..31 from (hA, oA), chain B = steps 32..63 from (hB, oB); step s reads the
    class of input byte s, takes the product transition, and places what the entry says: input byte s-K (if the entry copies) at
    the cursor, a job for the constant (if one follows), cursor += bytes appended.  Job noting is k_emit's branch-free form
    (piece_sweep2b): rank among the lanes with a constant, slot address, -1 for lanes without one, one store."""
    L = []
    ap = L.append
    base = {"A": 0, "B": 32}
    E = DF_WALK_E
    pair = lambda ch, k: "v[%s:%s]" % (E[ch][k][0][1:], E[ch][k][1][1:])
    def cls_issue(ch, t, slot):
        ap("v_lshlrev_b32_sdwa %%[x%s], 0, %%[w%d] %s src0_sel:DWORD src1_sel:BYTE_%d" % (ch, t >> 2, SD, t & 3))
        ap("ds_read_u8 %%[c%s%d], %%[x%s]" % (ch, slot, ch))
    twof = {"A": None, "B": None}
    ap("s_setprio 3")
    for d in (0, 1):
        for ch in "AB":
            cls_issue(ch, base[ch] + d, d)
    ap("s_waitcnt lgkmcnt(2)")
    for ch in "AB":
        if wide:
            ap("v_lshlrev_b32 %%[c%s0], 3, %%[c%s0]" % (ch, ch))
        ap("v_add_u32 %%[a%s0], %%[h%s], %%[c%s0]" % (ch, ch, ch))
        ap("ds_read_b64 %s, %%[a%s0]" % (pair(ch, 0), ch))
    for j in range(32):
        cur, nxt = j & 1, (j + 1) & 1
        if j + 2 < 32:
            for ch in "AB":
                cls_issue(ch, base[ch] + j + 2, (j + 2) % 3)
        # in flight, oldest first: [previous step's 2 byte stores + 2 job stores], class j+1 (2), entry j (2), class j+2 (2, if issued):
        # wait until the entries of step j are in
        # (counted slots: a job store of a step in which no lane has a constant may not count — wait as if it had not been issued)
        ap("s_waitcnt lgkmcnt(%d)" % (2 if j == 0 else (4 if counted else 6) if j + 2 < 32 else (2 if counted else 4)))
        if j + 1 < 32:
            for ch in "AB":
                if wide:
                    ap("v_lshlrev_b32 %%[c%s%d], 3, %%[c%s%d]" % (ch, (j + 1) % 3, ch, (j + 1) % 3))
                ap("v_add_u32_sdwa %%[a%s%d], %s, %%[c%s%d] %s src0_sel:WORD_0 src1_sel:DWORD" % (ch, nxt, E[ch][cur][0], ch, (j + 1) % 3, SD))
                ap("ds_read_b64 %s, %%[a%s%d]" % (pair(ch, nxt), ch, nxt))
        for ch in "AB":
            t = base[ch] + j
            so = t - K
            hi, o, a, x = E[ch][cur][1], "%%[o%s]" % ch, "%%[a%s%d]" % (ch, cur), "%%[x%s]" % ch
            if so >= 0:
                wreg, by = "%%[w%d]" % (so >> 2), so & 3
            else:
                wreg, by = "%[wp]", (4 + so) & 3
            if by in (1, 3) and twof[ch] != wreg:
                # (a shifted copy of the dword serves its bytes 1 and 3)
                ap("v_lshrrev_b32 %%[tw%s], 8, %s" % (ch, wreg))
                twof[ch] = wreg
            src = "%%[tw%s]" % ch if by in (1, 3) else wreg
            wr = "ds_write_b8_d16_hi" if by >= 2 else "ds_write_b8"
            if counted:
                # COUNTED SLOTS: the piece records carry how many constants each half of the piece appends, so every lane owns its job
                # slots before the walk (a prefix sum in k_demit) and a step with a constant stores its job at the lane's own pointer
                # — no rank among the lanes, no clamp, no wave-wide counter; only the lanes with a constant take part in the store
                pp = "%%[p%s]" % ch
                ap("v_cmp_gt_i32_sdwa vcc, 0, sext(%s) src0_sel:DWORD src1_sel:BYTE_2" % hi)
                ap("v_lshl_or_b32 %s, %s, 16, %s" % (a, o, a))
                ap("v_lshl_or_b32 %s, %s, 31, %s" % (x, hi, o))
                ap("v_add_u32_sdwa %s, %s, %s %s src0_sel:DWORD src1_sel:BYTE_3" % (o, o, hi, SD))
                ap("%s %s, %s" % (wr, x, src))
                ap("s_and_saveexec_b64 %[sv], vcc")
                ap("ds_write_b32 %s, %s" % (pp, a))
                ap("s_mov_b64 exec, %[sv]")
                ap("v_add_u32_sdwa %s, %s, %s %s src0_sel:DWORD src1_sel:BYTE_2" % (pp, pp, E[ch][cur][0], SD))
                continue
            ap("v_cmp_gt_i32_sdwa vcc, 0, sext(%s) src0_sel:DWORD src1_sel:BYTE_2" % hi)
            ap("v_lshl_or_b32 %s, %s, 16, %s" % (a, o, a))              # the job word: cursor << 16 | entry address
            ap("v_lshl_or_b32 %s, %s, 31, %s" % (x, hi, o))             # where the copied byte goes (out of range if nothing is copied)
            ap("v_add_u32_sdwa %s, %s, %s %s src0_sel:DWORD src1_sel:BYTE_3" % (o, o, hi, SD))
            ap("%s %s, %s" % (wr, x, src))
            ap("v_mbcnt_lo_u32_b32 %s, vcc_lo, 0" % hi)
            ap("v_mbcnt_hi_u32_b32 %s, vcc_hi, %s" % (hi, hi))
            ap("v_lshl_add_u32 %s, %s, 2, %%[jb]" % (hi, hi))
            ap("v_min_u32 %s, %%[jlim], %s" % (hi, hi))
            ap("v_cndmask_b32_e32 %s, -1, %s, vcc" % (hi, hi))
            ap("ds_write_b32 %s, %s" % (hi, a))
            ap("s_bcnt1_i32_b64 %[st], vcc")
            ap("s_lshl2_add_u32 %[jb], %[st], %[jb]")
    ap("s_setprio 0")
    return L


def dfwalk1(K, wide=False):
    """k_demit's walk in HALF-PIECE mode (a lane owns 32 input bytes: programs whose pieces are too large for a wave's staging area run
    one walk over twice as many lanes instead of several walks over a part of them): ONE chain of 32 steps from (h, o), the placing and
    the counted job slots of dfwalk2."""
    L = []
    ap = L.append
    E = DF_WALK_E["A"]
    pair = lambda k: "v[%s:%s]" % (E[k][0][1:], E[k][1][1:])
    def cls_issue(t, slot):
        ap("v_lshlrev_b32_sdwa %%[x], 0, %%[w%d] %s src0_sel:DWORD src1_sel:BYTE_%d" % (t >> 2, SD, t & 3))
        ap("ds_read_u8 %%[c%d], %%[x]" % slot)
    twof = None
    ap("s_setprio 3")
    for d in (0, 1, 2):
        cls_issue(d, d)
    ap("s_waitcnt lgkmcnt(2)")
    if wide:
        ap("v_lshlrev_b32 %[c0], 3, %[c0]")
    ap("v_add_u32 %[a0], %[h], %[c0]")
    ap("ds_read_b64 %s, %%[a0]" % pair(0))
    for j in range(32):
        cur, nxt = j & 1, (j + 1) & 1
        if j + 3 < 32:
            cls_issue(j + 3, (j + 3) % 3)
        # in flight behind the entry read of step j: the previous step's byte store (and its job store, which may not count) and the
        # class reads issued since
        ap("s_waitcnt lgkmcnt(%d)" % ((1 if j + 3 < 32 else 0) + (0 if j == 0 else 1)))
        if j + 1 < 32:
            if wide:
                ap("v_lshlrev_b32 %%[c%d], 3, %%[c%d]" % ((j + 1) % 3, (j + 1) % 3))
            ap("v_add_u32_sdwa %%[a%d], %s, %%[c%d] %s src0_sel:WORD_0 src1_sel:DWORD" % (nxt, E[cur][0], (j + 1) % 3, SD))
            ap("ds_read_b64 %s, %%[a%d]" % (pair(nxt), nxt))
        so = j - K
        hi, a = E[cur][1], "%%[a%d]" % cur
        if so >= 0:
            wreg, by = "%%[w%d]" % (so >> 2), so & 3
        else:
            wreg, by = "%[wp]", (4 + so) & 3
        if by in (1, 3) and twof != wreg:
            ap("v_lshrrev_b32 %%[tw], 8, %s" % wreg)
            twof = wreg
        src = "%[tw]" if by in (1, 3) else wreg
        wr = "ds_write_b8_d16_hi" if by >= 2 else "ds_write_b8"
        ap("v_cmp_gt_i32_sdwa vcc, 0, sext(%s) src0_sel:DWORD src1_sel:BYTE_2" % hi)
        ap("v_lshl_or_b32 %s, %%[o], 16, %s" % (a, a))
        ap("v_lshl_or_b32 %%[x], %s, 31, %%[o]" % hi)
        ap("v_add_u32_sdwa %%[o], %%[o], %s %s src0_sel:DWORD src1_sel:BYTE_3" % (hi, SD))
        ap("%s %%[x], %s" % (wr, src))
        ap("s_and_saveexec_b64 %[sv], vcc")
        ap("ds_write_b32 %%[p], %s" % a)
        ap("s_mov_b64 exec, %[sv]")
        ap("v_add_u32_sdwa %%[p], %%[p], %s %s src0_sel:DWORD src1_sel:BYTE_2" % (E[cur][0], SD))
    ap("s_setprio 0")
    return L


def main7(out):
    tmp = ["c0", "c1", "c2", "x", "e0", "e1"]
    emit_fn(out, "piece_dfrun1w",
            "const uint32_t (&w)[16], uint32_t& h, uint32_t& mid, uint32_t& lenA, uint32_t& sum",
            "uint32_t " + ", ".join(tmp) + ";",
            dfrun1(True),
            ['[%s] "=&v"(%s)' % (t, t) for t in tmp] + ['[mid] "=&v"(mid)', '[lenA] "=&v"(lenA)', '[h] "+v"(h)', '[sum] "+v"(sum)'],
            ['[w%d] "v"(w[%d])' % (i, i) for i in range(16)],
            '"memory"')
    emit_fn(out, "piece_dfrun1",
            "const uint32_t (&w)[16], uint32_t& h, uint32_t& mid, uint32_t& lenA, uint32_t& sum",
            "uint32_t " + ", ".join(tmp) + ";",
            dfrun1(),
            ['[%s] "=&v"(%s)' % (t, t) for t in tmp] + ['[mid] "=&v"(mid)', '[lenA] "=&v"(lenA)', '[h] "+v"(h)', '[sum] "+v"(sum)'],
            ['[w%d] "v"(w[%d])' % (i, i) for i in range(16)],
            '"memory"')
    tmp1 = ["c0", "c1", "c2", "x", "a0", "a1", "tw"]
    clob1 = ", ".join('"%s"' % r for pr in DF_WALK_E["A"] for r in pr)
    for K, wd in ((1, False), (2, False), (1, True), (2, True)):
        emit_fn(out, "piece_dfwalk1c%s_k%d" % ("w" if wd else "", K),
                "const uint32_t (&w)[8], uint32_t wp, uint32_t h, uint32_t& o, uint32_t& p",
                "uint32_t " + ", ".join(tmp1) + "; unsigned long long sv;",
                dfwalk1(K, wd),
                ['[%s] "=&v"(%s)' % (t, t) for t in tmp1] + ['[sv] "=&s"(sv)', '[o] "+v"(o)', '[p] "+v"(p)'],
                ['[w%d] "v"(w[%d])' % (i, i) for i in range(8)] + ['[wp] "v"(wp)', '[h] "v"(h)'],
                '"vcc", "scc", "memory", ' + clob1)
    tmp = ["cA0", "cA1", "cA2", "cB0", "cB1", "cB2", "xA", "xB", "aA0", "aA1", "aB0", "aB1", "twA", "twB"]
    clob = ", ".join('"%s"' % r for ch in "AB" for pr in DF_WALK_E[ch] for r in pr)
    for K, wd in ((1, False), (2, False), (1, True), (2, True)):
        emit_fn(out, "piece_dfwalk2c%s_k%d" % ("w" if wd else "", K),
                "const uint32_t (&w)[16], uint32_t wp, uint32_t hA, uint32_t& oA, uint32_t hB, uint32_t& oB, uint32_t& pA, uint32_t& pB",
                "uint32_t " + ", ".join(tmp) + "; unsigned long long sv;",
                dfwalk2(K, counted=True, wide=wd),
                ['[%s] "=&v"(%s)' % (t, t) for t in tmp] + ['[sv] "=&s"(sv)', '[oA] "+v"(oA)', '[oB] "+v"(oB)', '[pA] "+v"(pA)', '[pB] "+v"(pB)'],
                ['[w%d] "v"(w[%d])' % (i, i) for i in range(16)] + ['[wp] "v"(wp)', '[hA] "v"(hA)', '[hB] "v"(hB)'],
                '"vcc", "scc", "memory", ' + clob)


if __name__ == "__main__":
    main()
    main2(sys.stdout)
    main6(sys.stdout)
    main7(sys.stdout)
