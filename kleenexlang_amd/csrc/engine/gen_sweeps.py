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


def sweep2():
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
        ap("s_waitcnt lgkmcnt(%d)" % (0 if j == 31 else 2))
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
            ap("v_sub_u32_sdwa %s, %s, %s %s src0_sel:DWORD src1_sel:BYTE_3" % (o, o, e, SD))
            ap("v_cmp_gt_i32_sdwa %s, 0, sext(%s) src0_sel:DWORD src1_sel:BYTE_2" % ("%[mA]" if ch == "A" else "vcc", e))
            ap("v_lshl_or_b32 %s, %s, 31, %s" % (e, e, o))
            ap("%s %s, %s" % (wr, e, src))
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


def walk1():
    """backward walk that only measures: leaf chain + appended-byte sum, with the state in the middle of the piece"""
    L = []
    ap = L.append
    ap("s_setprio 2")
    for t in range(63, -1, -1):
        ap("v_add_u32_sdwa %%[a], %%[bo%d], %%[leaf] %s src0_sel:WORD_%d src1_sel:DWORD" % (t >> 1, SD, t & 1))
        ap("ds_read_b32 %[e], %[a]")
        ap("s_waitcnt lgkmcnt(0)")
        ap("v_and_b32 %[leaf], 0x3fc, %[e]")
        ap("v_add_u32_sdwa %%[sum], %%[sum], %%[e] %s src0_sel:DWORD src1_sel:BYTE_3" % SD)
        if t == 32:
            ap("v_mov_b32 %[lmid], %[leaf]")
            ap("v_mov_b32 %[shi], %[sum]")
    ap("s_setprio 0")
    return L


def place1():
    """The output stage's dense sweep (k_emit2): no table is consulted.  info holds one byte per step of the piece —
    bits 0-6 the bytes that step appends (1 for a plain copy), bit 7 "the input byte is not copied" — prepared from
    the piece's break records.  The cursor runs down the staging area; per step: cursor -= appended; the input byte
    is stored at the cursor (bit 7, sign-extended and or-ed in, sends the store of a step that copies nothing out of
    range: the hardware drops it).  3.75 instructions per byte, no LDS read, no wait."""
    L = []
    ap = L.append
    for t in range(63, -1, -1):
        wi, k = t >> 2, t & 3
        if k == 3:
            ap("v_and_b32 %%[d], %%[c7f], %%[i%d]" % wi)
            ap("v_and_b32 %%[n], %%[c80], %%[i%d]" % wi)
            ap("v_lshrrev_b32 %%[tw], 8, %%[w%d]" % wi)
        ap("v_sub_u32_sdwa %%[o], %%[o], %%[d] %s src0_sel:DWORD src1_sel:BYTE_%d" % (SD, k))
        ap("v_or_b32_sdwa %%[a], %%[o], sext(%%[n]) %s src0_sel:DWORD src1_sel:BYTE_%d" % (SD, k))
        src = "%[tw]" if k in (1, 3) else "%%[w%d]" % wi
        ap("%s %%[a], %s" % ("ds_write_b8_d16_hi" if k >= 2 else "ds_write_b8", src))
    return L


def scatter16():
    """k_emit2: the lane's first sixteen break records (descending step order) become info bytes and constant jobs.
    record r = appended bytes (bits 0-6) | not copied (bit 7) | step t (bits 8-13) | action id (bits 16-31).
      info byte:  lds8[ial + t] = r.byte0
      job:        slot k of the lane's own slots gets action id << 16 | low 16 bits of E (action id 0: nothing to copy),
                  E = LDS address one past the end of this step's output
                    = c1 + k + t - (bytes appended by the lane's records 0..k-1),  c1 = piece's staging end - plen + 1
    Lanes without a record k are masked; the wave leaves as soon as no lane has one (nmax = most records of any lane)."""
    L = []
    ap = L.append
    ap("s_mov_b64 %[sv], exec")
    for k in range(16):
        r = "%%[r%d]" % k
        ap("s_cmp_le_u32 %%[nmax], %d" % k)
        ap("s_cbranch_scc1 9f")
        ap("v_cmp_lt_u32 vcc, %d, %%[nrec]" % k)
        ap("s_and_b64 exec, %[sv], vcc")
        ap("v_add_u32_sdwa %%[a], %%[ial], %s %s src0_sel:DWORD src1_sel:BYTE_1" % (r, SD))
        ap("ds_write_b8 %%[a], %s" % r)
        ap("v_sub_u32 %[x], %[c1], %[S]")
        ap("v_and_b32 %%[dl], 0x7f, %s" % r)
        ap("v_add_u32 %[S], %[S], %[dl]")
        ap("v_add_u32_sdwa %%[x], %%[x], %s %s src0_sel:DWORD src1_sel:BYTE_1" % (r, SD))
        if k:
            ap("v_add_u32 %%[x], %d, %%[x]" % k)
        ap("v_bfi_b32 %%[x], %%[cff], %%[x], %s" % r)
        ap("ds_write_b32 %%[ja], %%[x] offset:%d" % (4 * k))
    ap("9:")
    ap("s_mov_b64 exec, %[sv]")
    return L


def run_trace():
    """k_forward2's piece: 64 chained transitions (as piece_run1) that also write the run trace.  Per step, after the
    transition word e arrived:  head = (e != e of the previous step);  a head appends (e & 0xffff0000) | position to the
    lane's ring of 32 entries in LDS (cnt4 = 4 x entries appended so far; a non-head stores out of range).  After every
    16 steps a lane with 16 or more entries pending writes one whole 64-byte sector of them to memory (ga = byte offset of
    the lane's trace area from tbase, fl4 = 4 x entries written out): smaller writes cost the memory a read-modify-write."""
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
        cur, nxt = "%%[e%d]" % (j & 1), "%%[e%d]" % ((j + 1) & 1)
        if j + 3 < 64:
            cls_issue(j + 3)
        pend = (1 if j + 3 < 64 else 0) + (1 if j > 0 else 0)
        ap("s_waitcnt lgkmcnt(%d)" % pend)
        # head test against the previous step's word (still in the other register) before that register is reused
        ap("v_cmp_ne_u32 vcc, %s, %s" % (cur, nxt))
        if j + 1 < 64:
            ap("v_add_u32_sdwa %%[x], %s, %%[c%d] %s src0_sel:WORD_0 src1_sel:DWORD" % (cur, (j + 1) % 3, SD))
            ap("ds_read_b32 %s, %%[x]" % nxt)
        ap("v_add_u32 %%[t], %d, %%[posl]" % j)
        ap("v_bfi_b32 %%[t], %%[cff], %%[t], %s" % cur)
        ap("v_bfi_b32 %[ra], %[c7c], %[cnt4], %[ring]")
        ap("v_cndmask_b32 %[ra], %[oob], %[ra], vcc")
        ap("ds_write_b32 %[ra], %[t]")
        ap("v_cndmask_b32 %[t], 0, %[four], vcc")
        ap("v_add_u32 %[cnt4], %[cnt4], %[t]")
        if j & 15 == 15:
            ap("v_sub_u32 %[t], %[cnt4], %[fl4]")
            ap("v_cmp_lt_u32 vcc, 63, %[t]")
            ap("s_and_saveexec_b64 %[sv], vcc")
            ap("s_cbranch_execz 1f")
            ap("v_and_b32 %[t], 64, %[fl4]")
            ap("v_add_u32 %[t], %[t], %[ring]")
            ap("ds_read_b128 %[x0], %[t]")
            ap("ds_read_b128 %[x1], %[t] offset:16")
            ap("ds_read_b128 %[x2], %[t] offset:32")
            ap("ds_read_b128 %[x3], %[t] offset:48")
            ap("v_add_u32 %[t], %[ga], %[fl4]")
            ap("v_add_u32 %[fl4], 64, %[fl4]")
            ap("s_waitcnt lgkmcnt(0)")
            ap("global_store_dwordx4 %[t], %[x0], %[tbase]")
            ap("global_store_dwordx4 %[t], %[x1], %[tbase] offset:16")
            ap("global_store_dwordx4 %[t], %[x2], %[tbase] offset:32")
            ap("global_store_dwordx4 %[t], %[x3], %[tbase] offset:48")
            ap("1:")
            ap("s_mov_b64 exec, %[sv]")
    ap("s_waitcnt lgkmcnt(0)")
    return L


def backloop():
    """k_backlen2's flat loop (programs without wide entries): every lane walks its own block backward at its own pace,
    one table access per trip.  A trip handles step p and, when the entry read is flagged E_FIXED, every step below it
    down to the first step of its run (never past the start of the piece, so that every piece gets its record):
      a = run.row + leaf;  e = lds[a];  leaf = e & 0x3fc
      lo = fixed(e) ? max(run.pos, floor) : p;  cum += e.appended + (p - lo);  p = lo - 1
      break record (steps that append a constant or do not copy) into the lane's ring of 16
      p < run.pos: next run — the entry fetched one run ahead becomes current, the one below it is requested
      p < floor:   piece complete — {cum, record offset} into the lane's ring of 8 piece records
    Every 4th / 8th trip is a wave-wide checkpoint where whole 32-byte sectors leave the rings.  Runs are read as
    packed entries (row << 16 | position) and used through SDWA operands without unpacking."""
    L = []
    ap = L.append
    ap("s_mov_b64 %[sfull], exec")
    ap("0:")
    ap("v_cmp_ge_i32 vcc, %[p], %[plim]")
    ap("s_and_b64 exec, %[sfull], vcc")
    ap("s_cbranch_execz 9f")
    ap("v_add_u32_sdwa %%[a], %%[leaf], %%[qcur] %s src0_sel:DWORD src1_sel:WORD_1" % SD)
    ap("ds_read_b32 %[e], %[a]")
    ap("v_max_i32_sdwa %%[lo], %%[floor], %%[qcur] %s src0_sel:DWORD src1_sel:WORD_0" % SD)
    ap("v_and_b32 %[tt], 63, %[p]")
    ap("s_waitcnt lgkmcnt(0)")
    ap("v_and_b32 %[t1], 2, %[e]")
    ap("v_cmp_ne_u32 vcc, 0, %[t1]")
    ap("v_cndmask_b32 %[lo], %[p], %[lo], vcc")
    ap("v_and_b32 %[leaf], 0x3fc, %[e]")
    ap("v_lshrrev_b32 %[dl], 24, %[e]")
    ap("v_sub_u32 %[t1], %[p], %[lo]")
    ap("v_add3_u32 %[cum], %[cum], %[dl], %[t1]")
    ap("v_add_u32 %[p], -1, %[lo]")
    # break record
    ap("v_and_b32 %[t1], %[cbrk], %[e]")
    ap("v_cmp_ne_u32 vcc, 0, %[t1]")
    ap("v_and_b32 %[t2], 1, %[e]")
    ap("v_lshl_or_b32 %[rec], %[t2], 7, %[dl]")
    ap("v_lshl_or_b32 %[rec], %[tt], 8, %[rec]")
    ap("v_bfe_u32 %[t2], %[e], 10, 13")
    ap("v_lshl_or_b32 %[rec], %[t2], 16, %[rec]")
    ap("v_and_b32 %[t2], 15, %[boff]")
    ap("v_lshl_add_u32 %[t2], %[t2], 2, %[rring]")
    ap("v_cndmask_b32 %[t2], %[oob], %[t2], vcc")
    ap("ds_write_b32 %[t2], %[rec]")
    ap("v_addc_co_u32 %[boff], vcc, 0, %[boff], vcc")
    # next run
    ap("v_cmp_lt_i32_sdwa vcc, %[p], %[qcur] src0_sel:DWORD src1_sel:WORD_0")
    ap("s_and_saveexec_b64 %[sv], vcc")
    ap("s_cbranch_execz 1f")
    ap("s_waitcnt vmcnt(0)")
    ap("v_sub_u32 %[qcur], %[qn], %[ceo]")      # (rows are image offsets; the tables are staged from off_ent on)
    ap("global_load_dword %[qn], %[va], %[tbase]")
    ap("v_add_u32 %[va], -4, %[va]")
    ap("1:")
    ap("s_mov_b64 exec, %[sv]")
    # piece complete
    ap("v_cmp_lt_i32 vcc, %[p], %[floor]")
    ap("s_and_saveexec_b64 %[sv], vcc")
    ap("s_cbranch_execz 2f")
    ap("v_and_b32 %[t1], 7, %[pp]")
    ap("v_lshl_add_u32 %[t1], %[t1], 3, %[pring]")
    ap("v_sub_u32 %[t2], %[boff], %[pb0]")
    ap("v_max_u32 %[kmax], %[kmax], %[t2]")
    ap("v_lshl_or_b32 %[t2], %[pb0], 14, %[t2]")
    ap("ds_write_b32 %[t1], %[cum]")
    ap("ds_write_b32 %[t1], %[t2] offset:4")
    ap("v_mov_b32 %[pb0], %[boff]")
    ap("v_add_u32 %[pp], -1, %[pp]")
    ap("v_add_u32 %[floor], -64, %[floor]")
    ap("2:")
    ap("s_mov_b64 exec, %[sv]")
    # trip counter; checkpoints
    ap("s_add_u32 %[tick], %[tick], 1")
    ap("s_and_b32 %[st], %[tick], 3")
    ap("s_cmp_lg_u32 %[st], 0")
    ap("s_cbranch_scc1 0b")
    ap("s_mov_b64 exec, %[sfull]")
    ap("s_waitcnt lgkmcnt(0)")
    # piece records: pieces [ptop-4, ptop) complete?  (ptop - 4 > pp)
    ap("v_add_u32 %[t1], -4, %[ptop]")
    ap("v_cmp_gt_i32 vcc, %[t1], %[pp]")
    ap("s_and_saveexec_b64 %[sv], vcc")
    ap("s_cbranch_execz 3f")
    ap("v_and_b32 %[t2], 4, %[t1]")
    ap("v_lshl_add_u32 %[t2], %[t2], 3, %[pring]")
    ap("ds_read_b128 %[x0], %[t2]")
    ap("ds_read_b128 %[x1], %[t2] offset:16")
    ap("v_add_u32 %[pa], -32, %[pa]")
    ap("v_mov_b32 %[ptop], %[t1]")
    ap("s_waitcnt lgkmcnt(0)")
    ap("global_store_dwordx4 %[pa], %[x0], %[pbase]")
    ap("global_store_dwordx4 %[pa], %[x1], %[pbase] offset:16")
    ap("3:")
    ap("s_mov_b64 exec, %[sv]")
    ap("s_and_b32 %[st], %[tick], 7")
    ap("s_cmp_lg_u32 %[st], 0")
    ap("s_cbranch_scc1 0b")
    # break records: eight pending?
    ap("v_sub_u32 %[t1], %[boff], %[bfl]")
    ap("v_cmp_lt_u32 vcc, 7, %[t1]")
    ap("s_and_saveexec_b64 %[sv], vcc")
    ap("s_cbranch_execz 4f")
    ap("v_and_b32 %[t2], 8, %[bfl]")
    ap("v_lshl_add_u32 %[t2], %[t2], 2, %[rring]")
    ap("ds_read_b128 %[x0], %[t2]")
    ap("ds_read_b128 %[x1], %[t2] offset:16")
    ap("v_add_u32 %[bfl], 8, %[bfl]")
    ap("s_waitcnt lgkmcnt(0)")
    ap("global_store_dwordx4 %[ba], %[x0], %[bbase]")
    ap("global_store_dwordx4 %[ba], %[x1], %[bbase] offset:16")
    ap("v_add_u32 %[ba], 32, %[ba]")
    ap("4:")
    ap("s_mov_b64 exec, %[sv]")
    ap("s_branch 0b")
    ap("9:")
    ap("s_mov_b64 exec, %[sfull]")
    ap("s_waitcnt vmcnt(0) lgkmcnt(0)")   # (the entry requested last must not land in a register the compiler has reused)
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
    emit_fn(out, "piece_walk1",
            "const uint32_t (&bo)[32], uint32_t& leaf, uint32_t& sum, uint32_t& lmid, uint32_t& shi",
            "uint32_t a, e;",
            walk1(),
            ['[a] "=&v"(a)', '[e] "=&v"(e)', '[lmid] "=&v"(lmid)', '[shi] "=&v"(shi)', '[leaf] "+v"(leaf)', '[sum] "+v"(sum)'],
            ['[bo%d] "v"(bo[%d])' % (i, i) for i in range(32)],
            '"memory"')


def main3(out):
    tmp = ["d", "n", "a", "tw"]
    emit_fn(out, "piece_place1",
            "const uint32_t (&w)[16], const uint32_t (&info)[16], uint32_t o",
            "uint32_t " + ", ".join(tmp) + ";",
            place1(),
            ['[%s] "=&v"(%s)' % (t, t) for t in tmp] + ['[o] "+v"(o)'],
            ['[w%d] "v"(w[%d])' % (i, i) for i in range(16)] + ['[i%d] "v"(info[%d])' % (i, i) for i in range(16)] +
            ['[c7f] "s"(0x7f7f7f7fu)', '[c80] "s"(0x80808080u)'],
            '"memory"')
    tmp = ["a", "x", "dl"]
    emit_fn(out, "piece_scatter16",
            "const uint32_t (&r)[16], uint32_t nrec, uint32_t nmax, uint32_t ial, uint32_t c1, uint32_t ja, uint32_t& S",
            "uint32_t " + ", ".join(tmp) + "; unsigned long long sv;",
            scatter16(),
            ['[%s] "=&v"(%s)' % (t, t) for t in tmp] + ['[S] "+v"(S)', '[sv] "=&s"(sv)'],
            ['[r%d] "v"(r[%d])' % (i, i) for i in range(16)] + ['[nrec] "v"(nrec)', '[nmax] "s"(nmax)', '[ial] "v"(ial)', '[c1] "v"(c1)',
                                                               '[ja] "v"(ja)', '[cff] "s"(0xffffu)'],
            '"vcc", "scc", "memory"')


def main5(out):
    tmp = ["c0", "c1", "c2", "x", "t", "ra", "x0", "x1", "x2", "x3"]
    emit_fn(out, "piece_run_trace",
            "const uint32_t (&w)[16], uint32_t h, uint32_t& e0, uint32_t& e1, uint32_t posl, uint32_t& cnt4, uint32_t& fl4, uint32_t ring, uint32_t ga, unsigned long long tbase",
            "uint32_t c0, c1, c2, x, t, ra; u32x4 x0, x1, x2, x3; unsigned long long sv;",
            run_trace(),
            ['[%s] "=&v"(%s)' % (t, t) for t in tmp] + ['[sv] "=&s"(sv)', '[e0] "=&v"(e0)', '[e1] "+v"(e1)', '[cnt4] "+v"(cnt4)', '[fl4] "+v"(fl4)'],
            ['[w%d] "v"(w[%d])' % (i, i) for i in range(16)] + ['[h] "v"(h)', '[posl] "v"(posl)', '[ring] "v"(ring)', '[ga] "v"(ga)', '[tbase] "s"(tbase)',
                                                               '[cff] "s"(0xffffu)', '[c7c] "s"(0x7cu)', '[oob] "v"(0x80000000u)', '[four] "v"(4u)'],
            '"vcc", "memory"')


def main4(out):
    emit_fn(out, "back_loop",
            "BackState& S, uint32_t plim, uint32_t ceo, unsigned long long tbase, unsigned long long pbase, unsigned long long bbase",
            "uint32_t a, e, lo, tt, t1, t2, dl, rec, qd, st; u32x4 x0, x1; unsigned long long sfull, sv;",
            backloop(),
            ['[%s] "=&v"(%s)' % (t, t) for t in ("a", "e", "lo", "tt", "t1", "t2", "dl", "rec", "qd", "x0", "x1")] +
            ['[st] "=&s"(st)', '[sfull] "=&s"(sfull)', '[sv] "=&s"(sv)'] +
            ['[%s] "+v"(S.%s)' % (t, t) for t in ("p", "floor", "pp", "leaf", "cum", "qcur", "qn", "va", "boff", "bfl", "pb0", "kmax", "ptop", "pa", "ba")] +
            ['[tick] "+s"(S.tick)'],
            ['[plim] "v"(plim)', '[rring] "v"(S.rring)', '[pring] "v"(S.pring)', '[oob] "v"(0x80000000u)', '[cbrk] "s"(0x800001u)', '[ceo] "s"(ceo)',
             '[tbase] "s"(tbase)', '[pbase] "s"(pbase)', '[bbase] "s"(bbase)'],
            '"vcc", "scc", "memory"')


if __name__ == "__main__":
    main()
    main2(sys.stdout)
    main3(sys.stdout)
    main4(sys.stdout)
    main5(sys.stdout)
    main6(sys.stdout)
