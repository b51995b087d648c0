"""Pure-Python reader of KXP blobs (include/kxp_format.h) + a CPU stand-in for one shard.

Test infrastructure: lets the sharded hand-off protocol (kleenexlang_amd/sharded.py) run in
multi-process gloo tests without a GPU.  Small inputs only (pure Python loops)."""
import struct

import numpy as np

NO_STATE = 0xFFFF
NO_LEAF = 0xFF
NOFAIL = 0xFFFFFFFFFFFFFFFF


def _pad4(n):
    return (n + 3) & ~3


class Stage:
    pass


def parse(blob):
    assert blob[:8] == b"KXPBLOB1"
    ver, ns, il = struct.unpack_from("<III", blob, 8)
    off = 20 + _pad4(il)
    stages = []
    for _ in range(ns):
        h = struct.unpack_from("<16I", blob, off); off += 64
        assert h[0] == 0x5453584B
        s = Stage()
        (s.nstates, s.nclasses, s.q0, s.nregs, nact, nops, nconsts, cpl, s.maxleaves, nback, npc, pcpl, nsync,
         s.sync_complete) = h[1:15]
        sc = s.nstates * s.nclasses

        def take(dtype, count):
            nonlocal off
            a = np.frombuffer(blob, dtype=dtype, count=count, offset=off)
            off += _pad4(a.nbytes)
            return a
        s.cls = take(np.uint8, 256)
        s.delta = take(np.uint16, sc).reshape(s.nstates, s.nclasses)
        s.act = take(np.uint32, sc)
        s.final_act = take(np.uint32, s.nstates)
        s.act_off = take(np.uint32, nact + 1); s.ops = take(np.uint32, nops * 2)
        s.const_off = take(np.uint32, nconsts + 1); s.cpool = bytes(take(np.uint8, cpl))
        s.pback = take(np.uint32, sc).reshape(s.nstates, s.nclasses)
        s.nleaves = take(np.uint8, s.nstates)
        s.fin_leaf = take(np.uint8, s.nstates)
        s.back = take(np.uint32, nback * s.maxleaves).reshape(nback, s.maxleaves)
        s.pconst_off = take(np.uint32, npc + 1)
        s.pool = bytes(take(np.uint8, pcpl))
        s.init_const = take(np.uint32, s.maxleaves)
        s.sync_next = take(np.uint32, nsync * s.nclasses).reshape(nsync, s.nclasses)
        s.sync_state = take(np.uint32, nsync)
        s.actions = h[15]
        s.tables = None
        if h[15] & 2:   # KXP_STAGE_HAS_TABLES: u32 ntables ; u8 table[ntables][256]
            nt = int(take(np.uint32, 1)[0])
            s.tables = take(np.uint8, nt * 256).reshape(nt, 256)
        stages.append(s)
    return stages


class _Fwd:
    pass


class _Bwd:
    pass


class CpuShard:
    """Same phase protocol as kleenexlang_amd.host.Shard, evaluated sequentially on the CPU."""

    def __init__(self, stage, data, is_first, is_last):
        self.t, self.d, self.first, self.last = stage, bytes(data), is_first, is_last
        self.n = len(self.d)
        self.states = [None] * (self.n + 1)
        self.fail = NOFAIL

    def _const(self, c):
        return self.t.pool[self.t.pconst_off[c]:self.t.pconst_off[c + 1]]

    def _run(self, q, a, b):
        t = self.t
        for i in range(a, b):
            self.states[i] = q
            nq = int(t.delta[q, t.cls[self.d[i]]])
            if nq == NO_STATE:
                self.fail = min(self.fail, i)
                return None
            q = nq
        self.states[b] = q
        return q

    def _summary(self):
        f = _Fwd()
        f.synced = 1 if self.end is not None or self.fail != NOFAIL else 0
        f.end_state = self.end if self.end is not None else 0
        f.head_len = self.head
        f.fail_pos = self.fail
        return f

    def forward(self):
        t = self.t
        if self.first:
            self.head, self.c0 = 0, t.q0
            self.end = self._run(t.q0, 0, self.n)
        else:
            sid, pos, st = 0, 0, int(t.sync_state[0])
            while st == 0xFFFFFFFF and pos < self.n:
                sid = int(t.sync_next[sid, t.cls[self.d[pos]]]); pos += 1
                if sid == 0xFFFFFFFD:
                    st = sid; break
                st = int(t.sync_state[sid])
            if st < 0xFFFF:
                self.head, self.c0 = pos, st
                self.end = self._run(st, pos, self.n)
            else:
                self.head, self.c0, self.end = self.n, None, None
        return self._summary()

    def fix_head(self, incoming):
        if not self.first and self.head > 0:
            q = self._run(incoming, 0, self.head)
            if q is not None:
                if self.c0 is not None:
                    assert q == self.c0, "sync property violated"
                else:
                    self.end = q
        elif not self.first and self.n == 0:
            self.end = incoming
        if self.last and self.fail == NOFAIL and self.end is not None and self.t.fin_leaf[self.end] == NO_LEAF:
            self.fail = self.n
        return self._summary()

    def _walk(self, leaf):
        t, out = self.t, []
        for i in range(self.n - 1, -1, -1):
            q = self.states[i]
            e = int(t.back[t.pback[q, t.cls[self.d[i]]], leaf])
            assert e != 0xFFFFFFFF
            tb, pc = (e >> 24, (e >> 9) & 0x7FFF) if t.tables is not None else (0, e >> 9)
            sym = int(t.tables[tb - 1, self.d[i]]) if tb else self.d[i]
            piece = (bytes([sym]) if e & 0x100 else b"") + self._const(pc)
            out.append(piece)
            leaf = e & 0xFF
        return leaf, b"".join(reversed(out))

    def backward(self):
        b = _Bwd()
        nle = int(self.t.nleaves[self.end])
        self.walks = {e: self._walk(e) for e in range(nle)}
        if self.last:   # a last shard ends in the final state's leaf whatever the neighbour assumes
            starts = [self.walks[int(self.t.fin_leaf[self.end])][0]] * nle
        else:
            starts = [self.walks[e][0] for e in range(nle)]
        b.nleaves = nle
        b.constant = 1 if len(set(starts)) == 1 else 0
        b.start_leaf = bytes(starts + [0] * (256 - nle))
        return b

    def resolve(self, end_leaf):
        if self.last:
            end_leaf = int(self.t.fin_leaf[self.end])
        start, body = self.walks[end_leaf]
        if self.first:
            body = self._const(int(self.t.init_const[start])) + body
        self.out = body
        return len(body)

    def emit(self):
        return self.out
