"""Lock-step simulation of the nondeterministic transducer — an independent CPU oracle.

TEST INFRASTRUCTURE.  Restates the reference's `--sim=lockstep` simulator
(src/KMC/SymbolicFST.hs:243-262 rightClosure, :361-380 run): all live paths are advanced in
priority order, a state reached a second time is dropped (the earlier, higher-priority path
wins), and at end of input the first path standing in a final state gives the output.  It shares
the parser/desugarer/transducer construction with the compiler but nothing of determinization,
`optimize`, lowering, the path form or the engines — so agreement with them on random programs
and inputs pins the greedy-leftmost disambiguation independently.  Pure Python: small inputs only.
"""


def _right_closure(fst, q):
    """Ordered ε-closure with output (SymbolicFST.hs:243-262)."""
    eps = fst["eps"]
    out = []

    def go(vis, acc, s):
        edges = eps[s]
        if not edges:
            out.append((acc, s))
            return
        for o, t in edges:
            if t in vis:
                continue
            vis.add(t)
            go(vis, acc + bytes(o), t)

    go(set(), b"", q)
    return out


def _close(fst, paths):
    seen, res = set(), []
    for acc, q in paths:
        for o, t in _right_closure(fst, q):
            if t in seen:
                continue
            seen.add(t)
            res.append((acc + o, t))
    return res


def run_stage(fst, data):
    """bytes → bytes, or None when no path accepts (a match error in the compiled program)."""
    sym = fst["sym"]
    final = set(fst["final"])
    paths = _close(fst, [(b"", fst["init"])])
    for b in data:
        stepped = []
        for acc, q in paths:
            for edge in sym[q]:
                ranges, copy, t = edge[:3]
                tbl = edge[3] if len(edge) > 3 else -1     # >= 0: the symbol leaves through that table (CodeArg / AppendTblI)
                if any(lo <= b <= hi for lo, hi in ranges):
                    stepped.append((acc + (bytes([fst["tables"][tbl][b] if tbl >= 0 else b]) if copy else b""), t))
        paths = _close(fst, stepped)
        if not paths:
            return None
    for acc, q in paths:
        if q in final:
            return acc
    return None


def run(fsts, data):
    cur = bytes(data)
    for f in fsts:
        cur = run_stage(f, cur)
        if cur is None:
            return None
        if f.get("tokens"):          # the stage has register actions: its output is a token stream
            cur = replay_actions(cur)
    return cur


def replay_actions(tokens):
    """The action post-pass on a token stream (include/kxp_format.h), stated directly from the reference's semantics
    (src/KMC/Kleenex/Actions.hs:28-38): state = (register store, stack of buffers), start (empty, [empty]);
    inj w appends to the top buffer; psh pushes an empty buffer; pop r stores the top buffer in r and pops it;
    wr r appends register r to the top buffer and clears r.  The result is the bottom buffer."""
    store, stack = {}, [bytearray()]
    i, n = 0, len(tokens)
    while i < n:
        b = tokens[i]
        i += 1
        if b != 0xFF:
            stack[-1].append(b)
            continue
        k = tokens[i]
        i += 1
        if k == 0xFF:
            stack[-1].append(0xFF)
        elif k == 0:
            stack.append(bytearray())
        else:
            r = tokens[i]
            i += 1
            if k == 1:
                store[r] = stack.pop()
            else:
                stack[-1] += store.get(r, b"")
                store[r] = bytearray()
    return bytes(stack[0])
