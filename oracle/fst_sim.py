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
            for ranges, copy, t in sym[q]:
                if any(lo <= b <= hi for lo, hi in ranges):
                    stepped.append((acc + (bytes([b]) if copy else b""), t))
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
    return cur
