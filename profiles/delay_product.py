"""Round 5 prototype: the K-delayed forward transducer of a path-form stage (fixed delay K: step s emits the constant of step
s-K and the copy of step s-K+1).  Prints the number of product states, escapes (contexts that K symbols do not resolve).
  python profiles/delay_product.py PROGRAM K"""
import sys, numpy as np, collections
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import kxp
from kleenexlang_amd import compile_file
name = sys.argv[1]; K = int(sys.argv[2])
st = kxp.parse(compile_file(name) if not name.endswith(".kex") else compile_file(name))[0]
back = st.back
def kind(e):  # (copy, const id, table)
    return ((e >> 8) & 1, (e >> 9) & 0x7FFF, e >> 24)
def norm(g):  # tuple over leaves -> value if constant
    s = set(g)
    return ("v", g[0]) if len(s) == 1 else ("f", tuple(g))
# product state: (q, pend) pend = tuple of K items, item j = step s-K+j; each ("v",kind) or ("f",tuple over leaves(q))
nl0 = int(st.nleaves[st.q0])
init = tuple(norm([(0, int(st.init_const[l]), 0) for l in range(nl0)]) if j == K - 1 else ("v", (0, 0, 0)) for j in range(K))
start = (st.q0, init)
ids = {start: 0}; todo = [start]; trans = {}; nesc = 0; ncopyesc = 0
while todo:
    s = todo.pop(); q, pend = s
    for c in range(st.nclasses):
        t = int(st.delta[q, c])
        if t == 0xFFFF: continue
        r = int(st.pback[q, c]); nl = int(st.nleaves[t])
        par = [int(back[r, l]) & 0xFF for l in range(nl)]
        newp = []
        for it in pend:
            newp.append(it if it[0] == "v" else norm([it[1][p] for p in par]))
        newp.append(norm([kind(int(back[r, l])) for l in range(nl)]))
        emit_const = newp[0]          # step s-K: must be resolved now
        esc = emit_const[0] != "v"
        # copy flag of step s-K+1 must be resolved
        nxt = newp[1]
        if nxt[0] == "f" and len(set(k[0] for k in nxt[1])) > 1: esc = True; ncopyesc += 1
        if esc: nesc += 1; continue
        ns = (t, tuple(newp[1:]))
        if ns not in ids: ids[ns] = len(ids); todo.append(ns)
        trans[(ids[s], c)] = ids[ns]
print(name, "K", K, "original states", st.nstates, "product states", len(ids), "transitions", len(trans), "escapes", nesc, "(copy-flag escapes", ncopyesc, ")",
      "table bytes", (len(ids) + 1) * st.nclasses * 4)
