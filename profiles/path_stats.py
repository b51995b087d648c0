"""Statistics of the path form along a run (CPU, numpy): leaves per state, distinct kinds, candidates per 64-byte piece and how
often a piece's backward map is constant.  Evidence for DESIGN §5d (why k_backlen cannot be fused into a piece-parallel k_emit).
  python profiles/path_stats.py PROGRAM [NBYTES]"""
import sys, numpy as np, collections
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import kxp
from kleenexlang_amd import compile_file, workloads
name = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
blob = compile_file(name)
st = kxp.parse(blob)[0]
data = workloads.generate(workloads.PROGRAM_INPUT[name], N, seed=3)
d = np.frombuffer(data, dtype=np.uint8)
print(name, "states", st.nstates, "classes", st.nclasses, "maxleaves", st.maxleaves, "nback", st.back.shape, "npconsts", len(st.pconst_off) - 1)
# forward
q = st.q0; n = len(d)
states = np.zeros(n + 1, dtype=np.int32)
cls = st.cls[d]
delta = st.delta
for i in range(n):
    states[i] = q
    q = int(delta[q, cls[i]])
    assert q != 0xFFFF
states[n] = q
rows = st.pback[states[:n], cls]
nl = st.nleaves[states]
print("nleaves hist along run:", np.bincount(nl))
print("distinct states visited", len(set(states.tolist())), "distinct rows", len(set(rows.tolist())))
# backward true path
leaf = int(st.fin_leaf[q]); leaves = np.zeros(n + 1, dtype=np.int32); leaves[n] = leaf
back = st.back
dl = np.zeros(n, dtype=np.int32); hc = np.zeros(n, dtype=np.int8); cp = np.zeros(n, dtype=np.int8)
plen = np.diff(st.pconst_off)
for i in range(n - 1, -1, -1):
    e = int(back[rows[i], leaf])
    leaf = e & 0xFF; leaves[i] = leaf
    c = e >> 9
    cp[i] = (e >> 8) & 1; hc[i] = 1 if plen[c] > 0 else 0
    dl[i] = cp[i] + plen[c]
print("out/in", dl.sum() / n, "copy frac", cp.mean(), "const steps frac", hc.mean(), "const bytes / in", (dl.sum() - cp.sum()) / n)
print("dl hist", np.bincount(dl)[:40])
# distinct entry kinds (copy, pconst)
kinds = set()
for r in range(back.shape[0]):
    for l in range(st.maxleaves):
        e = int(back[r, l])
        if e != 0xFFFFFFFF: kinds.add(((e >> 8) & 1, e >> 9))
print("distinct kinds (copy,const) in table", len(kinds), " used on data:", len(set(zip(cp.tolist(), [int(back[rows[i], leaves[i+1]]) >> 9 for i in range(n)]))))
# piece maps: for each 64-byte piece, map end leaf -> start leaf for all candidates; constant?
P = 64
nconst_map = 0; npieces = 0; ncand = collections.Counter(); merged_at = []
for p0 in range(0, n - P + 1, P):
    qe = states[p0 + P]; nc = int(st.nleaves[qe])
    cur = list(range(nc)); mstep = None
    for k, i in enumerate(range(p0 + P - 1, p0 - 1, -1)):
        cur = [int(back[rows[i], c]) & 0xFF for c in cur]
        if mstep is None and len(set(cur)) == 1: mstep = k + 1
    npieces += 1; ncand[nc] += 1
    if len(set(cur)) == 1: nconst_map += 1; merged_at.append(mstep)
print("pieces", npieces, "constant piece-map frac", nconst_map / npieces, "candidates hist", sorted(ncand.items()))
if merged_at: print("merge step (when merged) mean", np.mean(merged_at), "pcts", np.percentile(merged_at, [10, 50, 90]))
# do different candidates give different lengths when not merged?
