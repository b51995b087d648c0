"""Offline placement for the KX_TUNE_FILE experiment: the greedy of profiles/bank_model.py turned into a state permutation and
a row order / start banks.  python profiles/bank_place.py PROGRAM OUTFILE [NBYTES]"""
import sys, numpy as np, collections
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import kxp
from kleenexlang_amd import compile_file, workloads
name, outf = sys.argv[1], sys.argv[2]; N = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 18
st = kxp.parse(compile_file(name))[0]
d = np.frombuffer(workloads.generate(workloads.PROGRAM_INPUT[name], N, seed=3), dtype=np.uint8)
n = len(d); C = st.nclasses; cls = st.cls[d]
q = st.q0; states = np.zeros(n + 1, dtype=np.int64)
for i in range(n):
    states[i] = q; q = int(st.delta[q, cls[i]])
rows = st.pback[states[:n], cls].astype(np.int64)
leaf = int(st.fin_leaf[q]) if st.fin_leaf[q] != 0xFF else 0; lin = np.zeros(n, dtype=np.int64)
for i in range(n - 1, -1, -1):
    lin[i] = leaf; leaf = int(st.back[rows[i], leaf]) & 0xFF
# states: hot ones (by weight) to the free position whose banks are least loaded
cs = collections.Counter(zip(states[:n].tolist(), cls.tolist()))
hot = collections.defaultdict(dict)
for (s, c), k in cs.items(): hot[s][c] = k
load = np.zeros(64); free = set(range(st.nstates)); perm = [-1] * st.nstates
for s in sorted(hot, key=lambda s: -sum(hot[s].values())):
    best = min(sorted(free), key=lambda pos: max(load[(pos * C + c) % 64] for c in hot[s]))
    perm[s] = best; free.discard(best)
    for c, k in hot[s].items(): load[(best * C + c) % 64] += k
rest = sorted(free)
for s in range(st.nstates):
    if perm[s] < 0: perm[s] = rest.pop(0)
# rows: hot ones first, each padded to its best start bank
cr = collections.Counter(zip(rows.tolist(), lin.tolist()))
hr = collections.defaultdict(dict)
for (r, l), k in cr.items(): hr[r][l] = k
nback = st.back.shape[0]
load2 = np.zeros(64); order = []; want = [255] * nback
for r in sorted(hr, key=lambda r: -sum(hr[r].values())):
    b = min(range(64), key=lambda b: max(load2[(b + l) % 64] for l in hr[r]))
    want[r] = b; order.append(r)
    for l, k in hr[r].items(): load2[(b + l) % 64] += k
order += [r for r in range(nback) if r not in hr]
with open(outf, "w") as f:
    f.write(" ".join(map(str, perm)) + "\n%d\n" % nback + " ".join(map(str, order)) + "\n" + " ".join(map(str, want)) + "\n")
print(name, "hot states", len(hot), "hot rows", len(hr), "of", nback)
