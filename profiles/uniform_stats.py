"""Round 5: how much of a run is FORWARD-DETERMINED?  A transition row is output-uniform when every live leaf of its target
appends the same (copy, constant); a step on a non-uniform row is a DECISION, resolved once the forward-composed leaf map is
constant.  Prints the share of uniform steps, decisions per 64-byte piece and the distribution of decision delays.
  python profiles/uniform_stats.py PROGRAM [NBYTES]"""
import sys, numpy as np, collections
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import kxp
from kleenexlang_amd import compile_file, workloads
name = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
blob = compile_file(name)
st = kxp.parse(blob)[0]
data = workloads.generate(workloads.PROGRAM_INPUT[name], N, seed=3)
d = np.frombuffer(data, dtype=np.uint8)
n = len(d); cls = st.cls[d]; q = st.q0
states = np.zeros(n + 1, dtype=np.int32)
for i in range(n):
    states[i] = q; q = int(st.delta[q, cls[i]])
states[n] = q
rows = st.pback[states[:n], cls]
back = st.back
# static: which (state,class) transitions are uniform / map-constant
nrows_total = 0; nuni = 0
uni_row = {}
for s in range(st.nstates):
    for c in range(st.nclasses):
        t = int(st.delta[s, c])
        if t == 0xFFFF: continue
        r = int(st.pback[s, c]); nl = int(st.nleaves[t])
        ents = [int(back[r, l]) for l in range(nl)]
        kinds = set(((e >> 8) & 1, st.pool[int(st.pconst_off[(e>>9)&0x7FFF]):int(st.pconst_off[((e>>9)&0x7FFF)+1])], e>>24) for e in ents if e != 0xFFFFFFFF)
        parents = set(e & 0xFF for e in ents if e != 0xFFFFFFFF)
        uni_row[(s, c)] = (len(kinds) == 1, len(parents) == 1)
        nrows_total += 1; nuni += len(kinds) == 1
print(name, "transitions", nrows_total, "output-uniform", nuni, "(%.1f %%)" % (100 * nuni / nrows_total))
uni = np.array([uni_row[(int(states[i]), int(cls[i]))][0] for i in range(n)])
mc = np.array([uni_row[(int(states[i]), int(cls[i]))][1] for i in range(n)])
print("steps on uniform rows: %.4f   steps on map-constant rows: %.4f" % (uni.mean(), mc.mean()))
dec = np.nonzero(~uni)[0]
print("decisions:", len(dec), " per 64 B piece: %.3f" % (len(dec) / (n / 64)))
delays = []; maxl = 0
for t in dec[:20000]:
    # forward composition from t+1: M[l'] = leaf at t+1 that l' descends from
    nl = int(st.nleaves[states[t + 1]]); M = list(range(nl)); k = t + 1; maxl = max(maxl, nl)
    while len(set(M)) > 1 and k < n:
        r = rows[k]; nl2 = int(st.nleaves[states[k + 1]])
        M = [M[int(back[r, l]) & 0xFF] for l in range(nl2)]
        k += 1
    if len(set(M)) > 1:  # ran into EOF: final leaf decides
        pass
    delays.append(k - (t + 1))
delays = np.array(delays)
if len(delays):
    print("decision delay (steps after the decision until its leaf is known): mean %.2f max %d  pcts" % (delays.mean(), delays.max()),
          np.percentile(delays, [50, 90, 99, 99.9]), " leaves at a decision ≤", maxl)
    print("delay hist", np.bincount(delays)[:24])
if "-v" in sys.argv:
    seen = collections.Counter()
    for i in range(n):
        if not uni[i]: seen[(int(states[i]), int(cls[i]))] += 1
    for (s, c), cnt in seen.most_common(8):
        t = int(st.delta[s, c]); r = int(st.pback[s, c]); nl = int(st.nleaves[t])
        print("state", s, "class", c, "->", t, "visits", cnt)
        for l in range(nl):
            e = int(back[r, l])
            if e == 0xFFFFFFFF: print("   leaf", l, "dead"); continue
            pc = (e >> 9) & 0x7FFF
            print("   leaf", l, "parent", e & 0xFF, "copy", (e >> 8) & 1, "const", repr(st.pool[int(st.pconst_off[pc]):int(st.pconst_off[pc + 1])]))
# output-resolution delay: smallest k such that the OUTPUT of step t is the same for every leaf alive after step t+k
def outkind(e):
    pc = (e >> 9) & 0x7FFF
    return ((e >> 8) & 1, st.pool[int(st.pconst_off[pc]):int(st.pconst_off[pc + 1])], e >> 24)
ks = []
lim = min(n - 1100, 60000)
for t in range(lim):
    r = rows[t]; nl = int(st.nleaves[states[t + 1]])
    G = [outkind(int(back[r, l])) for l in range(nl)]   # per leaf at t+1: output of step t
    k = 0; j = t + 1
    while len(set(G)) > 1:
        r2 = rows[j]; nl2 = int(st.nleaves[states[j + 1]])
        G = [G[int(back[r2, l]) & 0xFF] for l in range(nl2)]
        j += 1; k += 1
    ks.append(k)
ks = np.array(ks)
print("output-resolution delay k: hist", np.bincount(ks)[:16], " max", ks.max(), " mean", ks.mean(), " P(k>=2) %.5f" % (ks >= 2).mean(), " P(k>=3) %.5f" % (ks >= 3).mean())
