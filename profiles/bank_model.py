"""LDS bank-conflict model of the table reads (CPU, numpy): how many LDS passes a wave's fwd / ent reads take with the engine's
table layout, and with a placement that puts the HOT entries of a run into distinct banks.  Lanes of a wave stand at
independent positions of the input, so a wave-read is modelled as 64 independent draws from the run's empirical access
distribution; the LDS serves 32 lanes per pass and one distinct dword per bank and pass (the probe figures of
profiles/r02b_lds_cost_table.txt: conflict-free 2.5 cycles, random 5.4 ~ 0.5 + the two half-waves' maximal bank loads).
  python profiles/bank_model.py PROGRAM [NBYTES]"""
import sys, numpy as np, collections
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import kxp
from kleenexlang_amd import compile_file, workloads
name = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 19
st = kxp.parse(compile_file(name))[0]
d = np.frombuffer(workloads.generate(workloads.PROGRAM_INPUT[name], N, seed=3), dtype=np.uint8)
n = len(d); C = st.nclasses; cls = st.cls[d]
q = st.q0; states = np.zeros(n + 1, dtype=np.int64)
for i in range(n):
    states[i] = q; q = int(st.delta[q, cls[i]])
states[n] = q
rows = st.pback[states[:n], cls].astype(np.int64)
leaf = int(st.fin_leaf[q]); lin = np.zeros(n, dtype=np.int64)     # leaf the step is entered with (index into its row)
for i in range(n - 1, -1, -1):
    lin[i] = leaf; leaf = int(st.back[rows[i], leaf]) & 0xFF
# engine layout (parseStage): fwd word index = 64 + q*C + c ; ent rows packed in order, row length = last live leaf + 1
nback = st.back.shape[0]
rl = [max([j + 1 for j in range(st.maxleaves) if st.back[b, j] != 0xFFFFFFFF] + [1]) for b in range(nback)]
rowoff = np.concatenate([[0], np.cumsum(rl)])[:-1]
off_ent_w = 64 + (st.nstates + 1) * C
fwd_w = 64 + states[:n] * C + cls
ent_w = off_ent_w + rowoff[rows] + lin
rng = np.random.default_rng(1)
def passes(words, trials=4000):
    """mean over random waves of sum over the two half-waves of the max number of distinct dwords in one bank"""
    tot = 0
    for _ in range(trials):
        w = words[rng.integers(0, len(words), 64)]
        for h in (w[:32], w[32:]):
            u = np.unique(h)
            tot += np.bincount(u % 64, minlength=64).max()
    return tot / trials
def greedy(items, weight, span):
    """items -> bank offset so that heavy items avoid each other: item = (key, list of relative dword offsets); returns {key: start bank}"""
    load = np.zeros(64); place = {}
    for k in sorted(items, key=lambda k: -weight[k]):
        best = min(range(64), key=lambda b: (max(load[(b + o) % 64] for o in items[k]), b))
        place[k] = best
        for o in items[k]: load[(best + o) % 64] += weight[k] / len(items[k])
    return place
print(name, "n", n, "fwd: distinct entries", len(np.unique(fwd_w)), " ent: distinct entries", len(np.unique(ent_w)))
print("  passes per wave-read now:      fwd %.2f   ent %.2f   (conflict-free = 2, uniform random = %.2f)" % (passes(fwd_w), passes(ent_w), passes(rng.integers(0, 3456, n))))
# optimised: states renumbered so that each state's hot classes land on free banks; rows placed likewise
cs = collections.Counter(zip(states[:n].tolist(), cls.tolist()))
it = collections.defaultdict(list); wt = collections.Counter()
for (s, c), k in cs.items(): it[s].append(c); wt[s] += k
pl = greedy(it, wt, C)
# a state's bank = (64 + pi(s)*C) % 64 must equal pl[s]: pi(s)*C = pl[s] (mod 64) has a solution for odd C; else approximate by padding
fwd_opt = np.array([pl[s] for s in states[:n]]) + cls
cr = collections.Counter(zip(rows.tolist(), lin.tolist()))
it2 = collections.defaultdict(list); wt2 = collections.Counter()
for (r, l), k in cr.items(): it2[r].append(l); wt2[r] += k
pl2 = greedy(it2, wt2, st.maxleaves)
ent_opt = np.array([pl2[r] for r in rows]) + lin
# (distinctness must be kept: two different entries in one bank conflict even at equal bank offset -> encode identity above bit 6)
fwd_id = (states[:n] * C + cls) * 64 + fwd_opt % 64
ent_id = (rowoff[rows] + lin) * 64 + ent_opt % 64
def passes_id(ids, trials=4000):
    tot = 0
    for _ in range(trials):
        w = ids[rng.integers(0, len(ids), 64)]
        for h in (w[:32], w[32:]):
            u = np.unique(h)
            tot += np.bincount(u % 64, minlength=64).max()
    return tot / trials
print("  passes with hot entries placed: fwd %.2f   ent %.2f" % (passes_id(fwd_id), passes_id(ent_id)))
