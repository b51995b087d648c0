// simulate.cpp — `kexc simulate --sim lockstep|backtrack`: the reference's two FST simulators, the user-visible CPU oracle of
// the language.  They run the NONDETERMINISTIC transducer of every stage directly — nothing of
// determinization, optimization, lowering, the path form or the HIP engine is involved — which is what makes them worth
// having beside `--sim sst` (the compiled program on the engine): three routes, one function (Tests/Regression.hs:45-53).
//   lockstep    src/KMC/SymbolicFST.hs:243-262 (rightClosure), :361-380 (run): all live paths advance together in priority
//               order; a state reached a second time in one closure is dropped; at end of input the first path that stands
//               in a final state wins                                                           (simulateLockstep, Commands.hs:277-289)
//   backtrack   src/KMC/SymbolicFST.hs:407-429 (runBacktracking) over KMC/Backtracking.hs: depth first in priority order,
//               a (state, input index) pair that has been entered is never entered again (barrier / fBarrier: it either
//               failed or lies on the current ε-loop), success = final state at end of input     (simulateBacktrack, :291-302)
// A stage with register actions writes its token stream (automata.cpp) and the stream is replayed with the semantics of
// Kleenex/Actions.hs:28-38 — what `runAction . mconcat . map adjActionSem` does in both simulators.
#include <unordered_set>

#include "kexc.h"

namespace kexc {

namespace {

struct Path { std::string out; int q; };

// SymbolicFST.hs:243-262 — ordered ε-closure below q with output
void rightClosure(const FST& f, int q, const std::string& acc, std::vector<char>& vis, std::vector<Path>& res) {
  // (explicit stack: ε-chains can be as long as the program)
  struct Fr { int q; size_t k; size_t len; };
  std::string cur = acc;
  std::vector<Fr> st{{q, 0, acc.size()}};
  while (!st.empty()) {
    Fr& fr = st.back();
    const auto& es = f.eps[(size_t)fr.q];
    if (es.empty()) { res.push_back({cur, fr.q}); st.pop_back(); if (!st.empty()) cur.resize(st.back().len); continue; }
    if (fr.k == es.size()) { st.pop_back(); if (!st.empty()) cur.resize(st.back().len); continue; }
    const FST::Eps& e = es[fr.k++];
    if (vis[(size_t)e.to]) continue;
    vis[(size_t)e.to] = 1;
    cur.resize(fr.len);
    cur += e.out;
    st.push_back({e.to, 0, cur.size()});
  }
}

std::vector<Path> closeAll(const FST& f, const std::vector<Path>& paths) {
  std::vector<Path> res, one;
  std::vector<char> seen((size_t)f.nstates, 0), vis;
  for (const Path& p : paths) {
    one.clear(); vis.assign((size_t)f.nstates, 0);
    rightClosure(f, p.q, p.out, vis, one);
    for (Path& x : one) if (!seen[(size_t)x.q]) { seen[(size_t)x.q] = 1; res.push_back(std::move(x)); }
  }
  return res;
}

uint8_t symOut(const FST& f, const FST::Sym& e, uint8_t b) { return e.tbl >= 0 ? f.tables[(size_t)e.tbl][b] : b; }

bool lockstep(const FST& f, const std::string& in, std::string& out) {
  std::vector<Path> paths = closeAll(f, {{"", f.init}});
  for (unsigned char b : in) {
    std::vector<Path> next;
    for (Path& p : paths)
      for (const FST::Sym& e : f.sym[(size_t)p.q])
        if (e.pred.has(b)) { Path n{p.out, e.to}; if (e.copy) n.out.push_back((char)symOut(f, e, b)); next.push_back(std::move(n)); }
    paths = closeAll(f, next);
    if (paths.empty()) return false;
  }
  for (Path& p : paths) if (f.is_final[(size_t)p.q]) { out = std::move(p.out); return true; }
  return false;
}

bool backtrack(const FST& f, const std::string& in, std::string& out) {
  struct Fr { int q; size_t i, k, len; };
  std::unordered_set<uint64_t> entered;
  auto enter = [&](int q, size_t i) { return entered.insert(((uint64_t)(uint32_t)q << 40) | (uint64_t)i).second; };
  std::string cur;
  std::vector<Fr> st;
  enter(f.init, 0);
  st.push_back({f.init, 0, 0, 0});
  while (!st.empty()) {
    Fr& fr = st.back();
    cur.resize(fr.len);
    const auto& es = f.eps[(size_t)fr.q];
    const auto& ss = f.sym[(size_t)fr.q];
    if (es.empty() && ss.empty()) {   // a state without edges: `pure mempty`, then `<* eof`
      if (fr.i == in.size() && f.is_final[(size_t)fr.q]) { out = cur; return true; }
      st.pop_back(); continue;
    }
    if (!es.empty()) {
      if (fr.k == es.size()) { st.pop_back(); continue; }
      const FST::Eps& e = es[fr.k++];
      if (!enter(e.to, fr.i)) continue;
      cur += e.out;
      st.push_back({e.to, fr.i, 0, cur.size()});
      continue;
    }
    if (fr.k == ss.size() || fr.i == in.size()) { st.pop_back(); continue; }
    const FST::Sym& e = ss[fr.k++];
    const uint8_t b = (uint8_t)in[fr.i];
    if (!e.pred.has(b) || !enter(e.to, fr.i + 1)) continue;
    if (e.copy) cur.push_back((char)symOut(f, e, b));
    st.push_back({e.to, fr.i + 1, 0, cur.size()});
  }
  return false;
}

// Kleenex/Actions.hs:28-38 on the token stream of kxp_format.h; false = "non-singleton stack on termination"
bool replayActions(const std::string& t, std::string& out) {
  std::map<int, std::string> store;
  std::vector<std::string> stack(1);
  for (size_t i = 0; i < t.size();) {
    const uint8_t b = (uint8_t)t[i++];
    if (b != KXP_ESC) { stack.back().push_back((char)b); continue; }
    if (i >= t.size()) break;
    const uint8_t k = (uint8_t)t[i++];
    if (k == KXP_ESC) stack.back().push_back((char)0xFF);
    else if (k == KXP_TOK_PUSH) stack.emplace_back();
    else {
      if (i >= t.size()) break;
      const int r = (uint8_t)t[i++];
      if (k == KXP_TOK_POP) { if (stack.size() > 1) { store[r] = std::move(stack.back()); stack.pop_back(); } }
      else { stack.back() += store[r]; store[r].clear(); }
    }
  }
  if (stack.size() != 1) return false;
  out = std::move(stack[0]);
  return true;
}

}  // namespace

// 0 = accepted (`out` holds the pipeline's output); 1 = some stage rejects its input; 2 = malformed action program
int simulateFST(const std::string& src, const std::string& srcname, const Options& o, bool backtracking, const std::string& input, std::string& out) {
  std::vector<std::pair<FST, bool>> stages;
  if (o.regex) {   // (the FST simulators run the regex's own transducer — tuTransducers — which copies what it reads)
    RProg rp = parseRegexProgram(src, srcname);
    stages.push_back({constructTransducer(rp, rp.pipeline[0], false), false});
  } else {
    RProg rp = desugar(parseKleenex(src, srcname));
    for (int start : rp.pipeline) {
      const bool acts = stageHasActions(rp, start);
      stages.push_back({constructTransducer(rp, start, acts), acts});
    }
  }
  std::string cur = input;
  for (auto& [f, acts] : stages) {
    std::string nxt;
    if (!(backtracking ? backtrack(f, cur, nxt) : lockstep(f, cur, nxt))) return 1;
    if (acts) { std::string fin; if (!replayActions(nxt, fin)) return 2; nxt = std::move(fin); }
    cur = std::move(nxt);
  }
  out = std::move(cur);
  return 0;
}

}  // namespace kexc
