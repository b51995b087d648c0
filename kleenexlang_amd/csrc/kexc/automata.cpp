// automata.cpp — reduced grammar → FST → streaming string transducer.
//
//   constructTransducer   src/KMC/SymbolicFST/Transducer.hs:57-107 (+ projectTransducer :114-133)
//   determinize           src/KMC/Determinization.hs:38-257, src/KMC/TreeWriter.hs:30-119,
//                         src/KMC/SymbolicFST.hs:229-241,298-312 (tests), Theories.hs:58-80
//   optimizeSST           src/KMC/SymbolicSST.hs:180-331
//
// Besides the register form the reference produces, determinize() records for
// every transition the *path form*: for each leaf of the target path tree, the
// leaf of the source tree it extends and the output appended on the way.  The
// concatenation of a root→leaf path never changes except by appending at the
// leaf (LCP hoisting only moves atoms towards the root), so the program's
// output is the concatenation of these per-step suffixes along the surviving
// path — which is what the GPU engine evaluates.
#include <algorithm>
#include <deque>
#include <functional>
#include <tuple>

#include "kexc.h"

namespace kexc {

// ======================================================================= FST
// Register actions.  The reference runs a program with actions as two machines per stage: an oracle that codes the path
// taken and an action machine that replays it on a stack of buffers (ActionSST.hs:47-104, Actions.hs:28-38).  Here the
// transducer itself stays one machine whose OUTPUT carries the actions in band, and the stage is followed by an action
// interpreter (include/kxp_format.h, "action post-pass").  Token code, escape byte 0xFF:
//   FF FF      the byte 0xFF (from a constant or copied from the input)
//   FF 00      Push            FF 01 r   Pop r            FF 02 r   Write r
bool stageHasActions(const RProg& rp, int start) {
  std::set<int> seen; std::vector<int> todo{start};
  while (!todo.empty()) {
    int i = todo.back(); todo.pop_back();
    if (!seen.insert(i).second) continue;
    auto it = rp.decls.find(i);
    if (it == rp.decls.end()) continue;
    if (it->second.kind == RTerm::RConst && it->second.c.kind != 0) return true;
    for (int j : it->second.ids) todo.push_back(j);
  }
  return false;
}

FST constructTransducer(const RProg& rp, int start, bool tokens) {
  using Stack = std::vector<int>;
  auto getDecl = [&](int i) -> const RTerm& {
    auto it = rp.decls.find(i);
    if (it == rp.decls.end()) throw CompileError("internal error: identifier without declaration: " + std::to_string(i));
    return it->second;
  };
  // Transducer.hs:98-103 — contract RSeq heads and single-alternative RSum heads
  std::function<Stack(Stack)> follow = [&](Stack st) -> Stack {
    int guard = 0;
    while (!st.empty()) {
      const RTerm& d = getDecl(st[0]);
      if (d.kind == RTerm::RSeq) { Stack n(d.ids); n.insert(n.end(), st.begin() + 1, st.end()); st = std::move(n); }
      else if (d.kind == RTerm::RSum && d.ids.size() == 1) st[0] = d.ids[0];
      else break;
      if (++guard > 1000000 || st.size() > 100000) throw CompileError("grammar is not right-regular (unbounded continuation stack)");
    }
    return st;
  };
  struct Edge { Stack from; bool is_sym; ByteSet pred; bool copy; std::string out; Stack to; };
  std::set<Stack> states, ws;
  std::vector<Edge> edges;
  ws.insert(Stack{start});
  while (!ws.empty()) {
    Stack q = *ws.begin(); ws.erase(ws.begin());
    if (states.count(q)) continue;
    states.insert(q);
    if (states.size() > 2000000) throw CompileError("transducer too large (grammar not right-regular?)");
    if (q.empty()) continue;
    Stack rest(q.begin() + 1, q.end());
    const RTerm& d = getDecl(q[0]);
    switch (d.kind) {
      case RTerm::RConst: {
        std::string out;
        if (d.c.kind != 0) {
          if (!tokens)  // Commands.hs:165-168
            throw CompileError("Transducer contains action symbols - direct SST generation not supported");
          if (d.c.kind != 1 && (d.c.arg < 0 || d.c.arg >= (int)KXP_MAX_ACTION_REGS)) throw CompileError("too many registers (at most " + std::to_string(KXP_MAX_ACTION_REGS) + ")");
          out = d.c.kind == 1 ? std::string("\xFF\x00", 2) : std::string(d.c.kind == 2 ? "\xFF\x01" : "\xFF\x02", 2) + char(d.c.arg);
        } else if (tokens && (d.c.arg & 0xFF) == 0xFF) out = "\xFF\xFF";
        else out = std::string(1, char(d.c.arg));
        Stack t = follow(rest); ws.insert(t);
        edges.push_back({q, false, {}, false, out, t}); break;
      }
      case RTerm::RRead: {
        Stack t = follow(rest); ws.insert(t);
        edges.push_back({q, true, d.pred, d.copy, "", t}); break;
      }
      case RTerm::RSeq: {
        Stack n(d.ids); n.insert(n.end(), rest.begin(), rest.end());
        Stack t = follow(n); ws.insert(t);
        edges.push_back({q, false, {}, false, "", t}); break;
      }
      case RTerm::RSum:
        for (int j : d.ids) {  // alternative order = priority order
          Stack n{j}; n.insert(n.end(), rest.begin(), rest.end());
          Stack t = follow(n); ws.insert(t);
          edges.push_back({q, false, {}, false, "", t});
        }
        break;
    }
  }
  // enumerateStates (SymbolicFST.hs:314-326): number in Ord order of the stacks
  std::map<Stack, int> id;
  int n = 0;
  for (auto& s : states) id[s] = n++;
  FST f;
  f.nstates = n; f.init = id[Stack{start}];
  f.is_final.assign(n, 0);
  if (id.count(Stack{})) f.is_final[id[Stack{}]] = 1;
  f.eps.resize(n); f.sym.resize(n);
  for (auto& e : edges) {
    if (e.is_sym) f.sym[id[e.from]].push_back({e.pred, e.copy, id[e.to]});
    else f.eps[id[e.from]].push_back({e.out, id[e.to]});
  }
  if (tokens) {
    // a copied input byte 0xFF must leave the transducer escaped: a copying symbol edge whose predicate contains 0xFF is
    // split into (predicate without 0xFF, copy) and (0xFF, no copy) followed by the constant FF FF
    const int n0 = f.nstates;
    for (int q = 0; q < n0; ++q) {
      if (f.sym[q].size() != 1 || !f.sym[q][0].copy || !f.sym[q][0].pred.has(0xFF)) continue;
      const FST::Sym e = f.sym[q][0];
      f.sym[q].clear();
      auto fresh = [&]() { f.eps.emplace_back(); f.sym.emplace_back(); f.is_final.push_back(0); return f.nstates++; };
      ByteSet rest = e.pred.minus(ByteSet::single(0xFF));
      if (!rest.empty()) { int a = fresh(); f.sym[a].push_back({rest, true, e.to}); f.eps[q].push_back({"", a}); }
      int b = fresh(), c = fresh();
      f.sym[b].push_back({ByteSet::single(0xFF), false, c});
      f.eps[c].push_back({std::string("\xFF\xFF", 2), e.to});
      f.eps[q].push_back({"", b});
    }
  }
  return f;
}

// OracleMachine.hs:47-61.  `symsym`: a copying symbol edge over a predicate p with more than one member outputs the
// fixed-width code of the symbol's index in p (CodeArg p), every other symbol edge outputs nothing; `epseps`: the k-th of
// n > 1 ε-alternatives outputs the fixed-width code of k, a lone ε-edge nothing.  Digits are base 256 (Frontend.hs:117),
// so a code is one byte up to 256 alternatives (Util/Coding.hs:13-19,66-73: width = least w with 256^w >= n, big-endian).
// CodeArg p is a TABLE ATOM (round 3): the edge keeps its predicate and carries the table "symbol -> its index in p"
// (one byte: |p| <= 256), which the SST, the blob and the engine's output stage know as output = table[symbol]
// (AppendTblI, IL.hs:44; C.hs:421-430) — the program keeps its few byte classes instead of one per member of p.
FST oracleTransducer(const FST& f) {
  auto code = [](int n, int k) {
    int w = 0; long long cap = 1;
    while (cap < n) { cap *= 256; ++w; }
    std::string c((size_t)w, '\0');
    for (int i = w - 1; i >= 0; --i) { c[(size_t)i] = char(k & 0xFF); k >>= 8; }
    return c;
  };
  // The engine's entries name at most KXP_ENGINE_TABLES tables beside each other (kxp_format.h).  A regex with more distinct
  // multi-member predicates keeps table atoms for the LARGEST predicates (each of them would cost |p| byte classes written
  // out) and writes the others out the way rounds 1-2 wrote all of them: |p| single-symbol edges, each followed by a lone
  // ε-edge carrying that symbol's code — the same relation, symbol for symbol (`[a-c][d-f]…[v-x]` loads again).
  auto tableOf = [](const ByteSet& p) { std::array<uint8_t, 256> t{}; int idx = 0; for (int x = 0; x < 256; ++x) if (p.has(x)) t[(size_t)x] = (uint8_t)idx++; return t; };
  std::vector<std::pair<int, std::array<uint8_t, 256>>> cand;   // (−|p|, table), distinct
  for (int q = 0; q < f.nstates; ++q)
    for (const auto& e : f.sym[(size_t)q]) {
      if (!e.copy || e.pred.size() <= 1) continue;
      auto t = tableOf(e.pred);
      bool seen = false;
      for (auto& c : cand) seen = seen || c.second == t;
      if (!seen) cand.push_back({-e.pred.size(), t});
    }
  size_t keep = KXP_ENGINE_TABLES;
  if (const char* ev = getenv("KEXC_MAX_TABLE_ATOMS")) keep = (size_t)std::max(0, atoi(ev));
  std::stable_sort(cand.begin(), cand.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  if (cand.size() > keep) cand.resize(keep);
  FST o;
  o.nstates = f.nstates; o.init = f.init; o.is_final = f.is_final;
  o.eps.resize((size_t)f.nstates); o.sym.resize((size_t)f.nstates);
  auto fresh = [&]() { o.eps.emplace_back(); o.sym.emplace_back(); o.is_final.push_back(0); return o.nstates++; };
  for (auto& c : cand) o.tables.push_back(c.second);
  for (int q = 0; q < f.nstates; ++q) {
    const auto& es = f.eps[(size_t)q];
    for (size_t k = 0; k < es.size(); ++k)
      o.eps[(size_t)q].push_back({es.size() > 1 ? code((int)es.size(), (int)k) : std::string(), es[k].to});
    for (const auto& e : f.sym[(size_t)q]) {
      const int n = e.pred.size();
      if (!e.copy || n <= 1) { o.sym[(size_t)q].push_back({e.pred, false, e.to}); continue; }
      const auto t = tableOf(e.pred);
      size_t k = 0;
      while (k < o.tables.size() && o.tables[k] != t) ++k;
      if (k < o.tables.size()) { o.sym[(size_t)q].push_back({e.pred, true, e.to, (int)k}); continue; }
      int idx = 0;
      for (int x = 0; x < 256; ++x) {
        if (!e.pred.has(x)) continue;
        const int s = fresh();
        o.sym[(size_t)q].push_back({ByteSet::single(x), false, s});
        o.eps[(size_t)s].push_back({code(n, idx++), e.to});
      }
    }
  }
  return o;
}

// =============================================================== path trees
namespace {

struct Tree {  // TreeWriter.hs:30-33; `tag` rides along with the leaf value
  UpdateString out;
  bool tip = true;
  int st = 0, tag = -1;
  std::vector<Tree> kids;
};
using MTree = std::optional<Tree>;

void prepend(const UpdateString& w, Tree& t) { t.out.insert(t.out.begin(), w.begin(), w.end()); }

// x >>= f on trees (TreeWriter.hs:78-92): leaves are visited left to right
MTree bindTree(const Tree& t, const std::function<MTree(int, int)>& f) {
  if (t.tip) {
    MTree r = f(t.st, t.tag);
    if (!r) return std::nullopt;
    prepend(t.out, *r);
    return r;
  }
  std::vector<Tree> ts;
  for (auto& k : t.kids) { MTree r = bindTree(k, f); if (r) ts.push_back(std::move(*r)); }
  if (ts.empty()) return std::nullopt;
  if (ts.size() == 1) { prepend(t.out, ts[0]); return std::move(ts[0]); }
  Tree r; r.tip = false; r.out = t.out; r.kids = std::move(ts);
  return r;
}

// Determinization.hs:65-71 — hoist the longest common prefix of sibling outputs
void reduceTree(Tree& t) {
  if (t.tip) return;
  for (auto& k : t.kids) reduceTree(k);
  size_t n = t.kids[0].out.size();
  for (auto& k : t.kids) n = std::min(n, k.out.size());
  size_t l = 0;
  for (; l < n; ++l) {
    bool same = true;
    for (size_t i = 1; i < t.kids.size() && same; ++i) same = t.kids[i].out[l] == t.kids[0].out[l];
    if (!same) break;
  }
  if (l) {
    t.out.insert(t.out.end(), t.kids[0].out.begin(), t.kids[0].out.begin() + l);
    for (auto& k : t.kids) k.out.erase(k.out.begin(), k.out.begin() + l);
  }
}

Atom constAtom(const std::string& b) { Atom a; a.kind = Atom::CONST; a.bytes = b; return a; }
Atom varAtom(int v) { Atom a; a.kind = Atom::VAR; a.var = v; return a; }
Atom funcAtom(int f) { Atom a; a.kind = Atom::FUNC; a.func = f; return a; }

struct Determinizer {
  const FST& f;
  explicit Determinizer(const FST& fst) : f(fst) {}

  // Determinization.hs:82-93 — ε-closure below one leaf; `vis` is shared by the whole tree walk
  MTree genclosure(int q, int tag, std::set<int>& vis) {
    const auto& es = f.eps[q];
    if (es.empty()) { Tree t; t.st = q; t.tag = tag; return t; }
    std::vector<Tree> ts;
    for (auto& e : es) {
      if (vis.count(e.to)) continue;   // visit: a state reached a second time is dropped
      vis.insert(e.to);
      MTree sub = genclosure(e.to, tag, vis);
      if (!sub) continue;
      prepend({constAtom(e.out)}, *sub);
      ts.push_back(std::move(*sub));
    }
    if (ts.empty()) return std::nullopt;
    if (ts.size() == 1) return std::move(ts[0]);
    Tree r; r.tip = false; r.kids = std::move(ts);
    return r;
  }
  MTree closeTree(const Tree& t) {
    std::set<int> vis;
    MTree r = bindTree(t, [&](int q, int tag) { return genclosure(q, tag, vis); });
    if (r) reduceTree(*r);
    return r;
  }
  MTree consumeTree(const Tree& t, const ByteSet& p, int idx = 0) {  // Determinization.hs:108-114; idx = `inj i`
    std::set<int> vis;
    MTree r = bindTree(t, [&](int q, int tag) -> MTree {
      const FST::Sym* hit = nullptr;
      for (auto& e : f.sym[q]) if (p.subsetOf(e.pred)) {
        if (hit) throw CompileError("Stepping for FSTs with read-fanout greater than one is not supported yet");
        hit = &e;
      }
      if (!hit) return std::nullopt;
      if (vis.count(hit->to)) return std::nullopt;
      vis.insert(hit->to);
      Tree n; n.st = hit->to; n.tag = tag; n.out = {funcAtom(hit->copy ? 0 : 1)}; n.out[0].sym = idx; n.out[0].tbl = hit->copy ? hit->tbl : -1;
      return n;
    });
    if (r) reduceTree(*r);
    return r;
  }
  MTree eofTree(const Tree& t) {  // Determinization.hs:116-118,144-145
    std::set<int> vis;
    MTree r = bindTree(t, [&](int q, int tag) -> MTree {
      if (!f.is_final[q] || vis.count(q)) return std::nullopt;
      vis.insert(q);
      Tree n; n.st = q; n.tag = tag; return n;
    });
    if (r) reduceTree(*r);
    return r;
  }
};

// Theories.hs:58-80 — coarsest partition of the union refining every predicate
std::vector<ByteSet> coarsestPartition(const std::vector<ByteSet>& preds) {
  std::map<std::vector<bool>, ByteSet> blocks;
  for (int b = 0; b < 256; ++b) {
    std::vector<bool> sig(preds.size());
    bool any = false;
    for (size_t i = 0; i < preds.size(); ++i) { sig[i] = preds[i].has(b); any = any || sig[i]; }
    if (any) blocks[sig].add(b);
  }
  std::vector<ByteSet> out;
  for (auto& kv : blocks) out.push_back(kv.second);
  std::sort(out.begin(), out.end());
  return out;
}

void leavesOf(const Tree& t, std::vector<const Tree*>& out) {
  if (t.tip) { out.push_back(&t); return; }
  for (auto& k : t.kids) leavesOf(k, out);
}
void tagLeaves(Tree& t, int& n) {
  if (t.tip) { t.tag = n++; return; }
  for (auto& k : t.kids) tagLeaves(k, n);
}

// state identity = tree shape + leaf FST states (register names are positional)
void shapeKey(const Tree& t, std::string& k) {
  if (t.tip) { k += "L" + std::to_string(t.st) + ";"; return; }
  k += "(";
  for (auto& c : t.kids) shapeKey(c, k);
  k += ")";
}

}  // namespace

SST determinize(const FST& f, size_t table_word_cap) {
  Determinizer D(f);
  // Fail fast for a back end with a table limit (the engine: (states + 1) x classes <= 65535 words even in its BIG form).  The
  // byte classes are the coarsest partition refining every edge predicate, so the partition of the predicates SEEN SO FAR
  // only gets finer and the states only get more: once (states so far) x (classes so far) is over the limit the finished
  // machine is too (bench/kleenex/src/syntax.kex ran for 40 minutes before saying so).
  uint8_t cls_of[256] = {0}; int ncls_lb = 1;
  auto refine = [&](const ByteSet& p) {
    int split[256]; for (int c = 0; c < ncls_lb; ++c) split[c] = -2;   // -2 unseen, -1 seen outside p only / inside p only, >= 0 new class for members
    bool in0[256] = {false}, out0[256] = {false};
    for (int b = 0; b < 256; ++b) (p.has(b) ? in0 : out0)[cls_of[b]] = true;
    for (int c = 0, n = ncls_lb; c < n; ++c) if (in0[c] && out0[c]) split[c] = ncls_lb++;
    for (int b = 0; b < 256; ++b) if (p.has(b) && split[cls_of[b]] >= 0) cls_of[b] = (uint8_t)split[cls_of[b]];
  };
  std::map<std::vector<int>, int> varIds;          // Var [Int] (reversed path) → register id
  auto varOf = [&](const std::vector<int>& path) {
    auto it = varIds.find(path);
    if (it != varIds.end()) return it->second;
    int id = (int)varIds.size(); varIds[path] = id; return id;
  };
  varOf({});                                       // root = designated output register 0

  // abstract (Determinization.hs:165-177): name nodes by position; returns the state skeleton
  std::function<void(const Tree&, std::vector<int>&, std::map<int, UpdateString>*, Tree&)> abstractT =
      [&](const Tree& t, std::vector<int>& pos, std::map<int, UpdateString>* kappa, Tree& skel) {
        int v = varOf(pos);
        if (kappa) (*kappa)[v] = t.out;
        skel.out = {varAtom(v)};
        skel.tip = t.tip; skel.st = t.st; skel.tag = -1;
        skel.kids.clear();
        if (!t.tip) {
          skel.kids.resize(t.kids.size());
          for (size_t m = 0; m < t.kids.size(); ++m) {
            pos.insert(pos.begin(), (int)m);       // Var (m:v)
            abstractT(t.kids[m], pos, kappa, skel.kids[m]);
            pos.erase(pos.begin());
          }
        }
      };

  SST sst;
  std::map<std::string, int> ids;
  std::vector<Tree> skels;                         // abstracted tree of every SST state (outputs = own VAR)
  std::deque<int> work;
  auto intern = [&](const Tree& skel) {
    std::string k; shapeKey(skel, k);
    auto it = ids.find(k);
    if (it != ids.end()) return it->second;
    int id = (int)skels.size();
    ids[k] = id; skels.push_back(skel); sst.states.emplace_back(); work.push_back(id);
    if (skels.size() > 60000) throw CompileError("SST has more than 60000 states");
    return id;
  };
  {
    Tree t0; t0.st = f.init; t0.out = {varAtom(0)};
    sst.init = intern(t0);
  }
  while (!work.empty()) {
    int sid = work.front(); work.pop_front();
    // closureAbstractTree (Determinization.hs:120-130); identity except for the initial state
    MTree tcl = D.closeTree(skels[sid]);
    if (!tcl) continue;                            // no live path: dead state
    int nl = 0; tagLeaves(*tcl, nl);
    std::vector<const Tree*> leaves; leavesOf(*tcl, leaves);
    sst.states[sid].nleaves = nl;
    if (sid == sst.init) {                         // path form of the initial closure
      std::function<void(const Tree&, std::string)> walk = [&](const Tree& t, std::string acc) {
        for (auto& a : t.out) if (a.kind == Atom::CONST) acc += a.bytes;
        if (t.tip) { sst.init_path.push_back(acc); return; }
        for (auto& k : t.kids) walk(k, acc);
      };
      walk(*tcl, "");
    }
    // final output (Determinization.hs:250)
    if (MTree e = D.eofTree(*tcl); e && e->tip) {
      sst.states[sid].is_final = true;
      sst.states[sid].final_upd = e->out;
      sst.states[sid].final_leaf = e->tag;
    }
    // tests: one per block of the coarsest partition of the leaves' predicates (--la=false)
    std::set<ByteSet> pset;
    for (auto* l : leaves) for (auto& e : f.sym[l->st]) pset.insert(e.pred);
    std::vector<ByteSet> preds(pset.begin(), pset.end());
    for (const ByteSet& p : coarsestPartition(preds)) {
      // consumeTreeMany (Determinization.hs:213-228): close → consume → close
      MTree a = D.closeTree(*tcl); if (!a) continue;
      MTree b = D.consumeTree(*a, p); if (!b) continue;
      MTree c = D.closeTree(*b); if (!c) continue;
      SSTEdge edge; edge.pred = p;
      if (table_word_cap) {
        if (ncls_lb < 256) refine(p);
        if ((skels.size() + 1) * (size_t)ncls_lb > table_word_cap)
          throw CompileError("program outside engine limits: more than " + std::to_string(skels.size()) + " SST states x " + std::to_string(ncls_lb) +
                             " byte classes (the engine's state table holds " + std::to_string(table_word_cap) + " transition words); --backend=c compiles it for the CPU");
      }
      // path form: per leaf of the new tree, origin leaf + the atoms after the VAR prefix
      std::function<void(const Tree&, UpdateString)> walk = [&](const Tree& t, UpdateString acc) {
        acc.insert(acc.end(), t.out.begin(), t.out.end());
        if (!t.tip) { for (auto& k : t.kids) walk(k, acc); return; }
        PathStep ps{t.tag, false, ""};
        bool seen_new = false;
        for (auto& at : acc) {
          if (at.kind == Atom::VAR) { if (seen_new) throw CompileError("internal: register after new output on a path"); continue; }
          seen_new = true;
          if (at.kind == Atom::FUNC) {
            if (!ps.bytes.empty() || ps.copy) throw CompileError("internal: symbol function not first on a path");
            ps.copy = at.func == 0; ps.tbl = at.tbl;
          } else ps.bytes += at.bytes;
        }
        edge.path.push_back(ps);
      };
      if (sid == sst.init) {
        // the initial tree is the only one that is not already closed: its closure output is
        // accounted for in init_path, so measure the step against the closed tree alone
        Tree marked = *tcl;
        std::function<void(Tree&)> mark = [&](Tree& t) { t.out = {varAtom(-1)}; for (auto& k : t.kids) mark(k); };
        mark(marked);
        MTree a2 = D.closeTree(marked);
        MTree b2 = a2 ? D.consumeTree(*a2, p) : std::nullopt;
        MTree c2 = b2 ? D.closeTree(*b2) : std::nullopt;
        if (!c2) throw CompileError("internal: path form diverges from register form");
        walk(*c2, {});
      } else walk(*c, {});
      // register form: abstract, specialize, normalize (Determinization.hs:165-204, SymbolicSST.hs:109-120)
      Tree skel; std::vector<int> pos; std::map<int, UpdateString> kappa;
      abstractT(*c, pos, &kappa, skel);
      bool single = p.size() == 1;
      for (auto& kv : kappa) {
        UpdateString norm;
        for (auto at : kv.second) {
          if (at.kind == Atom::FUNC) {
            if (at.func == 1) at = constAtom("");
            else if (single) at = constAtom(std::string(1, char(at.tbl >= 0 ? f.tables[(size_t)at.tbl][(size_t)p.first()] : p.first())));
          }
          if (at.kind == Atom::CONST) {
            if (at.bytes.empty()) continue;
            if (!norm.empty() && norm.back().kind == Atom::CONST) { norm.back().bytes += at.bytes; continue; }
          }
          norm.push_back(at);
        }
        kv.second = std::move(norm);
      }
      edge.upd = std::move(kappa);
      edge.to = intern(skel);
      sst.states[sid].edges.push_back(std::move(edge));
    }
  }
  for (auto& st : sst.states) {                    // normalize final updates
    UpdateString norm;
    for (auto& at : st.final_upd) {
      if (at.kind == Atom::CONST) {
        if (at.bytes.empty()) continue;
        if (!norm.empty() && norm.back().kind == Atom::CONST) { norm.back().bytes += at.bytes; continue; }
      }
      norm.push_back(at);
    }
    st.final_upd = std::move(norm);
  }
  sst.nregs = (int)varIds.size();
  sst.tables = f.tables;
  return sst;
}

// ================================================================ --la=true
// sstFromFST … singletonMode=False (Determinization.hs:233-257): the same saturation as determinize(), but a state's
// tests are words — prefixTests (SymbolicFST.hs:296-312) — and each test first kills the leaves whose longest
// deterministic prefix it does not entail.  Only the path form is kept: it is all the engine's back end needs, and
// it is what kexc_emit_pipeline takes from a front end that compiles with the reference's default flags.
WordSST determinizeWords(const FST& f) {
  Determinizer D(f);
  WordSST w;
  std::map<std::string, int> ids;
  std::vector<Tree> skels;
  std::deque<int> work;
  auto intern = [&](const Tree& skel) {
    std::string k; shapeKey(skel, k);
    auto it = ids.find(k);
    if (it != ids.end()) return it->second;
    int id = (int)skels.size();
    ids[k] = id; skels.push_back(skel); w.states.emplace_back(); work.push_back(id);
    if (skels.size() > 60000) throw CompileError("SST has more than 60000 states");
    return id;
  };
  std::function<void(const Tree&, Tree&)> skeleton = [&](const Tree& t, Tree& s) {
    s.out = {varAtom(0)}; s.tip = t.tip; s.st = t.st; s.tag = -1;
    s.kids.assign(t.kids.size(), Tree());
    for (size_t m = 0; m < t.kids.size(); ++m) skeleton(t.kids[m], s.kids[m]);
  };
  // rightInputClosure (SymbolicFST.hs:281-294): the ε-free states below q, no output
  std::map<int, std::set<int>> clMemo;
  auto inputClosure = [&](int q) -> const std::set<int>& {
    auto it = clMemo.find(q);
    if (it != clMemo.end()) return it->second;
    std::set<int> out, vis{q};
    std::vector<int> st{q};
    while (!st.empty()) {
      int x = st.back(); st.pop_back();
      if (f.eps[x].empty()) { out.insert(x); continue; }
      for (auto& e : f.eps[x]) if (vis.insert(e.to).second) st.push_back(e.to);
    }
    return clMemo[q] = out;
  };
  auto predsOf = [&](const std::set<int>& ctx) {   // coarsestPredicateSet
    std::set<ByteSet> ps;
    for (int q : ctx) for (auto& e : f.sym[q]) ps.insert(e.pred);
    return coarsestPartition(std::vector<ByteSet>(ps.begin(), ps.end()));
  };
  auto stepAll = [&](const ByteSet& p, const std::set<int>& ctx) {   // SymbolicFST.hs:274-278
    std::set<int> out;
    for (int q : ctx) for (auto& e : f.sym[q]) if (p.subsetOf(e.pred)) { auto& c = inputClosure(e.to); out.insert(c.begin(), c.end()); }
    return out;
  };
  auto ldp = [&](std::set<int> ctx, int q) {   // SymbolicFST.hs:264-272; the cap only bounds a pathological cycle of symbol edges
    std::vector<ByteSet> word;
    while (word.size() < 255 && f.eps[q].empty() && f.sym[q].size() == 1) {
      const ByteSet& p = f.sym[q][0].pred;
      auto part = predsOf(ctx);
      if (std::find(part.begin(), part.end(), p) == part.end()) break;
      word.push_back(p);
      ctx = stepAll(p, ctx);
      q = f.sym[q][0].to;
    }
    return word;
  };
  {
    Tree t0; t0.st = f.init; t0.out = {varAtom(0)};
    w.init = intern(t0);
  }
  while (!work.empty()) {
    const int sid = work.front(); work.pop_front();
    MTree tcl = D.closeTree(skels[sid]);
    if (!tcl) continue;
    int nl = 0; tagLeaves(*tcl, nl);
    std::vector<const Tree*> leaves; leavesOf(*tcl, leaves);
    w.states[sid].nleaves = nl;
    if (sid == w.init) {
      std::function<void(const Tree&, std::string)> walk = [&](const Tree& t, std::string acc) {
        for (auto& a : t.out) if (a.kind == Atom::CONST) acc += a.bytes;
        if (t.tip) { w.init_path.push_back(acc); return; }
        for (auto& k : t.kids) walk(k, acc);
      };
      walk(*tcl, "");
    }
    if (MTree e = D.eofTree(*tcl); e && e->tip) w.states[sid].final_leaf = e->tag;
    // the closed tree with every output forgotten: what a step appends is then all that is on its paths
    Tree closed = *tcl;
    std::function<void(Tree&)> mark = [&](Tree& t) { t.out = {varAtom(0)}; for (auto& k : t.kids) mark(k); };
    mark(closed);
    // prefixTests (SymbolicFST.hs:296-312)
    std::set<int> ctx;
    for (auto* l : leaves) ctx.insert(l->st);
    std::vector<std::pair<std::vector<ByteSet>, int>> ldps;
    std::set<std::vector<ByteSet>> tests;
    for (const ByteSet& p : predsOf(ctx)) tests.insert({p});
    for (auto* l : leaves) { ldps.push_back({ldp(ctx, l->st), l->st}); tests.insert(ldps.back().first); }
    for (const auto& ps : tests) {
      if (ps.empty()) continue;
      std::set<int> kills;
      for (auto& [lw, q] : ldps) {   // entails: the leaf's prefix must be a prefix of the test
        bool ok = lw.size() <= ps.size();
        for (size_t i = 0; ok && i < lw.size(); ++i) ok = ps[i] == lw[i];
        if (!ok) kills.insert(q);
      }
      MTree cur = bindTree(closed, [&](int q, int tag) -> MTree {   // killTree (Determinization.hs:122-124,147-151)
        if (kills.count(q)) return std::nullopt;
        Tree n; n.st = q; n.tag = tag; return n;
      });
      if (cur) reduceTree(*cur);
      for (size_t i = 0; cur && i < ps.size(); ++i) {   // consumeTreeMany (Determinization.hs:213-228)
        MTree a = D.closeTree(*cur);
        MTree b = a ? D.consumeTree(*a, ps[i], (int)i) : std::nullopt;
        cur = b ? D.closeTree(*b) : std::nullopt;
      }
      if (!cur) continue;
      WordEdge edge; edge.word = ps;
      std::function<void(const Tree&, UpdateString)> walk = [&](const Tree& t, UpdateString acc) {
        acc.insert(acc.end(), t.out.begin(), t.out.end());
        if (!t.tip) { for (auto& k : t.kids) walk(k, acc); return; }
        WordPath wp; wp.parent = t.tag; wp.steps.resize(ps.size());
        int at = -1;
        for (auto& a : acc) {
          if (a.kind == Atom::VAR) { if (at >= 0) throw CompileError("internal: register after new output on a path"); continue; }
          if (a.kind == Atom::FUNC) {
            if (a.sym != at + 1) throw CompileError("internal: symbols of a test out of order on a path");
            at = a.sym; wp.steps[(size_t)at].copy = a.func == 0; wp.steps[(size_t)at].tbl = a.tbl;
          } else if (at < 0) { if (!a.bytes.empty()) throw CompileError("internal: output before the first symbol of a test"); }
          else wp.steps[(size_t)at].bytes += a.bytes;
        }
        if (at + 1 != (int)ps.size()) throw CompileError("internal: a path of a test does not read all its symbols");
        edge.path.push_back(std::move(wp));
      };
      walk(*cur, {});
      Tree skel; skeleton(*cur, skel);
      edge.to = intern(skel);
      w.states[sid].edges.push_back(std::move(edge));
    }
  }
  w.tables = f.tables;
  return w;
}

FST leafGraph(const WordSST& w) {
  using Word = std::vector<ByteSet>;
  using Cons = std::set<Word>;   // words the input must NOT start with from here: longer tests that would have fired instead
  FST g;
  g.tables = w.tables;
  auto fresh = [&]() { g.eps.emplace_back(); g.sym.emplace_back(); g.is_final.push_back(0); return g.nstates++; };
  if (w.init < 0 || (size_t)w.init >= w.states.size() || (int)w.init_path.size() != w.states[(size_t)w.init].nleaves)
    throw CompileError("initial state and initial path constants disagree");
  for (auto& st : w.states)
    for (auto& e : st.edges) {
      if (e.to < 0 || (size_t)e.to >= w.states.size() || (int)e.path.size() != w.states[(size_t)e.to].nleaves || e.word.empty())
        throw CompileError("a test's target and its leaf annotation disagree");
      for (auto& p : e.word) if (p.empty()) throw CompileError("empty predicate in a test");
      for (auto& wp : e.path) if (wp.steps.size() != e.word.size() || wp.parent < 0 || wp.parent >= st.nleaves) throw CompileError("a leaf's steps and its test disagree");
    }
  // A node = (state, leaf, constraint).  A block takes the LONGEST of its tests that matches (the nested IfI's of
  // compileTransitions, SSTCompiler.hs:113-127, try the extensions of a word before the word's own action), so the
  // alternative of a shorter test carries the longer ones as a constraint on what follows: with it every input takes
  // exactly the test the word machine takes, and no path of the transducer is represented twice.
  std::map<std::tuple<int, int, Cons>, int> ids;
  std::deque<std::tuple<int, int, Cons>> work;
  auto nodeOf = [&](int q, int l, const Cons& c) {
    auto key = std::make_tuple(q, l, c);
    auto it = ids.find(key);
    if (it != ids.end()) return it->second;
    if (ids.size() > 2000000) throw CompileError("the unrolled lookahead machine has more than 2000000 nodes");
    const int id = fresh();
    ids.emplace(key, id); work.push_back(key);
    return id;
  };
  g.init = fresh();
  for (int l = 0; l < w.states[(size_t)w.init].nleaves; ++l) {
    const int n = nodeOf(w.init, l, {});   // (sequenced: nodeOf grows g.eps)
    g.eps[(size_t)g.init].push_back({w.init_path[(size_t)l], n});
  }

  // A node about to read symbol i of its alternatives reads it ONCE where it can — one symbol edge per distinct predicate
  // (blocks of one partition: equal or disjoint), the alternatives behind it in their order — so that a leaf of the word
  // machine stays one leaf of the table machine; where alternatives do not agree (overlapping predicates, different copy
  // flags: no machine of the reference's does that) they are read apart, in order.
  struct Alt { const WordEdge* e; int lt; Cons c; };
  struct Group { ByteSet pred; bool copy; int tbl; std::vector<Alt> alts; };
  std::function<void(int, const std::vector<Alt>&, size_t, bool)> fill;
  auto after = [&](int m, const std::vector<Alt>& alts, size_t i) {   // m: symbol i has been read; what follows it, in order
    for (size_t k = 0; k < alts.size();) {
      const WordEdge& e = *alts[k].e;
      const std::string& bytes = e.path[(size_t)alts[k].lt].steps[i].bytes;
      if (e.word.size() == i + 1) { const int n = nodeOf(e.to, alts[k].lt, alts[k].c); g.eps[(size_t)m].push_back({bytes, n}); ++k; continue; }
      std::vector<Alt> run;
      while (k < alts.size() && alts[k].e->word.size() > i + 1 && alts[k].e->path[(size_t)alts[k].lt].steps[i].bytes == bytes) run.push_back(alts[k++]);
      const int n = fresh();
      g.eps[(size_t)m].push_back({bytes, n});
      fill(n, run, i + 1, false);
    }
  };
  fill = [&](int node, const std::vector<Alt>& alts, size_t i, bool fin) {
    std::vector<std::vector<Group>> runs(1);
    for (const Alt& a : alts) {
      const ByteSet& p = a.e->word[i];
      const bool c = a.e->path[(size_t)a.lt].steps[i].copy;
      const int tb = c ? a.e->path[(size_t)a.lt].steps[i].tbl : -1;
      Group* hit = nullptr; bool clash = false;
      for (auto& gr : runs.back()) {
        if (gr.pred == p) { if (gr.copy == c && gr.tbl == tb) hit = &gr; else clash = true; }
        else if (!(gr.pred & p).empty()) clash = true;
      }
      if (clash) { runs.emplace_back(); hit = nullptr; }
      if (hit) hit->alts.push_back(a); else runs.back().push_back(Group{p, c, tb, {a}});
    }
    auto reader = [&](int r, const std::vector<Group>& run) {
      for (const Group& gr : run) {
        // the predicate, cut along the first symbols of the constraints of its alternatives
        std::set<ByteSet> firsts;
        for (auto& a : gr.alts) for (auto& f : a.c) firsts.insert(f[0]);
        std::vector<ByteSet> cuts(firsts.begin(), firsts.end()), parts;
        if (cuts.empty()) parts.push_back(gr.pred);
        else {
          std::map<std::vector<bool>, ByteSet> blocks;
          for (int b = 0; b < 256; ++b) {
            if (!gr.pred.has(b)) continue;
            std::vector<bool> sig(cuts.size());
            for (size_t x = 0; x < cuts.size(); ++x) sig[x] = cuts[x].has(b);
            blocks[sig].add(b);
          }
          for (auto& kv : blocks) parts.push_back(kv.second);
        }
        for (const ByteSet& part : parts) {
          std::vector<Alt> live;
          for (auto& a : gr.alts) {
            Cons next; bool dead = false;
            for (auto& f : a.c) {
              if (!part.subsetOf(f[0])) continue;   // the forbidden word does not start like this: it no longer applies
              if (f.size() == 1) { dead = true; break; }
              next.insert(Word(f.begin() + 1, f.end()));
            }
            if (!dead) live.push_back(Alt{a.e, a.lt, std::move(next)});
          }
          if (live.empty()) continue;
          const int m = fresh();
          g.sym[(size_t)r].push_back({part, gr.copy, m, gr.tbl});
          after(m, live, i);
        }
      }
    };
    if (runs.size() == 1) { reader(node, runs[0]); if (fin) g.is_final[(size_t)node] = 1; return; }
    for (auto& run : runs) { const int r = fresh(); g.eps[(size_t)node].push_back({"", r}); reader(r, run); }
    if (fin) { const int f = fresh(); g.is_final[(size_t)f] = 1; g.eps[(size_t)node].push_back({"", f}); }
  };
  while (!work.empty()) {
    auto [q, lp, c] = work.front(); work.pop_front();
    const int node = ids.at(std::make_tuple(q, lp, c));
    const WordState& st = w.states[(size_t)q];
    std::vector<const WordEdge*> order;
    for (auto& e : st.edges) order.push_back(&e);
    std::stable_sort(order.begin(), order.end(), [](const WordEdge* a, const WordEdge* b) { return a->word.size() > b->word.size(); });
    std::vector<Alt> alts;
    for (const WordEdge* e : order) {
      Cons mine = c;
      for (const WordEdge* o : order) {   // the tests that extend this one
        if (o->word.size() <= e->word.size()) break;
        if (std::equal(e->word.begin(), e->word.end(), o->word.begin())) mine.insert(o->word);
      }
      for (size_t lt = 0; lt < e->path.size(); ++lt) if (e->path[lt].parent == lp) alts.push_back(Alt{e, (int)lt, mine});
    }
    fill(node, alts, 0, st.final_leaf == lp);
  }
  return g;
}

// ================================================================== optimize
namespace {
struct AbsVal { bool exact; std::string v; bool operator==(const AbsVal& o) const { return exact == o.exact && (!exact || v == o.v); } };
using AbsValuation = std::map<int, AbsVal>;
}  // namespace

int optimizeSST(SST& s, int level) {  // SymbolicSST.hs:323-331
  if (level <= 0) return 0;
  bool weak = level < 3;
  size_t n = s.states.size();
  std::vector<AbsValuation> gamma(n);
  // liftAbstractValuation (SymbolicSST.hs:210-219): nullopt = a referenced register has no info yet
  auto lift = [](const AbsValuation& rho, const UpdateString& us) -> std::optional<AbsVal> {
    AbsVal acc{true, ""};
    // the reference folds from the right, but a missing register anywhere yields Nothing
    for (auto& a : us) if (a.kind == Atom::VAR && !rho.count(a.var)) return std::nullopt;
    for (auto& a : us) {
      if (a.kind == Atom::VAR) { const AbsVal& x = rho.at(a.var); if (!x.exact) acc.exact = false; else acc.v += x.v; }
      else if (a.kind == Atom::FUNC) { if (a.func == 0) acc.exact = false; }
      else acc.v += a.bytes;
    }
    if (!acc.exact) acc.v.clear();
    return acc;
  };
  // NB: `go` in the reference short-circuits on the first non-constant FuncA (→ Just Ambiguous even
  // if a later register is missing).  Reproduce that order-sensitivity exactly:
  auto liftExact = [&](const AbsValuation& rho, const UpdateString& us) -> std::optional<AbsVal> {
    // evaluate right-to-left recursion of `go`: go (x:xs) combines x with go xs
    std::function<std::optional<AbsVal>(size_t)> go = [&](size_t i) -> std::optional<AbsVal> {
      if (i == us.size()) return AbsVal{true, ""};
      const Atom& a = us[i];
      if (a.kind == Atom::FUNC && a.func == 0) return AbsVal{false, ""};
      if (a.kind == Atom::VAR) {
        auto it = rho.find(a.var);
        if (it == rho.end()) return std::nullopt;
        auto rest = go(i + 1);
        if (!rest) return std::nullopt;
        if (!it->second.exact || !rest->exact) return AbsVal{false, ""};
        return AbsVal{true, it->second.v + rest->v};
      }
      auto rest = go(i + 1);
      if (!rest) return std::nullopt;
      if (!rest->exact) return rest;
      return AbsVal{true, (a.kind == Atom::CONST ? a.bytes : std::string()) + rest->v};
    };
    return go(0);
  };
  (void)lift;
  auto lub = [](const AbsVal& a, const AbsVal& b) { return (a.exact && b.exact && a.v == b.v) ? a : AbsVal{false, ""}; };
  auto lte = [](const AbsVal& a, const AbsVal& b) { return !b.exact || a == b; };

  std::set<int> states;
  for (size_t i = 0; i < n; ++i) states.insert((int)i);
  int iters = 0;
  while (!states.empty()) {  // abstractInterpretation (SymbolicSST.hs:286-299)
    std::map<int, AbsValuation> contrib;
    for (int r : states) {
      const AbsValuation& rho_r = gamma[r];
      for (auto& e : s.states[r].edges) {
        AbsValuation rho2 = rho_r;  // M.union (mapMaybe lift kappa) rho
        for (auto& kv : e.upd) {
          AbsValuation base = rho_r;
          if (weak) base[kv.first] = AbsVal{false, ""};
          auto v = liftExact(base, kv.second);
          if (v) rho2[kv.first] = *v;
        }
        auto it = contrib.find(e.to);
        if (it == contrib.end()) contrib[e.to] = rho2;
        else for (auto& kv : rho2) {  // M.unionWith lub
          auto jt = it->second.find(kv.first);
          if (jt == it->second.end()) it->second[kv.first] = kv.second; else jt->second = lub(jt->second, kv.second);
        }
      }
    }
    std::set<int> next;
    for (auto& [st, rho2] : contrib) {
      AbsValuation& cur = gamma[st];
      bool sub = true;  // isSubmapOfBy lte rho2 cur
      for (auto& kv : rho2) { auto jt = cur.find(kv.first); if (jt == cur.end() || !lte(kv.second, jt->second)) { sub = false; break; } }
      if (sub) continue;
      for (auto& kv : rho2) { auto jt = cur.find(kv.first); if (jt == cur.end()) cur[kv.first] = kv.second; else jt->second = lub(jt->second, kv.second); }
      next.insert(st);
    }
    states.swap(next);
    ++iters;
  }
  // applyAbstractEnvironment (SymbolicSST.hs:301-321)
  auto apply = [](const AbsValuation& rho, const UpdateString& us) {
    UpdateString out;
    for (auto a : us) {
      if (a.kind == Atom::VAR) { auto it = rho.find(a.var); if (it != rho.end() && it->second.exact) a = constAtom(it->second.v); }
      if (a.kind == Atom::CONST) {
        if (a.bytes.empty()) continue;
        if (!out.empty() && out.back().kind == Atom::CONST) { out.back().bytes += a.bytes; continue; }
      }
      out.push_back(a);
    }
    return out;
  };
  for (size_t q = 0; q < n; ++q) {
    for (auto& e : s.states[q].edges) {
      std::map<int, UpdateString> k2;
      for (auto& kv : e.upd) {
        auto it = gamma[e.to].find(kv.first);
        if (it != gamma[e.to].end() && it->second.exact) continue;  // statically known at the destination
        k2[kv.first] = apply(gamma[q], kv.second);
      }
      e.upd = std::move(k2);
    }
    if (s.states[q].is_final) s.states[q].final_upd = apply(gamma[q], s.states[q].final_upd);
  }
  return iters;
}

}  // namespace kexc
