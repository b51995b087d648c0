// lower.cpp — SST → table form → KXP blob.
//
// Register form follows the reference's SST → IL lowering
// (src/KMC/SSTCompiler.hs:67-156): assignments are ordered so that a register
// is read before it is overwritten (orderAssignments, :85-104), `v := v ++ …`
// appends in place, anything else resets first (compileAssignment, :71-80).
// With --la=false every IL block is `NextI 1 1 fallback; [IfI …]; FailI` over
// pairwise-disjoint single-symbol tests, which is exactly a (state, byte
// class) → (next, action) table.
#include <algorithm>
#include <cstring>
#include <functional>

#include "../../../include/kxp_format.h"
#include "kexc.h"

namespace kexc {

namespace {

// SSTCompiler.hs:85-104
std::vector<int> orderAssignments(const std::map<int, UpdateString>& ru) {
  std::set<int> mark;
  std::vector<int> acc;  // built by prepending
  std::function<void(int, std::set<int>)> visit = [&](int v, std::set<int> temp) {
    if (temp.count(v)) throw CompileError("Not a DAG");
    if (mark.count(v) || !ru.count(v)) { mark.insert(v); return; }
    temp.insert(v);
    for (auto& a : ru.at(v)) if (a.kind == Atom::VAR && a.var != v) visit(a.var, temp);
    mark.insert(v);
    acc.insert(acc.begin(), v);
  };
  for (auto& kv : ru) visit(kv.first, {});
  return acc;
}

struct Lowerer {
  StageTables& t;
  std::map<std::string, uint32_t> cmap;
  std::map<std::vector<uint64_t>, uint32_t> amap;
  explicit Lowerer(StageTables& tt) : t(tt) {}

  uint32_t constId(const std::string& s) {
    auto it = cmap.find(s);
    if (it != cmap.end()) return it->second;
    uint32_t id = (uint32_t)t.consts.size(); cmap[s] = id; t.consts.push_back(s); return id;
  }
  void assignment(int var, const UpdateString& us, std::vector<MicroOp>& ops) {  // SSTCompiler.hs:67-80
    size_t i = 0;
    if (!us.empty() && us[0].kind == Atom::VAR && us[0].var == var) i = 1;
    else ops.push_back({KXP_OP_RESET, (uint16_t)var, 0});
    for (; i < us.size(); ++i) {
      const Atom& a = us[i];
      if (a.kind == Atom::VAR) ops.push_back({KXP_OP_CONCAT, (uint16_t)var, (uint32_t)a.var});
      else if (a.kind == Atom::CONST) ops.push_back({KXP_OP_APPEND_CONST, (uint16_t)var, constId(a.bytes)});
      else if (a.func == 0 && a.tbl >= 0) ops.push_back({KXP_OP_APPEND_TBL, (uint16_t)var, (uint32_t)a.tbl});   // AppendTblI (IL.hs:44)
      else if (a.func == 0) ops.push_back({KXP_OP_APPEND_SYM, (uint16_t)var, 0});
    }
  }
  uint32_t actionId(const std::vector<MicroOp>& ops) {
    std::vector<uint64_t> key;
    for (auto& o : ops) key.push_back(((uint64_t)o.op << 56) | ((uint64_t)o.dst << 32) | o.arg);
    auto it = amap.find(key);
    if (it != amap.end()) return it->second;
    uint32_t id = (uint32_t)t.actions.size(); amap[key] = id; t.actions.push_back(ops); return id;
  }
};

}  // namespace

// The synchronising automaton of a stage: subsets of states reachable from "any state" (what k_sync runs from a
// segment start until one state remains).  Needs only delta, so it is also built for programs that arrive as tables
// (kexc_emit_pipeline).
void buildSync(StageTables& t) {
  const size_t n = (size_t)t.nstates;
  t.sync_next.clear(); t.sync_state.clear(); t.sync_complete = true;
  {
    const size_t CAP = 4096;
    std::map<std::vector<uint16_t>, uint32_t> ids;
    std::vector<std::vector<uint16_t>> subs;
    auto intern = [&](const std::vector<uint16_t>& v) -> uint32_t {
      auto it = ids.find(v);
      if (it != ids.end()) return it->second;
      if (subs.size() >= CAP) { t.sync_complete = false; return KXP_SYNC_UNKNOWN; }
      uint32_t id = (uint32_t)subs.size(); ids[v] = id; subs.push_back(v); return id;
    };
    std::vector<uint16_t> all(n);
    for (size_t i = 0; i < n; ++i) all[i] = (uint16_t)i;
    intern(all);
    for (size_t i = 0; i < subs.size(); ++i) {
      std::vector<uint16_t> cur = subs[i];
      t.sync_state.push_back(cur.size() == 1 ? cur[0] : cur.empty() ? KXP_SYNC_EMPTY : KXP_SYNC_MULTI);
      for (int c = 0; c < t.nclasses; ++c) {
        uint32_t nx;
        if (cur.size() <= 1) nx = (uint32_t)i;  // terminal for the engine (singleton / empty)
        else {
          std::vector<uint16_t> v;
          for (uint16_t q : cur) { uint16_t d = t.delta[(size_t)q * t.nclasses + c]; if (d != KXP_NO_STATE) v.push_back(d); }
          std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end());
          nx = intern(v);
        }
        t.sync_next.push_back(nx);
      }
    }
  }
}

StageTables lower(const SST& s, const SST&) {
  StageTables t;
  size_t n = s.states.size();
  t.nstates = (int)n; t.q0 = s.init;
  t.tables = s.tables;
  if (t.tables.size() > 254) throw CompileError("more than 254 symbol tables");
  // global byte classes: coarsest partition refining every predicate of every state
  {
    std::set<ByteSet> preds;
    for (auto& st : s.states) for (auto& e : st.edges) preds.insert(e.pred);
    std::vector<int> cls(256, 0);
    int ncls = 1;
    for (auto& p : preds) {
      std::map<std::pair<int, bool>, int> ren;
      for (int b = 0; b < 256; ++b) {
        auto key = std::make_pair(cls[b], p.has(b));
        auto it = ren.find(key);
        if (it == ren.end()) it = ren.emplace(key, (int)ren.size()).first;
        cls[b] = it->second;
      }
      ncls = (int)ren.size();
    }
    if (ncls > 256) throw CompileError("internal: more than 256 byte classes");
    t.nclasses = ncls;
    for (int b = 0; b < 256; ++b) t.cls[b] = (uint8_t)cls[b];
  }
  std::vector<int> rep(t.nclasses, -1);
  for (int b = 255; b >= 0; --b) rep[t.cls[b]] = b;

  // registers actually mentioned (after optimize some vanish); renumber densely, root stays 0
  std::map<int, int> regmap; regmap[0] = 0;
  auto noteReg = [&](int v) { if (!regmap.count(v)) { int id = (int)regmap.size(); regmap[v] = id; } };
  for (auto& st : s.states) {
    for (auto& e : st.edges) for (auto& kv : e.upd) { noteReg(kv.first); for (auto& a : kv.second) if (a.kind == Atom::VAR) noteReg(a.var); }
    for (auto& a : st.final_upd) if (a.kind == Atom::VAR) noteReg(a.var);
  }
  t.nregs = (int)regmap.size();
  if (t.nregs > 65535) throw CompileError("too many registers");
  auto renUS = [&](const UpdateString& us) { UpdateString o = us; for (auto& a : o) if (a.kind == Atom::VAR) a.var = regmap[a.var]; return o; };

  Lowerer L(t);
  t.delta.assign(n * t.nclasses, KXP_NO_STATE);
  t.act.assign(n * t.nclasses, 0);
  t.final_act.assign(n, KXP_NOT_FINAL);
  t.pback.assign(n * t.nclasses, 0);
  t.nleaves.assign(n, 0); t.fin_leaf.assign(n, KXP_NO_LEAF);
  t.maxleaves = 1;
  for (auto& st : s.states) t.maxleaves = std::max(t.maxleaves, st.nleaves);
  if (t.maxleaves > 254) throw CompileError("more than 254 simultaneous paths in one state");

  std::map<std::string, uint32_t> pcmap;
  auto pconstId = [&](const std::string& b) {
    auto it = pcmap.find(b);
    if (it != pcmap.end()) return it->second;
    uint32_t id = (uint32_t)t.pconsts.size(); pcmap[b] = id; t.pconsts.push_back(b); return id;
  };
  pconstId("");
  std::map<std::vector<uint32_t>, uint32_t> bmap;

  for (size_t q = 0; q < n; ++q) {
    const SSTState& st = s.states[q];
    t.nleaves[q] = (uint8_t)st.nleaves;
    if (st.is_final) {
      std::vector<MicroOp> ops;
      L.assignment(0, renUS(st.final_upd), ops);
      for (auto& o : ops) if (o.op == KXP_OP_RESET && o.dst == 0) throw CompileError("internal: final update resets the stream register");
      t.final_act[q] = L.actionId(ops);
      t.fin_leaf[q] = (uint8_t)st.final_leaf;
    }
    for (auto& e : st.edges) {
      std::map<int, UpdateString> ru;
      for (auto& kv : e.upd) ru[regmap[kv.first]] = renUS(kv.second);
      std::vector<MicroOp> ops;
      for (int v : orderAssignments(ru)) L.assignment(v, ru[v], ops);
      for (auto& o : ops) if (o.op == KXP_OP_RESET && o.dst == 0) throw CompileError("internal: transition resets the stream register");
      uint32_t aid = L.actionId(ops);
      std::vector<uint32_t> row(t.maxleaves, KXP_DEAD_LEAF);
      if ((int)e.path.size() != s.states[e.to].nleaves) throw CompileError("internal: path form leaf count mismatch");
      for (size_t j = 0; j < e.path.size(); ++j) {
        uint32_t pc = pconstId(e.path[j].bytes);
        if (pc >= (t.tables.empty() ? 1u << 23 : 1u << 15)) throw CompileError("too many path constants");
        const int tb = e.path[j].copy ? e.path[j].tbl : -1;
        if (tb >= (int)t.tables.size()) throw CompileError("internal: symbol table out of range");
        row[j] = (uint32_t)e.path[j].parent | (e.path[j].copy ? 0x100u : 0u) | (pc << 9) | ((uint32_t)(tb + 1) << 24);
      }
      auto it = bmap.find(row);
      if (it == bmap.end()) { it = bmap.emplace(row, (uint32_t)bmap.size()).first; t.back.insert(t.back.end(), row.begin(), row.end()); }
      for (int c = 0; c < t.nclasses; ++c) {
        if (!e.pred.has(rep[c])) continue;
        size_t ix = q * t.nclasses + c;
        if (t.delta[ix] != KXP_NO_STATE) throw CompileError("internal: overlapping tests in one state");
        t.delta[ix] = (uint16_t)e.to; t.act[ix] = aid; t.pback[ix] = it->second;
      }
    }
  }
  t.init_const.assign(t.maxleaves, 0);
  for (size_t j = 0; j < s.init_path.size(); ++j) t.init_const[j] = pconstId(s.init_path[j]);

  buildSync(t);
  return t;
}

// ------------------------------------------------------------------- blob
namespace {
struct W {
  std::vector<uint8_t> b;
  void u32(uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back((v >> (8 * i)) & 0xFF); }
  void u16(uint16_t v) { b.push_back(v & 0xFF); b.push_back(v >> 8); }
  void raw(const void* p, size_t n) { const uint8_t* c = (const uint8_t*)p; b.insert(b.end(), c, c + n); }
  void pad() { while (b.size() % 4) b.push_back(0); }
};
void pool(W& w, const std::vector<std::string>& cs) {
  uint32_t off = 0;
  for (auto& c : cs) { w.u32(off); off += (uint32_t)c.size(); }
  w.u32(off);
  for (auto& c : cs) w.raw(c.data(), c.size());
  w.pad();
}
}  // namespace

std::vector<uint8_t> writeBlob(const std::vector<StageTables>& stages, const std::string& info) {
  W w;
  w.raw(KXP_MAGIC, 8);
  w.u32(KXP_VERSION); w.u32((uint32_t)stages.size()); w.u32((uint32_t)info.size());
  w.raw(info.data(), info.size()); w.pad();
  for (auto& t : stages) {
    size_t nops = 0, cpl = 0, pcpl = 0;
    for (auto& a : t.actions) nops += a.size();
    for (auto& c : t.consts) cpl += c.size();
    for (auto& c : t.pconsts) pcpl += c.size();
    uint32_t nback = (uint32_t)(t.back.size() / t.maxleaves);
    uint32_t nsync = (uint32_t)t.sync_state.size();
    w.u32(KXP_STAGE_MAGIC); w.u32(t.nstates); w.u32(t.nclasses); w.u32(t.q0); w.u32(t.nregs);
    w.u32((uint32_t)t.actions.size()); w.u32((uint32_t)nops); w.u32((uint32_t)t.consts.size()); w.u32((uint32_t)cpl);
    w.u32(t.maxleaves); w.u32(nback); w.u32((uint32_t)t.pconsts.size()); w.u32((uint32_t)pcpl);
    w.u32(nsync); w.u32(t.sync_complete ? 1 : 0); w.u32((t.act_regs >= 0 ? 1u | ((uint32_t)t.act_regs << 8) : 0u) | (t.tables.empty() ? 0u : KXP_STAGE_HAS_TABLES));
    w.raw(t.cls, 256);
    for (auto v : t.delta) w.u16(v);
    w.pad();
    for (auto v : t.act) w.u32(v);
    for (auto v : t.final_act) w.u32(v);
    uint32_t off = 0;
    for (auto& a : t.actions) { w.u32(off); off += (uint32_t)a.size(); }
    w.u32(off);
    for (auto& a : t.actions) for (auto& o : a) { w.u32(((uint32_t)o.op << 24) | o.dst); w.u32(o.arg); }
    pool(w, t.consts);
    for (auto v : t.pback) w.u32(v);
    w.raw(t.nleaves.data(), t.nleaves.size()); w.pad();
    w.raw(t.fin_leaf.data(), t.fin_leaf.size()); w.pad();
    for (auto v : t.back) w.u32(v);
    pool(w, t.pconsts);
    for (auto v : t.init_const) w.u32(v);
    for (auto v : t.sync_next) w.u32(v);
    for (auto v : t.sync_state) w.u32(v);
    if (!t.tables.empty()) { w.u32((uint32_t)t.tables.size()); for (auto& tb : t.tables) w.raw(tb.data(), 256); }
  }
  return w.b;
}

}  // namespace kexc
