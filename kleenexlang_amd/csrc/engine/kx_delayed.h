// kx_delayed.h — the DELAYED FORM of a path-form stage (round 5; host side, no device code).
//
// The path form (include/kxp_format.h) attributes to input step t the bytes δ(q_t, c_t, ℓ_{t+1}) that the surviving path
// appends there; ℓ_{t+1} is only known from the FUTURE, which is why the general engine runs a backward pass (k_backlen) and
// a backward sweep (k_emit).  For most programs that dependence is shallow: two leaves of a state differ only in what a
// LATER symbol decides (csv2json: "does the field end here" = is the next byte a comma), or not at all as far as the output
// goes (apache_log's two live alternatives "another record follows / this is the last record" append the same bytes for a
// whole line).  Measured on the BASELINE inputs (profiles/uniform_stats.py): the output of EVERY step of apache_log,
// csv2json and iso_datetime_to_json is determined by at most two further input symbols; thousand_sep (the digit count of a
// whole number decides where the commas go) is the counter-example.
//
// The delayed form makes that a table.  It is the deterministic FORWARD transducer with fixed delay K:
//     state  (q, g_1 … g_K)   q = SST state; g_j = what step s-K+j-1 appends, as a function of the leaf of q that survives
//                             (a constant function is stored as its value)
//     on a symbol of class c (transition row r of the path form, target q', parent map p_r : leaves(q') → leaves(q)):
//         g_j' = g_j ∘ p_r,  g_new = (leaf' ↦ δ(r, leaf'));   g_1' must now be CONSTANT — its value is what this step emits —
//         and the next state is (q', g_2' … g_K', g_new).
// So step s writes the bytes the path form attributes to step s-K (the copied byte is input byte s-K), no leaf is ever
// needed, and output offsets are a forward prefix sum: one forward pass for lengths (k_dforward), one fused walk that places
// the bytes (k_demit) — no backward pass, no re-derivation, no second sweep.  Where g_1' is NOT constant the context needs
// more than K symbols of lookahead: the transition goes to the absorbing ESCAPE state, the run notices (Flags::df_esc) and the
// shard is redone by the general engine — exactness never rests on the delayed form being applicable.
// At the end of a shard the K pending functions are evaluated at the shard's end leaf (the same hand-off value the general
// engine uses; at end of input the final state's leaf) and written behind the kernel's output by the host ("tail").
//
// MERGED CONSTANTS (round 6).  A constant costs k_demit a job (a slot, a lane of its constants phase) however short it is, and the
// path form cuts a program's constants where its STEPS fall: apache_log writes `"` `,` `"date":"` on three steps of which none
// copies a byte in between — 15 constants per line where 8 would do.  A constant may wait as long as no copied byte can come
// between it and the next one: the product state carries a queue `def` of constants that are due, each with its age; a step
// whose successor cannot copy (no kind in the new oldest pending slot has the copy bit) keeps the queue, any other step writes
// all of it in front of its own constant, and a constant that has waited J steps is written whatever follows.  Each constant's
// step is thus a function of the J steps behind it alone: a lane that starts K + J symbols before its own part is in the same
// product state as the run that came from the start of the input — the argument that lets a segment, a window or a shard begin
// in (q, nothing pending, nothing due).  A shard's tail writes what is still due in front of the pending steps.
// The reference has no counterpart: its SST parks undecided output in registers (SymbolicSST.hs:400-446, crt.c append/concat).
#ifndef KX_DELAYED_H
#define KX_DELAYED_H
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace kxdf {

constexpr uint32_t DF_OFF_ROWS = 256;     // the image starts with the 256-byte class table (class * 8); state rows follow
constexpr uint32_t DF_MAX_K = 4;

// what a step appends: copy | canonical path-constant id << 1  (constants compared by CONTENT)
using Kind = uint32_t;
using Slot = std::vector<Kind>;           // size 1: resolved value; else one kind per leaf of the state

struct DfState { uint32_t q; std::vector<Slot> pend; std::vector<std::pair<uint32_t, uint32_t>> def; };   // def: (canonical path-constant id, age) still due, oldest first

struct DfBuild {
  uint32_t K = 0, C = 0, nP = 0;
  uint32_t J = 0;                          // merge window: steps a constant may wait for the next one (0: every constant is written on its own step)
  std::vector<std::string> outs;           // what a step writes behind its copied byte (merged constants included); outs[0] = nothing
  bool wide = false;                       // more than 31 byte classes: the class table holds class indices (else class * 8)
  std::vector<uint32_t> img;               // LDS image: [class*8 u8[256] | rows (nP + 2) x C x {lo, hi} | pool]
  // pool: every path constant FOUR times — copy s = the constant from its byte s on, zero-padded to plen = (length + 15) & ~15
  // bytes, the copies one behind the other from a 16-byte boundary — so that k_demit can store the part of a constant behind the
  // next 4-byte boundary of its destination as ALIGNED DWORDS fetched with one aligned 16-byte read (copy s, s = bytes up to
  // that boundary); an entry names copy 0
  std::vector<uint32_t> pool_off;          // [outs.size()] offset of copy 0 inside the pool
  uint32_t off_pool = 0, deadh = 0, esch = 0, starth = 0;
  std::vector<uint16_t> start_of_state;    // [nstates] handle of (q, nothing pending); 0xFFFF = not in the table
  std::vector<DfState> states;             // host copy: original state and pending functions of every product state
  uint32_t ntrans = 0, nesc = 0;           // over the whole table
  uint32_t ntrans_start = 0, nesc_start = 0;   // over the part reachable from the program's start state alone
  uint32_t capped = 0;                     // transitions into states beyond the size budget (they escape)
  uint32_t handleOf(uint32_t idx) const { return DF_OFF_ROWS + idx * C * 8; }
  uint32_t indexOf(uint32_t handle) const { return (handle - DF_OFF_ROWS) / (C * 8); }
};

struct DfInput {   // views into a parsed stage (kx_engine.hip: parseStage)
  uint32_t nstates, C, q0, Lm, nback, npc;
  const uint8_t* cls; const uint16_t* delta; const uint32_t* pback; const uint8_t* nleaves; const uint32_t* back;
  const uint32_t* pcoff; const uint8_t* pcpool; const uint32_t* init_const;
  const uint32_t* apoff;     // [npc] byte offset of every path constant in the engine's 16-byte-aligned pool
  const uint8_t* apool; uint32_t apool_bytes;
  bool has_tbl;
};

// Returns "" and fills `out`, or the reason why the stage has no delayed form.
inline std::string buildDelayed(const DfInput& in, uint32_t K, uint32_t J, size_t image_budget, DfBuild& out) {
  const uint32_t C = in.C, Lm = in.Lm;
  if (K == 0 || K > DF_MAX_K) return "delay out of range";
  if (in.has_tbl) return "symbol tables";
  if (C > 255) return "more than 255 byte classes";
  out.wide = C > 31;   // class * 8 no longer fits the class table's byte: it holds the class index and the kernels shift
  if ((in.apool_bytes >> 4) >= (1u << 13)) return "constant pool too large";
  auto pcOf = [&](uint32_t e) { return e >> 9; };
  // canonical constant ids (by content), lengths
  std::vector<uint32_t> canon(in.npc), clen(in.npc);
  {
    std::map<std::string, uint32_t> seen;
    for (uint32_t pc = 0; pc < in.npc; ++pc) {
      clen[pc] = in.pcoff[pc + 1] - in.pcoff[pc];
      std::string s((const char*)in.pcpool + in.pcoff[pc], clen[pc]);
      auto it = seen.find(s);
      if (it == seen.end()) { seen.emplace(s, pc); canon[pc] = pc; } else canon[pc] = it->second;
    }
  }
  uint32_t empty_pc = in.npc;
  for (uint32_t pc = 0; pc < in.npc; ++pc) if (clen[pc] == 0) { empty_pc = canon[pc]; break; }
  for (uint32_t pc = 0; pc < in.npc; ++pc) if (clen[pc] + 1 > 126) return "a path constant longer than 125 bytes";
  // the kind "nothing appended": needs an empty constant id; programs without one get a virtual id npc
  const Kind NOTHING = (empty_pc << 1);
  auto kindOf = [&](uint32_t e) -> Kind { return ((e >> 8) & 1u) | (canon[pcOf(e)] << 1); };
  auto normalise = [](Slot& s) { bool same = true; for (size_t i = 1; i < s.size(); ++i) same = same && s[i] == s[0]; if (same && s.size() > 1) s.resize(1); };
  auto keyOf = [](const DfState& s) {
    std::vector<uint32_t> k; k.push_back(s.q);
    for (auto& sl : s.pend) { k.push_back((uint32_t)sl.size()); k.insert(k.end(), sl.begin(), sl.end()); }
    k.push_back(0xFFFFFFFEu);
    for (auto& d : s.def) { k.push_back(d.first); k.push_back(d.second); }
    return k;
  };
  // what a step writes behind its copied byte, interned by content
  out.outs.assign(1, std::string());
  std::map<std::string, uint32_t> out_ids; out_ids.emplace(std::string(), 0u);
  auto outId = [&](const std::string& t) { auto it = out_ids.find(t); if (it != out_ids.end()) return it->second; const uint32_t id = (uint32_t)out.outs.size(); out_ids.emplace(t, id); out.outs.push_back(t); return id; };
  auto bytesOf = [&](uint32_t pc) { return pc < in.npc ? std::string((const char*)in.pcpool + in.pcoff[pc], clen[pc]) : std::string(); };
  bool too_long = false;
  const size_t row_bytes = (size_t)C * 8;
  size_t maxP = (65536 - DF_OFF_ROWS) / row_bytes;
  const size_t pool_est = 4 * (size_t)in.apool_bytes + 64 + (J ? 2048 : 0);   // (four copies of every constant; merged ones on top)
  if (image_budget > DF_OFF_ROWS + pool_est + 3 * row_bytes) {
    const size_t byb = (image_budget - DF_OFF_ROWS - pool_est) / row_bytes;
    if (byb < maxP) maxP = byb;
  } else return "no room for the table";
  if (maxP < 4) return "no room for the table";
  maxP -= 2;   // dead and escape rows
  std::map<std::vector<uint32_t>, uint32_t> ids;
  std::vector<DfState>& S = out.states; S.clear();
  std::vector<uint32_t> todo;
  auto add = [&](DfState&& st) -> int64_t {
    auto k = keyOf(st);
    auto it = ids.find(k);
    if (it != ids.end()) return it->second;
    if (S.size() >= maxP) return -1;
    const uint32_t id = (uint32_t)S.size();
    ids.emplace(std::move(k), id); S.push_back(std::move(st)); todo.push_back(id);
    return id;
  };
  struct Tr { int64_t next; uint32_t copy, out; };   // next: state index, -1 = no transition (dead), -2 = escape; what the step writes: the copied byte?, outs[out]
  std::vector<std::vector<Tr>> trans;
  auto expand = [&]() {
    while (!todo.empty()) {
      const uint32_t id = todo.back(); todo.pop_back();
      if (trans.size() <= id) trans.resize(id + 1);
      std::vector<Tr> row(C, Tr{-1, 0, 0});
      for (uint32_t c = 0; c < C; ++c) {
        const DfState cur = S[id];   // (copy: S may grow)
        const uint16_t t = in.delta[(size_t)cur.q * C + c];
        if (t == 0xFFFFu) continue;
        const uint32_t r = in.pback[(size_t)cur.q * C + c], nl = in.nleaves[t];
        DfState nx; nx.q = t;
        std::vector<Slot> np;
        for (const Slot& g : cur.pend) {
          if (g.size() == 1) { np.push_back(g); continue; }
          Slot h(nl);
          bool ok = true;
          for (uint32_t l = 0; l < nl; ++l) {
            const uint32_t e = in.back[(size_t)r * Lm + l];
            if (e == 0xFFFFFFFFu) { h[l] = 0xFFFFFFFFu; continue; }   // a dead leaf of the target: never the survivor
            const uint32_t p = e & 0xFFu;
            if (p >= g.size()) { ok = false; break; }
            h[l] = g[p];
          }
          if (!ok) { np.clear(); break; }
          // dead leaves take the value of any live one (they cannot be the end leaf)
          Kind live = 0xFFFFFFFFu; for (Kind k : h) if (k != 0xFFFFFFFFu) { live = k; break; }
          if (live == 0xFFFFFFFFu) live = NOTHING;   // (every leaf of the target dead: nothing survives to ask, as for g_new below)
          for (Kind& k : h) if (k == 0xFFFFFFFFu) k = live;
          normalise(h); np.push_back(std::move(h));
        }
        if (np.size() != cur.pend.size()) { row[c] = Tr{-2, 0, 0}; ++out.nesc; continue; }
        {
          Slot h(nl);
          Kind live = 0xFFFFFFFFu;
          for (uint32_t l = 0; l < nl; ++l) { const uint32_t e = in.back[(size_t)r * Lm + l]; h[l] = e == 0xFFFFFFFFu ? 0xFFFFFFFFu : kindOf(e); if (h[l] != 0xFFFFFFFFu && live == 0xFFFFFFFFu) live = h[l]; }
          if (live == 0xFFFFFFFFu) live = NOTHING;
          for (Kind& k : h) if (k == 0xFFFFFFFFu) k = live;
          normalise(h); np.push_back(std::move(h));
        }
        if (np[0].size() != 1) { row[c] = Tr{-2, 0, 0}; ++out.nesc; continue; }   // K symbols do not decide step s-K
        const Kind emit = np[0][0];
        nx.pend.assign(np.begin() + 1, np.end());
        // the constants due: those that waited, then this step's own.  (A state with constants due never copies: they were kept
        // because no kind of the slot that is resolved NOW had the copy bit, and composing with a parent map only selects among them.)
        std::vector<std::pair<uint32_t, uint32_t>> due = cur.def;
        if ((emit >> 1) < in.npc && clen[emit >> 1]) due.emplace_back(canon[emit >> 1], 0u);
        bool next_may_copy = J == 0;
        for (Kind k : nx.pend[0]) next_may_copy = next_may_copy || (k & 1u);
        std::string text;
        size_t nout = due.size();
        if (!next_may_copy) { nout = 0; while (nout < due.size() && due[nout].second >= J) ++nout; }
        for (size_t i = 0; i < nout; ++i) text += bytesOf(due[i].first);
        for (size_t i = nout; i < due.size(); ++i) nx.def.emplace_back(due[i].first, due[i].second + 1);
        if (text.size() + 1 > 126) { too_long = true; text.resize(125); }
        const uint32_t oid = outId(text);
        const int64_t ni = add(std::move(nx));
        if (ni < 0) { row[c] = Tr{-2, 0, 0}; ++out.nesc; ++out.capped; continue; }
        row[c] = Tr{ni, emit & 1u, oid}; ++out.ntrans;
      }
      trans[id] = std::move(row);
    }
  };
  // the program's own start: nothing pending but the initial closure's output (a virtual step -1 that copies nothing)
  {
    DfState st; st.q = in.q0;
    for (uint32_t j = 0; j + 1 < K; ++j) st.pend.push_back(Slot{NOTHING});
    const uint32_t nl = in.nleaves[in.q0];
    Slot g(nl);
    for (uint32_t l = 0; l < nl; ++l) g[l] = canon[in.init_const[l]] << 1;
    normalise(g); st.pend.push_back(std::move(g));
    add(std::move(st));
  }
  expand();
  out.ntrans_start = out.ntrans; out.nesc_start = out.nesc;
  // (q, nothing pending) for every SST state the start-reachable part visits: where a segment, a window or a shard may begin
  out.start_of_state.assign(in.nstates, 0xFFFFu);
  {
    std::vector<uint32_t> qs;
    std::vector<uint8_t> have(in.nstates, 0);
    for (const DfState& s : S) if (!have[s.q]) { have[s.q] = 1; qs.push_back(s.q); }
    for (uint32_t q : qs) {
      DfState st; st.q = q;
      for (uint32_t j = 0; j < K; ++j) st.pend.push_back(Slot{NOTHING});
      const int64_t id = add(std::move(st));
      if (id >= 0) out.start_of_state[q] = 1;   // (handle filled in below)
    }
    expand();
  }
  if (too_long) return "a merged constant longer than 125 bytes";
  const uint32_t nP = (uint32_t)S.size();
  out.K = K; out.C = C; out.nP = nP; out.J = J;
  out.deadh = out.handleOf(nP); out.esch = out.handleOf(nP + 1); out.starth = out.handleOf(0);
  for (uint32_t q = 0; q < in.nstates; ++q) if (out.start_of_state[q] != 0xFFFFu) {
    DfState st; st.q = q; for (uint32_t j = 0; j < K; ++j) st.pend.push_back(Slot{NOTHING});
    out.start_of_state[q] = (uint16_t)out.handleOf(ids.at(keyOf(st)));
  }
  // image
  const size_t rows_end = DF_OFF_ROWS + (size_t)(nP + 2) * row_bytes;
  const size_t off_pool = (rows_end + 15) & ~(size_t)15;
  std::vector<uint8_t> pool;
  out.pool_off.assign(out.outs.size(), 0);
  for (size_t oi = 1; oi < out.outs.size(); ++oi) {
    const std::string& t = out.outs[oi];
    const uint32_t L = (uint32_t)t.size(), plen = (L + 15) & ~15u;
    out.pool_off[oi] = (uint32_t)pool.size();
    for (uint32_t sft = 0; sft < 4; ++sft) {
      const size_t at = pool.size();
      pool.resize(at + plen, 0);
      if (sft < L) memcpy(&pool[at], t.data() + sft, L - sft);
    }
  }
  pool.resize(pool.size() + 32, 0);   // (reads of 16 bytes behind the last copy stay inside)
  if ((pool.size() >> 4) >= (1u << 13)) return "constant pool too large";
  if (off_pool + pool.size() > 65536 + 32768) return "image too large";
  out.off_pool = (uint32_t)off_pool;
  out.img.assign((off_pool + pool.size() + 3) / 4, 0u);
  uint8_t* ib = (uint8_t*)out.img.data();
  for (int b = 0; b < 256; ++b) ib[b] = (uint8_t)(out.wide ? in.cls[b] : in.cls[b] * 8);
  auto entryHi = [&](uint32_t copy, uint32_t oid) -> uint32_t {
    const uint32_t cl = (uint32_t)out.outs[oid].size();
    uint32_t e = (copy ? 0u : 1u) | ((cl + copy) << 24);
    if (cl) e |= ((out.pool_off[oid] >> 4) << 10) | (1u << 23);
    return e;
  };
  uint32_t* rows = out.img.data() + DF_OFF_ROWS / 4;
  for (uint32_t i = 0; i < nP + 2; ++i)
    for (uint32_t c = 0; c < C; ++c) {
      uint32_t lo, hi = 1u;   // (nothing copied, nothing appended)
      if (i >= nP) lo = out.handleOf(i);
      else {
        const Tr& t = trans[i][c];
        if (t.next == -1) lo = out.deadh; else if (t.next == -2) lo = out.esch; else { lo = out.handleOf((uint32_t)t.next); hi = entryHi(t.copy, t.out); }
      }
      // the upper half of lo repeats what the measuring pass adds up in ONE add: bytes appended << 8 | 4 x "a constant follows"
      lo |= (hi & 0xFF000000u) | (((hi >> 23) & 1u) << 18);
      rows[((size_t)i * C + c) * 2] = lo; rows[((size_t)i * C + c) * 2 + 1] = hi;
    }
  memcpy(ib + off_pool, pool.data(), pool.size());
  return "";
}

}  // namespace kxdf
#endif
