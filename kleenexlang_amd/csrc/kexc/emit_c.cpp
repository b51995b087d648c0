// emit_c.cpp — `--backend=c`: print the register form as C in the shape the
// reference generates, for the CPU baseline.
//
// Shape follows src/KMC/Program/Backends/C.hs: program template :41-66, match
// template :72-83, labels `l<phase>_<block>` :172-173, stream-buffer writes
// printed as outputconst/outputarray/output and others as
// append/appendarray/concat/reset :241-311, tests as range/equality
// expressions (SSTCompiler/Classes.hs:56-83), constants as
// `const buffer_unit_t const_<phase>_<id>[]` :439-460, `init_buffer` for every
// non-stream register :480-484.  Unlike the reference, which splices the text
// of crt/crt.c, the output says `#include "crt.c"`: the runtime is chosen by
// the include path at build time (the reference's own crt/crt.c for
// oracle/_ref, the restated runtime otherwise).
#include <algorithm>
#include <map>
#include <sstream>

#include "../../../include/kxp_format.h"
#include "kexc.h"

namespace kexc {

namespace {
std::string cchar(int n) {  // C.hs:397-401
  if (n >= 32 && n < 127 && n != '\\' && n != '\'') return std::string("'") + char(n) + "'";
  return std::to_string(n);
}
std::string predExpr(const ByteSet& p) {
  std::ostringstream o;
  bool first = true;
  for (int b = 0; b < 256;) {
    if (!p.has(b)) { ++b; continue; }
    int e = b;
    while (e + 1 < 256 && p.has(e + 1)) ++e;
    if (!first) o << " || ";
    first = false;
    if (e == b) o << "(next[0] == " << cchar(b) << ")";
    else o << "((" << cchar(b) << " <= next[0]) && (next[0] <= " << cchar(e) << "))";
    b = e + 1;
  }
  return first ? "0" : o.str();
}
}  // namespace

std::string emitC(const std::vector<StageTables>& stages, const std::string& info) {
  for (auto& t : stages) if (t.act_regs >= 0) throw CompileError("--backend=c prints direct-mode programs only (this one has register actions)");
  std::ostringstream o;
  o << "\n#define NUM_PHASES " << stages.size() << "\n#define BUFFER_UNIT_T uint8_t\n#include \"crt.c\"\n";
  {   // prettyTableDecl (C.hs:412-430): one array per phase that has tables
    bool any = false;
    for (size_t ph = 0; ph < stages.size(); ++ph) {
      auto& t = stages[ph];
      if (t.tables.empty()) continue;
      any = true;
      o << "const uint8_t tbl" << ph + 1 << "[" << t.tables.size() << "][256] =\n{";
      for (size_t k = 0; k < t.tables.size(); ++k) {
        o << (k ? ",\n{" : "{");
        for (int b = 0; b < 256; ++b) { char buf[8]; snprintf(buf, sizeof buf, "0x%x", t.tables[k][(size_t)b]); o << (b ? "," : "") << buf; }
        o << "}";
      }
      o << "};\n";
    }
    if (!any) o << "/* no tables */\n";
  }
  int maxregs = 0;
  for (auto& t : stages) maxregs = std::max(maxregs, t.nregs);
  for (int r = 0; r < maxregs; ++r) o << "buffer_t buf_" << r << ";\n";
  for (size_t ph = 0; ph < stages.size(); ++ph) {
    auto& t = stages[ph];
    for (size_t c = 0; c < t.consts.size(); ++c) {
      o << "const buffer_unit_t const_" << ph + 1 << "_" << c << "[" << t.consts[c].size() << "] = {";
      for (size_t i = 0; i < t.consts[c].size(); ++i) {
        char buf[8]; snprintf(buf, sizeof buf, "0x%x", (unsigned char)t.consts[c][i]);
        o << (i ? "," : "") << buf;
      }
      o << "};\n";
    }
  }
  std::string esc;
  for (char ch : info) { if (ch == '"') esc += "\\\""; else if (ch == '%') esc += "%%"; else esc += ch; }
  o << "void printCompilationInfo()\n{\n  fprintf(stdout, \"" << esc << "\\n\");\n}\n\nvoid init()\n{\n";
  for (int r = 1; r < maxregs; ++r) o << "init_buffer(&buf_" << r << ");\n";
  o << "}\n";
  for (size_t ph = 0; ph < stages.size(); ++ph) {
    auto& t = stages[ph];
    int P = (int)ph + 1;
    auto ops = [&](uint32_t aid) {
      std::ostringstream b;
      for (auto& m : t.actions[aid]) {
        bool stream = m.dst == 0;
        switch (m.op) {
          case KXP_OP_RESET: b << "reset(&buf_" << m.dst << ");\n"; break;
          case KXP_OP_APPEND_CONST: {
            size_t bits = t.consts[m.arg].size() * 8;
            if (stream) b << "outputarray(const_" << P << "_" << m.arg << "," << bits << ");\n";
            else b << "appendarray(&buf_" << m.dst << ",const_" << P << "_" << m.arg << "," << bits << ");\n";
            break;
          }
          case KXP_OP_APPEND_SYM:
            if (stream) b << "outputconst(next[0],8);\n"; else b << "append(&buf_" << m.dst << ",next[0],8);\n";
            break;
          case KXP_OP_APPEND_TBL:   // prettyAppendTbl (C.hs:228-252)
            if (stream) b << "outputconst(tbl" << P << "[" << m.arg << "][next[0]],8);\n";
            else b << "append(&buf_" << m.dst << ",tbl" << P << "[" << m.arg << "][next[0]],8);\n";
            break;
          case KXP_OP_CONCAT:
            if (stream) b << "output(&buf_" << m.arg << ");\n"; else b << "concat(&buf_" << m.dst << ",&buf_" << m.arg << ");\n";
            break;
        }
      }
      return b.str();
    };
    o << "void match" << P << "()\n{\n  int i = 0;\ngoto l" << P << "_" << t.q0 << ";\n";
    for (int q = 0; q < t.nstates; ++q) {
      o << "l" << P << "_" << q << ": if (!readnext(1, 1))\n{\n";
      if (t.final_act[q] != KXP_NOT_FINAL) o << ops(t.final_act[q]) << "   goto accept" << P << ";\n";
      else o << "   goto fail" << P << ";\n";
      o << "}\n";
      // one test per distinct (target, action): the reference has one per predicate of the state
      std::map<std::pair<uint32_t, uint32_t>, ByteSet> groups;
      for (int b = 0; b < 256; ++b) {
        size_t ix = (size_t)q * t.nclasses + t.cls[b];
        if (t.delta[ix] == KXP_NO_STATE) continue;
        groups[{t.delta[ix], t.act[ix]}].add(b);
      }
      std::vector<std::pair<ByteSet, std::pair<uint32_t, uint32_t>>> tests;
      for (auto& g : groups) tests.push_back({g.second, g.first});
      std::sort(tests.begin(), tests.end(), [](auto& a, auto& b) { return a.first.first() < b.first.first(); });
      for (auto& ts : tests) {
        o << "if (((avail >= 1) && (" << predExpr(ts.first) << ")))\n{\n" << ops(ts.second.second)
          << "   consume(1);\n   goto l" << P << "_" << ts.second.first << ";\n}\n";
      }
      o << "goto fail" << P << ";\n";
    }
    o << "  accept" << P << ":\n    return;\n  fail" << P
      << ":\n    fprintf(stderr, \"Match error at input symbol %zu!\\n\", count);\n    exit(1);\n}\n";
  }
  o << "\nvoid match(int phase)\n{\n  switch(phase) {\n";
  for (size_t ph = 0; ph < stages.size(); ++ph) o << "    case " << ph + 1 << ": match" << ph + 1 << "(); break;\n";
  o << "    default:\n      fprintf(stderr, \"Invalid phase: %d given\\n\", phase);\n      exit(1);\n  }\n}\n";
  return o.str();
}

}  // namespace kexc
