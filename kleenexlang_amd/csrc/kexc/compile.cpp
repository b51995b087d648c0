// compile.cpp — pipeline driver + C-ABI of the compiler library (libkexc.so).
//
// Mirrors the direct-mode compile path of the reference:
//   createProgram → buildTransducers → generateDirectSSTs → compileDirect
//   (src/KMC/Frontend/Commands.hs:50-68,82-115,155-180,182-210).
#include <dlfcn.h>
#include <sys/stat.h>

#include <cstring>
#include <fstream>
#include <sstream>

#include "../../../include/kexc_api.h"
#include "kexc.h"

namespace kexc {

Compiled compileSource(const std::string& src, const std::string& srcname, const Options& o) {
  Compiled out;
  if (o.regex) {
    // createProgram (RegexFlavor) → buildTransducers → generateOracleSSTs → compileCoder (Commands.hs:69-79,117-136,246-275;
    // kexc.hs:46-48): the program's output is the code of the greedy parse, not a rewriting of the input
    RProg rp = parseRegexProgram(src, srcname);
    FST f = oracleTransducer(constructTransducer(rp, rp.pipeline[0], false));
    if (o.la) f = leafGraph(determinizeWords(f));
    SST sst = determinize(f);
    optimizeSST(sst, o.opt);
    out.sst_states.push_back((int)sst.states.size());
    out.stages.push_back(lower(sst, sst));
    std::ostringstream info;                  // Commands.hs:258-266
    info << "Options:\\n--opt " << o.opt << " --la=" << (o.la ? "true" : "false") << " --wordsize 8\\n\\nSource file: " << srcname << "\\nOracle SST states:  " << out.sst_states[0];
    out.info = info.str();
    return out;
  }
  Prog ast = parseKleenex(src, srcname);
  RProg rp = desugar(ast);
  for (int start : rp.pipeline) {
    const bool acts = stageHasActions(rp, start);   // register actions: in-band tokens + the action post-pass
    FST f = constructTransducer(rp, start, acts && o.act);   // (--act=false = compileDirect: refuses them, Commands.hs:165-168)
    // --la=true: the lookahead machine (word tests) is built as the reference builds it, then unrolled over its leaves
    // into single-symbol steps — the tables stay (state, class) tables and write what the lookahead machine writes
    if (o.la) f = leafGraph(determinizeWords(f));
    SST sst = determinize(f, o.backend == "hip" && !getenv("KEXC_NO_TABLE_CAP") ? 65535 : 0);   // singletonMode
    optimizeSST(sst, o.opt);
    out.sst_states.push_back((int)sst.states.size());
    out.stages.push_back(lower(sst, sst));
    if (acts) out.stages.back().act_regs = (int)rp.regnames.size();
  }
  std::ostringstream info;                    // Commands.hs:191-199
  info << "Options:\\n--opt " << o.opt << " --la=" << (o.la ? "true" : "false") << " --act=" << (o.act ? "true" : "false") << "\\n\\nSource file: " << srcname
       << "\\nSST states:  ";
  for (size_t i = 0; i < out.sst_states.size(); ++i) info << (i ? ", " : "") << out.sst_states[i];
  out.info = info.str();
  return out;
}

void writeBinary(const std::string& out, const std::vector<uint8_t>& blob, const std::string& dir) {
  std::ifstream drv(dir + "/kxrun", std::ios::binary);
  if (!drv) throw CompileError("host driver not found: " + dir + "/kxrun (run __graft_entry__.build())");
  std::ofstream bin(out, std::ios::binary | std::ios::trunc);
  if (!bin) throw CompileError("cannot write " + out);
  bin << drv.rdbuf();
  bin.write((const char*)blob.data(), blob.size());
  bin.write(dir.data(), dir.size());
  uint64_t bl = blob.size(), dl = dir.size();
  bin.write((const char*)&bl, 8); bin.write((const char*)&dl, 8); bin.write("KXRUNTRL", 8);
  bin.close();
  if (!bin.good()) { unlink(out.c_str()); throw CompileError("write error on " + out + " (disk full?)"); }
  chmod(out.c_str(), 0755);
}

}  // namespace kexc

// ------------------------------------------------------------------ C ABI
namespace {
thread_local std::string g_err;
char* dupBytes(const void* p, size_t n) {
  char* r = (char*)malloc(n ? n : 1);
  if (r && n) memcpy(r, p, n);
  return r;
}
}  // namespace

extern "C" {

int kexc_compile(const char* source, size_t source_len, const char* source_name, int opt_level,
                 unsigned char** blob, size_t* blob_len) {
  try {
    kexc::Options o; o.opt = opt_level;
    auto c = kexc::compileSource(std::string(source, source_len), source_name ? source_name : "<memory>", o);
    auto b = kexc::writeBlob(c.stages, c.info);
    *blob = (unsigned char*)dupBytes(b.data(), b.size());
    *blob_len = b.size();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return 1;
  }
}

int kexc_emit_c(const char* source, size_t source_len, const char* source_name, int opt_level,
                char** c_text, size_t* c_len) {
  try {
    kexc::Options o; o.opt = opt_level;
    auto c = kexc::compileSource(std::string(source, source_len), source_name ? source_name : "<memory>", o);
    std::string txt = kexc::emitC(c.stages, c.info);
    *c_text = dupBytes(txt.data(), txt.size());
    *c_len = txt.size();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return 1;
  }
}

// The nondeterministic transducer of every pipeline stage as JSON — for tests that simulate the FST
// directly (an evaluation route that shares nothing with determinization, lowering or the engines).
static void fstJson(std::ostringstream& o, const kexc::FST& f, bool toks) {
  o << "{\"tokens\":" << (toks ? "true" : "false") << ",\"nstates\":" << f.nstates << ",\"init\":" << f.init << ",\"final\":[";
  bool first = true;
  for (int q = 0; q < f.nstates; ++q) if (f.is_final[q]) { o << (first ? "" : ",") << q; first = false; }
  o << "],\"eps\":[";
  for (int q = 0; q < f.nstates; ++q) {
    o << (q ? "," : "") << "[";
    for (size_t k = 0; k < f.eps[q].size(); ++k) {
      o << (k ? "," : "") << "[[";
      for (size_t b = 0; b < f.eps[q][k].out.size(); ++b) o << (b ? "," : "") << (int)(unsigned char)f.eps[q][k].out[b];
      o << "]," << f.eps[q][k].to << "]";
    }
    o << "]";
  }
  o << "],\"sym\":[";
  for (int q = 0; q < f.nstates; ++q) {
    o << (q ? "," : "") << "[";
    for (size_t k = 0; k < f.sym[q].size(); ++k) {
      o << (k ? "," : "") << "[[";
      bool fr = true;
      for (int b = 0; b < 256;) {
        if (!f.sym[q][k].pred.has(b)) { ++b; continue; }
        int e = b; while (e + 1 < 256 && f.sym[q][k].pred.has(e + 1)) ++e;
        o << (fr ? "" : ",") << "[" << b << "," << e << "]"; fr = false; b = e + 1;
      }
      o << "]," << (f.sym[q][k].copy ? 1 : 0) << "," << f.sym[q][k].to << "," << f.sym[q][k].tbl << "]";
    }
    o << "]";
  }
  o << "],\"tables\":[";
  for (size_t k = 0; k < f.tables.size(); ++k) {
    o << (k ? "," : "") << "[";
    for (int b = 0; b < 256; ++b) o << (b ? "," : "") << (int)f.tables[k][(size_t)b];
    o << "]";
  }
  o << "]}";
}

int kexc_dump_fst(const char* source, size_t source_len, const char* source_name, char** json, size_t* json_len) {
  try {
    std::string name = source_name ? source_name : "<memory>";
    kexc::Prog ast = kexc::parseKleenex(std::string(source, source_len), name);
    kexc::RProg rp = kexc::desugar(ast);
    std::ostringstream o;
    o << "[";
    for (size_t si = 0; si < rp.pipeline.size(); ++si) {
      const bool toks = kexc::stageHasActions(rp, rp.pipeline[si]);
      kexc::FST f = kexc::constructTransducer(rp, rp.pipeline[si], toks);
      o << (si ? "," : "");
      fstJson(o, f, toks);
    }
    o << "]";
    std::string txt = o.str();
    *json = dupBytes(txt.data(), txt.size());
    *json_len = txt.size();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return 1;
  }
}

// kexc_compile with the flags spelled out: lookahead = `--la`, regex = the source is one regular expression (bit-coder)
int kexc_compile_flags(const char* source, size_t source_len, const char* source_name, int opt_level, int lookahead, int regex,
                       unsigned char** blob, size_t* blob_len) {
  try {
    kexc::Options o; o.opt = opt_level; o.la = lookahead != 0; o.regex = regex != 0;
    auto c = kexc::compileSource(std::string(source, source_len), source_name ? source_name : "<memory>", o);
    auto b = kexc::writeBlob(c.stages, c.info);
    *blob = (unsigned char*)dupBytes(b.data(), b.size());
    *blob_len = b.size();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return 1;
  }
}

// The lookahead machine (`--la=true`) of every stage in its path form as JSON — what a front end that compiles with
// the reference's default flags has in hand, for tests that marshal it into kexc_il_program's block form.
int kexc_dump_words(const char* source, size_t source_len, const char* source_name, int regex, char** json, size_t* json_len) {
  try {
    std::string name = source_name ? source_name : "<memory>";
    std::vector<std::pair<kexc::WordSST, int>> ws;
    if (regex) {
      kexc::RProg rp = kexc::parseRegexProgram(std::string(source, source_len), name);
      ws.push_back({kexc::determinizeWords(kexc::oracleTransducer(kexc::constructTransducer(rp, rp.pipeline[0], false))), -1});
    } else {
      kexc::RProg rp = kexc::desugar(kexc::parseKleenex(std::string(source, source_len), name));
      for (int start : rp.pipeline) {
        const bool acts = kexc::stageHasActions(rp, start);
        ws.push_back({kexc::determinizeWords(kexc::constructTransducer(rp, start, acts)), acts ? (int)rp.regnames.size() : -1});
      }
    }
    std::ostringstream o;
    auto bytes = [&](const std::string& b) { o << "["; for (size_t i = 0; i < b.size(); ++i) o << (i ? "," : "") << (int)(unsigned char)b[i]; o << "]"; };
    o << "[";
    for (size_t si = 0; si < ws.size(); ++si) {
      const kexc::WordSST& w = ws[si].first;
      o << (si ? "," : "") << "{\"init\":" << w.init << ",\"action_regs\":" << ws[si].second << ",\"init_path\":[";
      for (size_t l = 0; l < w.init_path.size(); ++l) { o << (l ? "," : ""); bytes(w.init_path[l]); }
      o << "],\"states\":[";
      for (size_t q = 0; q < w.states.size(); ++q) {
        const auto& st = w.states[q];
        o << (q ? "," : "") << "{\"nleaves\":" << st.nleaves << ",\"final_leaf\":" << st.final_leaf << ",\"edges\":[";
        for (size_t k = 0; k < st.edges.size(); ++k) {
          const auto& e = st.edges[k];
          o << (k ? "," : "") << "{\"to\":" << e.to << ",\"word\":[";
          for (size_t i = 0; i < e.word.size(); ++i) {   // each predicate as 32 bytes, bit b of the little-endian set = byte b
            o << (i ? "," : "") << "[";
            for (int x = 0; x < 32; ++x) o << (x ? "," : "") << (int)((e.word[i].w[x >> 3] >> (8 * (x & 7))) & 0xFF);
            o << "]";
          }
          o << "],\"path\":[";
          for (size_t l = 0; l < e.path.size(); ++l) {
            o << (l ? "," : "") << "{\"parent\":" << e.path[l].parent << ",\"steps\":[";
            for (size_t i = 0; i < e.path[l].steps.size(); ++i) { o << (i ? "," : "") << "[" << (e.path[l].steps[i].copy ? 1 : 0) << ","; bytes(e.path[l].steps[i].bytes); o << "," << e.path[l].steps[i].tbl << "]"; }
            o << "]}";
          }
          o << "]}";
        }
        o << "]}";
      }
      o << "],\"tables\":[";
      for (size_t k = 0; k < w.tables.size(); ++k) {
        o << (k ? "," : "") << "[";
        for (int b = 0; b < 256; ++b) o << (b ? "," : "") << (int)w.tables[k][(size_t)b];
        o << "]";
      }
      o << "]}";
    }
    o << "]";
    std::string txt = o.str();
    *json = dupBytes(txt.data(), txt.size());
    *json_len = txt.size();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return 1;
  }
}

// Regex flavour (`.re` / `.rx` / `--re`): the bit-coder — kexc.hs:46-48 → generateOracleSSTs → compileCoder.
int kexc_compile_regex(const char* regex, size_t regex_len, const char* source_name, int opt_level,
                       unsigned char** blob, size_t* blob_len) {
  try {
    kexc::Options o; o.opt = opt_level; o.regex = true;
    auto c = kexc::compileSource(std::string(regex, regex_len), source_name ? source_name : "<command line>", o);
    auto b = kexc::writeBlob(c.stages, c.info);
    *blob = (unsigned char*)dupBytes(b.data(), b.size());
    *blob_len = b.size();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return 1;
  }
}

// flavour 0: the regex's transducer before `oracle` (symbols copied); 1: its oracle machine
int kexc_dump_regex_fst(const char* regex, size_t regex_len, int oracle, char** json, size_t* json_len) {
  try {
    kexc::RProg rp = kexc::parseRegexProgram(std::string(regex, regex_len), "<command line>");
    kexc::FST f = kexc::constructTransducer(rp, rp.pipeline[0], false);
    if (oracle) f = kexc::oracleTransducer(f);
    std::ostringstream o;
    o << "[";
    fstJson(o, f, false);
    o << "]";
    std::string txt = o.str();
    *json = dupBytes(txt.data(), txt.size());
    *json_len = txt.size();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return 1;
  }
}

// compileProgram's seam (Backends/C.hs:529-540) for callers that bring their own SSTs: the Pipeline arrives as tables.
int kexc_emit_pipeline(int buffer_unit_bits, int cc_opt_level, void (*info)(const char* line, void* ctx), void* info_ctx,
                       const kexc_pipeline* pl, const char* env_info, const char* cc, const char* out_path,
                       const char* srcout_path, int word_alignment) {
  (void)cc_opt_level; (void)cc; (void)word_alignment;   // meaningful to a C compiler only; kept so that the argument list stays one-to-one
  try {
    using namespace kexc;
    if (!pl || !pl->programs || pl->nprograms == 0) throw CompileError("empty pipeline");
    if (pl->program_size != sizeof(kexc_il_program))
      throw CompileError("kexc_pipeline.program_size is " + std::to_string(pl->program_size) + ", this library's kexc_il_program has " +
                         std::to_string(sizeof(kexc_il_program)) + " bytes: the caller was built against another version of include/kexc_api.h");
    if (buffer_unit_bits != 8) throw CompileError("buffer unit must be 8 bits (UInt8T): the engine's output is a byte stream (--wordsize 8)");
    if (pl->is_oracle_action) throw CompileError("oracle/action pipelines (Right [(Program, Program)]) are not taken: the action program's registers hold data (ActionSST.hs:47-104), "
                                                    "not path choices, so it has no path form; hand over the transducer with its actions in band instead (has_actions, kxp_format.h) "
                                                    "or compile with --act=false");
    std::vector<StageTables> stages;
    for (uint32_t pi = 0; pi < pl->nprograms; ++pi) {
      const kexc_il_program& P = pl->programs[pi];
      auto bad = [&](const std::string& what) { throw CompileError("program " + std::to_string(pi) + ": " + what); };
      if (P.ntests && P.ntables) bad("block form (ntests) with symbol tables: AppendTblI does not occur in a lookahead program (kexc_api.h)");
      if (P.ntests) {
        // block form (--la=true): tests are words of predicates.  The annotation alone defines the function: unroll it
        // over the leaves into single-symbol steps and build the tables from that (kexc.h: leafGraph); the blocks'
        // register updates say the same thing a second time and are not read.
        if (!P.nstates || P.nstates >= 0xFFFF || P.init_state >= P.nstates || !P.maxleaves || P.maxleaves > 254) bad("state/leaf counts out of range");
        if (!P.nleaves || !P.final_leaf || !P.pconst_off || !P.init_const || !P.test_block || !P.test_target || !P.test_len || !P.test_preds || !P.test_back ||
            (P.npconsts && P.pconst_off[P.npconsts] && !P.pconst_pool)) bad("missing table");
        std::vector<std::string> pc;
        for (uint32_t c = 0; c < P.npconsts; ++c) {
          if (P.pconst_off[c + 1] < P.pconst_off[c]) bad("path constant offsets out of order");
          pc.emplace_back((const char*)P.pconst_pool + P.pconst_off[c], P.pconst_off[c + 1] - P.pconst_off[c]);
        }
        WordSST w;
        w.states.resize(P.nstates); w.init = (int)P.init_state;
        for (uint32_t q = 0; q < P.nstates; ++q) {
          if (P.nleaves[q] == 0 || P.nleaves[q] > P.maxleaves || (P.final_leaf[q] != 0xFF && P.final_leaf[q] >= P.nleaves[q])) bad("leaf counts out of range");
          w.states[q].nleaves = P.nleaves[q]; w.states[q].final_leaf = P.final_leaf[q] == 0xFF ? -1 : (int)P.final_leaf[q];
        }
        for (uint32_t l = 0; l < P.nleaves[P.init_state]; ++l) {
          if (P.init_const[l] >= P.npconsts) bad("initial constant out of range");
          w.init_path.push_back(pc[P.init_const[l]]);
        }
        size_t row = 0;
        for (uint32_t k = 0; k < P.ntests; ++k) {
          const uint32_t q = P.test_block[k], to = P.test_target[k], n = P.test_len[k];
          if (q >= P.nstates || to >= P.nstates || n == 0 || n > 255) bad("test out of range");
          WordEdge e; e.to = (int)to;
          for (uint32_t i = 0; i < n; ++i) {
            ByteSet p; memcpy(p.w, P.test_preds + 32 * (row + i), 32);
            if (p.empty()) bad("empty predicate in a test");
            e.word.push_back(p);
          }
          for (uint32_t l = 0; l < P.nleaves[to]; ++l) {
            WordPath wp; wp.steps.resize(n);
            for (uint32_t i = 0; i < n; ++i) {
              const uint32_t v = P.test_back[(row + i) * P.maxleaves + l];
              if (v == 0xFFFFFFFFu || (v >> 9) >= P.npconsts) bad("backward entry out of range");
              if (i == 0) { wp.parent = (int)(v & 0xFF); if ((uint32_t)wp.parent >= P.nleaves[q]) bad("backward entry out of range"); }
              wp.steps[i].copy = (v >> 8) & 1; wp.steps[i].bytes = pc[v >> 9];
            }
            e.path.push_back(std::move(wp));
          }
          w.states[q].edges.push_back(std::move(e));
          row += n;
        }
        SST sst = determinize(leafGraph(w));
        optimizeSST(sst, 3);
        StageTables t = lower(sst, sst);
        if (P.has_actions && P.action_regs > KXP_MAX_ACTION_REGS) bad("too many action registers (at most " + std::to_string(KXP_MAX_ACTION_REGS) + ")");
        t.act_regs = P.has_actions ? (int)P.action_regs : -1;
        stages.push_back(std::move(t));
        continue;
      }
      if (!P.nstates || P.nstates >= 0xFFFF || !P.nclasses || P.nclasses > 256 || P.init_state >= P.nstates) bad("state/class counts out of range");
      if (!P.maxleaves || P.maxleaves > 254) bad("maxleaves out of range");
      if (!P.class_of || !P.delta || !P.action || !P.final_action || !P.action_off || !P.const_off || !P.back_row || !P.nleaves || !P.final_leaf ||
          !P.back || !P.pconst_off || !P.init_const) bad("missing table");
      if ((P.nactions && P.action_off[P.nactions] && !P.ops) || (P.nconsts && P.const_off[P.nconsts] && !P.const_pool) ||
          (P.npconsts && P.pconst_off[P.npconsts] && !P.pconst_pool)) bad("missing table (micro-ops or constant pool)");
      StageTables t;
      t.nstates = (int)P.nstates; t.nclasses = (int)P.nclasses; t.q0 = (int)P.init_state; t.nregs = (int)P.nregs; t.maxleaves = (int)P.maxleaves;
      const size_t sc = (size_t)P.nstates * P.nclasses;
      for (int b = 0; b < 256; ++b) { if (P.class_of[b] >= P.nclasses) bad("byte class out of range"); t.cls[b] = P.class_of[b]; }
      t.delta.assign(P.delta, P.delta + sc); t.act.assign(P.action, P.action + sc); t.final_act.assign(P.final_action, P.final_action + P.nstates);
      t.pback.assign(P.back_row, P.back_row + sc);
      for (uint32_t c = 0; c < P.nconsts; ++c) {
        if (P.const_off[c + 1] < P.const_off[c]) bad("constant offsets out of order");
        t.consts.emplace_back((const char*)P.const_pool + P.const_off[c], P.const_off[c + 1] - P.const_off[c]);
      }
      for (uint32_t c = 0; c < P.npconsts; ++c) {
        if (P.pconst_off[c + 1] < P.pconst_off[c]) bad("path constant offsets out of order");
        t.pconsts.emplace_back((const char*)P.pconst_pool + P.pconst_off[c], P.pconst_off[c + 1] - P.pconst_off[c]);
      }
      if (P.ntables && (!P.tbl_width || !P.tbl_data || !P.back_table)) bad("missing table data");
      std::vector<size_t> tbl_at(P.ntables, 0);
      for (uint32_t k = 0, at = 0; k < P.ntables; ++k) { if (P.tbl_width[k] > 8) bad("table digits wider than 8"); tbl_at[k] = at; at += 256 * P.tbl_width[k]; }
      auto tblEntry = [&](uint32_t k, int sym) { return std::string((const char*)P.tbl_data + tbl_at[k] + (size_t)sym * P.tbl_width[k], P.tbl_width[k]); };
      // what a table entry becomes when it is written out as a constant: in a stage with register actions the output is a token
      // stream whose escape byte is FF, so a table value FF leaves as FF FF like any other data byte (kxp_format.h)
      auto tblOut = [&](uint32_t k, int sym) {
        std::string v = tblEntry(k, sym);
        if (!P.has_actions) return v;
        std::string e;
        for (char ch : v) { e += ch; if ((uint8_t)ch == KXP_ESC) e += ch; }
        return e;
      };
      bool uses_tables = false;
      for (uint32_t a = 0; a < P.nactions; ++a) {
        if (P.action_off[a + 1] < P.action_off[a]) bad("action offsets out of order");
        std::vector<MicroOp> ops;
        for (uint32_t k = P.action_off[a]; k < P.action_off[a + 1]; ++k) {
          const uint32_t w0 = P.ops[2 * k], arg = P.ops[2 * k + 1], op = w0 >> 24, dst = w0 & 0xFFFFFF;
          if (op > 4 || dst >= P.nregs || (op == 1 && arg >= P.nconsts) || (op == 3 && arg >= P.nregs) || (op == 4 && arg >= P.ntables)) bad("malformed micro-op");
          uses_tables |= op == 4;
          ops.push_back(MicroOp{(uint8_t)op, (uint16_t)dst, arg});
        }
        t.actions.push_back(ops);
      }
      t.back.assign(P.back, P.back + (size_t)P.nback * P.maxleaves);
      std::vector<uint32_t> btab(t.back.size(), 0xFFFFFFFFu);
      if (P.ntables) {
        btab.assign(P.back_table, P.back_table + t.back.size());
        for (size_t i = 0; i < btab.size(); ++i)
          if (btab[i] != 0xFFFFFFFFu) {
            if (btab[i] >= P.ntables || t.back[i] == 0xFFFFFFFFu || ((t.back[i] >> 8) & 1)) bad("backward table entry out of range");
            uses_tables = true;
          }
      }
      for (uint32_t q = 0; q < P.nstates; ++q)
        if (t.final_act[q] != 0xFFFFFFFFu && t.final_act[q] < P.nactions)
          for (auto& m : t.actions[t.final_act[q]]) if (m.op == 4) bad("AppendTblI in a final action (no symbol to index with)");
      // (a stage with register actions takes no table atoms — the engine's parseStage refuses the pair — and the engine's entries
      //  name at most KXP_ENGINE_TABLES tables: both go the written-out way)
      bool native_tables = uses_tables && !getenv("KEXC_LOWER_TABLES") && !P.has_actions && P.ntables <= KXP_ENGINE_TABLES;
      for (uint32_t k = 0; k < P.ntables; ++k) native_tables = native_tables && P.tbl_width[k] == 1;
      for (uint32_t e : t.back) native_tables = native_tables && (e == 0xFFFFFFFFu || (e >> 9) < (1u << 15));
      if (native_tables) {
        // one-byte tables stay TABLE ATOMS (round 3): micro-op 4 as it is, the path entry copies through its table
        for (uint32_t k = 0; k < P.ntables; ++k) { std::array<uint8_t, 256> tb; memcpy(tb.data(), P.tbl_data + tbl_at[k], 256); t.tables.push_back(tb); }
        for (size_t i = 0; i < btab.size(); ++i) if (btab[i] != 0xFFFFFFFFu) t.back[i] |= 0x100u | ((btab[i] + 1) << 24);
      } else if (uses_tables) {
        // AppendTblI → AppendI (tables of wider digits): refine the byte classes until every table is constant on each class, then give each
        // (state, class) its own copy of the action / backward row with the table entries written out as constants
        for (size_t i = 0; i < sc; ++i) if (t.delta[i] != 0xFFFF && (t.act[i] >= P.nactions || t.pback[i] >= P.nback)) bad("transition out of range");
        std::map<std::string, int> sig2cls;
        uint8_t ncls[256]; std::vector<int> rep, oldc;
        for (int b = 0; b < 256; ++b) {
          std::string sig(1, (char)t.cls[b]);
          for (uint32_t k = 0; k < P.ntables; ++k) sig += tblEntry(k, b);
          auto it = sig2cls.find(sig);
          if (it == sig2cls.end()) { it = sig2cls.emplace(sig, (int)rep.size()).first; rep.push_back(b); oldc.push_back(t.cls[b]); }
          ncls[b] = (uint8_t)it->second;
        }
        const int NC = (int)rep.size();
        std::map<std::string, uint32_t> constId, pconstId;
        for (uint32_t c = 0; c < t.consts.size(); ++c) constId.emplace(t.consts[c], c);
        for (uint32_t c = 0; c < t.pconsts.size(); ++c) pconstId.emplace(t.pconsts[c], c);
        auto internC = [&](const std::string& v) { auto it = constId.find(v); if (it != constId.end()) return it->second; t.consts.push_back(v); return constId[v] = (uint32_t)t.consts.size() - 1; };
        auto internP = [&](const std::string& v) { auto it = pconstId.find(v); if (it != pconstId.end()) return it->second; t.pconsts.push_back(v); return pconstId[v] = (uint32_t)t.pconsts.size() - 1; };
        std::map<std::pair<uint32_t, std::string>, uint32_t> actMemo, rowMemo;
        std::vector<uint16_t> nd((size_t)P.nstates * NC, 0xFFFF);
        std::vector<uint32_t> na(nd.size(), 0), nb(nd.size(), 0);
        const uint32_t ML = P.maxleaves;
        for (uint32_t q = 0; q < P.nstates; ++q)
          for (int c = 0; c < NC; ++c) {
            const size_t o = (size_t)q * P.nclasses + oldc[c], n = (size_t)q * NC + c;
            nd[n] = t.delta[o];
            if (nd[n] == 0xFFFF) continue;
            std::string key;
            for (uint32_t k = 0; k < P.ntables; ++k) key += tblEntry(k, rep[c]) + '\x01';
            {
              bool any = false;
              for (auto& m : t.actions[t.act[o]]) any |= m.op == 4;
              if (!any) na[n] = t.act[o];
              else {
                auto it = actMemo.find({t.act[o], key});
                if (it == actMemo.end()) {
                  std::vector<MicroOp> ops = t.actions[t.act[o]];
                  for (auto& m : ops) if (m.op == 4) { m.arg = internC(tblOut(m.arg, rep[c])); m.op = 1; }
                  t.actions.push_back(ops);
                  it = actMemo.emplace(std::make_pair(t.act[o], key), (uint32_t)t.actions.size() - 1).first;
                }
                na[n] = it->second;
              }
            }
            {
              const uint32_t row = t.pback[o];
              bool any = false;
              for (uint32_t l = 0; l < ML; ++l) any |= btab[(size_t)row * ML + l] != 0xFFFFFFFFu;
              if (!any) nb[n] = row;
              else {
                auto it = rowMemo.find({row, key});
                if (it == rowMemo.end()) {
                  const uint32_t nr = (uint32_t)(t.back.size() / ML);
                  for (uint32_t l = 0; l < ML; ++l) {
                    uint32_t e = t.back[(size_t)row * ML + l];
                    const uint32_t tb = btab[(size_t)row * ML + l];
                    if (tb != 0xFFFFFFFFu) {
                      if ((e >> 9) >= P.npconsts) bad("backward entry out of range");
                      e = (e & 0x1FF) | (internP(tblOut(tb, rep[c]) + t.pconsts[e >> 9]) << 9);
                    }
                    t.back.push_back(e);
                  }
                  it = rowMemo.emplace(std::make_pair(row, key), nr).first;
                }
                nb[n] = it->second;
              }
            }
          }
        t.nclasses = NC; memcpy(t.cls, ncls, 256);
        t.delta = std::move(nd); t.act = std::move(na); t.pback = std::move(nb);
      }
      const size_t sc2 = (size_t)t.nstates * t.nclasses, nback2 = t.back.size() / P.maxleaves;
      for (size_t i = 0; i < sc2; ++i) {
        if (t.delta[i] == 0xFFFF) continue;
        if (t.delta[i] >= P.nstates || t.act[i] >= t.actions.size() || t.pback[i] >= nback2) bad("transition out of range");
      }
      for (uint32_t q = 0; q < P.nstates; ++q) {
        if (t.final_act[q] != 0xFFFFFFFFu && t.final_act[q] >= P.nactions) bad("final action out of range");
        if (P.nleaves[q] == 0 || P.nleaves[q] > P.maxleaves || (P.final_leaf[q] != 0xFF && P.final_leaf[q] >= P.nleaves[q])) bad("leaf counts out of range");
        if ((t.final_act[q] == 0xFFFFFFFFu) != (P.final_leaf[q] == 0xFF)) bad("final action and final leaf disagree");
      }
      t.nleaves.assign(P.nleaves, P.nleaves + P.nstates); t.fin_leaf.assign(P.final_leaf, P.final_leaf + P.nstates);
      for (uint32_t e : t.back) if (e != 0xFFFFFFFFu && ((e & 0xFF) >= P.maxleaves || ((e >> 9) & (t.tables.empty() ? 0x7FFFFFu : 0x7FFFu)) >= t.pconsts.size())) bad("backward entry out of range");
      t.init_const.assign(P.init_const, P.init_const + P.maxleaves);
      for (uint32_t v : t.init_const) if (v >= P.npconsts) bad("initial constant out of range");
      if (P.has_actions && P.action_regs > KXP_MAX_ACTION_REGS) bad("too many action registers (at most " + std::to_string(KXP_MAX_ACTION_REGS) + ")");
      t.act_regs = P.has_actions ? (int)P.action_regs : -1;
      buildSync(t);
      stages.push_back(std::move(t));
    }
    const std::string infotxt = env_info ? env_info : "";
    std::vector<uint8_t> blob = writeBlob(stages, infotxt);
    if (info) {   // the reference reports the size of what it generated (C.hs:553)
      const std::string line = "Generated a KXP table blob of " + std::to_string(blob.size()) + " bytes (" + std::to_string(stages.size()) + " phase(s)).";
      info(line.c_str(), info_ctx);
    }
    if (srcout_path && *srcout_path) {
      std::ofstream f(srcout_path, std::ios::binary | std::ios::trunc);
      if (!f) throw CompileError(std::string("cannot write ") + srcout_path);
      f.write((const char*)blob.data(), blob.size());
      f.close();
      if (!f.good()) throw CompileError(std::string("write error on ") + srcout_path);
    }
    if (out_path && *out_path) {
      Dl_info di;
      std::string dir = ".";
      if (dladdr((const void*)&kexc_emit_pipeline, &di) && di.dli_fname) { dir = di.dli_fname; size_t k = dir.rfind('/'); dir = k == std::string::npos ? "." : dir.substr(0, k); }
      writeBinary(out_path, blob, dir);
    }
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return 1;
  }
}

const char* kexc_last_error(void) { return g_err.c_str(); }
void kexc_free(void* p) { free(p); }

}  // extern "C"
