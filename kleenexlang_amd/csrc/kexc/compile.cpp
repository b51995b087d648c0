// compile.cpp — pipeline driver + C-ABI of the compiler library (libkexc.so).
//
// Mirrors the direct-mode compile path of the reference:
//   createProgram → buildTransducers → generateDirectSSTs → compileDirect
//   (src/KMC/Frontend/Commands.hs:50-68,82-115,155-180,182-210).
#include <cstring>
#include <sstream>

#include "../../../include/kexc_api.h"
#include "kexc.h"

namespace kexc {

Compiled compileSource(const std::string& src, const std::string& srcname, const Options& o) {
  Compiled out;
  Prog ast = parseKleenex(src, srcname);
  RProg rp = desugar(ast);
  for (int start : rp.pipeline) {
    FST f = constructTransducer(rp, start);
    SST sst = determinize(f);                 // --la=false semantics (singletonMode)
    optimizeSST(sst, o.opt);
    out.sst_states.push_back((int)sst.states.size());
    out.stages.push_back(lower(sst, sst));
  }
  std::ostringstream info;                    // Commands.hs:191-199
  info << "Options:\\n--opt " << o.opt << " --la=false --act=false (direct mode)\\n\\nSource file: " << srcname
       << "\\nSST states:  ";
  for (size_t i = 0; i < out.sst_states.size(); ++i) info << (i ? ", " : "") << out.sst_states[i];
  out.info = info.str();
  return out;
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
int kexc_dump_fst(const char* source, size_t source_len, const char* source_name, char** json, size_t* json_len) {
  try {
    std::string name = source_name ? source_name : "<memory>";
    kexc::Prog ast = kexc::parseKleenex(std::string(source, source_len), name);
    kexc::RProg rp = kexc::desugar(ast);
    std::ostringstream o;
    o << "[";
    for (size_t si = 0; si < rp.pipeline.size(); ++si) {
      kexc::FST f = kexc::constructTransducer(rp, rp.pipeline[si]);
      o << (si ? "," : "") << "{\"nstates\":" << f.nstates << ",\"init\":" << f.init << ",\"final\":[";
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
          o << "]," << (f.sym[q][k].copy ? 1 : 0) << "," << f.sym[q][k].to << "]";
        }
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

const char* kexc_last_error(void) { return g_err.c_str(); }
void kexc_free(void* p) { free(p); }

}  // extern "C"
