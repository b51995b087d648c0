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

const char* kexc_last_error(void) { return g_err.c_str(); }
void kexc_free(void* p) { free(p); }

}  // extern "C"
