// kxrun.cpp — host driver of a compiled Kleenex binary (`kexc compile … --out BIN`).
//
// Keeps the command-line contract of the reference's generated binaries
// (crt/crt.c:326-467): `BIN < in > out`; `-i` prints compile info and exits 2;
// `-t` prints "time (ms): N" on stderr; `-h`/unknown prints usage on stdout and
// exits 1; a rejected input prints "Match error at input symbol <count>!" on
// stderr and exits 1.  `-p/--phase K` runs only phase K, stdin to stdout, as in
// the reference (crt.c:390-393,408-411); without it the phases of a pipeline run
// as chained device-resident stages inside one process (instead of
// fork()+pipe(), crt.c:414-454).  `--gpus N` (no counterpart in the reference) shards a regular
// file on stdin over N GPUs (kx_run_fd_sharded).
//
// BIN = this executable ++ KXP blob ++ libdir ++ trailer (see kexc main.cpp).
// The engine is loaded with dlopen so that this file carries no HIP dependency.
#include <dlfcn.h>
#include <getopt.h>
#include <sys/time.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/kxhip.h"

// The engine's switches are fields of kx_config (include/kxhip.h); the library reads no environment variable for them.  A produced
// binary keeps honouring the variable names that earlier rounds' scripts use: they are read HERE, once, and mapped onto the struct.
static kx_config configFromEnv() {
  kx_config c{};
  auto num = [](const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; };
  auto tri = [&](const char* name) -> uint32_t { const char* e = getenv(name); return !e ? 0u : atoi(e) ? 2u : 1u; };   // unset: auto, 0: off, else: on
  if (const char* e = getenv("KX_DF")) { const int v = atoi(e); c.delayed_form = v == 0 ? 1u : v == 2 ? 2u : 0u; }
  c.delay = (uint32_t)num("KX_DF_K", 0);
  if (getenv("KX_DF_J")) c.merge_window = (uint32_t)num("KX_DF_J", 0) + 1;
  c.inline_consts = tri("KX_INL");
  c.job_stride = getenv("KX_JL") ? tri("KX_JL") : getenv("KX_JL_AUTO_OFF") ? 1u : 0u;
  if (getenv("KX_NO_DIRECT")) c.disable |= KX_OFF_DIRECT;
  if (getenv("KX_NO_PAIR")) c.disable |= KX_OFF_PAIR;
  if (getenv("KX_NO_CMPX")) c.disable |= KX_OFF_CMPX;
  if (getenv("KX_NO_COOP")) c.disable |= KX_OFF_COOP;
  if (getenv("KX_NO_SLOW")) c.disable |= KX_OFF_SLOW;
  if (getenv("KX_FORCE_BIG")) c.force |= KX_FORCE_BIG;
  if (getenv("KX_FORCE_TBLMODE")) c.force |= KX_FORCE_TBLMODE;
  if (getenv("KX_ACT_SEQ")) c.force |= KX_FORCE_ACT_SEQ;
  if (getenv("KX_SHARD_SAME_DEVICE")) c.force |= KX_FORCE_SAME_DEVICE;
  c.emit_waves = (uint32_t)num("KX_EMIT_WAVES", 0);
  c.emit_half = tri("KX_EMIT_HALF");
  c.emit_inplace = tri("KX_EMIT_INPLACE");
  c.emit_staging = (uint32_t)num("KX_EMIT_STG", 0);
  c.df_backoff = (uint32_t)num("KX_DF_BACKOFF_OFF", 0) ? 1u : 0u;
  c.debug_flags = (uint32_t)num("KX_DEBUG_FLAGS", 0);
  c.act_par_min = (uint32_t)num("KX_ACT_PAR_MIN", 0);
  c.act_prefix3_min = (uint32_t)num("KX_ACT_PREFIX3_MIN", 0);
  c.act_lanes = tri("KX_ACT_LANES");
  c.act_chunk = (uint32_t)num("KX_ACT_CHUNK", 0);
  return c;
}

static void usage(const char* name) {
  fprintf(stdout, "Normal usage: %s < infile > outfile\n", name);
  fprintf(stdout, "- \"%s\": reads from stdin and writes to stdout.\n", name);
  fprintf(stdout, "- \"%s -i\": prints compilation info.\n", name);
  fprintf(stdout, "- \"%s -t\": runs normally, but prints timing to stderr.\n", name);
}

int main(int argc, char** argv) {
  // locate the payload appended to this executable
  FILE* self = fopen("/proc/self/exe", "rb");
  if (!self) { perror("/proc/self/exe"); return 1; }
  fseek(self, 0, SEEK_END);
  long size = ftell(self);
  char trailer[24];
  if (size < 24 || fseek(self, size - 24, SEEK_SET) || fread(trailer, 1, 24, self) != 24 || memcmp(trailer + 16, "KXRUNTRL", 8)) {
    fprintf(stderr, "%s: no compiled program attached (use `kexc compile prog.kex --out BIN`)\n", argv[0]);
    return 1;
  }
  uint64_t bl, dl;
  memcpy(&bl, trailer, 8); memcpy(&dl, trailer + 8, 8);
  if (bl < 20 || bl > (uint64_t)size || dl > (uint64_t)size || bl + dl + 24 > (uint64_t)size) { fprintf(stderr, "%s: corrupt payload trailer\n", argv[0]); return 1; }
  std::vector<unsigned char> blob(bl);
  std::string libdir(dl, '\0');
  fseek(self, size - 24 - (long)dl - (long)bl, SEEK_SET);
  if (fread(blob.data(), 1, bl, self) != bl || fread(&libdir[0], 1, dl, self) != dl) { fprintf(stderr, "corrupt payload\n"); return 1; }
  fclose(self);

  static struct option long_options[] = {{"phase", required_argument, 0, 'p'}, {"gpus", required_argument, 0, 'g'}, {0, 0, 0, 0}};
  bool timing = false;
  long phase = 0, gpus = 0;
  int c;
  while ((c = getopt_long(argc, argv, "ihtp:", long_options, nullptr)) != -1) {
    switch (c) {
      case 'i': {
        uint32_t il; memcpy(&il, blob.data() + 16, 4);
        if ((uint64_t)il + 20 > blob.size()) { fprintf(stderr, "%s: corrupt payload\n", argv[0]); return 1; }
        std::string info((const char*)blob.data() + 20, il);
        for (size_t p; (p = info.find("\\n")) != std::string::npos;) info.replace(p, 2, "\n");
        fprintf(stdout, "%s\n", info.c_str());
        return 2;
      }
      case 't': timing = true; break;
      case 'g': gpus = atol(optarg); if (gpus < 1 || gpus > 64) { fprintf(stderr, "Invalid number of GPUs: %ld given\n", gpus); return 1; } break;
      case 'p': phase = atol(optarg); if (phase < 1) { fprintf(stderr, "Invalid phase: %ld given\n", phase); return 1; } break;
      case 'h':
      default: usage(argv[0]); return 1;
    }
  }
  struct timeval t0, t1;
  if (timing) gettimeofday(&t0, nullptr);

  const char* env = getenv("KXHIP_LIB");
  std::string lib = env ? env : libdir + "/libkxhip.so";
  void* h = dlopen(lib.c_str(), RTLD_NOW);
  if (!h) { fprintf(stderr, "%s: cannot load the HIP engine: %s\n", argv[0], dlerror()); return 1; }
  auto load = (int (*)(const void*, size_t, const kx_config*, kx_program**))dlsym(h, "kx_load_config");
  auto run = (int (*)(kx_program*, int, int, kx_stats*))dlsym(h, "kx_run_fd");
  auto lasterr = (const char* (*)(void))dlsym(h, "kx_last_error");
  if (!load || !run || !lasterr) { fprintf(stderr, "%s: engine library lacks required symbols\n", argv[0]); return 1; }
  kx_stats st;
  int rc;
  kx_config cfg = configFromEnv();
  if (gpus) {
    auto runs = (int (*)(const void*, size_t, const kx_config*, int, int, int, kx_stats*))dlsym(h, "kx_run_fd_sharded_cfg");
    if (phase) { fprintf(stderr, "%s: --gpus cannot be combined with --phase\n", argv[0]); return 1; }
    if (!runs) { fprintf(stderr, "%s: this libkxhip.so has no kx_run_fd_sharded_cfg (--gpus needs the engine library of round 6 or later)\n", argv[0]); return 1; }
    rc = runs(blob.data(), blob.size(), &cfg, (int)gpus, STDIN_FILENO, STDOUT_FILENO, &st);
  } else {
  kx_program* prog = nullptr;
  if (load(blob.data(), blob.size(), &cfg, &prog)) { fprintf(stderr, "%s: %s\n", argv[0], lasterr()); return 1; }
  if (phase) {
    auto setcfg = (int (*)(kx_program*, const kx_config*))dlsym(h, "kx_set_config");
    auto nst = (uint32_t (*)(const kx_program*))dlsym(h, "kx_num_stages");
    if (!setcfg || !nst) { fprintf(stderr, "%s: engine library lacks required symbols\n", argv[0]); return 1; }
    if ((unsigned long)phase > nst(prog)) { fprintf(stderr, "Invalid phase: %ld given\n", phase); return 1; }   // (crt.c match(): default case)
    cfg.phase = (uint32_t)phase;
    if (setcfg(prog, &cfg)) { fprintf(stderr, "%s: %s\n", argv[0], lasterr()); return 1; }
  }
  rc = run(prog, STDIN_FILENO, STDOUT_FILENO, &st);
  }
  if (rc == KX_MATCH_ERROR) {
    // `kexc simulate` runs its program through this driver and wants the reference simulators' words instead of the
    // compiled binary's (Commands.hs:285,298 "Reject"; SymbolicSST.hs:425,427)
    const char* sim = getenv("KX_SIM_MESSAGES");
    if (sim && strcmp(sim, "sst") != 0) fprintf(stderr, "Reject\n");
    else if (sim) fprintf(stderr, "%s\n", st.fail_stage == 0 && st.fail_pos >= st.in_bytes ? "End of input reached, but final state is not accepting." : "No match");
    else fprintf(stderr, "Match error at input symbol %zu!\n", (size_t)st.fail_pos);
    return 1;
  }
  if (rc) { fprintf(stderr, "%s: %s\n", argv[0], lasterr()); return 1; }
  if (timing) {
    gettimeofday(&t1, nullptr);
    long ms = (t1.tv_sec - t0.tv_sec) * 1000 + (t1.tv_usec - t0.tv_usec) / 1000;
    fprintf(stderr, "time (ms): %ld\n", ms);
  }
  return 0;
}
