// main.cpp — the `kexc` command line.
//
// Keeps the reference's outer CLI for the compile path (src/kexc.hs:12-50,
// options src/KMC/Frontend/Options.hs:146-181):
//     kexc compile [main opts] [compile opts] FILE.kex --out BIN
// Flags may appear before or after the subcommand and after the positional
// file; booleans accept `--quiet` or `--quiet=true`.  Exactly one positional
// argument, else usage + exit 1 (kexc.hs:22-27).
//
// Back ends:
//   --backend=hip (default)  BIN = the generic host driver `kxrun` with the KXP
//                            blob appended; at run time it drives libkxhip.so
//                            (stdin → HBM → HIP engine → stdout).
//   --backend=c              reference-shaped C piped to `cc -O3 -xc -o BIN
//                            "-D FLAG_WORDALIGNED" -` exactly as C.hs:556-568
//                            does; needs --crt-dir (directory holding crt.c).
#include <signal.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <unistd.h>

#include <climits>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "kexc.h"

using namespace kexc;

static std::string selfDir() {
  char buf[PATH_MAX];
  ssize_t n = readlink("/proc/self/exe", buf, sizeof buf - 1);
  if (n <= 0) return ".";
  buf[n] = 0;
  std::string s(buf);
  return s.substr(0, s.rfind('/'));
}

static int usage() {
  std::cout << "Usage: kexc compile [--quiet] [--opt N] [--la[=BOOL]] [--act[=BOOL]] [--copt N] [--cc CC]\n"
               "                    [--backend=hip|c] [--crt-dir DIR] [--srcout FILE] [--blob FILE] [--wordsize 8] FILE.kex --out BIN\n"
               "       kexc compile … FILE.re|FILE.rx --out BIN   |   kexc compile … --re 'REGEX' --out BIN\n"
               "                    (regex flavour: BIN writes the code of the greedy parse, one byte per choice)\n"
               "       kexc simulate|interpret [--sim lockstep|backtrack|sst] [--quiet] [--opt N] [--act[=BOOL]] FILE.kex  < in > out\n"
               "                    (sst: the compiled program on the HIP engine; lockstep (default), backtrack: the FST simulators,\n"
               "                     on the CPU; `interpret` = `simulate --quiet`)\n"
               "--la defaults to false here (the reference: true, Options.hs:153): both machines write the same bytes\n"
               "(Tests/Regression.hs:45-53); the direct tables have fewer states, and the path form has no registers for lookahead to save.\n"
               "The reference's `visualize` subcommand is not part of this build.\n";
  return 1;
}

static bool parseBool(const std::string& v) {
  if (v == "true" || v == "True" || v == "1") return true;
  if (v == "false" || v == "False" || v == "0") return false;
  throw CompileError("bad boolean value: " + v);
}

int main(int argc, char** argv) {
  std::vector<std::string> pos;
  Options o;
  std::string crtdir, sub, sim = "lockstep";
  bool report = false, expr = false;
  try {
    for (int i = 1; i < argc; ++i) {
      std::string a = argv[i];
      if (a.rfind("--", 0) != 0) { if (sub.empty()) sub = a; else pos.push_back(a); continue; }
      std::string key = a.substr(2), val; bool hasval = false;
      size_t eq = key.find('=');
      if (eq != std::string::npos) { val = key.substr(eq + 1); key = key.substr(0, eq); hasval = true; }
      auto need = [&]() { if (hasval) return val; if (i + 1 >= argc) throw CompileError("option --" + key + " needs a value"); return std::string(argv[++i]); };
      auto flag = [&]() { return hasval ? parseBool(val) : true; };
      if (key == "quiet") o.quiet = flag();
      else if (key == "report") report = flag();
      else if (key == "opt") o.opt = std::stoi(need());
      else if (key == "la") o.la = flag();
      else if (key == "act") o.act = flag();
      else if (key == "re") expr = flag();                  // the argument is the regular expression itself (Options.hs optExpressionArg)
      else if (key == "func" || key == "sb" || key == "ite" || key == "rmidtbls") flag();
      else if (key == "wordsize") {                         // Options.hs:130-144
        const std::string w = need();
        if (w != "8" && w != "16" && w != "32" && w != "64") throw CompileError("\"" + w + "\" is not a valid word size");
        o.wordsize = std::stoi(w);
      }
      else if (key == "metric" || key == "approxmode") need();
      else if (key == "sim") { sim = need(); if (sim != "lockstep" && sim != "backtrack" && sim != "sst") throw CompileError("\"" + sim + "\" is not a valid simulation type"); }
      else if (key == "copt") o.copt = std::stoi(need());
      else if (key == "out") o.out = need();
      else if (key == "srcout") o.srcout = need();
      else if (key == "cc") o.cc = need();
      else if (key == "backend") o.backend = need();
      else if (key == "crt-dir") crtdir = need();
      else if (key == "blob") o.blobout = need();
      else if (key == "help") return usage();
      else throw CompileError("unknown option --" + key);
    }
    const bool simulate = sub == "simulate" || sub == "interpret";
    if ((sub != "compile" && !simulate) || pos.size() != 1) return usage();
    if (sub == "interpret") o.quiet = true;
    const std::string& file = pos[0];
    if (o.wordsize != 8) {
      // The reference always compiles with -D FLAG_WORDALIGNED (kexc.hs:40-48 pass True; C.hs:567), under which a write of
      // fewer bits than the buffer unit does not advance to the next unit (crt.c:143-155): with 8-bit symbols a unit wider
      // than 8 bits makes successive writes overwrite each other.  There is no defined output to reproduce.
      std::cerr << "--wordsize " << o.wordsize << ": only a buffer unit of 8 bits gives defined output for 8-bit symbols (the reference's word-aligned "
                   "runtime overwrites partial units, crt/crt.c:143-155); use --wordsize 8\n";
      return 1;
    }
    std::stringstream ss;
    std::string srcname = file;
    if (expr) { ss << file; srcname = "<command line>"; o.regex = true; }   // Commands.hs:70-72
    else {
      // getCompileFlavor (Frontend.hs:140-152)
      const size_t slash = file.find_last_of('/'), dot = file.find_last_of('.');
      const std::string ext = dot == std::string::npos || (slash != std::string::npos && dot < slash) ? "" : file.substr(dot);
      if (ext == ".re" || ext == ".rx") o.regex = true;
      else if (ext != ".kex") { std::cerr << "Unknown extension: '" << ext << "'.\nExpects one of '.kex', '.re', or '.rx'.\n"; return 1; }
      std::ifstream in(file, std::ios::binary);
      if (!in) { std::cerr << file << ": openFile: does not exist (No such file or directory)\n"; return 1; }
      ss << in.rdbuf();
    }
    if (!o.quiet && !simulate) std::cout << "Compile: " << srcname << (o.regex ? " (regex flavour: bit-coder; " : " (direct mode; ") << (o.la ? "--la=true: word tests unrolled over the path tree's leaves)\n" : "--la=false)\n");
    if (simulate && sim != "sst") {
      // Commands.hs:277-302: the FST simulators run the nondeterministic transducers themselves, on the CPU, as the
      // reference's do (simulate.cpp) — the language's user-visible oracle, independent of everything the compiler does after
      // the transducer; `--sim sst` below is the compiled program on the HIP engine
      std::string input((std::istreambuf_iterator<char>(std::cin)), std::istreambuf_iterator<char>()), output;
      const int rc = simulateFST(ss.str(), srcname, o, sim == "backtrack", input, output);
      if (rc == 1) { std::cerr << "Reject\n"; return 1; }
      if (rc == 2) { std::cerr << "Malformed action program: non-singleton stack on termination\n"; return 1; }
      std::cout.write(output.data(), (std::streamsize)output.size());
      std::cout.flush();
      return 0;
    }
    if (simulate) {
      // Commands.hs:304-323 (`--sim sst`): stdin → the compiled pipeline → stdout on the HIP engine
      o.quiet = true;
      Compiled cs = compileSource(ss.str(), srcname, o);
      std::vector<uint8_t> sblob = writeBlob(cs.stages, cs.info);
      const char* td = getenv("TMPDIR");
      std::string path = std::string(td && *td ? td : "/tmp") + "/kexc-sim-XXXXXX";
      int fd = mkstemp(&path[0]);
      if (fd < 0) { std::cerr << "cannot create a temporary file in " << (td && *td ? td : "/tmp") << "\n"; return 1; }
      close(fd);
      int rc = 1;
      try {
        writeBinary(path, sblob, selfDir());
        pid_t pid = fork();
        if (pid == 0) {
          setenv("KX_SIM_MESSAGES", sim.c_str(), 1);
          const char* av[] = {path.c_str(), nullptr};
          execv(path.c_str(), const_cast<char* const*>(av));
          _exit(127);
        }
        int st = 0;
        if (pid > 0) { waitpid(pid, &st, 0); rc = WIFEXITED(st) ? WEXITSTATUS(st) : 1; }
      } catch (...) { unlink(path.c_str()); throw; }
      unlink(path.c_str());
      return rc;
    }
    Compiled c = compileSource(ss.str(), srcname, o);
    if (!o.quiet) {
      std::cout << (o.regex ? "Oracle SST states: " : "SST states: ");
      for (size_t i = 0; i < c.sst_states.size(); ++i) std::cout << (i ? ", " : "") << c.sst_states[i];
      std::cout << "\n";
    }
    (void)report;
    if (o.backend == "c") {
      std::string txt = emitC(c.stages, c.info);
      if (!o.srcout.empty()) { std::ofstream f(o.srcout); f << txt; }
      if (o.out.empty()) return 0;
      if (crtdir.empty()) { std::cerr << "--backend=c needs --crt-dir DIR (directory with crt.c)\n"; return 1; }
      // cc -O<copt> -xc -o <out> "-D FLAG_WORDALIGNED" -, source on stdin (C.hs:556-568).  fork/exec with an argument
      // vector: no shell ever sees --out, --crt-dir or --cc.
      const std::string optflag = "-O" + std::to_string(o.copt), incflag = "-I" + crtdir;
      const char* av[] = {o.cc.c_str(), optflag.c_str(), "-xc", "-o", o.out.c_str(), "-w", "-D FLAG_WORDALIGNED", incflag.c_str(), "-", nullptr};
      int fds[2];
      if (pipe(fds)) { std::cerr << "cannot run " << o.cc << "\n"; return 1; }
      pid_t pid = fork();
      if (pid < 0) { std::cerr << "cannot run " << o.cc << "\n"; return 1; }
      if (pid == 0) {
        dup2(fds[0], STDIN_FILENO); close(fds[0]); close(fds[1]);
        execvp(av[0], const_cast<char* const*>(av));
        std::cerr << "cannot run " << o.cc << "\n";
        _exit(127);
      }
      close(fds[0]);
      signal(SIGPIPE, SIG_IGN);
      for (size_t off = 0; off < txt.size();) {
        ssize_t w = write(fds[1], txt.data() + off, txt.size() - off);
        if (w <= 0) break;
        off += (size_t)w;
      }
      close(fds[1]);
      int rc = 0;
      waitpid(pid, &rc, 0);
      return WIFEXITED(rc) ? WEXITSTATUS(rc) : 1;
    }
    if (o.backend != "hip") { std::cerr << "unknown backend " << o.backend << "\n"; return 1; }
    std::vector<uint8_t> blob = writeBlob(c.stages, c.info);
    if (!o.blobout.empty()) { std::ofstream f(o.blobout, std::ios::binary); f.write((const char*)blob.data(), blob.size()); }
    if (!o.srcout.empty()) { std::ofstream f(o.srcout, std::ios::binary); f.write((const char*)blob.data(), blob.size()); }
    if (o.out.empty()) return 0;
    // BIN = kxrun ++ blob ++ libdir ++ trailer{blob_len u64, libdir_len u64, "KXRUNTRL"}
    writeBinary(o.out, blob, selfDir());
    return 0;
  } catch (const CompileError& e) {
    std::cerr << e.what() << "\n";
    return 1;
  } catch (const std::exception& e) {
    std::cerr << "kexc: " << e.what() << "\n";
    return 1;
  }
}
