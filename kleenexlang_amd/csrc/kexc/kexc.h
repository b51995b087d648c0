// kexc.h — shared types of the Kleenex → streaming-string-transducer compiler.
//
// This is a from-scratch C++ restatement of the *compile-time* half of the
// reference (Haskell) pipeline, needed because the hot path (the compiled
// state loop) only exists as compiler output.  Each stage cites the reference
// module whose behaviour it follows (paths relative to the reference root):
//   surface syntax      src/KMC/Kleenex/Parser.hs:154-222
//   regex literals      consumer side src/KMC/Kleenex/Desugaring.hs:73-118
//   desugaring          src/KMC/Kleenex/Desugaring.hs:125-207
//   transducer          src/KMC/SymbolicFST/Transducer.hs:57-107
//   determinization     src/KMC/Determinization.hs:38-257, src/KMC/TreeWriter.hs
//   optimize            src/KMC/SymbolicSST.hs:180-331
//   lowering            src/KMC/SSTCompiler.hs:67-156
// The back end does not print C (src/KMC/Program/Backends/C.hs) but a table
// blob for the HIP engine (see kxp_format.h); `--backend=c` prints C in the
// reference's shape for the CPU baseline.
#pragma once
#include <array>
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>
#include <unistd.h>
#include "../../../include/kxp_format.h"

namespace kexc {

struct CompileError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// ---------------------------------------------------------------- byte sets
// Predicates over Word8 (reference: KMC.RangeSet with sigma = Word8).
struct ByteSet {
  uint64_t w[4] = {0, 0, 0, 0};
  static ByteSet single(int b) { ByteSet s; s.add(b); return s; }
  static ByteSet range(int lo, int hi) { ByteSet s; for (int b = lo; b <= hi; ++b) s.add(b); return s; }
  static ByteSet universe() { ByteSet s; for (auto& x : s.w) x = ~0ull; return s; }
  void add(int b) { w[b >> 6] |= 1ull << (b & 63); }
  bool has(int b) const { return (w[b >> 6] >> (b & 63)) & 1; }
  bool empty() const { return !(w[0] | w[1] | w[2] | w[3]); }
  int size() const { int n = 0; for (auto x : w) n += __builtin_popcountll(x); return n; }
  int first() const { for (int b = 0; b < 256; ++b) if (has(b)) return b; return -1; }
  ByteSet operator&(const ByteSet& o) const { ByteSet s; for (int i = 0; i < 4; ++i) s.w[i] = w[i] & o.w[i]; return s; }
  ByteSet operator|(const ByteSet& o) const { ByteSet s; for (int i = 0; i < 4; ++i) s.w[i] = w[i] | o.w[i]; return s; }
  ByteSet operator~() const { ByteSet s; for (int i = 0; i < 4; ++i) s.w[i] = ~w[i]; return s; }
  ByteSet minus(const ByteSet& o) const { return *this & ~o; }
  bool subsetOf(const ByteSet& o) const { return minus(o).empty(); }
  bool operator==(const ByteSet& o) const { for (int i = 0; i < 4; ++i) if (w[i] != o.w[i]) return false; return true; }
  bool operator<(const ByteSet& o) const { for (int i = 0; i < 4; ++i) if (w[i] != o.w[i]) return w[i] < o.w[i]; return false; }
};

// ------------------------------------------------------------- surface AST
struct Regex;
using RegexP = std::shared_ptr<Regex>;
struct Regex {  // the AST shape Desugaring.hs:73-118 consumes
  enum Kind { One, Dot, Chr, Group, Concat, Branch, Class, Star, LazyStar, Plus, LazyPlus,
              Question, LazyQuestion, Range, Suppress } kind;
  int chr = 0;              // Chr: one *byte* (multi-byte UTF-8 chars become Concat of Chr)
  ByteSet cls;              // Class (already complemented when negative)
  RegexP a, b;              // operands
  int lo = 0; int hi = -1;  // Range: hi == -1 means unbounded
  bool has_hi = false;
};

struct Term;
using TermP = std::shared_ptr<Term>;
struct UpdAtom { bool is_reg; std::string s; };
struct Term {  // Kleenex/Syntax.hs:45-62
  enum Kind { Var, Constant, RE, Seq, Sum, Star, Plus, Question, Approx, Range, Suppress, One,
              UpdateReg, WriteReg, RedirectReg } kind;
  std::string name;          // Var ident / register name
  std::string bytes;         // Constant
  RegexP re;
  TermP a, b;
  int lo = -1, hi = -1;      // Range: -1 = absent
  int k = 0;                 // Approx
  std::vector<UpdAtom> upd;  // UpdateReg
};
struct Decl { std::string name; TermP term; };
struct Prog { std::vector<std::string> pipeline; std::vector<Decl> decls; };

Prog parseKleenex(const std::string& src, const std::string& srcname);

// ------------------------------------------------------- reduced grammar
struct RAct { int kind; /*0 byte,1 push,2 pop r,3 write r*/ int arg;
  bool operator<(const RAct& o) const { return kind != o.kind ? kind < o.kind : arg < o.arg; }
  bool operator==(const RAct& o) const { return kind == o.kind && arg == o.arg; } };
struct RTerm {  // Kleenex/Syntax.hs:95-102
  enum Kind { RConst, RRead, RSeq, RSum } kind;
  RAct c{0, 0};
  ByteSet pred; bool copy = false;
  std::vector<int> ids;
  bool operator<(const RTerm& o) const;
};
struct RProg { std::vector<int> pipeline; std::map<int, RTerm> decls; std::vector<std::string> regnames; };
RProg desugar(const Prog& p);
// regex flavour (`.re` / `.rx` / `--re`): the whole source is one regular expression (parseRegex, Kleenex/Parser.hs:204-206,
// 229-230) desugared with output enabled (desugarRegex, Desugaring.hs:231-239)
RProg parseRegexProgram(const std::string& src, const std::string& srcname);

// ---------------------------------------------------------------- the FST
struct FST {  // SymbolicFST.hs:49-55; states are ints after enumerateStates
  int nstates = 0, init = 0;
  std::vector<char> is_final;
  // per state: ordered ε-edges (output bytes, target) — or symbol edges (pred, copy?, target)
  struct Eps { std::string out; int to; };
  // tbl >= 0 (with copy): the symbol leaves through tables[tbl] — output = table[symbol], one byte (AppendTblI, IL.hs:44)
  struct Sym { ByteSet pred; bool copy; int to; int tbl = -1; };
  std::vector<std::array<uint8_t, 256>> tables;
  std::vector<std::vector<Eps>> eps;
  std::vector<std::vector<Sym>> sym;
};
bool stageHasActions(const RProg& rp, int start);
// tokens = false: direct mode, fails on register actions; true: actions and the byte 0xFF leave as escape tokens (automata.cpp)
FST constructTransducer(const RProg& rp, int start, bool tokens = false);
// the oracle of a transducer (SymbolicFST/OracleMachine.hs:47-61): symbol outputs dropped, every nondeterministic choice
// coded — one base-256 digit per choice, as the front end instantiates it (Frontend.hs:117: digit = Word8)
FST oracleTransducer(const FST& f);

// ---------------------------------------------------------------- the SST
struct Atom {  // SymbolicSST.hs:52-56
  enum Kind { VAR, CONST, FUNC } kind;
  int var = 0;          // VAR
  std::string bytes;    // CONST
  int func = 0;         // FUNC: 0 = copy next[0] (CopyArg), 1 = CopyConst []
  int sym = 0;          // FUNC: which symbol of a multi-symbol test it reads (`inj i`, Determinization.hs:224-226)
  int tbl = -1;         // FUNC 0: >= 0 = the symbol leaves through that table of the transducer (CodeArg / AppendTblI)
  bool operator==(const Atom& o) const { return kind == o.kind && var == o.var && bytes == o.bytes && func == o.func && sym == o.sym && tbl == o.tbl; }
};
using UpdateString = std::vector<Atom>;
struct PathStep { int parent; bool copy; std::string bytes; int tbl = -1; };  // per new leaf: origin leaf + appended output
struct SSTEdge {
  ByteSet pred; int to;
  std::map<int, UpdateString> upd;   // register update (parallel assignment)
  std::vector<PathStep> path;        // path form: one entry per leaf of the target state
};
struct SSTState {
  std::vector<SSTEdge> edges;
  bool is_final = false;
  UpdateString final_upd;   // VAR / CONST atoms
  int nleaves = 0;          // leaves of the (closed) path tree
  int final_leaf = -1;      // leaf whose path is emitted at end of input
};
struct SST {
  std::vector<SSTState> states; int init = 0; int nregs = 0;
  std::vector<std::string> init_path;  // output accumulated on each leaf of the initial closure
  std::vector<std::array<uint8_t, 256>> tables;   // the transducer's symbol tables (FST::tables)
};
SST determinize(const FST& f, size_t table_word_cap = 0);   // table_word_cap != 0: give up as soon as (states + 1) x classes must exceed it             // sstFromFST … singletonMode=True  (--la=false)

// The path form of the lookahead machine (sstFromFST … singletonMode=False, `--la=true`, the reference's default): a
// block tests WORDS of predicates — the single symbols of the coarsest partition plus the longest deterministic prefix
// of every leaf (prefixTests / ldp, SymbolicFST.hs:264-312) — and a leaf of the target extends a leaf of the source by
// one (copy?, constant) step per symbol of the word (consumeTreeMany, Determinization.hs:213-228).
struct WordStep { bool copy = false; std::string bytes; int tbl = -1; };
struct WordPath { int parent = 0; std::vector<WordStep> steps; };
struct WordEdge { std::vector<ByteSet> word; int to = 0; std::vector<WordPath> path; };   // path: one per leaf of `to`
struct WordState { std::vector<WordEdge> edges; int nleaves = 0, final_leaf = -1; };
struct WordSST { std::vector<WordState> states; int init = 0; std::vector<std::string> init_path; std::vector<std::array<uint8_t, 256>> tables; };
WordSST determinizeWords(const FST& f);
// The same function as a prioritized single-symbol transducer over (state, leaf) nodes: every (test, target leaf) becomes
// an alternative of its parent leaf's node — a chain of one symbol edge per symbol of the word — longer words first
// (the blocks try them first: a leaf killed by a short test dies within the longer word anyway), then in leaf order.
// determinize() of it is a table machine (one symbol per step) that writes what the lookahead machine writes.
FST leafGraph(const WordSST& w);
int optimizeSST(SST& s, int level);        // SymbolicSST.optimize; returns #iterations

// ------------------------------------------------------------ table form
struct MicroOp { uint8_t op; uint16_t dst; uint32_t arg; };  // op codes in kxp_format.h
struct StageTables {
  int nstates = 0, nclasses = 0, q0 = 0, nregs = 0;
  uint8_t cls[256];
  std::vector<uint16_t> delta;        // [nstates*nclasses], 0xFFFF = no transition
  std::vector<uint32_t> act;          // [nstates*nclasses] action id
  std::vector<uint32_t> final_act;    // [nstates], 0xFFFFFFFF = not final
  std::vector<std::vector<MicroOp>> actions;
  std::vector<std::string> consts;
  // path form
  int maxleaves = 0;
  std::vector<uint32_t> pback;        // [nstates*nclasses] backward-row id
  std::vector<uint8_t> nleaves, fin_leaf;
  std::vector<uint32_t> back;         // [nback*maxleaves]: parent | copy<<8 | pconst<<9 | (table+1)<<24 ; ~0u = dead
  std::vector<std::array<uint8_t, 256>> tables;   // symbol tables (KXP_OP_APPEND_TBL / the table field of a back entry)
  std::vector<std::string> pconsts;
  std::vector<uint32_t> init_const;   // [maxleaves] pconst id per leaf of q0
  // synchronising automaton over state subsets (all states → …)
  std::vector<uint32_t> sync_next;    // [nsync*nclasses]
  std::vector<uint32_t> sync_state;   // [nsync]: state id if singleton, 0xFFFFFFFE empty, 0xFFFFFFFF otherwise
  bool sync_complete = true;          // false if the subset construction hit its cap
  int act_regs = -1;                  // ≥ 0: the stage's output is a token stream for the action interpreter with that many registers
};
StageTables lower(const SST& s, const SST& path_src);
void buildSync(StageTables& t);            // fills sync_next / sync_state from delta
// BIN = kxrun ++ blob ++ libdir ++ trailer (the produced binary of `kexc compile --out BIN`); dir = where kxrun and libkxhip.so lie
void writeBinary(const std::string& out, const std::vector<uint8_t>& blob, const std::string& dir);
std::vector<uint8_t> writeBlob(const std::vector<StageTables>& stages, const std::string& info);
std::string emitC(const std::vector<StageTables>& stages, const std::string& info);

struct Options {
  int opt = 3; bool la = false; bool act = true; bool quiet = false;
  std::string out, srcout, cc = "cc", backend = "hip", blobout;
  int copt = 3;
  bool regex = false;          // regex flavour: the bit-coder (kexc.hs:46-48, compileCoder Commands.hs:246-275)
  int wordsize = 8;            // --wordsize: the run-time buffer unit (Options.hs:130-144); only 8 produces defined output, see main.cpp
};
struct Compiled { std::vector<StageTables> stages; std::string info; std::vector<int> sst_states; };
Compiled compileSource(const std::string& src, const std::string& srcname, const Options& o);
// the reference's FST simulators on the CPU (simulate.cpp): 0 accepted, 1 rejected, 2 malformed action program
int simulateFST(const std::string& src, const std::string& srcname, const Options& o, bool backtracking, const std::string& input, std::string& out);

}  // namespace kexc
