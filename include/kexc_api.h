/* kexc_api.h — C ABI of the compiler library (libkexc.so).
 *
 * Replaces, for the hot path, the seam the reference exposes between its
 * front end and its code generator:
 *   compileProgram :: CType -> Int -> (String -> m ()) -> Pipeline -> Maybe String
 *                  -> FilePath -> Maybe FilePath -> Maybe FilePath -> Bool -> m ExitCode
 *   (src/KMC/Program/Backends/C.hs:529-540; sole callers
 *    src/KMC/Frontend/Commands.hs:202-210,236-244,267-275).
 * There the `Pipeline` of IL programs is printed as C and piped to `cc`; here
 * the same information leaves the compiler as a KXP table blob
 * (include/kxp_format.h) that the HIP engine (include/kxhip.h) loads.
 * Two entries: kexc_emit_pipeline is compileProgram itself — a front end that has its own SSTs (the reference's
 * Haskell one) marshals its `Pipeline` into tables and calls it with compileProgram's argument list, one to one;
 * kexc_compile puts the restated front half (Kleenex source → SST, C++: no Haskell toolchain exists in this
 * environment) in front of the same back end.
 */
#ifndef KEXC_API_H
#define KEXC_API_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* One IL `Program` (src/KMC/Program/IL.hs:69-88) of a `--la=false` pipeline in table form, plus the path-tree annotation.
 *
 * With `--la=false` every block is `NextI 1 1 fallback ; [IfI (avail>=1 && pred_j(next[0])) (updates_j ++ [ConsumeI 1,
 * GotoI s_j])]_j ; FailI` with pairwise disjoint predicates (SSTCompiler.hs:138-156), i.e. a table:
 *   class_of      coarsest partition of all test predicates (byte -> class)
 *   delta/action  per (block, class): GotoI target (0xFFFF = FailI) and the register updates, as an index into `actions`
 *   final_action  the NextI fallback: updates then AcceptI; 0xFFFFFFFF = FailI
 *   actions       ordered micro-ops {op << 24 | dst register, arg}: 0 ResetI dst; 1 AppendI dst, constant arg;
 *                 2 AppendSymI dst (the current input byte); 3 ConcatI dst, register arg; 4 AppendTblI dst, table arg
 *                 (below)      (IL.hs:40-47; register 0 = progStreamBuffer, the output)
 *   constants     progConstants as offsets into one pool (const_off[nconsts+1])
 * The annotation is what the determinizer knows when it builds a transition (Determinization.hs:165-177, TreeWriter.hs)
 * and the IL no longer says: which leaf of the SOURCE state's path tree each leaf of the TARGET state extends, and by
 * what.  It lets the engine replace the registers by a backward pass:
 *   nleaves[q], final_leaf[q] (0xFF = not final), maxleaves
 *   back_row[block*nclasses+class] -> row of `back`; back[row*maxleaves + leaf] = parent leaf | copies input byte << 8 |
 *                 path constant << 9 (0xFFFFFFFF = leaf does not exist); path constants as offsets into a second pool
 *   init_const[leaf] path constant emitted by the initial closure on each leaf of the start state */
typedef struct kexc_il_program {
  uint32_t nstates, nclasses, init_state, nregs;
  const uint8_t* class_of;        /* [256] */
  const uint16_t* delta;          /* [nstates*nclasses] */
  const uint32_t* action;         /* [nstates*nclasses] */
  const uint32_t* final_action;   /* [nstates] */
  uint32_t nactions; const uint32_t* action_off; const uint32_t* ops;   /* action_off[nactions+1]; ops[2*nops] */
  uint32_t nconsts; const uint32_t* const_off; const uint8_t* const_pool;
  uint32_t maxleaves, nback;
  const uint32_t* back_row;       /* [nstates*nclasses] */
  const uint8_t* nleaves; const uint8_t* final_leaf;   /* [nstates] */
  const uint32_t* back;           /* [nback*maxleaves] */
  uint32_t npconsts; const uint32_t* pconst_off; const uint8_t* pconst_pool;
  const uint32_t* init_const;     /* [maxleaves] */
  /* register actions: non-zero = the program's output is a token stream (escape byte 0xFF: FF FF = byte FF, FF 00 Push,
   * FF 01 r Pop r, FF 02 r Write r; kxp_format.h) to be replayed by the action interpreter with `action_regs` registers */
  uint32_t has_actions, action_regs;
  /* AppendTblI (src/KMC/Program/IL.hs:44,84 progTables; built by SSTCompiler/Classes.hs:102-125 for the coder's
   * `CodeArg p`; printed by C.hs:421-430): micro-op 4 = AppendTblI dst, table arg — appends tables[arg][next[0]], a string
   * of tbl_width[arg] digits (bytes: progOutBits = 8).  Table t starts at tbl_data + 256 * (tbl_width[0] + … +
   * tbl_width[t-1]); the entry of symbol s lies at + s * tbl_width[t].  In the annotation, back_table[row*maxleaves + leaf]
   * names the table whose entry for the symbol read is appended on that path step BEFORE the path constant (0xFFFFFFFF =
   * none; not together with the copy bit); NULL when ntables == 0.  Tables of ONE-byte entries (every table the coder's
   * CodeArg makes: |p| <= 256) stay table atoms — kxp_format.h: KXP_OP_APPEND_TBL, the table field of a back entry — and
   * the engine's output stage looks the symbol up.  Tables of wider entries (or KEXC_LOWER_TABLES=1) are
   * written out: the byte classes are refined until every table is constant on each class and op 4 becomes op 1. */
  uint32_t ntables; const uint32_t* tbl_width; const uint8_t* tbl_data; const uint32_t* back_table;
  /* Block form — `--la=true`, the reference's default.  A block then tests WORDS: `IfI (avail>=n && p_0(next[0]) && … &&
   * p_{n-1}(next[n-1])) (updates ++ [ConsumeI n, GotoI s])`, nested by common prefix, longer words first (prefixTests /
   * ldp, SymbolicFST.hs:264-312; the kv-tree of tests, SSTCompiler.hs:96-135) — not a row of a (state, class) table.
   * ntests != 0 selects this form: the tests come with their annotation and the tables are built HERE from the
   * annotation (one symbol per step again, writing what the word machine writes; kexc.h: leafGraph).  Read then:
   * nstates, init_state, maxleaves, nleaves, final_leaf, the path-constant pool, init_const, has_actions/action_regs,
   * and
   *   test_block[t], test_target[t], test_len[t]   the block, its GotoI target, n of ConsumeI n (1..255)
   *   test_preds     32 bytes per symbol of every test (tests in order, symbols in order): bit b of the 256-bit set,
   *                  least significant byte first = byte b passes
   *   test_back      one row of maxleaves entries per symbol of every test, in the same order: entry [leaf of the target]
   *                  of the row of symbol i = parent leaf in the block's tree (row 0 only) | symbol i is copied << 8 |
   *                  path constant appended after symbol i << 9
   * class_of … back and the register-form arrays may be NULL (the blocks' register updates state the same function a
   * second time and are not read); AppendTblI does not occur (tables belong to the coder, whose tests — ranges split to
   * single symbols — are marshalled like any others).  Within a block, tests of equal length must be disjoint. */
  uint32_t ntests; const uint32_t* test_block; const uint32_t* test_target; const uint32_t* test_len;
  const uint8_t* test_preds; const uint32_t* test_back;
} kexc_il_program;

/* What a front end has to do to hand its pipeline over (INTEGRATION.md §1 has the Haskell): read the annotation off the finished
 * SST — its states are their own path trees and every transition's registers expand to the parent leaf's node positions followed
 * by what the step appends (no change to the determinizer) — and marshal it: with `--la=false` (`sstFromFST fst True`,
 * Determinization.hs:233-257) as the (state, class) tables above, with the default `--la=true` as the block form (ntests).
 * The two machines write the same bytes (Tests/Regression.hs:45-53); so do the tables built from either.
 *
 * `type Pipeline = Either [Program] [(Program, Program)]` (IL.hs:90): Left = direct / coder pipelines, one phase per
 * program; Right = (oracle, action) pairs, programs[2i] and programs[2i+1] — accepted by the type, refused: the action
 * program (actionToSST, src/KMC/SymbolicSST/ActionSST.hs:47-104) is a register machine whose registers hold data (the
 * contents of `r@t`), not the pending output of undecided paths, so it has no path form.  Register actions reach the
 * engine the other way: ONE program per stage whose output carries the actions in band (has_actions below) and the
 * action post-pass on the device (`kexc_compile` does this for Kleenex source). */
/* program_size = sizeof(kexc_il_program) as the CALLER was compiled: programs[] is an array of records of that size, and the record
 * has grown from round to round (symbol tables, the block form) — a caller built against an older header is refused with a
 * message instead of being read with the wrong stride. */
typedef struct kexc_pipeline { int is_oracle_action; uint32_t nprograms; const kexc_il_program* programs; uint32_t program_size; } kexc_pipeline;

/* compileProgram (src/KMC/Program/Backends/C.hs:529-540), argument for argument:
 *   CType buffer unit        -> buffer_unit_bits (8; 16/32/64 are refused: `--wordsize` other than 8 is not built)
 *   Int cc -O level          -> cc_opt_level     (no C compiler runs for the HIP back end: accepted, unused)
 *   String -> m () info      -> info(line, ctx)  (called with the progress line the reference prints, C.hs:553)
 *   Pipeline                 -> pipeline
 *   Maybe String             -> env_info         (text that `BIN -i` prints; NULL = Nothing)
 *   FilePath cc              -> cc               (accepted, unused)
 *   Maybe FilePath binary    -> out_path         (NULL = Nothing: like the reference, only srcout is written then)
 *   Maybe FilePath source    -> srcout_path      (the KXP blob is this back end's "source")
 *   Bool word alignment      -> word_alignment   (accepted, unused: the device output is a byte stream)
 * Returns the exit code (0 = ExitSuccess); on failure the message is in kexc_last_error(). */
/* The exported symbol carries the version of the records it reads: kexc_pipeline grew a field (program_size) in round 4, and a
 * caller built against the older, shorter struct must not reach code that reads the longer one — it now fails to link / to look
 * the symbol up instead.  Source keeps the reference's name through the macro. */
#define kexc_emit_pipeline kexc_emit_pipeline_v2
int kexc_emit_pipeline_v2(int buffer_unit_bits, int cc_opt_level, void (*info)(const char* line, void* ctx), void* info_ctx,
                          const kexc_pipeline* pipeline, const char* env_info, const char* cc, const char* out_path,
                          const char* srcout_path, int word_alignment);

/* Compile Kleenex source text (direct mode, --la=false semantics) to a KXP blob.
 * opt_level = the reference's `--opt` (0..3, SymbolicSST.optimize).
 * Returns 0 and a malloc'd blob (release with kexc_free), or 1 with the
 * message available from kexc_last_error() (parse / well-formedness errors,
 * register actions in direct mode: Commands.hs:57-62,165-168). */
int kexc_compile(const char* source, size_t source_len, const char* source_name, int opt_level,
                 unsigned char** blob, size_t* blob_len);

/* kexc_compile with the flags spelled out.  lookahead = `--la`: non-zero builds the lookahead machine the way the reference does
 * (word tests: prefixTests, SymbolicFST.hs:296-312; kills and consumeTreeMany, Determinization.hs:213-257) and then the tables
 * from its path form, exactly as kexc_emit_pipeline does for a block-form program.  regex: non-zero = kexc_compile_regex. */
int kexc_compile_flags(const char* source, size_t source_len, const char* source_name, int opt_level, int lookahead, int regex,
                       unsigned char** blob, size_t* blob_len);

/* Test support: the lookahead machine of every stage in path form, JSON:
 * [{"init","action_regs" (-1: none),"init_path":[[bytes]..],"states":[{"nleaves","final_leaf","edges":[{"to","word":[[32 bytes]..],
 *   "path":[{"parent","steps":[[copy,[bytes]]..]}..]}..]}..]}] */
int kexc_dump_words(const char* source, size_t source_len, const char* source_name, int regex, char** json, size_t* json_len);

/* `--backend=c`: the same program printed as C in the reference's generated
 * shape (C.hs:72-83,267-311,486-493), to be compiled together with a `crt.c`
 * runtime for the CPU baseline. */
int kexc_emit_c(const char* source, size_t source_len, const char* source_name, int opt_level,
                char** c_text, size_t* c_len);

/* Test support: the nondeterministic transducers (one per pipeline stage, after
 * constructTransducer, src/KMC/SymbolicFST/Transducer.hs:57-107) as JSON:
 * [{"nstates","init","final":[..],"eps":[[ [[out bytes],to], ..] per state],"sym":[[ [[[lo,hi],..],copy,to], ..] per state]}] */
int kexc_dump_fst(const char* source, size_t source_len, const char* source_name, char** json, size_t* json_len);

/* Regex flavour — what `kexc compile FILE.re|FILE.rx` / `kexc compile --re EXPR` builds (src/kexc.hs:46-48;
 * createProgram's RegexFlavor branch, generateOracleSSTs, compileCoder: src/KMC/Frontend/Commands.hs:69-79,117-136,
 * 246-275): the program that reads a string matching the (anchored) regular expression and writes the code of its greedy
 * parse — one base-256 digit for every choice the parse makes (the index of the ε-alternative taken, the index of the
 * symbol within a predicate with more than one member; src/KMC/SymbolicFST/OracleMachine.hs:47-61 with digit = Word8,
 * src/KMC/Frontend.hs:117).  Same blob format, same engine. */
int kexc_compile_regex(const char* regex, size_t regex_len, const char* source_name, int opt_level,
                       unsigned char** blob, size_t* blob_len);

/* Test support: the regex's transducer (oracle = 0) or its oracle machine (oracle = 1), JSON as kexc_dump_fst. */
int kexc_dump_regex_fst(const char* regex, size_t regex_len, int oracle, char** json, size_t* json_len);

const char* kexc_last_error(void);
void kexc_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
