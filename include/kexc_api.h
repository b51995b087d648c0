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
 * Because no Haskell toolchain exists in this environment the front half
 * (Kleenex source → SST) is restated in C++ behind the same ABI, so the entry
 * point takes source text rather than a marshalled `Pipeline`.
 */
#ifndef KEXC_API_H
#define KEXC_API_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Compile Kleenex source text (direct mode, --la=false semantics) to a KXP blob.
 * opt_level = the reference's `--opt` (0..3, SymbolicSST.optimize).
 * Returns 0 and a malloc'd blob (release with kexc_free), or 1 with the
 * message available from kexc_last_error() (parse / well-formedness errors,
 * register actions in direct mode: Commands.hs:57-62,165-168). */
int kexc_compile(const char* source, size_t source_len, const char* source_name, int opt_level,
                 unsigned char** blob, size_t* blob_len);

/* `--backend=c`: the same program printed as C in the reference's generated
 * shape (C.hs:72-83,267-311,486-493), to be compiled together with a `crt.c`
 * runtime for the CPU baseline. */
int kexc_emit_c(const char* source, size_t source_len, const char* source_name, int opt_level,
                char** c_text, size_t* c_len);

/* Test support: the nondeterministic transducers (one per pipeline stage, after
 * constructTransducer, src/KMC/SymbolicFST/Transducer.hs:57-107) as JSON:
 * [{"nstates","init","final":[..],"eps":[[ [[out bytes],to], ..] per state],"sym":[[ [[[lo,hi],..],copy,to], ..] per state]}] */
int kexc_dump_fst(const char* source, size_t source_len, const char* source_name, char** json, size_t* json_len);

const char* kexc_last_error(void);
void kexc_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
