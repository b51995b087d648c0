/* kxhip.h — C ABI of the MI355X streaming-SST engine (libkxhip.so).
 *
 * This is the drop-in boundary for the reference's run-time hot path.  In the
 * reference the generated code talks to its runtime through the "program
 * interface" of crt/crt.c:101-105 (`init`, `match(phase)`) and the runtime
 * exports readnext/consume/outputconst/outputarray/output/append/appendarray/
 * concat/reset (crt/crt.c:107-324); the whole thing is driven by `main`/`run`
 * (crt/crt.c:356-467).  Here the seam sits one level up: a compiled program
 * (KXP blob, include/kxp_format.h — the table form of the IL `Pipeline`,
 * src/KMC/Program/IL.hs:69-90) is handed to an engine that owns the state
 * loop, the registers and the output buffer on the GPU.
 *
 *   reference                                   this ABI
 *   ---------                                   --------
 *   compileProgram … Pipeline (C.hs:529-540)    kx_load(blob)
 *   run(phase): init_outbuf, init, match,       kx_run_device / kx_run_host /
 *     flush_outbuf        (crt.c:356-364)         kx_run_fd
 *   main: -t timing, phase pipeline             kx_run_fd + kx_stats (kxrun.cpp
 *     (crt.c:372-467)                             is the `main`)
 *   fail<K>: "Match error at input symbol %zu"  return 1, stats->fail_pos
 *     exit(1)             (C.hs:79-81)
 *
 * Plain pointers and sizes only; no framework types.  Device pointers are HIP
 * device pointers on the current device; `stream` is a hipStream_t (NULL =
 * default stream).  One kx_program may be used by one host thread at a time.
 *
 * Return codes: 0 accepted; 1 match error (stats->fail_pos = number of input
 * symbols consumed before the failing state, stats->fail_stage = pipeline
 * stage; no output is produced — the reference itself loses the un-flushed
 * tail, crt/crt.c:217-227); <0 runtime error, message in kx_last_error().
 */
#ifndef KXHIP_H
#define KXHIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct kx_program kx_program;
typedef struct kx_shard kx_shard;

#define KX_OK 0
#define KX_MATCH_ERROR 1
#define KX_E_BLOB (-1)     /* malformed / unsupported program blob          */
#define KX_E_HIP (-2)      /* HIP runtime error                             */
#define KX_E_CAPACITY (-3) /* output buffer too small; *out_len = needed    */
#define KX_E_ARG (-4)
#define KX_E_IO (-5)

enum { KX_K_SYNC = 0, KX_K_FORWARD = 1, KX_K_HEAD = 2, KX_K_BACKLEN = 3, KX_K_RESOLVE = 4, KX_K_EMIT = 5, KX_NKERNELS = 6 };

typedef struct kx_stats {
  uint64_t fail_pos;            /* valid when the call returned 1                           */
  uint32_t fail_stage;
  uint32_t unsynced_segments;   /* segments whose start state had to be chained sequentially */
  uint64_t in_bytes, out_bytes;
  float kernel_ms[KX_NKERNELS]; /* HIP-event time of each kernel group, summed over the stages  */
  float total_ms;               /* first launch → last kernel done, all stages               */
  uint32_t emit_overflow_pieces; /* output stage: pieces that needed a second sweep (last stage run) */
} kx_stats;

/* tuning knobs (0 = default, in every field).  Round 6: what used to be environment variables of the library (KX_DF, KX_DF_K, KX_INL,
 * KX_JL, KX_EMIT_*, KX_NO_*, KX_FORCE_*, KX_ACT_*, KX_DEBUG_FLAGS) is here; the library itself reads only KX_DEBUG (chatter on stderr),
 * KX_FD_TRACE, KX_WINDOW_BYTES and KX_READ_THREADS (kx_run_fd's host side).  The produced binary (kxrun.cpp) and the Python test
 * binding (kleenexlang_amd/host.py) map the old variable names onto this struct, so scripts keep working.
 * LOAD-TIME fields decide how the tables are built: they take effect in kx_load_config; kx_set_config refuses to change them. */
typedef struct kx_config {
  uint32_t segment_bytes;  /* input bytes per lane; multiple of 64; 0 = one round of lanes (4-64 KiB)  */
  uint32_t block_threads;  /* workgroup size (power of two); default 512        */
  uint32_t collect_timing; /* record per-kernel HIP events into kx_stats        */
  uint32_t phase;          /* kx_run_fd: 0 = the whole pipeline; K = only phase K, stdin -> stdout (`BIN --phase K`, crt/crt.c:390-393,408-411) */
  uint64_t window_bytes;   /* kx_run_fd: input bytes resident at a time; 0 = 1 GiB (env KX_WINDOW_BYTES overrides;
                              KX_READ_THREADS = pread threads per chunk of a regular file, default 4; KX_FD_TRACE=1 prints where
                              the reader / compute / writer threads spent their time) */
  /* ---- which engine (load time) ---- */
  uint32_t delayed_form;   /* 0 auto (taken where at most a quarter of the start-reachable transitions are undecided), 1 never,
                              2 whatever that share                                                          [KX_DF=0 / 2] */
  uint32_t delay;          /* 0 auto (1 where one symbol decides everything the start state reaches, else 2), 1, 2   [KX_DF_K] */
  uint32_t merge_window;   /* merged constants (kx_delayed.h): 0 auto (the largest J <= 8 whose table image leaves k_dforward two
                              blocks per CU), else J = merge_window - 1 (1: none)                                   [KX_DF_J = J] */
  uint32_t inline_consts;  /* general engine, inline-constant entry layout: 0 auto, 1 off, 2 on                    [KX_INL=0 / 1] */
  uint32_t job_stride;     /* general engine, job-stride entry layout: 0 auto (a second image, chosen shard by shard from the
                              constants per piece), 1 off, 2 on (only that image)                [KX_JL_AUTO_OFF / KX_JL=0 / 1] */
  uint32_t disable;        /* KX_OFF_* bits                                                                             */
  uint32_t force;          /* KX_FORCE_* bits                                                                           */
  /* ---- the output stage's plan (run time) ---- */
  uint32_t emit_waves;     /* waves per CU of k_emit / k_demit: 0 auto (12), 4, 8, 12, 16 (16: delayed form only)   [KX_EMIT_WAVES] */
  uint32_t emit_half;      /* k_demit, a lane takes half a piece: 0 auto (where 64 pieces outgrow a wave's staging), 1 off, 2 on [KX_EMIT_HALF] */
  uint32_t emit_inplace;   /* k_emit, constants copied by the sweeping lane instead of jobs: 0 auto, 1 off, 2 on    [KX_EMIT_INPLACE] */
  uint32_t emit_staging;   /* k_emit, staging bytes per wave: 0 auto                                                 [KX_EMIT_STG] */
  uint32_t df_backoff;     /* after a shard left the delayed form: 0 = the stage skips 2^k - 1 shards after the k-th fall-back in a
                              row (at most 63), 1 = every shard tries the form again                                        */
  uint32_t debug_flags;    /* 64: per-phase shader-clock timeline of k_emit / k_demit (printed under KX_DEBUG)     [KX_DEBUG_FLAGS] */
  /* ---- the action post-pass (run time) ---- */
  uint32_t act_par_min;    /* bytes from which a token stream is replayed chunk-parallel: 0 = 1 MiB                 [KX_ACT_PAR_MIN] */
  uint32_t act_prefix3_min;/* blocks from which the safe-point prefix runs in three steps: 0 = 65536             [KX_ACT_PREFIX3_MIN] */
  uint32_t act_lanes;      /* one lane per chunk (token-dense streams): 0 auto, 1 off, 2 on                           [KX_ACT_LANES] */
  uint32_t act_chunk;      /* wanted chunk bytes: 0 auto (2048 / 4096)                                                 [KX_ACT_CHUNK] */
  uint32_t reserved[4];    /* must be 0 */
} kx_config;
#define KX_OFF_DIRECT 1u     /* load: entries of the plain layout name their action, not the constant's pool slot    [KX_NO_DIRECT] */
#define KX_OFF_PAIR 2u       /* load: no two-symbol table for k_forward                                                 [KX_NO_PAIR] */
#define KX_OFF_CMPX 4u       /* load: constants of at most 16 bytes are not stored by the v_cmpx sequences              [KX_NO_CMPX] */
#define KX_OFF_COOP 8u       /* run:  k_forward without cooperative line loads                                          [KX_NO_COOP] */
#define KX_OFF_SLOW 16u      /* load: no exact slow path in the delayed form's forward pass: a context that the delay does not decide
                                sends the whole shard to the general engine, as in round 5                              [KX_NO_SLOW] */
#define KX_FORCE_BIG 1u      /* load: tables stay in global memory whatever their size (the BIG instances; tests)     [KX_FORCE_BIG] */
#define KX_FORCE_TBLMODE 2u  /* load: per-entry symbol-table ids even where one table would do (tests)            [KX_FORCE_TBLMODE] */
#define KX_FORCE_ACT_SEQ 4u  /* run:  the action post-pass on one wave                                                  [KX_ACT_SEQ] */
#define KX_FORCE_SAME_DEVICE 8u /* kx_run_fd_sharded_cfg: every rank on the caller's current device (a one-GPU box)  [KX_SHARD_SAME_DEVICE] */

int kx_load(const void* blob, size_t blob_len, kx_program** prog);   /* = kx_load_config(blob, blob_len, NULL, prog) */
/* kx_load with the load-time fields of `cfg` (NULL: all defaults); its run-time fields become the program's configuration. */
int kx_load_config(const void* blob, size_t blob_len, const kx_config* cfg, kx_program** prog);
/* Structural check of a blob without touching a device: every section inside the blob, every index inside its table,
 * the engine's size limits.  0 or KX_E_BLOB (kx_load performs the same checks). */
int kx_validate(const void* blob, size_t blob_len);
void kx_free(kx_program* prog);
const char* kx_last_error(void);
int kx_set_config(kx_program* prog, const kx_config* cfg);   /* run-time fields; KX_E_ARG if a load-time field differs from what the program was loaded with */
uint32_t kx_num_stages(const kx_program* prog);
/* 1: the stage uses register actions (`r@t`, `!r`, `[r <- …]`).  Its transducer output is a token stream (kxp_format.h) that
 * kx_run_device / kx_run_host / kx_run_fd replay with the action post-pass before it leaves the stage; the replay is
 * sequential over the whole stream (registers are unbounded), so such a stage is not run through the kx_shard_* protocol
 * across GPUs (SURVEY §8e "pathological case": the boundary tuple would be O(N)). */
int kx_stage_has_actions(const kx_program* prog, uint32_t stage);

/* ---- the delayed form of a stage (round 5; kleenexlang_amd/csrc/engine/kx_delayed.h) -----------------------------------
 * Where every step's output is decided by at most K further input symbols the engine runs the stage as a forward transducer
 * with fixed delay K — a forward pass for the lengths and one fused walk that places the bytes; no backward pass.  A context
 * that K symbols do not decide is noticed at run time and the shard is redone by the general engine (and the stage backs off
 * from the form for a while): results never depend on which engine ran.  kx_config::delayed_form / delay / merge_window select it.
 * kx_df_describe needs no device: it builds the form from the blob (as kx_load does; the `_cfg` variants with the load-time fields of a
 * configuration, the plain ones with the defaults) and reports it; `image`, if not NULL,
 * receives up to image_cap bytes of the table image (class*8 u8[256] — the class index where a stage has more than 31 byte classes — | rows of C x {lo = handle of the next state's row,
 * hi = what the step writes: bit 0 no byte copied, bits 10-22 pool offset/16 of the constant, bit 23 a constant follows,
 * bits 24-30 bytes appended} | pool).  kx_df_pending: what slot j (0 = oldest) of product state `state` still owes, per leaf
 * of its SST state: copy | path-constant id << 1 (n_out = 1: the same for every leaf) — the host evaluates these at the end
 * leaf to write a shard's last K steps.  Returns 0, KX_E_BLOB, or KX_E_ARG (no such stage / state / slot). */
typedef struct kx_df_info {
  uint32_t available;      /* 1: the stage has a delayed form and the engine takes it */
  uint32_t delay;          /* K */
  uint32_t nstates;        /* product states (the dead and the escape row follow them in the image) */
  uint32_t nclasses;
  uint32_t image_bytes, off_pool, start_handle, dead_handle, escape_handle;
  uint32_t transitions, escapes;               /* over the whole table */
  uint32_t transitions_start, escapes_start;   /* over the part reachable from the program's start state */
  uint32_t merge_window;   /* J: steps a constant may wait to be written together with the next one (0: no merged constants) */
  char reason[96];         /* why not, if available == 0 */
} kx_df_info;
int kx_df_describe(const void* blob, size_t blob_len, uint32_t stage, kx_df_info* info, void* image, size_t image_cap);
int kx_df_pending(const void* blob, size_t blob_len, uint32_t stage, uint32_t state, uint32_t slot, uint32_t* sst_state,
                  uint32_t* kinds, uint32_t* n_out);
int kx_df_describe_cfg(const void* blob, size_t blob_len, uint32_t stage, const kx_config* cfg, kx_df_info* info, void* image, size_t image_cap);
int kx_df_pending_cfg(const void* blob, size_t blob_len, uint32_t stage, const kx_config* cfg, uint32_t state, uint32_t slot,
                      uint32_t* sst_state, uint32_t* kinds, uint32_t* n_out);
/* Merged constants (round 6): a constant may wait up to J = merge_window steps for the next one as long as no copied byte can come
 * between them, so that one job of the output stage places both (apache_log: 15 constants per line become 8).  A product state then
 * also holds the constants that are DUE; kx_df_deferred returns their bytes, oldest first (a shard's tail writes them in front of
 * what the pending steps append).  *n_out = their length (which may exceed cap; at most cap bytes are stored). */
int kx_df_deferred(const void* blob, size_t blob_len, uint32_t stage, const kx_config* cfg, uint32_t state, void* bytes, size_t cap, size_t* n_out);
/* The handle of (SST state q, nothing pending, nothing due): where a segment, a window or a shard may begin.  *handle = 0xFFFF where
 * the table does not hold it (the part reachable from the start state never visits q). */
int kx_df_start_of_state(const void* blob, size_t blob_len, uint32_t stage, const kx_config* cfg, uint32_t sst_state, uint32_t* handle);
/* of a loaded program: 0 the stage has no delayed form, 1 its next shard runs on it, 2 a shard gave it up (escape) and the stage is backing
 * off: after the k-th fall-back in a row the next 2^k - 1 shards (at most 63) go straight to the general engine, then the form is tried again
 * (kx_config::df_backoff = 1: at once); 3 a shard left the form in (nearly) every segment (at least one lane in 16, and 8 lanes): the
 * program's output hangs on unbounded lookahead and the stage stays on the general engine until kx_stage_reset_delayed_form */
int kx_stage_delayed_form(const kx_program* prog, uint32_t stage);
/* forget the back-off: the stage's next shard tries the delayed form again (no-op for a stage without one) */
void kx_stage_reset_delayed_form(kx_program* prog, uint32_t stage);

/* Whole program (all pipeline stages) over one device-resident input.
 * d_out may be NULL with cap 0 to query the exact output size (returned in
 * *out_len with KX_E_CAPACITY).  Blocks until the result is complete. */
int kx_run_device(kx_program* prog, const void* d_in, size_t n, void* d_out, size_t cap, size_t* out_len,
                  kx_stats* stats, void* stream);

/* Host-buffer convenience: H2D, run, D2H.  *out is malloc'd (free with kx_host_free). */
int kx_run_host(kx_program* prog, const void* in, size_t n, void** out, size_t* out_len, kx_stats* stats);
void kx_host_free(void* p);

/* stdin → stdout contract of the produced binary (crt/crt.c:294-312,107-136): reads in_fd to EOF,
 * writes the output to out_fd. */
int kx_run_fd(kx_program* prog, int in_fd, int out_fd, kx_stats* stats);

/* ---- sharded execution: one contiguous shard of the input per GPU -------------
 * Per stage and per rank:
 *   kx_shard_begin → kx_shard_forward → [exchange kx_fwd_summary] → kx_shard_fix_head
 *   → kx_shard_backward → [exchange kx_bwd_summary, last rank first] → kx_shard_resolve
 *   → [exchange out_len] → kx_shard_emit → kx_shard_end
 * The only cross-shard data are the two fixed-size summaries below (the chunk-boundary
 * hand-off); the data path needs no collective. */
#define KX_MAX_LEAVES 256
typedef struct kx_fwd_summary {
  uint32_t synced;      /* 1: end_state does not depend on the incoming state                  */
  uint32_t end_state;   /* state entering the byte after this shard (valid if synced or fixed) */
  uint64_t head_len;    /* leading bytes that still need the incoming state (0 on first shard)  */
  uint64_t fail_pos;    /* UINT64_MAX = no failure seen so far                                 */
} kx_fwd_summary;
typedef struct kx_bwd_summary {
  uint32_t constant;                     /* 1: start leaf is the same for every end leaf */
  uint32_t nleaves;                      /* leaves of the state at the shard end         */
  uint8_t start_leaf[KX_MAX_LEAVES];     /* start leaf of the shard, per end leaf        */
} kx_bwd_summary;

int kx_shard_begin(kx_program* prog, uint32_t stage, const void* d_in, size_t n, int is_first, int is_last,
                   void* stream, kx_shard** shard);
int kx_shard_forward(kx_shard* s, kx_fwd_summary* out);
int kx_shard_fix_head(kx_shard* s, uint32_t incoming_state, kx_fwd_summary* out);
int kx_shard_backward(kx_shard* s, kx_bwd_summary* out);
int kx_shard_resolve(kx_shard* s, uint32_t end_leaf, uint64_t* out_len);
int kx_shard_emit(kx_shard* s, void* d_out, size_t cap);
void kx_shard_stats(kx_shard* s, kx_stats* stats);
void kx_shard_end(kx_shard* s);

/* ---- the multi-GPU driver: the protocol above, run by the library itself ---------------------------------------
 * One rank per GPU (a process, or a thread of one process); rank r holds shard r of the input on its own current device.
 * kx_run_sharded runs every pipeline stage over the rank's shard and takes part in the boundary hand-off — four
 * all-gathers of fixed-size records per stage (40, 40, 272 and 16 bytes per rank; one more of 16 bytes behind the last stage's emit), nothing else crosses ranks — through
 * the all-gather it is given:
 *   kx_comm_*   RCCL: `ncclAllGather` on the communicator's own device buffers and stream (xGMI between the GPUs of a node;
 *               librccl is dlopen'ed).  Rank 0 makes the 128-byte id (kx_comm_unique_id) and the launcher hands it to every
 *               rank (ncclGetUniqueId / ncclCommInitRank's contract); every rank then calls kx_comm_init on its device.
 *   kx_group_*  the threads of one process (the produced binary's `--gpus N`): exchange through host memory.
 * With world = 1 no exchange takes place (ag may be NULL).  Returns as kx_run_device: 0, 1 (match error: res->stats.fail_pos
 * is the GLOBAL position, the same on every rank), or < 0 (KX_E_CAPACITY: res->out_len = bytes this rank needs).
 * The rank's output slice [out_offset, out_offset + out_len) of the total_out output bytes stays on its device. */
typedef int (*kx_allgather_fn)(void* ctx, const void* send, void* recv, size_t bytes);   /* recv holds world * bytes; 0 = ok */
typedef struct kx_sharded_result {
  uint64_t out_len, out_offset, total_out;
  float boundary_ms;   /* wall time this rank spent inside the all-gathers (waiting for the slowest rank included) */
  kx_stats stats;      /* kernel times summed over the stages; fail_pos / fail_stage on a match error */
} kx_sharded_result;
/* (collective return code: every record of the exchanges carries the sender's status and one more status word follows the last
 *  stage's emit, so a local failure — out of memory,
 *  a HIP error — ends the call on EVERY rank instead of leaving the others inside an all-gather) */
int kx_run_sharded(kx_program* prog, int rank, int world, kx_allgather_fn ag, void* ag_ctx, const void* d_in, size_t n,
                   void* d_out, size_t cap, kx_sharded_result* res, void* stream);

/* The produced binary's `--gpus N`: stdin must be a regular file; it is cut into N contiguous shards (4 KiB multiples), one
 * thread and one program instance per GPU, hand-off through host memory (kx_group_*); the output is written at each rank's
 * offset (regular file) or in rank order (pipe).  Same return codes and match-error position as kx_run_fd.
 * Limits: every rank holds its WHOLE shard and its output in device memory (no windows, unlike kx_run_fd): an input beyond
 * about N x 100 GB fails with "cannot place the shard on the device".  stats: kernel times of the slowest rank, counts summed.
 * The return code is collective (kx_run_sharded): a rank that fails locally says so in the next exchange and every rank
 * returns — its own code and message, or KX_E_IO naming the failing rank. */
int kx_run_fd_sharded(const void* blob, size_t blob_len, int ngpus, int in_fd, int out_fd, kx_stats* stats);
int kx_run_fd_sharded_cfg(const void* blob, size_t blob_len, const kx_config* cfg, int ngpus, int in_fd, int out_fd, kx_stats* stats);

typedef struct kx_comm kx_comm;
int kx_comm_unique_id(void* id128);
int kx_comm_init(kx_comm** comm, int rank, int world, const void* id128);
void kx_comm_free(kx_comm* comm);
int kx_comm_allgather(void* comm, const void* send, void* recv, size_t bytes);            /* a kx_allgather_fn */

typedef struct kx_group kx_group;
typedef struct kx_group_member kx_group_member;
kx_group* kx_group_create(int world);
void kx_group_free(kx_group* g);
kx_group_member* kx_group_join(kx_group* g, int rank);
void kx_group_leave(kx_group_member* m);
void kx_group_abort(kx_group* g);   /* a member that fails outside the protocol wakes the others: their all-gathers return an error */
int kx_group_allgather(void* member, const void* send, void* recv, size_t bytes);         /* a kx_allgather_fn */

#ifdef __cplusplus
}
#endif
#endif
