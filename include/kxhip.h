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

/* tuning knobs (0 = default) */
typedef struct kx_config {
  uint32_t segment_bytes;  /* input bytes per lane; multiple of 64; 0 = one round of lanes (4-64 KiB)  */
  uint32_t block_threads;  /* workgroup size (power of two); default 512        */
  uint32_t collect_timing; /* record per-kernel HIP events into kx_stats        */
  uint32_t phase;          /* kx_run_fd: 0 = the whole pipeline; K = only phase K, stdin -> stdout (`BIN --phase K`, crt/crt.c:390-393,408-411) */
  uint64_t window_bytes;   /* kx_run_fd: input bytes resident at a time; 0 = 1 GiB (env KX_WINDOW_BYTES overrides;
                              KX_READ_THREADS = pread threads per chunk of a regular file, default 4; KX_FD_TRACE=1 prints where
                              the reader / compute / writer threads spent their time) */
} kx_config;

int kx_load(const void* blob, size_t blob_len, kx_program** prog);
/* Structural check of a blob without touching a device: every section inside the blob, every index inside its table,
 * the engine's size limits.  0 or KX_E_BLOB (kx_load performs the same checks). */
int kx_validate(const void* blob, size_t blob_len);
void kx_free(kx_program* prog);
const char* kx_last_error(void);
int kx_set_config(kx_program* prog, const kx_config* cfg);
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
 * from the form for a while): results never depend on which engine ran.  Environment: KX_DF=0 switches the delayed form off,
 * KX_DF=2 takes it whatever the share of undecided contexts, KX_DF_K=1|2 pins the delay (default: 1 where one symbol decides every
 * transition the start state reaches, else 2).
 * kx_df_describe needs no device: it builds the form from the blob (as kx_load does) and reports it; `image`, if not NULL,
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
  char reason[96];         /* why not, if available == 0 */
} kx_df_info;
int kx_df_describe(const void* blob, size_t blob_len, uint32_t stage, kx_df_info* info, void* image, size_t image_cap);
int kx_df_pending(const void* blob, size_t blob_len, uint32_t stage, uint32_t state, uint32_t slot, uint32_t* sst_state,
                  uint32_t* kinds, uint32_t* n_out);
/* of a loaded program: 0 the stage has no delayed form, 1 its next shard runs on it, 2 a shard gave it up (escape) and the stage is backing
 * off: after the k-th fall-back in a row the next 2^k - 1 shards (at most 63) go straight to the general engine, then the form is tried again */
int kx_stage_delayed_form(const kx_program* prog, uint32_t stage);

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
