// kx_sharded.cpp — the multi-GPU driver behind include/kxhip.h (`kx_run_sharded`): one contiguous shard of the input per
// GPU, one rank (process or thread) per GPU, and nothing but the chunk-boundary hand-off between them.
//
// The reference has no counterpart: its binary is one process per phase reading stdin (crt/crt.c:356-467).  What makes
// sharding possible is that an SST transition is an element of an associative monoid (composeRegisterUpdate,
// src/KMC/SymbolicSST.hs:122-136); in the path form that monoid element of a whole shard is tiny — the state it ends in
// (or that this depends on the state it starts in) and, backward, the leaf it starts in per leaf it ends in.
//
// Per pipeline stage a rank runs the kx_shard_* phases on its own shard and takes part in four all-gathers of fixed-size
// records (40, 40, 272 and 16 bytes per rank): states forward (before and after the shard heads are fixed), leaves backward,
// output sizes.  The all-gather is a callback:
//   kx_comm_*   RCCL (`ncclAllGather` on small device buffers of its own communicator; librccl is dlopen'ed so that the
//               engine library has no link-time dependency on it) — one process per GPU, xGMI between them;
//   kx_group_*  threads of one process (the produced binary's `--gpus N`), exchange through host memory.
// With one rank no exchange happens at all.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <sys/stat.h>
#include <unistd.h>

#include "../../../include/kxhip.h"

namespace {

extern "C" void kx_internal_set_error(const char* m);   // (kx_engine.hip: kx_last_error's thread-local message)
int sErr(int code, const std::string& m) { kx_internal_set_error(m.c_str()); return code; }

// ---------------------------------------------------------------- RCCL through dlopen
struct NcclId { char internal[128]; };
struct Rccl {
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // the process may already hold an RCCL (PyTorch ships one, built against the HIP runtime it also ships): use that very
    // library — by symbol if it is globally visible, else by the path it is mapped from — and only otherwise the system's
    void* h = dlsym(RTLD_DEFAULT, "ncclAllGather") ? RTLD_DEFAULT : nullptr;
    if (!h) {
      if (FILE* f = fopen("/proc/self/maps", "r")) {
        char line[4096];
        while (!h && fgets(line, sizeof line, f)) {
          char* path = strchr(line, '/');
          if (!path) continue;
          path[strcspn(path, "\n")] = 0;
          const char* base = strrchr(path, '/');
          if (base && strncmp(base + 1, "librccl.so", 10) == 0) h = dlopen(path, RTLD_NOW | RTLD_NOLOAD);
        }
        fclose(f);
      }
    }
    if (!h) for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return;
    r.GetUniqueId = (int (*)(NcclId*))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(void**, int, NcclId, int))dlsym(h, "ncclCommInitRank");
    r.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
    r.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    r.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.AllGather && r.CommDestroy;
  });
  return r;
}
constexpr int NCCL_UINT8 = 1;
constexpr size_t MSG_MAX = 512;   // largest per-rank record of the protocol (272 bytes)

}  // namespace

struct kx_comm {
  void* nccl = nullptr; int rank = 0, world = 1;
  uint8_t *send = nullptr, *recv = nullptr;       // pinned host staging
  uint8_t *d_send = nullptr, *d_recv = nullptr;   // device buffers: what RCCL is handed (the documented kind of buffer)
  hipStream_t stream = nullptr;
};

struct kx_group {   // ranks = threads of one process
  int world;
  std::mutex m; std::condition_variable cv;
  int arrived = 0; uint64_t gen = 0; bool aborted = false;
  std::vector<uint8_t> buf;
  explicit kx_group(int w) : world(w), buf((size_t)w * MSG_MAX) {}
  bool barrier() {   // false: a member gave up (its error would otherwise leave the others waiting for ever)
    std::unique_lock<std::mutex> lk(m);
    const uint64_t g = gen;
    if (aborted) return false;
    if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g || aborted; });
    return !aborted;
  }
  void abort() { std::lock_guard<std::mutex> lk(m); aborted = true; cv.notify_all(); }
};
struct kx_group_member { kx_group* g; int rank; };

extern "C" {

int kx_comm_unique_id(void* id128) {
  if (!id128) return sErr(KX_E_ARG, "null argument");
  if (!rccl().ok) return sErr(KX_E_HIP, "RCCL (librccl.so) is not available");
  NcclId id;
  int rc = rccl().GetUniqueId(&id);
  if (rc) return sErr(KX_E_HIP, std::string("ncclGetUniqueId: ") + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "failed"));
  memcpy(id128, &id, 128);
  return 0;
}

int kx_comm_init(kx_comm** out, int rank, int world, const void* id128) {
  if (!out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id128)) return sErr(KX_E_ARG, "bad communicator arguments");
  auto* c = new kx_comm;
  c->rank = rank; c->world = world;
  if (world > 1) {
    if (!rccl().ok) { delete c; return sErr(KX_E_HIP, "RCCL (librccl.so) is not available"); }
    NcclId id; memcpy(&id, id128, 128);
    int rc = rccl().CommInitRank(&c->nccl, world, id, rank);
    if (rc) { delete c; return sErr(KX_E_HIP, std::string("ncclCommInitRank: ") + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "failed")); }
    if (hipHostMalloc((void**)&c->send, MSG_MAX, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&c->recv, MSG_MAX * (size_t)world, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&c->d_send, MSG_MAX) != hipSuccess || hipMalloc((void**)&c->d_recv, MSG_MAX * (size_t)world) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c; return sErr(KX_E_HIP, "cannot allocate the communicator's buffers");
    }
  }
  *out = c;
  return 0;
}

void kx_comm_free(kx_comm* c) {
  if (!c) return;
  if (c->nccl) rccl().CommDestroy(c->nccl);
  if (c->send) (void)hipHostFree(c->send);
  if (c->recv) (void)hipHostFree(c->recv);
  if (c->d_send) (void)hipFree(c->d_send);
  if (c->d_recv) (void)hipFree(c->d_recv);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// kx_allgather_fn over RCCL: ctx = kx_comm*
int kx_comm_allgather(void* ctx, const void* send, void* recv, size_t bytes) {
  auto* c = (kx_comm*)ctx;
  if (!c || bytes > MSG_MAX) return sErr(KX_E_ARG, "bad all-gather arguments");
  if (c->world == 1) { memcpy(recv, send, bytes); return 0; }
  memcpy(c->send, send, bytes);
  if (hipMemcpyAsync(c->d_send, c->send, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return sErr(KX_E_HIP, "hipMemcpyAsync failed");
  int rc = rccl().AllGather(c->d_send, c->d_recv, bytes, NCCL_UINT8, c->nccl, c->stream);
  if (rc) return sErr(KX_E_HIP, std::string("ncclAllGather: ") + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "failed"));
  if (hipMemcpyAsync(c->recv, c->d_recv, bytes * (size_t)c->world, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return sErr(KX_E_HIP, "hipMemcpyAsync failed");
  if (hipStreamSynchronize(c->stream) != hipSuccess) return sErr(KX_E_HIP, "hipStreamSynchronize after ncclAllGather failed");
  memcpy(recv, c->recv, bytes * (size_t)c->world);
  return 0;
}

kx_group* kx_group_create(int world) { return world >= 1 ? new kx_group(world) : nullptr; }
void kx_group_free(kx_group* g) { delete g; }
kx_group_member* kx_group_join(kx_group* g, int rank) { return g && rank >= 0 && rank < g->world ? new kx_group_member{g, rank} : nullptr; }
void kx_group_leave(kx_group_member* m) { delete m; }

// kx_allgather_fn among the threads of a group: ctx = kx_group_member*
int kx_group_allgather(void* ctx, const void* send, void* recv, size_t bytes) {
  auto* mb = (kx_group_member*)ctx;
  if (!mb || bytes > MSG_MAX) return sErr(KX_E_ARG, "bad all-gather arguments");
  kx_group* g = mb->g;
  memcpy(g->buf.data() + (size_t)mb->rank * bytes, send, bytes);
  if (!g->barrier()) return sErr(KX_E_IO, "another rank of the group failed");
  memcpy(recv, g->buf.data(), bytes * (size_t)g->world);
  if (!g->barrier()) return sErr(KX_E_IO, "another rank of the group failed");
  return 0;
}
void kx_group_abort(kx_group* g) { if (g) g->abort(); }

// ------------------------------------------------------------------------------------ the driver
// records of the three exchanges.  Every record carries the sender's STATUS: a rank whose local step failed (out of memory, a HIP
// error, a bad leaf) still takes part in the next exchange and says so there, so that every rank leaves the collective with an
// error instead of waiting in an all-gather for a peer that has already returned.  The return code is therefore
// collective: the failing rank returns its own code and message, the others KX_E_IO naming the lowest failing rank.
struct FwdMsg { uint32_t synced, end_state; uint64_t head_len, fail_pos, n; uint32_t fixed, status; };   // 40 bytes
struct BwdMsg { uint32_t constant, nleaves; uint8_t start_leaf[KX_MAX_LEAVES]; uint32_t status, pad; };  // 272 bytes
struct LenMsg { uint64_t len; uint32_t status, pad; };                                                   // 16 bytes

static int runSharded(kx_program* p, int rank, int world, kx_allgather_fn ag, void* ag_ctx, const void* d_in, size_t n,
                      void* d_out, size_t cap, kx_sharded_result* res, void* stream, void** res_alloc) {
  if (!p || !res || world < 1 || rank < 0 || rank >= world || (world > 1 && !ag)) return sErr(KX_E_ARG, "bad arguments");
  memset(res, 0, sizeof *res);
  res->stats.fail_pos = UINT64_MAX;
  const uint32_t ns = kx_num_stages(p);
  for (uint32_t st = 0; st < ns; ++st)
    if (kx_stage_has_actions(p, st)) return sErr(KX_E_ARG, "a stage with register actions cannot be sharded (kx_stage_has_actions)");
  double boundary = 0;
  auto gather = [&](const void* send, void* recv, size_t bytes) -> int {
    if (world == 1) { memcpy(recv, send, bytes); return 0; }
    const auto t0 = std::chrono::steady_clock::now();
    int rc = ag(ag_ctx, send, recv, bytes);
    boundary += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
  };
  const void* cur = d_in; size_t curn = n;
  void* stagebuf[2] = {nullptr, nullptr};
  struct FreeBufs { void** b; ~FreeBufs() { for (int i = 0; i < 2; ++i) if (b[i]) (void)hipFree(b[i]); } } free_bufs{stagebuf};
  std::vector<FwdMsg> fwd(world);
  std::vector<BwdMsg> bwd(world);
  std::vector<LenMsg> lens(world);
  int err = 0;                 // this rank's first local failure (its message stays in the thread's error slot)
  std::string err_msg;
  auto local = [&](int rc) { if (rc && !err) { err = rc; err_msg = kx_last_error(); } return rc; };
  // after an exchange: has any rank failed?  (every rank sees the same records, so every rank takes the same way out)
  auto collective = [&](auto& recs) -> int {
    for (int r = 0; r < world; ++r)
      if (recs[r].status) {
        res->boundary_ms = (float)boundary;
        if (r == rank || err) return sErr(err ? err : KX_E_IO, err_msg.empty() ? "local failure" : err_msg);
        return sErr(KX_E_IO, "rank " + std::to_string(r) + " of the sharded run failed (its own return code says why)");
      }
    return 0;
  };
  int rc = 0;
  for (uint32_t st = 0; st < ns; ++st) {
    kx_shard* s = nullptr;
    if (!err) local(kx_shard_begin(p, st, cur, curn, rank == 0, rank == world - 1, stream, &s));   // (not on the half-written buffer of a stage whose emit failed)
    struct EndShard { kx_shard* s; kx_stats* acc; ~EndShard() {
      if (!s) return;
      kx_stats ss; kx_shard_stats(s, &ss);
      for (int i = 0; i < KX_NKERNELS; ++i) { acc->kernel_ms[i] += ss.kernel_ms[i]; acc->total_ms += ss.kernel_ms[i]; }
      acc->unsynced_segments += ss.unsynced_segments;
      kx_shard_end(s); } } end_shard{s, &res->stats};
    // forward: the state entering every shard.  A shard whose end state depends on its start state (no synchronisation
    // point in the whole shard: never for the workloads) is fixed from the left in extra rounds.
    kx_fwd_summary fs{};
    if (!err) local(kx_shard_forward(s, &fs));
    bool fixed_me = false;
    for (;;) {
      FwdMsg mine{fs.synced, fs.end_state, fs.head_len, fs.fail_pos, (uint64_t)curn, fixed_me ? 1u : 0u, err ? 1u : 0u};
      rc = gather(&mine, fwd.data(), sizeof(FwdMsg));
      if (rc) return rc;                     // (the transport itself failed: nothing left to tell the peers with)
      if ((rc = collective(fwd))) return rc;
      bool all = true;
      for (int r = 0; r < world; ++r) all = all && fwd[r].fixed;
      if (all) break;
      // ranks whose left neighbour's end state is known (synchronised or already fixed) fix their head now
      const bool left_known = rank == 0 || fwd[rank - 1].fixed || fwd[rank - 1].synced;
      if (!fixed_me && left_known) {
        if (!err) local(kx_shard_fix_head(s, rank ? fwd[rank - 1].end_state : 0, &fs));
        fixed_me = true;
      }
    }
    {
      uint64_t off = 0, fail = UINT64_MAX;
      for (int r = 0; r < world; ++r) {
        if (fwd[r].fail_pos != UINT64_MAX && fail == UINT64_MAX) fail = off + fwd[r].fail_pos;   // lowest rank = earliest position
        off += fwd[r].n;
      }
      if (fail != UINT64_MAX) {
        res->stats.fail_pos = fail; res->stats.fail_stage = st;
        res->boundary_ms = (float)boundary;
        return KX_MATCH_ERROR;
      }
    }
    // backward: the leaf every shard ends in = the leaf its right neighbour starts in
    kx_bwd_summary bs{};
    if (!err) local(kx_shard_backward(s, &bs));
    {
      BwdMsg mine{}; mine.constant = bs.constant; mine.nleaves = bs.nleaves; memcpy(mine.start_leaf, bs.start_leaf, KX_MAX_LEAVES);
      mine.status = err ? 1u : 0u;
      rc = gather(&mine, bwd.data(), sizeof(BwdMsg));
      if (rc) return rc;
      if ((rc = collective(bwd))) return rc;
    }
    uint32_t end_leaf = 0;
    {
      std::vector<uint32_t> ends(world, 0);
      for (int r = world - 2; r >= 0; --r) ends[r] = bwd[r + 1].start_leaf[ends[r + 1]];
      end_leaf = ends[rank];
    }
    uint64_t ol = 0;
    if (!err) local(kx_shard_resolve(s, end_leaf, &ol));
    // emit: the last stage into the caller's buffer, the others into a buffer that is the next stage's input — the buffers are
    // claimed BEFORE the lengths are exchanged, so that a rank without room says so in that exchange
    void* dst = d_out; size_t dcap = cap;
    if (!err) {
      if (st + 1 < ns) {
        const int slot = st & 1;
        if (stagebuf[slot]) { (void)hipFree(stagebuf[slot]); stagebuf[slot] = nullptr; }
        if (hipMalloc(&stagebuf[slot], ol + 256) != hipSuccess) local(sErr(KX_E_HIP, "hipMalloc(stage buffer) failed"));
        dst = stagebuf[slot]; dcap = ol + 256;
      } else if (!d_out && cap == 0 && res_alloc) {   // (kx_run_fd_sharded: the library allocates the rank's output itself)
        if (hipMalloc(&dst, ol + 256) != hipSuccess) local(sErr(KX_E_HIP, "hipMalloc(output) failed"));
        else { *res_alloc = dst; dcap = ol + 256; }
      } else if (ol > cap || (ol && !d_out)) local(sErr(KX_E_CAPACITY, "output buffer too small"));
    }
    {
      LenMsg mine{ol, err ? 1u : 0u, 0u};
      rc = gather(&mine, lens.data(), sizeof(LenMsg));
      if (rc) return rc;
      if ((rc = collective(lens))) return rc;
    }
    if (st + 1 == ns) {
      uint64_t off = 0, total = 0;
      for (int r = 0; r < world; ++r) { if (r < rank) off += lens[r].len; total += lens[r].len; }
      res->out_len = ol; res->out_offset = off; res->total_out = total;
    }
    // (a failure of the emit itself is local again: it is reported by the next stage's first exchange, or — on the last
    //  stage — by one more exchange of status words, so that the return code is collective for the whole call and no caller
    //  assembles a truncated output from the ranks that succeeded)
    if (!err) local(kx_shard_emit(s, dst, dcap));
    if (st + 1 == ns && world > 1) {
      LenMsg mine{0, err ? 1u : 0u, 0u};
      rc = gather(&mine, lens.data(), sizeof(LenMsg));
      res->boundary_ms = (float)boundary;   // (this exchange is part of the hand-off time, also on the ways out below)
      if (rc) return rc;
      if ((rc = collective(lens))) return rc;
    } else if (st + 1 == ns && err) { res->boundary_ms = (float)boundary; return sErr(err, err_msg); }
    cur = dst; curn = ol;
  }
  res->stats.in_bytes = n; res->stats.out_bytes = res->out_len;
  res->boundary_ms = (float)boundary;
  return 0;
}

int kx_run_sharded(kx_program* p, int rank, int world, kx_allgather_fn ag, void* ag_ctx, const void* d_in, size_t n,
                   void* d_out, size_t cap, kx_sharded_result* res, void* stream) {
  return runSharded(p, rank, world, ag, ag_ctx, d_in, n, d_out, cap, res, stream, nullptr);
}

// `BIN --gpus N < file > out`: the input (a regular file) cut into N contiguous shards at 4 KiB multiples, one thread and
// one program instance per GPU, the hand-off among the threads through host memory; every rank writes its slice of the
// output at its own offset (a regular file) or in rank order (a pipe).  kx_config::force & KX_FORCE_SAME_DEVICE stacks the ranks on
// the current device (validation on a one-GPU box; the produced binary maps KX_SHARD_SAME_DEVICE=1 onto it).
int kx_run_fd_sharded(const void* blob, size_t blob_len, int ngpus, int in_fd, int out_fd, kx_stats* stats) {
  return kx_run_fd_sharded_cfg(blob, blob_len, nullptr, ngpus, in_fd, out_fd, stats);
}

int kx_run_fd_sharded_cfg(const void* blob, size_t blob_len, const kx_config* cfg, int ngpus, int in_fd, int out_fd, kx_stats* stats) {
  if (!blob || ngpus < 1) return sErr(KX_E_ARG, "bad arguments");
  struct stat sti, sto;
  if (fstat(in_fd, &sti) || !S_ISREG(sti.st_mode)) return sErr(KX_E_ARG, "--gpus needs a regular file on stdin (each GPU reads its own shard)");
  const bool out_seekable = fstat(out_fd, &sto) == 0 && S_ISREG(sto.st_mode);
  const off_t out_base = out_seekable ? lseek(out_fd, 0, SEEK_CUR) : 0;   // (output starts where the descriptor stands)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return sErr(KX_E_HIP, "no HIP device available: the engine has no CPU fallback");
  const bool same = cfg && (cfg->force & KX_FORCE_SAME_DEVICE);
  if (!same && ngpus > ndev) return sErr(KX_E_ARG, "--gpus " + std::to_string(ngpus) + " but " + std::to_string(ndev) + " visible device(s)");
  const uint64_t n = (uint64_t)sti.st_size;
  const uint64_t L = ngpus > 1 ? (n / ngpus) / 4096 * 4096 : n;
  kx_group* grp = kx_group_create(ngpus);
  std::vector<int> rcs(ngpus, 0);
  std::vector<std::string> errs(ngpus);
  std::vector<kx_sharded_result> results(ngpus);
  std::mutex wm; std::condition_variable wcv; int next_writer = 0; bool abort_write = false;
  int base_dev = 0; (void)hipGetDevice(&base_dev);   // the CALLER's current device (a fresh thread starts on device 0)
  auto body = [&](int r) {
    auto fail = [&](int code, const std::string& m) { rcs[r] = code; errs[r] = m; };
    if (hipSetDevice(same ? base_dev : r) != hipSuccess) { fail(KX_E_HIP, "hipSetDevice failed"); }
    const uint64_t start = (uint64_t)r * L, len = r == ngpus - 1 ? n - start : L;
    kx_program* prog = nullptr; void *d_in = nullptr, *d_out = nullptr; std::vector<char> host;
    kx_group_member* mb = kx_group_join(grp, r);
    if (!rcs[r] && kx_load_config(blob, blob_len, cfg, &prog)) fail(KX_E_BLOB, kx_last_error());
    if (!rcs[r]) {
      host.resize(len ? len : 1);
      uint64_t got = 0;
      while (got < len) { ssize_t k = pread(in_fd, host.data() + got, len - got, (off_t)(start + got)); if (k <= 0) { fail(KX_E_IO, "read error on the input file"); break; } got += (uint64_t)k; }
    }
    if (!rcs[r] && len && (hipMalloc(&d_in, len) != hipSuccess || hipMemcpy(d_in, host.data(), len, hipMemcpyHostToDevice) != hipSuccess)) fail(KX_E_HIP, "cannot place the shard on the device");
    // (a rank that failed before this point must still take part in the exchanges: it would deadlock the others otherwise —
    //  so such failures abort the whole run below, before anyone enters the protocol)
    {
      int bad = rcs[r] ? 1 : 0;
      std::vector<int> all(ngpus);
      kx_group_allgather(mb, &bad, all.data(), sizeof(int));
      bool stop = false; for (int x : all) stop = stop || x;
      if (stop) { if (!rcs[r]) fail(KX_E_IO, "another rank failed to set up"); }
    }
    if (!rcs[r]) {
      int rc = runSharded(prog, r, ngpus, kx_group_allgather, mb, d_in, len, nullptr, 0, &results[r], nullptr, &d_out);
      if (rc) { fail(rc, kx_last_error()); if (rc != KX_MATCH_ERROR) kx_group_abort(grp); }   // (a match error is seen by every rank; anything else by this one only)
    }
    if (!rcs[r]) {
      const uint64_t ol = results[r].out_len;
      host.resize(ol ? ol : 1);
      if (ol && hipMemcpy(host.data(), d_out, ol, hipMemcpyDeviceToHost) != hipSuccess) fail(KX_E_HIP, "D2H copy failed");
    }
    // output: at the rank's own offset, or in rank order
    if (out_seekable) {
      if (!rcs[r]) {
        uint64_t put = 0; const uint64_t ol = results[r].out_len;
        while (put < ol) { ssize_t k = pwrite(out_fd, host.data() + put, ol - put, out_base + (off_t)(results[r].out_offset + put)); if (k <= 0) { fail(KX_E_IO, "write error"); break; } put += (uint64_t)k; }
      }
    } else {
      std::unique_lock<std::mutex> lk(wm);
      wcv.wait(lk, [&] { return next_writer == r; });
      if (rcs[r]) abort_write = true;
      if (!abort_write) {
        uint64_t put = 0; const uint64_t ol = results[r].out_len;
        while (put < ol) { ssize_t k = write(out_fd, host.data() + put, ol - put); if (k <= 0) { fail(KX_E_IO, "write error"); abort_write = true; break; } put += (uint64_t)k; }
      }
      ++next_writer; wcv.notify_all();
    }
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (prog) kx_free(prog);
    kx_group_leave(mb);
  };
  std::vector<std::thread> th;
  for (int r = 0; r < ngpus; ++r) th.emplace_back(body, r);
  for (auto& t : th) t.join();
  kx_group_free(grp);
  int rc = 0;
  for (int r = 0; r < ngpus; ++r) if (rcs[r] && (!rc || rcs[r] == KX_MATCH_ERROR)) { rc = rcs[r]; if (rc != KX_MATCH_ERROR) kx_internal_set_error(errs[r].c_str()); }
  if (stats) {   // kernel times: the slowest rank's (the ranks run side by side); counts: summed
    *stats = results[0].stats;
    for (int r = 1; r < ngpus; ++r) {
      for (int i = 0; i < KX_NKERNELS; ++i) if (results[r].stats.kernel_ms[i] > stats->kernel_ms[i]) stats->kernel_ms[i] = results[r].stats.kernel_ms[i];
      if (results[r].stats.total_ms > stats->total_ms) stats->total_ms = results[r].stats.total_ms;
      stats->unsynced_segments += results[r].stats.unsynced_segments;
    }
    stats->in_bytes = n; stats->out_bytes = results[0].total_out;
  }
  if (rc == 0 && out_seekable) {   // (pwrite does not move the descriptor's offset: leave it behind the output, as write would)
    (void)lseek(out_fd, out_base + (off_t)results[0].total_out, SEEK_SET);
  }
  return rc;
}

}  // extern "C"

