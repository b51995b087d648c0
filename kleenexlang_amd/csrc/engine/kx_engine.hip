// kx_engine.hip — MI355X (gfx950) streaming-SST execution engine behind include/kxhip.h.
//
// What the reference does per input symbol on one CPU core — `match<K>()`'s compare-and-branch
// cascade plus crt.c's append/concat/output calls (Backends/C.hs:72-83,486-493; crt/crt.c:171-283)
// — is re-designed here for a chip with 256 CUs and the whole input resident in HBM:
//
//   the SST's registers exist because a one-pass CPU run cannot know yet which of the live paths
//   of the path tree will survive (Determinization.hs:24-28); with the input resident we resolve
//   that by a second, backward sweep instead of parking bytes.  The engine therefore executes the
//   *path form* of the program (include/kxp_format.h): forward = the SST's state sequence,
//   backward = which leaf of each state's path tree lies on the surviving path, output = the
//   per-step suffixes along that path, written at prefix-summed offsets.  No register bytes ever
//   move; results are bit-identical to the register form (tests/).
//
// Kernels (table image staged in LDS; DESIGN.md §3 has the layout and the byte accounting):
//   k_sync     start state of every segment: run the compile-time synchronising automaton from the
//              segment start until every possible start state has converged (speculation resolved
//              without enumerating states at run time)
//   k_forward  lane = segment: state sequence from the sync point; one state checkpoint per 64-byte
//              piece; failure position (first symbol without transition) by atomicMin
//   k_head     (sharded runs) the few leading bytes of a shard that need the previous shard's state
//   k_backlen  lane = block: output length and start leaf for every candidate end leaf (candidates
//              merge after about one record); per-piece {running length, end leaf} records
//   k_resolve  end leaf per block from its successor's summary; exclusive scan of the block output
//              lengths (k_scan_*); k_fixtail re-walks the few pieces above each block's merge point
//   k_emit     lane = piece, persistent: re-derive the piece, walk it backward along the resolved
//              path, assemble the wave's contiguous output in LDS, flush with 16-byte stores
// DESIGN.md has the data layout, the byte accounting and the roofline for each.
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../../include/kxhip.h"
#include "../../../include/kxp_format.h"

namespace {

thread_local std::string g_err;
int setErr(int code, const std::string& m) { g_err = m; return code; }

#define HIPCHECK(expr)                                                                          \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return setErr(KX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));               \
  } while (0)

constexpr int PIECE = 64;                 // bytes per lane-step of the backward kernels (one piece record each)
constexpr int HALF = 32;                  // bytes between state checkpoints (two dependency chains per piece)
constexpr uint32_t OFF_FWD = 256;         // the table image starts with the 256-byte class table; state rows follow
constexpr uint64_t UNSYNC = ~0ull;
constexpr uint64_t NOFAIL = ~0ull;
constexpr int EMIT_STG = 6144;            // k_emit: least staging bytes per wave (one wave-iteration of apache_log fits)
constexpr int EMIT_JOBS_MIN = 4096;       // k_emit: least bytes of constant-copy job slots per wave (4 B each)
constexpr int EMIT_WAVE_LDS_MIN = EMIT_STG + 16 + EMIT_JOBS_MIN;

// ------------------------------------------------------------------ device-side program view
// The table image is copied verbatim into LDS (at LDS address 0 of the dynamic segment) and is
// addressed by *byte offsets* that are stored pre-scaled inside the tables themselves, so that a
// lookup is one add + one ds_read:
//   state handle h   = byte offset of the state's row in the image     (h = off_fwd + state·C·4)
//   cls4[byte]       = class·4 (u8 table at LDS address 0; programs with more than 64 byte classes
//                      store the class index and shift: DevTables::cshift)
//   e = fwd[h + cls4[b]] :  low 16 = next handle, high 16 = byte offset of the transition's back row
//   back entry x = ent[row + leaf·4] :
//     bit 0 "no input byte copied" | bits 2-9 leaf' (so x & 0x3FC = leaf'·4) | bits 10-22 pool offset |
//     bit 23 "a constant follows" | bits 24-30 appended bytes (copy included); bit 31 = 0
//     appended = 127 escapes to the wide side tables wlen/woff (global memory) for long constants.
//   The layout serves k_emit's step: the byte count is a whole byte (SDWA operand), bit 0 shifted to
//   bit 31 turns the staging address of a step that copies nothing into an out-of-range LDS address
//   (the hardware drops such stores), and bit 23 is the sign of byte 2 (one SDWA compare).
struct DevTables {
  const uint32_t* packed;     // [cls4 | fwd | ent | pool] image
  uint32_t packed_words;
  uint32_t off_fwd, off_ent, off_pool, off_cls;  // byte offsets inside the image
  uint32_t nstates, nclasses, q0h, maxleaves, deadh, nullrow, cshift;
  const uint32_t* wlen;       // [nent] appended byte count / pool offset per back entry (wide form)
  const uint32_t* woff;
  const uint8_t* cls;         // global copy for kernels that do not stage the big image
  const uint8_t* nleaves;     // [nstates+1]  by state id
  const uint8_t* fin_leaf;    // [nstates+1]
  const uint16_t* sync16;     // [(nsync+1)*C next | (nsync+1) state]: subsets < sync_multi are undecided
  uint32_t nsync, sync_multi, sync_words;  // sync_words = u32 words of the sync16 image
  const uint32_t* init_off;   // [maxleaves] pool offset / length of the initial closure output
  const uint32_t* init_len;
};

struct Lds {
  const uint8_t* base; uint32_t cls, pool, ent, csh;
  __device__ __forceinline__ uint32_t w(uint32_t off) const { return *reinterpret_cast<const uint32_t*>(base + off); }
  // one byte per input symbol: ASCII text then never collides in an LDS bank (a u16 table aliases b and b+64)
  __device__ __forceinline__ uint32_t c4(uint32_t byte) const { return (uint32_t)base[cls + byte] << csh; }
  __device__ __forceinline__ uint32_t next(uint32_t h, uint32_t byte) const { return w(h + c4(byte)); }
  __device__ __forceinline__ uint8_t pb(uint32_t off) const { return base[pool + off]; }
};
#define E_LEAF4(e) ((e) & 0x3FCu)
#define E_COPY(e) (~(e) & 1u)
#define E_DLEN7(e) ((e) >> 24)
#define E_OFF13(e) (((e) >> 10) & 0x1FFFu)
#define E_HASCONST(e) (((e) >> 23) & 1u)
// WIDE = the program has back entries in the escaped wide form (long constants); programs without
// them (all five workloads) run kernel instances in which the escape test does not exist at all.
template <bool WIDE>
__device__ __forceinline__ uint32_t ent_dlen(uint32_t e, uint32_t addr, const Lds& L, const DevTables& T) {
  uint32_t d = E_DLEN7(e);
  if (WIDE) { if (__builtin_expect(d == 127u, 0)) d = T.wlen[(addr - L.ent) >> 2]; }
  return d;
}
template <bool WIDE>
__device__ __forceinline__ uint32_t ent_off(uint32_t e, uint32_t addr, const Lds& L, const DevTables& T) {
  if (WIDE) { if (__builtin_expect(E_DLEN7(e) == 127u, 0)) return T.woff[(addr - L.ent) >> 2]; }
  return E_OFF13(e);
}

// GENERAL = kernel instance for programs with wide back entries or more than 64 byte classes; the
// common instances know at compile time that neither exists (class shift 0, no escape tests).
template <bool GENERAL>
__device__ __forceinline__ Lds stage_tables(const DevTables& T, uint32_t* smem) {
  for (uint32_t i = threadIdx.x; i < T.packed_words; i += blockDim.x) smem[i] = T.packed[i];
  __syncthreads();
  Lds L;
  L.base = reinterpret_cast<const uint8_t*>(smem); L.cls = T.off_cls; L.pool = T.off_pool; L.ent = T.off_ent;
  L.csh = GENERAL ? T.cshift : 0u;
  return L;
}

// per-piece result of the backward length pass: end leaf and running output length (see k_backlen)
struct __attribute__((aligned(8))) PieceRec { int32_t cum; uint32_t leaf; };

struct Flags {               // one per shard, device memory
  unsigned long long fail_pos;
  uint32_t end_state;        // handle of the state entering the byte after the shard
  uint32_t unsynced;
  unsigned long long total_len;
  uint32_t first_merged;
  uint32_t emit_ovf;         // k_emit: pieces whose constants did not fit their job slots
  uint32_t emit_kmax;        // k_fixtail's sample: most constants met in one piece,
  uint32_t emit_ksum, emit_kn;   //   constants / pieces over a subsample (their average)
  uint32_t pad[3];
};

// Bit-field extract that the scheduler may not issue before `dep` exists.  The sweeps below are one
// long dependency chain; without this the compiler unpacks all 64 bytes / 64 offsets of a piece up
// front and holds them in ~128 VGPRs (measured: 181 VGPRs + scratch vs < 100).
template <int OFF, int WIDTH>
__device__ __forceinline__ uint32_t bfe_after(uint32_t word, uint32_t dep) {
  uint32_t r;
  asm("v_bfe_u32 %0, %1, %2, %3 ; after %4" : "=v"(r) : "v"(word), "n"(OFF), "n"(WIDTH), "v"(dep));
  return r;
}
#define BYTE_AT_DEP(w, t, dep) bfe_after<((t) & 3) * 8, 8>((w)[(t) >> 2], (dep))
#define BO_GET_DEP(bo, t, dep) bfe_after<((t) & 1) * 16, 16>((bo)[(t) >> 1], (dep))
// Make `x` depend on `a` (no instruction emitted): forces side accumulations of a chain step to be
// retired before the next step instead of keeping every step's table word alive until the end.
__device__ __forceinline__ void tie(uint32_t& x, uint32_t a) { asm("; tie" : "+v"(x) : "v"(a)); }
// compile-time step loop: the step index is a constant expression inside the body
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// ------------------------------------------------------------------------------- k_sync
// Segment k starts at byte k*seg with an unknown state.  sync16 is the subset automaton "set of all
// states → …" built by the compiler, renumbered so that undecided subsets come first and decided
// ones (singleton / empty / capped) are absorbing: a lane just counts how many steps stay undecided.
// Segments that do not converge inside their own bytes are marked UNSYNC and are run through by the
// preceding lane.
template <bool IN_LDS>
__global__ void k_sync(const uint8_t* __restrict__ in, uint64_t n, uint64_t seg, uint32_t nseg, int first_known,
                       uint64_t* __restrict__ seg_pos, uint16_t* __restrict__ seg_state, Flags* flags, DevTables T) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint16_t* tab = T.sync16;
  const uint8_t* cls = T.cls;
  if (IN_LDS) {
    const uint32_t* src = (const uint32_t*)T.sync16;
    for (uint32_t i = threadIdx.x; i < T.sync_words; i += blockDim.x) smem[i] = src[i];
    const uint32_t* csrc = (const uint32_t*)T.cls;
    for (uint32_t i = threadIdx.x; i < 64; i += blockDim.x) smem[T.sync_words + i] = csrc[i];
    __syncthreads();
    tab = (const uint16_t*)smem;
    cls = (const uint8_t*)(smem + T.sync_words);
  }
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nseg) return;
  if (k == 0 && first_known) { seg_pos[0] = 0; seg_state[0] = (uint16_t)T.q0h; return; }
  const uint32_t C = T.nclasses, M = T.sync_multi;
  const uint16_t* state_of = tab + (size_t)(T.nsync + 1) * C;
  uint64_t pos = (uint64_t)k * seg;
  const uint64_t limit = pos + seg < n ? pos + seg : n;
  uint32_t sid = 0, cnt = 0;
  while (sid < M && pos + 16 <= limit) {
    const uint4 v = *reinterpret_cast<const uint4*>(in + pos);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      sid = tab[sid * C + cls[(w[b >> 2] >> ((b & 3) * 8)) & 0xFFu]];
      cnt += sid < M ? 1u : 0u;
    }
    pos += 16;
  }
  while (sid < M && pos < limit) {
    sid = tab[sid * C + cls[in[pos]]];
    cnt += sid < M ? 1u : 0u;
    ++pos;
  }
  const uint32_t st = sid < M ? 0xFFFFu : state_of[sid];
  if (st < 0xFFF0u) { seg_pos[k] = (uint64_t)k * seg + cnt + (M ? 1 : 0); seg_state[k] = (uint16_t)(OFF_FWD + st * C * 4); }
  else { seg_pos[k] = UNSYNC; seg_state[k] = 0; atomicAdd(&flags->unsynced, 1u); }
}

// the hand-scheduled per-piece instruction sequences (generated text, see gen_sweeps.py)
#include "kx_sweeps.inc"

// ---------------------------------------------------------------------------- k_forward
// The absorbing dead handle stands for "no transition", so the hot loop has no failure branch; the
// exact position is recovered by re-running the 64-byte piece in which the run died.
__device__ __forceinline__ void load_piece(const uint8_t* __restrict__ in, uint64_t n, uint64_t pstart, uint32_t (&w)[16]) {
  if (pstart + PIECE <= n) {
    const uint4* p = reinterpret_cast<const uint4*>(in + pstart);
#pragma unroll
    for (int i = 0; i < 4; ++i) { uint4 v = p[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      uint32_t x = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) { uint64_t a = pstart + 4 * i + b; if (a < n) x |= (uint32_t)in[a] << (8 * b); }
      w[i] = x;
    }
  }
}

// two pieces = one 128-byte line per lane and trip: lanes are a whole segment apart, so a line that is
// only half consumed gets evicted from L2 before its other half is wanted (measured: 2x over-fetch)
__device__ __forceinline__ void load_pair(const uint8_t* __restrict__ in, uint64_t n, uint64_t pstart, uint32_t (&wa)[16],
                                          uint32_t (&wb)[16]) {
  if (pstart + 2 * PIECE <= n) {
    const uint4* p = reinterpret_cast<const uint4*>(in + pstart);
    uint4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = p[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wa[4 * i] = v[i].x; wa[4 * i + 1] = v[i].y; wa[4 * i + 2] = v[i].z; wa[4 * i + 3] = v[i].w;
      wb[4 * i] = v[4 + i].x; wb[4 * i + 1] = v[4 + i].y; wb[4 * i + 2] = v[4 + i].z; wb[4 * i + 3] = v[4 + i].w;
    }
  } else {
    load_piece(in, n, pstart, wa);
    load_piece(in, n, pstart + PIECE, wb);
  }
}

template <bool GENERAL>
__global__ void k_forward(const uint8_t* __restrict__ in, uint64_t n, uint32_t nseg,
                          const uint64_t* __restrict__ seg_pos, const uint16_t* __restrict__ seg_state,
                          uint16_t* __restrict__ chk, Flags* flags, DevTables T) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  Lds L = stage_tables<GENERAL>(T, smem);
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nseg) return;
  uint64_t pos = seg_pos[k];
  if (pos == UNSYNC) return;
  uint32_t j = k + 1;
  while (j < nseg && seg_pos[j] == UNSYNC) ++j;
  const bool last = j >= nseg;
  const uint64_t end = last ? n : seg_pos[j];
  const uint32_t dead = T.deadh;
  uint32_t h = seg_state[k];
  bool failed = false;
  while (pos < end && (pos & (PIECE - 1))) {
    if ((pos & (HALF - 1)) == 0) chk[pos >> 5] = (uint16_t)h;
    uint32_t nh = L.next(h, in[pos]) & 0xFFFFu;
    if (nh == dead) { failed = true; break; }
    h = nh; ++pos;
  }
  // 64 chained transitions over one piece held in registers; `mid` = the state after the first 32
  auto run_piece = [&](const uint32_t (&w)[16], uint32_t hh, uint32_t& mid) {
    if constexpr (!GENERAL) { piece_run1(w, hh, mid); return hh; }
    uint32_t c[8], cn[8];
    static_for<0, 8>([&](auto ic) { constexpr int i = decltype(ic)::value; c[i] = L.c4(BYTE_AT_DEP(w, i, hh)); });
    static_for<0, PIECE / 8>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      if constexpr (g + 1 < PIECE / 8)
        static_for<0, 8>([&](auto ic) { constexpr int i = decltype(ic)::value; cn[i] = L.c4(BYTE_AT_DEP(w, 8 * (g + 1) + i, hh)); });
      static_for<0, 8>([&](auto ic) { constexpr int i = decltype(ic)::value; hh = L.w(hh + c[i]) & 0xFFFFu; });
      if constexpr (g == HALF / 8 - 1) mid = hh;
      if constexpr (g + 1 < PIECE / 8)
        static_for<0, 8>([&](auto ic) { constexpr int i = decltype(ic)::value; c[i] = cn[i]; });
    });
    return hh;
  };
  auto locate = [&](uint32_t h0) {   // the run died inside the piece at `pos`: find the exact symbol
    h = h0;
    for (int t = 0; t < PIECE; ++t) {
      uint32_t nh = L.next(h, in[pos]) & 0xFFFFu;
      if (nh == dead) break;
      h = nh; ++pos;
    }
    failed = true;
  };
  while (!failed && pos + PIECE <= end) {
    if ((pos & (8 * PIECE - 1)) == 0 && pos + 8 * PIECE <= end) {
      // eight pieces per trip: their sixteen checkpoints leave as two aligned 16-byte stores instead of
      // scattered 2-byte stores (each of which dirties a whole 64-byte sector)
      uint32_t cw[8];
      bool died = false;
      static_for<0, 4>([&](auto pc) {
        constexpr int k2 = decltype(pc)::value;
        if (!died) {
          uint32_t wa[16], wb[16], mid = 0;
          load_pair(in, n, pos, wa, wb);   // one full 128-byte line per lane
          uint32_t h0 = h;
          h = run_piece(wa, h, mid);
          cw[2 * k2] = h0 | (mid << 16);
          if (h == dead) { locate(h0); died = true; }
          else {
            pos += PIECE;
            h0 = h;
            h = run_piece(wb, h, mid);
            cw[2 * k2 + 1] = h0 | (mid << 16);
            if (h == dead) { locate(h0); died = true; } else pos += PIECE;
          }
        }
      });
      if (!died) {
        uint4* dst = reinterpret_cast<uint4*>(chk + ((pos >> 5) - 16));
        dst[0] = make_uint4(cw[0], cw[1], cw[2], cw[3]);
        dst[1] = make_uint4(cw[4], cw[5], cw[6], cw[7]);
      }
      continue;
    }
    uint32_t w[16], mid = 0;
    load_piece(in, n, pos, w);
    const uint32_t h0 = h;
    h = run_piece(w, h, mid);
    if (h == dead) { locate(h0); break; }
    *reinterpret_cast<uint32_t*>(chk + (pos >> 5)) = h0 | (mid << 16);
    pos += PIECE;
  }
  while (!failed && pos < end) {
    if ((pos & (HALF - 1)) == 0) chk[pos >> 5] = (uint16_t)h;
    uint32_t nh = L.next(h, in[pos]) & 0xFFFFu;
    if (nh == dead) { failed = true; break; }
    h = nh; ++pos;
  }
  if (failed) { atomicMin(&flags->fail_pos, (unsigned long long)pos); return; }
  if (last) {
    flags->end_state = h;
    if ((n & (HALF - 1)) == 0) chk[n >> 5] = (uint16_t)h;
  }
}

// sequential run over the head of a shard (bytes before the first synchronised segment)
template <bool GENERAL>
__global__ void k_head(const uint8_t* __restrict__ in, uint64_t n, uint64_t head_len, uint32_t h,
                       uint16_t* __restrict__ chk, Flags* flags, DevTables T) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  Lds L = stage_tables<GENERAL>(T, smem);
  if (blockIdx.x || threadIdx.x) return;
  const uint32_t dead = T.deadh;
  for (uint64_t pos = 0; pos < head_len; ++pos) {
    if ((pos & (HALF - 1)) == 0) chk[pos >> 5] = (uint16_t)h;
    uint32_t nh = L.next(h, in[pos]) & 0xFFFFu;
    if (nh == dead) { atomicMin(&flags->fail_pos, (unsigned long long)pos); return; }
    h = nh;
  }
  if (head_len == n) {
    flags->end_state = h;
    if ((n & (HALF - 1)) == 0) chk[n >> 5] = (uint16_t)h;
  }
}

// --------------------------------------------------------------- piece helpers (backward sweeps)
// A piece is 64 input bytes held in 16 VGPRs; forward re-derivation from its checkpoint yields the
// back-row offset of every step (kept in registers; all loops are fully unrolled so that the
// arrays are statically indexed and never spill to scratch).
// Opaque touch: stops the compiler from keeping per-byte / per-step extractions of a previous phase
// alive (it would otherwise hold 64+64 unpacked values across the phases and spill to scratch).
template <int N>
__device__ __forceinline__ void launder(uint32_t (&a)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(a[i]));
}

// back-row offsets are 16-bit: two steps share a VGPR (keeps the sweeps at 4+ waves per SIMD)
constexpr int BOW = PIECE / 2;
__device__ __forceinline__ void bo_set(uint32_t (&bo)[BOW], int t, uint32_t v) {
  bo[t >> 1] = (t & 1) ? ((bo[t >> 1] & 0xFFFFu) | (v << 16)) : ((bo[t >> 1] & 0xFFFF0000u) | v);
}
// one chain, scheduled by the compiler: k_backlen keeps two pieces of input in registers and has no room
// for the two-chain sequence's 62 simultaneously live registers
__device__ __forceinline__ void piece_forward(const uint32_t (&w)[16], uint32_t h, const Lds& L, uint32_t (&bo)[BOW]) {
  // byte-class lookups do not depend on the state, so they are prefetched one group of 8 ahead of
  // the dependent fwd[] chain — and no further (all 64 at once would cost 64 VGPRs)
  uint32_t c[8], cn[8];
  static_for<0, 8>([&](auto ic) { constexpr int i = decltype(ic)::value; c[i] = L.c4(BYTE_AT_DEP(w, i, h)); });
  static_for<0, PIECE / 8>([&](auto gc) {
    constexpr int g = decltype(gc)::value;
    if constexpr (g + 1 < PIECE / 8)
      static_for<0, 8>([&](auto ic) { constexpr int i = decltype(ic)::value; cn[i] = L.c4(BYTE_AT_DEP(w, 8 * (g + 1) + i, h)); });
    static_for<0, 4>([&](auto ic) {
      constexpr int i = 2 * decltype(ic)::value;
      const uint32_t e0 = L.w(h + c[i]);
      h = e0 & 0xFFFFu;
      const uint32_t e1 = L.w(h + c[i + 1]);
      h = e1 & 0xFFFFu;
      bo[(8 * g + i) >> 1] = (e0 >> 16) | (e1 & 0xFFFF0000u);
    });
    if constexpr (g + 1 < PIECE / 8)
      static_for<0, 8>([&](auto ic) { constexpr int i = decltype(ic)::value; c[i] = cn[i]; });
  });
}
// Steps beyond the end of a partial piece point at the identity row (parent = leaf, nothing
// appended), so the sweeps below need no per-step bounds test.
__device__ __forceinline__ void mask_tail(uint32_t (&bo)[BOW], int plen, uint32_t nullrow) {
  if (plen < PIECE) {
#pragma unroll
    for (int t = 0; t < PIECE; ++t) if (t >= plen) bo_set(bo, t, nullrow);
  }
}

// ---------------------------------------------------------------------------- k_backlen
// back_lo[row + leaf] = parent | copy<<8 | (bytes appended on this step)<<9.
// For block m and every leaf the block could end in: where the path enters the block (start leaf)
// and how many output bytes the block contributes.  Candidates are advanced piece by piece and
// collapse to one as soon as they agree; from there on the per-piece end leaf and the running
// output length are final and are written out for k_emit (pleaf/pcum).  Pieces above the merge
// point (the block's tail) are finished by k_fixtail once the block's end leaf is known.
template <bool WIDE>
__device__ __forceinline__ uint32_t walk_len(const uint32_t (&bo)[BOW], uint32_t& leaf, const Lds& L, const DevTables& T) {
  uint32_t sum = 0;
  static_for<0, PIECE>([&](auto ic) {
    constexpr int t = PIECE - 1 - decltype(ic)::value;
    const uint32_t a = BO_GET_DEP(bo, t, leaf) + leaf;
    const uint32_t e = L.w(a); sum += ent_dlen<WIDE>(e, a, L, T); leaf = E_LEAF4(e); tie(leaf, sum);
  });
  return sum;
}
// The same walk, also reporting where it stood in the middle of the piece (leaf entering step 31 and
// the bytes appended by steps 63..32): k_emit starts its second dependency chain there.
template <bool WIDE, bool COUNT = false>   // COUNT: also count the steps that append a constant
__device__ __forceinline__ uint32_t walk_len_mid(const uint32_t (&bo)[BOW], uint32_t& leaf, uint32_t& leaf_mid, uint32_t& sum_hi,
                                                 const Lds& L, const DevTables& T, uint32_t* nconst = nullptr) {
  uint32_t sum = 0, k = 0;
  static_for<0, PIECE>([&](auto ic) {
    constexpr int t = PIECE - 1 - decltype(ic)::value;
    const uint32_t a = BO_GET_DEP(bo, t, leaf) + leaf;
    const uint32_t e = L.w(a); sum += ent_dlen<WIDE>(e, a, L, T); leaf = E_LEAF4(e); tie(leaf, sum);
    if constexpr (COUNT) { k += E_HASCONST(e); tie(leaf, k); }
    if constexpr (t == HALF) { leaf_mid = leaf; sum_hi = sum; }
  });
  if constexpr (COUNT) *nconst = k;
  return sum;
}
// piece record word: end leaf | leaf in the middle << 8 | min(bytes of the upper half, 0xFFFF) << 16
__device__ __forceinline__ uint32_t rec_word(uint32_t leaf_end, uint32_t leaf_mid4, uint32_t sum_hi) {
  return leaf_end | ((leaf_mid4 >> 2) << 8) | ((sum_hi < 0xFFFFu ? sum_hi : 0xFFFFu) << 16);
}

template <int MAXC, bool WIDE>
__global__ void k_backlen(const uint8_t* __restrict__ in, uint64_t n, uint64_t blk, uint32_t nblk,
                          const uint16_t* __restrict__ chk, const Flags* flags, int is_last,
                          uint8_t* __restrict__ bs_start, uint32_t* __restrict__ bs_len, uint8_t* __restrict__ bs_merged,
                          uint8_t* __restrict__ bs_mstart, uint32_t Lc, PieceRec* __restrict__ prec,
                          uint16_t* __restrict__ merge_piece, uint32_t* __restrict__ ctot, DevTables T) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  Lds L = stage_tables<WIDE>(T, smem);
  // per-lane staging behind the table image: 8 piece records (64 B) and 8 piece-start checkpoints (16 B), so
  // that both move as whole aligned lines instead of scattered 1-4 byte accesses
  uint4* lrec = reinterpret_cast<uint4*>(smem + ((T.packed_words + 3) & ~3u)) + (size_t)threadIdx.x * 5;
  uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= nblk) return;
  const uint64_t bstart = (uint64_t)m * blk;
  const uint64_t bend = bstart + blk < n ? bstart + blk : n;
  const uint32_t qe = ((bend == n ? flags->end_state : chk[bend >> 5]) - OFF_FWD) / (T.nclasses * 4);
  // candidate 0 lives in registers (after merging it is the only one left); candidates 1.. in private memory
  uint16_t cl[MAXC]; uint32_t clen[MAXC]; uint32_t pre[MAXC];   // cl = leaf·4
  uint32_t nc, nact, leaf0, len0 = 0, pre0 = 0;
  uint32_t known = 0xFFFFFFFFu;
  if (bend == n && is_last) { nc = 1; known = T.fin_leaf[qe]; leaf0 = known * 4; }
  else {
    nc = T.nleaves[qe]; if (nc > MAXC) nc = MAXC;
    leaf0 = 0;
    for (uint32_t j = 1; j < nc; ++j) { cl[j] = (uint16_t)(j * 4); clen[j] = 0; pre[j] = 0; }
  }
  nact = nc;
  const uint32_t npieces = (uint32_t)((bend - bstart + PIECE - 1) / PIECE);
  const uint64_t piece0 = bstart >> 6;
  const bool batch = (blk & (8 * PIECE - 1)) == 0;   // blocks start on an 8-piece boundary
  uint32_t mp = nact == 1 ? npieces : 0;   // pieces [mp, npieces) form the unresolved tail
  for (uint32_t pp = (npieces + 1) / 2; pp-- > 0;) {
    uint32_t wa[16], wb[16];
    load_pair(in, n, bstart + (uint64_t)pp * 2 * PIECE, wa, wb);
#pragma unroll
    for (int hf = 1; hf >= 0; --hf) {
      const uint32_t p = 2 * pp + hf;
      if (p < npieces) {
        const uint32_t (&w)[16] = hf ? wb : wa;
        const uint64_t pstart = bstart + (uint64_t)p * PIECE;
        const int plen = (int)(bend - pstart < PIECE ? bend - pstart : PIECE);
        uint32_t bo[BOW];
        // checkpoints arrive 16 at a time (k_forward stores them that way); the piece index is 8-aligned
        // with the block, so group (p | 7) is fetched when the sweep first enters it
        uint32_t hA;
        if (batch) {
          if ((p & 7) == 7 || p == npieces - 1) {
            const uint4* src = reinterpret_cast<const uint4*>(chk + 2 * ((piece0 + p) & ~7ull));
            const uint4 c0 = src[0], c1 = src[1];   // (start, middle) handles of 8 pieces: keep the starts
            lrec[4] = make_uint4((c0.x & 0xFFFFu) | (c0.y << 16), (c0.z & 0xFFFFu) | (c0.w << 16),
                                 (c1.x & 0xFFFFu) | (c1.y << 16), (c1.z & 0xFFFFu) | (c1.w << 16));
          }
          hA = reinterpret_cast<const uint16_t*>(lrec + 4)[p & 7];
        } else hA = chk[2 * (piece0 + p)];
        if constexpr (WIDE) piece_forward(w, hA, L, bo);
        else piece_forward1(w, hA, 0xFFFF0000u, bo);
        mask_tail(bo, plen, T.nullrow);
        if (nact == 1) {
          const uint32_t leaf_end = leaf0 >> 2;
          uint32_t lmid = 0, shi = 0;
          if constexpr (WIDE) len0 += walk_len_mid<WIDE>(bo, leaf0, lmid, shi, L, T);
          else { uint32_t sm = 0; piece_walk1(bo, leaf0, sm, lmid, shi); len0 += sm; }
          const PieceRec rec{(int32_t)(len0 - pre0), rec_word(leaf_end, lmid, shi)};
          if (batch) {
            reinterpret_cast<PieceRec*>(lrec)[p & 7] = rec;
            if ((p & 7) == 0) {   // records p..p+7 are complete (or belong to the unresolved tail: k_fixtail rewrites those)
              uint4* dst = reinterpret_cast<uint4*>(prec + piece0 + p);
              dst[0] = lrec[0]; dst[1] = lrec[1]; dst[2] = lrec[2]; dst[3] = lrec[3];
            }
          } else prec[piece0 + p] = rec;
        } else {
          len0 += walk_len<WIDE>(bo, leaf0, L, T);
          bool same = true;
          for (uint32_t j = 1; j < nact; ++j) {
            uint32_t lf = cl[j];
            clen[j] += walk_len<WIDE>(bo, lf, L, T);
            cl[j] = (uint16_t)lf;
            same = same && lf == leaf0;
          }
          if (same) { pre0 = len0; for (uint32_t j = 1; j < nc; ++j) pre[j] = clen[j]; nact = 1; mp = p; }
        }
      }
    }
  }
  const bool merged = nact == 1;
  const uint32_t tail = len0 - pre0;
  if (known != 0xFFFFFFFFu) {
    bs_start[(size_t)m * Lc + known] = (uint8_t)(leaf0 >> 2); bs_len[(size_t)m * Lc + known] = len0;
  } else {
    bs_start[(size_t)m * Lc] = (uint8_t)(leaf0 >> 2); bs_len[(size_t)m * Lc] = len0;
    for (uint32_t j = 1; j < nc; ++j) {   // after merging only candidate 0 kept accumulating
      bs_start[(size_t)m * Lc + j] = (uint8_t)((merged ? leaf0 : (uint32_t)cl[j]) >> 2);
      bs_len[(size_t)m * Lc + j] = merged ? pre[j] + tail : clen[j];
    }
  }
  bs_merged[m] = merged ? 1 : 0;
  bs_mstart[m] = (uint8_t)(leaf0 >> 2);
  merge_piece[m] = (uint16_t)(merged ? mp : 0);
  ctot[m] = merged ? tail : 0;
}

// ---------------------------------------------------------------------------- k_resolve
// End leaf of block m = start leaf of block m+1 under *its* end leaf; blocks whose candidates merged
// cut the dependency chain, so this is a short local walk (usually one step).
__global__ void k_resolve(uint32_t nblk, uint32_t end_leaf, const uint8_t* __restrict__ bs_start,
                          const uint32_t* __restrict__ bs_len, const uint8_t* __restrict__ bs_merged,
                          const uint8_t* __restrict__ bs_mstart, uint32_t Lc, uint8_t* __restrict__ E,
                          uint32_t* __restrict__ len, unsigned long long* __restrict__ wsum) {
  __shared__ unsigned long long red[1024];
  uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t mylen = 0;
  if (m < nblk) {
    uint32_t e;
    if (m == nblk - 1) e = end_leaf;
    else {
      uint32_t j = m + 1;
      while (!bs_merged[j] && j < nblk - 1) ++j;
      uint32_t leaf = bs_merged[j] ? bs_mstart[j] : bs_start[(size_t)j * Lc + end_leaf];
      for (uint32_t i = j - 1; i > m; --i) leaf = bs_start[(size_t)i * Lc + leaf];
      e = leaf;
    }
    E[m] = (uint8_t)e;
    mylen = bs_len[(size_t)m * Lc + e];
    len[m] = mylen;
  }
  red[threadIdx.x] = mylen;
  __syncthreads();
  for (uint32_t s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) wsum[blockIdx.x] = red[0];
}

// exclusive scan of the per-workgroup sums (one workgroup), then of the blocks inside each group
__global__ void k_scan_groups(uint32_t ngroups, const unsigned long long* __restrict__ wsum,
                              unsigned long long* __restrict__ woff, Flags* flags) {
  __shared__ unsigned long long buf[1024];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < ngroups; base += blockDim.x) {
    uint32_t i = base + threadIdx.x;
    unsigned long long v = i < ngroups ? wsum[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t d = 1; d < blockDim.x; d <<= 1) {
      unsigned long long x = threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
      __syncthreads();
      buf[threadIdx.x] += x;
      __syncthreads();
    }
    if (i < ngroups) woff[i] = carry + buf[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) carry += buf[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x == 0) flags->total_len = carry;
}

__global__ void k_scan_blocks(uint32_t nblk, const uint32_t* __restrict__ len, const unsigned long long* __restrict__ woff,
                              unsigned long long* __restrict__ off) {
  __shared__ unsigned long long buf[1024];
  uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long v = m < nblk ? len[m] : 0;
  buf[threadIdx.x] = v;
  __syncthreads();
  for (uint32_t d = 1; d < blockDim.x; d <<= 1) {
    unsigned long long x = threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
    __syncthreads();
    buf[threadIdx.x] += x;
    __syncthreads();
  }
  if (m < nblk) off[m] = woff[blockIdx.x] + buf[threadIdx.x] - v;
}

// start leaf of the whole shard per end leaf (for the neighbouring rank); one thread
__global__ void k_shard_map(uint32_t nblk, uint32_t nleaves_end, const uint8_t* __restrict__ bs_start,
                            const uint8_t* __restrict__ bs_merged, const uint8_t* __restrict__ bs_mstart, uint32_t Lc,
                            uint8_t* __restrict__ map_out, uint32_t* __restrict__ constant_out) {
  if (blockIdx.x || threadIdx.x) return;
  uint32_t first = nblk;
  for (uint32_t m = 0; m < nblk; ++m) if (bs_merged[m]) { first = m; break; }
  if (first < nblk) {
    uint32_t leaf = bs_mstart[first];
    for (uint32_t i = first; i-- > 0;) leaf = bs_start[(size_t)i * Lc + leaf];
    for (uint32_t e = 0; e < nleaves_end; ++e) map_out[e] = (uint8_t)leaf;
    *constant_out = 1;
  } else {
    for (uint32_t e = 0; e < nleaves_end; ++e) {
      uint32_t leaf = e;
      for (uint32_t i = nblk; i-- > 0;) leaf = bs_start[(size_t)i * Lc + leaf];
      map_out[e] = (uint8_t)leaf;
    }
    *constant_out = 0;
  }
}

// ------------------------------------------------------------------------------ k_fixtail
// With the block's end leaf E known, walk the unresolved tail pieces [merge_piece, npieces) again
// and write their end leaf and offset in the convention of k_backlen:
//   output offset of piece p inside its block = ctot[m] - pcum[p].
// The same walk counts the constants of the pieces it visits, plus those of every eighth block's first
// piece: a sample from which k_emit's job slots are dimensioned.
template <bool WIDE>
__global__ void k_fixtail(const uint8_t* __restrict__ in, uint64_t n, uint64_t blk, uint32_t nblk,
                          const uint16_t* __restrict__ chk, const uint8_t* __restrict__ E,
                          const uint32_t* __restrict__ len, const uint16_t* __restrict__ merge_piece,
                          const uint32_t* __restrict__ ctot, PieceRec* __restrict__ prec, Flags* flags, DevTables T) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  Lds L = stage_tables<WIDE>(T, smem);
  uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= nblk) return;
  const uint64_t bstart = (uint64_t)m * blk;
  const uint64_t bend = bstart + blk < n ? bstart + blk : n;
  const uint32_t npieces = (uint32_t)((bend - bstart + PIECE - 1) / PIECE);
  const uint32_t mp = merge_piece[m] < npieces ? merge_piece[m] : npieces;
  const uint64_t piece0 = bstart >> 6;
  uint32_t leaf = (uint32_t)E[m] * 4, suffix = 0, kmax = 0, ksum = 0;
  const int32_t base = (int32_t)ctot[m] - (int32_t)len[m];
  const uint32_t ntail = npieces - mp, nvisit = ntail + (mp > 0 && (m & 7u) == 0 ? 1u : 0u);
  for (uint32_t i = 0; i < nvisit; ++i) {
    const bool tail = i < ntail;
    const uint32_t p = tail ? npieces - 1 - i : 0u;
    if (!tail) leaf = (prec[piece0].leaf & 0xFFu) * 4;   // sample only: piece 0 was finished by k_backlen
    const uint64_t pstart = bstart + (uint64_t)p * PIECE;
    const int plen = (int)(bend - pstart < PIECE ? bend - pstart : PIECE);
    uint32_t w[16], bo[BOW];
    load_piece(in, n, pstart, w);
    const uint32_t hh = reinterpret_cast<const uint32_t*>(chk)[pstart >> 6];
    if constexpr (WIDE) piece_forward(w, hh & 0xFFFFu, L, bo);
    else piece_forward2(w, hh & 0xFFFFu, hh >> 16, 0xFFFF0000u, bo);
    mask_tail(bo, plen, T.nullrow);
    const uint32_t leaf_end = leaf >> 2;
    uint32_t lmid = 0, shi = 0, k = 0;
    suffix += walk_len_mid<WIDE, true>(bo, leaf, lmid, shi, L, T, &k);
    kmax = k > kmax ? k : kmax; ksum += k;
    if (tail) prec[piece0 + p] = PieceRec{base + (int32_t)suffix, rec_word(leaf_end, lmid, shi)};
  }
  if (kmax > flags->emit_kmax) atomicMax(&flags->emit_kmax, kmax);
  if ((m & 255u) == 0 && nvisit) { atomicAdd(&flags->emit_ksum, ksum); atomicAdd(&flags->emit_kn, nvisit); }
}

// ------------------------------------------------------------------------------- k_emit
// One lane = one 64-byte piece; one wave-iteration = 64 consecutive pieces = 4 KiB of contiguous
// input and a contiguous stretch of output, assembled in an LDS staging buffer and flushed with
// aligned 16-byte stores.  Every lane re-derives the back rows of its piece, then walks them
// backward from its resolved end leaf *and places the output in the same sweep*: the piece's output
// range is known beforehand from the piece records (start = its own record, end = the next piece's
// start), so the write cursor simply runs down from the end.  Copied input bytes are stored by
// the walking lane; constants are noted as (entry, cursor) pairs in lane-private job slots and
// copied after the sweep, one lane per constant (the slots are handed out wave-wide, densely).
// Persistent workgroups: tables are staged once per CU.
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

// One step of the sweep (program without wide entries; piece_sweep2 in kx_sweeps.inc runs two chains
// of them per lane): 6 VALU + 1 LDS read + 1 LDS write,
//   a = row(t) + leaf;  e = lds[a];  leaf = e & 0x3FC;  o -= e.byte3;
//   lds8[o | e<<31] = input byte t          (a step that copies nothing addresses out of range: dropped)
//   if (e.bit23) { job slot[wave count + rank among such lanes] <- o<<16 | a }   (clamped to the last slot)
// The step in plain C++ for the other program shapes (one chain).  CONSTS: 0 = note a job, 1 = copy the
// constant in place (short constants, or the second attempt of a lane whose job slots overflowed).
template <int T_, bool WIDE, int CONSTS>
__device__ __forceinline__ void emit_step_gen(const uint32_t (&bo)[BOW], const uint32_t (&w)[16], uint32_t& leaf, uint32_t& o,
                                              uint32_t& jb, uint32_t jlim, const Lds& L, const DevTables& T) {
  const uint32_t a = BO_GET_DEP(bo, T_, leaf) + leaf;
  const uint32_t e = L.w(a);
  leaf = E_LEAF4(e);
  const uint32_t dl = ent_dlen<WIDE>(e, a, L, T), cp = E_COPY(e);
  o -= dl;
  uint8_t* stg0 = const_cast<uint8_t*>(L.base);
  if (cp) stg0[o] = (uint8_t)BYTE_AT_DEP(w, T_, e);
  const bool hc = E_HASCONST(e) != 0;
  if constexpr (CONSTS == 0) {
    const unsigned long long m = __ballot(hc);   // (inactive lanes vote 0)
    if (hc) {
      const uint32_t slot = jb + 4 * __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      *reinterpret_cast<uint32_t*>(stg0 + (slot < jlim ? slot : jlim)) = (o << 16) | a;
    }
    jb += 4 * (uint32_t)__popcll(m);
  } else if (hc) {
    const uint32_t cl = dl - cp, src = ent_off<WIDE>(e, a, L, T);
    for (uint32_t i = 0; i < cl; ++i) stg0[o + cp + i] = L.pb(src + i);
  }
}

template <int WAVES, bool WIDE>   // WIDE: the GENERAL instance (wide back entries / class shift), compiler-scheduled sweeps
__global__ __launch_bounds__(WAVES * 64) void k_emit(const uint8_t* __restrict__ in, uint64_t n, uint64_t blk, uint32_t blk_shift,
                                                     uint64_t npieces_total, const uint16_t* __restrict__ chk,
                                                     const PieceRec* __restrict__ prec,
                                                     const uint32_t* __restrict__ ctot,
                                                     const unsigned long long* __restrict__ off, Flags* __restrict__ flags,
                                                     uint32_t stgb, uint32_t jbytes, uint32_t maxcnt, uint32_t nup, uint32_t init_shift,
                                                     uint32_t init_leaf, int is_first, uint8_t* __restrict__ out, DevTables T) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  Lds L = stage_tables<WIDE>(T, smem);
  // the sweeps address LDS absolutely (16-bit row offsets, staging cursors): the image must sit at LDS address 0
  if ((uint32_t)(uintptr_t)smem != 0u) __builtin_trap();
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t stga = ((T.packed_words + 3) & ~3u) * 4 + wave * (stgb + 16 + jbytes);   // LDS address of this wave's staging area
  uint8_t* stg = (uint8_t*)smem + stga;
  const uint32_t jarea = stga + stgb + 16;
  if (is_first && blockIdx.x == 0 && threadIdx.x < init_shift) out[threadIdx.x] = L.pb(T.init_off[init_leaf] + threadIdx.x);
  if (is_first && blockIdx.x == 0 && init_shift > blockDim.x)
    for (uint32_t i = blockDim.x + threadIdx.x; i < init_shift; i += blockDim.x) out[i] = L.pb(T.init_off[init_leaf] + i);
  const uint64_t nwi = (npieces_total + 63) / 64;
  const uint64_t oend_all = (uint64_t)init_shift + flags->total_len;
  auto piece_ostart = [&](uint64_t pc, const PieceRec& r) {
    const uint64_t m = blk_shift < 64u ? (pc * PIECE) >> blk_shift : pc * PIECE / blk;   // (a 64-bit division costs ≈100 instructions)
    return (uint64_t)init_shift + off[m] + (uint64_t)(int64_t)((int32_t)ctot[m] - r.cum);
  };
  for (uint64_t it = (uint64_t)blockIdx.x * WAVES + wave; it < nwi; it += (uint64_t)gridDim.x * WAVES) {
    const uint64_t piece = it * 64 + lane;
    const bool valid = piece < npieces_total;
    const uint64_t pstart = piece * PIECE;
    const int plen = valid ? (int)(n - pstart < PIECE ? n - pstart : PIECE) : 0;
    uint32_t w[16], bo[BOW];
    load_piece(in, n, valid ? pstart : n, w);
    const PieceRec rec = valid ? prec[piece] : PieceRec{0, 0};
    uint64_t ostart = valid ? piece_ostart(piece, rec) : 0, oend;
    {   // end of the piece's output = start of the next piece's
      const uint32_t lo = __shfl_down((uint32_t)ostart, 1), hi = __shfl_down((uint32_t)(ostart >> 32), 1);
      oend = ((uint64_t)hi << 32) | lo;
      if (valid && (lane == 63 || piece + 1 == npieces_total))
        oend = piece + 1 == npieces_total ? oend_all : piece_ostart(piece + 1, prec[piece + 1]);
      if (!valid) oend = ostart;
    }
    const uint32_t hh = valid ? reinterpret_cast<const uint32_t*>(chk)[piece] : T.deadh * 0x10001u;
    if constexpr (WIDE) piece_forward(w, hh & 0xFFFFu, L, bo);
    else piece_forward2(w, hh & 0xFFFFu, hh >> 16, 0xFFFF0000u, bo);
    mask_tail(bo, plen, T.nullrow);
    // Touch the next iteration's input, checkpoints and records now, so that their HBM latency passes during the
    // sweep.  The three destination registers stay reserved (they are operands of the wait below), the loads are
    // older than every later vector-memory operation of the wave, so the compiler's own waits stay valid.
    uint32_t pf0, pf1, pf2;
    {
      const uint64_t npiece = piece + (uint64_t)gridDim.x * WAVES * 64;
      const uint64_t tp = npiece < npieces_total ? npiece : piece;
      asm volatile("global_load_dword %0, %3, off\n\tglobal_load_dword %1, %4, off\n\tglobal_load_dword %2, %5, off"
                   : "=&v"(pf0), "=&v"(pf1), "=&v"(pf2)
                   : "v"(in + (valid ? tp * PIECE : 0)), "v"(reinterpret_cast<const uint32_t*>(chk) + (valid ? tp : 0)), "v"(prec + (valid ? tp : 0)));
    }
    const uint32_t leaf_end4 = (rec.leaf & 0xFFu) * 4, leaf_mid4 = ((rec.leaf >> 8) & 0xFFu) * 4, len_hi = rec.leaf >> 16;
    const unsigned long long vmask = __ballot(valid);
    const uint32_t nvalid = (uint32_t)__popcll(vmask);
    uint32_t first = 0;
    bool pf_pending = true;
    while (first < nvalid) {
      const uint64_t gs = __shfl(ostart, first);
      const uint64_t abase = gs & ~15ull;
      const bool fits = valid && lane >= first && (oend - abase) <= (uint64_t)stgb && len_hi != 0xFFFFu;
      const unsigned long long fm = __ballot(fits) >> first;
      uint32_t cnt = fm == ~0ull ? 64u - first : (uint32_t)__builtin_ctzll(~fm);   // leading run of fitting lanes
      if (cnt > maxcnt) cnt = maxcnt;   // (the job slots are dimensioned for maxcnt lanes per round)
      if (cnt == 0) {
        // a single piece larger than the staging area: its lane writes straight to global memory
        if (lane == first) {
          uint32_t leaf = leaf_end4;
          static_for<0, PIECE>([&](auto ic) {
            constexpr int t = PIECE - 1 - decltype(ic)::value;
            const uint32_t a = BO_GET_DEP(bo, t, leaf) + leaf;
            bo_set(bo, t, a);
            leaf = E_LEAF4(L.w(a));
          });
          uint64_t o = ostart;
          static_for<0, PIECE>([&](auto ic) {
            constexpr int t = decltype(ic)::value;
            const uint32_t a = BO_GET_DEP(bo, t, (uint32_t)o);
            const uint32_t e = L.w(a);
            const uint32_t cp = E_COPY(e), cl = ent_dlen<WIDE>(e, a, L, T) - cp, src = ent_off<WIDE>(e, a, L, T);
            if (cp) out[o++] = (uint8_t)BYTE_AT_DEP(w, t, a);
            for (uint32_t i = 0; i < cl; ++i) out[o++] = L.pb(src + i);
          });
        }
        first += 1;
        continue;
      }
      const uint32_t lastl = first + cnt - 1;
      const bool active = lane >= first && lane <= lastl;
      const uint32_t oe = stga + (uint32_t)(oend - abase);   // LDS address one past the piece's staged output
      // job slots: jbytes/4 four-byte slots per wave and round, handed out in the order the constants are met
      const uint32_t jlim = __builtin_amdgcn_readfirstlane(jarea + jbytes - 4);
      uint32_t jb = __builtin_amdgcn_readfirstlane(jarea);
      // sweep: copied bytes into staging, constants into the job slots
      if (active) {
        if constexpr (!WIDE) {
          piece_sweep2(bo, w, leaf_end4, oe, leaf_mid4, oe - len_hi, jb, jlim);
        } else {
          uint32_t leaf = leaf_end4, o = oe;
          static_for<0, PIECE>([&](auto ic) {
            constexpr int t = PIECE - 1 - decltype(ic)::value;
            emit_step_gen<t, WIDE, 0>(bo, w, leaf, o, jb, jlim, L, T);
          });
        }
      }
      jb = __builtin_amdgcn_readfirstlane(__shfl(jb, (int)first));   // (lane `first` took part in the sweep; lane 0 may not have)
      uint32_t njobs = (jb - jarea) >> 2;
      if (njobs > jbytes / 4) {
        // more constants than slots (the last slot was overwritten): sweep once more, copying constants in place
        if (lane == first) atomicAdd(&flags->emit_ovf, cnt);
        if (active) {
          uint32_t leaf = leaf_end4, o = oe, jq = 0;
          static_for<0, PIECE>([&](auto ic) {
            constexpr int t = PIECE - 1 - decltype(ic)::value;
            emit_step_gen<t, WIDE, 1>(bo, w, leaf, o, jq, 0u, L, T);
          });
        }
        njobs = 0;
      }
      wave_lds_fence();
      // constants: one lane per job; the first 16 bytes of a constant are fetched in one go (four independent
      // unaligned reads; what lies beyond a constant is read and dropped), so a job is three dependent LDS round trips, not seven
      auto job_of = [&](uint32_t j, uint32_t& l, uint8_t*& dst, const uint8_t*& sp) {
        const uint32_t jw = *reinterpret_cast<const uint32_t*>((const uint8_t*)smem + jarea + 4 * j);
        const uint32_t a = jw & 0xFFFFu, e = L.w(a);
        const uint32_t cp = E_COPY(e), src = ent_off<WIDE>(e, a, L, T);
        l = ent_dlen<WIDE>(e, a, L, T) - cp;
        dst = (uint8_t*)smem + stga + (((jw >> 16) - stga) & 0xFFFFu) + cp;
        sp = L.base + L.pool + src;
      };
      if (nup > 1) {   // (uniform) programs with constants longer than 4 bytes
        for (uint32_t j = lane; j < njobs; j += 64) {
          uint32_t l; uint8_t* dst; const uint8_t* sp;
          job_of(j, l, dst, sp);
          uint32_t v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) __builtin_memcpy(&v[q], sp + 4 * q, 4);
          const uint32_t nw = l >> 2;
#pragma unroll
          for (int q = 0; q < 4; ++q) if ((uint32_t)q < nw) __builtin_memcpy(dst + 4 * q, &v[q], 4);
          if (nw < 4) {
            uint32_t vt = nw == 0 ? v[0] : nw == 1 ? v[1] : nw == 2 ? v[2] : v[3];
            uint8_t* dt = dst + 4 * nw;
            if (l & 2) { const uint16_t h2 = (uint16_t)vt; __builtin_memcpy(dt, &h2, 2); dt += 2; vt >>= 16; }
            if (l & 1) *dt = (uint8_t)vt;
          } else {
            uint32_t i = 16;
            for (; i + 4 <= l; i += 4) { uint32_t x; __builtin_memcpy(&x, sp + i, 4); __builtin_memcpy(dst + i, &x, 4); }
            if (l & 2) { uint16_t x; __builtin_memcpy(&x, sp + i, 2); __builtin_memcpy(dst + i, &x, 2); i += 2; }
            if (l & 1) dst[i] = sp[i];
          }
        }
      } else {         // every constant fits one dword
        for (uint32_t j = lane; j < njobs; j += 64) {
          uint32_t l; uint8_t* dst; const uint8_t* sp;
          job_of(j, l, dst, sp);
          uint32_t vt;
          __builtin_memcpy(&vt, sp, 4);
          if (l == 4) __builtin_memcpy(dst, &vt, 4);
          else {
            uint8_t* dt = dst;
            if (l & 2) { const uint16_t h2 = (uint16_t)vt; __builtin_memcpy(dt, &h2, 2); dt += 2; vt >>= 16; }
            if (l & 1) *dt = (uint8_t)vt;
          }
        }
      }
      wave_lds_fence();
      if (pf_pending) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf0), "+v"(pf1), "+v"(pf2)); pf_pending = false; }
      // flush [gs, ge): partial head and tail windows bytewise, everything between as aligned 16 B
      const uint64_t ge = __shfl(oend, lastl);
      const uint64_t fs = (gs + 15) & ~15ull, fe = ge & ~15ull;
      if (fs >= fe) {
        for (uint64_t x = gs + lane; x < ge; x += 64) out[x] = stg[x - abase];
      } else {
        if (gs + lane < fs) out[gs + lane] = stg[gs + lane - abase];
        if (fe + lane < ge) out[fe + lane] = stg[fe + lane - abase];
        for (uint64_t x = fs + (uint64_t)lane * 16; x < fe; x += 1024)
          *reinterpret_cast<uint4*>(out + x) = *reinterpret_cast<const uint4*>(stg + (x - abase));
      }
      wave_lds_fence();
      first = lastl + 1;
    }
    if (pf_pending) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf0), "+v"(pf1), "+v"(pf2));   // (a wave-iteration without valid pieces)
  }
}

// =============================================================================== host side
struct Stage {
  uint32_t nstates = 0, nclasses = 0, q0 = 0, maxleaves = 1;
  std::vector<uint8_t> h_nleaves, h_fin_leaf;          // host copies for the control path
  std::vector<uint32_t> h_init_off, h_init_len;
  std::vector<uint8_t> h_pool;
  void* d_all = nullptr;                               // one allocation holding every table
  DevTables T{};
  size_t lds_bytes = 0;                                // packed table image
  size_t sync_lds_bytes = 0;                           // 0 = sync tables stay in global memory
  uint32_t emit_nup = 1;                               // k_emit: dwords of a constant fetched up front (longest constant / 4, at most 4)
  bool general = false;                                   // wide back entries or > 64 byte classes: run the GENERAL kernel instances
};

struct Arena {  // grow-only device workspace, reused across runs
  char* base = nullptr; size_t cap = 0, used = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return 0;
    if (base) (void)hipFree(base);
    base = nullptr; cap = 0;
    size_t want = bytes + bytes / 8 + (1 << 20);
    HIPCHECK(hipMalloc((void**)&base, want));
    cap = want;
    return 0;
  }
  void reset() { used = 0; }
  template <typename Ty> Ty* take(size_t count) {
    used = (used + 255) & ~(size_t)255;
    Ty* p = reinterpret_cast<Ty*>(base + used);
    used += count * sizeof(Ty);
    return p;
  }
};

}  // namespace

struct kx_program {
  std::vector<Stage> stages;
  kx_config cfg{0, 512, 0, 0, 0};
  std::vector<Arena*> arenas;      // shard workspaces (grow-only), handed out to open shards and taken back
  void* stagebuf[2] = {nullptr, nullptr};   // ping-pong buffers between pipeline stages
  size_t stagecap[2] = {0, 0};
  hipEvent_t ev[KX_NKERNELS + 1] = {};
  bool have_events = false;
  int ncu = 256;

};

struct kx_shard {
  kx_program* prog; Stage* st; uint32_t stage; Arena* arena = nullptr;
  const uint8_t* in; uint64_t n; int is_first, is_last; hipStream_t stream;
  uint64_t seg; uint32_t nseg, nblk, Lc, ngroups;
  // workspace
  uint64_t* seg_pos; uint16_t* seg_state; uint16_t* chk; Flags* flags;
  uint8_t *bs_start, *bs_merged, *bs_mstart, *E, *d_map; uint32_t *bs_len, *len, *d_const, *ctot;
  PieceRec* prec; uint16_t* merge_piece;
  unsigned long long *off, *wsum, *woff;
  // control state mirrored on the host
  Flags hflags{}; uint64_t head_len = 0; bool have_end = false; uint64_t out_len = 0; uint32_t init_shift = 0, init_leaf = 0;
  uint32_t endId() const { return (hflags.end_state - OFF_FWD) / (st->nclasses * 4); }
  kx_stats stats{};
};

namespace {

size_t pad4(size_t x) { return (x + 3) & ~(size_t)3; }

int parseStage(const uint8_t*& c, const uint8_t* end, Stage& S) {
  if (c + 64 > end) return setErr(KX_E_BLOB, "truncated stage header");
  uint32_t h[16];
  memcpy(h, c, 64); c += 64;
  if (h[0] != KXP_STAGE_MAGIC) return setErr(KX_E_BLOB, "bad stage magic");
  const uint32_t nstates = h[1], C = h[2], q0 = h[3], nactions = h[5], nops = h[6], nconsts = h[7], cpl = h[8];
  const uint32_t Lm = h[9], nback = h[10], npc = h[11], pcpl = h[12], nsync = h[13];
  const size_t sc = (size_t)nstates * C;
  const uint8_t* cls = c; c += 256;
  const uint16_t* delta = (const uint16_t*)c; c += pad4(sc * 2);
  c += sc * 4;                                   // act        (register form: not used by the engine)
  c += (size_t)nstates * 4;                      // final_act
  c += ((size_t)nactions + 1) * 4;               // act_off
  c += (size_t)nops * 8;                         // ops
  c += ((size_t)nconsts + 1) * 4; c += pad4(cpl);  // consts
  const uint32_t* pback = (const uint32_t*)c; c += sc * 4;
  const uint8_t* nleaves = c; c += pad4(nstates);
  const uint8_t* fin_leaf = c; c += pad4(nstates);
  const uint32_t* back = (const uint32_t*)c; c += (size_t)nback * Lm * 4;
  const uint32_t* pcoff = (const uint32_t*)c; c += ((size_t)npc + 1) * 4;
  const uint8_t* pcpool = c; c += pad4(pcpl);
  const uint32_t* init_const = (const uint32_t*)c; c += (size_t)Lm * 4;
  const uint32_t* sync_next = (const uint32_t*)c; c += (size_t)nsync * C * 4;
  const uint32_t* sync_state = (const uint32_t*)c; c += (size_t)nsync * 4;
  if (c > end) return setErr(KX_E_BLOB, "truncated stage body");
  if (nstates == 0 || C == 0 || Lm == 0 || Lm > 254 || (size_t)(nstates + 1) * C * 4 + OFF_FWD > 0xFFF0)
    return setErr(KX_E_BLOB, "program outside engine limits (states x classes / leaves)");

  S.nstates = nstates; S.nclasses = C; S.q0 = q0; S.maxleaves = Lm;
  // ragged back rows: row b keeps entries up to its last live leaf.  Compact entry (see DevTables);
  // entries that do not fit (≥127 appended bytes, pool offset ≥ 8 KiB) escape to the wide side tables.
  std::vector<uint32_t> rowoff(nback), ent, wlen, woff;
  auto pushEnt = [&](uint32_t parent, uint32_t copy, uint32_t dlen, uint32_t poff) {
    const bool wide = dlen >= 127 || poff >= (1u << 13);
    if (wide) S.general = true;
    { const uint32_t need = (dlen - copy + 3) / 4; if (need > S.emit_nup) S.emit_nup = need < 4 ? need : 4; }
    ent.push_back((copy ? 0u : 1u) | (parent << 2) | ((wide ? 0u : poff) << 10) | (dlen > copy ? 1u << 23 : 0u) |
                  ((wide ? 127u : dlen) << 24));
    wlen.push_back(dlen); woff.push_back(poff);
  };
  for (uint32_t b = 0; b < nback; ++b) {
    uint32_t rl = 1;
    for (uint32_t j = 0; j < Lm; ++j) if (back[(size_t)b * Lm + j] != KXP_DEAD_LEAF) rl = j + 1;
    rowoff[b] = (uint32_t)ent.size();
    for (uint32_t j = 0; j < rl; ++j) {
      uint32_t e = back[(size_t)b * Lm + j];
      if (e == KXP_DEAD_LEAF) { pushEnt(0, 0, 0, 0); continue; }
      uint32_t parent = e & 0xFF, copy = (e >> 8) & 1, pc = e >> 9;
      uint32_t clen = pcoff[pc + 1] - pcoff[pc];
      if (clen + copy >= (1u << 23)) return setErr(KX_E_BLOB, "path constant too long");
      pushEnt(parent, copy, clen + copy, pcoff[pc]);
    }
  }
  // identity row (parent = leaf, nothing appended): steps past the end of a partial piece point here;
  // a second, all-zero row pads the table so that a probe on a dead path stays inside it
  const uint32_t nullrow_idx = (uint32_t)ent.size();
  for (uint32_t j = 0; j < Lm; ++j) pushEnt(j, 0, 0, 0);
  for (uint32_t j = 0; j < Lm; ++j) pushEnt(0, 0, 0, 0);
  // LDS image: cls4 (256 B, at LDS address 0: the hand-scheduled sweeps read it with no base) | fwd | ent | pool
  const uint32_t off_fwd = OFF_FWD, fwd_bytes = (nstates + 1) * C * 4;
  const uint32_t off_ent = off_fwd + fwd_bytes;
  if ((size_t)off_ent + ent.size() * 4 > 65532)
    return setErr(KX_E_BLOB, "program tables exceed the engine's 16-bit LDS addressing (states x classes + path table > 64 KiB)");
  const uint32_t deadh = off_fwd + nstates * C * 4;    // handle of the absorbing "no transition" state
  const uint32_t cshift = C <= 64 ? 0u : 2u;           // class·4 fits a byte up to 64 classes
  if (cshift) S.general = true;
  std::vector<uint32_t> packed(OFF_FWD / 4 + (size_t)(nstates + 1) * C);
  for (int b = 0; b < 256; ++b) ((uint8_t*)packed.data())[b] = (uint8_t)(cshift ? cls[b] : cls[b] * 4);
  {
    uint32_t* fwd = packed.data() + OFF_FWD / 4;
    for (uint32_t q = 0; q < nstates; ++q)
      for (uint32_t k = 0; k < C; ++k) {
        uint16_t d = delta[(size_t)q * C + k];
        fwd[(size_t)q * C + k] = d == KXP_NO_STATE ? deadh
                                 : ((off_fwd + (uint32_t)d * C * 4) | ((off_ent + rowoff[pback[(size_t)q * C + k]] * 4) << 16));
      }
    for (uint32_t k = 0; k < C; ++k) fwd[(size_t)nstates * C + k] = deadh;
  }
  packed.insert(packed.end(), ent.begin(), ent.end());
  const uint32_t off_pool = (uint32_t)packed.size() * 4;
  packed.resize(packed.size() + pad4(pcpl + 4) / 4, 0);
  memcpy((char*)packed.data() + off_pool, pcpool, pcpl);
  const uint32_t off_cls = 0;
  S.lds_bytes = packed.size() * 4;
  if (S.lds_bytes > 150 * 1024) return setErr(KX_E_BLOB, "program tables exceed the LDS budget (150 KiB)");
  const uint32_t nullrow = off_ent + nullrow_idx * 4;

  S.h_nleaves.assign(nleaves, nleaves + nstates); S.h_nleaves.push_back(1);
  S.h_fin_leaf.assign(fin_leaf, fin_leaf + nstates); S.h_fin_leaf.push_back(KXP_NO_LEAF);
  S.h_pool.assign(pcpool, pcpool + pcpl);
  S.h_init_off.resize(Lm); S.h_init_len.resize(Lm);
  for (uint32_t j = 0; j < Lm; ++j) { S.h_init_off[j] = pcoff[init_const[j]]; S.h_init_len[j] = pcoff[init_const[j] + 1] - pcoff[init_const[j]]; }

  // synchronising automaton, renumbered: undecided subsets first, decided ones absorbing; one extra
  // absorbing row stands for "subset construction was capped here"
  if (nsync == 0 || nsync >= 0xFFF0) return setErr(KX_E_BLOB, "bad synchronising automaton");
  std::vector<uint32_t> newid(nsync);
  uint32_t nmulti = 0;
  for (uint32_t i = 0; i < nsync; ++i) if (sync_state[i] == KXP_SYNC_MULTI) newid[i] = nmulti++;
  { uint32_t t = nmulti; for (uint32_t i = 0; i < nsync; ++i) if (sync_state[i] != KXP_SYNC_MULTI) newid[i] = t++; }
  std::vector<uint16_t> sync16(((size_t)(nsync + 1) * (C + 1) + 1) & ~(size_t)1, 0);
  for (uint32_t i = 0; i < nsync; ++i) {
    const bool multi = sync_state[i] == KXP_SYNC_MULTI;
    for (uint32_t k = 0; k < C; ++k) {
      uint32_t nx = sync_next[(size_t)i * C + k];
      sync16[(size_t)newid[i] * C + k] = (uint16_t)(!multi ? newid[i] : nx == KXP_SYNC_UNKNOWN ? nsync : newid[nx]);
    }
    uint32_t st = sync_state[i];
    sync16[(size_t)(nsync + 1) * C + newid[i]] = (uint16_t)(st == KXP_SYNC_MULTI ? 0xFFFF : st == KXP_SYNC_EMPTY ? 0xFFFE : st);
  }
  for (uint32_t k = 0; k < C; ++k) sync16[(size_t)nsync * C + k] = (uint16_t)nsync;
  sync16[(size_t)(nsync + 1) * C + nsync] = 0xFFFD;
  const size_t sync_bytes = sync16.size() * 2;
  S.sync_lds_bytes = sync_bytes + 256 <= 64 * 1024 ? sync_bytes + 256 : 0;

  // one device allocation: packed | cls | nleaves | fin_leaf | sync16 | init_off | init_len
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t o_packed = 0, o_cls = o_packed + al(packed.size() * 4), o_nl = o_cls + 256, o_fl = o_nl + al(nstates + 1),
         o_sn = o_fl + al(nstates + 1), o_io = o_sn + al(sync_bytes),
         o_il = o_io + al(Lm * 4), o_wl = o_il + al(Lm * 4), o_wo = o_wl + al(wlen.size() * 4),
         total = o_wo + al(woff.size() * 4);
  std::vector<uint8_t> img(total, 0);
  memcpy(&img[o_packed], packed.data(), packed.size() * 4);
  memcpy(&img[o_cls], cls, 256);
  memcpy(&img[o_nl], S.h_nleaves.data(), nstates + 1);
  memcpy(&img[o_fl], S.h_fin_leaf.data(), nstates + 1);
  memcpy(&img[o_sn], sync16.data(), sync_bytes);
  memcpy(&img[o_io], S.h_init_off.data(), Lm * 4);
  memcpy(&img[o_il], S.h_init_len.data(), Lm * 4);
  memcpy(&img[o_wl], wlen.data(), wlen.size() * 4);
  memcpy(&img[o_wo], woff.data(), woff.size() * 4);
  HIPCHECK(hipMalloc(&S.d_all, total));
  HIPCHECK(hipMemcpy(S.d_all, img.data(), total, hipMemcpyHostToDevice));
  char* d = (char*)S.d_all;
  DevTables& T = S.T;
  T.packed = (const uint32_t*)(d + o_packed); T.packed_words = (uint32_t)packed.size();
  T.off_ent = off_ent; T.off_pool = off_pool; T.off_cls = off_cls;
  T.wlen = (const uint32_t*)(d + o_wl); T.woff = (const uint32_t*)(d + o_wo);
T.nstates = nstates; T.nclasses = C; T.off_fwd = off_fwd; T.q0h = off_fwd + q0 * C * 4; T.maxleaves = Lm; T.deadh = deadh; T.nullrow = nullrow; T.cshift = cshift;
  T.cls = (const uint8_t*)(d + o_cls); T.nleaves = (const uint8_t*)(d + o_nl); T.fin_leaf = (const uint8_t*)(d + o_fl);
  T.sync16 = (const uint16_t*)(d + o_sn); T.nsync = nsync; T.sync_multi = nmulti; T.sync_words = (uint32_t)(sync_bytes / 4);
  T.init_off = (const uint32_t*)(d + o_io); T.init_len = (const uint32_t*)(d + o_il);
  return 0;
}

int setLds(const void* fn, size_t bytes) {
  HIPCHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return 0;
}

float evMs(hipEvent_t a, hipEvent_t b) { float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms; }

}  // namespace

// ================================================================================= C ABI
extern "C" {

const char* kx_last_error(void) { return g_err.c_str(); }

int kx_load(const void* blob, size_t blob_len, kx_program** prog) {
  if (!blob || !prog) return setErr(KX_E_ARG, "null argument");
  const uint8_t* b = (const uint8_t*)blob;
  if (blob_len < 20 || memcmp(b, KXP_MAGIC, 8)) return setErr(KX_E_BLOB, "not a KXP blob");
  uint32_t ver, ns, il;
  memcpy(&ver, b + 8, 4); memcpy(&ns, b + 12, 4); memcpy(&il, b + 16, 4);
  if (ver != KXP_VERSION || ns == 0) return setErr(KX_E_BLOB, "unsupported KXP version");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return setErr(KX_E_HIP, "no HIP device available: the engine has no CPU fallback");
  const uint8_t* c = b + 20 + pad4(il);
  auto* p = new kx_program;
  p->stages.resize(ns);
  for (uint32_t s = 0; s < ns; ++s) {
    int rc = parseStage(c, b + blob_len, p->stages[s]);
    if (rc) { kx_free(p); return rc; }
  }
  size_t lds = 0, slds = 0;
  for (auto& s : p->stages) { lds = s.lds_bytes > lds ? s.lds_bytes : lds; slds = s.sync_lds_bytes > slds ? s.sync_lds_bytes : slds; }
  hipDeviceProp_t prop;
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) p->ncu = prop.multiProcessorCount;
  // k_emit: as many waves per CU as the LDS left over by the tables allows (one workgroup per CU)
  const size_t lds_cap = 160 * 1024;
  const size_t tab = (lds + 15) & ~(size_t)15;
  if (tab + 4 * (size_t)EMIT_WAVE_LDS_MIN > lds_cap) { kx_free(p); return setErr(KX_E_BLOB, "program tables leave no LDS for the output stage"); }
  int rc = setLds((const void*)k_forward<false>, lds); if (rc) { kx_free(p); return rc; }
  rc = setLds((const void*)k_forward<true>, lds); if (rc) { kx_free(p); return rc; }
  rc = setLds((const void*)k_head<false>, lds); if (rc) { kx_free(p); return rc; }
  rc = setLds((const void*)k_head<true>, lds); if (rc) { kx_free(p); return rc; }
  rc = setLds((const void*)k_backlen<32, false>, tab + 1024 * 80); if (rc) { kx_free(p); return rc; }
  rc = setLds((const void*)k_backlen<32, true>, tab + 1024 * 80); if (rc) { kx_free(p); return rc; }
  rc = setLds((const void*)k_backlen<256, false>, tab + 1024 * 80); if (rc) { kx_free(p); return rc; }
  rc = setLds((const void*)k_backlen<256, true>, tab + 1024 * 80); if (rc) { kx_free(p); return rc; }
  rc = setLds((const void*)k_fixtail<false>, lds); if (rc) { kx_free(p); return rc; }
  rc = setLds((const void*)k_fixtail<true>, lds); if (rc) { kx_free(p); return rc; }
  if (slds) { rc = setLds((const void*)k_sync<true>, slds); if (rc) { kx_free(p); return rc; } }
  const size_t elds = lds_cap;   // waves, staging and job slots per wave are chosen per run (output/input ratio)
  bool anywide = false;
  for (auto& s : p->stages) anywide = anywide || s.general;
#define KX_EMIT_ATTR(WV) (rc = setLds((const void*)k_emit<WV, false>, elds), rc ? rc : (anywide ? setLds((const void*)k_emit<WV, true>, elds) : 0))
  rc = KX_EMIT_ATTR(12); rc = rc ? rc : KX_EMIT_ATTR(8); rc = rc ? rc : KX_EMIT_ATTR(4);
#undef KX_EMIT_ATTR
  if (rc) { kx_free(p); return rc; }
  *prog = p;
  return 0;
}

void kx_free(kx_program* p) {
  if (!p) return;
  for (auto& s : p->stages) if (s.d_all) (void)hipFree(s.d_all);
  for (Arena* a : p->arenas) { if (a->base) (void)hipFree(a->base); delete a; }
  for (int i = 0; i < 2; ++i) if (p->stagebuf[i]) (void)hipFree(p->stagebuf[i]);
  if (p->have_events) for (auto& e : p->ev) (void)hipEventDestroy(e);
  delete p;
}

int kx_set_config(kx_program* p, const kx_config* cfg) {
  if (!p || !cfg) return setErr(KX_E_ARG, "null argument");
  kx_config c = *cfg;
  if (c.block_threads == 0) c.block_threads = 512;
  if (c.segment_bytes % PIECE || c.segment_bytes > 65535u * PIECE || c.block_threads % 64 || c.block_threads > 1024 ||
      (c.block_threads & (c.block_threads - 1)))
    return setErr(KX_E_ARG, "segment_bytes: multiple of 64 (≤ 4 MiB); block_threads: power of two in [64, 1024]");
  p->cfg = c;
  return 0;
}

uint32_t kx_num_stages(const kx_program* p) { return p ? (uint32_t)p->stages.size() : 0; }

int kx_shard_begin(kx_program* p, uint32_t stage, const void* d_in, size_t n, int is_first, int is_last,
                   void* stream, kx_shard** out) {
  if (!p || !out || stage >= p->stages.size()) return setErr(KX_E_ARG, "bad program/stage");
  if (n && (!d_in || ((uintptr_t)d_in & 15))) return setErr(KX_E_ARG, "input must be 16-byte aligned");
  auto* s = new kx_shard;
  s->prog = p; s->st = &p->stages[stage]; s->stage = stage;
  s->in = (const uint8_t*)d_in; s->n = n; s->is_first = is_first; s->is_last = is_last; s->stream = (hipStream_t)stream;
  s->seg = p->cfg.segment_bytes;
  if (s->seg == 0) s->seg = (n >> 14) >= (512u << 10) ? 16384 : (n >> 13) >= (512u << 10) ? 8192 : 4096;   // keep ≥ 512 Ki lanes
  uint64_t ns = n ? (n + s->seg - 1) / s->seg : 1;
  if (ns > 0x7FFFFFF0ull) { delete s; return setErr(KX_E_ARG, "input too large for the configured segment size"); }
  s->nseg = (uint32_t)ns; s->nblk = s->nseg; s->Lc = s->st->maxleaves;
  const uint32_t bt = p->cfg.block_threads;
  s->ngroups = (s->nblk + bt - 1) / bt;
  // workspace layout
  if (p->arenas.empty()) p->arenas.push_back(new Arena);
  s->arena = p->arenas.back(); p->arenas.pop_back();
  Arena& A = *s->arena;
  size_t need = 4096 + (size_t)s->nseg * (8 + 2) + (n / HALF + 32) * 2 + sizeof(Flags) +
                (size_t)s->nblk * ((size_t)s->Lc * 5 + 3 + 4 + 8 + 2 + 4) + (size_t)s->ngroups * 16 + KX_MAX_LEAVES + 64 +
                (n / PIECE + 16) * 8 + 256 * 32;
  int rc = A.reserve(need);
  if (rc) { p->arenas.push_back(s->arena); delete s; return rc; }
  A.reset();
  s->seg_pos = A.take<uint64_t>(s->nseg); s->seg_state = A.take<uint16_t>(s->nseg);
  s->chk = A.take<uint16_t>(n / HALF + 32); s->flags = A.take<Flags>(1);
  s->bs_start = A.take<uint8_t>((size_t)s->nblk * s->Lc); s->bs_len = A.take<uint32_t>((size_t)s->nblk * s->Lc);
  s->bs_merged = A.take<uint8_t>(s->nblk); s->bs_mstart = A.take<uint8_t>(s->nblk); s->E = A.take<uint8_t>(s->nblk);
  s->len = A.take<uint32_t>(s->nblk); s->off = A.take<unsigned long long>(s->nblk);
  s->wsum = A.take<unsigned long long>(s->ngroups); s->woff = A.take<unsigned long long>(s->ngroups);
  s->d_map = A.take<uint8_t>(KX_MAX_LEAVES); s->d_const = A.take<uint32_t>(4);
  s->prec = A.take<PieceRec>(n / PIECE + 16);
  s->merge_piece = A.take<uint16_t>(s->nblk); s->ctot = A.take<uint32_t>(s->nblk);
  if (!p->have_events) {
    for (auto& e : p->ev) if (hipEventCreate(&e) != hipSuccess) { p->arenas.push_back(s->arena); delete s; return setErr(KX_E_HIP, "hipEventCreate failed"); }
    p->have_events = true;
  }
  s->stats.in_bytes = n;
  *out = s;
  return 0;
}

static int readFlags(kx_shard* s) {
  HIPCHECK(hipMemcpyAsync(&s->hflags, s->flags, sizeof(Flags), hipMemcpyDeviceToHost, s->stream));
  HIPCHECK(hipStreamSynchronize(s->stream));
  return 0;
}

static void fillFwd(kx_shard* s, kx_fwd_summary* out) {
  out->synced = s->have_end ? 1 : 0;
  out->end_state = s->have_end ? s->endId() : 0;
  out->head_len = s->head_len;
  out->fail_pos = s->hflags.fail_pos;
}

int kx_shard_forward(kx_shard* s, kx_fwd_summary* out) {
  if (!s || !out) return setErr(KX_E_ARG, "null argument");
  kx_program* p = s->prog; Stage& S = *s->st;
  const uint32_t bt = p->cfg.block_threads;
  const bool timing = p->cfg.collect_timing;
  Flags init{}; init.fail_pos = NOFAIL; init.end_state = OFF_FWD + S.q0 * S.nclasses * 4; init.first_merged = 0xFFFFFFFFu;
  HIPCHECK(hipMemcpyAsync(s->flags, &init, sizeof(Flags), hipMemcpyHostToDevice, s->stream));
  if (s->n == 0) {  // nothing to scan: the state entering byte 0 is also the end state
    s->hflags = init; s->head_len = 0; s->have_end = s->is_first != 0;
    fillFwd(s, out);
    return 0;
  }
  const uint32_t grid = (s->nseg + bt - 1) / bt;
  if (timing) HIPCHECK(hipEventRecord(p->ev[0], s->stream));
  if (S.sync_lds_bytes)
    hipLaunchKernelGGL((k_sync<true>), dim3(grid), dim3(bt), S.sync_lds_bytes, s->stream, s->in, s->n, s->seg, s->nseg,
                       s->is_first, s->seg_pos, s->seg_state, s->flags, S.T);
  else
    hipLaunchKernelGGL((k_sync<false>), dim3(grid), dim3(bt), 0, s->stream, s->in, s->n, s->seg, s->nseg, s->is_first,
                       s->seg_pos, s->seg_state, s->flags, S.T);
  if (timing) HIPCHECK(hipEventRecord(p->ev[1], s->stream));
  hipLaunchKernelGGL(S.general ? k_forward<true> : k_forward<false>, dim3(grid), dim3(bt), S.lds_bytes, s->stream, s->in, s->n, s->nseg, s->seg_pos,
                     s->seg_state, s->chk, s->flags, S.T);
  if (timing) HIPCHECK(hipEventRecord(p->ev[2], s->stream));
  HIPCHECK(hipGetLastError());
  int rc = readFlags(s);
  if (rc) return rc;
  if (timing) { s->stats.kernel_ms[KX_K_SYNC] = evMs(p->ev[0], p->ev[1]); s->stats.kernel_ms[KX_K_FORWARD] = evMs(p->ev[1], p->ev[2]); }
  s->stats.unsynced_segments = s->hflags.unsynced;
  // head = bytes before the first synchronised segment (0 on a first shard)
  if (s->is_first) { s->head_len = 0; s->have_end = true; }
  else {
    // find the first synced segment: unsynced ones are rare, probe from the front
    uint64_t hp = UNSYNC;
    for (uint32_t k = 0; k < s->nseg; ++k) {
      HIPCHECK(hipMemcpy(&hp, s->seg_pos + k, 8, hipMemcpyDeviceToHost));
      if (hp != UNSYNC) break;
    }
    s->head_len = hp == UNSYNC ? s->n : hp;
    s->have_end = hp != UNSYNC;
  }
  fillFwd(s, out);
  return 0;
}

int kx_shard_fix_head(kx_shard* s, uint32_t incoming_state, kx_fwd_summary* out) {
  if (!s || !out) return setErr(KX_E_ARG, "null argument");
  kx_program* p = s->prog; Stage& S = *s->st;
  if (!s->is_first && s->head_len > 0) {
    if (incoming_state > S.nstates) return setErr(KX_E_ARG, "incoming state out of range");
    const bool timing = p->cfg.collect_timing;
    if (timing) HIPCHECK(hipEventRecord(p->ev[0], s->stream));
    hipLaunchKernelGGL(S.general ? k_head<true> : k_head<false>, dim3(1), dim3(64), S.lds_bytes, s->stream, s->in, s->n, s->head_len,
                       OFF_FWD + incoming_state * S.nclasses * 4, s->chk, s->flags, S.T);
    if (timing) HIPCHECK(hipEventRecord(p->ev[1], s->stream));
    HIPCHECK(hipGetLastError());
    int rc = readFlags(s);
    if (rc) return rc;
    if (timing) s->stats.kernel_ms[KX_K_HEAD] = evMs(p->ev[0], p->ev[1]);
  } else if (!s->is_first && s->n == 0) {
    s->hflags.end_state = OFF_FWD + incoming_state * S.nclasses * 4;
  }
  s->have_end = true;
  if (s->is_last && s->hflags.fail_pos == NOFAIL && S.h_fin_leaf[s->endId()] == KXP_NO_LEAF)
    s->hflags.fail_pos = s->n;  // end of input in a non-final state (C.hs NextI fallback → FailI)
  fillFwd(s, out);
  return 0;
}

int kx_shard_backward(kx_shard* s, kx_bwd_summary* out) {
  if (!s || !out) return setErr(KX_E_ARG, "null argument");
  kx_program* p = s->prog; Stage& S = *s->st;
  memset(out, 0, sizeof *out);
  const uint32_t nle = S.h_nleaves[s->endId()];
  out->nleaves = nle;
  if (s->n == 0) {  // identity map
    out->constant = 0;
    for (uint32_t e = 0; e < nle; ++e) out->start_leaf[e] = (uint8_t)e;
    return 0;
  }
  const uint32_t bt = p->cfg.block_threads;
  const uint32_t grid = (s->nblk + bt - 1) / bt;
  const bool timing = p->cfg.collect_timing;
  if (timing) HIPCHECK(hipEventRecord(p->ev[0], s->stream));
  const size_t blds = ((S.lds_bytes + 15) & ~(size_t)15) + (size_t)bt * 80;
#define KX_LAUNCH_BACKLEN(MC, WD)                                                                                  \
  hipLaunchKernelGGL((k_backlen<MC, WD>), dim3(grid), dim3(bt), blds, s->stream, s->in, s->n, s->seg, s->nblk, s->chk, \
                     s->flags, s->is_last, s->bs_start, s->bs_len, s->bs_merged, s->bs_mstart, s->Lc, s->prec,        \
                     s->merge_piece, s->ctot, S.T)
  if (s->Lc <= 32) { if (S.general) KX_LAUNCH_BACKLEN(32, true); else KX_LAUNCH_BACKLEN(32, false); }
  else { if (S.general) KX_LAUNCH_BACKLEN(256, true); else KX_LAUNCH_BACKLEN(256, false); }
#undef KX_LAUNCH_BACKLEN
  if (timing) HIPCHECK(hipEventRecord(p->ev[1], s->stream));
  HIPCHECK(hipGetLastError());
  if (!s->is_first) {  // the neighbouring rank needs our start leaf as a function of our end leaf
    hipLaunchKernelGGL(k_shard_map, dim3(1), dim3(64), 0, s->stream, s->nblk, nle, s->bs_start, s->bs_merged,
                       s->bs_mstart, s->Lc, s->d_map, s->d_const);
    HIPCHECK(hipMemcpyAsync(out->start_leaf, s->d_map, nle, hipMemcpyDeviceToHost, s->stream));
    HIPCHECK(hipMemcpyAsync(&out->constant, s->d_const, 4, hipMemcpyDeviceToHost, s->stream));
  }
  HIPCHECK(hipStreamSynchronize(s->stream));
  if (timing) s->stats.kernel_ms[KX_K_BACKLEN] = evMs(p->ev[0], p->ev[1]);
  return 0;
}

int kx_shard_resolve(kx_shard* s, uint32_t end_leaf, uint64_t* out_len) {
  if (!s || !out_len) return setErr(KX_E_ARG, "null argument");
  kx_program* p = s->prog; Stage& S = *s->st;
  if (s->is_last) end_leaf = S.h_fin_leaf[s->endId()];
  if (end_leaf >= S.h_nleaves[s->endId()]) return setErr(KX_E_ARG, "end leaf out of range");
  if (s->n == 0) {
    s->init_shift = s->is_first ? S.h_init_len[end_leaf] : 0;
    s->out_len = s->init_shift;
    *out_len = s->out_len;
    s->init_leaf = end_leaf;  // no input: the start leaf is the end leaf
    return 0;
  }
  const uint32_t bt = p->cfg.block_threads;
  const bool timing = p->cfg.collect_timing;
  if (timing) HIPCHECK(hipEventRecord(p->ev[0], s->stream));
  hipLaunchKernelGGL(k_resolve, dim3(s->ngroups), dim3(bt), 0, s->stream, s->nblk, end_leaf, s->bs_start, s->bs_len,
                     s->bs_merged, s->bs_mstart, s->Lc, s->E, s->len, s->wsum);
  hipLaunchKernelGGL(k_scan_groups, dim3(1), dim3(1024), 0, s->stream, s->ngroups, s->wsum, s->woff, s->flags);
  hipLaunchKernelGGL(k_scan_blocks, dim3(s->ngroups), dim3(bt), 0, s->stream, s->nblk, s->len, s->woff, s->off);
  if (S.general)
    hipLaunchKernelGGL((k_fixtail<true>), dim3(s->ngroups), dim3(bt), S.lds_bytes, s->stream, s->in, s->n, s->seg, s->nblk,
                       s->chk, s->E, s->len, s->merge_piece, s->ctot, s->prec, s->flags, S.T);
  else
    hipLaunchKernelGGL((k_fixtail<false>), dim3(s->ngroups), dim3(bt), S.lds_bytes, s->stream, s->in, s->n, s->seg, s->nblk,
                       s->chk, s->E, s->len, s->merge_piece, s->ctot, s->prec, s->flags, S.T);
  if (timing) HIPCHECK(hipEventRecord(p->ev[1], s->stream));
  HIPCHECK(hipGetLastError());
  uint8_t e0 = 0;
  HIPCHECK(hipMemcpyAsync(&e0, s->E, 1, hipMemcpyDeviceToHost, s->stream));
  int rc = readFlags(s);
  if (rc) return rc;
  if (timing) s->stats.kernel_ms[KX_K_RESOLVE] = evMs(p->ev[0], p->ev[1]);
  s->init_shift = 0;
  if (s->is_first) {
    uint8_t l0 = 0;
    HIPCHECK(hipMemcpy(&l0, s->bs_start + (size_t)e0, 1, hipMemcpyDeviceToHost));
    s->init_shift = S.h_init_len[l0];
    s->init_leaf = l0;
  }
  s->out_len = s->hflags.total_len + s->init_shift;
  *out_len = s->out_len;
  return 0;
}

int kx_shard_emit(kx_shard* s, void* d_out, size_t cap) {
  if (!s) return setErr(KX_E_ARG, "null argument");
  kx_program* p = s->prog; Stage& S = *s->st;
  if (cap < s->out_len || (s->out_len && !d_out)) return setErr(KX_E_CAPACITY, "output buffer too small");
  if ((uintptr_t)d_out & 15) return setErr(KX_E_ARG, "output buffer must be 16-byte aligned");
  s->stats.out_bytes = s->out_len;
  if (s->n == 0) {
    if (s->init_shift) {
      uint32_t leaf = s->init_leaf;
      HIPCHECK(hipMemcpyAsync(d_out, S.h_pool.data() + S.h_init_off[leaf], s->init_shift, hipMemcpyHostToDevice, s->stream));
      HIPCHECK(hipStreamSynchronize(s->stream));
    }
    return 0;
  }
  const bool timing = p->cfg.collect_timing;
  const uint64_t npieces = (s->n + PIECE - 1) / PIECE;
  const uint64_t nwi = (npieces + 63) / 64;
  // Waves per CU (one persistent workgroup each), staging bytes and job-slot bytes per wave.  A wave-iteration
  // turns 4 KiB of input into ratio x 4 KiB of output; when that fits the staging buffer the iteration is one
  // sweep + one flush, otherwise every extra round repeats the sweep code for the lanes it did not cover.  The
  // kernel needs more than 128 VGPRs, so 12 waves (three per SIMD) is the most a CU holds.
  const size_t lds_cap = 160 * 1024, tab = (S.lds_bytes + 15) & ~(size_t)15;
  int W = tab + 12 * (size_t)EMIT_WAVE_LDS_MIN <= lds_cap ? 12 : tab + 8 * (size_t)EMIT_WAVE_LDS_MIN <= lds_cap ? 8 : 4;
  if (const char* ev = getenv("KX_EMIT_WAVES")) { int v = atoi(ev); if ((v == 4 || v == 8 || v == 12) && v <= W) W = v; }
  const size_t per_wave = ((lds_cap - tab) / W) & ~(size_t)255;
  // Rounds per iteration R = 1, 2, …: a round covers up to ceil(64/R) lanes, needs staging for their output
  // (average piece output x 1.08) and job slots for their constants (k_fixtail's sample: average per piece x 1.25
  // + 1.5, at most the sampled maximum + 1); take the smallest R that fits the wave's LDS.
  const double olen = 64.0 * (double)s->out_len / (double)(s->n ? s->n : 1);
  const uint32_t kmax = s->hflags.emit_kmax < 64 ? s->hflags.emit_kmax : 64;
  double kper = s->hflags.emit_kn ? 1.25 * (double)s->hflags.emit_ksum / (double)s->hflags.emit_kn + 1.5 : (double)kmax + 1;
  if (kper > kmax + 1) kper = kmax + 1;
  size_t stgb = 0, jbytes = 0; uint32_t maxcnt = 64;
  for (uint32_t R = 1; R <= 64; ++R) {
    const uint32_t c = (64 + R - 1) / R;
    size_t st = (((size_t)(c * olen * 1.08) + 64) + 255) & ~(size_t)255, jb = ((size_t)(c * kper + 16) * 4 + 255) & ~(size_t)255;
    if (st < 1024) st = 1024;
    if (st + 16 + jb <= per_wave || c == 1) {
      maxcnt = c; jbytes = jb; stgb = per_wave - 16 - jb;   // the rest goes to staging (bigger pieces fit, fewer oversize ones)
      break;
    }
  }
  if (const char* ev = getenv("KX_EMIT_STG")) { size_t v = (size_t)atoi(ev) & ~(size_t)255; if (v >= 1024 && v + 16 + jbytes <= per_wave) stgb = v; }
  if (stgb > 49152) stgb = 49152;   // job records keep 16 bits of the staging address
  stgb &= ~(size_t)15;
  uint64_t wantg = (nwi + W - 1) / W;
  const uint32_t grid = (uint32_t)(wantg < (uint64_t)p->ncu ? wantg : (uint64_t)p->ncu);
  const size_t elds = tab + (size_t)W * (stgb + 16 + jbytes);
  const uint32_t blk_shift = (s->seg & (s->seg - 1)) == 0 ? (uint32_t)__builtin_ctzll(s->seg) : 0xFFu;
  if (timing) HIPCHECK(hipEventRecord(p->ev[0], s->stream));
#define KX_LAUNCH_EMIT(WV, WD)                                                                                       \
  hipLaunchKernelGGL((k_emit<WV, WD>), dim3(grid), dim3(WV * 64), elds, s->stream, s->in, s->n, s->seg, blk_shift, npieces, s->chk, \
                     s->prec, s->ctot, s->off, s->flags, (uint32_t)stgb, (uint32_t)jbytes, maxcnt, S.emit_nup, s->init_shift, s->init_leaf, s->is_first, (uint8_t*)d_out, S.T)
  if (S.general) { if (W == 12) KX_LAUNCH_EMIT(12, true); else if (W == 8) KX_LAUNCH_EMIT(8, true); else KX_LAUNCH_EMIT(4, true); }
  else { if (W == 12) KX_LAUNCH_EMIT(12, false); else if (W == 8) KX_LAUNCH_EMIT(8, false); else KX_LAUNCH_EMIT(4, false); }
#undef KX_LAUNCH_EMIT
  if (timing) HIPCHECK(hipEventRecord(p->ev[1], s->stream));
  HIPCHECK(hipGetLastError());
  uint32_t ovf = 0;
  const bool debug = getenv("KX_DEBUG") != nullptr;
  if (timing || debug) HIPCHECK(hipMemcpyAsync(&ovf, &s->flags->emit_ovf, 4, hipMemcpyDeviceToHost, s->stream));
  HIPCHECK(hipStreamSynchronize(s->stream));
  if (timing) s->stats.kernel_ms[KX_K_EMIT] = evMs(p->ev[0], p->ev[1]);
  s->stats.emit_overflow_pieces = ovf;
  if (debug) fprintf(stderr, "[kx] emit: W=%d stg=%zu jobs=%zu maxcnt=%u kmax=%u kavg=%.2f olen=%.1f overflow=%u of %llu pieces\n", W, stgb, jbytes, maxcnt, kmax, s->hflags.emit_kn ? (double)s->hflags.emit_ksum / s->hflags.emit_kn : -1.0, olen, ovf, (unsigned long long)npieces);
  return 0;
}

void kx_shard_stats(kx_shard* s, kx_stats* st) {
  if (!s || !st) return;
  *st = s->stats;
  st->fail_pos = s->hflags.fail_pos; st->fail_stage = s->stage;
}

void kx_shard_end(kx_shard* s) {
  if (!s) return;
  if (s->arena) s->prog->arenas.push_back(s->arena);
  delete s;
}

// whole program on one device: every stage is a single shard that is both first and last
static int runPipeline(kx_program* p, const void* d_in, size_t n, void* d_out, size_t cap, bool alloc_final,
                       void** d_final, size_t* out_len, kx_stats* stats, void* stream) {
  kx_stats total{}; total.fail_pos = NOFAIL; total.in_bytes = n;
  const void* cur = d_in; size_t curn = n;
  void* aligned_copy = nullptr;
  if (n && ((uintptr_t)d_in & 15)) {  // engine wants 16-byte aligned pieces
    HIPCHECK(hipMalloc(&aligned_copy, n));
    HIPCHECK(hipMemcpyAsync(aligned_copy, d_in, n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    cur = aligned_copy;
  }
  int rc = 0;
  const uint32_t ns = (uint32_t)p->stages.size();
  for (uint32_t st = 0; st < ns && rc == 0; ++st) {
    kx_shard* s = nullptr;
    rc = kx_shard_begin(p, st, cur, curn, 1, 1, stream, &s);
    if (rc) break;
    kx_fwd_summary fs; kx_bwd_summary bs; uint64_t ol = 0;
    rc = kx_shard_forward(s, &fs);
    if (!rc) rc = kx_shard_fix_head(s, 0, &fs);
    if (!rc && fs.fail_pos != NOFAIL) { total.fail_pos = fs.fail_pos; total.fail_stage = st; rc = KX_MATCH_ERROR; }
    if (!rc) rc = kx_shard_backward(s, &bs);
    if (!rc) rc = kx_shard_resolve(s, 0, &ol);
    if (!rc) {
      void* dst = nullptr; size_t dcap = 0;
      if (st + 1 == ns) {
        *out_len = ol;
        if (alloc_final) {
          if (hipMalloc(d_final, ol ? ol : 16) != hipSuccess) rc = setErr(KX_E_HIP, "hipMalloc(output) failed");
          dst = *d_final; dcap = ol;
        } else {
          if (ol > cap || (ol && !d_out)) rc = setErr(KX_E_CAPACITY, "output buffer too small");
          dst = d_out; dcap = cap;
        }
      } else {
        int slot = st & 1;
        if (p->stagecap[slot] < ol + 16) {
          if (p->stagebuf[slot]) (void)hipFree(p->stagebuf[slot]);
          p->stagebuf[slot] = nullptr; p->stagecap[slot] = 0;
          if (hipMalloc(&p->stagebuf[slot], ol + ol / 8 + 4096) != hipSuccess) rc = setErr(KX_E_HIP, "hipMalloc(stage buffer) failed");
          else p->stagecap[slot] = ol + ol / 8 + 4096;
        }
        dst = p->stagebuf[slot]; dcap = p->stagecap[slot];
      }
      if (!rc) rc = kx_shard_emit(s, dst, dcap);
      if (!rc) { cur = dst; curn = ol; }
    }
    kx_stats ss; kx_shard_stats(s, &ss);
    for (int i = 0; i < KX_NKERNELS; ++i) { total.kernel_ms[i] = ss.kernel_ms[i]; total.total_ms += ss.kernel_ms[i]; }
    total.unsynced_segments += ss.unsynced_segments;
    total.out_bytes = ol;
    kx_shard_end(s);
  }
  if (aligned_copy) (void)hipFree(aligned_copy);
  if (stats) *stats = total;
  return rc;
}

int kx_run_device(kx_program* p, const void* d_in, size_t n, void* d_out, size_t cap, size_t* out_len, kx_stats* stats,
                  void* stream) {
  if (!p || !out_len) return setErr(KX_E_ARG, "null argument");
  return runPipeline(p, d_in, n, d_out, cap, false, nullptr, out_len, stats, stream);
}

int kx_run_host(kx_program* p, const void* in, size_t n, void** out, size_t* out_len, kx_stats* stats) {
  if (!p || !out || !out_len) return setErr(KX_E_ARG, "null argument");
  *out = nullptr; *out_len = 0;
  void* d_in = nullptr; void* d_out = nullptr;
  if (n) { HIPCHECK(hipMalloc(&d_in, n)); HIPCHECK(hipMemcpy(d_in, in, n, hipMemcpyHostToDevice)); }
  size_t ol = 0;
  int rc = runPipeline(p, d_in, n, nullptr, 0, true, &d_out, &ol, stats, nullptr);
  if (rc == 0) {
    *out = malloc(ol ? ol : 1);
    if (!*out) rc = setErr(KX_E_IO, "out of host memory");
    else if (ol && hipMemcpy(*out, d_out, ol, hipMemcpyDeviceToHost) != hipSuccess) rc = setErr(KX_E_HIP, "D2H copy failed");
    if (!rc) *out_len = ol;
  }
  if (d_out) (void)hipFree(d_out);
  if (d_in) (void)hipFree(d_in);
  return rc;
}

void kx_host_free(void* p) { free(p); }

// stdin → pinned host → HBM → engine → pinned host → stdout.  Reads and H2D copies overlap (two
// pinned staging buffers), and so do D2H copies and writes.  The whole input is resident in HBM
// while the engine runs (SURVEY §8f rank 1 asks for windowed streaming on top; not built yet).
// ------------------------------------------------------------------ streaming: stdin → HBM → stdout
// The input is processed in windows (default 4 GiB; everything up to that is one window).  A window is one
// shard of the sharded protocol: its start state is the previous window's end state; its end leaf is the
// next window's start leaf, so a window's output is placed once the next window's backward summary is known
// (immediately when that summary does not depend on the end leaf — "constant" — which is the rule; otherwise
// windows queue up until one is, or until the last).  Multi-stage programs chain such streams: the output
// windows of stage i are the input windows of stage i+1.  Reads overlap the host→device copies and
// device→host copies overlap the writes through pinned staging buffers.
namespace {

struct FdStream {
  kx_program* p = nullptr; int in_fd = -1, out_fd = -1;
  size_t CH = 64u << 20, window = 0;
  char* pin[4] = {nullptr, nullptr, nullptr, nullptr};   // 0,1: input staging; 2,3: output staging
  hipEvent_t ev[4] = {}; bool have_ev = false;
  hipStream_t cs = nullptr;
  int rk = 0; size_t carry = 0; bool eof = false;          // reader: chunk read ahead in pin[rk]
  struct Win { kx_shard* sh; char* d_in; size_t n; kx_bwd_summary bs; };
  struct StageQ { std::vector<Win> pend; uint32_t end_state = 0; bool first = true; uint64_t consumed = 0; };
  std::vector<StageQ> q;
  kx_stats total{};

  ~FdStream() {
    for (auto& sq : q) for (auto& w : sq.pend) { if (w.sh) kx_shard_end(w.sh); if (w.d_in) (void)hipFree(w.d_in); }
    for (auto& b : pin) if (b) (void)hipHostFree(b);
    if (have_ev) for (auto& e : ev) (void)hipEventDestroy(e);
    if (cs) (void)hipStreamDestroy(cs);
  }
  int init() {
    HIPCHECK(hipStreamCreate(&cs));
    for (auto& e : ev) HIPCHECK(hipEventCreate(&e));
    have_ev = true;
    for (auto& b : pin) HIPCHECK(hipHostMalloc((void**)&b, CH, hipHostMallocDefault));
    q.resize(p->stages.size());
    total.fail_pos = NOFAIL;
    return 0;
  }
  int readChunk(char* dst, size_t* got) {
    size_t g = 0;
    while (g < CH) {
      ssize_t r = read(in_fd, dst + g, CH - g);
      if (r < 0) { if (errno == EINTR) continue; return setErr(KX_E_IO, "read failed"); }
      if (r == 0) break;
      g += (size_t)r;
    }
    *got = g;
    return 0;
  }
  // next window of the input: device buffer (owned by the caller afterwards), its length, whether it is the last
  int nextWindow(char** d_out, size_t* n_out, bool* last) {
    size_t cap = window, n = 0;
    struct stat stt;
    if (carry == 0 && !eof && fstat(in_fd, &stt) == 0 && S_ISREG(stt.st_mode)) {   // regular file: do not over-allocate
      off_t pos = lseek(in_fd, 0, SEEK_CUR);
      if (pos >= 0 && (size_t)(stt.st_size - pos) + CH < cap) cap = ((size_t)(stt.st_size > pos ? stt.st_size - pos : 0) / CH + 1) * CH;
    }
    char* d = nullptr;
    HIPCHECK(hipMalloc((void**)&d, cap + 16));
    while (n < cap && !eof) {
      size_t got = carry;
      if (!got) {
        if (hipEventSynchronize(ev[rk]) != hipSuccess) { (void)hipFree(d); return setErr(KX_E_HIP, "hipEventSynchronize failed"); }
        int rc = readChunk(pin[rk], &got);
        if (rc) { (void)hipFree(d); return rc; }
      }
      carry = 0;
      if (got) {
        if (hipMemcpyAsync(d + n, pin[rk], got, hipMemcpyHostToDevice, cs) != hipSuccess || hipEventRecord(ev[rk], cs) != hipSuccess) {
          (void)hipFree(d); return setErr(KX_E_HIP, "host to device copy failed");
        }
        n += got; rk ^= 1;
      }
      if (got < CH) eof = true;
    }
    if (!eof) {   // the window is full: read one chunk ahead to learn whether anything follows
      if (hipEventSynchronize(ev[rk]) != hipSuccess) { (void)hipFree(d); return setErr(KX_E_HIP, "hipEventSynchronize failed"); }
      int rc = readChunk(pin[rk], &carry);
      if (rc) { (void)hipFree(d); return rc; }
      if (carry == 0) eof = true;
    }
    if (hipStreamSynchronize(cs) != hipSuccess) { (void)hipFree(d); return setErr(KX_E_HIP, "hipStreamSynchronize failed"); }
    *d_out = d; *n_out = n; *last = eof && carry == 0;
    return 0;
  }
  int writeOut(const char* d, size_t ol) {
    size_t done = 0, issued = 0, clen[2] = {0, 0};
    int k = 0;
    if (issued < ol) {
      clen[0] = ol - issued < CH ? ol - issued : CH;
      HIPCHECK(hipMemcpyAsync(pin[2], d + issued, clen[0], hipMemcpyDeviceToHost, cs));
      HIPCHECK(hipEventRecord(ev[2], cs));
      issued += clen[0];
    }
    while (done < ol) {
      if (issued < ol) {   // next chunk flies while this one is written
        clen[k ^ 1] = ol - issued < CH ? ol - issued : CH;
        HIPCHECK(hipMemcpyAsync(pin[2 + (k ^ 1)], d + issued, clen[k ^ 1], hipMemcpyDeviceToHost, cs));
        HIPCHECK(hipEventRecord(ev[2 + (k ^ 1)], cs));
        issued += clen[k ^ 1];
      }
      HIPCHECK(hipEventSynchronize(ev[2 + k]));
      size_t w = 0;
      while (w < clen[k]) {
        ssize_t r = write(out_fd, pin[2 + k] + w, clen[k] - w);
        if (r < 0) { if (errno == EINTR) continue; return setErr(KX_E_IO, "write failed"); }
        w += (size_t)r;
      }
      done += clen[k];
      k ^= 1;
    }
    return 0;
  }
  // window `d` (ownership taken) enters stage `st`
  int push(uint32_t st, char* d, size_t n, bool last) {
    StageQ& sq = q[st];
    Win w{nullptr, d, n, {}};
    int rc = kx_shard_begin(p, st, d, n, sq.first ? 1 : 0, last ? 1 : 0, nullptr, &w.sh);
    if (rc) { (void)hipFree(d); return rc; }
    sq.pend.push_back(w);
    Win& W = sq.pend.back();
    kx_fwd_summary fs;
    rc = kx_shard_forward(W.sh, &fs);
    if (!rc) rc = kx_shard_fix_head(W.sh, sq.first ? 0 : sq.end_state, &fs);
    if (rc) return rc;
    if (fs.fail_pos != NOFAIL) { total.fail_pos = sq.consumed + fs.fail_pos; total.fail_stage = st; return KX_MATCH_ERROR; }
    sq.end_state = fs.end_state; sq.first = false; sq.consumed += n;
    rc = kx_shard_backward(W.sh, &W.bs);
    if (rc) return rc;
    accumulate(W.sh);
    // which queued windows now know the leaf they end in?
    size_t nres = 0;
    std::vector<uint32_t> ends(sq.pend.size(), 0);
    if (last) nres = sq.pend.size();                     // the last window resolves its own end leaf from the final state
    else if (W.bs.constant) nres = sq.pend.size() - 1;   // everything before this window
    if (nres == 0) {
      if (sq.pend.size() > 64) return setErr(KX_E_ARG, "the program's output stays undetermined across more than 64 windows; use a larger window");
      return 0;
    }
    for (size_t j = nres; j-- > 0;) {
      if (j + 1 < sq.pend.size()) { const Win& nx = sq.pend[j + 1]; ends[j] = nx.bs.start_leaf[nx.bs.constant ? 0 : ends[j + 1]]; }
    }
    for (size_t j = 0; j < nres; ++j) {
      Win& R = sq.pend[j];
      const bool rlast = last && j + 1 == sq.pend.size();
      uint64_t ol = 0;
      rc = kx_shard_resolve(R.sh, ends[j], &ol);
      if (rc) return rc;
      char* dout = nullptr;
      HIPCHECK(hipMalloc((void**)&dout, (size_t)ol + 16));
      rc = kx_shard_emit(R.sh, dout, (size_t)ol);
      accumulate(R.sh);
      kx_shard_end(R.sh); R.sh = nullptr;
      (void)hipFree(R.d_in); R.d_in = nullptr;
      if (rc) { (void)hipFree(dout); return rc; }
      if (st + 1 < q.size()) {
        if (ol || rlast) rc = push(st + 1, dout, (size_t)ol, rlast);   // (push takes the buffer)
        else (void)hipFree(dout);
      } else {
        total.out_bytes += ol;
        rc = writeOut(dout, (size_t)ol);
        (void)hipFree(dout);
      }
      if (rc) return rc;
    }
    sq.pend.erase(sq.pend.begin(), sq.pend.begin() + nres);
    return 0;
  }
  void accumulate(kx_shard* sh) {
    kx_stats ss; kx_shard_stats(sh, &ss);
    total.unsynced_segments = total.unsynced_segments > ss.unsynced_segments ? total.unsynced_segments : ss.unsynced_segments;
  }
};

}  // namespace

extern "C" int kx_run_fd(kx_program* p, int in_fd, int out_fd, kx_stats* stats) {
  if (!p) return setErr(KX_E_ARG, "null argument");
  FdStream fsr;
  fsr.p = p; fsr.in_fd = in_fd; fsr.out_fd = out_fd;
  size_t window = p->cfg.window_bytes ? p->cfg.window_bytes : (size_t)4 << 30;
  if (const char* ev = getenv("KX_WINDOW_BYTES")) { long long v = atoll(ev); if (v > 0) window = (size_t)v; }
  if (window < 4096) window = 4096;
  if (fsr.CH > window) fsr.CH = (window + 4095) & ~(size_t)4095;
  fsr.window = (window + fsr.CH - 1) / fsr.CH * fsr.CH;
  int rc = fsr.init();
  bool last = false;
  while (!rc && !last) {
    char* d = nullptr; size_t n = 0;
    rc = fsr.nextWindow(&d, &n, &last);
    if (!rc) { fsr.total.in_bytes += n; rc = fsr.push(0, d, n, last); }
  }
  if (stats) *stats = fsr.total;
  return rc;
}

}  // extern "C"
