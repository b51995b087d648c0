/* kx_oracle.c — CPU ORACLE (test infrastructure, never shipped, never timed as product).
 *
 * Sequential restatement of the reference's run-time hot path over a KXP
 * blob (include/kxp_format.h).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product path (libkxhip.so)
 * never does.
 *
 * kxo_run  — the register form, executed the way the reference's generated
 *   `match<K>()` + crt/crt.c do:
 *     state loop                  generated code, Backends/C.hs:72-83,486-493
 *     readnext/consume            crt/crt.c:285-312 (window logic elided: input is in memory)
 *     reset/append/appendarray/concat   crt/crt.c:171-193,229-259 (grow ×2 from 32 KiB)
 *     outputconst/outputarray/output    crt/crt.c:217-227,261-283 (16 KiB flush granularity)
 *     accept / "Match error at input symbol %zu!"   Backends/C.hs:76-81
 *   SST meaning: src/KMC/SymbolicSST.hs:400-446 (parallel assignment, output
 *   register streamed, final update at EOF).
 * kxo_run_path — the path form evaluated sequentially (forward states,
 *   backward leaf resolution, in-order emission): a CPU cross-check that the
 *   compiler's two forms describe the same function, and the algorithm the HIP
 *   engine parallelises.
 *
 * PARITY PIN: checked in tests/test_oracle_golden.py against every action-free
 * I/O vector the reference's tests hold (test/test_compiled, test/test_simulated,
 * test/Tests/Regression.hs, README add-commas) and against outputs of the
 * reference's Perl twins on the reference's sample data (tests/golden/).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/kxp_format.h"

typedef struct {
  uint32_t nstates, nclasses, q0, nregs, nactions, nops, nconsts, constpool_len;
  uint32_t maxleaves, nback, npconsts, pconstpool_len, nsync, sync_complete, actions;
  const uint8_t* cls;
  const uint16_t* delta;
  const uint32_t *act, *final_act, *act_off, *ops, *const_off;
  const uint8_t* constpool;
  const uint32_t* pback;
  const uint8_t *nleaves, *fin_leaf;
  const uint32_t *back, *pconst_off;
  const uint8_t* pconstpool;
  const uint32_t *init_const, *sync_next, *sync_state;
  uint32_t ntables; const uint8_t* tables; /* [ntables][256]: AppendTblI's tables (IL.hs:44,84) */
} stage_t;

typedef struct { uint32_t nstages; stage_t* st; } prog_t;

static size_t pad4(size_t n) { return (n + 3) & ~(size_t)3; }

static int parse_blob(const uint8_t* b, size_t len, prog_t* p) {
  if (len < 20 || memcmp(b, KXP_MAGIC, 8)) return -1;
  const uint8_t* c = b + 8;
  uint32_t ver, ns, il;
  memcpy(&ver, c, 4); memcpy(&ns, c + 4, 4); memcpy(&il, c + 8, 4); c += 12;
  if (ver != KXP_VERSION) return -1;
  c += pad4(il);
  p->nstages = ns;
  p->st = (stage_t*)calloc(ns, sizeof(stage_t));
  for (uint32_t s = 0; s < ns; ++s) {
    stage_t* t = &p->st[s];
    uint32_t h[16];
    memcpy(h, c, 64); c += 64;
    if (h[0] != KXP_STAGE_MAGIC) return -1;
    t->nstates = h[1]; t->nclasses = h[2]; t->q0 = h[3]; t->nregs = h[4]; t->nactions = h[5]; t->nops = h[6];
    t->nconsts = h[7]; t->constpool_len = h[8]; t->maxleaves = h[9]; t->nback = h[10]; t->npconsts = h[11];
    t->pconstpool_len = h[12]; t->nsync = h[13]; t->sync_complete = h[14]; t->actions = h[15];
    size_t sc = (size_t)t->nstates * t->nclasses;
    t->cls = c; c += 256;
    t->delta = (const uint16_t*)c; c += pad4(sc * 2);
    t->act = (const uint32_t*)c; c += sc * 4;
    t->final_act = (const uint32_t*)c; c += (size_t)t->nstates * 4;
    t->act_off = (const uint32_t*)c; c += ((size_t)t->nactions + 1) * 4;
    t->ops = (const uint32_t*)c; c += (size_t)t->nops * 8;
    t->const_off = (const uint32_t*)c; c += ((size_t)t->nconsts + 1) * 4;
    t->constpool = c; c += pad4(t->constpool_len);
    t->pback = (const uint32_t*)c; c += sc * 4;
    t->nleaves = c; c += pad4(t->nstates);
    t->fin_leaf = c; c += pad4(t->nstates);
    t->back = (const uint32_t*)c; c += (size_t)t->nback * t->maxleaves * 4;
    t->pconst_off = (const uint32_t*)c; c += ((size_t)t->npconsts + 1) * 4;
    t->pconstpool = c; c += pad4(t->pconstpool_len);
    t->init_const = (const uint32_t*)c; c += (size_t)t->maxleaves * 4;
    t->sync_next = (const uint32_t*)c; c += (size_t)t->nsync * t->nclasses * 4;
    t->sync_state = (const uint32_t*)c; c += (size_t)t->nsync * 4;
    if (t->actions & KXP_STAGE_HAS_TABLES) { memcpy(&t->ntables, c, 4); c += 4; t->tables = c; c += (size_t)t->ntables * 256; }
    if ((size_t)(c - b) > len) return -1;
  }
  return 0;
}

/* ---- registers: crt/crt.c:46-51,161-215 (byte units, word aligned) ---- */
typedef struct { uint8_t* data; size_t size, len; } buf_t;
#define INITIAL_BUFFER_SIZE (4096 * 8) /* crt/crt.c:19 */

static void buf_init(buf_t* b) { b->data = (uint8_t*)malloc(INITIAL_BUFFER_SIZE); b->size = INITIAL_BUFFER_SIZE; b->len = 0; }
static void buf_reserve(buf_t* b, size_t extra) { /* appendarray growth rule, crt/crt.c:229-243 */
  if (b->len + extra + 1 >= b->size) {
    size_t ns = b->size;
    while (b->len + extra + 1 >= ns) ns <<= 1;
    b->data = (uint8_t*)realloc(b->data, ns);
    b->size = ns;
  }
}
static void buf_append(buf_t* b, const uint8_t* p, size_t n) { buf_reserve(b, n); memcpy(b->data + b->len, p, n); b->len += n; }

static int run_stage_reg(const stage_t* t, const uint8_t* in, size_t n, buf_t* out, uint64_t* fail_pos) {
  buf_t* regs = (buf_t*)calloc(t->nregs ? t->nregs : 1, sizeof(buf_t));
  for (uint32_t r = 1; r < t->nregs; ++r) buf_init(&regs[r]); /* init(): all but the stream buffer */
  uint32_t q = t->q0;
  size_t count = 0;
  int rc = 0;
  for (;;) {
    uint32_t aid;
    int at_end = count >= n; /* !readnext(1,1) */
    if (at_end) {
      aid = t->final_act[q];
      if (aid == KXP_NOT_FINAL) { rc = 1; break; }
    } else {
      size_t ix = (size_t)q * t->nclasses + t->cls[in[count]];
      if (t->delta[ix] == KXP_NO_STATE) { rc = 1; break; }
      aid = t->act[ix];
    }
    for (uint32_t k = t->act_off[aid]; k < t->act_off[aid + 1]; ++k) {
      uint32_t op = t->ops[2 * k] >> 24, dst = t->ops[2 * k] & 0xFFFFFF, arg = t->ops[2 * k + 1];
      buf_t* d = dst == 0 ? out : &regs[dst];
      switch (op) {
        case KXP_OP_RESET: d->len = 0; break;
        case KXP_OP_APPEND_CONST: buf_append(d, t->constpool + t->const_off[arg], t->const_off[arg + 1] - t->const_off[arg]); break;
        case KXP_OP_APPEND_SYM: buf_append(d, &in[count], 1); break;
        case KXP_OP_APPEND_TBL: if (arg >= t->ntables) { rc = -3; break; } buf_append(d, &t->tables[(size_t)arg * 256 + in[count]], 1); break; /* C.hs:228-252 */
        case KXP_OP_CONCAT: buf_append(d, regs[arg].data, regs[arg].len); break; /* src keeps its value, crt.c:255-259 */
      }
    }
    if (at_end || rc) break;
    q = t->delta[(size_t)q * t->nclasses + t->cls[in[count]]];
    ++count; /* consume(1) */
  }
  if (rc) *fail_pos = count;
  for (uint32_t r = 1; r < t->nregs; ++r) free(regs[r].data);
  free(regs);
  return rc;
}

/* path form, sequentially: the algorithm of DESIGN.md §2 without any parallelism */
static int run_stage_path(const stage_t* t, const uint8_t* in, size_t n, buf_t* out, uint64_t* fail_pos) {
  uint16_t* qs = (uint16_t*)malloc((n + 1) * sizeof(uint16_t));
  uint8_t* leaf = (uint8_t*)malloc(n + 1);
  uint32_t q = t->q0;
  size_t i;
  int rc = 0;
  for (i = 0; i < n; ++i) {
    qs[i] = (uint16_t)q;
    uint16_t d = t->delta[(size_t)q * t->nclasses + t->cls[in[i]]];
    if (d == KXP_NO_STATE) { rc = 1; *fail_pos = i; goto done; }
    q = d;
  }
  qs[n] = (uint16_t)q;
  if (t->fin_leaf[q] == KXP_NO_LEAF) { rc = 1; *fail_pos = n; goto done; }
  leaf[n] = t->fin_leaf[q];
  for (i = n; i-- > 0;) {
    uint32_t b = t->pback[(size_t)qs[i] * t->nclasses + t->cls[in[i]]];
    uint32_t e = t->back[(size_t)b * t->maxleaves + leaf[i + 1]];
    if (e == KXP_DEAD_LEAF) { rc = -2; goto done; }
    leaf[i] = (uint8_t)(e & 0xFF);
  }
  {
    uint32_t c0 = t->init_const[leaf[0]];
    buf_append(out, t->pconstpool + t->pconst_off[c0], t->pconst_off[c0 + 1] - t->pconst_off[c0]);
  }
  for (i = 0; i < n; ++i) {
    uint32_t b = t->pback[(size_t)qs[i] * t->nclasses + t->cls[in[i]]];
    uint32_t e = t->back[(size_t)b * t->maxleaves + leaf[i + 1]];
    uint32_t tb = t->ntables ? e >> 24 : 0, c = t->ntables ? (e >> 9) & 0x7FFF : e >> 9;
    if (tb > t->ntables) { rc = -3; goto done; }
    if (e & 0x100) buf_append(out, tb ? &t->tables[(size_t)(tb - 1) * 256 + in[i]] : &in[i], 1);
    buf_append(out, t->pconstpool + t->pconst_off[c], t->pconst_off[c + 1] - t->pconst_off[c]);
  }
done:
  free(qs); free(leaf);
  return rc;
}

/* The action post-pass of a stage (kxp_format.h): replays the token stream on a stack of buffers and a register bank —
 * the semantics of src/KMC/Kleenex/Actions.hs:28-38 (inj / psh / pop / wr, starting from ([], [mempty])); the result is
 * the bottom buffer.  What the reference's action SST (ActionSST.hs:85-104) computes with stack-indexed registers. */
static void run_actions(uint32_t nregs, const uint8_t* in, size_t n, buf_t* out) {
  buf_t* regs = (buf_t*)calloc(nregs ? nregs : 1, sizeof(buf_t));
  for (uint32_t r = 0; r < nregs; ++r) buf_init(&regs[r]);
  size_t cap = 16, depth = 1;
  buf_t* stack = (buf_t*)calloc(cap, sizeof(buf_t));
  buf_init(&stack[0]);
  for (size_t i = 0; i < n; ++i) {
    const uint8_t b = in[i];
    if (b != KXP_ESC) { buf_append(&stack[depth - 1], &b, 1); continue; }       /* inj */
    const uint8_t k = i + 1 < n ? in[++i] : KXP_ESC;
    if (k == KXP_ESC) { buf_append(&stack[depth - 1], &k, 1); continue; }       /* inj 0xFF */
    if (k == KXP_TOK_PUSH) {                                                     /* psh */
      if (depth == cap) { cap *= 2; stack = (buf_t*)realloc(stack, cap * sizeof(buf_t)); }
      buf_init(&stack[depth++]);
      continue;
    }
    const uint8_t r = i + 1 < n ? in[++i] : 0;
    if (r >= nregs) continue;
    if (k == KXP_TOK_POP && depth > 1) {                                         /* pop r */
      free(regs[r].data);
      regs[r] = stack[--depth];
    } else if (k == KXP_TOK_WRITE) {                                             /* wr r */
      buf_append(&stack[depth - 1], regs[r].data, regs[r].len);
      regs[r].len = 0;
    }
  }
  buf_append(out, stack[0].data, stack[0].len);
  for (size_t d = 0; d < depth; ++d) free(stack[d].data);
  for (uint32_t r = 0; r < nregs; ++r) free(regs[r].data);
  free(stack); free(regs);
}

static int run_all(const uint8_t* blob, size_t blob_len, const uint8_t* in, size_t n, uint8_t** outp, size_t* out_len,
                   uint64_t* fail_pos, uint32_t* fail_stage, int path) {
  prog_t p;
  if (parse_blob(blob, blob_len, &p)) return -1;
  const uint8_t* cur = in; size_t curn = n;
  uint8_t* owned = NULL;
  int rc = 0;
  for (uint32_t s = 0; s < p.nstages; ++s) {
    buf_t out; buf_init(&out);
    uint64_t fp = 0;
    rc = path ? run_stage_path(&p.st[s], cur, curn, &out, &fp) : run_stage_reg(&p.st[s], cur, curn, &out, &fp);
    free(owned);
    owned = out.data; cur = owned; curn = out.len;
    if (rc) { *fail_pos = fp; *fail_stage = s; break; }
    if (p.st[s].actions & 1u) {
      buf_t fin; buf_init(&fin);
      run_actions(p.st[s].actions >> 8, cur, curn, &fin);
      free(owned);
      owned = fin.data; cur = owned; curn = fin.len;
    }
  }
  free(p.st);
  *outp = owned; *out_len = curn;
  return rc;
}

/* 0 = accepted; 1 = match error (fail_pos = symbols consumed before the failing state;
 * *out then holds what had been appended to the stream so far — the reference would have
 * flushed only the first out_len - out_len % 16384 bytes of it, crt/crt.c:17,217-227);
 * <0 = malformed blob / internal error.  *out is malloc'd; release with kxo_free. */
int kxo_run(const uint8_t* blob, size_t blob_len, const uint8_t* in, size_t n, uint8_t** out, size_t* out_len,
            uint64_t* fail_pos, uint32_t* fail_stage) {
  return run_all(blob, blob_len, in, n, out, out_len, fail_pos, fail_stage, 0);
}
int kxo_run_path(const uint8_t* blob, size_t blob_len, const uint8_t* in, size_t n, uint8_t** out, size_t* out_len,
                 uint64_t* fail_pos, uint32_t* fail_stage) {
  return run_all(blob, blob_len, in, n, out, out_len, fail_pos, fail_stage, 1);
}
void kxo_free(void* p) { free(p); }

/* summary numbers used by tests / DESIGN.md tables */
int kxo_info(const uint8_t* blob, size_t blob_len, uint32_t stage, uint32_t* v /* [8] */) {
  prog_t p;
  if (parse_blob(blob, blob_len, &p) || stage >= p.nstages) return -1;
  const stage_t* t = &p.st[stage];
  v[0] = t->nstates; v[1] = t->nclasses; v[2] = t->nregs; v[3] = t->nactions;
  v[4] = t->maxleaves; v[5] = t->nback; v[6] = t->nsync; v[7] = t->sync_complete;
  free(p.st);
  return 0;
}

#ifdef KXO_MAIN
/* kx_oracle prog.kxp [--path] < in > out */
int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s prog.kxp [--path] < in > out\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  fseek(f, 0, SEEK_END); long bl = ftell(f); fseek(f, 0, SEEK_SET);
  uint8_t* blob = (uint8_t*)malloc(bl);
  if (fread(blob, 1, bl, f) != (size_t)bl) return 2;
  fclose(f);
  size_t cap = 1 << 20, n = 0;
  uint8_t* in = (uint8_t*)malloc(cap);
  for (;;) {
    size_t r = fread(in + n, 1, cap - n, stdin);
    n += r;
    if (r == 0) break;
    if (n == cap) { cap <<= 1; in = (uint8_t*)realloc(in, cap); }
  }
  uint8_t* out; size_t ol; uint64_t fp = 0; uint32_t fs = 0;
  int path = argc > 2 && !strcmp(argv[2], "--path");
  int rc = run_all(blob, bl, in, n, &out, &ol, &fp, &fs, path);
  if (rc == 0) { fwrite(out, 1, ol, stdout); return 0; }
  if (rc == 1) {
    fwrite(out, 1, ol - ol % 16384, stdout);
    fprintf(stderr, "Match error at input symbol %zu!\n", (size_t)fp);
    return 1;
  }
  fprintf(stderr, "oracle: bad program blob\n");
  return 2;
}
#endif
