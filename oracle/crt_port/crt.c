/* crt.c (port) — ORACLE / CPU-baseline runtime, restated from the behaviour of the reference's
 * crt/crt.c so that `kexc --backend=c` output can be built where the reference tree is absent.
 * Test infrastructure only.  Interface kept: the generated code defines printCompilationInfo(),
 * init(), match(int); this file provides next/avail/count, readnext, consume, cmp, reset,
 * init_buffer, append, appendarray, concat, outputconst, outputarray, output and main()
 * (reference: crt/crt.c:101-105 program interface, :285-324 input window, :161-283 buffers,
 * :326-467 main).  Byte-unit, word-aligned configuration only (BUFFER_UNIT_T = uint8_t with
 * FLAG_WORDALIGNED — the only one the reference's front end ever selects, src/kexc.hs:42-48).
 * Behavioural constants kept: 2×16 KiB sliding input window, 16 KiB output flush granularity,
 * 32 KiB initial registers growing by doubling, exit codes 0/1/2, message texts. */
#include <getopt.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <sys/wait.h>
#include <unistd.h>

#ifndef NUM_PHASES
#error "NUM_PHASES not defined."
#endif
#ifndef OUTSTREAM
#define OUTSTREAM stdout
#endif
typedef uint8_t buffer_unit_t;
typedef struct { buffer_unit_t* data; size_t size; size_t bitpos; } buffer_t;

#define WIN (16 * 1024)
static unsigned char window[2 * WIN];
static size_t win_fill = 0;   /* valid bytes in the upper half */
static long win_cur = 0;      /* cursor relative to the middle of the window */
unsigned char* next;
size_t count = 0;
#define avail ((long)win_fill - win_cur)

static buffer_t outbuf;

void printCompilationInfo();
void init();
void match(int phase);

static inline int readnext(int minc, int maxc) {
  if (minc == 0) return 1;
  if (avail < maxc) {               /* slide what is left below the middle, refill above it */
    long rest = avail;
    memmove(&window[WIN - rest], &window[WIN + win_cur], (size_t)rest);
    win_cur = -rest;
    win_fill = fread(&window[WIN], 1, WIN, stdin);
  }
  if (avail < minc) return 0;
  next = &window[WIN + win_cur];
  return 1;
}
static inline void consume(int c) { count += (size_t)c; win_cur += c; next += c; }
static inline int cmp(unsigned char* a, unsigned char* b, int l) { return memcmp(a, b, (size_t)l) == 0; }

static inline void reset(buffer_t* b) { b->bitpos = 0; }
void init_buffer(buffer_t* b) { b->size = 4096 * 8; b->data = malloc(b->size); b->bitpos = 0; }
static void grow(buffer_t* b, size_t need_bytes) {
  size_t ns = b->size;
  while (need_bytes + 1 >= ns) ns <<= 1;
  if (ns != b->size) { b->data = realloc(b->data, ns); b->size = ns; }
}
static inline void appendarray(buffer_t* d, const buffer_unit_t* a, size_t bits) {
  size_t n = bits / 8, len = d->bitpos / 8;
  grow(d, len + n);
  memcpy(d->data + len, a, n);
  d->bitpos += bits;
}
static inline void append(buffer_t* d, buffer_unit_t w, size_t bits) { (void)bits; appendarray(d, &w, 8); }
static inline void concat(buffer_t* d, buffer_t* s) { appendarray(d, s->data, s->bitpos); }

static void flush_some(void) {
  size_t n = outbuf.bitpos / 8;
  if (n && fwrite(outbuf.data, 1, n, OUTSTREAM) != n) { fprintf(stderr, "Error writing to output stream.\n"); exit(1); }
  outbuf.bitpos = 0;
}
static inline void outputconst(buffer_unit_t w, size_t bits) {
  (void)bits;
  outbuf.data[outbuf.bitpos / 8] = w;
  outbuf.bitpos += 8;
  if (outbuf.bitpos / 8 >= WIN) flush_some();   /* 16 KiB granularity */
}
static inline void outputarray(const buffer_unit_t* a, size_t bits) { for (size_t i = 0; i < bits / 8; ++i) outputconst(a[i], 8); }
static inline void output(buffer_t* b) { outputarray(b->data, b->bitpos); }

static void run(int phase) {
  outbuf.size = WIN + 1; outbuf.data = malloc(outbuf.size); outbuf.bitpos = 0;
  init();
  match(phase);
  flush_some();
  fflush(OUTSTREAM);
}

#ifndef FLAG_NOMAIN
static void usage(char* name) {
  fprintf(stdout, "Normal usage: %s < infile > outfile\n", name);
  fprintf(stdout, "- \"%s\": reads from stdin and writes to stdout.\n", name);
  fprintf(stdout, "- \"%s -i\": prints compilation info.\n", name);
  fprintf(stdout, "- \"%s -t\": runs normally, but prints timing to stderr.\n", name);
}
int main(int argc, char* argv[]) {
  static struct option lo[] = {{"phase", required_argument, 0, 'p'}, {0, 0, 0, 0}};
  int timing = 0, phase = 0, c;
  while ((c = getopt_long(argc, argv, "ihtp:", lo, NULL)) != -1) {
    switch (c) {
      case 'i': printCompilationInfo(); return 2;
      case 't': timing = 1; break;
      case 'p': phase = atoi(optarg); break;
      default: usage(argv[0]); return 1;
    }
  }
  struct timeval t0, t1;
  if (timing) gettimeofday(&t0, NULL);
  if (phase) run(phase);
  else {
    /* stdin → phase 1 → pipe → … → phase n → stdout, one process per phase */
    for (int i = 1; i < NUM_PHASES; ++i) {
      int fd[2];
      if (pipe(fd)) { fprintf(stderr, "Error creating pipe %d.", i); return 1; }
      pid_t pid = fork();
      if (pid == 0) { close(fd[0]); dup2(fd[1], STDOUT_FILENO); close(fd[1]); run(i); exit(0); }
      close(fd[1]); dup2(fd[0], STDIN_FILENO); close(fd[0]);
    }
    run(NUM_PHASES);
  }
  if (timing) {
    gettimeofday(&t1, NULL);
    fprintf(stderr, "time (ms): %ld\n", (long)((t1.tv_sec - t0.tv_sec) * 1000 + (t1.tv_usec - t0.tv_usec) / 1000));
  }
  return 0;
}
#endif
