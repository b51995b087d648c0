// LDS cost table probe: cycles per wave-instruction per CU for the access patterns the engine uses or may use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REPS 2000
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// kinds
enum { RD32_RAND, RD32_HOT, RD8_ASCII, WR8_STRIDE80, WR8_STRIDE80_M3, WR8_STRIDE80_M10, WR32_UNAL80, WR32_AL80, RD32_UNAL_RAND,
       RD64_RAND, WR32_M11_RAND, WR128_CONTIG, RD16_RAND, WR8_JIT, WR32_UNAL_JIT, RD32_UNAL_STRIDE64, RD128_CONTIG, WR64_UNAL_JIT,
       WR8_COL, WR8_P132, WR8_P132_JIT, WR32_COL, RD32_COL, WR8_P132_M3, WR8_RANDOM, NKINDS };
static const char* kname[] = {"ds_read_b32 random 13.5KB", "ds_read_b32 50 hot entries", "ds_read_u8 ascii cls", "ds_write_b8 lane*80+k",
  "ds_write_b8 lane*80+k 1/3 lanes", "ds_write_b8 lane*80+k 1/10 lanes", "ds_write_b32 unaligned lane*80+4k+1", "ds_write_b32 aligned lane*80+4k",
  "ds_read_b32 unaligned random", "ds_read_b64 random", "ds_write_b32 1/11 lanes random", "ds_write_b128 contiguous", "ds_read_u16 random",
  "ds_write_b8 jitter(70..90)*lane", "ds_write_b32 unaligned jitter", "ds_read_b32 unaligned lane*64+k (input gather)", "ds_read_b128 contiguous", "ds_write_b64 unaligned jitter",
  "ds_write_b8 column (bank = lane)", "ds_write_b8 lane*132+k", "ds_write_b8 lane*132+drift(0..23)+k", "ds_write_b32 column (bank = lane)", "ds_read_b32 column (bank = lane)",
  "ds_write_b8 lane*132+drift+k 1/3 lanes", "ds_write_b8 random 6 KB"};

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int KIND>
__global__ void k(uint32_t* out, uint64_t* ticks) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t i = threadIdx.x; i < 8192; i += blockDim.x) smem[i] = i * 2654435761u;
  __syncthreads();
  uint32_t seed = (blockIdx.x * 1024 + threadIdx.x) * 7919u + 17u;
  uint32_t a[8];
  const uint32_t wbase = 32768 + wave * 6144;   // per-wave staging region (bytes), after a 16 KB table
  uint32_t jit = 0; { uint32_t s2 = lane * 77u + 5u; for (uint32_t i = 0; i < lane; ++i) { s2 = s2 * 1664525u + 1013904223u; jit += 70 + (s2 >> 24) % 21; } }
  bool active = true;
  for (int j = 0; j < 8; ++j) {
    uint32_t r = lcg(seed);
    switch (KIND) {
      case RD32_RAND: a[j] = (r % 3456) * 4; break;
      case RD32_HOT: a[j] = ((r % 50) * 67 % 3456) * 4; break;
      case RD8_ASCII: a[j] = 32 + r % 96; break;
      case WR8_STRIDE80: a[j] = wbase + lane * 80 + j; break;
      case WR8_STRIDE80_M3: a[j] = wbase + lane * 80 + j; active = lane % 3 == 0; break;
      case WR8_STRIDE80_M10: a[j] = wbase + lane * 80 + j; active = lane % 10 == 0; break;
      case WR32_UNAL80: a[j] = wbase + lane * 80 + 4 * j + 1; break;
      case WR32_AL80: a[j] = wbase + lane * 80 + 4 * j; break;
      case RD32_UNAL_RAND: a[j] = r % 13800; break;
      case RD64_RAND: a[j] = (r % 1728) * 8; break;
      case WR32_M11_RAND: a[j] = wbase + (r % 1024) * 4; active = (lane * 7 + 3) % 11 == 0; break;
      case WR128_CONTIG: a[j] = wbase + lane * 16 + (j & 3) * 1024; break;
      case RD16_RAND: a[j] = (r % 6900) * 2; break;
      case WR8_JIT: a[j] = wbase + jit + j; break;
      case WR32_UNAL_JIT: a[j] = wbase + jit + 4 * j + (lane & 3); break;
      case RD32_UNAL_STRIDE64: a[j] = wbase + lane * 64 + 4 * j + (lane * 5 & 3); break;
      case RD128_CONTIG: a[j] = wbase + lane * 16 + (j & 3) * 1024; break;
      case WR64_UNAL_JIT: a[j] = wbase + jit + 8 * j + (lane & 3); break;
      case WR8_COL: a[j] = wbase + lane * 4 + (j >> 2) * 256 + (j & 3); break;
      case WR8_P132: a[j] = wbase + lane * 132 + j; break;
      case WR8_P132_JIT: a[j] = wbase + lane * 132 + (lane * 2654435761u >> 8) % 24 + j; break;
      case WR8_P132_M3: a[j] = wbase + lane * 132 + (lane * 2654435761u >> 8) % 24 + j; active = lane % 3 == 0; break;
      case WR32_COL: a[j] = wbase + lane * 4 + j * 256; break;
      case RD32_COL: a[j] = wbase + lane * 4 + j * 256; break;
      case WR8_RANDOM: a[j] = wbase + r % 6144; break;
    }
  }
  uint32_t acc = 0, v = lane * 0x01010101u;
  uint64_t t0 = __builtin_readcyclecounter();
  if (active) {
    for (int it = 0; it < REPS; ++it) {
      uint32_t r0, r1, r2, r3, r4, r5, r6, r7;
      if constexpr (KIND == RD32_RAND || KIND == RD32_HOT || KIND == RD32_UNAL_RAND || KIND == RD32_UNAL_STRIDE64 || KIND == RD32_COL) {
        asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %9\n ds_read_b32 %2, %10\n ds_read_b32 %3, %11\n ds_read_b32 %4, %12\n ds_read_b32 %5, %13\n ds_read_b32 %6, %14\n ds_read_b32 %7, %15\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
        acc += r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
      } else if constexpr (KIND == RD8_ASCII) {
        asm volatile("ds_read_u8 %0, %8\n ds_read_u8 %1, %9\n ds_read_u8 %2, %10\n ds_read_u8 %3, %11\n ds_read_u8 %4, %12\n ds_read_u8 %5, %13\n ds_read_u8 %6, %14\n ds_read_u8 %7, %15\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
        acc += r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
      } else if constexpr (KIND == RD16_RAND) {
        asm volatile("ds_read_u16 %0, %8\n ds_read_u16 %1, %9\n ds_read_u16 %2, %10\n ds_read_u16 %3, %11\n ds_read_u16 %4, %12\n ds_read_u16 %5, %13\n ds_read_u16 %6, %14\n ds_read_u16 %7, %15\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
        acc += r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
      } else if constexpr (KIND == RD64_RAND) {
        uint64_t q0, q1, q2, q3, q4, q5, q6, q7;
        asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %9\n ds_read_b64 %2, %10\n ds_read_b64 %3, %11\n ds_read_b64 %4, %12\n ds_read_b64 %5, %13\n ds_read_b64 %6, %14\n ds_read_b64 %7, %15\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7)
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
        acc += (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7);
      } else if constexpr (KIND == RD128_CONTIG) {
        u32x4 q0, q1, q2, q3;
        asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %5\n ds_read_b128 %2, %6\n ds_read_b128 %3, %7\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
        acc += q0.x ^ q1.y ^ q2.z ^ q3.w;
      } else if constexpr (KIND == WR8_STRIDE80 || KIND == WR8_STRIDE80_M3 || KIND == WR8_STRIDE80_M10 || KIND == WR8_JIT || KIND == WR8_COL || KIND == WR8_P132 || KIND == WR8_P132_JIT || KIND == WR8_P132_M3 || KIND == WR8_RANDOM) {
        asm volatile("ds_write_b8 %0, %8\n ds_write_b8 %1, %8\n ds_write_b8 %2, %8\n ds_write_b8 %3, %8\n ds_write_b8 %4, %8\n ds_write_b8 %5, %8\n ds_write_b8 %6, %8\n ds_write_b8 %7, %8\n s_waitcnt lgkmcnt(0)"
                     :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(v) : "memory");
      } else if constexpr (KIND == WR128_CONTIG) {
        u32x4 q = {v, v, v, v};
        asm volatile("ds_write_b128 %0, %4\n ds_write_b128 %1, %4\n ds_write_b128 %2, %4\n ds_write_b128 %3, %4\n s_waitcnt lgkmcnt(0)"
                     :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(q) : "memory");
      } else if constexpr (KIND == WR64_UNAL_JIT) {
        uint64_t q = v;
        asm volatile("ds_write_b64 %0, %8\n ds_write_b64 %1, %8\n ds_write_b64 %2, %8\n ds_write_b64 %3, %8\n ds_write_b64 %4, %8\n ds_write_b64 %5, %8\n ds_write_b64 %6, %8\n ds_write_b64 %7, %8\n s_waitcnt lgkmcnt(0)"
                     :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(q) : "memory");
      } else {
        asm volatile("ds_write_b32 %0, %8\n ds_write_b32 %1, %8\n ds_write_b32 %2, %8\n ds_write_b32 %3, %8\n ds_write_b32 %4, %8\n ds_write_b32 %5, %8\n ds_write_b32 %6, %8\n ds_write_b32 %7, %8\n s_waitcnt lgkmcnt(0)"
                     :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(v) : "memory");
      }
      v += acc;
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0 && wave == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + v;
}

template <int KIND>
void run(int waves, uint32_t* d_out, uint64_t* d_ticks) {
  const int nops = (KIND == WR128_CONTIG || KIND == RD128_CONTIG) ? 4 : 8;
  size_t lds = 32768 + 16 * 6144 + 4096;
  hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(waves * 64), lds, 0, d_out, d_ticks);
  { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess || hipGetLastError() != hipSuccess) printf("launch error %s\n", hipGetErrorString(e)); }
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(waves * 64), lds, 0, d_out, d_ticks);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  uint64_t ticks; hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost);
  double ninstr = (double)waves * REPS * nops;   // per CU
  printf("%-48s waves=%2d  %7.3f ms  %6.2f ns/instr/CU = %5.2f cyc@2.4GHz  (wave0 ticks/instr-round %.1f)\n", kname[KIND], waves, ms,
         ms * 1e6 / ninstr, ms * 1e6 / ninstr * 2.4, (double)ticks / REPS);
}

template <int K> void all(uint32_t* o, uint64_t* t) { run<K>(4, o, t); run<K>(12, o, t); run<K>(16, o, t); if constexpr (K + 1 < NKINDS) all<K + 1>(o, t); }

int main() {
  uint32_t* d; uint64_t* t;
  hipMalloc(&d, 256 * 1024 * 4); hipMalloc(&t, 64);
  all<0>(d, t);
  return 0;
}
