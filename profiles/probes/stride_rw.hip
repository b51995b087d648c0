// Probe: k_backlen's memory pattern with the kind bytes added.  262144 lanes (256 CUs x 1024), lane k walks its own segment
// backward, 128 B of input per visit (two pieces), and leaves 64 B of output per piece.  Variants of the read (per-lane
// line / 8 lanes per line) and of the write (per-lane 4 x 16 B / 4 lanes per 64-byte line / 8 lanes per 128-byte line /
// wave-contiguous 4 KiB) are timed alone and together.  Build: hipcc --offload-arch=gfx950 -O3 -o stride_rw stride_rw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// RD: 0 none, 1 per-lane 128 B, 2 cooperative (8 lanes per line).  WR: 0 none, 1 per-lane 4 x 16 B per piece,
// 2 four lanes per 64-byte line, 3 eight lanes per 128-byte line (two pieces at once), 4 wave-contiguous (lane-interleaved layout)
template <int RD, int WR>
__global__ void k_rw(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t seg, uint32_t nseg, uint32_t* sink) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, k0 = k & ~63u;
  if (k0 >= nseg) return;
  uint32_t acc = 0;
  for (uint64_t off = seg; off >= 128; off -= 128) {
    const uint64_t o = off - 128;
    uint4 v[8];
    if (RD == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const uint4*>(in + (uint64_t)k * seg + o + i * 16);
    } else if (RD == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const uint4*>(in + (uint64_t)(k0 + 8 * i + (lane >> 3)) * seg + o + (lane & 7) * 16);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = make_uint4(k + i, lane, (uint32_t)o, acc);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    const uint4 w = make_uint4(acc, acc + 1, acc + 2, acc + 3);
    if (WR == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(out + (uint64_t)k * seg + o + i * 16) = w;
    } else if (WR == 2) {   // 2 pieces x 4 instructions: lane l writes quarter l&3 of the line of lane 16 i + (l >> 2)
#pragma unroll
      for (int pc = 0; pc < 2; ++pc)
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(out + (uint64_t)(k0 + 16 * i + (lane >> 2)) * seg + o + pc * 64 + (lane & 3) * 16) = w;
    } else if (WR == 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(out + (uint64_t)(k0 + 8 * i + (lane >> 3)) * seg + o + (lane & 7) * 16) = w;
    } else if (WR == 4) {   // layout [wave][visit][lane][128 B]: the wave's 64 x 128 B of one visit are contiguous
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(out + (uint64_t)k0 * seg + o * 64 + (uint64_t)i * 1024 + lane * 16) = w;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
int main(int argc, char** argv) {
  const uint64_t n = (argc > 1 ? atof(argv[1]) : 8.0) * (1ull << 30);
  uint8_t *d, *o; uint32_t* sink;
  hipMalloc(&d, n); hipMalloc(&o, n); hipMalloc(&sink, 64); hipMemset(d, 1, n);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const uint32_t nseg = 256 * 1024; const uint64_t seg = (n / nseg) & ~(uint64_t)1023;
  auto run = [&](const char* name, auto kern, double bytes) {
    hipLaunchKernelGGL(kern, dim3(nseg / 512), dim3(512), 0, 0, d, o, seg, nseg, sink); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(nseg / 512), dim3(512), 0, 0, d, o, seg, nseg, sink); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-64s %8.3f ms  %8.1f GB/s (reads + writes)\n", name, ms, bytes / ms / 1e6);
  };
  printf("-- 1024 lanes per CU, segment %llu bytes\n", (unsigned long long)seg);
  const double N = (double)seg * nseg;
  run("read per-lane line", k_rw<1, 0>, N);
  run("read 8 lanes per line", k_rw<2, 0>, N);
  run("write per-lane 16 B x 8", k_rw<0, 1>, N);
  run("write 4 lanes per 64-byte line", k_rw<0, 2>, N);
  run("write 8 lanes per 128-byte line", k_rw<0, 3>, N);
  run("write wave-contiguous", k_rw<0, 4>, N);
  run("read per-lane + write per-lane", k_rw<1, 1>, 2 * N);
  run("read per-lane + write 4 lanes per 64-byte line", k_rw<1, 2>, 2 * N);
  run("read per-lane + write 8 lanes per 128-byte line", k_rw<1, 3>, 2 * N);
  run("read per-lane + write wave-contiguous", k_rw<1, 4>, 2 * N);
  run("read coop + write 4 lanes per 64-byte line", k_rw<2, 2>, 2 * N);
  run("read coop + write 8 lanes per 128-byte line", k_rw<2, 3>, 2 * N);
  run("read coop + write wave-contiguous", k_rw<2, 4>, 2 * N);
  return 0;
}
