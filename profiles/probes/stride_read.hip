// Probe: what does a lane-per-segment read pattern cost on MI355X?  262144 lanes (256 CUs x 1024), lane k streams through
// its own segment of SEG bytes, CH bytes per visit (CH/16 global_load_dwordx4 back to back), versus the same bytes read
// with adjacent lanes on adjacent 16-byte words.  Build: hipcc --offload-arch=gfx950 -O3 -o stride_read stride_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int CH>
__global__ void k_lane_seg(const uint8_t* __restrict__ in, uint64_t seg, uint32_t nseg, uint32_t* sink) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nseg) return;
  const uint4* p = reinterpret_cast<const uint4*>(in + (uint64_t)k * seg);
  uint32_t acc = 0;
  for (uint64_t off = 0; off < seg; off += CH) {
    uint4 v[CH / 16];
#pragma unroll
    for (int i = 0; i < CH / 16; ++i) v[i] = p[off / 16 + i];
#pragma unroll
    for (int i = 0; i < CH / 16; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
// the wave reads 8 segments x 128 bytes per instruction (lane l: segment base + (l >> 3), word l & 7), 8 instructions per visit
__global__ void k_coop8(const uint8_t* __restrict__ in, uint64_t seg, uint32_t nseg, uint32_t* sink) {
  const uint32_t k0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63u, lane = threadIdx.x & 63;
  if (k0 >= nseg) return;
  uint32_t acc = 0;
  for (uint64_t off = 0; off < seg; off += 128) {
    uint4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const uint4*>(in + (uint64_t)(k0 + 8 * i + (lane >> 3)) * seg + off + (lane & 7) * 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void k_stream(const uint8_t* __restrict__ in, uint64_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += (uint64_t)gridDim.x * blockDim.x * 16) {
    const uint4 v = *reinterpret_cast<const uint4*>(in + i);
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
int main(int argc, char** argv) {
  const uint64_t n = (argc > 1 ? atof(argv[1]) : 8.0) * (1ull << 30);
  uint8_t* d; uint32_t* sink;
  hipMalloc(&d, n); hipMalloc(&sink, 64); hipMemset(d, 1, n);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](const char* name, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-40s %8.3f ms  %8.1f GB/s\n", name, ms, n / ms / 1e6);
  };
  for (uint32_t lanes_per_cu : {1024u, 2048u}) {
    const uint32_t nseg = 256 * lanes_per_cu; const uint64_t seg = (n / nseg) & ~(uint64_t)511;
    printf("-- %u lanes per CU, segment %llu bytes, 512-lane groups\n", lanes_per_cu, (unsigned long long)seg);
    run("lane=segment, 128 B per visit", [&] { hipLaunchKernelGGL(k_lane_seg<128>, dim3(nseg / 512), dim3(512), 0, 0, d, seg, nseg, sink); });
    run("lane=segment, 256 B per visit", [&] { hipLaunchKernelGGL(k_lane_seg<256>, dim3(nseg / 512), dim3(512), 0, 0, d, seg, nseg, sink); });
    run("lane=segment, 512 B per visit", [&] { hipLaunchKernelGGL(k_lane_seg<512>, dim3(nseg / 512), dim3(512), 0, 0, d, seg, nseg, sink); });
    run("lane=segment, 64 B per visit", [&] { hipLaunchKernelGGL(k_lane_seg<64>, dim3(nseg / 512), dim3(512), 0, 0, d, seg, nseg, sink); });
    run("wave reads 8 segments x 128 B per load", [&] { hipLaunchKernelGGL(k_coop8, dim3(nseg / 512), dim3(512), 0, 0, d, seg, nseg, sink); });
  }
  run("adjacent lanes, adjacent words", [&] { hipLaunchKernelGGL(k_stream, dim3(256 * 8), dim3(512), 0, 0, d, n, sink); });
  return 0;
}
