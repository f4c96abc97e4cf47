// What does one scattered VMEM instruction cost a wave?  256 workgroups x 4 waves (one per SIMD) issue NI loads / stores of a given lane->address
// pattern back to back; s_memrealtime before, after the last issue, and after vmcnt(0).  Patterns (bytes per lane, lanes per contiguous segment, segment stride):
//   0: 8 B, 64 lanes contiguous (512 B)            1: 8 B, 4-lane segments of 32 B, stride 8320 B (16 segments)
//   2: 8 B, lane (n + 16 q) -> row n, 8 q (64 single-lane pieces: 4 per 32-B segment but lanes far apart)
//   3: 16 B, 8-lane segments of 128 B, stride 8320 B (8 segments)      4: 16 B, 64 lanes contiguous (1 KB)
//   5: 16 B, 4-lane segments of 64 B, stride 8320 (16 segments)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned u32x2;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
constexpr int NI = 12;
template <int PAT, bool STORE>
__global__ __launch_bounds__(256, 1) void k(char* buf, size_t per_wg, unsigned long long* out, int reps) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* base = buf + (size_t)blockIdx.x * per_wg + (size_t)wave * (per_wg / 4);
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(per_wg / 4), 0x00020000);
  unsigned voff;
  const unsigned RS = 8320;
  if (PAT == 0) voff = lane * 8;
  else if (PAT == 1) voff = (lane >> 2) * RS + (lane & 3) * 8;
  else if (PAT == 2) voff = (lane & 15) * RS + (lane >> 4) * 8;
  else if (PAT == 3) voff = (lane >> 3) * RS + (lane & 7) * 16;
  else if (PAT == 4) voff = lane * 16;
  else voff = (lane >> 2) * RS + (lane & 3) * 16;
  constexpr bool W16 = PAT >= 3;
  unsigned long long t_issue = 0, t_done = 0;
  unsigned acc = 0;
  for (int rep = 0; rep < reps; ++rep) {
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    u32x2 v2[NI]; u32x4 v4[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const unsigned so = (unsigned)(i * 1024 + (rep & 7) * 16 * RS);       // another gate / another block of rows each time
      if (STORE) {
        if (W16) __builtin_amdgcn_raw_buffer_store_b128(u32x4{acc, 1u, 2u, 3u}, r, voff, so, 0);
        else __builtin_amdgcn_raw_buffer_store_b64(u32x2{acc, 1u}, r, voff, so, 0);
      } else {
        if (W16) v4[i] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, so, 0);
        else v2[i] = __builtin_amdgcn_raw_buffer_load_b64(r, voff, so, 0);
      }
    }
    asm volatile("" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (!STORE) {
#pragma unroll
      for (int i = 0; i < NI; ++i) acc += W16 ? v4[i][0] : v2[i][0];
    }
    if (rep >= 2) { t_issue += t1 - t0; t_done += t2 - t0; }
  }
  if (lane == 0) {
    out[(blockIdx.x * 4 + wave) * 2] = t_issue;
    out[(blockIdx.x * 4 + wave) * 2 + 1] = t_done + (acc == 0x12345u);
  }
}
template <int PAT, bool STORE> void run(char* buf, size_t per_wg, unsigned long long* dout, int wgs) {
  const int reps = 34;
  hipLaunchKernelGGL((k<PAT, STORE>), dim3(wgs), dim3(256), 0, 0, buf, per_wg, dout, reps);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(wgs * 8);
  hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  double si = 0, sd = 0;
  for (int i = 0; i < wgs * 4; ++i) { si += h[2 * i]; sd += h[2 * i + 1]; }
  si /= (wgs * 4.0 * (reps - 2)); sd /= (wgs * 4.0 * (reps - 2));
  printf("pattern %d %s  wgs %3d: issue of %d instructions %7.0f cycles (%5.0f each), all complete after %7.0f cycles\n", PAT, STORE ? "store" : "load ", wgs, NI, si, si / NI, sd);
}
int main() {
  const size_t per_wg = 4 * (size_t)(8 * 16 * 8320 + NI * 1024 + 65536);
  char* buf; unsigned long long* dout;
  hipMalloc(&buf, per_wg * 256); hipMemset(buf, 0, per_wg * 256); hipMalloc(&dout, 256 * 8 * 8);
  for (int wgs : {1, 256}) {
    run<0, false>(buf, per_wg, dout, wgs); run<1, false>(buf, per_wg, dout, wgs); run<2, false>(buf, per_wg, dout, wgs); run<3, false>(buf, per_wg, dout, wgs);
    run<4, false>(buf, per_wg, dout, wgs); run<5, false>(buf, per_wg, dout, wgs);
    run<0, true>(buf, per_wg, dout, wgs); run<1, true>(buf, per_wg, dout, wgs); run<2, true>(buf, per_wg, dout, wgs); run<3, true>(buf, per_wg, dout, wgs);
    run<4, true>(buf, per_wg, dout, wgs); run<5, true>(buf, per_wg, dout, wgs);
  }
  return 0;
}
