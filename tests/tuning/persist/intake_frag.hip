// Intake probe for the weights-resident recurrence (b = 128): every CU of a "layer" (64 CUs) reads the SAME [128 rows x 1024 k] bf16 panel per step
// (a fresh panel every step, as hs[l][t] is), as MFMA B fragments straight into registers: lane (n = lane & 15, kg = lane >> 4) loads the 16 bytes
// act[row 16 ni + n][k0 + 8 kg ..]: 64-byte row segments per instruction.  Wave w of 4 owns k in [256 w, 256 w + 256).
//   mode 0: plain global_load_dwordx4       mode 1: global_load_dwordx4 sc1       mode 2: sc0 sc1
//   mode 3: LDS-DMA (buffer_load ... lds, full 128-byte row segments, 8 rows per instruction), no fragment reads
// Prints us per step and GB/s per CU.   hipcc --offload-arch=gfx950 -O3 intake_frag.hip -o intake_frag && ./intake_frag
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void lds_void_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ROWS = 128, LD = 1088, K = 1024;        // elements

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256, 1) void k(const uint16_t* __restrict__ buf, int steps, int panels, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int layer = (blockIdx.x & 7) >> 1;            // blocks b, b+8 share an XCD (observed): a layer = one XCD pair
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int s = 0; s < steps; ++s) {
    const uint16_t* p = buf + (size_t)((s * 4 + layer) % panels) * ROWS * LD;
    if (MODE < 3) {
      const char* base = reinterpret_cast<const char*>(p) + (size_t)(lane & 15) * LD * 2 + wave * 512 + (lane >> 4) * 16;
      // 64 loads per lane: ni 0..7 (row tile), kb 0..7 (64-byte column block)
      uint4 v[DEPTH];
#pragma unroll
      for (int i = 0; i < 64 + DEPTH; ++i) {
        if (i >= DEPTH) { const uint4 x = v[i % DEPTH]; acc.x ^= x.x; acc.y ^= x.y; acc.z ^= x.z; acc.w ^= x.w; }
        if (i < 64) {
          const int ni = i >> 3, kb = i & 7;
          const char* a = base + (size_t)ni * 16 * LD * 2 + kb * 64;
          uint4 x;
          if (MODE == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x) : "v"(a));
          if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(x) : "v"(a));
          if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(x) : "v"(a));
          v[i % DEPTH] = x;
        }
        if (i + 1 >= DEPTH && i + 1 < 64 + DEPTH) {
          // wait until the load that the NEXT iteration consumes has landed: at most min(DEPTH, remaining) - 1 younger ones may be in flight
          const int inflight = (i + 1 < 64) ? DEPTH - 1 : (64 + DEPTH - 1 - (i + 1));
          if (inflight >= DEPTH - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 1) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
    } else {
      // LDS-DMA: this wave's 128 rows x 512 bytes, 8 rows x 128 B... one instruction = 64 lanes x 16 B = 2 rows x 512 B
      __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p), 0, ROWS * LD * 2, 0x00020000);
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        const uint32_t row = (uint32_t)i * 2 + (lane >> 5);
        const uint32_t o = row * LD * 2 + wave * 512 + (lane & 31) * 16;
        lds_void_t* dst = (lds_void_t*)(smem + wave * 32768 + (i & 31) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, o, 0, 0, 0);
        if ((i & 15) == 15) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  if (MODE == 3) acc.x ^= *reinterpret_cast<uint32_t*>(smem + lane * 4);
  if (acc.x == 0x12345678u && acc.y == 7u) sink[blockIdx.x] = acc.z ^ acc.w;
}

template <int MODE, int DEPTH> int run(const uint16_t* buf, int panels, uint32_t* sink, const char* name) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int steps = 400;
  hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(256), dim3(256), 131072, 0, buf, 8, panels, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(256), dim3(256), 131072, 0, buf, steps, panels, sink);
  CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / steps, gbs = (double)ROWS * K * 2 / (us * 1e-6) / 1e9;
  printf("%-58s %7.2f us per 256 KB panel  %7.1f GB/s per CU  %6.2f TB/s chip\n", name, us, gbs, gbs * 256 / 1e3);
  return 0;
}

int main() {
  const int panels = 480;                                // 480 x 272 KB = 130 MB: each panel is fresh when read (as hs[l][t] is)
  uint16_t* buf; uint32_t* sink;
  CK(hipMalloc(&buf, (size_t)panels * ROWS * LD * 2)); CK(hipMemset(buf, 1, (size_t)panels * ROWS * LD * 2)); CK(hipMalloc(&sink, 4096));
  if (run<0, 16>(buf, panels, sink, "plain global_load -> VGPR fragments, 16 in flight")) return 1;
  if (run<0, 32>(buf, panels, sink, "plain global_load -> VGPR fragments, 32 in flight")) return 1;
  if (run<1, 16>(buf, panels, sink, "sc1 global_load -> VGPR fragments, 16 in flight")) return 1;
  if (run<1, 32>(buf, panels, sink, "sc1 global_load -> VGPR fragments, 32 in flight")) return 1;
  if (run<2, 32>(buf, panels, sink, "sc0 sc1 global_load -> VGPR fragments, 32 in flight")) return 1;
  if (run<3, 1>(buf, panels, sink, "LDS-DMA full rows, <= 24 in flight")) return 1;
  return 0;
}
