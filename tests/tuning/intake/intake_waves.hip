// Per-CU LDS-DMA intake against the NUMBER of issuing waves and the bytes in flight (gfx950): 256 workgroups, each streaming a 4 MB slice
// shared by the 8 workgroups of an XCD (L2 / Infinity-Cache resident, like an operand panel) four times.  NW waves issue
// buffer_load_dwordx4 ... lds; the ring is DEPTH stages of STAGE_KB each: before issuing stage s the wave waits (counted vmcnt) until at
// most DEPTH - 1 stages are in flight.  With SYNC, a workgroup barrier per stage (what a consumer hand-off costs).
//   hipcc --offload-arch=gfx950 -O3 intake_waves.hip -o intake_waves && ./intake_waves
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void lds_void_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NW issuing waves; STAGE_KB per stage; pieces per wave per stage P = STAGE_KB / NW (1 KB per wave instruction)
template <int NW, int STAGE_KB, int DEPTH, bool SYNC>
__global__ __launch_bounds__(NW * 64) void intake_kernel(const char* buf, size_t slice_bytes, size_t total_bytes, int iters, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int P = STAGE_KB / NW;
  static_assert(P >= 1 && P * NW == STAGE_KB && (DEPTH - 1) * P < 64, "pieces");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const size_t base = ((size_t)(blockIdx.x % 32) * slice_bytes) % total_bytes;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(buf + base), 0, (int)slice_bytes, 0x00020000);
  const int steps = (int)(slice_bytes / (STAGE_KB * 1024));
  for (int it = 0; it < iters; ++it) {
    for (int s = 0; s < steps; ++s) {
      wait_vm<(DEPTH - 1) * P>();          // at most DEPTH - 1 stages of this wave still in flight -> the slot of stage s is free again
      if (SYNC) __builtin_amdgcn_s_barrier();
      const uint32_t off0 = (uint32_t)s * (STAGE_KB * 1024u);
      char* stage = smem + (s % DEPTH) * (STAGE_KB * 1024);
#pragma unroll
      for (int i = 0; i < P; ++i) {
        const uint32_t o = off0 + (uint32_t)(((i * NW + wave) * 64 + lane) * 16);
        lds_void_t* dst = (lds_void_t*)(stage + ((i * NW + wave) * 64) * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, o, 0, 0, 0);
      }
    }
  }
  wait_vm<0>();
  __syncthreads();
  uint32_t x = *reinterpret_cast<uint32_t*>(smem + lane * 4);
  if (x == 0x12345678u) sink[blockIdx.x] = x;
}

template <int NW, int STAGE_KB, int DEPTH, bool SYNC> int run(const char* buf, size_t slice, size_t total, uint32_t* sink) {
  auto k = intake_kernel<NW, STAGE_KB, DEPTH, SYNC>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 4;
  const size_t lds = (size_t)DEPTH * STAGE_KB * 1024;
  hipLaunchKernelGGL(k, dim3(256), dim3(NW * 64), lds, 0, buf, slice, total, 1, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k, dim3(256), dim3(NW * 64), lds, 0, buf, slice, total, iters, sink);
  CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double per_cu = (double)slice * iters / (ms * 1e-3) / 1e9;
  printf("waves %2d  stage %2d KB  ring %d (%3d KB in flight)  %s  %8.3f ms  %6.1f GB/s per CU  %6.2f TB/s chip\n", NW, STAGE_KB, DEPTH,
         (DEPTH - 1) * STAGE_KB, SYNC ? "barrier/stage" : "no barrier   ", ms, per_cu, per_cu * 256 / 1e3);
  return 0;
}

int main() {
  const size_t slice = 4u << 20, total = 128u << 20;
  char* buf; uint32_t* sink;
  CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total)); CK(hipMalloc(&sink, 4096));
  // the step kernels' shape: 32 KB stages, ring of 4 (96 KB in flight)
  if (run<2, 32, 4, false>(buf, slice, total, sink)) return 1;
  if (run<4, 32, 4, false>(buf, slice, total, sink)) return 1;
  if (run<8, 32, 4, false>(buf, slice, total, sink)) return 1;
  if (run<16, 32, 4, false>(buf, slice, total, sink)) return 1;
  if (run<4, 32, 4, true>(buf, slice, total, sink)) return 1;
  if (run<8, 32, 4, true>(buf, slice, total, sink)) return 1;
  // depth
  if (run<4, 32, 2, false>(buf, slice, total, sink)) return 1;
  if (run<4, 32, 3, false>(buf, slice, total, sink)) return 1;
  if (run<4, 32, 5, false>(buf, slice, total, sink)) return 1;
  if (run<8, 32, 5, false>(buf, slice, total, sink)) return 1;
  // small stages
  if (run<4, 16, 4, false>(buf, slice, total, sink)) return 1;
  if (run<4, 16, 8, false>(buf, slice, total, sink)) return 1;
  if (run<4, 64, 2, false>(buf, slice, total, sink)) return 1;
  if (run<8, 64, 2, false>(buf, slice, total, sink)) return 1;
  return 0;
}
