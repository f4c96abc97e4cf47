// Per-CU operand intake micro-benchmark (gfx950): 256 workgroups x 512 threads, each streams its own slice of a buffer that is re-read by
// other workgroups (L2 / Infinity-Cache resident working set), through
//   mode 0: LDS-DMA only (buffer_load_dwordx4 ... lds), all 8 waves
//   mode 1: global_load_dwordx4 -> VGPR only (xor-accumulated, no LDS write), all 8 waves
//   mode 2: waves 0-3 LDS-DMA, waves 4-7 global_load -> VGPR -> ds_write_b128
//   mode 3: global_load -> VGPR -> ds_write_b128, all 8 waves
// Prints GB/s per CU.   hipcc --offload-arch=gfx950 -O3 intake.hip -o intake && ./intake
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void lds_void_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void intake_kernel(const char* buf, size_t slice_bytes, size_t total_bytes, int iters, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // the slice of this workgroup: shared with 7 others (same slice index modulo): re-reads hit L2 / MALL like a weight panel does
  const size_t base = ((size_t)(blockIdx.x % 32) * slice_bytes) % total_bytes;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(buf + base), 0, (int)slice_bytes, 0x00020000);
  uint4 acc = make_uint4(0, 0, 0, 0);
  const int PIECES = 8;                           // 16-byte pieces per thread per step: 512 thr x 8 x 16 B = 64 KB per step
  const int steps = (int)(slice_bytes / 65536);
  for (int it = 0; it < iters; ++it) {
    for (int s = 0; s < steps; ++s) {
      const uint32_t off0 = (uint32_t)s * 65536u;
      char* stage = smem + (s & 1) * 65536;
      if (MODE == 0 || (MODE == 2 && wave < 4)) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
          const uint32_t o = off0 + (uint32_t)((i * 512 + tid) * 16);
          lds_void_t* dst = (lds_void_t*)(stage + (i * 512 + wave * 64) * 16);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, o, 0, 0, 0);
        }
      } else {
        uint4 v[PIECES];
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
          const uint32_t o = off0 + (uint32_t)((i * 512 + tid) * 16);
          v[i] = *reinterpret_cast<const uint4*>(buf + base + o);
        }
        if (MODE == 1) {
#pragma unroll
          for (int i = 0; i < PIECES; ++i) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
        } else {
#pragma unroll
          for (int i = 0; i < PIECES; ++i) *reinterpret_cast<uint4*>(stage + (i * 512 + tid) * 16) = v[i];
        }
      }
      if ((s & 3) == 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE != 1) acc.x ^= *reinterpret_cast<uint32_t*>(smem + lane * 4);
  if (acc.x == 0x12345678u && acc.y == 7u) sink[blockIdx.x] = acc.z ^ acc.w;
}

template <int MODE> int run(const char* buf, size_t slice, size_t total, uint32_t* sink, const char* name) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(intake_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 4;
  hipLaunchKernelGGL(intake_kernel<MODE>, dim3(256), dim3(512), 131072, 0, buf, slice, total, 1, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(intake_kernel<MODE>, dim3(256), dim3(512), 131072, 0, buf, slice, total, iters, sink);
  CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double per_cu = (double)slice * iters / (ms * 1e-3) / 1e9;
  printf("%-44s %8.3f ms  %7.1f GB/s per CU  %6.2f TB/s chip\n", name, ms, per_cu, per_cu * 256 / 1e3);
  return 0;
}

int main() {
  const size_t slice = 4u << 20, total = 128u << 20;       // 4 MB per workgroup per pass (a backward step tile streams 4 MB), 128 MB buffer
  char* buf; uint32_t* sink;
  CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total)); CK(hipMalloc(&sink, 4096));
  if (run<0>(buf, slice, total, sink, "LDS-DMA, 8 waves")) return 1;
  if (run<1>(buf, slice, total, sink, "global_load -> VGPR (no LDS), 8 waves")) return 1;
  if (run<2>(buf, slice, total, sink, "4 waves LDS-DMA + 4 waves load/ds_write")) return 1;
  if (run<3>(buf, slice, total, sink, "global_load -> VGPR -> ds_write, 8 waves")) return 1;
  return 0;
}
