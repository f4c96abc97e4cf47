// Do a wave's vector-memory operations retire in issue order ACROSS loads and stores (gfx950)?  Each wave issues one load from a cold
// (HBM-resident, never touched) line, then four stores to a small hot buffer, then `s_waitcnt vmcnt(4)`.  If retirement is in order the load
// has arrived behind that wait; if store acknowledgements can overtake an older load the destination register still holds the sentinel.
//   hipcc --offload-arch=gfx950 -O3 vmcnt_order.hip -o vmcnt_order && ./vmcnt_order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void probe(const uint32_t* cold, uint32_t* hot, uint32_t* early, uint32_t* late, size_t stride) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t* src = cold + gid * stride;               // one 4-byte word per thread, lines far apart: HBM misses
  uint32_t* dst = hot + (gid & 4095) * 4;
  uint32_t v = 0xdeadbeefu, v2;
  asm volatile("global_load_dword %0, %1, off" : "+v"(v) : "v"(src) : "memory");
  asm volatile("global_store_dword %0, %1, off\n\tglobal_store_dword %0, %1, off offset:4\n\tglobal_store_dword %0, %1, off offset:8\n\tglobal_store_dword %0, %1, off offset:12"
               :: "v"(dst), "v"((uint32_t)gid) : "memory");
  asm volatile("s_waitcnt vmcnt(4)\n\tv_mov_b32 %0, %1" : "=v"(v2) : "v"(v) : "memory");      // snapshot of the load's register behind the counted wait
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  early[gid] = v2;
  late[gid] = v;
}

int main() {
  const int blocks = 1024, n = blocks * 256;
  const size_t stride = 1024;                              // 4 KB between the words of neighbouring threads
  uint32_t *cold, *hot, *early, *late;
  CK(hipMalloc(&cold, (size_t)n * stride * 4)); CK(hipMemset(cold, 0x5a, (size_t)n * stride * 4));
  CK(hipMalloc(&hot, 4096 * 16)); CK(hipMalloc(&early, n * 4)); CK(hipMalloc(&late, n * 4));
  // evict: touch another large buffer
  uint32_t* junk; CK(hipMalloc(&junk, 1u << 30)); CK(hipMemset(junk, 1, 1u << 30)); CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, cold, hot, early, late, stride);
  CK(hipDeviceSynchronize());
  uint32_t* he = (uint32_t*)malloc(n * 4); uint32_t* hl = (uint32_t*)malloc(n * 4);
  CK(hipMemcpy(he, early, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hl, late, n * 4, hipMemcpyDeviceToHost));
  long stale = 0, bad_late = 0;
  for (int i = 0; i < n; ++i) { if (he[i] != 0x5a5a5a5au) ++stale; if (hl[i] != 0x5a5a5a5au) ++bad_late; }
  printf("threads %d: load not yet arrived behind s_waitcnt vmcnt(4) with 4 younger stores: %ld (%.2f %%); after vmcnt(0): %ld wrong\n", n, stale,
         100.0 * stale / n, bad_late);
  printf(stale ? "=> store acknowledgements overtake an older load: vmcnt is NOT in issue order across loads and stores\n"
               : "=> no overtaking observed in this run\n");
  return 0;
}
