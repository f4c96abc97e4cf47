// Helpers shared by the weights-resident dataflow kernels (rnn_persist.hip: forward, rnn_persist_bwd.hip: backward): LDS accesses the compiler's
// waitcnt pass must not see, bounded flag polls, compile-time slot unrolling.
#pragma once
#include "common.hpp"
#include "tile_pipe.hpp"
#include <utility>

namespace {

__device__ __forceinline__ void lds_write128(uint32_t addr, const f32x4& v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 lds_read128f(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
template <int OFF> __device__ __forceinline__ f32x4 lds_rd(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

// Poll the 64 flag words of one (layer, step): one 256-byte sc1 load per round (lane i reads word i).  Returns false on timeout / abort.
__device__ __forceinline__ bool wait_flags(const uint32_t* f, const uint32_t* status, uint32_t limit, int lane) {
  for (uint32_t it = 0; it < limit; ++it) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(f + lane) : "memory");
    if (__builtin_amdgcn_ballot_w64(v != 0u) == ~0ull) return true;
    if ((it & 63) == 63) {                    // somebody else gave up: drain too
      uint32_t s;
      asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(s) : "v"(status) : "memory");
      if (__builtin_amdgcn_readfirstlane(s) != 0u) return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  return false;
}

__device__ __forceinline__ void wait_vm(int n) {      // n is a compile-time constant after unrolling: the switch folds to one s_waitcnt
  switch (n) {
#define WV(N) case N: wait_vmcnt<N>(); break;
    WV(1) WV(2) WV(3) WV(4) WV(5) WV(6) WV(7) WV(8) WV(9) WV(10) WV(11) WV(12) WV(13) WV(14) WV(15) WV(16) WV(17) WV(18) WV(19) WV(20)
    WV(21) WV(22) WV(23) WV(24) WV(25) WV(26) WV(27) WV(28) WV(29) WV(30)
#undef WV
    default: wait_vmcnt<0>(); break;
  }
}
__device__ __forceinline__ void wait_lgkm(int n) {
  switch (n) {
    case 7: wait_lgkmcnt<7>(); break; case 6: wait_lgkmcnt<6>(); break; case 5: wait_lgkmcnt<5>(); break; case 4: wait_lgkmcnt<4>(); break;
    case 3: wait_lgkmcnt<3>(); break; case 2: wait_lgkmcnt<2>(); break; case 1: wait_lgkmcnt<1>(); break; default: wait_lgkmcnt<0>(); break;
  }
}

template <typename F, int... S> __device__ __forceinline__ void for_each_slot(F&& f, std::integer_sequence<int, S...>) {
  (f(std::integral_constant<int, S>{}), ...);
}

}  // namespace
