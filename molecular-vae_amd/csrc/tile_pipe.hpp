// Deep-pipelined tile mainloop for gfx950: operands go HBM/L2 -> LDS directly (buffer_load_dwordx4 ... lds),
// NBUF-deep LDS ring, counted s_waitcnt vmcnt + raw s_barrier (one barrier per K-step), so NBUF-1 K-steps of
// loads (tens of KB per CU) stay in flight under the MFMAs.  The LDS image of a stage is linear in lane order
// (a wave instruction writes 1 KiB = 8 rows x 128 B); the bank-conflict swizzle is applied to the per-lane
// SOURCE address and undone by the same XOR on the fragment reads (tile.hpp swz()).
// Out-of-matrix rows rely on the buffer descriptor's range check (they read as zero).
// Requirements (checked on the host): every K extent is a multiple of the K-step (128 bytes), operand
// buffers are < 2 GiB.
#pragma once
#include "tile.hpp"

constexpr uint32_t PIPE_OOB = 0x80000000u;   // voffset of an invalid row: beyond any num_records we accept

template <int BM, int BN> struct PipeSeg {
  __amdgpu_buffer_rsrc_t ra, rb;
  uint32_t offA[BM * 8 / 256];
  uint32_t offB[BN * 8 / 256];
  int nk;
};

// rowoff(r) -> byte offset of tile row r (k = 0) inside the operand buffer, or PIPE_OOB.
template <typename T, int BM, int BN, typename RowOffA, typename RowOffB>
__device__ __forceinline__ void pipe_seg_init(PipeSeg<BM, BN>& s, const void* A, uint32_t bytesA, const void* B, uint32_t bytesB,
                                              RowOffA rowoffA, RowOffB rowoffB, int K, int tid) {
  constexpr int KE = KB / (int)sizeof(T);
  s.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(A), 0, (int)bytesA, 0x00020000);
  s.rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(B), 0, (int)bytesB, 0x00020000);
  s.nk = (A != nullptr) ? K / KE : 0;
#pragma unroll
  for (int i = 0; i < BM * 8 / 256; ++i) {
    const int id = tid + i * 256, row = id >> 3, pos = id & 7;
    const uint32_t ro = rowoffA(row);
    s.offA[i] = (ro == PIPE_OOB) ? PIPE_OOB : ro + (uint32_t)(((pos ^ row) & 7) << 4);
  }
#pragma unroll
  for (int i = 0; i < BN * 8 / 256; ++i) {
    const int id = tid + i * 256, row = id >> 3, pos = id & 7;
    const uint32_t ro = rowoffB(row);
    s.offB[i] = (ro == PIPE_OOB) ? PIPE_OOB : ro + (uint32_t)(((pos ^ row) & 7) << 4);
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef __attribute__((address_space(3))) void lds_void_t;

// ---- fragment reads in inline asm -------------------------------------------------------------------------
// hipcc's waitcnt pass treats every LDS-DMA as a pending LDS write and puts s_waitcnt vmcnt(0) in front of any
// ds_read it can see, which would drain the ring every K-step.  The fragment reads are therefore issued from asm
// (invisible to that pass) and retired by our own counted s_waitcnt lgkmcnt; a sched_barrier keeps the MFMAs
// (register-only, so not ordered by a "memory" clobber) below the wait.
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int OFF> __device__ __forceinline__ u32x4 lds_read128(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int I, int N, int STRIDE> struct FragRead {
  static __device__ __forceinline__ void run(u32x4 (&dst)[N], uint32_t addr) {
    dst[I] = lds_read128<I * STRIDE>(addr);
    FragRead<I + 1, N, STRIDE>::run(dst, addr);
  }
};
template <int N, int STRIDE> struct FragRead<N, N, STRIDE> {
  static __device__ __forceinline__ void run(u32x4 (&)[N], uint32_t) {}
};
template <int N> __device__ __forceinline__ void wait_lgkmcnt() {
  __builtin_amdgcn_sched_barrier(0);     // nothing (in particular no earlier MFMA) may sink below / later MFMA rise above the wait
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// MFMAs of one K-step from LDS byte addresses (stage base already added).  BSTRIDE = LDS rows between
// consecutive n-sub-tiles of this wave.  a_lane / b_lane: per-lane byte offsets of (row l&15, chunk kk*4 + l>>4)
// for kk = 0,1 relative to the wave's first A / B row.
template <typename T, int MI, int NI, int BSTRIDE>
__device__ __forceinline__ void tile_mma_asm(uint32_t a_base, uint32_t b_base, const uint32_t (&a_lane)[2], const uint32_t (&b_lane)[2],
                                             f32x4 (&acc)[MI][NI]) {
  u32x4 a0[MI], b0[NI], a1[MI], b1[NI];
  FragRead<0, MI, 16 * KB>::run(a0, a_base + a_lane[0]);
  FragRead<0, NI, BSTRIDE * KB>::run(b0, b_base + b_lane[0]);
  FragRead<0, MI, 16 * KB>::run(a1, a_base + a_lane[1]);
  FragRead<0, NI, BSTRIDE * KB>::run(b1, b_base + b_lane[1]);
  wait_lgkmcnt<(MI + NI) < 16 ? (MI + NI) : 15>();
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      mma16<T>(__builtin_bit_cast(uint4, a0[mi]), __builtin_bit_cast(uint4, b0[ni]), acc[mi][ni]);
  wait_lgkmcnt<0>();
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      mma16<T>(__builtin_bit_cast(uint4, a1[mi]), __builtin_bit_cast(uint4, b1[ni]), acc[mi][ni]);
}

// Issue the (BM + BN) * 8 / 256 LDS-direct loads of pipeline stage `st` (K-steps of segment 0 first, then segment 1).
template <int BM, int BN, int NBUF>
__device__ __forceinline__ void pipe_issue_stage(char* smem, const PipeSeg<BM, BN>& s0, const PipeSeg<BM, BN>& s1, int st, int wave) {
#if defined(__HIP_DEVICE_COMPILE__)   // device pass only: the host pass cannot type-check LDS address-space casts / gfx950 builtins
  char* stage = smem + (st % NBUF) * ((BM + BN) * KB);
  const bool first = st < s0.nk;
  const PipeSeg<BM, BN>& s = first ? s0 : s1;
  const uint32_t kbyte = (uint32_t)(first ? st : st - s0.nk) * KB;
#pragma unroll
  for (int i = 0; i < BM * 8 / 256; ++i) {
    lds_void_t* dst = (lds_void_t*)(stage + (i * 256 + wave * 64) * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.ra, dst, 16, s.offA[i] + kbyte, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < BN * 8 / 256; ++i) {
    lds_void_t* dst = (lds_void_t*)(stage + BM * KB + (i * 256 + wave * 64) * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rb, dst, 16, s.offB[i] + kbyte, 0, 0, 0);
  }
#endif
}

// acc += sum over segment 0 then segment 1 of A_tile . B_tile^T.   smem: NBUF * (BM + BN) * 128 bytes.
template <typename T, int BM, int BN, int MI, int NI, int NBUF, int BSTRIDE>
__device__ __forceinline__ void tile_gemm_pipe(char* smem, const PipeSeg<BM, BN>& s0, const PipeSeg<BM, BN>& s1, int arow0,
                                               int brow0, f32x4 (&acc)[MI][NI], int tid) {
  constexpr int STAGE = (BM + BN) * KB;
  constexpr int LPS = (BM + BN) * 8 / 256;          // buffer loads per thread per stage
  static_assert((NBUF - 2) * LPS < 64, "vmcnt range");
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = s0.nk + s1.nk;
  if (nk <= 0) return;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;      // LDS byte address of the ring
  const int lr = lane & 15, lk = lane >> 4;
  uint32_t a_lane[2], b_lane[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    a_lane[kk] = (uint32_t)swz(arow0 + lr, kk * 4 + lk);
    b_lane[kk] = (uint32_t)swz(brow0 + lr, kk * 4 + lk) + BM * KB;
  }
#pragma unroll
  for (int s = 0; s < NBUF - 1; ++s)
    if (s < nk) pipe_issue_stage<BM, BN, NBUF>(smem, s0, s1, s, wave);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + NBUF - 2 < nk) wait_vmcnt<(NBUF - 2) * LPS>();   // stage kt landed; NBUF-2 newer stages stay in flight
    else wait_vmcnt<0>();                                     // tail: fewer stages were issued, drain
    __builtin_amdgcn_s_barrier();                             // every wave's part of stage kt landed; everyone is done reading stage kt-1
    asm volatile("" ::: "memory");                            // keep the fragment reads below the barrier
    if (kt + NBUF - 1 < nk) pipe_issue_stage<BM, BN, NBUF>(smem, s0, s1, kt + NBUF - 1, wave);   // refill the buffer stage kt-1 used
    const uint32_t st = lds0 + (uint32_t)((kt % NBUF) * STAGE);
    tile_mma_asm<T, MI, NI, BSTRIDE>(st, st, a_lane, b_lane, acc);
  }
  __builtin_amdgcn_s_barrier();                               // LDS free for the caller (epilogue scratch / next use)
}
