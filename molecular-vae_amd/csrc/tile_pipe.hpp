// Deep-pipelined tile mainloop for gfx950: operands go HBM/L2 -> LDS directly (buffer_load_dwordx4 ... lds),
// NBUF-deep LDS ring, counted s_waitcnt vmcnt + raw s_barrier (one barrier per K-step), so NBUF-1 K-steps of
// loads (tens of KB per CU) stay in flight under the MFMAs.  The LDS image of a stage is linear in lane order
// (a wave instruction writes 1 KiB = 8 rows x 128 B); the bank-conflict swizzle is applied to the per-lane
// SOURCE address and undone by the same XOR on the fragment reads (tile.hpp swz()).
// Out-of-matrix rows rely on the buffer descriptor's range check (they read as zero).
// Requirements (checked on the host): every K extent is a multiple of the K-step (128 bytes), operand
// buffers are < 2 GiB.
#pragma once
#include <type_traits>
#include "tile.hpp"

constexpr uint32_t PIPE_OOB = 0x80000000u;   // voffset of an invalid row: beyond any num_records we accept

// NT = threads that issue the LDS-DMA pieces of a stage (256: one piece set per wave 0-3; 512: all eight waves of a 512-thread workgroup)
template <int BM, int BN, int NT = 256> struct PipeSeg {
  __amdgpu_buffer_rsrc_t ra, rb;
  uint32_t offA[BM * 8 / NT];
  uint32_t offB[BN * 8 / NT];
  int nk;
  // optional hole in the K range: local stages >= hole_st read hole_bytes further along K (a zero block of the operands that is skipped:
  // GRU gate slots, rnn.hip).  Default: no hole.
  int hole_st; uint32_t hole_bytes;
};

// rowoff(r) -> byte offset of tile row r (k = 0) inside the operand buffer, or PIPE_OOB.
template <typename T, int BM, int BN, int NT = 256, typename RowOffA, typename RowOffB>
__device__ __forceinline__ void pipe_seg_init(PipeSeg<BM, BN, NT>& s, const void* A, uint32_t bytesA, const void* B, uint32_t bytesB,
                                              RowOffA rowoffA, RowOffB rowoffB, int K, int tid) {
  constexpr int KE = KB / (int)sizeof(T);
  s.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(A), 0, (int)bytesA, 0x00020000);
  s.rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(B), 0, (int)bytesB, 0x00020000);
  s.nk = (A != nullptr) ? K / KE : 0;
  s.hole_st = 0x7fffffff; s.hole_bytes = 0;
#pragma unroll
  for (int i = 0; i < BM * 8 / NT; ++i) {
    const int id = tid + i * NT, row = id >> 3, pos = id & 7;
    const uint32_t ro = rowoffA(row);
    s.offA[i] = (ro == PIPE_OOB) ? PIPE_OOB : ro + (uint32_t)(((pos ^ row) & 7) << 4);
  }
#pragma unroll
  for (int i = 0; i < BN * 8 / NT; ++i) {
    const int id = tid + i * NT, row = id >> 3, pos = id & 7;
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
// B fragments whose n-sub-tiles form groups: sub-tile I lives at LDS row (I / JS) * BOUT + (I % JS) * 16 of the wave's B region
// (plain GEMM: JS = N, one group of consecutive 16-row sub-tiles; LSTM forward: one group per gate, BOUT = rows per gate).
template <int I, int N, int JS, int BOUT> struct FragReadB {
  static __device__ __forceinline__ void run(u32x4 (&dst)[N], uint32_t addr) {
    dst[I] = lds_read128<((I / JS) * BOUT + (I % JS) * 16) * KB>(addr);
    FragReadB<I + 1, N, JS, BOUT>::run(dst, addr);
  }
};
template <int N, int JS, int BOUT> struct FragReadB<N, N, JS, BOUT> {
  static __device__ __forceinline__ void run(u32x4 (&)[N], uint32_t) {}
};
template <int N> __device__ __forceinline__ void wait_lgkmcnt() {
  __builtin_amdgcn_sched_barrier(0);     // nothing (in particular no earlier MFMA) may sink below / later MFMA rise above the wait
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// fp32 operands multiplied on the bf16 MFMA as  hi.hi + hi.lo + lo.hi  (x = hi + lo, hi = x truncated to bf16, lo = bf16(x - hi)):
// about 16 mantissa bits per product at 3/16 of the f32 MFMA's cycles.  Eight floats per lane (the two 16-byte chunks of a K-step) become
// one bf16x8 fragment each for hi and lo; A and B use the same element order, which is all a dot product needs.
__device__ __forceinline__ void split_x3(const u32x4& c0, const u32x4& c1, uint4& hi, uint4& lo) {
  const uint32_t x[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
  uint32_t h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t h0 = x[2 * j] & 0xffff0000u, h1 = x[2 * j + 1] & 0xffff0000u;
    const float l0 = __builtin_bit_cast(float, x[2 * j]) - __builtin_bit_cast(float, h0);
    const float l1 = __builtin_bit_cast(float, x[2 * j + 1]) - __builtin_bit_cast(float, h1);
    h[j] = (h0 >> 16) | h1;
    l[j] = (uint32_t)f2bf(l0) | ((uint32_t)f2bf(l1) << 16);
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]); lo = make_uint4(l[0], l[1], l[2], l[3]);
}
template <int MI, int NI, int JS, int BOUT>
__device__ __forceinline__ void tile_mma_asm_x3(uint32_t a_base, uint32_t b_base, const uint32_t (&a_lane)[2], const uint32_t (&b_lane)[2],
                                                f32x4 (&acc)[MI][NI]) {
  u32x4 a0[MI], b0[NI], a1[MI], b1[NI];
  FragRead<0, MI, 16 * KB>::run(a0, a_base + a_lane[0]);
  FragReadB<0, NI, JS, BOUT>::run(b0, b_base + b_lane[0]);
  FragRead<0, MI, 16 * KB>::run(a1, a_base + a_lane[1]);
  FragReadB<0, NI, JS, BOUT>::run(b1, b_base + b_lane[1]);
  wait_lgkmcnt<0>();
  uint4 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) split_x3(a0[mi], a1[mi], ah[mi], al[mi]);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) split_x3(b0[ni], b1[ni], bh[ni], bl[ni]);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      mma16<bf16_t>(al[mi], bh[ni], acc[mi][ni]);       // small terms first
      mma16<bf16_t>(ah[mi], bl[ni], acc[mi][ni]);
      mma16<bf16_t>(ah[mi], bh[ni], acc[mi][ni]);
    }
}

// MFMAs of one K-step from LDS byte addresses (stage base already added).  JS / BOUT: layout of the wave's n-sub-tiles
// (FragReadB).  a_lane / b_lane: per-lane byte offsets of (row l&15, chunk kk*4 + l>>4)
// for kk = 0,1 relative to the wave's first A / B row.
template <typename T, int MI, int NI, int JS, int BOUT>
__device__ __forceinline__ void tile_mma_asm(uint32_t a_base, uint32_t b_base, const uint32_t (&a_lane)[2], const uint32_t (&b_lane)[2],
                                             f32x4 (&acc)[MI][NI]) {
  u32x4 a0[MI], b0[NI], a1[MI], b1[NI];
  FragRead<0, MI, 16 * KB>::run(a0, a_base + a_lane[0]);
  FragReadB<0, NI, JS, BOUT>::run(b0, b_base + b_lane[0]);
  FragRead<0, MI, 16 * KB>::run(a1, a_base + a_lane[1]);
  FragReadB<0, NI, JS, BOUT>::run(b1, b_base + b_lane[1]);
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

__device__ __forceinline__ void ws_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// Issue the (BM + BN) * 8 / 256 LDS-direct loads of pipeline stage `st` (K-steps of segment 0 first, then segment 1).
template <int BM, int BN, int NBUF, int NT = 256>
__device__ __forceinline__ void pipe_issue_stage(char* smem, const PipeSeg<BM, BN, NT>& s0, const PipeSeg<BM, BN, NT>& s1, int st, int wave) {
#if defined(__HIP_DEVICE_COMPILE__)   // device pass only: the host pass cannot type-check LDS address-space casts / gfx950 builtins
  char* stage = smem + (st % NBUF) * ((BM + BN) * KB);
  const bool first = st < s0.nk;
  const PipeSeg<BM, BN, NT>& s = first ? s0 : s1;
  const int ls = first ? st : st - s0.nk;
  const uint32_t kbyte = (uint32_t)ls * KB + (ls >= s.hole_st ? s.hole_bytes : 0u);
#pragma unroll
  for (int i = 0; i < BM * 8 / NT; ++i) {
    lds_void_t* dst = (lds_void_t*)(stage + (i * NT + wave * 64) * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.ra, dst, 16, s.offA[i] + kbyte, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < BN * 8 / NT; ++i) {
    lds_void_t* dst = (lds_void_t*)(stage + BM * KB + (i * NT + wave * 64) * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rb, dst, 16, s.offB[i] + kbyte, 0, 0, 0);
  }
#endif
}

// acc += sum over segment 0 then segment 1 of A_tile . B_tile^T.   smem: NBUF * (BM + BN) * 128 bytes.
template <typename T, int BM, int BN, int MI, int NI, int NBUF, int JS, int BOUT, bool X3 = false>
__device__ __forceinline__ void tile_gemm_pipe(char* smem, const PipeSeg<BM, BN>& s0, const PipeSeg<BM, BN>& s1, int arow0,
                                               int brow0, f32x4 (&acc)[MI][NI], int tid) {
  static_assert(!X3 || std::is_same<T, float>::value, "the 3 x bf16 product form takes fp32 operands");
  constexpr int STAGE = (BM + BN) * KB;
  constexpr int LPS = (BM + BN) * 8 / 256;          // buffer loads per thread per stage
  static_assert((NBUF - 2) * LPS < 64, "vmcnt range");
  static_assert((BM + BN) * 8 % 256 == 0 && BM * 8 % 256 == 0, "whole LDS-DMA pieces per thread");
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
    if constexpr (X3) tile_mma_asm_x3<MI, NI, JS, BOUT>(st, st, a_lane, b_lane, acc);
    else tile_mma_asm<T, MI, NI, JS, BOUT>(st, st, a_lane, b_lane, acc);
  }
  __builtin_amdgcn_s_barrier();                               // LDS free for the caller (epilogue scratch / next use)
}


// Pieces [LO, HI) of the LPS = (BM + BN) * 8 / NT LDS-DMA pieces a thread issues per stage (A pieces first).
template <int BM, int BN, int NBUF, int NT, int LO, int HI>
__device__ __forceinline__ void pipe_issue_part(char* smem, const PipeSeg<BM, BN, NT>& s0, const PipeSeg<BM, BN, NT>& s1, int st, int wave) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NA = BM * 8 / NT, NB = BN * 8 / NT;
  char* stage = smem + (st % NBUF) * ((BM + BN) * KB);
  const bool first = st < s0.nk;
  const PipeSeg<BM, BN, NT>& s = first ? s0 : s1;
  const int ls = first ? st : st - s0.nk;
  const uint32_t kbyte = (uint32_t)ls * KB + (ls >= s.hole_st ? s.hole_bytes : 0u);
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    if (i < LO || i >= HI) continue;
    lds_void_t* dst = (lds_void_t*)(stage + (i * NT + wave * 64) * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.ra, dst, 16, s.offA[i] + kbyte, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    if (NA + i < LO || NA + i >= HI) continue;
    lds_void_t* dst = (lds_void_t*)(stage + BM * KB + (i * NT + wave * 64) * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rb, dst, 16, s.offB[i] + kbyte, 0, 0, 0);
  }
#endif
}

__device__ __forceinline__ void SB() { __builtin_amdgcn_sched_barrier(0); }     // pins the DMA / MFMA interleave as written

// MFMAs of fragment rows [M0, M1) of one K-half
template <typename T, int MI, int NI, int M0, int M1>
__device__ __forceinline__ void mma_rows(const u32x4 (&a)[MI], const u32x4 (&b)[NI], f32x4 (&acc)[MI][NI]) {
#pragma unroll
  for (int mi = M0; mi < M1; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      mma16<T>(__builtin_bit_cast(uint4, a[mi]), __builtin_bit_cast(uint4, b[ni]), acc[mi][ni]);
}

// All-waves form: NT threads (512 = eight waves, two per SIMD; 256 = four waves for small tiles), EVERY wave both issues its share of
// the LDS-DMA pieces and computes.  For tiles whose accumulators need all eight waves' registers (256 x 256: 128 accumulator VGPRs per
// wave).  A K-step is four MFMA blocks -- (K-half 0 | 1) x (fragment rows 0..MI/2-1 | MI/2..MI-1) -- and the A fragments of block i+1
// are read into the second of two MI/2-fragment register sets under the MFMAs of block i (B fragments: one set per K-half), so the
// fragment registers are 2 x (MI/2 + NI) instead of 2 x (MI + NI): what lets a 128 x 64 wave tile fit 256 VGPRs with room to spare.
//   barrier(kt) sits between blocks 2 and 3 of K-step kt-1: every wave has waited for its own pieces of stage kt (counted vmcnt) and
//   holds everything it needs of stage kt-1 in registers -> stage kt is readable, the buffer of stage kt-1 is free and is refilled with
//   stage kt+NBUF-1.
// The refill's DMA pieces cost their issuing wave ~100 cycles each, during which it issues no MFMA, and the two waves of a SIMD run
// this loop in lockstep (same barrier).  So (MODE bit 0) the pieces are spread over the rows of an MFMA block instead of issued as one
// burst, and (MODE bit 1) they are STAGGERED between the SIMD partners: waves 0..NT/128-1 interleave them with block 3 right behind the
// barrier, the other half of the waves with block 0 of the NEXT K-step -- while one partner issues DMA the other issues MFMAs.  (The
// counted vmcnt is the same for both: the deferred pieces are issued before the next wait.)
// One MFMA block (fragment rows M0 .. M0+GA-1 of one K-half) with the refill of stage `st` issued beside it when `refill` (wave-uniform)
// is set.  Only the DMA issue sits under the branch, never an MFMA: the accumulators stay in one straight-line live range.
template <typename T, int BM, int BN, int NBUF, int NT, int MI, int NI, int GA, bool SPREAD, int M0>
__device__ __forceinline__ void pipe_block_with_refill(char* smem, const PipeSeg<BM, BN, NT>& s0, const PipeSeg<BM, BN, NT>& s1, bool refill, int st,
                                                       int wave, const u32x4 (&a)[GA], const u32x4 (&b)[NI], f32x4 (&acc)[MI][NI]) {
  constexpr int LPS = (BM + BN) * 8 / NT;
  if constexpr (SPREAD && GA == 4 && LPS % 4 == 0) {
    constexpr int Q = LPS / 4;
    if (refill) pipe_issue_part<BM, BN, NBUF, NT, 0, Q>(smem, s0, s1, st, wave);
    SB();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) mma16<T>(__builtin_bit_cast(uint4, a[0]), __builtin_bit_cast(uint4, b[ni]), acc[M0 + 0][ni]);
    SB();
    if (refill) pipe_issue_part<BM, BN, NBUF, NT, Q, 2 * Q>(smem, s0, s1, st, wave);
    SB();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) mma16<T>(__builtin_bit_cast(uint4, a[1]), __builtin_bit_cast(uint4, b[ni]), acc[M0 + 1][ni]);
    SB();
    if (refill) pipe_issue_part<BM, BN, NBUF, NT, 2 * Q, 3 * Q>(smem, s0, s1, st, wave);
    SB();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) mma16<T>(__builtin_bit_cast(uint4, a[2]), __builtin_bit_cast(uint4, b[ni]), acc[M0 + 2][ni]);
    SB();
    if (refill) pipe_issue_part<BM, BN, NBUF, NT, 3 * Q, LPS>(smem, s0, s1, st, wave);
    SB();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) mma16<T>(__builtin_bit_cast(uint4, a[3]), __builtin_bit_cast(uint4, b[ni]), acc[M0 + 3][ni]);
  } else {
    if (refill) pipe_issue_stage<BM, BN, NBUF, NT>(smem, s0, s1, st, wave);
    SB();
#pragma unroll
    for (int r = 0; r < GA; ++r)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) mma16<T>(__builtin_bit_cast(uint4, a[r]), __builtin_bit_cast(uint4, b[ni]), acc[M0 + r][ni]);
  }
}
template <typename T, int MI, int NI, int GA, int M0>
__device__ __forceinline__ void pipe_block(const u32x4 (&a)[GA], const u32x4 (&b)[NI], f32x4 (&acc)[MI][NI]) {
#pragma unroll
  for (int r = 0; r < GA; ++r)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) mma16<T>(__builtin_bit_cast(uint4, a[r]), __builtin_bit_cast(uint4, b[ni]), acc[M0 + r][ni]);
}

template <typename T, int BM, int BN, int MI, int NI, int NBUF, int NT, int MODE = 3>
__device__ __forceinline__ void tile_gemm_pipe_all(char* smem, const PipeSeg<BM, BN, NT>& s0, const PipeSeg<BM, BN, NT>& s1, int arow0,
                                                   int brow0, f32x4 (&acc)[MI][NI], int tid) {
  constexpr int STAGE = (BM + BN) * KB;
  constexpr int LPS = (BM + BN) * 8 / NT;
  constexpr int GA = MI / 2;
  constexpr bool SPREAD = (MODE & 1) != 0, STAGGER = (MODE & 2) != 0 && NT == 512;
  static_assert(MI % 2 == 0 && NBUF >= 2 && (NBUF - 2) * LPS < 64, "vmcnt range");
  static_assert(GA + NI <= 15, "lgkmcnt range");
  static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "whole LDS-DMA pieces per thread");
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool late = STAGGER && wave >= NT / 128;                          // wave-uniform
  const int nk = s0.nk + s1.nk;
  if (nk <= 0) return;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  const int lr = lane & 15, lk = lane >> 4;
  uint32_t a_lane[2], b_lane[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    a_lane[kk] = (uint32_t)swz(arow0 + lr, kk * 4 + lk);
    b_lane[kk] = (uint32_t)swz(brow0 + lr, kk * 4 + lk) + BM * KB;
  }
  constexpr uint32_t G1 = GA * 16 * KB;                                   // byte offset of fragment rows GA.. inside the wave's A rows
#pragma unroll
  for (int s = 0; s < NBUF - 1; ++s)
    if (s < nk) pipe_issue_stage<BM, BN, NBUF, NT>(smem, s0, s1, s, wave);
  u32x4 aX[GA], aY[GA], bh0[NI], bh1[NI];
  if (NBUF - 2 < nk) wait_vmcnt<(NBUF - 2) * LPS>(); else wait_vmcnt<0>();
  ws_barrier();                                                           // barrier(0)
  int pend = (NBUF - 1 < nk) ? NBUF - 1 : -1;                             // stage whose refill is due (its buffer is free)
  if (pend >= 0 && !late) { pipe_issue_stage<BM, BN, NBUF, NT>(smem, s0, s1, pend, wave); pend = -1; }
  FragRead<0, GA, 16 * KB>::run(aX, lds0 + a_lane[0]);
  FragReadB<0, NI, NI, 0>::run(bh0, lds0 + b_lane[0]);
  for (int kt = 0; kt < nk; ++kt) {
    const uint32_t st = lds0 + (uint32_t)((kt % NBUF) * STAGE);
    FragRead<0, GA, 16 * KB>::run(aY, st + a_lane[0] + G1);              // (h0, rows GA..)
    FragReadB<0, NI, NI, 0>::run(bh1, st + b_lane[1]);                   // (h1)
    wait_lgkmcnt<GA + NI>();                                              // aX, bh0 arrived
    if constexpr (STAGGER) {                                              // block 0 (+ the staggered waves' refill)
      pipe_block_with_refill<T, BM, BN, NBUF, NT, MI, NI, GA, SPREAD, 0>(smem, s0, s1, late && pend >= 0, pend, wave, aX, bh0, acc);
      if (late) pend = -1;
    } else pipe_block<T, MI, NI, GA, 0>(aX, bh0, acc);
    SB();
    FragRead<0, GA, 16 * KB>::run(aX, st + a_lane[1]);                   // (h1, rows 0..GA-1)
    wait_lgkmcnt<NI + GA>();                                              // aY arrived
    pipe_block<T, MI, NI, GA, GA>(aY, bh0, acc);                          // block 1
    SB();
    FragRead<0, GA, 16 * KB>::run(aY, st + a_lane[1] + G1);              // (h1, rows GA..)
    wait_lgkmcnt<GA>();                                                   // aX, bh1 arrived
    pipe_block<T, MI, NI, GA, 0>(aX, bh1, acc);                           // block 2
    wait_lgkmcnt<0>();                                                    // aY arrived: every read of stage kt is done
    if (kt + 1 < nk) {
      if (kt + 1 + NBUF - 2 < nk) wait_vmcnt<(NBUF - 2) * LPS>(); else wait_vmcnt<0>();    // my pieces of stage kt+1 have landed
      ws_barrier();                                                       // barrier(kt + 1): the buffer stage kt used is free
      pend = (kt + NBUF < nk) ? kt + NBUF : -1;
      const uint32_t sn = lds0 + (uint32_t)(((kt + 1) % NBUF) * STAGE);
      FragRead<0, GA, 16 * KB>::run(aX, sn + a_lane[0]);
      FragReadB<0, NI, NI, 0>::run(bh0, sn + b_lane[0]);
    }
    pipe_block_with_refill<T, BM, BN, NBUF, NT, MI, NI, GA, SPREAD, GA>(smem, s0, s1, !late && pend >= 0, pend, wave, aY, bh1, acc);   // block 3
    if (!late) pend = -1;
    SB();
  }
}

// =====================================================================================================================
// TN form (bf16): C[m][n] = sum_k A[k][m] * B[k][n] with both operands K-major in memory (row = k, m / n contiguous):
// the weight-gradient contraction dW = dG^T . X reads dG [T*B, 4H] and X [T*B, H] exactly as the step kernels wrote them,
// so no transposed copies exist anywhere.  LDS stage = 64 k-rows x 128 columns per operand (256-byte rows); the MFMA
// fragments (8 consecutive k of one column per lane) come from the hardware transposed read ds_read_b64_tr_b16
// (4 k-rows x 16 columns per 16-lane group).  Bank conflicts: 8 rows of one half-wave would hit the same 8 banks, so the
// 32-byte column block index is XOR-ed with f(k) = (k & 3) | ((k >> 3) & 1) << 2 -- applied to the LDS-DMA SOURCE address
// (the LDS image itself is lane-linear) and to the read address.
// =====================================================================================================================
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

template <int OFF> __device__ __forceinline__ u32x2 lds_read_tr64(uint32_t addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

// Tile = (32*MI) x 128 (MI = 4: 128 x 128, MI = 8: 256 x 128); 4 waves as 2 (m) x 2 (n), wave tile (16*MI) x 64.
// Stage = 64 k-rows of A (row = 64*MI bytes) followed by 64 k-rows of B (256-byte rows).
template <int MI> struct PipeSegTN {
  __amdgpu_buffer_rsrc_t ra, rb;
  uint32_t offA[MI], offB[4];      // per-thread byte offsets of its MI + 4 LDS-DMA pieces (1 KiB per wave instruction) at k-step 0
  uint32_t kstepA, kstepB;         // bytes per 64-row K-step
  int nk;
};

__device__ __forceinline__ uint32_t tn_f(int krow) { return (uint32_t)((krow & 3) | (((krow >> 3) & 1) << 2)); }

// A: [K][lda] elements (bf16), tile columns [m0, m0 + 32*MI); B likewise with [n0, n0 + 128).  K rows beyond the matrix read as zero
// through the descriptor bound (num_records = K * ld * 2).
template <int MI>
__device__ __forceinline__ void pipe_seg_tn_init(PipeSegTN<MI>& s, const void* A, long lda, int m0, const void* B, long ldb, int n0,
                                                 int K, int tid) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int RA = 64 * MI;                    // bytes per A tile row
  constexpr int RPP = 1024 / RA;                 // A k-rows per 1-KiB piece (4 or 2)
  constexpr int CPR = RA / 16;                   // 16-byte chunks per A row (16 or 32)
  s.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(A), 0, (int)((long)K * lda * 2), 0x00020000);
  s.rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(B), 0, (int)((long)K * ldb * 2), 0x00020000);
  s.nk = (K + 63) / 64;
  s.kstepA = (uint32_t)(64 * lda * 2);
  s.kstepB = (uint32_t)(64 * ldb * 2);
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int ii = i * 4 + wave;                 // piece index inside the A part of the stage
    const int krow = ii * RPP + lane / CPR, pc = lane % CPR;
    // physical 16-byte chunk pc holds logical chunk lc: the low 3 bits of the 32-byte block index are XOR-ed with f(k)
    const uint32_t lc = (uint32_t)((((pc >> 1) ^ tn_f(krow)) << 1) | (pc & 1));
    s.offA[i] = (uint32_t)krow * (uint32_t)(lda * 2) + (uint32_t)(m0 * 2) + lc * 16u;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ii = i * 4 + wave;
    const int krow = ii * 4 + (lane >> 4), pc = lane & 15;
    const uint32_t lc = (uint32_t)((((pc >> 1) ^ tn_f(krow)) << 1) | (pc & 1));
    s.offB[i] = (uint32_t)krow * (uint32_t)(ldb * 2) + (uint32_t)(n0 * 2) + lc * 16u;
  }
#endif
}

template <int NBUF, int MI>
__device__ __forceinline__ void pipe_tn_issue(char* smem, const PipeSegTN<MI>& s, int st, int wave) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int ABYTES = 64 * 64 * MI, STAGE = ABYTES + 16384;
  char* stage = smem + (st % NBUF) * STAGE;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    lds_void_t* dst = (lds_void_t*)(stage + (i * 4 + wave) * 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.ra, dst, 16, s.offA[i] + (uint32_t)st * s.kstepA, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lds_void_t* dst = (lds_void_t*)(stage + ABYTES + (i * 4 + wave) * 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rb, dst, 16, s.offB[i] + (uint32_t)st * s.kstepB, 0, 0, 0);
  }
#endif
}

// fragment I of N: two transposed reads (k rows 8g..8g+3 and 8g+4..8g+7); ROWB = bytes per tile row
template <int I, int N, int ROWB> struct FragReadTN {
  template <int KOFF>
  static __device__ __forceinline__ void run(u32x4 (&dst)[N], const uint32_t (&addr)[N], uint32_t st) {
    const u32x2 lo = lds_read_tr64<KOFF>(st + addr[I]);
    const u32x2 hi = lds_read_tr64<KOFF + 4 * ROWB>(st + addr[I]);
    dst[I] = u32x4{lo[0], lo[1], hi[0], hi[1]};
    FragReadTN<I + 1, N, ROWB>::template run<KOFF>(dst, addr, st);
  }
};
template <int N, int ROWB> struct FragReadTN<N, N, ROWB> {
  template <int KOFF> static __device__ __forceinline__ void run(u32x4 (&)[N], const uint32_t (&)[N], uint32_t) {}
};

// smem: NBUF * (4096*MI + 16384) bytes.
template <int NBUF, int MI>
__device__ __forceinline__ void tile_gemm_pipe_tn(char* smem, const PipeSegTN<MI>& s, int wm, int wn, f32x4 (&acc)[MI][4], int tid) {
  constexpr int LPS = MI + 4;
  constexpr int RA = 64 * MI, ABYTES = 64 * RA, STAGE = ABYTES + 16384;
  static_assert((NBUF - 2) * LPS < 64, "vmcnt range");
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = s.nk;
  if (nk <= 0) return;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const uint32_t f = (uint32_t)(q | ((g & 1) << 2));
  uint32_t addrA[MI], addrB[4];
#pragma unroll
  for (int i = 0; i < MI; ++i) addrA[i] = (uint32_t)(8 * g + q) * (uint32_t)RA + ((((uint32_t)(wm * MI + i)) ^ f) << 5) + (uint32_t)pp * 8u;
#pragma unroll
  for (int i = 0; i < 4; ++i) addrB[i] = (uint32_t)(8 * g + q) * 256u + ((((uint32_t)(wn * 4 + i)) ^ f) << 5) + (uint32_t)pp * 8u + (uint32_t)ABYTES;
#pragma unroll
  for (int st = 0; st < NBUF - 1; ++st)
    if (st < nk) pipe_tn_issue<NBUF, MI>(smem, s, st, wave);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + NBUF - 2 < nk) wait_vmcnt<(NBUF - 2) * LPS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + NBUF - 1 < nk) pipe_tn_issue<NBUF, MI>(smem, s, kt + NBUF - 1, wave);
    const uint32_t st = lds0 + (uint32_t)((kt % NBUF) * STAGE);
    u32x4 a0[MI], b0[4], a1[MI], b1[4];
    FragReadTN<0, MI, RA>::template run<0>(a0, addrA, st);          // k rows 0..31 of the stage
    FragReadTN<0, 4, 256>::template run<0>(b0, addrB, st);
    wait_lgkmcnt<0>();
    FragReadTN<0, MI, RA>::template run<32 * RA>(a1, addrA, st);    // k rows 32..63: in flight under the first MFMAs
    FragReadTN<0, 4, 256>::template run<32 * 256>(b1, addrB, st);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        mma16<bf16_t>(__builtin_bit_cast(uint4, a0[mi]), __builtin_bit_cast(uint4, b0[ni]), acc[mi][ni]);
    wait_lgkmcnt<0>();
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        mma16<bf16_t>(__builtin_bit_cast(uint4, a1[mi]), __builtin_bit_cast(uint4, b1[ni]), acc[mi][ni]);
  }
  __builtin_amdgcn_s_barrier();
}

// =====================================================================================================================
// Wave-specialised main loops (512-thread workgroups): waves 0-3 are CONSUMERS (fragment reads + MFMA only), waves 4-7 are
// LOADERS (LDS-DMA issue + counted vmcnt only).  An LDS-DMA piece costs its issuing wave ~60-180 cycles of issue time; with
// one wave per SIMD doing both jobs those cycles come straight out of the MFMA stream.  With a loader wave beside every
// consumer wave on each SIMD the MFMA stream only stops for data.  One s_barrier per K-step carries both hand-offs:
// barrier(kt) = "stage kt has landed" (loaders waited for it) + "nobody still reads stage kt-1" (consumers drained lgkmcnt).
// The consumers are software-pipelined across the barrier: the first half of stage kt+1 is read under the MFMAs of the second
// half of stage kt, so no fragment-read latency is exposed after the prologue.
// =====================================================================================================================
template <int NBUF, int MI>
__device__ __forceinline__ void tile_gemm_ws_tn(char* smem, const PipeSegTN<MI>& s, int wm, int wn, f32x4 (&acc)[MI][4], int tid) {
  constexpr int LPS = MI + 4;
  constexpr int RA = 64 * MI, ABYTES = 64 * RA, STAGE = ABYTES + 16384;
  static_assert((NBUF - 2) * LPS < 64, "vmcnt range");
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = s.nk;
  if (nk <= 0) return;
  if (wave >= 4) {
    // ---- loader
#pragma unroll
    for (int st = 0; st < NBUF - 1; ++st)
      if (st < nk) pipe_tn_issue<NBUF, MI>(smem, s, st, wave - 4);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + NBUF - 2 < nk) wait_vmcnt<(NBUF - 2) * LPS>();
      else wait_vmcnt<0>();
      ws_barrier();                                                       // barrier(kt)
      if (kt + NBUF - 1 < nk) pipe_tn_issue<NBUF, MI>(smem, s, kt + NBUF - 1, wave - 4);
    }
  } else {
    // ---- consumer
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const uint32_t f = (uint32_t)(q | ((g & 1) << 2));
    uint32_t addrA[MI], addrB[4];
#pragma unroll
    for (int i = 0; i < MI; ++i) addrA[i] = (uint32_t)(8 * g + q) * (uint32_t)RA + ((((uint32_t)(wm * MI + i)) ^ f) << 5) + (uint32_t)pp * 8u;
#pragma unroll
    for (int i = 0; i < 4; ++i) addrB[i] = (uint32_t)(8 * g + q) * 256u + ((((uint32_t)(wn * 4 + i)) ^ f) << 5) + (uint32_t)pp * 8u + (uint32_t)ABYTES;
    u32x4 a0[MI], b0[4], a1[MI], b1[4];
    ws_barrier();                                                         // barrier(0)
    FragReadTN<0, MI, RA>::template run<0>(a0, addrA, lds0);
    FragReadTN<0, 4, 256>::template run<0>(b0, addrB, lds0);
    for (int kt = 0; kt < nk; ++kt) {
      const uint32_t st = lds0 + (uint32_t)((kt % NBUF) * STAGE);
      wait_lgkmcnt<0>();                                                  // first half of stage kt (issued under the previous MFMAs)
      FragReadTN<0, MI, RA>::template run<32 * RA>(a1, addrA, st);
      FragReadTN<0, 4, 256>::template run<32 * 256>(b1, addrB, st);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          mma16<bf16_t>(__builtin_bit_cast(uint4, a0[mi]), __builtin_bit_cast(uint4, b0[ni]), acc[mi][ni]);
      wait_lgkmcnt<0>();                                                  // second half: every read of stage kt is done
      if (kt + 1 < nk) {
        ws_barrier();                                                     // barrier(kt + 1)
        const uint32_t sn = lds0 + (uint32_t)(((kt + 1) % NBUF) * STAGE);
        FragReadTN<0, MI, RA>::template run<0>(a0, addrA, sn);
        FragReadTN<0, 4, 256>::template run<0>(b0, addrB, sn);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          mma16<bf16_t>(__builtin_bit_cast(uint4, a1[mi]), __builtin_bit_cast(uint4, b1[ni]), acc[mi][ni]);
    }
  }
  ws_barrier();
}

// One half K-step of a consumer wave with its fragment reads INTERLEAVED: MFMA m of the MI x NI block on fragments (a, b) is followed, for
// m < MI + NI, by ONE ds_read_b128 of the next fragment set (na, nb).  A consumer wave issues in order: eight reads in a block cost their
// LDS issue slots on top of the MFMA block (measured on the fused backward step, tune build: barriers alone 21.5 us per launch, + reads
// 34.0, + MFMAs 49.5, both 63.6 -- purely additive); issued one per MFMA gap they sit under the matrix pipe's 16 cycles per instruction.
// Order of the reads: a[0], b[0..NI-1], a[1..MI-1] -- what the next block's first MFMAs need comes first.
// read r of a fragment set, in the order a[0], b[0..NI-1], a[1..MI-1]
template <int MI, int NI, int JS, int BOUT, int R>
__device__ __forceinline__ void ilv_read(u32x4 (&na)[MI], u32x4 (&nb)[NI], uint32_t a_addr, uint32_t b_addr) {
  if constexpr (R == 0) na[0] = lds_read128<0>(a_addr);
  else if constexpr (R <= NI) nb[R - 1] = lds_read128<(((R - 1) / JS) * BOUT + ((R - 1) % JS) * 16) * KB>(b_addr);
  else if constexpr (R < MI + NI) na[R - NI] = lds_read128<(R - NI) * 16 * KB>(a_addr);
}
template <typename T, int MI, int NI, int JS, int BOUT, int M, bool RD = true> struct MmaIlv {
  static constexpr int RPG = (MI + NI + MI * NI - 1) / (MI * NI);      // reads per MFMA gap (1 unless the block has fewer MFMAs than fragments)
  static __device__ __forceinline__ void run(const u32x4 (&a)[MI], const u32x4 (&b)[NI], f32x4 (&acc)[MI][NI], u32x4 (&na)[MI], u32x4 (&nb)[NI],
                                             uint32_t a_addr, uint32_t b_addr) {
    // RD: this block requests fragments at all (not in the last K-step's second half)
    constexpr int mi = M / NI, ni = M % NI;
    mma16<T>(__builtin_bit_cast(uint4, a[mi]), __builtin_bit_cast(uint4, b[ni]), acc[mi][ni]);
    if constexpr (RD && M * RPG < MI + NI) {
      __builtin_amdgcn_sched_barrier(0);
      ilv_read<MI, NI, JS, BOUT, M * RPG>(na, nb, a_addr, b_addr);
      if constexpr (RPG > 1) ilv_read<MI, NI, JS, BOUT, M * RPG + 1>(na, nb, a_addr, b_addr);
      static_assert(RPG <= 2, "at most two reads per gap");
      __builtin_amdgcn_sched_barrier(0);
    }
    MmaIlv<T, MI, NI, JS, BOUT, M + 1, RD>::run(a, b, acc, na, nb, a_addr, b_addr);
  }
};
template <typename T, int MI, int NI, int JS, int BOUT, bool RD> struct MmaIlv<T, MI, NI, JS, BOUT, MI * NI, RD> {
  static __device__ __forceinline__ void run(const u32x4 (&)[MI], const u32x4 (&)[NI], f32x4 (&)[MI][NI], u32x4 (&)[MI], u32x4 (&)[NI], uint32_t, uint32_t) {}
};

// NT form of the wave-specialised loop (same contract as tile_gemm_pipe, 512 threads; s0 / s1 must have been initialised
// with tid & 255, so that loader wave w + 4 issues the pieces wave w issues in the 256-thread form).
// The two roles as separate functions (a caller that gives the roles different work around the loop -- the fused backward step, whose
// loader waves hold prefetched epilogue operands in their otherwise idle registers -- branches ONCE on the wave index and calls one of them;
// each must be followed by the closing ws_barrier() that tile_gemm_ws issues).
template <typename T, int BM, int BN, int NBUF>
__device__ __forceinline__ void tile_gemm_ws_loader(char* smem, const PipeSeg<BM, BN>& s0, const PipeSeg<BM, BN>& s1, int lwave, int dbg = 0) {
  constexpr int LPS = (BM + BN) * 8 / 256;
  static_assert((NBUF - 2) * LPS < 64, "vmcnt range");
  const int nk = s0.nk + s1.nk;
#ifdef MVAE_TUNING
  const bool issue = !(dbg & 8);            // diagnostic build, MVAE_DBG bit 3: the loader waves keep the barriers but move nothing
#else
  constexpr bool issue = true;
#endif
#pragma unroll
  for (int s = 0; s < NBUF - 1; ++s)
    if (s < nk && issue) pipe_issue_stage<BM, BN, NBUF>(smem, s0, s1, s, lwave);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + NBUF - 2 < nk) wait_vmcnt<(NBUF - 2) * LPS>();
    else wait_vmcnt<0>();
    ws_barrier();                                                       // barrier(kt)
    if (kt + NBUF - 1 < nk && issue) pipe_issue_stage<BM, BN, NBUF>(smem, s0, s1, kt + NBUF - 1, lwave);
  }
}
// Loader loop with a TAIL: right after the last stage has been issued the wave issues NTAIL more vector loads of its own (`tail()`: operands
// of the caller's epilogue, whose latency then runs under the last NBUF - 1 K-steps instead of in front of the first).  They are the
// YOUNGEST entries of the wave's memory queue, so the counted waits of the remaining iterations allow for them: in iteration
// nk - NBUF + j (j = 1 .. NBUF - 1) stage kt has landed once at most (NBUF - 1 - j) stages + NTAIL loads are outstanding.
// NTAIL must be exact (a smaller real count would let a stage be read before it has landed).  Requires nk >= NBUF.
template <typename T, int BM, int BN, int NBUF, int NTAIL, typename Tail>
__device__ __forceinline__ void tile_gemm_ws_loader_tail(char* smem, const PipeSeg<BM, BN>& s0, const PipeSeg<BM, BN>& s1, int lwave, Tail tail) {
  constexpr int LPS = (BM + BN) * 8 / 256;
  static_assert((NBUF - 2) * LPS + NTAIL < 64, "vmcnt range");
  static_assert(NBUF == 4, "tail waits written out for a ring of four");
  const int nk = s0.nk + s1.nk;
#pragma unroll
  for (int s = 0; s < NBUF - 1; ++s) pipe_issue_stage<BM, BN, NBUF>(smem, s0, s1, s, lwave);
  for (int kt = 0; kt <= nk - NBUF; ++kt) {
    wait_vmcnt<(NBUF - 2) * LPS>();
    ws_barrier();                                                       // barrier(kt)
    pipe_issue_stage<BM, BN, NBUF>(smem, s0, s1, kt + NBUF - 1, lwave);
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");                                        // the tail loads stay behind the last stage's
  tail();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  wait_vmcnt<2 * LPS + NTAIL>(); ws_barrier();                          // barrier(nk - 3)
  wait_vmcnt<1 * LPS + NTAIL>(); ws_barrier();                          // barrier(nk - 2)
  wait_vmcnt<NTAIL>(); ws_barrier();                                    // barrier(nk - 1)
}
template <typename T, int BM, int BN, int MI, int NI, int NBUF, int JS, int BOUT, bool ILV = false>
__device__ __forceinline__ void tile_gemm_ws_consumer(char* smem, int nk, int arow0, int brow0, f32x4 (&acc)[MI][NI], int lane, int dbg = 0) {
#ifdef MVAE_TUNING
  // diagnostic build: MVAE_DBG bit 4 = no MFMAs, bit 5 = no fragment reads (barriers kept) -- which side of the hand-off sets the K-step
  const bool do_mma = !(dbg & 16), do_rd = !(dbg & 32);
#else
  constexpr bool do_mma = true, do_rd = true;
#endif
  constexpr int STAGE = (BM + BN) * KB;
  static_assert(MI + NI <= 15, "lgkmcnt range");
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  const int lr = lane & 15, lk = lane >> 4;
  uint32_t a_lane[2], b_lane[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    a_lane[kk] = (uint32_t)swz(arow0 + lr, kk * 4 + lk);
    b_lane[kk] = (uint32_t)swz(brow0 + lr, kk * 4 + lk) + BM * KB;
  }
  u32x4 a0[MI], b0[NI], a1[MI], b1[NI];
  ws_barrier();                                                         // barrier(0)
  if (ILV && do_mma && do_rd) {            // (the diagnostic build's partial loops take the blocked form below)
  FragRead<0, MI, 16 * KB>::run(a0, lds0 + a_lane[0]);
  FragReadB<0, NI, JS, BOUT>::run(b0, lds0 + b_lane[0]);
  wait_lgkmcnt<0>();
  for (int kt = 0; kt < nk; ++kt) {
    const uint32_t st = lds0 + (uint32_t)((kt % NBUF) * STAGE);
    // first half of stage kt on (a0, b0); the second-half fragments (a1, b1) of the same stage are requested in its first MFMA gaps
    MmaIlv<T, MI, NI, JS, BOUT, 0>::run(a0, b0, acc, a1, b1, st + a_lane[1], st + b_lane[1]);
    wait_lgkmcnt<0>();                                                  // every read of stage kt is done
    if (kt + 1 < nk) {
      ws_barrier();                                                     // barrier(kt + 1)
      const uint32_t sn = lds0 + (uint32_t)(((kt + 1) % NBUF) * STAGE);
      // second half of stage kt on (a1, b1); the first-half fragments of stage kt + 1 are requested in its first MFMA gaps
      MmaIlv<T, MI, NI, JS, BOUT, 0>::run(a1, b1, acc, a0, b0, sn + a_lane[0], sn + b_lane[0]);
      wait_lgkmcnt<0>();
    } else {
      MmaIlv<T, MI, NI, JS, BOUT, 0, false>::run(a1, b1, acc, a0, b0, 0u, 0u);      // last K-step: nothing left to request
    }
  }
  } else {
  if (do_rd) {
  FragRead<0, MI, 16 * KB>::run(a0, lds0 + a_lane[0]);
  FragReadB<0, NI, JS, BOUT>::run(b0, lds0 + b_lane[0]);
  }
  for (int kt = 0; kt < nk; ++kt) {
    const uint32_t st = lds0 + (uint32_t)((kt % NBUF) * STAGE);
    if (do_rd) {
    FragRead<0, MI, 16 * KB>::run(a1, st + a_lane[1]);
    FragReadB<0, NI, JS, BOUT>::run(b1, st + b_lane[1]);
    }
    wait_lgkmcnt<MI + NI>();                                            // first half arrived, second half in flight
    if (do_mma) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        mma16<T>(__builtin_bit_cast(uint4, a0[mi]), __builtin_bit_cast(uint4, b0[ni]), acc[mi][ni]);
    }
    wait_lgkmcnt<0>();                                                  // every read of stage kt is done
    if (kt + 1 < nk) {
      ws_barrier();                                                     // barrier(kt + 1)
      const uint32_t sn = lds0 + (uint32_t)(((kt + 1) % NBUF) * STAGE);
      if (do_rd) {
      FragRead<0, MI, 16 * KB>::run(a0, sn + a_lane[0]);
      FragReadB<0, NI, JS, BOUT>::run(b0, sn + b_lane[0]);
      }
    }
    if (do_mma) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        mma16<T>(__builtin_bit_cast(uint4, a1[mi]), __builtin_bit_cast(uint4, b1[ni]), acc[mi][ni]);
    }
  }
  }
}

template <typename T, int BM, int BN, int MI, int NI, int NBUF, int JS, int BOUT, bool ILV = false>
__device__ __forceinline__ void tile_gemm_ws(char* smem, const PipeSeg<BM, BN>& s0, const PipeSeg<BM, BN>& s1, int arow0, int brow0,
                                             f32x4 (&acc)[MI][NI], int tid) {
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = s0.nk + s1.nk;
  if (nk <= 0) return;
  if (wave >= 4) tile_gemm_ws_loader<T, BM, BN, NBUF>(smem, s0, s1, wave - 4);
  else tile_gemm_ws_consumer<T, BM, BN, MI, NI, NBUF, JS, BOUT, ILV>(smem, nk, arow0, brow0, acc, lane);
  ws_barrier();
}

// =====================================================================================================================
// TN form, 256 x 256 tile, 512 threads = 8 waves as 2 (m) x 4 (n), wave tile 128 x 64, every wave both loads and computes
// (two waves per SIMD: one wave's LDS-DMA issue stalls sit under the other's MFMAs).  A 256 x 256 tile needs 64 KB of operands
// per 64-deep K-step for 8.4 MFLOP -- a third less per FLOP than 256 x 128 -- which is what matters once the loop is bound by
// the per-CU L2->LDS intake.  Stage = 64 k-rows of A (512-byte rows) + 64 k-rows of B (512-byte rows) = 64 KiB, double-buffered.
// =====================================================================================================================
struct PipeSegTN2 {
  __amdgpu_buffer_rsrc_t ra, rb;
  uint32_t offA[4], offB[4];
  uint32_t kstepA, kstepB;
  int nk;
};

__device__ __forceinline__ void pipe_seg_tn2_init(PipeSegTN2& s, const void* A, long lda, int m0, const void* B, long ldb, int n0, int K, int tid) {
#if defined(__HIP_DEVICE_COMPILE__)
  s.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(A), 0, (int)((long)K * lda * 2), 0x00020000);
  s.rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(B), 0, (int)((long)K * ldb * 2), 0x00020000);
  s.nk = (K + 63) / 64;
  s.kstepA = (uint32_t)(64 * lda * 2);
  s.kstepB = (uint32_t)(64 * ldb * 2);
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ii = i * 8 + wave;                 // 1-KiB piece = 2 k-rows x 512 bytes
    const int krow = ii * 2 + (lane >> 5), pc = lane & 31;
    const uint32_t lc = (uint32_t)((((pc >> 1) ^ tn_f(krow)) << 1) | (pc & 1));
    s.offA[i] = (uint32_t)krow * (uint32_t)(lda * 2) + (uint32_t)(m0 * 2) + lc * 16u;
    s.offB[i] = (uint32_t)krow * (uint32_t)(ldb * 2) + (uint32_t)(n0 * 2) + lc * 16u;
  }
#endif
}

__device__ __forceinline__ void pipe_tn2_issue(char* smem, const PipeSegTN2& s, int st, int wave) {
#if defined(__HIP_DEVICE_COMPILE__)
  char* stage = smem + (st & 1) * 65536;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lds_void_t* dst = (lds_void_t*)(stage + (i * 8 + wave) * 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.ra, dst, 16, s.offA[i] + (uint32_t)st * s.kstepA, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lds_void_t* dst = (lds_void_t*)(stage + 32768 + (i * 8 + wave) * 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rb, dst, 16, s.offB[i] + (uint32_t)st * s.kstepB, 0, 0, 0);
  }
#endif
}

// Phase 1 of a 256 x 256 TN K-step with the NEXT half's fragment reads interleaved: MFMA m (mi = m / 4, ni = m % 4) on (a0, b0) is followed,
// for m < 24, by ONE ds_read_b64_tr_b16 -- read r = m fills half (r & 1) of fragment r >> 1 of the second K-half (fragments 0-7 = a1, 8-11 = b1).
// In a block in front of the MFMAs those 24 reads cost their issue slots on top of the matrix pipe's time (the waves issue in order).
template <int M> struct TnIlv {
  static __device__ __forceinline__ void run(const u32x4 (&a0)[8], const u32x4 (&b0)[4], f32x4 (&acc)[8][4], u32x2 (&lo)[12], u32x2 (&hi)[12],
                                             const uint32_t (&addrA)[8], const uint32_t (&addrB)[4], uint32_t st) {
    constexpr int mi = M / 4, ni = M % 4;
    mma16<bf16_t>(__builtin_bit_cast(uint4, a0[mi]), __builtin_bit_cast(uint4, b0[ni]), acc[mi][ni]);
    if constexpr (M < 24) {
      constexpr int F = M >> 1;
      __builtin_amdgcn_sched_barrier(0);
      const uint32_t ad = st + (F < 8 ? addrA[F < 8 ? F : 0] : addrB[F < 8 ? 0 : F - 8]);
      if constexpr ((M & 1) == 0) lo[F] = lds_read_tr64<32 * 512>(ad);
      else hi[F] = lds_read_tr64<32 * 512 + 4 * 512>(ad);
      __builtin_amdgcn_sched_barrier(0);
    }
    TnIlv<M + 1>::run(a0, b0, acc, lo, hi, addrA, addrB, st);
  }
};
template <> struct TnIlv<32> {
  static __device__ __forceinline__ void run(const u32x4 (&)[8], const u32x4 (&)[4], f32x4 (&)[8][4], u32x2 (&)[12], u32x2 (&)[12], const uint32_t (&)[8],
                                             const uint32_t (&)[4], uint32_t) {}
};
// smem: 2 x 64 KiB.  wm in {0,1}, wn in {0..3}.
// COLSUM variant: the column sums of A (sum_k A[k][m], the bias gradient when A = dG) ride along.  The 16 (tile column, wave column)
// pairs that share one A row panel split its 8 fragments x 2 K-halves between them: pair c does fragment cs_mi = c % 8 of half
// cs_half = c / 8 with ONE extra MFMA per K-step against a fragment of ones (accb: every n column identical) -- balanced over all
// waves (+1.5 % MFMAs), no extra pass over dG.  The two halves of a fragment are added by the split-K reduction.
template <bool COLSUM>
__device__ __forceinline__ void tile_gemm_tn_256(char* smem, const PipeSegTN2& s, int wm, int wn, f32x4 (&acc)[8][4], int cs_mi, int cs_half, f32x4& accb,
                                                 int tid) {
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = s.nk;
  if (nk <= 0) return;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const uint32_t f = (uint32_t)(q | ((g & 1) << 2));
  uint32_t addrA[8], addrB[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) addrA[i] = (uint32_t)(8 * g + q) * 512u + ((((uint32_t)(wm * 8 + i)) ^ f) << 5) + (uint32_t)pp * 8u;
#pragma unroll
  for (int i = 0; i < 4; ++i) addrB[i] = (uint32_t)(8 * g + q) * 512u + ((((uint32_t)(wn * 4 + i)) ^ f) << 5) + (uint32_t)pp * 8u + 32768u;
  const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);      // eight bf16 1.0
  // the fragment whose column sums this wave accumulates is read once more from LDS (its index is only known at run time)
  uint32_t addrS[1];
  addrS[0] = (uint32_t)(8 * g + q) * 512u + ((((uint32_t)(wm * 8 + (cs_mi & 7))) ^ f) << 5) + (uint32_t)pp * 8u + (cs_half == 1 ? 32u * 512u : 0u);
  pipe_tn2_issue(smem, s, 0, wave);
  for (int kt = 0; kt < nk; ++kt) {
    wait_vmcnt<0>();                                         // my pieces of stage kt have landed
    ws_barrier();                                            // everyone's have; nobody still reads stage kt-1
    const uint32_t st = lds0 + (uint32_t)((kt & 1) * 65536);
    u32x4 a0[8], b0[4], a1[8], b1[4], as[1];
    u32x2 lo[12], hi[12];
    FragReadTN<0, 8, 512>::template run<0>(a0, addrA, st);
    FragReadTN<0, 4, 512>::template run<0>(b0, addrB, st);
    if constexpr (COLSUM) FragReadTN<0, 1, 512>::template run<0>(as, addrS, st);
    // the refill of the other buffer goes out BEHIND the first half's reads: their LDS latency runs under the LDS-DMA issue
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    if (kt + 1 < nk) pipe_tn2_issue(smem, s, kt + 1, wave);  // into the buffer stage kt-1 used
    wait_lgkmcnt<0>();
    TnIlv<0>::run(a0, b0, acc, lo, hi, addrA, addrB, st);
    if constexpr (COLSUM) {
      if (cs_half >= 0) mma16<bf16_t>(__builtin_bit_cast(uint4, as[0]), ones, accb);      // wave-uniform
    }
    wait_lgkmcnt<0>();
#pragma unroll
    for (int i = 0; i < 8; ++i) a1[i] = u32x4{lo[i][0], lo[i][1], hi[i][0], hi[i][1]};
#pragma unroll
    for (int i = 0; i < 4; ++i) b1[i] = u32x4{lo[8 + i][0], lo[8 + i][1], hi[8 + i][0], hi[8 + i][1]};
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        mma16<bf16_t>(__builtin_bit_cast(uint4, a1[mi]), __builtin_bit_cast(uint4, b1[ni]), acc[mi][ni]);
  }
  ws_barrier();
}
