// Workgroup-level tile machinery shared by the GEMM and the recurrent-step kernels.
//   256 threads = 4 waves (2 x 2); operands are K-contiguous ("NT" form: C[m][n] = sum_k A[m][k] * B[n][k]).
//   One K-step = 128 bytes of K per row (64 bf16 / 32 f32), register-staged into a swizzled LDS image,
//   double-buffered: global loads of step k+1 are issued before the MFMAs of step k and written to the
//   other LDS buffer after them; one barrier per K-step.
#pragma once
#include "common.hpp"

template <int ROWS> struct StageRegs { uint4 r[ROWS * 8 / 256]; };

// global -> registers for one K-step of a ROWS-row tile. rowptr(r) returns the address of tile row r at
// element k = 0 (or nullptr when the row is outside the matrix). [k0, kend) is the valid element range;
// a chunk is loaded when its FIRST element is inside the range (buffers are zero-padded to chunk size).
template <typename T, int ROWS, typename RowPtr>
__device__ __forceinline__ void stage_load(StageRegs<ROWS>& s, RowPtr rowptr, long k0, long kend, int tid) {
  constexpr int EPC = TT<T>::EPC;
#pragma unroll
  for (int i = 0; i < ROWS * 8 / 256; ++i) {
    const int id = tid + i * 256;
    const int row = id >> 3, c = id & 7;
    const T* p = rowptr(row);
    const long k = k0 + c * EPC;
    if (p != nullptr && k < kend) s.r[i] = *reinterpret_cast<const uint4*>(p + k);
    else s.r[i] = make_uint4(0u, 0u, 0u, 0u);
  }
}

template <int ROWS>
__device__ __forceinline__ void stage_store(char* lds, const StageRegs<ROWS>& s, int tid) {
#pragma unroll
  for (int i = 0; i < ROWS * 8 / 256; ++i) {
    const int id = tid + i * 256;
    const int row = id >> 3, c = id & 7;
    *reinterpret_cast<uint4*>(lds + swz(row, c)) = s.r[i];
  }
}

// MFMAs of one K-step for one wave: MI x NI 16x16 sub-tiles.
// arow0: LDS row of the wave's first A row; brow[ni]: LDS row of the first B row of n-sub-tile ni.
template <typename T, int MI, int NI>
__device__ __forceinline__ void tile_mma(const char* As, const char* Bs, int arow0, const int (&brow)[NI],
                                         f32x4 (&acc)[MI][NI], int lane) {
  const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int chunk = kk * 4 + lk;
    uint4 a[MI], b[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const uint4*>(As + swz(arow0 + mi * 16 + lr, chunk));
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const uint4*>(Bs + swz(brow[ni] + lr, chunk));
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) mma16<T>(a[mi], b[ni], acc[mi][ni]);
  }
}

// acc += A_tile[BM x K] * B_tile[BN x K]^T over the element range [kbeg, kend) (kbeg a multiple of the
// K-step).  smem: 2 * (BM + BN) * 128 bytes.  Ends with a barrier, so it can be called again (next segment).
template <typename T, int BM, int BN, int MI, int NI, typename RowA, typename RowB>
__device__ __forceinline__ void tile_gemm_segment(char* smem, RowA rowA, RowB rowB, long kbeg, long kend, int arow0,
                                                  const int (&brow)[NI], f32x4 (&acc)[MI][NI], int tid) {
  constexpr int KE = KB / (int)sizeof(T);
  const int lane = tid & 63;
  char* As[2] = {smem, smem + (BM + BN) * KB};
  char* Bs[2] = {smem + BM * KB, smem + (BM + BN) * KB + BM * KB};
  const int nk = (int)((kend - kbeg + KE - 1) / KE);
  if (nk <= 0) return;
  StageRegs<BM> ra;
  StageRegs<BN> rb;
  stage_load<T, BM>(ra, rowA, kbeg, kend, tid);
  stage_load<T, BN>(rb, rowB, kbeg, kend, tid);
  stage_store<BM>(As[0], ra, tid);
  stage_store<BN>(Bs[0], rb, tid);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1 < nk);
    if (more) {
      stage_load<T, BM>(ra, rowA, kbeg + (long)(kt + 1) * KE, kend, tid);
      stage_load<T, BN>(rb, rowB, kbeg + (long)(kt + 1) * KE, kend, tid);
    }
    tile_mma<T, MI, NI>(As[cur], Bs[cur], arow0, brow, acc, lane);
    if (more) {
      stage_store<BM>(As[cur ^ 1], ra, tid);
      stage_store<BN>(Bs[cur ^ 1], rb, tid);
    }
    __syncthreads();
    cur ^= 1;
  }
}

