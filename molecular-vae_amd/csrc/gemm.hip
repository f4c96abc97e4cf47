// NT GEMM on MFMA for gfx950:  C[M,N] = act(A[M,K] . B[N,K]^T + bias[N]),  fp32 accumulate.
// bf16 operands -> v_mfma_f32_16x16x32_bf16, f32 operands -> v_mfma_f32_16x16x4_f32 (exact f32).
// Tiles 128x128 or 64x64 per 256-thread workgroup, split-K with a deterministic fp32 slab reduction when the
// output alone cannot fill 256 CUs.  Blocks are remapped so that consecutive tiles of one XCD share A rows.
#include "tile_pipe.hpp"
#include "kernels.hpp"
#include <stdlib.h>

struct GemmArgs {
  const void* A; const void* B; void* C; const float* bias; float* partial;
  long lda, ldb, ldc;
  int M, N, K;
  int c_dtype, act, accumulate;
  int tiles_m, tiles_n, splits;
  long kper;   // K elements per split (multiple of the K-step)
  // Row groups ("sliding windows"): when group > 0, operand row r starts at element (r / group) * gstride + (r % group) * ld
  // instead of r * ld -- rows of one group may OVERLAP (ld smaller than the row length), which is how a channels-last Conv1d
  // reads its im2col matrix straight out of the activation tensor.  a_total / b_total: elements in the operand buffer (bounds).
  int a_group, b_group;
  long a_gstride, b_gstride, a_total, b_total;
  // column sums of A (out[m] = sum_k A[k][m]) produced alongside C; per-split partials when splits > 1 (colsum_nparts of them: the bf16
  // 256 x 256 kernel writes two K-halves per split, the f32 TN kernel one).  f32 TN kernel: the sums come from a virtual column of ONES
  // appended to B at index ones_col = N (-1: none) -- C[m][N] = sum_k A[k][m] . 1 falls out of the same MFMAs, in the slack of the last tile
  float* colsum_out; float* colsum_partial; int colsum_acc; int colsum_nparts; int ones_col;
};
__device__ __forceinline__ long grow(int r, int group, long gstride, long ld) {
  return group > 0 ? (long)(r / group) * gstride + (long)(r % group) * ld : (long)r * ld;
}

__device__ __forceinline__ void store_out(void* C, long off, int c_dtype, float v, int accumulate) {
  if (c_dtype == MVAE_F32) {
    float* p = reinterpret_cast<float*>(C) + off;
    *p = accumulate ? (*p + v) : v;
  } else {
    reinterpret_cast<bf16_t*>(C)[off].x = f2bf(v);
  }
}

// XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs, so give each XCD a contiguous
// range of tiles (neighbouring tiles share an A row panel / B column panel in that XCD's L2).  Bijective for any count.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
  const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// X3 (fp32 operands, LDS-direct loop only): products on the bf16 MFMA as hi.hi + hi.lo + lo.hi, see tile_pipe.hpp split_x3
template <typename T, int BM, int BN, bool PIPE, bool X3 = false>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 16, NI = WN / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int ntile = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntile);
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int split = blockIdx.y;
  const long kbeg = (long)split * p.kper;
  const long kend = (kbeg + p.kper < (long)p.K) ? (kbeg + p.kper) : (long)p.K;

  const T* A = reinterpret_cast<const T*>(p.A);
  const T* B = reinterpret_cast<const T*>(p.B);
  auto rowA = [&](int r) -> const T* { const int gm = m0 + r; return gm < p.M ? A + grow(gm, p.a_group, p.a_gstride, p.lda) : nullptr; };
  auto rowB = [&](int r) -> const T* { const int gn = n0 + r; return gn < p.N ? B + (long)gn * p.ldb : nullptr; };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int brow[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) brow[j] = wn * WN + j * 16;

  if constexpr (PIPE) {
    constexpr int NBUF = 4;
    PipeSeg<BM, BN> s0, s1;
    const uint32_t sz = (uint32_t)sizeof(T);
    auto offA = [&](int r) -> uint32_t { const int gm = m0 + r; return gm < p.M ? (uint32_t)grow(gm, p.a_group, p.a_gstride, p.lda) * sz : PIPE_OOB; };
    auto offB = [&](int r) -> uint32_t { const int gn = n0 + r; return gn < p.N ? (uint32_t)gn * (uint32_t)p.ldb * sz : PIPE_OOB; };
    const uint32_t bytesA = (uint32_t)(p.a_total * sz - kbeg * sz), bytesB = (uint32_t)((long)p.N * p.ldb * sz - kbeg * sz);
    pipe_seg_init<T, BM, BN>(s0, A + kbeg, bytesA, B + kbeg, bytesB, offA, offB, (int)(kend - kbeg), tid);
    s1 = s0; s1.nk = 0;
    tile_gemm_pipe<T, BM, BN, MI, NI, NBUF, NI, 0, X3>(smem, s0, s1, wm * WM, wn * WN, acc, tid);
  } else {
    tile_gemm_segment<T, BM, BN, MI, NI>(smem, rowA, rowB, kbeg, kend, wm * WM, brow, acc, tid);
  }

  const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = n0 + wn * WN + j * 16 + lr;
      if (col >= p.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * WM + i * 16 + lq * 4 + r;
        if (row >= p.M) continue;
        float v = acc[i][j][r];
        if (p.splits > 1) {
          p.partial[((long)split * p.M + row) * p.N + col] = v;
        } else {
          if (p.bias) v += p.bias[col];
          if (p.act == MVAE_ACT_SELU) v = selu_f(v); else if (p.act == MVAE_ACT_RELU) v = fmaxf(v, 0.f);
          store_out(p.C, (long)row * p.ldc + col, p.c_dtype, v, p.accumulate);
        }
      }
    }
}

__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmArgs p) {
  const long n = (long)p.M * p.N;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int row = (int)(i / p.N), col = (int)(i - (long)row * p.N);
    float v = 0.f;
    for (int s = 0; s < p.splits; ++s) v += p.partial[(long)s * n + i];   // fixed order: deterministic
    if (p.bias) v += p.bias[col];
    if (p.act == MVAE_ACT_SELU) v = selu_f(v); else if (p.act == MVAE_ACT_RELU) v = fmaxf(v, 0.f);
    store_out(p.C, (long)row * p.ldc + col, p.c_dtype, v, p.accumulate);
  }
  if (p.colsum_out && p.colsum_partial) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.M; i += (long)gridDim.x * 256) {
      float v = 0.f;
      for (int s = 0; s < p.colsum_nparts; ++s) v += p.colsum_partial[(long)s * p.M + i];   // per-split (x K-half) partials, fixed order
      p.colsum_out[i] = p.colsum_acc ? p.colsum_out[i] + v : v;
    }
  }
}

// C[M,N] = A^T . B with A [K][lda], B [K][ldb] (bf16, K-major): the weight-gradient contraction (see tile_pipe.hpp, TN form).
template <int NBUF, int MI, bool WS>
__global__ __launch_bounds__(WS ? 512 : 256) void gemm_tn_bf16_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 32 * MI;
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, wm = wave >> 1, wn = wave & 1;
  const int ntile = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntile);
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * 128;
  const int split = blockIdx.y;
  const long kbeg = (long)split * p.kper;
  const long kend = (kbeg + p.kper < (long)p.K) ? (kbeg + p.kper) : (long)p.K;
  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  PipeSegTN<MI> s;
  pipe_seg_tn_init<MI>(s, reinterpret_cast<const bf16_t*>(p.A) + kbeg * p.lda, p.lda, m0, reinterpret_cast<const bf16_t*>(p.B) + kbeg * p.ldb,
                       p.ldb, n0, (int)(kend - kbeg), tid & 255);
  if constexpr (WS) {
    tile_gemm_ws_tn<NBUF, MI>(smem, s, wm, wn, acc, tid);
    if (tid >= 256) return;                       // loader waves own no accumulators
  } else {
    tile_gemm_pipe_tn<NBUF, MI>(smem, s, wm, wn, acc, tid);
  }
  const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wn * 64 + j * 16 + lr;
      if (col >= p.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * (16 * MI) + i * 16 + lq * 4 + r;
        if (row >= p.M) continue;
        float v = acc[i][j][r];
        if (p.splits > 1) {
          p.partial[((long)split * p.M + row) * p.N + col] = v;
        } else {
          if (p.bias) v += p.bias[col];
          if (p.act == MVAE_ACT_SELU) v = selu_f(v); else if (p.act == MVAE_ACT_RELU) v = fmaxf(v, 0.f);
          store_out(p.C, (long)row * p.ldc + col, p.c_dtype, v, p.accumulate);
        }
      }
    }
}

// 256 x 256 tile, 8 waves (tile_gemm_tn_256): used when it still fills the chip through split-K
template <bool COLSUM>
__global__ __launch_bounds__(512) void gemm_tn_bf16_256_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
  const int ntile = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntile);
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const int split = blockIdx.y;
  const long kbeg = (long)split * p.kper;
  const long kend = (kbeg + p.kper < (long)p.K) ? (kbeg + p.kper) : (long)p.K;
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  PipeSegTN2 s;
  pipe_seg_tn2_init(s, reinterpret_cast<const bf16_t*>(p.A) + kbeg * p.lda, p.lda, m0, reinterpret_cast<const bf16_t*>(p.B) + kbeg * p.ldb, p.ldb, n0,
                    (int)(kend - kbeg), tid);
  f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};
  // column sums of A: the 16 (tn, wn) pairs of a row panel share the 8 fragments x 2 K-halves (launcher guarantees tiles_n == 4)
  const int cs_pair = tn * 4 + wn, cs_mi = cs_pair & 7, cs_half = (COLSUM && p.colsum_out != nullptr) ? (cs_pair >> 3) : -1;
  tile_gemm_tn_256<COLSUM>(smem, s, wm, wn, acc, cs_mi, cs_half, accb, tid);
  const int lr = lane & 15, lq = lane >> 4;
  if (COLSUM && cs_half >= 0 && lr == 0) {        // accb[r] (any n column) = sum over this wave's K-half of tile row 16 cs_mi + 4 q + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wm * 128 + cs_mi * 16 + lq * 4 + r;
      if (row < p.M) p.colsum_partial[((long)split * 2 + cs_half) * p.M + row] = accb[r];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wn * 64 + j * 16 + lr;
      if (col >= p.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 128 + i * 16 + lq * 4 + r;
        if (row >= p.M) continue;
        float v = acc[i][j][r];
        if (p.splits > 1) {
          p.partial[((long)split * p.M + row) * p.N + col] = v;
        } else {
          if (p.bias) v += p.bias[col];
          if (p.act == MVAE_ACT_SELU) v = selu_f(v); else if (p.act == MVAE_ACT_RELU) v = fmaxf(v, 0.f);
          store_out(p.C, (long)row * p.ldc + col, p.c_dtype, v, p.accumulate);
        }
      }
    }
}

// ---- grouped form: several weight-gradient contractions of one shape class in ONE launch, each 256 x 256 tile accumulated in registers
// over its group's FULL K -- no split-K slabs, no reduce launch.  The 7 dW_ih / dW_hh GEMMs of the decoder stack are 7 x 64 tiles.
struct TnGroup {
  const void* A; const void* B; float* C; float* cs_partial;      // cs_partial: [2][M] K-half column-sum partials (nullptr: no column sums)
  long lda, ldb, ldc;
  int M, N, K, accumulate, tiles_m, tiles_n, tile0;               // tile0: first tile index of this group in the launch
};
struct TnGroupArgs { TnGroup g[MVAE_MAX_LAYERS * 2]; int ngroups, ntiles; };

template <bool COLSUM>
__global__ __launch_bounds__(512) void gemm_tn_bf16_256_grouped_kernel(TnGroupArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
  // One tile per workgroup -- or, when the launch was given fewer workgroups than tiles (a launch that must leave compute units to the kernels
  // of another stream), a loop over tiles bt, bt + gridDim.x, ...
#pragma unroll 1
  for (int bt = blockIdx.x; bt < p.ntiles; bt += gridDim.x) {
  if (bt != (int)blockIdx.x) __syncthreads();              // the previous tile's last LDS stage has been read by every wave
  const int gt = (gridDim.x == (unsigned)p.ntiles) ? xcd_remap(bt, p.ntiles) : bt;
  int gi = 0;
#pragma unroll 1
  for (int i = 1; i < p.ngroups; ++i) if (gt >= p.g[i].tile0) gi = i;
  const TnGroup& q = p.g[gi];
  const int tile = gt - q.tile0;
  const int tm = tile / q.tiles_n, tn = tile - tm * q.tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  PipeSegTN2 s;
  pipe_seg_tn2_init(s, q.A, q.lda, m0, q.B, q.ldb, n0, q.K, tid);
  f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};
  const int cs_pair = tn * 4 + wn, cs_mi = cs_pair & 7, cs_half = (COLSUM && q.cs_partial != nullptr) ? (cs_pair >> 3) : -1;
  tile_gemm_tn_256<COLSUM>(smem, s, wm, wn, acc, cs_mi, cs_half, accb, tid);
  const int lr = lane & 15, lq = lane >> 4;
  if (COLSUM && cs_half >= 0 && lr == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wm * 128 + cs_mi * 16 + lq * 4 + r;
      if (row < q.M) q.cs_partial[(long)cs_half * q.M + row] = accb[r];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wn * 64 + j * 16 + lr;
      if (col >= q.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 128 + i * 16 + lq * 4 + r;
        if (row >= q.M) continue;
        float* c = q.C + (long)row * q.ldc + col;
        *c = q.accumulate ? (*c + acc[i][j][r]) : acc[i][j][r];
      }
    }
  }
}
// out[m] (+)= partial[0][m] + partial[1][m], fixed order
struct TnColsumFinish { const float* partial[MVAE_MAX_LAYERS * 2]; float* out[MVAE_MAX_LAYERS * 2]; int M[MVAE_MAX_LAYERS * 2], acc[MVAE_MAX_LAYERS * 2]; int n; };
__global__ __launch_bounds__(256) void gemm_tn_colsum_finish_kernel(TnColsumFinish f) {
  const int gi = blockIdx.y;
  if (gi >= f.n) return;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < f.M[gi]; i += gridDim.x * 256) {
    const float v = f.partial[gi][i] + f.partial[gi][f.M[gi] + i];
    f.out[gi][i] = f.acc[gi] ? f.out[gi][i] + v : v;
  }
}

namespace {
struct Plan { int bm, bn, tiles_m, tiles_n, splits; long kper; };

Plan make_plan(int M, int N, int K, int dtype) {
  Plan pl;
  const int ke = (dtype == MVAE_BF16) ? 64 : 32;
  // 128 x 128 tiles when they fill the chip AND their last round of workgroups is not mostly empty (288 tiles on 256 CUs = 2 rounds at
  // 56 %); otherwise 64 x 64 (four times the tiles: finer tail, and the f32 MFMA rate leaves room for the extra operand traffic)
  const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128), t64 = (long)((M + 63) / 64) * ((N + 63) / 64);
  auto eff = [](long t) { return (double)t / (double)(((t + 255) / 256) * 256); };
  pl.bm = (t128 >= 192 && eff(t128) >= 0.8 * eff(t64)) ? 128 : 64;
  pl.bn = pl.bm;
  // narrow outputs in f32 (the conv stack: 9 / 10 output channels padded to 32): a 128 x 32 tile, so that the slow f32 MFMAs are not
  // spent on the 75 % (128-wide tile) or 50 % (64-wide) of columns that lie outside the matrix
  if (dtype == MVAE_F32 && N <= 32 && M >= 128) { pl.bm = 128; pl.bn = 32; }
  pl.tiles_m = (M + pl.bm - 1) / pl.bm;
  pl.tiles_n = (N + pl.bn - 1) / pl.bn;
  const long tiles = (long)pl.tiles_m * pl.tiles_n;
  long ksteps = (K + ke - 1) / ke;
  if (ksteps < 1) ksteps = 1;                 // K == 0: one empty pass, the epilogue still writes bias / zeros
  int splits = 1;
  if (tiles < 128 && ksteps >= 16) {
    splits = (int)((256 + tiles - 1) / tiles);
    const long maxs = ksteps / 8;            // at least 8 K-steps per split
    if (splits > maxs) splits = (int)maxs;
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
  }
  long per = (ksteps + splits - 1) / splits;
  splits = (int)((ksteps + per - 1) / per);
  pl.splits = splits;
  pl.kper = per * ke;
  return pl;
}
}  // namespace

size_t gemm_nt_workspace_bytes(int M, int N, int K, int dtype) {
  if (dtype == MVAE_F32X3) dtype = MVAE_F32;
  Plan pl = make_plan(M, N, K, dtype);
  return pl.splits > 1 ? (size_t)pl.splits * M * N * sizeof(float) : 0;
}

int launch_gemm_nt(int dtype, int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                   int c_dtype, const float* bias, int act, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  return launch_gemm_nt_grouped(dtype, M, N, K, A, lda, 0, 0, (long)M * lda, B, ldb, C, ldc, c_dtype, bias, act, accumulate, ws, ws_bytes, st);
}

int launch_gemm_nt_grouped(int dtype, int M, int N, int K, const void* A, long lda, int a_group, long a_gstride, long a_total,
                           const void* B, long ldb, void* C, long ldc, int c_dtype, const float* bias, int act, int accumulate,
                           void* ws, size_t ws_bytes, hipStream_t st) {
  if (M <= 0 || N <= 0) return MVAE_OK;
  if (K < 0 || !A || !B || !C) return MVAE_ERR_INVALID;
  const bool x3 = dtype == MVAE_F32X3;           // fp32 operands, 3 x bf16 products (only the LDS-direct loop has the form; otherwise exact fp32)
  if (x3) dtype = MVAE_F32;
  if (dtype != MVAE_F32 && dtype != MVAE_BF16) return MVAE_ERR_INVALID;
  if (c_dtype != MVAE_F32 && c_dtype != MVAE_BF16) return MVAE_ERR_INVALID;
  if (accumulate && c_dtype != MVAE_F32) return MVAE_ERR_INVALID;
  Plan pl = make_plan(M, N, K, dtype);
  GemmArgs p;
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.partial = nullptr;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.c_dtype = c_dtype; p.act = act; p.accumulate = accumulate;
  p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n; p.splits = pl.splits; p.kper = pl.kper;
  p.a_group = a_group; p.a_gstride = a_gstride; p.a_total = a_total; p.b_group = 0; p.b_gstride = 0; p.b_total = (long)N * ldb;
  p.colsum_out = nullptr; p.colsum_partial = nullptr; p.colsum_acc = 0; p.colsum_nparts = 0; p.ones_col = -1;
  if (pl.splits > 1) {
    const size_t need = (size_t)pl.splits * M * N * sizeof(float);
    if (!ws || ws_bytes < need) return MVAE_ERR_WORKSPACE;
    p.partial = reinterpret_cast<float*>(ws);
  }
  dim3 grid(pl.tiles_m * pl.tiles_n, pl.splits), block(256);
  const int sz = (dtype == MVAE_BF16) ? 2 : 4, ke = KB / sz;
  // deep-pipelined LDS-direct path: whole K-steps only, 16-byte aligned rows, operands < 2 GiB
  const bool pipe = (K % ke == 0) && (pl.kper % ke == 0) && (lda % (16 / sz) == 0) && (ldb % (16 / sz) == 0) &&
                    ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) % 16 == 0) &&
                    (a_total * sz < (1L << 31)) && ((long)N * ldb * sz < (1L << 31)) && K >= 4 * ke &&
                    (a_group == 0 || a_gstride % (16 / sz) == 0);
  const size_t lds = (size_t)(pipe ? 4 : 2) * (pl.bm + pl.bn) * KB;
#define MVAE_GEMM_LAUNCH2(TT_, BM_, BN_, PIPE_)                                                       \
  do {                                                                                                \
    auto kern = gemm_nt_kernel<TT_, BM_, BN_, PIPE_>;                                                 \
    static bool attr_set = false;                                                                     \
    if (!attr_set) { MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; } \
    hipLaunchKernelGGL(kern, grid, block, lds, st, p);                                                \
  } while (0)
#define MVAE_GEMM_LAUNCH(TT_, BM_, PIPE_) MVAE_GEMM_LAUNCH2(TT_, BM_, BM_, PIPE_)
#define MVAE_GEMM_LAUNCH3(BM_)                                                                        \
  do {                                                                                                \
    auto kern = gemm_nt_kernel<float, BM_, BM_, true, true>;                                          \
    static bool attr_set = false;                                                                     \
    if (!attr_set) { MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; } \
    hipLaunchKernelGGL(kern, grid, block, lds, st, p);                                                \
  } while (0)
  if (pl.bn == 32) {
    if (pipe) MVAE_GEMM_LAUNCH2(float, 128, 32, true); else MVAE_GEMM_LAUNCH2(float, 128, 32, false);
  } else if (dtype == MVAE_BF16) {
    if (pl.bm == 128) { if (pipe) MVAE_GEMM_LAUNCH(bf16_t, 128, true); else MVAE_GEMM_LAUNCH(bf16_t, 128, false); }
    else { if (pipe) MVAE_GEMM_LAUNCH(bf16_t, 64, true); else MVAE_GEMM_LAUNCH(bf16_t, 64, false); }
  } else if (x3 && pipe) {
    if (pl.bm == 128) MVAE_GEMM_LAUNCH3(128); else MVAE_GEMM_LAUNCH3(64);
  } else {
    if (pl.bm == 128) { if (pipe) MVAE_GEMM_LAUNCH(float, 128, true); else MVAE_GEMM_LAUNCH(float, 128, false); }
    else { if (pipe) MVAE_GEMM_LAUNCH(float, 64, true); else MVAE_GEMM_LAUNCH(float, 64, false); }
  }
#undef MVAE_GEMM_LAUNCH
#undef MVAE_GEMM_LAUNCH2
#undef MVAE_GEMM_LAUNCH3
  MVAE_CHECK_HIP(hipGetLastError());
  if (pl.splits > 1) {
    long n = (long)M * N;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), block, 0, st, p);
    MVAE_CHECK_HIP(hipGetLastError());
  }
  return MVAE_OK;
}

// ---- TN (bf16): plan = 128x128 tiles, K-steps of 64 rows, split-K when the output alone cannot fill the chip
namespace {
Plan make_plan_tn(int M, int N, int K) {
  Plan pl;
  // 256 x 128 tiles (half the operand bytes per output of 128 x 128) whenever they still give >= 128 tiles: split-K then fills the chip
  const char* fe = mvae_knob("MVAE_TN_BM");
  const int force_bm = fe ? atoi(fe) : 0;
  pl.bm = ((long)((M + 255) / 256) * ((N + 127) / 128) >= 128) ? 256 : 128;
  // 256 x 256 (pl.bm = 512 as its tag): another third less operand traffic per FLOP; needs whole tiles' worth of work and enough K to split
  if ((long)((M + 255) / 256) * ((N + 255) / 256) >= 32 && N % 256 == 0 && K >= 64 * 64) pl.bm = 512;
  if (force_bm == 128 || force_bm == 256 || force_bm == 512) pl.bm = force_bm;
  const int bm_rows = pl.bm == 512 ? 256 : pl.bm, bn_cols = pl.bm == 512 ? 256 : 128;
  pl.tiles_m = (M + bm_rows - 1) / bm_rows; pl.tiles_n = (N + bn_cols - 1) / bn_cols;
  const long tiles = (long)pl.tiles_m * pl.tiles_n;
  long ksteps = (K + 63) / 64;
  if (ksteps < 1) ksteps = 1;
  int splits = 1;
  if (tiles < 192 && ksteps >= 16) {
    splits = (int)((256 + tiles - 1) / tiles);
    const long maxs = ksteps / 8;
    if (splits > maxs) splits = (int)maxs;
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
  }
  const long per = (ksteps + splits - 1) / splits;
  pl.splits = (int)((ksteps + per - 1) / per);
  pl.kper = per * 64;
  return pl;
}
}  // namespace

size_t gemm_tn_workspace_bytes(int M, int N, int K) {
  Plan pl = make_plan_tn(M, N, K);
  return pl.splits > 1 ? (size_t)pl.splits * M * N * sizeof(float) : 0;
}

int launch_gemm_tn_bf16(int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc, int c_dtype,
                        const float* bias, int act, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  return launch_gemm_tn_bf16_colsum(M, N, K, A, lda, B, ldb, C, ldc, c_dtype, bias, act, accumulate, nullptr, 0, ws, ws_bytes, st);
}

// colsum_out (optional, fp32 [M]): out[m] (+)= sum_k A[k][m].  Only the 256 x 256 tile produces it: MVAE_ERR_UNSUPPORTED when the plan
// for this shape is another tile (the caller then uses mvae_colsum_t).
bool gemm_tn_colsum_supported(int M, int N, int K) {
  Plan pl = make_plan_tn(M, N, K);
  return pl.bm == 512 && pl.tiles_n == 4 && pl.splits > 1;       // 16 (tile column, wave column) pairs per row panel; partials summed by the reduction
}
size_t gemm_tn_colsum_workspace_bytes(int M, int N, int K) {
  Plan pl = make_plan_tn(M, N, K);
  return (size_t)pl.splits * M * (N + 2) * sizeof(float);
}
int launch_gemm_tn_bf16_colsum(int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc, int c_dtype,
                               const float* bias, int act, int accumulate, float* colsum_out, int colsum_acc, void* ws, size_t ws_bytes,
                               hipStream_t st) {
  if (M <= 0 || N <= 0) return MVAE_OK;
  if (K < 0 || !A || !B || !C) return MVAE_ERR_INVALID;
  if (c_dtype != MVAE_F32 && c_dtype != MVAE_BF16) return MVAE_ERR_INVALID;
  if (accumulate && c_dtype != MVAE_F32) return MVAE_ERR_INVALID;
  if ((lda % 8) || (ldb % 8) || ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15)) return MVAE_ERR_INVALID;
  if ((long)K * lda * 2 >= (1L << 31) || (long)K * ldb * 2 >= (1L << 31)) {
    // operands beyond the 2 GiB a buffer descriptor addresses (BASELINE configs[4]: K = T * B = 524288 rows): equal K-chunks, each a
    // launch of its own that accumulates onto C (and onto the column sums)
    const long ldm = lda > ldb ? lda : ldb;
    const long kmax = (((1L << 31) - 1) / (2 * ldm)) / 4096 * 4096;
    if (kmax < 4096 || bias || act != MVAE_ACT_NONE || c_dtype != MVAE_F32) return MVAE_ERR_UNSUPPORTED;
    const long nch = (K + kmax - 1) / kmax;
    const long kc = ((K + nch - 1) / nch + 63) / 64 * 64;
    for (long k0 = 0; k0 < K; k0 += kc) {
      const int kk = (int)(K - k0 < kc ? K - k0 : kc);
      const int rc = launch_gemm_tn_bf16_colsum(M, N, kk, reinterpret_cast<const char*>(A) + k0 * lda * 2, lda, reinterpret_cast<const char*>(B) + k0 * ldb * 2, ldb,
                                                C, ldc, c_dtype, nullptr, MVAE_ACT_NONE, (k0 > 0) ? 1 : accumulate, colsum_out, (k0 > 0) ? 1 : colsum_acc,
                                                ws, ws_bytes, st);
      if (rc != MVAE_OK) return rc;
    }
    return MVAE_OK;
  }
  Plan pl = make_plan_tn(M, N, K);
  GemmArgs p;
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.partial = nullptr;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.c_dtype = c_dtype; p.act = act; p.accumulate = accumulate;
  p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n; p.splits = pl.splits; p.kper = pl.kper;
  p.a_group = 0; p.b_group = 0; p.a_gstride = 0; p.b_gstride = 0; p.a_total = (long)K * lda; p.b_total = (long)K * ldb;
  p.colsum_out = nullptr; p.colsum_partial = nullptr; p.colsum_acc = colsum_acc; p.colsum_nparts = 2 * pl.splits; p.ones_col = -1;
  if (colsum_out && !(pl.bm == 512 && pl.tiles_n == 4 && pl.splits > 1)) return MVAE_ERR_UNSUPPORTED;
  if (pl.splits > 1) {
    const size_t need = (size_t)pl.splits * M * (N + (colsum_out ? 2 : 0)) * sizeof(float);
    if (!ws || ws_bytes < need) return MVAE_ERR_WORKSPACE;
    p.partial = reinterpret_cast<float*>(ws);
    if (colsum_out) p.colsum_partial = p.partial + (size_t)pl.splits * M * N;
  }
  p.colsum_out = colsum_out;
  dim3 grid(pl.tiles_m * pl.tiles_n, pl.splits);
  static bool attr_set = false;
  if (!attr_set) {
    MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_bf16_kernel<4, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_bf16_kernel<3, 8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_bf16_kernel<4, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_bf16_kernel<3, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_bf16_256_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_bf16_256_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const char* we = mvae_knob("MVAE_TN_WS");
  const bool wspec = we ? atoi(we) != 0 : true;      // loader / consumer wave specialisation (512-thread workgroups)
  if (pl.bm == 512) {
    if (colsum_out) hipLaunchKernelGGL(gemm_tn_bf16_256_kernel<true>, grid, dim3(512), 2 * 65536, st, p);
    else hipLaunchKernelGGL(gemm_tn_bf16_256_kernel<false>, grid, dim3(512), 2 * 65536, st, p);
  }
  else if (wspec) {
    if (pl.bm == 256) hipLaunchKernelGGL((gemm_tn_bf16_kernel<3, 8, true>), grid, dim3(512), 3 * (32768 + 16384), st, p);
    else {
      const char* ne = mvae_knob("MVAE_NBUF_TN");
      const int nb = ne ? atoi(ne) : 4;
      if (nb == 3) hipLaunchKernelGGL((gemm_tn_bf16_kernel<3, 4, true>), grid, dim3(512), 3 * 32768, st, p);
      else if (nb == 5) {
        static bool a5 = false;
        if (!a5) { MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_bf16_kernel<5, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); a5 = true; }
        hipLaunchKernelGGL((gemm_tn_bf16_kernel<5, 4, true>), grid, dim3(512), 5 * 32768, st, p);
      } else hipLaunchKernelGGL((gemm_tn_bf16_kernel<4, 4, true>), grid, dim3(512), 4 * 32768, st, p);
    }
  } else {
    if (pl.bm == 256) hipLaunchKernelGGL((gemm_tn_bf16_kernel<3, 8, false>), grid, dim3(256), 3 * (32768 + 16384), st, p);   // 144 KiB ring
    else hipLaunchKernelGGL((gemm_tn_bf16_kernel<4, 4, false>), grid, dim3(256), 4 * 32768, st, p);
  }
  MVAE_CHECK_HIP(hipGetLastError());
  if (pl.splits > 1) {
    long n = (long)M * N;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
    MVAE_CHECK_HIP(hipGetLastError());
  }
  return MVAE_OK;
}


// =====================================================================================================================
// TN GEMM, f32 (exact, v_mfma_f32_16x16x4_f32):  C[M,N] = sum_r A[r][m] * B[r][n], both operands R-major (row = r), optionally
// with row groups (GemmArgs).  This is the Conv1d weight gradient dW = dz^T . windows(x): the reduction runs over the
// B * Wout output positions, A rows are dz[b, w, :], B rows are the overlapping windows x[b, w : w + k, :] -- neither the
// im2col matrix nor a transpose of it is ever materialised.
// 64 x 64 tile per 256-thread workgroup (4 waves, 2 x 2, wave tile 32 x 32), 32 r-rows per stage, register-staged double
// buffer.  LDS image [r][64 + 16] floats: the MFMA fragment of lane l is element [r0 + (l >> 4)][m0 + (l & 15)], one
// ds_read_b32; the +16 pad puts the four r-rows of a wave read on disjoint banks.  Split over r with fp32 slabs.
// =====================================================================================================================
constexpr int TNF_BM = 64, TNF_KS = 32, TNF_LD = 80;

// X3: the products go to the bf16 MFMA as hi.hi + hi.lo + lo.hi (split_x3 in tile_pipe.hpp).  The 16x16x32 fragment of lane l is eight k-values
// of column (l & 15) for k-slice (l >> 4); WHICH eight is free as long as A and B agree, so slice f takes rows f, f + 4, ..., f + 28 of the
// stage -- exactly the eight values the fp32 form already reads for its eight 16x16x4 MFMAs.  12 bf16 MFMAs per stage instead of 32 fp32 ones.
// CS: instantiation that can produce the column sums of A through a virtual ones column of B (p.ones_col); the plain instantiation carries no
// trace of it -- this loop has no VALU slack (four compares and selects per staged chunk cost the f32 decoder's weight gradients +28 %).
template <bool X3, bool CS>
__device__ __forceinline__ void gemm_tn_f32_body(const GemmArgs& p, int tile, int split, float (&As)[2][TNF_KS * TNF_LD], float (&Bs)[2][TNF_KS * TNF_LD]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * TNF_BM, n0 = tn * TNF_BM;
  const long rbeg = (long)split * p.kper;
  const long rend = (rbeg + p.kper < (long)p.K) ? (rbeg + p.kper) : (long)p.K;
  const float* A = reinterpret_cast<const float*>(p.A);
  const float* B = reinterpret_cast<const float*>(p.B);
  // each thread stages 2 float4 of A and 2 of B per stage: r-row = id >> 4, column chunk = id & 15 (id = tid, tid + 256)
  const int c4 = (tid & 15) * 4;
  const bool a_vec = (m0 + c4 + 3 < p.M), b_vec = (n0 + c4 + 3 < p.N);     // whole chunk inside the matrix (host checked alignment)
  const bool ones_here = CS && p.ones_col >= n0 && p.ones_col < n0 + TNF_BM;
  float4 ra[2], rb[2];
  // per-thread row offsets, kept incrementally: a runtime integer division per row and stage (grouped operands) costs as many issue
  // cycles as the stage's MFMAs
  long oa[2], ob[2]; int ia[2], ib[2];                    // element offset of the row start; position of the row inside its group
  auto row_init = [&](long r, int group, long gstride, long ld, long& off, int& in_g) {
    if (group > 0) { const long qg = r / group; in_g = (int)(r - qg * group); off = qg * gstride + (long)in_g * ld; }
    else { in_g = 0; off = r * ld; }
  };
  auto row_step = [&](int group, long gstride, long ld, long& off, int& in_g) {       // advance by TNF_KS rows
    if (group > 0) {
      in_g += TNF_KS; off += (long)TNF_KS * ld;
      while (in_g >= group) { in_g -= group; off += gstride - (long)group * ld; }
    } else off += (long)TNF_KS * ld;
  };
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const long r = rbeg + (tid >> 4) + i * 16;
    row_init(r, p.a_group, p.a_gstride, p.lda, oa[i], ia[i]);
    row_init(r, p.b_group, p.b_gstride, p.ldb, ob[i], ib[i]);
  }
  auto load_stage = [&](long r0) {                        // stages are loaded in order: the row state advances by one stage per call
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long r = r0 + (tid >> 4) + i * 16;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
      if (r < rend) {
        const float* pa = A + oa[i] + m0 + c4;
        const float* pb = B + ob[i] + n0 + c4;
        if (a_vec) va = *reinterpret_cast<const float4*>(pa);
        else { if (m0 + c4 < p.M) va.x = pa[0]; if (m0 + c4 + 1 < p.M) va.y = pa[1]; if (m0 + c4 + 2 < p.M) va.z = pa[2]; }
        if (b_vec) vb = *reinterpret_cast<const float4*>(pb);
        else { if (n0 + c4 < p.N) vb.x = pb[0]; if (n0 + c4 + 1 < p.N) vb.y = pb[1]; if (n0 + c4 + 2 < p.N) vb.z = pb[2]; }
        if constexpr (CS) {
          if (ones_here) {                                    // workgroup-uniform: only the tile column that holds the virtual ones column
            const int od = p.ones_col - (n0 + c4);
            if (od == 0) vb.x = 1.f; else if (od == 1) vb.y = 1.f; else if (od == 2) vb.z = 1.f; else if (od == 3) vb.w = 1.f;
          }
        }
      }
      ra[i] = va; rb[i] = vb;
      row_step(p.a_group, p.a_gstride, p.lda, oa[i], ia[i]);
      row_step(p.b_group, p.b_gstride, p.ldb, ob[i], ib[i]);
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (tid >> 4) + i * 16;
      *reinterpret_cast<float4*>(&As[buf][r * TNF_LD + c4]) = ra[i];
      *reinterpret_cast<float4*>(&Bs[buf][r * TNF_LD + c4]) = rb[i];
    }
  };
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nst = (int)((rend - rbeg + TNF_KS - 1) / TNF_KS);
  if (nst > 0) {
    load_stage(rbeg);
    store_stage(0);
    __syncthreads();
    const int fr = lane >> 4, fc = lane & 15;
    for (int s = 0; s < nst; ++s) {
      const int cur = s & 1;
      if (s + 1 < nst) load_stage(rbeg + (long)(s + 1) * TNF_KS);
      const float* as = &As[cur][fr * TNF_LD + wm * 32 + fc];
      const float* bs = &Bs[cur][fr * TNF_LD + wn * 32 + fc];
      // all 32 fragment reads of the stage first (one LDS round trip), then 32 MFMAs back to back; left to itself the compiler
      // alternates two reads / wait / four MFMAs and exposes the LDS latency eight times per stage
      float af[TNF_KS / 4][2], bf[TNF_KS / 4][2];
#pragma unroll
      for (int k4 = 0; k4 < TNF_KS / 4; ++k4) {
        af[k4][0] = as[k4 * 4 * TNF_LD]; af[k4][1] = as[k4 * 4 * TNF_LD + 16];
        bf[k4][0] = bs[k4 * 4 * TNF_LD]; bf[k4][1] = bs[k4 * 4 * TNF_LD + 16];
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (X3) {
        static_assert(TNF_KS == 32, "one 16x16x32 MFMA step per stage");
        uint4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const u32x4 a0 = {__builtin_bit_cast(uint32_t, af[0][i]), __builtin_bit_cast(uint32_t, af[1][i]), __builtin_bit_cast(uint32_t, af[2][i]), __builtin_bit_cast(uint32_t, af[3][i])};
          const u32x4 a1 = {__builtin_bit_cast(uint32_t, af[4][i]), __builtin_bit_cast(uint32_t, af[5][i]), __builtin_bit_cast(uint32_t, af[6][i]), __builtin_bit_cast(uint32_t, af[7][i])};
          const u32x4 b0 = {__builtin_bit_cast(uint32_t, bf[0][i]), __builtin_bit_cast(uint32_t, bf[1][i]), __builtin_bit_cast(uint32_t, bf[2][i]), __builtin_bit_cast(uint32_t, bf[3][i])};
          const u32x4 b1 = {__builtin_bit_cast(uint32_t, bf[4][i]), __builtin_bit_cast(uint32_t, bf[5][i]), __builtin_bit_cast(uint32_t, bf[6][i]), __builtin_bit_cast(uint32_t, bf[7][i])};
          split_x3(a0, a1, ah[i], al[i]);
          split_x3(b0, b1, bh[i], bl[i]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            mma16<bf16_t>(al[i], bh[j], acc[i][j]);
            mma16<bf16_t>(ah[i], bl[j], acc[i][j]);
            mma16<bf16_t>(ah[i], bh[j], acc[i][j]);
          }
      } else {
#pragma unroll
      for (int k4 = 0; k4 < TNF_KS / 4; ++k4) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[k4][0], bf[k4][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[k4][0], bf[k4][1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[k4][1], bf[k4][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[k4][1], bf[k4][1], acc[1][1], 0, 0, 0);
      }
      }
      if (s + 1 < nst) store_stage(cur ^ 1);
      __syncthreads();
    }
  }
  const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 32 + j * 16 + lr;
      if (col >= p.N && !(CS && col == p.ones_col)) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 32 + i * 16 + lq * 4 + r;
        if (row >= p.M) continue;
        const float v = acc[i][j][r];
        if (CS && col == p.ones_col) {                        // column sums of A
          if (p.splits > 1) p.colsum_partial[(long)split * p.M + row] = v;
          else p.colsum_out[row] = p.colsum_acc ? p.colsum_out[row] + v : v;
        } else if (p.splits > 1) p.partial[((long)split * p.M + row) * p.N + col] = v;
        else store_out(p.C, (long)row * p.ldc + col, p.c_dtype, v, p.accumulate);
      }
    }
}
template <bool X3, bool CS = false>
__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(GemmArgs p) {
  __shared__ float As[2][TNF_KS * TNF_LD];
  __shared__ float Bs[2][TNF_KS * TNF_LD];
  gemm_tn_f32_body<X3, CS>(p, xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n), blockIdx.y, As, Bs);
}

// ---- several problems of different shapes in ONE launch (mvae_gemm_tn_f32_multi): the argument block carries every problem's GemmArgs (kernel
// arguments: no device-side table to build or copy); block b serves problem g with first[g] <= b < first[g + 1], tile = local % tiles,
// split = local / tiles.  flags[g]: bit 0 = x3, bit 1 = column sums.  One second launch sums every problem's split-K slabs (grid.y = problem).
struct TnF32Multi {
  GemmArgs p[MVAE_TN_F32_MULTI_MAX];
  int first[MVAE_TN_F32_MULTI_MAX + 1];
  int flags[MVAE_TN_F32_MULTI_MAX];
  int n;
};
static_assert(sizeof(TnF32Multi) <= 4000, "kernel argument block");
__global__ __launch_bounds__(256) void gemm_tn_f32_multi_kernel(TnF32Multi m) {
  __shared__ float As[2][TNF_KS * TNF_LD];
  __shared__ float Bs[2][TNF_KS * TNF_LD];
  int g = 0;
  while (g + 1 < m.n && (int)blockIdx.x >= m.first[g + 1]) ++g;          // workgroup-uniform
  const GemmArgs& p = m.p[g];
  const int local = (int)blockIdx.x - m.first[g], tiles = p.tiles_m * p.tiles_n;
  const int split = local / tiles, tile = local - split * tiles;
  switch (m.flags[g]) {
    case 0: gemm_tn_f32_body<false, false>(p, tile, split, As, Bs); break;
    case 1: gemm_tn_f32_body<true, false>(p, tile, split, As, Bs); break;
    case 2: gemm_tn_f32_body<false, true>(p, tile, split, As, Bs); break;
    default: gemm_tn_f32_body<true, true>(p, tile, split, As, Bs); break;
  }
}
__global__ __launch_bounds__(256) void gemm_splitk_reduce_multi_kernel(TnF32Multi m) {
  const GemmArgs& p = m.p[blockIdx.y];
  if (p.splits <= 1) return;
  const long n = (long)p.M * p.N;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int row = (int)(i / p.N), col = (int)(i - (long)row * p.N);
    float v = 0.f;
    for (int s = 0; s < p.splits; ++s) v += p.partial[(long)s * n + i];   // fixed order: deterministic
    store_out(p.C, (long)row * p.ldc + col, p.c_dtype, v, p.accumulate);
  }
  if (p.colsum_out && p.colsum_partial) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.M; i += (long)gridDim.x * 256) {
      float v = 0.f;
      for (int s = 0; s < p.colsum_nparts; ++s) v += p.colsum_partial[(long)s * p.M + i];
      p.colsum_out[i] = p.colsum_acc ? p.colsum_out[i] + v : v;
    }
  }
}

namespace {
// `target`: workgroups this problem should offer (768 = two or three per CU when it has the chip to itself; a problem of a multi launch
// gets its share of ~1536)
Plan make_plan_tn_f32(int M, int N, int R, int target = 768) {
  Plan pl;
  pl.bm = TNF_BM;
  pl.tiles_m = (M + TNF_BM - 1) / TNF_BM; pl.tiles_n = (N + TNF_BM - 1) / TNF_BM;
  const long tiles = (long)pl.tiles_m * pl.tiles_n;
  long steps = (R + TNF_KS - 1) / TNF_KS;
  if (steps < 1) steps = 1;
  int splits = 1;
  if (tiles < (target * 2) / 3 && steps >= 16) {           // two or more workgroups per CU: the f32 MFMA stream of one tile is short on parallelism
    splits = (int)((target + tiles - 1) / tiles);
    const long maxs = steps / 8;
    if (splits > maxs) splits = (int)maxs;
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
  }
  const long per = (steps + splits - 1) / splits;
  pl.splits = (int)((steps + per - 1) / per);
  pl.kper = per * TNF_KS;
  return pl;
}
}  // namespace

size_t gemm_tn_f32_workspace_bytes(int M, int N, int R) {     // sized for the column-sum form too (one more B column, one partial per split)
  Plan pl = make_plan_tn_f32(M, N + 1, R);
  Plan p0 = make_plan_tn_f32(M, N, R);
  const int sp = pl.splits > p0.splits ? pl.splits : p0.splits;
  return sp > 1 ? (size_t)sp * M * (N + 1) * sizeof(float) : 0;
}

// A: rows of M floats, B: rows of N floats; group == 0 -> plain row-major [R][ld].  Row starts must be 16-byte aligned
// (ld, gstride multiples of 4, bases 16-byte aligned) -- checked here.
int launch_gemm_tn_f32(int M, int N, int R, const float* A, long lda, int a_group, long a_gstride, const float* B, long ldb, int b_group,
                       long b_gstride, float* C, long ldc, int accumulate, void* ws, size_t ws_bytes, hipStream_t st, bool x3,
                       float* colsum_out, int colsum_acc) {
  if (M <= 0 || N <= 0) return MVAE_OK;
  if (R < 0 || !A || !B || !C) return MVAE_ERR_INVALID;
  if ((lda | ldb | a_gstride | b_gstride) & 3) return MVAE_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) return MVAE_ERR_INVALID;
  Plan pl = make_plan_tn_f32(M, colsum_out ? N + 1 : N, R);
  GemmArgs p;
  p.A = A; p.B = B; p.C = C; p.bias = nullptr; p.partial = nullptr;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = R;
  p.c_dtype = MVAE_F32; p.act = MVAE_ACT_NONE; p.accumulate = accumulate;
  p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n; p.splits = pl.splits; p.kper = pl.kper;
  p.a_group = a_group; p.a_gstride = a_gstride; p.b_group = b_group; p.b_gstride = b_gstride; p.a_total = 0; p.b_total = 0;
  p.colsum_out = colsum_out; p.colsum_partial = nullptr; p.colsum_acc = colsum_acc; p.colsum_nparts = pl.splits; p.ones_col = colsum_out ? N : -1;
  if (pl.splits > 1) {
    const size_t need = (size_t)pl.splits * M * (N + (colsum_out ? 1 : 0)) * sizeof(float);
    if (!ws || ws_bytes < need) return MVAE_ERR_WORKSPACE;
    p.partial = reinterpret_cast<float*>(ws);
    if (colsum_out) p.colsum_partial = p.partial + (size_t)pl.splits * M * N;
  }
  const dim3 tgrid(pl.tiles_m * pl.tiles_n, pl.splits);
  if (colsum_out) {
    if (x3) hipLaunchKernelGGL((gemm_tn_f32_kernel<true, true>), tgrid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_tn_f32_kernel<false, true>), tgrid, dim3(256), 0, st, p);
  } else {
    if (x3) hipLaunchKernelGGL((gemm_tn_f32_kernel<true, false>), tgrid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_tn_f32_kernel<false, false>), tgrid, dim3(256), 0, st, p);
  }
  MVAE_CHECK_HIP(hipGetLastError());
  if (pl.splits > 1) {
    long n = (long)M * N;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
    MVAE_CHECK_HIP(hipGetLastError());
  }
  return MVAE_OK;
}


// ---- multi launcher (exact-f32 TN problems of different shapes in one launch; see gemm_tn_f32_multi_kernel)
namespace {
int tnf_multi_target(int n) { int t = 1536 / (n > 0 ? n : 1); return t < 96 ? 96 : (t > 768 ? 768 : t); }
Plan tnf_multi_plan(const mvae_gemm_tn_f32_problem& q, int n) { return make_plan_tn_f32(q.M, q.colsum_out ? q.N + 1 : q.N, (int)q.K, tnf_multi_target(n)); }
size_t tnf_multi_need(const mvae_gemm_tn_f32_problem& q, const Plan& pl) {
  return pl.splits > 1 ? (((size_t)pl.splits * q.M * (q.N + (q.colsum_out ? 1 : 0)) * sizeof(float) + 255) & ~(size_t)255) : 0;
}
}  // namespace
extern "C" size_t mvae_gemm_tn_f32_multi_workspace(int n, const mvae_gemm_tn_f32_problem* pr) {
  if (n < 1 || n > MVAE_TN_F32_MULTI_MAX || !pr) return 0;
  size_t b = 0;
  for (int i = 0; i < n; ++i) if (pr[i].M > 0 && pr[i].N > 0 && pr[i].K >= 0 && pr[i].K < (1L << 31)) b += tnf_multi_need(pr[i], tnf_multi_plan(pr[i], n));
  return b;
}
extern "C" int mvae_gemm_tn_f32_multi(int n, const mvae_gemm_tn_f32_problem* pr, void* ws, size_t ws_bytes, void* stream) {
  if (n < 1 || n > MVAE_TN_F32_MULTI_MAX || !pr) return MVAE_ERR_INVALID;
  if (mvae_gemm_tn_f32_multi_workspace(n, pr) > ws_bytes || (mvae_gemm_tn_f32_multi_workspace(n, pr) && !ws)) return MVAE_ERR_WORKSPACE;
  TnF32Multi m;
  m.n = n;
  char* wsp = reinterpret_cast<char*>(ws);
  int blocks = 0, any_split = 0;
  long max_mn = 1;
  for (int i = 0; i < n; ++i) {
    const mvae_gemm_tn_f32_problem& q = pr[i];
    if (q.M < 1 || q.N < 1 || q.K < 0 || q.K >= (1L << 31) || !q.A || !q.B || !q.C) return MVAE_ERR_INVALID;
    if ((q.lda | q.ldb | q.a_gstride | q.b_gstride) & 3) return MVAE_ERR_INVALID;
    if ((reinterpret_cast<uintptr_t>(q.A) | reinterpret_cast<uintptr_t>(q.B)) & 15) return MVAE_ERR_INVALID;
    const Plan pl = tnf_multi_plan(q, n);
    GemmArgs& p = m.p[i];
    p.A = q.A; p.B = q.B; p.C = q.C; p.bias = nullptr; p.partial = nullptr;
    p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc; p.M = q.M; p.N = q.N; p.K = (int)q.K;
    p.c_dtype = MVAE_F32; p.act = MVAE_ACT_NONE; p.accumulate = q.accumulate;
    p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n; p.splits = pl.splits; p.kper = pl.kper;
    p.a_group = q.a_group; p.a_gstride = q.a_gstride; p.b_group = q.b_group; p.b_gstride = q.b_gstride; p.a_total = 0; p.b_total = 0;
    p.colsum_out = q.colsum_out; p.colsum_partial = nullptr; p.colsum_acc = q.colsum_accumulate; p.colsum_nparts = pl.splits; p.ones_col = q.colsum_out ? q.N : -1;
    if (pl.splits > 1) {
      p.partial = reinterpret_cast<float*>(wsp);
      if (q.colsum_out) p.colsum_partial = p.partial + (size_t)pl.splits * q.M * q.N;
      wsp += tnf_multi_need(q, pl);
      any_split = 1;
      if ((long)q.M * q.N > max_mn) max_mn = (long)q.M * q.N;
    }
    m.flags[i] = (q.x3 ? 1 : 0) | (q.colsum_out ? 2 : 0);
    m.first[i] = blocks;
    blocks += pl.tiles_m * pl.tiles_n * pl.splits;
  }
  m.first[n] = blocks;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gemm_tn_f32_multi_kernel, dim3(blocks), dim3(256), 0, st, m);
  MVAE_CHECK_HIP(hipGetLastError());
  if (any_split) {
    int bx = (int)((max_mn + 255) / 256);
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(gemm_splitk_reduce_multi_kernel, dim3(bx, n), dim3(256), 0, st, m);
    MVAE_CHECK_HIP(hipGetLastError());
  }
  return MVAE_OK;
}

// ---- grouped launcher (see gemm_tn_bf16_256_grouped_kernel).  Every problem: bf16 K-major operands, fp32 C, N a multiple of 256, and --
// for column sums -- exactly 4 tile columns (N == 1024).  Operands beyond 2 GiB (K * ld * 2 bytes) are not served here.
extern "C" int mvae_gemm_tn_grouped_supported(int M, int N, int64_t K, int64_t lda, int64_t ldb) {
  return (M >= 256 && N >= 256 && N % 256 == 0 && K >= 64 && (lda % 8) == 0 && (ldb % 8) == 0 && K * lda * 2 < (1L << 31) && K * ldb * 2 < (1L << 31)) ? 1 : 0;
}
extern "C" size_t mvae_gemm_tn_grouped_workspace(int n, const mvae_gemm_tn_problem* pr) {
  size_t b = 0;
  for (int i = 0; i < n; ++i) if (pr[i].colsum_out) b += (size_t)2 * pr[i].M * sizeof(float);
  return b;
}
extern "C" int mvae_gemm_tn_grouped(int n, const mvae_gemm_tn_problem* pr, void* ws, size_t ws_bytes, void* stream) {
  return mvae_gemm_tn_grouped_capped(n, pr, 0, ws, ws_bytes, stream);
}
extern "C" int mvae_gemm_tn_grouped_capped(int n, const mvae_gemm_tn_problem* pr, int max_workgroups, void* ws, size_t ws_bytes, void* stream) {
  if (n < 1 || n > MVAE_MAX_LAYERS * 2 || !pr) return MVAE_ERR_INVALID;
  if (mvae_gemm_tn_grouped_workspace(n, pr) > ws_bytes || (mvae_gemm_tn_grouped_workspace(n, pr) && !ws)) return MVAE_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  TnGroupArgs a;
  TnColsumFinish f;
  f.n = 0;
  float* wsp = reinterpret_cast<float*>(ws);
  int tiles = 0, maxM = 0;
  bool any_cs = false;
  for (int i = 0; i < n; ++i) {
    const mvae_gemm_tn_problem& q = pr[i];
    if (!q.A || !q.B || !q.C) return MVAE_ERR_INVALID;
    if ((reinterpret_cast<uintptr_t>(q.A) | reinterpret_cast<uintptr_t>(q.B)) & 15) return MVAE_ERR_INVALID;
    if (!mvae_gemm_tn_grouped_supported(q.M, q.N, q.K, q.lda, q.ldb)) return MVAE_ERR_UNSUPPORTED;
    TnGroup& g = a.g[i];
    g.A = q.A; g.B = q.B; g.C = q.C; g.lda = q.lda; g.ldb = q.ldb; g.ldc = q.ldc; g.M = q.M; g.N = q.N; g.K = (int)q.K; g.accumulate = q.accumulate;
    g.tiles_m = (q.M + 255) / 256; g.tiles_n = q.N / 256; g.tile0 = tiles; g.cs_partial = nullptr;
    if (q.colsum_out) {
      if (g.tiles_n != 4) return MVAE_ERR_UNSUPPORTED;
      g.cs_partial = wsp; wsp += (size_t)2 * q.M;
      f.partial[f.n] = g.cs_partial; f.out[f.n] = q.colsum_out; f.M[f.n] = q.M; f.acc[f.n] = q.colsum_accumulate; ++f.n;
      any_cs = true;
      if (q.M > maxM) maxM = q.M;
    }
    tiles += g.tiles_m * g.tiles_n;
  }
  a.ngroups = n; a.ntiles = tiles;
  static bool attr_set = false;
  if (!attr_set) {
    MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_bf16_256_grouped_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_bf16_256_grouped_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  int grid = tiles;
  if (max_workgroups > 0 && max_workgroups < grid) grid = max_workgroups;
  if (any_cs) hipLaunchKernelGGL(gemm_tn_bf16_256_grouped_kernel<true>, dim3(grid), dim3(512), 2 * 65536, st, a);
  else hipLaunchKernelGGL(gemm_tn_bf16_256_grouped_kernel<false>, dim3(grid), dim3(512), 2 * 65536, st, a);
  MVAE_CHECK_HIP(hipGetLastError());
  if (any_cs) {
    hipLaunchKernelGGL(gemm_tn_colsum_finish_kernel, dim3((maxM + 255) / 256, f.n), dim3(256), 0, st, f);
    MVAE_CHECK_HIP(hipGetLastError());
  }
  return MVAE_OK;
}
