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
};

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

template <typename T, int BM, int BN, bool PIPE>
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
  auto rowA = [&](int r) -> const T* { const int gm = m0 + r; return gm < p.M ? A + (long)gm * p.lda : nullptr; };
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
    auto offA = [&](int r) -> uint32_t { const int gm = m0 + r; return gm < p.M ? (uint32_t)gm * (uint32_t)p.lda * sz : PIPE_OOB; };
    auto offB = [&](int r) -> uint32_t { const int gn = n0 + r; return gn < p.N ? (uint32_t)gn * (uint32_t)p.ldb * sz : PIPE_OOB; };
    const uint32_t bytesA = (uint32_t)((long)p.M * p.lda * sz - kbeg * sz), bytesB = (uint32_t)((long)p.N * p.ldb * sz - kbeg * sz);
    pipe_seg_init<T, BM, BN>(s0, A + kbeg, bytesA, B + kbeg, bytesB, offA, offB, (int)(kend - kbeg), tid);
    s1 = s0; s1.nk = 0;
    tile_gemm_pipe<T, BM, BN, MI, NI, NBUF, NI, 0>(smem, s0, s1, wm * WM, wn * WN, acc, tid);
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

namespace {
struct Plan { int bm, tiles_m, tiles_n, splits; long kper; };

Plan make_plan(int M, int N, int K, int dtype) {
  Plan pl;
  const int ke = (dtype == MVAE_BF16) ? 64 : 32;
  const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  pl.bm = (t128 >= 192) ? 128 : 64;
  pl.tiles_m = (M + pl.bm - 1) / pl.bm;
  pl.tiles_n = (N + pl.bm - 1) / pl.bm;
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
  Plan pl = make_plan(M, N, K, dtype);
  return pl.splits > 1 ? (size_t)pl.splits * M * N * sizeof(float) : 0;
}

int launch_gemm_nt(int dtype, int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                   int c_dtype, const float* bias, int act, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  if (M <= 0 || N <= 0) return MVAE_OK;
  if (K < 0 || !A || !B || !C) return MVAE_ERR_INVALID;
  if (dtype != MVAE_F32 && dtype != MVAE_BF16) return MVAE_ERR_INVALID;
  if (c_dtype != MVAE_F32 && c_dtype != MVAE_BF16) return MVAE_ERR_INVALID;
  if (accumulate && c_dtype != MVAE_F32) return MVAE_ERR_INVALID;
  Plan pl = make_plan(M, N, K, dtype);
  GemmArgs p;
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.partial = nullptr;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.c_dtype = c_dtype; p.act = act; p.accumulate = accumulate;
  p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n; p.splits = pl.splits; p.kper = pl.kper;
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
                    ((long)M * lda * sz < (1L << 31)) && ((long)N * ldb * sz < (1L << 31)) && K >= 4 * ke;
  const size_t lds = (size_t)(pipe ? 4 : 2) * (pl.bm + pl.bm) * KB;
#define MVAE_GEMM_LAUNCH(TT_, BM_, PIPE_)                                                             \
  do {                                                                                                \
    auto kern = gemm_nt_kernel<TT_, BM_, BM_, PIPE_>;                                                 \
    static bool attr_set = false;                                                                     \
    if (!attr_set) { MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; } \
    hipLaunchKernelGGL(kern, grid, block, lds, st, p);                                                \
  } while (0)
  if (dtype == MVAE_BF16) {
    if (pl.bm == 128) { if (pipe) MVAE_GEMM_LAUNCH(bf16_t, 128, true); else MVAE_GEMM_LAUNCH(bf16_t, 128, false); }
    else { if (pipe) MVAE_GEMM_LAUNCH(bf16_t, 64, true); else MVAE_GEMM_LAUNCH(bf16_t, 64, false); }
  } else {
    if (pl.bm == 128) { if (pipe) MVAE_GEMM_LAUNCH(float, 128, true); else MVAE_GEMM_LAUNCH(float, 128, false); }
    else { if (pipe) MVAE_GEMM_LAUNCH(float, 64, true); else MVAE_GEMM_LAUNCH(float, 64, false); }
  }
#undef MVAE_GEMM_LAUNCH
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
  const char* fe = getenv("MVAE_TN_BM");
  const int force_bm = fe ? atoi(fe) : 0;
  pl.bm = ((long)((M + 255) / 256) * ((N + 127) / 128) >= 128) ? 256 : 128;
  if (force_bm == 128 || force_bm == 256) pl.bm = force_bm;
  pl.tiles_m = (M + pl.bm - 1) / pl.bm; pl.tiles_n = (N + 127) / 128;
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
  if (M <= 0 || N <= 0) return MVAE_OK;
  if (K < 0 || !A || !B || !C) return MVAE_ERR_INVALID;
  if (c_dtype != MVAE_F32 && c_dtype != MVAE_BF16) return MVAE_ERR_INVALID;
  if (accumulate && c_dtype != MVAE_F32) return MVAE_ERR_INVALID;
  if ((lda % 8) || (ldb % 8) || ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15)) return MVAE_ERR_INVALID;
  if ((long)K * lda * 2 >= (1L << 31) || (long)K * ldb * 2 >= (1L << 31)) return MVAE_ERR_UNSUPPORTED;
  Plan pl = make_plan_tn(M, N, K);
  GemmArgs p;
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.partial = nullptr;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.c_dtype = c_dtype; p.act = act; p.accumulate = accumulate;
  p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n; p.splits = pl.splits; p.kper = pl.kper;
  if (pl.splits > 1) {
    const size_t need = (size_t)pl.splits * M * N * sizeof(float);
    if (!ws || ws_bytes < need) return MVAE_ERR_WORKSPACE;
    p.partial = reinterpret_cast<float*>(ws);
  }
  dim3 grid(pl.tiles_m * pl.tiles_n, pl.splits);
  static bool attr_set = false;
  if (!attr_set) {
    MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_bf16_kernel<4, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_bf16_kernel<3, 8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_bf16_kernel<4, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_bf16_kernel<3, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const char* we = getenv("MVAE_TN_WS");
  const bool wspec = we ? atoi(we) != 0 : true;      // loader / consumer wave specialisation (512-thread workgroups)
  if (wspec) {
    if (pl.bm == 256) hipLaunchKernelGGL((gemm_tn_bf16_kernel<3, 8, true>), grid, dim3(512), 3 * (32768 + 16384), st, p);
    else {
      const char* ne = getenv("MVAE_NBUF_TN");
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
