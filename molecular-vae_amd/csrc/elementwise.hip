// HBM-bound helper kernels of the SMILES-VAE training path (gfx950): casts/transposes, embedding-table gather and
// its deterministic scatter, SELU derivative, reparameterisation, softmax head, ELBO, reductions,
// gradient-norm + Adam.  All reductions use fixed orders (wave shuffle -> LDS -> serial over blocks) so results
// are bitwise reproducible run to run.
#include <atomic>
#include "common.hpp"
#include "kernels.hpp"
#include <stdlib.h>

static inline int grid_for(long n, int per_block = 256, int cap = 2048) {
  long b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// ------------------------------------------------------------------------------------------- cast / transpose
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_transpose_kernel(int R, int C, const TS* src, long lds_, TD* dst, long ldd, TD* dstT, long ldt) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + i * 8, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < C) v = TT<TS>::ld(src + (long)r * lds_ + c);
    tile[ty + i * 8][tx] = v;
    if (dst && r < R && c < ldd) TT<TD>::st(dst + (long)r * ldd + c, v);   // zero fill of the pad columns
  }
  if (!dstT) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + i * 8, r = r0 + tx;     // dstT[c][r]
    if (c < C && r < ldt) TT<TD>::st(dstT + (long)c * ldt + r, tile[tx][ty + i * 8]);
  }
}

int launch_cast_transpose(int dsrc, int ddst, int R, int C, const void* src, long lds_, void* dst, long ldd, void* dstT,
                          long ldt, hipStream_t st) {
  if (R <= 0 || C <= 0) return MVAE_OK;
  if (!src || (!dst && !dstT)) return MVAE_ERR_INVALID;
  if (dst && ldd < C) return MVAE_ERR_INVALID;
  if (dstT && ldt < R) return MVAE_ERR_INVALID;
  const long cmax = (dst && ldd > C) ? ldd : C;     // cover pad columns of dst
  const long rmax = (dstT && ldt > R) ? ldt : R;    // cover pad columns of dstT
  dim3 grid((unsigned)((cmax + 31) / 32), (unsigned)((rmax + 31) / 32)), block(256);
  if (dsrc == MVAE_F32 && ddst == MVAE_F32)
    hipLaunchKernelGGL((cast_transpose_kernel<float, float>), grid, block, 0, st, R, C, (const float*)src, lds_, (float*)dst, ldd, (float*)dstT, ldt);
  else if (dsrc == MVAE_F32 && ddst == MVAE_BF16)
    hipLaunchKernelGGL((cast_transpose_kernel<float, bf16_t>), grid, block, 0, st, R, C, (const float*)src, lds_, (bf16_t*)dst, ldd, (bf16_t*)dstT, ldt);
  else if (dsrc == MVAE_BF16 && ddst == MVAE_BF16)
    hipLaunchKernelGGL((cast_transpose_kernel<bf16_t, bf16_t>), grid, block, 0, st, R, C, (const bf16_t*)src, lds_, (bf16_t*)dst, ldd, (bf16_t*)dstT, ldt);
  else if (dsrc == MVAE_BF16 && ddst == MVAE_F32)
    hipLaunchKernelGGL((cast_transpose_kernel<bf16_t, float>), grid, block, 0, st, R, C, (const bf16_t*)src, lds_, (float*)dst, ldd, (float*)dstT, ldt);
  else return MVAE_ERR_INVALID;
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}

// ------------------------------------------------------------------------------------------- multi-tensor pack
// ONE launch for a whole list of small weight-packing jobs (fp32 master -> bf16 / padded / transposed shadows, bias sums, block copies):
// a model refreshes 35-60 such shadows after every optimiser step, each a 3-5 us launch of its own before -- pure launch latency in front
// of the step (0.25 ms of a 6.3 ms MOSES step).  The job table lives in device memory (pointers are stable from step to step, so the host
// builds it once); block -> job by binary search over the jobs' first-block indices.
//   kind 0  cast / transpose: the 32 x 32 tile (by, bx) of  dst[r, c] = (Td) src[r, c] and / or dstT[c, r] = (Td) src[r, c] for r < R,
//           c < C ONLY -- unlike mvae_cast_transpose nothing outside the R x C block is touched (destinations may be sub-blocks of a
//           larger buffer whose padding the caller allocated zero)
//   kind 1  out[i] = a[i] + b[i]            (fp32, n = R * C elements; 1024 per block)
//   kind 2  dst[r, c] = src[r, c]           (fp32 block copy with leading dimensions; 32 x 32 tiles)
__global__ __launch_bounds__(256) void pack_multi_kernel(int njobs, const mvae_pack_job* __restrict__ jobs) {
  __shared__ float tile[32][33];
  int lo = 0, hi = njobs - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
  const mvae_pack_job q = jobs[lo];
  const int blk = blockIdx.x - q.block0;
  if (q.kind == 1) {
    const float* a = reinterpret_cast<const float*>(q.src); const float* b = reinterpret_cast<const float*>(q.src2);
    float* o = reinterpret_cast<float*>(q.dst);
    const long n = (long)q.R * q.C;
    for (int k = 0; k < 4; ++k) { const long i = (long)blk * 1024 + k * 256 + threadIdx.x; if (i < n) o[i] = a[i] + b[i]; }
    return;
  }
  const int tiles_x = (q.C + 31) / 32;
  const int c0 = (blk % tiles_x) * 32, r0 = (blk / tiles_x) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (q.kind == 2) {
    const float* src = reinterpret_cast<const float*>(q.src); float* dst = reinterpret_cast<float*>(q.dst);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int r = r0 + ty + i * 8, c = c0 + tx; if (r < q.R && c < q.C) dst[(long)r * q.ldd + c] = src[(long)r * q.lds + c]; }
    return;
  }
  auto ld = [&](int r, int c) -> float {
    return q.src_dtype == MVAE_BF16 ? TT<bf16_t>::ld(reinterpret_cast<const bf16_t*>(q.src) + (long)r * q.lds + c)
                                    : reinterpret_cast<const float*>(q.src)[(long)r * q.lds + c];
  };
  auto st = [&](void* base, long off, float v) {
    if (q.dst_dtype == MVAE_BF16) TT<bf16_t>::st(reinterpret_cast<bf16_t*>(base) + off, v); else reinterpret_cast<float*>(base)[off] = v;
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + i * 8, c = c0 + tx;
    float v = 0.f;
    if (r < q.R && c < q.C) v = ld(r, c);
    tile[ty + i * 8][tx] = v;
    if (q.dst && r < q.R && c < q.C) st(q.dst, (long)r * q.ldd + c, v);
  }
  if (!q.dstT) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + i * 8, r = r0 + tx;
    if (c < q.C && r < q.R) st(q.dstT, (long)c * q.ldt + r, tile[tx][ty + i * 8]);
  }
}

// ------------------------------------------------------------------------------------------- embedding table
__global__ __launch_bounds__(256) void gather_rows_tb_kernel(const int64_t* idx, int B, int L, int nrows, const float* table, int W,
                                                             const float* base, float* out) {
  const long n = (long)B * L * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long row = i / W; const int c = (int)(i - row * W);
    const int t = (int)(row / B), b = (int)(row - (long)t * B);
    long id = idx[(long)b * L + t];
    id = id < 0 ? 0 : (id >= nrows ? nrows - 1 : id);
    out[i] = table[id * W + c] + (base ? base[(long)b * W + c] : 0.f);
  }
}

// stage 1: block (k, s) owns rows [k*RPB, (k+1)*RPB) of d and the column slice [s*WS, (s+1)*WS); thread j owns one column of the slice;
// LDS table [nrows][WS] (per-thread column => no races, fixed order => deterministic).
template <typename T>
__global__ __launch_bounds__(256) void scatter_rows_tb_stage1(const int64_t* idx, int B, int L, int nrows, const T* d, long ldd, int W,
                                                              int rpb, int ws_cols, float* partial) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* tab = reinterpret_cast<float*>(smem_raw);   // [nrows][ws_cols]
  const long total = (long)B * L;
  const int c_lo = blockIdx.y * ws_cols;
  int c_hi = c_lo + ws_cols; if (c_hi > W) c_hi = W;
  const int wsl = c_hi - c_lo;
  for (int i = threadIdx.x; i < nrows * ws_cols; i += 256) tab[i] = 0.f;
  __syncthreads();
  const long rbeg = (long)blockIdx.x * rpb;
  long rend = rbeg + rpb; if (rend > total) rend = total;
  // eight rows per round: their ids and values are requested together (one load latency per round), then applied in row order
  for (int c = threadIdx.x; c < wsl; c += 256) {
    for (long row0 = rbeg; row0 < rend; row0 += 8) {
      float v[8]; int id[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const long row = row0 + r;
        if (row < rend) {
          const int t = (int)(row / B), b = (int)(row - (long)t * B);
          long q = idx[(long)b * L + t];
          id[r] = (int)(q < 0 ? 0 : (q >= nrows ? nrows - 1 : q));
          v[r] = TT<T>::ld(d + row * ldd + c_lo + c);
        } else { id[r] = 0; v[r] = 0.f; }
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) tab[id[r] * ws_cols + c] += v[r];
    }
  }
  __syncthreads();
  float* out = partial + (long)blockIdx.x * nrows * W;
  for (int i = threadIdx.x; i < nrows * wsl; i += 256) { const int r = i / wsl, c = i - r * wsl; out[(long)r * W + c_lo + c] = tab[r * ws_cols + c]; }
}
// out[i] = sum_k partial[k][i]: 64 elements x 4 interleaved k-slots per block, fixed combination order (deterministic)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* partial, int nparts, long n, float* out) {
  __shared__ float red[4][64];
  const int e = threadIdx.x & 63, slot = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + e;
  float v = 0.f;
  if (i < n) for (int k = slot; k < nparts; k += 4) v += partial[(long)k * n + i];
  red[slot][e] = v;
  __syncthreads();
  if (slot == 0 && i < n) out[i] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// out[n, b, a] = in[n, a, b]   (batched 2-D transpose; flatten order of models.py:6-10 <-> GEMM row order)
__global__ __launch_bounds__(256) void permute021_kernel(int N, int A, int Bd, const float* in, float* out) {
  const long n = (long)N * A * Bd;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int a = (int)(i % A); const long nb = i / A; const int b = (int)(nb % Bd); const long nn = nb / Bd;
    out[i] = in[(nn * A + a) * Bd + b];
  }
}

__global__ __launch_bounds__(256) void selu_bwd_kernel(long n, float* dy, const float* y) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dy[i] *= selu_grad_from_out(y[i]);
}
int launch_selu_bwd(long n, float* dy, const float* y, hipStream_t st) {
  if (n <= 0) return MVAE_OK;
  hipLaunchKernelGGL(selu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, n, dy, y);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}

// ------------------------------------------------------------------------------------------- reductions
// out[n] = sum_m X[m, n], two fixed-order stages: (column tile of 64) x (row chunk) partial sums, then a serial sum over chunks.
__global__ __launch_bounds__(256) void colsum_stage1_kernel(int M, int N, const float* X, long ldx, int rows_per_chunk, float* partial) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_chunk;
  int r1 = r0 + rows_per_chunk; if (r1 > M) r1 = M;
  float v = 0.f;
  if (c < N) {
    // eight independent loads in flight per thread; fixed combination order (deterministic)
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int m = r0 + g;
    for (; m + 28 < r1; m += 32) {
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += X[(long)(m + 4 * k) * ldx + c];
    }
    for (; m < r1; m += 4) a[0] += X[(long)m * ldx + c];
    v = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  red[g][threadIdx.x & 63] = v;
  __syncthreads();
  if (g == 0 && c < N) partial[(long)blockIdx.y * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
static int colsum_chunks(int M) { int c = (M + 127) / 128; if (c > 256) c = 256; if (c < 1) c = 1; return c; }
size_t colsum_workspace_bytes(int M, int N) { return (size_t)colsum_chunks(M) * N * sizeof(float); }
int launch_colsum(int M, int N, const float* X, long ldx, float* out, void* ws, size_t ws_bytes, hipStream_t st) {
  if (N <= 0) return MVAE_OK;
  const int chunks = colsum_chunks(M);
  if (!ws || ws_bytes < colsum_workspace_bytes(M, N)) return MVAE_ERR_WORKSPACE;
  const int rpc = (M + chunks - 1) / chunks;
  hipLaunchKernelGGL(colsum_stage1_kernel, dim3((N + 63) / 64, chunks), dim3(256), 0, st, M, N, X, ldx, rpc, (float*)ws);
  hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, st, (const float*)ws, chunks, (long)N, out);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
// out[n] = sum_m X[m, n] for X of type T read as 16-byte vectors: thread = 8 columns x (every 8th row of its chunk);
// block = 32 column groups (256 columns) x 8 row slots; fixed-order LDS + chunk reduction.
template <typename T>
__global__ __launch_bounds__(256) void colsum_t_stage1_kernel(int M, int N, const T* X, long ldx, int rows_per_chunk, float* partial) {
  __shared__ float red[8][256 + 8];
  const int cg = threadIdx.x & 31, rs = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 256 + cg * 8;
  const int r0 = blockIdx.y * rows_per_chunk;
  int r1 = r0 + rows_per_chunk; if (r1 > M) r1 = M;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < N) {
    const int n = (N - c0 < 8) ? (N - c0) : 8;
    for (int m = r0 + rs; m < r1; m += 8) {
      float v[8];
      const T* p = X + (long)m * ldx + c0;
      if (n == 8) {
        if constexpr (sizeof(T) == 2) {
          const uint4 u = *reinterpret_cast<const uint4*>(p);
          const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) { v[2 * i] = __builtin_bit_cast(float, w[i] << 16); v[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u); }
        } else {
          const float4 x0 = *reinterpret_cast<const float4*>(p), x1 = *reinterpret_cast<const float4*>(p + 4);
          v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (e < n) ? TT<T>::ld(p + e) : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rs][cg * 8 + e] = a[e];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < N) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) v += red[k][threadIdx.x];
    partial[(long)blockIdx.y * N + c] = v;
  }
}
static int colsum_t_chunks(int M) { int c = (M + 511) / 512; if (c > 64) c = 64; if (c < 1) c = 1; return c; }

// out[r] (+)= sum_c X[r, c]: one wave per row, 16-byte loads when the row is 16-byte aligned
template <typename T> __device__ __forceinline__ float sum_chunk16(const uint4& v);
template <> __device__ __forceinline__ float sum_chunk16<float>(const uint4& v) {
  return (__builtin_bit_cast(float, v.x) + __builtin_bit_cast(float, v.y)) + (__builtin_bit_cast(float, v.z) + __builtin_bit_cast(float, v.w));
}
template <> __device__ __forceinline__ float sum_chunk16<bf16_t>(const uint4& v) {
  float a = 0.f;
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) a += __builtin_bit_cast(float, w[i] << 16) + __builtin_bit_cast(float, w[i] & 0xffff0000u);
  return a;
}
template <typename T>
__global__ __launch_bounds__(256) void rowsum_kernel(int R, int C, const T* X, long ldx, float* out, int accumulate) {
  constexpr int EPC = TT<T>::EPC;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const T* p = X + (long)r * ldx;
  float v = 0.f;
  int c0 = 0;
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    const int nch = C / EPC;
    for (int i = lane; i < nch; i += 64) v += sum_chunk16<T>(*reinterpret_cast<const uint4*>(p + (long)i * EPC));
    c0 = nch * EPC;
  }
  for (int c = c0 + lane; c < C; c += 64) v += TT<T>::ld(p + c);
  v = wave_sum(v);
  if (lane == 0) out[r] = accumulate ? out[r] + v : v;
}
// out[b, w] = sum_t X[t, b, w]
template <typename T>
__global__ __launch_bounds__(256) void timesum_kernel(int T_, long BW, const T* X, float* out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < BW; i += (long)gridDim.x * 256) {
    float v = 0.f;
    for (int t = 0; t < T_; ++t) v += TT<T>::ld(X + (long)t * BW + i);
    out[i] = v;
  }
}

// 16-byte version (bf16): one thread per 8 consecutive elements, four time steps in flight
__global__ __launch_bounds__(256) void timesum_bf16x8_kernel(int T_, long BW, const bf16_t* X, float* out) {
  const long n8 = BW / 8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16_t* p = X + i * 8;
    int t = 0;
    for (; t + 4 <= T_; t += 4) {
      uint4 u[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) u[k] = *reinterpret_cast<const uint4*>(p + (long)(t + k) * BW);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t w[4] = {u[k].x, u[k].y, u[k].z, u[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[2 * e] += __builtin_bit_cast(float, w[e] << 16); acc[2 * e + 1] += __builtin_bit_cast(float, w[e] & 0xffff0000u); }
      }
    }
    for (; t < T_; ++t) {
      const uint4 u = *reinterpret_cast<const uint4*>(p + (long)t * BW);
      const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] += __builtin_bit_cast(float, w[e] << 16); acc[2 * e + 1] += __builtin_bit_cast(float, w[e] & 0xffff0000u); }
    }
    *reinterpret_cast<float4*>(out + i * 8) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(out + i * 8 + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}

// ------------------------------------------------------------------------------------------- Lambda (models.py:80-94)
// eps != nullptr: injected noise (device memory, or PINNED HOST memory read in place over the host link -- the "cpu" noise source needs no copy
// command in the stream).  eps == nullptr: the draw of models.py:92 happens HERE -- element i is scale * N(0,1) from the counter hash of
// (seed, offset + i) (common.hpp normal_draw; no generator state anywhere).  eps_out, when given, receives the noise used (what backward reads).
__global__ __launch_bounds__(256) void lambda_fwd_kernel(int B, int o, const float* mulv, const float* eps, float scale, uint32_t seed,
                                                         uint64_t offset, float* eps_out, float* z, float* mu, float* logvar) {
  const long n = (long)B * o;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / o; const int c = (int)(i - b * o);
    const float m = mulv[b * 2 * o + c], lv = mulv[b * 2 * o + o + c];
    const float e = eps ? eps[i] : scale * normal_draw(seed, offset + (uint64_t)i);
    if (eps_out) eps_out[i] = e;
    mu[i] = m; logvar[i] = lv;
    z[i] = m + expf(lv * 0.5f) * e;
  }
}
__global__ __launch_bounds__(256) void normal_fill_kernel(long n, float scale, uint32_t seed, uint64_t offset, float* out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = scale * normal_draw(seed, offset + (uint64_t)i);
}
__global__ __launch_bounds__(256) void lambda_bwd_kernel(int B, int o, const float* mulv, const float* eps, const float* dz, const float* dmu,
                                                         const float* dlogvar, float* dmulv) {
  const long n = (long)B * o;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / o; const int c = (int)(i - b * o);
    const float lv = mulv[b * 2 * o + o + c];
    const float g = dz ? dz[i] : 0.f;
    dmulv[b * 2 * o + c] = (dmu ? dmu[i] : 0.f) + g;
    dmulv[b * 2 * o + o + c] = (dlogvar ? dlogvar[i] : 0.f) + g * eps[i] * 0.5f * expf(lv * 0.5f);
  }
}

// ------------------------------------------------------------------------------------------- softmax head
// one wave per (t,b) row; C <= 64*4
// One wave handles SM_ROWS consecutive rows with all their loads issued before the first reduction: with one row per wave the kernel is a
// chain of dependent load -> shuffle-reduce -> exp -> reduce -> store per wave and runs at the occupancy x latency rate (1.6 TB/s measured at
// B*L = 524288 rows of 64 classes), not at the HBM rate.  C <= 64 * SM_CPL classes.
constexpr int SM_ROWS = 4, SM_CPL = 2;
__global__ __launch_bounds__(256) void softmax_tb_fwd_kernel(int B, int L, int C, const float* logits, long ldl, float* recon) {
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * SM_ROWS;
  const int lane = threadIdx.x & 63;
  const long rows = (long)B * L;
  float x[SM_ROWS][SM_CPL];
#pragma unroll
  for (int r = 0; r < SM_ROWS; ++r)
#pragma unroll
    for (int k = 0; k < SM_CPL; ++k) {
      const int c = lane + 64 * k;
      x[r][k] = (row0 + r < rows && c < C) ? logits[(row0 + r) * ldl + c] : -INFINITY;
    }
#pragma unroll
  for (int r = 0; r < SM_ROWS; ++r) {
    const long row = row0 + r;
    if (row >= rows) break;                                   // wave-uniform
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < SM_CPL; ++k) mx = fmaxf(mx, x[r][k]);
    mx = wave_max(mx);
    float e[SM_CPL], s = 0.f;
#pragma unroll
    for (int k = 0; k < SM_CPL; ++k) { e[k] = (lane + 64 * k < C) ? expf(x[r][k] - mx) : 0.f; s += e[k]; }
    s = wave_sum(s);
    const int t = (int)(row / B), b = (int)(row - (long)t * B);
    float* out = recon + ((long)b * L + t) * C;
#pragma unroll
    for (int k = 0; k < SM_CPL; ++k) if (lane + 64 * k < C) out[lane + 64 * k] = e[k] / s;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void softmax_tb_bwd_kernel(int B, int L, int C, const float* recon, const float* drecon, T* dl, long ldd,
                                                             T* dlT, long ldT) {
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * SM_ROWS;
  const int lane = threadIdx.x & 63;
  const long rows = (long)B * L;
  float p[SM_ROWS][SM_CPL], dp[SM_ROWS][SM_CPL];
#pragma unroll
  for (int r = 0; r < SM_ROWS; ++r) {
    const long row = row0 + r;
    const int t = (int)(row / B), b = (int)(row - (long)t * B);
    const long src = ((long)b * L + t) * C;
#pragma unroll
    for (int k = 0; k < SM_CPL; ++k) {
      const int c = lane + 64 * k;
      const bool ok = row < rows && c < C;
      p[r][k] = ok ? recon[src + c] : 0.f;
      dp[r][k] = ok ? drecon[src + c] : 0.f;
    }
  }
#pragma unroll
  for (int r = 0; r < SM_ROWS; ++r) {
    const long row = row0 + r;
    if (row >= rows) break;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < SM_CPL; ++k) s += dp[r][k] * p[r][k];
    s = wave_sum(s);
#pragma unroll
    for (int k = 0; k < SM_CPL; ++k) {
      const int c = lane + 64 * k;
      if (c < ldd) {
        const float v = (c < C) ? p[r][k] * (dp[r][k] - s) : 0.f;
        TT<T>::st(dl + row * ldd + c, v);
        if (dlT && c < C) TT<T>::st(dlT + (long)c * ldT + row, v);
      }
    }
  }
}

// ---- tiled forms (round 3).  The head's two layouts differ in ROW ORDER: logits / dl are time-major (row t*B + b, what the recurrent
// kernels produce and consume), recon / drecon batch-major ([B, L, C], what the reference returns).  A workgroup owns a tile of TT time
// steps x BB batch rows, ONE ROW PER THREAD: on the time-major side the BB rows of a time step are one contiguous run of BB*C floats, on
// the batch-major side the TT rows of a molecule are one contiguous run of TT*C floats, so both sides move as flat 16-byte streams through
// an LDS image [TT*BB][C|1] (odd row stride: the per-row passes of 64 lanes hit 64 different banks).  No lane idles for C < 64, no shuffle
// reductions, and the transposition costs no uncoalesced access.  (One wave per row -- the kernels above -- left 45 % of the lanes idle at
// C = 35 and paid two 6-step shuffle reductions per 140-byte row: 0.15 / 0.25 of the HBM roof at the headline shape.)
// Tile walker.  One side of a tile is `nruns` runs of contiguous memory (time-major side: one run per time step = the tile's batch rows;
// batch-major side: one run per molecule = the tile's time steps), each run `rows_in` rows of `W` units (floats, or 16-byte chunks), moved in
// pieces of `UPP` units; `rq` = pieces per run.  A thread visits pieces Q = tid, tid + NT, ...  The walker keeps (run, piece in run) and
// (row in run, unit in row) of the current piece and advances all four by the fixed stride WITHOUT a division (runtime-divisor integer
// divisions cost ~25 instructions each; two per 16-byte piece had made the first tiled form ALU-bound).
struct TileWalk {
  int run, q, row, c;           // current piece: run index, piece inside the run, row inside the run, unit inside the row (of its FIRST unit)
  int d_run, d_q, d_row, d_c;   // stride decomposed the same way
  int rq, W, rows_in;
  __device__ __forceinline__ void init(int Q0, int stride, int rq_, int W_, int rows_in_, int upp) {
    rq = rq_; W = W_; rows_in = rows_in_;
    run = Q0 / rq; q = Q0 - run * rq;
    const int f = q * upp; row = f / W; c = f - row * W;
    d_run = stride / rq; d_q = stride - d_run * rq;
    const int g = d_q * upp; d_row = g / W; d_c = g - d_row * W;
  }
  __device__ __forceinline__ void step() {
    run += d_run; q += d_q; row += d_row; c += d_c;
    if (c >= W) { c -= W; ++row; }
    if (q >= rq) { q -= rq; ++run; row -= rows_in; }
  }
};

// LDS image [TT_*BB_][C|1] (odd stride: the per-row passes of 64 lanes hit 64 banks).  Phase 1 fills it in TIME-major row order (row =
// ti * BB_ + bi); phase 2 (one row per thread, the row held in registers) computes and writes the result back in BATCH-major row order (row =
// bi * TT_ + ti) after a barrier -- the transposition happens there, in place; phase 3 streams the image out.  Every global access is a
// 16-byte piece of a contiguous run, all of a batch's loads are in flight before the first is used.
template <int TT_, int BB_, int CMAX>
__global__ __launch_bounds__(TT_ * BB_) void softmax_tile_fwd_kernel(int B, int L, int C, const float* __restrict__ logits, float* __restrict__ recon) {
  extern __shared__ __attribute__((aligned(16))) float sm_t[];
  constexpr int ROWS = TT_ * BB_, U = 8;
  const int CP = C | 1;
  const int tiles_b = (B + BB_ - 1) / BB_;
  const int tb = blockIdx.x % tiles_b, tt = blockIdx.x / tiles_b;
  const int b0 = tb * BB_, t0 = tt * TT_;
  const int nb = (B - b0 < BB_) ? (B - b0) : BB_, nt = (L - t0 < TT_) ? (L - t0) : TT_;
  const int tid = threadIdx.x;
  const int run = nb * C;
  if (((run & 3) == 0) && ((((long)B * C) & 3) == 0) && ((((long)b0 * C) & 3) == 0)) {
    const int rq = run >> 2, total = nt * rq;
    TileWalk w; w.init(tid, ROWS, rq, C, nb, 4);
    for (int Q0 = tid; Q0 < total; Q0 += ROWS * U) {
      float4 v[U]; int lrow[U], lc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (Q0 + u * ROWS < total) {
          v[u] = *reinterpret_cast<const float4*>(logits + ((long)(t0 + w.run) * B + b0) * C + 4 * w.q);
          lrow[u] = w.run * BB_ + w.row; lc[u] = w.c;
          w.step();
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (Q0 + u * ROWS < total) {
          const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
          int r = lrow[u], c = lc[u];
#pragma unroll
          for (int j = 0; j < 4; ++j) { sm_t[r * CP + c] = e[j]; if (++c == C) { c = 0; ++r; } }
        }
      }
    }
  } else {
    for (int q = tid; q < nt * run; q += ROWS) { const int ti = q / run, f = q - ti * run, r = f / C, c = f - r * C; sm_t[(ti * BB_ + r) * CP + c] = logits[((long)(t0 + ti) * B + b0) * C + f]; }
  }
  __syncthreads();
  const int ti_ = tid / BB_, bi_ = tid - ti_ * BB_;
  const bool live = ti_ < nt && bi_ < nb;
  float x[CMAX];
  if (live) {
    const float* rowp = sm_t + tid * CP;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) x[c] = (c < C) ? rowp[c] : -INFINITY;
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) mx = fmaxf(mx, x[c]);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) { x[c] = expf(x[c] - mx); sum += x[c]; }        // exp(-inf) = 0 for the columns past C
    const float inv = 1.f / sum;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) x[c] *= inv;
  }
  __syncthreads();                                                                  // every row is in registers: the image may be permuted
  if (live) {
    float* rowo = sm_t + (bi_ * TT_ + ti_) * CP;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) if (c < C) rowo[c] = x[c];
  }
  __syncthreads();
  const int runo = nt * C;
  if (((runo & 3) == 0) && ((((long)L * C) & 3) == 0) && ((((long)t0 * C) & 3) == 0)) {
    const int rq = runo >> 2, total = nb * rq;
    TileWalk w; w.init(tid, ROWS, rq, C, nt, 4);
    for (int Q = tid; Q < total; Q += ROWS) {
      int r = w.run * TT_ + w.row, c = w.c;
      float e[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { e[j] = sm_t[r * CP + c]; if (++c == C) { c = 0; ++r; } }
      *reinterpret_cast<float4*>(recon + ((long)(b0 + w.run) * L + t0) * C + 4 * w.q) = make_float4(e[0], e[1], e[2], e[3]);
      w.step();
    }
  } else {
    for (int q = tid; q < nb * runo; q += ROWS) { const int bi = q / runo, f = q - bi * runo, ti = f / C, c = f - ti * C; recon[((long)(b0 + bi) * L + t0) * C + f] = sm_t[(bi * TT_ + ti) * CP + c]; }
  }
}

// backward, bf16 dl: dl[(t*B + b), c] = p (dp - sum_c dp p) for c < C, zero for C <= c < C8 = C rounded up to 8; columns >= C8 are NOT written.
// Same three phases the other way round: in batch-major (row = bi * TT_ + ti), out time-major (row = ti * BB_ + bi).
template <int TT_, int BB_, int CMAX>
__global__ __launch_bounds__(TT_ * BB_) void softmax_tile_bwd_kernel(int B, int L, int C, const float* __restrict__ recon, const float* __restrict__ drecon,
                                                                     bf16_t* __restrict__ dl, long ldd) {
  extern __shared__ __attribute__((aligned(16))) float sm_t[];
  constexpr int ROWS = TT_ * BB_, U = 4;
  const int CP = C | 1;
  float* P = sm_t; float* D = sm_t + ROWS * CP;
  const int tiles_b = (B + BB_ - 1) / BB_;
  const int tb = blockIdx.x % tiles_b, tt = blockIdx.x / tiles_b;
  const int b0 = tb * BB_, t0 = tt * TT_;
  const int nb = (B - b0 < BB_) ? (B - b0) : BB_, nt = (L - t0 < TT_) ? (L - t0) : TT_;
  const int tid = threadIdx.x;
  const int runi = nt * C;
  if (((runi & 3) == 0) && ((((long)L * C) & 3) == 0) && ((((long)t0 * C) & 3) == 0)) {
    const int rq = runi >> 2, total = nb * rq;
    TileWalk w; w.init(tid, ROWS, rq, C, nt, 4);
    for (int Q0 = tid; Q0 < total; Q0 += ROWS * U) {
      float4 vp[U], vd[U]; int lrow[U], lc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (Q0 + u * ROWS < total) {
          const long src = ((long)(b0 + w.run) * L + t0) * C + 4 * w.q;
          vp[u] = *reinterpret_cast<const float4*>(recon + src); vd[u] = *reinterpret_cast<const float4*>(drecon + src);
          lrow[u] = w.run * TT_ + w.row; lc[u] = w.c;
          w.step();
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (Q0 + u * ROWS < total) {
          const float ep[4] = {vp[u].x, vp[u].y, vp[u].z, vp[u].w}, ed[4] = {vd[u].x, vd[u].y, vd[u].z, vd[u].w};
          int r = lrow[u], c = lc[u];
#pragma unroll
          for (int j = 0; j < 4; ++j) { P[r * CP + c] = ep[j]; D[r * CP + c] = ed[j]; if (++c == C) { c = 0; ++r; } }
        }
      }
    }
  } else {
    for (int q = tid; q < nb * runi; q += ROWS) {
      const int bi = q / runi, f = q - bi * runi, ti = f / C, c = f - ti * C;
      const long src = ((long)(b0 + bi) * L + t0) * C + f;
      P[(bi * TT_ + ti) * CP + c] = recon[src]; D[(bi * TT_ + ti) * CP + c] = drecon[src];
    }
  }
  __syncthreads();
  const int bi_ = tid / TT_, ti_ = tid - bi_ * TT_;            // the row this thread owns, in the order phase 1 wrote it
  const bool live = ti_ < nt && bi_ < nb;
  float pv[CMAX];
  if (live) {
    const float* pr = P + tid * CP; const float* dr = D + tid * CP;
    float dv[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) { pv[c] = (c < C) ? pr[c] : 0.f; dv[c] = (c < C) ? dr[c] : 0.f; }
    float sdot = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) sdot += dv[c] * pv[c];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) pv[c] = pv[c] * (dv[c] - sdot);
  }
  __syncthreads();
  if (live) {
    float* po = P + (ti_ * BB_ + bi_) * CP;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) if (c < C) po[c] = pv[c];
  }
  __syncthreads();
  // time-major side out: run = one time step = nb rows of c8n 16-byte chunks (row stride ldd in memory: the runs are not dense there, so
  // the walker's piece unit is a chunk and every piece carries its own row)
  const int c8n = (C + 7) >> 3, rq = nb * c8n, total = nt * rq;
  TileWalk w; w.init(tid, ROWS, rq, c8n, nb, 1);
  for (int Q = tid; Q < total; Q += ROWS) {
    const float* pr = P + (w.run * BB_ + w.row) * CP + w.c * 8;
    uint32_t wd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = w.c * 8 + 2 * j;
      const float lo = (c < C) ? pr[2 * j] : 0.f, hi = (c + 1 < C) ? pr[2 * j + 1] : 0.f;
      wd[j] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
    }
    *reinterpret_cast<uint4*>(dl + ((long)(t0 + w.run) * B + b0 + w.row) * ldd + w.c * 8) = make_uint4(wd[0], wd[1], wd[2], wd[3]);
    w.step();
  }
}

// ---- class counts that are a multiple of 4 with 16-byte aligned rows (BASELINE configs[4]: C = 64): no LDS at all.  A row is C / 4
// float4 pieces = LPR lanes (a power of two >= C / 4), so a wave owns 64 / LPR consecutive rows, the two row reductions are log2(LPR)
// shuffle steps inside the lane group, and every access is a 16-byte piece of a contiguous row on both sides.
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int LPR>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
constexpr int SV_RPT = 4;      // rows per lane group and thread: all their loads go out before the first reduction
template <int LPR>
__global__ __launch_bounds__(256) void softmax_vec_fwd_kernel(int B, int L, int C, const float* __restrict__ logits, float* __restrict__ recon) {
  constexpr int RPW = 64 / LPR;                                   // rows per wave and pass
  const long rows = (long)B * L;
  const int lane = threadIdx.x & 63, g = lane / LPR, j = lane - g * LPR;
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (RPW * SV_RPT) + g;
  const bool on = 4 * j < C;
  float4 x[SV_RPT];
#pragma unroll
  for (int r = 0; r < SV_RPT; ++r) {
    const long row = row0 + r * RPW;
    x[r] = (on && row < rows) ? *reinterpret_cast<const float4*>(logits + row * C + 4 * j) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  }
#pragma unroll
  for (int r = 0; r < SV_RPT; ++r) {
    const long row = row0 + r * RPW;
    const float mx = group_max<LPR>(fmaxf(fmaxf(x[r].x, x[r].y), fmaxf(x[r].z, x[r].w)));
    float4 e = make_float4(expf(x[r].x - mx), expf(x[r].y - mx), expf(x[r].z - mx), expf(x[r].w - mx));
    const float inv = 1.f / group_sum<LPR>((e.x + e.y) + (e.z + e.w));
    if (on && row < rows) {
      const int t = (int)(row / B), b = (int)(row - (long)t * B);
      *reinterpret_cast<float4*>(recon + ((long)b * L + t) * C + 4 * j) = make_float4(e.x * inv, e.y * inv, e.z * inv, e.w * inv);
    }
  }
}
template <int LPR>
__global__ __launch_bounds__(256) void softmax_vec_bwd_kernel(int B, int L, int C, const float* __restrict__ recon, const float* __restrict__ drecon,
                                                              bf16_t* __restrict__ dl, long ldd) {
  constexpr int RPW = 64 / LPR;
  const long rows = (long)B * L;
  const int lane = threadIdx.x & 63, g = lane / LPR, j = lane - g * LPR;
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (RPW * SV_RPT) + g;   // time-major row index t * B + b
  const bool on = 4 * j < C;
  float4 p[SV_RPT], d[SV_RPT];
#pragma unroll
  for (int r = 0; r < SV_RPT; ++r) {
    const long row = row0 + r * RPW;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on && row < rows) {
      const int t = (int)(row / B), b = (int)(row - (long)t * B);
      const long src = ((long)b * L + t) * C + 4 * j;
      p[r] = *reinterpret_cast<const float4*>(recon + src); d[r] = *reinterpret_cast<const float4*>(drecon + src);
    } else { p[r] = z; d[r] = z; }
  }
#pragma unroll
  for (int r = 0; r < SV_RPT; ++r) {
    const long row = row0 + r * RPW;
    const float sdot = group_sum<LPR>((p[r].x * d[r].x + p[r].y * d[r].y) + (p[r].z * d[r].z + p[r].w * d[r].w));
    if (on && row < rows) {
      const float v0 = p[r].x * (d[r].x - sdot), v1 = p[r].y * (d[r].y - sdot), v2 = p[r].z * (d[r].z - sdot), v3 = p[r].w * (d[r].w - sdot);
      uint2 w;
      w.x = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16); w.y = (uint32_t)f2bf(v2) | ((uint32_t)f2bf(v3) << 16);
      *reinterpret_cast<uint2*>(dl + row * ldd + 4 * j) = w;            // C % 4 == 0: the row's C columns exactly (no chunk padding needed when C % 8 == 0)
    }
  }
}

// ------------------------------------------------------------------------------------------- ELBO (train.py:31-38)
constexpr int LOSS_BLOCKS = 2048;
// log(p) and log1p(-p) on v_log_f32 (log2, 1 ulp): the libm forms (~25 instructions each, both evaluated for nearly every wave because a
// one-hot row mixes t = 0 and t = 1 lanes) had made this "HBM-bound" kernel ALU-bound: 33 us for 37 MB.  log1p(x) = x log(1 + x) / ((1 + x) - 1)
// keeps full relative accuracy for small x (Kahan); p == 0 / p == 1 give -inf and are clamped at -100 like BCELoss does.
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309f; }
__device__ __forceinline__ float fast_log1m(float p) {               // log(1 - p)
  const float u = 1.f - p, d = u - 1.f;
  return d == 0.f ? -p : fast_log(u) * (-p) * __builtin_amdgcn_rcpf(d);
}
__device__ __forceinline__ float bce_term(float p, float t) {
  // BCELoss clamps both logs at -100.  Hard targets (the one-hot block: every t is 0 or 1) need only ONE of the two logarithms -- same value
  // as the general form (the other term is multiplied by an exact zero).
  if (t == 0.f) return fmaxf(fast_log1m(p), -100.f);
  if (t == 1.f) return fmaxf(fast_log(p), -100.f);
  const float lp = fmaxf(fast_log(p), -100.f), l1p = fmaxf(fast_log1m(p), -100.f);
  return t * lp + (1.f - t) * l1p;
}
// ONE launch: every block writes its (xent, kl) partial sums, takes a ticket, and the block that draws the last ticket adds the partials up in
// a fixed order (bitwise reproducible) and writes the three loss values.  `ticket` lives behind the partials in the caller's workspace: zero
// before the first call, reset to zero by the finishing block (so zero again before the next).  Hand-off across XCDs (per-XCD L2s are not
// coherent): every handed-off value is a device-scope (sc1) store drained with vmcnt(0) before the ticket, and a device-scope load after it.
__global__ __launch_bounds__(256) void bce_kl_fused_kernel(long n, const float* recon, const float* target, long m, const float* mu,
                                                           const float* logvar, float max_len, float* partial, unsigned int* ticket, float* out) {
  __shared__ float red[4];
  __shared__ int s_last;
  float a = 0.f, k = 0.f;
  const long n4 = ((reinterpret_cast<uintptr_t>(recon) | reinterpret_cast<uintptr_t>(target)) & 15) ? 0 : n / 4;    // 16-byte vector part
  const float4* r4 = reinterpret_cast<const float4*>(recon);
  const float4* t4 = reinterpret_cast<const float4*>(target);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 p = r4[i], t = t4[i];
    a -= (bce_term(p.x, t.x) + bce_term(p.y, t.y)) + (bce_term(p.z, t.z) + bce_term(p.w, t.w));
  }
  for (long i = 4 * n4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) a -= bce_term(recon[i], target[i]);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < m; i += (long)gridDim.x * 256) {
    const float u = mu[i], v = logvar[i];
    k += 1.f + u - v * v - expf(u);                                             // mu / logvar swapped, as train.py:36-37
  }
  a = block_sum_256(a, red);
  k = block_sum_256(k, red);
  if (threadIdx.x == 0) {
    // my two partials as device-scope (sc1, write-through) stores, drained before my ticket; no agent-scope FENCE here: on gfx950 that is a
    // write-back of every dirty line of this XCD's L2 (the recon tensor the softmax has just written), once per block -- 69 us instead of 20
    __hip_atomic_store(partial + 2 * blockIdx.x, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(partial + 2 * blockIdx.x + 1, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  float sa = 0.f, sk = 0.f;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) {                     // device-scope (sc1) loads: past this CU's L1 and this XCD's L2
    sa += __hip_atomic_load(partial + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sk += __hip_atomic_load(partial + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  sa = block_sum_256(sa, red);
  sk = block_sum_256(sk, red);
  if (threadIdx.x == 0) {
    const float xent = max_len * (sa / (float)n);
    const float kl = -0.5f * (sk / (float)m);
    out[0] = xent + kl; out[1] = xent; out[2] = kl;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next call (stream order)
  }
}
__global__ __launch_bounds__(256) void bce_kl_bwd_kernel(long n, const float* recon, const float* target, long m, const float* mu,
                                                         const float* logvar, float max_len, const float* grad_out, float* drecon,
                                                         float* dmu, float* dlogvar) {
  const float g = grad_out ? grad_out[0] : 1.f;
  const float sr = g * max_len / (float)n, sl = g * (-0.5f) / (float)m;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float p = recon[i], t = target[i];
    drecon[i] = sr * (p - t) / fmaxf((1.f - p) * p, 1e-12f);
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < m; i += (long)gridDim.x * 256) {
    if (dmu) dmu[i] = sl * (1.f - expf(mu[i]));
    if (dlogvar) dlogvar[i] = sl * (-2.f * logvar[i]);
  }
}

// ------------------------------------------------------------------------------------------- autoregressive sampling step (mosesvae.py:236-253)
// ONE launch per generated token for everything behind the GRU step:  y = decoder_fc(h_top);  p = softmax(y / temp);  w ~ multinomial(p, 1);
// x[~eos, i] = w[~eos];  end_pads[new eos] = i + 1;  eos |= (w == eos)  -- and the NEXT step's layer-0 addend row  add[b] = table[w_b] + base[b]
// (the embedding folded into the input projection, mosesvae.py:239-240).  One wave per batch row; the V x H head sits in LDS once per workgroup.
// Explicit randomness, no hidden state: u(b, i) = hash(seed, i * B + b) / 2^32 (the same counter hash as mvae_dropout_keep); the sampled
// index is the first k with  cumsum_k(p) > u * sum(p)  in the fixed order k = 0 .. V - 1.
template <typename T>
__global__ __launch_bounds__(256) void moses_sample_step_kernel(int B, int V, int H, const T* __restrict__ h, long ldh, const T* __restrict__ wfc, long ldw,
                                                                const float* __restrict__ bias, float inv_temp, uint32_t seed, int step, int eos_id,
                                                                const float* __restrict__ table, int W, const float* __restrict__ base,
                                                                float* __restrict__ add_out, int64_t* __restrict__ x, long x_ld,
                                                                int64_t* __restrict__ end_pads, uint8_t* __restrict__ eos_mask, int64_t* __restrict__ w_out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* ws = reinterpret_cast<T*>(smem_raw);                    // [V][H]
  for (int i = threadIdx.x; i < V * H; i += 256) ws[i] = wfc[(long)(i / H) * ldw + (i % H)];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
    // logits: lane k-strided partial dot products, wave reduction per class (V <= 64 classes: class v ends up in lane v)
    float mine = -INFINITY;
    for (int v = 0; v < V; ++v) {
      float a = 0.f;
      for (int k = lane; k < H; k += 64) a += TT<T>::ld(h + (long)b * ldh + k) * TT<T>::ld(ws + (long)v * H + k);
      a = wave_sum(a);
      if (lane == v) mine = (a + (bias ? bias[v] : 0.f)) * inv_temp;
    }
    const float mx = wave_max(mine);
    const float e = (lane < V) ? __expf(mine - mx) : 0.f;
    // inclusive prefix sum over the classes (fixed order)
    float c = e;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float up = __shfl_up(c, o, 64); if (lane >= o) c += up; }
    const float tot = __shfl(c, 63, 64);
    const uint32_t hsh = drop_hash_u32(seed, (uint32_t)((long)step * B + b));
    const float u = (float)hsh * (1.0f / 4294967296.0f) * tot;
    const unsigned long long above = __ballot(lane < V && c > u);
    int w = above ? (int)__builtin_ctzll(above) : V - 1;     // (u rounds up to tot for hsh near 2^32: take the last class)
    if (lane == 0) {
      const bool done = eos_mask[b] != 0;
      if (!done) {
        x[(long)b * x_ld + step] = w;
        if (w == eos_id) { end_pads[b] = step + 1; eos_mask[b] = 1; }
      }
      w_out[b] = w;
    }
    // next step's layer-0 addend (the finished rows keep being fed their sampled token, as the reference does)
    const float* trow = table + (long)w * W;
    const float* brow = base + (long)b * W;
    float* orow = add_out + (long)b * W;
    for (int cidx = lane * 4; cidx < W; cidx += 256) {
      const float4 tv = *reinterpret_cast<const float4*>(trow + cidx), bv = *reinterpret_cast<const float4*>(brow + cidx);
      *reinterpret_cast<float4*>(orow + cidx) = make_float4(tv.x + bv.x, tv.y + bv.y, tv.z + bv.z, tv.w + bv.w);
    }
  }
}

// ------------------------------------------------------------------------------------------- grad-norm + Adam
constexpr int SUMSQ_CHUNK = 1 << 16;   // elements per partial
__global__ __launch_bounds__(256) void sumsq_kernel(long n, const float* g, float* partial) {
  __shared__ float red[4];
  const long beg = (long)blockIdx.x * SUMSQ_CHUNK;
  long end = beg + SUMSQ_CHUNK; if (end > n) end = n;
  // 16-byte loads, four independent accumulators per thread, fixed combination order (deterministic).  A SHORT last chunk adds its
  // elements in exactly the order (and into exactly the accumulators) the same chunk zero-padded to 64K elements would: the sharded
  // optimiser pads its flat buffer to whole chunks, the unsharded one does not, and both must form the same norm bit for bit.
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const long cnt = end - beg, n4 = cnt / 4, rem = cnt - 4 * n4;
  const float4* g4 = reinterpret_cast<const float4*>(g + beg);          // beg is a multiple of 65536 elements: 16-byte aligned
  if (cnt == SUMSQ_CHUNK) {
    for (long i = threadIdx.x; i < SUMSQ_CHUNK / 4; i += 1024) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float4 v = g4[i + 256 * k]; acc[k] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w); }
    }
  } else {
    for (long i = threadIdx.x; i <= n4; i += 1024) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long idx = i + 256 * k;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < n4) v = g4[idx];
        else if (idx == n4 && rem) { const float* t = g + beg + 4 * n4; v.x = t[0]; if (rem > 1) v.y = t[1]; if (rem > 2) v.z = t[2]; }
        acc[k] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
    }
  }
  float a = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = a;
}
__global__ __launch_bounds__(256) void clip_adam_kernel(long n, float* p, const float* g, float* m, float* v, const float* partial, long npartial,
                                                        float grad_scale, float max_norm, float lr, float b1, float b2, float eps,
                                                        float bc1, float bc2_sqrt, float* norm_out, int norm_out_len, float* poison_reset) {
  __shared__ float red[4];
  __shared__ float s_coef;
  // every block re-derives the global norm from the partials in the same fixed order
  float a = 0.f;
  for (long i = threadIdx.x; i < npartial; i += 256) a += partial[i];
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) {
    const float norm = sqrtf(a) * grad_scale;
    float coef = 1.f;
    if (max_norm > 0.f) { coef = max_norm / (norm + 1e-6f); if (coef > 1.f) coef = 1.f; }
    // a norm that is not finite (a poisoned step: mvae_rnn_*_desc.poison, or genuinely diverged gradients): skip the whole update -- every
    // block takes the same decision from the same partials
    const bool finite = norm < __builtin_huge_valf() && norm == norm;
    s_coef = finite ? coef * grad_scale : __builtin_nanf("");
    if (blockIdx.x == 0) {
      if (norm_out) { norm_out[0] = norm; if (!finite && norm_out_len >= 2) norm_out[1] += 1.f; }
      if (poison_reset) *poison_reset = 0.f;           // mvae_sumsq has read it (stream order); ready for the next step
    }
  }
  __syncthreads();
  const float coef = s_coef;
  if (!(coef == coef)) return;
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * coef;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}

// ------------------------------------------------------------------------------------------- device-side input pipeline
// rows[b] selects a molecule of the uint8 index store [N, L] resident in HBM; emits what MoleLoader.__getitem__ + the default collate
// yield (data_loader.py:26-31): int64 indices [B, L] and the float one-hot [B, L, C].
__global__ __launch_bounds__(256) void expand_indices_kernel(const uint8_t* store, const int64_t* rows, int B, int L, int C, int64_t* idx, float* ohe) {
  const long n = (long)B * L;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int b = (int)(i / L), t = (int)(i - (long)b * L);
    const int v = store[rows[b] * L + t];
    idx[i] = v;
    if (ohe) {
      float* o = ohe + i * C;
      for (int c = 0; c < C; ++c) o[c] = (c == v) ? 1.f : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------- MOSES path (mosesvae.py)
__global__ __launch_bounds__(256) void relu_bwd_kernel(long n, float* dy, const float* y) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dy[i] = y[i] > 0.f ? dy[i] : 0.f;
}
// out[b, t, :] = in[t, b, :]
__global__ __launch_bounds__(256) void permute102_kernel(int T_, int B, int V, const float* in, float* out) {
  const long n = (long)T_ * B * V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int v = (int)(i % V); const long bt = i / V; const int t = (int)(bt % T_); const int b = (int)(bt / T_);
    out[i] = in[((long)t * B + b) * V + v];
  }
}
// z = mu + exp(logvar/2) * eps ; kl = 0.5 * mean_b sum_j (exp(logvar) + mu^2 - 1 - logvar)      (mosesvae.py:158-162)
// one wave per sequence (row): z for the row, the row's KL sum to row_kl[b]; a second tiny launch adds the rows up in a fixed order.  (The
// single 1024-thread block this replaces walked all B * dz elements by itself: 54 us at B = 1024 -- pure latency.)
__global__ __launch_bounds__(256) void moses_latent_fwd_kernel(int B, int dz, const float* mu, const float* logvar, const float* eps, uint32_t seed,
                                                               uint64_t offset, float* eps_out, float* z, float* row_kl) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  float a = 0.f;
  for (int j = lane; j < dz; j += 64) {
    const long i = (long)b * dz + j;
    const float m = mu[i], lv = logvar[i];
    float e;                                                  // mosesvae.py:159 randn_like(mu): injected, or drawn here (see lambda_fwd_kernel)
    if (eps) e = eps[i];
    else { e = normal_draw(seed, offset + (uint64_t)i); eps_out[i] = e; }
    z[i] = m + expf(lv * 0.5f) * e;
    a += expf(lv) + m * m - 1.f - lv;
  }
  a = wave_sum(a);
  if (lane == 0) row_kl[b] = a;
}
__global__ __launch_bounds__(256) void moses_latent_kl_kernel(int B, const float* row_kl, float* kl_out) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) a += row_kl[i];
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) kl_out[0] = 0.5f * a / (float)B;
}
__global__ __launch_bounds__(256) void moses_latent_bwd_kernel(int B, int dz, const float* mu, const float* logvar, const float* eps,
                                                               const float* dzv, const float* dkl, const float* dlogvar_ext, float* dmu, float* dlogvar) {
  const long n = (long)B * dz;
  const float g = dkl ? dkl[0] : 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float m = mu[i], lv = logvar[i], d = dzv ? dzv[i] : 0.f;
    dmu[i] = d + g * m / (float)B;
    dlogvar[i] = d * eps[i] * 0.5f * expf(lv * 0.5f) + g * 0.5f * (expf(lv) - 1.f) / (float)B + (dlogvar_ext ? dlogvar_ext[i] : 0.f);
  }
}
// token cross-entropy of mosesvae.py:193-197: logits row (t*B+b) predicts x[b, t+1]; targets == pad are ignored; mean over the rest.
// a block = 64 consecutive (t, b) rows, 16 per wave; per-BLOCK partial sums (nll, count), added up in a fixed order by the final kernel.
// (One partial per row made the final kernel a 66 us serial walk over T * B * 2 floats by one block.)
constexpr int CE_RPB = 64;
__global__ __launch_bounds__(256) void ce_tb_fwd_kernel(int B, int T_, int V, const float* logits, long ldl, const int64_t* x, int pad,
                                                        float* nll, float* cnt) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long rows = (long)B * T_;
  float out = 0.f, c = 0.f;
  for (int k = 0; k < CE_RPB / 4; ++k) {
    const long row = (long)blockIdx.x * CE_RPB + w * (CE_RPB / 4) + k;
    if (row >= rows) break;                                  // wave-uniform
    const int t = (int)(row / B), b = (int)(row - (long)t * B);
    if (t + 1 >= T_) continue;
    const long tgt = x[(long)b * T_ + t + 1];
    if (tgt == pad) continue;
    const float* l = logits + row * ldl;
    float mx = -INFINITY;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, l[v]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(l[v] - mx);
    s = wave_sum(s);
    out += mx + logf(s) - l[tgt];
    c += 1.f;
  }
  // every lane of a wave holds the same (out, c): one value per wave into the block sum
  const float o4 = block_sum_256(lane == 0 ? out : 0.f, red);
  const float c4 = block_sum_256(lane == 0 ? c : 0.f, red);
  if (threadIdx.x == 0) { nll[blockIdx.x] = o4; cnt[blockIdx.x] = c4; }
}
__global__ __launch_bounds__(256) void ce_tb_final_kernel(long nparts, const float* nll, const float* cnt, float* out2) {
  __shared__ float red[4];
  float a = 0.f, c = 0.f;
  for (long i = threadIdx.x; i < nparts; i += 256) { a += nll[i]; c += cnt[i]; }
  a = block_sum_256(a, red);
  c = block_sum_256(c, red);
  if (threadIdx.x == 0) { out2[0] = a / c; out2[1] = c; }
}
// dlogits[(t*B+b), :] = g * (softmax - onehot(target)) / ntok for counted rows, 0 otherwise (+ optional external dy in [B,T,V] layout)
template <typename T>
__global__ __launch_bounds__(256) void ce_tb_bwd_kernel(int B, int T_, int V, const float* logits, long ldl, const int64_t* x, int pad,
                                                        const float* loss2, const float* g, const float* dy_ext, T* dl, long ldd) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long)B * T_) return;
  const int t = (int)(row / B), b = (int)(row - (long)t * B);
  const float scale = (g ? g[0] : 1.f) / loss2[1];
  bool on = false; long tgt = 0;
  if (t + 1 < T_) { tgt = x[(long)b * T_ + t + 1]; on = tgt != pad; }
  const float* l = logits + row * ldl;
  float mx = -INFINITY, s = 1.f;
  if (on) {
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, l[v]);
    mx = wave_max(mx);
    s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(l[v] - mx);
    s = wave_sum(s);
  }
  for (int v = lane; v < ldd; v += 64) {
    float d = 0.f;
    if (v < V) {
      if (on) d = scale * (expf(l[v] - mx) / s - (v == tgt ? 1.f : 0.f));
      if (dy_ext) d += dy_ext[((long)b * T_ + t) * V + v];
    }
    TT<T>::st(dl + row * ldd + v, d);
  }
}

// rows (t*B + b) with t >= lengths[b] := 0 (16-byte stores; ld in bytes, a multiple of 16)
__global__ __launch_bounds__(256) void mask_rows_tb_kernel(int T_, int B, long ld_bytes, const int* lengths, char* buf) {
  const long cpr = ld_bytes / 16, n = (long)T_ * B * cpr;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long row = i / cpr;
    const int t = (int)(row / B), b = (int)(row - (long)t * B);
    if (t >= lengths[b]) *reinterpret_cast<uint4*>(buf + row * ld_bytes + (i - row * cpr) * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
}

// fp32 one-hot rows in the order of idx itself (row r = flat position in idx): the exact-f32 table gradient dtable = onehot^T . d as a TN problem
__global__ __launch_bounds__(256) void onehot_f32_kernel(const int64_t* idx, long n, int nrows, float* out, long ld) {
  const int gpr = (int)(ld / 4);
  const long tot = n * gpr;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
    const long row = i / gpr; const int c0 = (int)(i - row * gpr) * 4;
    long id = idx[row];
    id = id < 0 ? 0 : (id >= nrows ? nrows - 1 : id);
    const int e = (int)id - c0;
    *reinterpret_cast<float4*>(out + row * ld + c0) = make_float4(e == 0 ? 1.f : 0.f, e == 1 ? 1.f : 0.f, e == 2 ? 1.f : 0.f, e == 3 ? 1.f : 0.f);
  }
}
// one thread per (row, 8-column group): 16-byte stores
__global__ __launch_bounds__(256) void onehot_tb_kernel(const int64_t* idx, int B, int L, int nrows, bf16_t* out, long ld) {
  const int gpr = (int)(ld / 8);
  const long n = (long)B * L * gpr;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long row = i / gpr; const int c0 = (int)(i - row * gpr) * 8;
    const int t = (int)(row / B), b = (int)(row - (long)t * B);
    long id = idx[(long)b * L + t];
    id = id < 0 ? 0 : (id >= nrows ? nrows - 1 : id);
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    const int e = (int)id - c0;
    if (e >= 0 && e < 8) w[e >> 1] = (e & 1) ? 0x3F800000u : 0x00003F80u;      // bf16 1.0 in the low / high half
    *reinterpret_cast<uint4*>(out + row * ld + c0) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// ------------------------------------------------------------------------------------------- extern "C" surface
extern "C" {

int mvae_cast_transpose(int dtype_src, int dtype_dst, int R, int C, const void* src, int64_t lds_, void* dst, int64_t ldd,
                        void* dstT, int64_t ldt, void* stream) {
  return launch_cast_transpose(dtype_src, dtype_dst, R, C, src, lds_, dst, ldd, dstT, ldt, (hipStream_t)stream);
}

int mvae_pack_job_blocks(const mvae_pack_job* j) {
  if (!j || j->R < 1 || j->C < 1) return 0;
  if (j->kind == 1) return (int)(((long)j->R * j->C + 1023) / 1024);
  return ((j->C + 31) / 32) * ((j->R + 31) / 32);
}
int mvae_pack_multi(int njobs, const mvae_pack_job* jobs_device, int total_blocks, void* stream) {
  if (njobs < 1 || !jobs_device || total_blocks < 1) return MVAE_ERR_INVALID;
  hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, njobs, jobs_device);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}

int mvae_permute021(int N, int A, int Bd, const float* in, float* out, void* stream) {
  if (!in || !out || N < 1 || A < 1 || Bd < 1) return MVAE_ERR_INVALID;
  hipLaunchKernelGGL(permute021_kernel, dim3(grid_for((long)N * A * Bd)), dim3(256), 0, (hipStream_t)stream, N, A, Bd, in, out);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}

int mvae_gather_rows_tb(const int64_t* idx, int B, int L, int nrows, const float* table, int W, const float* base, float* out, void* stream) {
  if (!idx || !table || !out || B < 1 || L < 1 || W < 1 || nrows < 1) return MVAE_ERR_INVALID;
  hipLaunchKernelGGL(gather_rows_tb_kernel, dim3(grid_for((long)B * L * W)), dim3(256), 0, (hipStream_t)stream, idx, B, L, nrows, table, W, base, out);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}

static int scatter_parts(int B, int L) { long rows = (long)B * L; long parts = (rows + 63) / 64; if (parts > 1024) parts = 1024; return (int)parts; }
size_t mvae_scatter_rows_tb_workspace(int B, int L, int nrows, int W) { return (size_t)scatter_parts(B, L) * nrows * W * sizeof(float); }
int mvae_scatter_rows_tb(int dtype, const int64_t* idx, int B, int L, int nrows, const void* d, int64_t ldd, int W, float* dtable,
                         void* ws, size_t ws_bytes, void* stream) {
  if (!idx || !d || !dtable || B < 1 || L < 1 || W < 1 || nrows < 1) return MVAE_ERR_INVALID;
  int ws_cols = (48 * 1024) / (nrows * (int)sizeof(float));        // column slice so that the LDS table stays <= 48 KiB
  if (ws_cols < 1) return MVAE_ERR_UNSUPPORTED;
  if (ws_cols > W) ws_cols = W;
  const int slices = (W + ws_cols - 1) / ws_cols;
  const size_t ldsb = (size_t)nrows * ws_cols * sizeof(float);
  const int parts = scatter_parts(B, L);
  if (!ws || ws_bytes < mvae_scatter_rows_tb_workspace(B, L, nrows, W)) return MVAE_ERR_WORKSPACE;
  const long rows = (long)B * L;
  const int rpb = (int)((rows + parts - 1) / parts);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MVAE_F32)
    hipLaunchKernelGGL((scatter_rows_tb_stage1<float>), dim3(parts, slices), dim3(256), ldsb, st, idx, B, L, nrows, (const float*)d, ldd, W, rpb, ws_cols, (float*)ws);
  else if (dtype == MVAE_BF16)
    hipLaunchKernelGGL((scatter_rows_tb_stage1<bf16_t>), dim3(parts, slices), dim3(256), ldsb, st, idx, B, L, nrows, (const bf16_t*)d, ldd, W, rpb, ws_cols, (float*)ws);
  else return MVAE_ERR_INVALID;
  hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)(((long)nrows * W + 63) / 64)), dim3(256), 0, st, (const float*)ws, parts, (long)nrows * W, dtable);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}

int mvae_rowsum(int dtype, int R, int C, const void* X, int64_t ldx, float* out, int accumulate, void* stream) {
  if (!X || !out || R < 1) return MVAE_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MVAE_F32) hipLaunchKernelGGL((rowsum_kernel<float>), dim3((R + 3) / 4), dim3(256), 0, st, R, C, (const float*)X, ldx, out, accumulate);
  else if (dtype == MVAE_BF16) hipLaunchKernelGGL((rowsum_kernel<bf16_t>), dim3((R + 3) / 4), dim3(256), 0, st, R, C, (const bf16_t*)X, ldx, out, accumulate);
  else return MVAE_ERR_INVALID;
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
int mvae_timesum(int dtype, int T, int B, int W, const void* X, float* out, void* stream) {
  if (!X || !out || T < 1 || B < 1 || W < 1) return MVAE_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const long BW = (long)B * W;
  if (dtype == MVAE_F32) hipLaunchKernelGGL((timesum_kernel<float>), dim3(grid_for(BW, 256, 8192)), dim3(256), 0, st, T, BW, (const float*)X, out);
  else if (dtype == MVAE_BF16) {
    if (BW % 8 == 0 && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(out)) & 15) == 0)
      hipLaunchKernelGGL(timesum_bf16x8_kernel, dim3(grid_for(BW / 8, 256, 8192)), dim3(256), 0, st, T, BW, (const bf16_t*)X, out);
    else hipLaunchKernelGGL((timesum_kernel<bf16_t>), dim3(grid_for(BW, 256, 8192)), dim3(256), 0, st, T, BW, (const bf16_t*)X, out);
  } else return MVAE_ERR_INVALID;
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
int mvae_onehot_f32(const int64_t* idx, int64_t n, int nrows, float* out, int64_t ld, void* stream) {
  if (!idx || !out || n < 1 || nrows < 1 || ld < nrows || (ld % 4) || (reinterpret_cast<uintptr_t>(out) & 15)) return MVAE_ERR_INVALID;
  hipLaunchKernelGGL(onehot_f32_kernel, dim3(grid_for((long)n * (ld / 4))), dim3(256), 0, (hipStream_t)stream, idx, (long)n, nrows, out, (long)ld);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
int mvae_onehot_tb(const int64_t* idx, int B, int L, int nrows, void* out, int64_t ld, void* stream) {
  if (!idx || !out || B < 1 || L < 1 || nrows < 1 || ld < nrows || (ld % 8) || (reinterpret_cast<uintptr_t>(out) & 15)) return MVAE_ERR_INVALID;
  hipLaunchKernelGGL(onehot_tb_kernel, dim3(grid_for((long)B * L * (ld / 8))), dim3(256), 0, (hipStream_t)stream, idx, B, L, nrows,
                     reinterpret_cast<bf16_t*>(out), (long)ld);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
size_t mvae_colsum_t_workspace(int M, int N) { return (size_t)colsum_t_chunks(M) * N * sizeof(float); }
int mvae_colsum_t(int dtype, int M, int N, const void* X, int64_t ldx, float* out, void* ws, size_t ws_bytes, void* stream) {
  if (!X || !out || M < 0 || N < 1) return MVAE_ERR_INVALID;
  if ((ldx % 8) || (reinterpret_cast<uintptr_t>(X) & 15)) return MVAE_ERR_INVALID;
  if (!ws || ws_bytes < mvae_colsum_t_workspace(M, N)) return MVAE_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = colsum_t_chunks(M);
  const int rpc = (M + chunks - 1) / chunks;
  dim3 grid((N + 255) / 256, chunks);
  if (dtype == MVAE_F32) hipLaunchKernelGGL((colsum_t_stage1_kernel<float>), grid, dim3(256), 0, st, M, N, (const float*)X, ldx, rpc, (float*)ws);
  else if (dtype == MVAE_BF16) hipLaunchKernelGGL((colsum_t_stage1_kernel<bf16_t>), grid, dim3(256), 0, st, M, N, (const bf16_t*)X, ldx, rpc, (float*)ws);
  else return MVAE_ERR_INVALID;
  hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, st, (const float*)ws, chunks, (long)N, out);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
size_t mvae_colsum_workspace(int M, int N) { return colsum_workspace_bytes(M, N); }
int mvae_colsum(int M, int N, const float* X, int64_t ldx, float* out, void* ws, size_t ws_bytes, void* stream) {
  if (!X || !out || M < 0) return MVAE_ERR_INVALID;
  return launch_colsum(M, N, X, ldx, out, ws, ws_bytes, (hipStream_t)stream);
}
int mvae_expand_indices(const uint8_t* store, const int64_t* rows, int B, int L, int C, int64_t* idx, float* ohe, void* stream) {
  if (!store || !rows || !idx || B < 1 || L < 1 || C < 1) return MVAE_ERR_INVALID;
  hipLaunchKernelGGL(expand_indices_kernel, dim3(grid_for((long)B * L)), dim3(256), 0, (hipStream_t)stream, store, rows, B, L, C, idx, ohe);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
int mvae_relu_bwd(int64_t n, float* dy, const float* y, void* stream) {
  if (!dy || !y) return MVAE_ERR_INVALID;
  if (n <= 0) return MVAE_OK;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (long)n, dy, y);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
int mvae_permute102(int T, int B, int V, const float* in, float* out, void* stream) {
  if (!in || !out || T < 1 || B < 1 || V < 1) return MVAE_ERR_INVALID;
  hipLaunchKernelGGL(permute102_kernel, dim3(grid_for((long)T * B * V)), dim3(256), 0, (hipStream_t)stream, T, B, V, in, out);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
size_t mvae_moses_latent_workspace(int B) { return (size_t)(B > 0 ? B : 0) * sizeof(float); }
int mvae_moses_latent_fwd(int B, int dz, const float* mu, const float* logvar, const float* eps, uint32_t seed, uint64_t offset, float* eps_out,
                          float* z, float* kl_out, void* ws, size_t ws_bytes, void* stream) {
  if (!mu || !logvar || (!eps && !eps_out) || !z || !kl_out || B < 1 || dz < 1) return MVAE_ERR_INVALID;
  if (!ws || ws_bytes < mvae_moses_latent_workspace(B)) return MVAE_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(moses_latent_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, st, B, dz, mu, logvar, eps, seed, offset, eps_out, z, (float*)ws);
  hipLaunchKernelGGL(moses_latent_kl_kernel, dim3(1), dim3(256), 0, st, B, (const float*)ws, kl_out);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
int mvae_moses_latent_bwd(int B, int dz, const float* mu, const float* logvar, const float* eps, const float* dzv, const float* dkl,
                          const float* dlogvar_ext, float* dmu, float* dlogvar, void* stream) {
  if (!mu || !logvar || !eps || !dmu || !dlogvar || B < 1 || dz < 1) return MVAE_ERR_INVALID;
  hipLaunchKernelGGL(moses_latent_bwd_kernel, dim3(grid_for((long)B * dz)), dim3(256), 0, (hipStream_t)stream, B, dz, mu, logvar, eps, dzv, dkl,
                     dlogvar_ext, dmu, dlogvar);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
size_t mvae_ce_loss_workspace(int B, int T) { return (size_t)B * T * 2 * sizeof(float); }
int mvae_ce_loss_fwd(int B, int T, int V, const float* logits, int64_t ldl, const int64_t* x, int pad, float* loss2, void* ws, size_t ws_bytes,
                     void* stream) {
  if (!logits || !x || !loss2 || B < 1 || T < 1 || V < 1) return MVAE_ERR_INVALID;
  if (!ws || ws_bytes < mvae_ce_loss_workspace(B, T)) return MVAE_ERR_WORKSPACE;
  const long rows = (long)B * T, nparts = (rows + CE_RPB - 1) / CE_RPB;
  float* nll = (float*)ws; float* cnt = nll + nparts;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_tb_fwd_kernel, dim3((unsigned)nparts), dim3(256), 0, st, B, T, V, logits, (long)ldl, x, pad, nll, cnt);
  hipLaunchKernelGGL(ce_tb_final_kernel, dim3(1), dim3(256), 0, st, nparts, (const float*)nll, (const float*)cnt, loss2);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
int mvae_ce_loss_bwd(int dtype, int B, int T, int V, const float* logits, int64_t ldl, const int64_t* x, int pad, const float* loss2,
                     const float* grad_out, const float* dy_ext, void* dl, int64_t ldd, void* stream) {
  if (!logits || !x || !loss2 || !dl || B < 1 || T < 1 || V < 1 || ldd < V) return MVAE_ERR_INVALID;
  const long rows = (long)B * T;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MVAE_F32)
    hipLaunchKernelGGL((ce_tb_bwd_kernel<float>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, B, T, V, logits, (long)ldl, x, pad, loss2, grad_out, dy_ext, (float*)dl, (long)ldd);
  else if (dtype == MVAE_BF16)
    hipLaunchKernelGGL((ce_tb_bwd_kernel<bf16_t>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, B, T, V, logits, (long)ldl, x, pad, loss2, grad_out, dy_ext, (bf16_t*)dl, (long)ldd);
  else return MVAE_ERR_INVALID;
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
int mvae_mask_rows_tb(int dtype, int T, int B, int64_t ld, const int32_t* lengths, void* buf, void* stream) {
  if (!lengths || !buf || T < 1 || B < 1 || ld < 1 || (dtype != MVAE_F32 && dtype != MVAE_BF16)) return MVAE_ERR_INVALID;
  const long ldb = (long)ld * (dtype == MVAE_BF16 ? 2 : 4);
  if ((ldb % 16) || (reinterpret_cast<uintptr_t>(buf) & 15)) return MVAE_ERR_INVALID;
  hipLaunchKernelGGL(mask_rows_tb_kernel, dim3(grid_for((long)T * B * (ldb / 16))), dim3(256), 0, (hipStream_t)stream, T, B, ldb, lengths, (char*)buf);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
int mvae_selu_bwd(int64_t n, float* dy, const float* y, void* stream) {
  if (!dy || !y) return MVAE_ERR_INVALID;
  return launch_selu_bwd(n, dy, y, (hipStream_t)stream);
}

int mvae_lambda_fwd(int B, int o, const float* mulv, const float* eps, float scale, uint32_t seed, uint64_t offset, float* eps_out, float* z,
                    float* mu, float* logvar, void* stream) {
  if (!mulv || (!eps && !eps_out) || !z || !mu || !logvar || B < 1 || o < 1) return MVAE_ERR_INVALID;
  hipLaunchKernelGGL(lambda_fwd_kernel, dim3(grid_for((long)B * o)), dim3(256), 0, (hipStream_t)stream, B, o, mulv, eps, scale, seed, offset, eps_out,
                     z, mu, logvar);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
int mvae_normal_fill(int64_t n, float scale, uint32_t seed, uint64_t offset, float* out, void* stream) {
  if (!out || n < 1) return MVAE_ERR_INVALID;
  hipLaunchKernelGGL(normal_fill_kernel, dim3(grid_for((long)n)), dim3(256), 0, (hipStream_t)stream, (long)n, scale, seed, offset, out);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
void mvae_normal_words(uint32_t seed, uint64_t counter, uint32_t* words2) {
  if (words2) normal_words(seed, counter, words2[0], words2[1]);
}
int mvae_lambda_bwd(int B, int o, const float* mulv, const float* eps, const float* dz, const float* dmu, const float* dlogvar,
                    float* dmulv, void* stream) {
  if (!mulv || !eps || !dmulv || B < 1 || o < 1) return MVAE_ERR_INVALID;
  hipLaunchKernelGGL(lambda_bwd_kernel, dim3(grid_for((long)B * o)), dim3(256), 0, (hipStream_t)stream, B, o, mulv, eps, dz, dmu, dlogvar, dmulv);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}

static bool softmax_tiled_ok() { const char* e = mvae_knob("MVAE_SOFTMAX_TILED"); return !e || atoi(e) != 0; }
int mvae_softmax_tb_fwd(int B, int L, int C, const float* logits, int64_t ldl, float* recon, void* stream) {
  if (!logits || !recon || B < 1 || L < 1 || C < 1) return MVAE_ERR_INVALID;
  const long rows = (long)B * L;
  if (C > 64 * SM_CPL) return MVAE_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (ldl == C && softmax_tiled_ok() && (C % 8) == 0 && C >= 16 && ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(recon)) & 15) == 0) {
    // rows are whole 16-byte pieces: the register-only form
    const int lpr = C <= 32 ? 8 : C <= 64 ? 16 : 32, rpb = 4 * (64 / lpr) * SV_RPT;
    const dim3 grid((unsigned)((rows + rpb - 1) / rpb));
    if (lpr == 8) hipLaunchKernelGGL((softmax_vec_fwd_kernel<8>), grid, dim3(256), 0, st, B, L, C, logits, recon);
    else if (lpr == 16) hipLaunchKernelGGL((softmax_vec_fwd_kernel<16>), grid, dim3(256), 0, st, B, L, C, logits, recon);
    else hipLaunchKernelGGL((softmax_vec_fwd_kernel<32>), grid, dim3(256), 0, st, B, L, C, logits, recon);
    MVAE_CHECK_HIP(hipGetLastError());
    return MVAE_OK;
  }
  if (ldl == C && softmax_tiled_ok() && ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(recon)) & 15) == 0) {
    // tiled form: densely packed logits (what MolDecoder produces); 16 x 16 tiles (256 threads) up to 40 classes, 8 x 16 beyond (LDS)
    static bool attr_set = false;
    if (!attr_set) {
      MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(softmax_tile_fwd_kernel<16, 16, 40>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(softmax_tile_fwd_kernel<8, 16, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(softmax_tile_fwd_kernel<8, 16, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set = true;
    }
    const int CP = C | 1;
    if (C <= 40) {
      const unsigned grid = (unsigned)(((B + 15) / 16) * ((L + 15) / 16));
      hipLaunchKernelGGL((softmax_tile_fwd_kernel<16, 16, 40>), dim3(grid), dim3(256), (size_t)256 * CP * sizeof(float), st, B, L, C, logits, recon);
    } else {
      const unsigned grid = (unsigned)(((B + 15) / 16) * ((L + 7) / 8));
      if (C <= 64) hipLaunchKernelGGL((softmax_tile_fwd_kernel<8, 16, 64>), dim3(grid), dim3(128), (size_t)128 * CP * sizeof(float), st, B, L, C, logits, recon);
      else hipLaunchKernelGGL((softmax_tile_fwd_kernel<8, 16, 128>), dim3(grid), dim3(128), (size_t)128 * CP * sizeof(float), st, B, L, C, logits, recon);
    }
    MVAE_CHECK_HIP(hipGetLastError());
    return MVAE_OK;
  }
  hipLaunchKernelGGL(softmax_tb_fwd_kernel, dim3((unsigned)((rows + 4 * SM_ROWS - 1) / (4 * SM_ROWS))), dim3(256), 0, st, B, L, C, logits, ldl, recon);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
int mvae_softmax_tb_bwd(int dtype, int B, int L, int C, const float* recon, const float* drecon, void* dl, int64_t ldd, void* dlT,
                        int64_t ldT, void* stream) {
  if (!recon || !drecon || !dl || B < 1 || L < 1 || C < 1 || ldd < C) return MVAE_ERR_INVALID;
  const long rows = (long)B * L;
  hipStream_t st = (hipStream_t)stream;
  if (C > 64 * SM_CPL || ldd > 64 * SM_CPL) return MVAE_ERR_UNSUPPORTED;
  if (dtype == MVAE_BF16 && !dlT && softmax_tiled_ok() && (C % 8) == 0 && C >= 16 && (ldd % 8) == 0 &&
      ((reinterpret_cast<uintptr_t>(recon) | reinterpret_cast<uintptr_t>(drecon) | reinterpret_cast<uintptr_t>(dl)) & 15) == 0) {
    const int lpr = C <= 32 ? 8 : C <= 64 ? 16 : 32, rpb = 4 * (64 / lpr) * SV_RPT;
    const dim3 grid((unsigned)((rows + rpb - 1) / rpb));
    if (lpr == 8) hipLaunchKernelGGL((softmax_vec_bwd_kernel<8>), grid, dim3(256), 0, st, B, L, C, recon, drecon, (bf16_t*)dl, (long)ldd);
    else if (lpr == 16) hipLaunchKernelGGL((softmax_vec_bwd_kernel<16>), grid, dim3(256), 0, st, B, L, C, recon, drecon, (bf16_t*)dl, (long)ldd);
    else hipLaunchKernelGGL((softmax_vec_bwd_kernel<32>), grid, dim3(256), 0, st, B, L, C, recon, drecon, (bf16_t*)dl, (long)ldd);
    MVAE_CHECK_HIP(hipGetLastError());
    return MVAE_OK;
  }
  if (dtype == MVAE_BF16 && !dlT && softmax_tiled_ok() && (ldd % 8) == 0 && ldd >= ((C + 7) & ~7) &&
      ((reinterpret_cast<uintptr_t>(recon) | reinterpret_cast<uintptr_t>(drecon) | reinterpret_cast<uintptr_t>(dl)) & 15) == 0) {
    static bool attr_set = false;
    if (!attr_set) {
      MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(softmax_tile_bwd_kernel<16, 16, 40>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(softmax_tile_bwd_kernel<8, 16, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(softmax_tile_bwd_kernel<8, 16, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set = true;
    }
    const int CP = C | 1;
    if (C <= 40) {
      const unsigned grid = (unsigned)(((B + 15) / 16) * ((L + 15) / 16));
      hipLaunchKernelGGL((softmax_tile_bwd_kernel<16, 16, 40>), dim3(grid), dim3(256), (size_t)2 * 256 * CP * sizeof(float), st, B, L, C, recon, drecon, (bf16_t*)dl, (long)ldd);
    } else {
      const unsigned grid = (unsigned)(((B + 15) / 16) * ((L + 7) / 8));
      if (C <= 64) hipLaunchKernelGGL((softmax_tile_bwd_kernel<8, 16, 64>), dim3(grid), dim3(128), (size_t)2 * 128 * CP * sizeof(float), st, B, L, C, recon, drecon, (bf16_t*)dl, (long)ldd);
      else hipLaunchKernelGGL((softmax_tile_bwd_kernel<8, 16, 128>), dim3(grid), dim3(128), (size_t)2 * 128 * CP * sizeof(float), st, B, L, C, recon, drecon, (bf16_t*)dl, (long)ldd);
    }
    MVAE_CHECK_HIP(hipGetLastError());
    return MVAE_OK;
  }
  const dim3 sgrid((unsigned)((rows + 4 * SM_ROWS - 1) / (4 * SM_ROWS)));
  if (dtype == MVAE_F32)
    hipLaunchKernelGGL((softmax_tb_bwd_kernel<float>), sgrid, dim3(256), 0, st, B, L, C, recon, drecon, (float*)dl, ldd, (float*)dlT, ldT);
  else if (dtype == MVAE_BF16)
    hipLaunchKernelGGL((softmax_tb_bwd_kernel<bf16_t>), sgrid, dim3(256), 0, st, B, L, C, recon, drecon, (bf16_t*)dl, ldd, (bf16_t*)dlT, ldT);
  else return MVAE_ERR_INVALID;
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}

size_t mvae_bce_kl_loss_workspace(int64_t, int64_t) { return (size_t)LOSS_BLOCKS * 2 * sizeof(float) + 16; }
int mvae_bce_kl_loss_fwd(int64_t n, const float* recon, const float* target, int64_t m, const float* mu, const float* logvar,
                         float max_len, float* loss_out, void* ws, size_t ws_bytes, void* stream) {
  if (!recon || !target || !mu || !logvar || !loss_out || n < 1 || m < 1) return MVAE_ERR_INVALID;
  if (!ws || ws_bytes < mvae_bce_kl_loss_workspace(n, m) || (reinterpret_cast<uintptr_t>(ws) & 3)) return MVAE_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  // enough blocks to keep the HBM queues full (>= 2 per CU), few enough that the tickets -- atomics on ONE address, which the memory side
  // serialises at ~10 ns each -- and the finishing block's pass over the partials stay short: 8 16-byte pieces per thread, at most 1024 blocks
  long want = (n / 4 + 256 * 8 - 1) / (256 * 8);
  const int blocks = (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
  float* partial = reinterpret_cast<float*>(ws);
  unsigned int* ticket = reinterpret_cast<unsigned int*>(partial + 2 * LOSS_BLOCKS);
  hipLaunchKernelGGL(bce_kl_fused_kernel, dim3(blocks), dim3(256), 0, st, (long)n, recon, target, (long)m, mu, logvar, max_len, partial, ticket, loss_out);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
int mvae_bce_kl_loss_bwd(int64_t n, const float* recon, const float* target, int64_t m, const float* mu, const float* logvar,
                         float max_len, const float* grad_out, float* drecon, float* dmu, float* dlogvar, void* stream) {
  if (!recon || !target || !mu || !logvar || !drecon || n < 1 || m < 1) return MVAE_ERR_INVALID;
  hipLaunchKernelGGL(bce_kl_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (long)n, recon, target, (long)m, mu, logvar, max_len,
                     grad_out, drecon, dmu, dlogvar);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}

int mvae_moses_sample_step(int dtype, int B, int V, int H, const void* h_top, int64_t ldh, const void* w_fc, int64_t ldw, const float* bias, float temp,
                           uint32_t seed, int step, int eos_id, const float* table, int W, const float* base, float* add_out, int64_t* x, int64_t x_ld,
                           int64_t* end_pads, uint8_t* eos_mask, int64_t* w_out, void* stream) {
  if (!h_top || !w_fc || !table || !base || !add_out || !x || !end_pads || !eos_mask || !w_out) return MVAE_ERR_INVALID;
  if (B < 1 || V < 1 || V > 64 || H < 1 || (W & 3) || W < 4 || !(temp > 0.f) || step < 0) return MVAE_ERR_INVALID;
  const size_t lds = (size_t)V * H * (dtype == MVAE_BF16 ? 2 : 4);
  if (lds > 160 * 1024) return MVAE_ERR_UNSUPPORTED;                 // the head must fit the CU's LDS (fp32, H = 512: V <= 80; bf16: V <= 64 by the check above)
  if (lds > 64 * 1024) {                                              // above the default dynamic-LDS limit: opt in (per device, like the persistent kernels)
    static std::atomic<bool> attr[64];
    int dev_id = 0;
    MVAE_CHECK_HIP(hipGetDevice(&dev_id));
    if (dev_id < 0 || dev_id >= 64 || !attr[dev_id].load(std::memory_order_acquire)) {
      MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(moses_sample_step_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(moses_sample_step_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      if (dev_id >= 0 && dev_id < 64) attr[dev_id].store(true, std::memory_order_release);
    }
  }
  int blocks = (B + 3) / 4; if (blocks > 1024) blocks = 1024;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MVAE_BF16)
    hipLaunchKernelGGL((moses_sample_step_kernel<bf16_t>), dim3(blocks), dim3(256), lds, st, B, V, H, (const bf16_t*)h_top, (long)ldh, (const bf16_t*)w_fc, (long)ldw,
                       bias, 1.f / temp, seed, step, eos_id, table, W, base, add_out, x, (long)x_ld, end_pads, eos_mask, w_out);
  else if (dtype == MVAE_F32)
    hipLaunchKernelGGL((moses_sample_step_kernel<float>), dim3(blocks), dim3(256), lds, st, B, V, H, (const float*)h_top, (long)ldh, (const float*)w_fc, (long)ldw,
                       bias, 1.f / temp, seed, step, eos_id, table, W, base, add_out, x, (long)x_ld, end_pads, eos_mask, w_out);
  else return MVAE_ERR_INVALID;
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}

size_t mvae_sumsq_workspace(int64_t n) { return (size_t)((n + SUMSQ_CHUNK - 1) / SUMSQ_CHUNK) * sizeof(float); }
int mvae_sumsq(int64_t n, const float* g, float* partial, void* stream) {
  if (!g || !partial || n < 1 || (reinterpret_cast<uintptr_t>(g) & 15)) return MVAE_ERR_INVALID;
  const int blocks = (int)((n + SUMSQ_CHUNK - 1) / SUMSQ_CHUNK);
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (long)n, g, partial);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
int mvae_clip_adam(int64_t n, float* p, const float* g, float* m, float* v, const float* partial, int64_t npartial, float grad_scale,
                   float max_norm, float lr, float beta1, float beta2, float eps, int step, float* norm_out, int norm_out_len, float* poison_reset,
                   void* stream) {
  if (!p || !g || !m || !v || !partial || n < 1 || npartial < 1 || step < 1) return MVAE_ERR_INVALID;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(clip_adam_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, (long)n, p, g, m, v, partial,
                     (long)npartial, grad_scale, max_norm, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), norm_out, norm_out_len, poison_reset);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}

}  // extern "C"
