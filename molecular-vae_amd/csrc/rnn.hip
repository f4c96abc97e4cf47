// Recurrent stack on gfx950: layer-wavefront schedule, one launch per wavefront step.
// Forward cell update  = MFMA tile loop over two K-segments (x_t . W_ih^T, h_{t-1} . W_hh^T) whose output tile
// holds the 4 gate pre-activations of the same (batch row, hidden unit), so the gate non-linearities and the
// state update run in the epilogue (models.py:128,164 nn.LSTM).
// Backward cell update = tile loop over (dG_{t+1} . W_hh, dG^{l+1}_t . W_ih^{l+1}) + gate derivative epilogue.
// Epilogues go through LDS: accumulators are re-laid out so that each thread owns 8 consecutive hidden units of
// one batch row and every global access of the saved state (gates, c, h, dG, dc) is a 16-byte coalesced vector.
#include "tile_pipe.hpp"
#include "kernels.hpp"
#include <stdlib.h>

// ---------------------------------------------------------------------------------------------- 8-wide vector I/O
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __builtin_bit_cast(float, w[i] << 16); v[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u); }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
  uint4 u;
  u.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16); u.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
  u.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16); u.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
  *reinterpret_cast<uint4*>(p) = u;
}
// n valid elements (n == 8 and `vec` -> one/two 16-byte accesses; otherwise element-wise)
template <typename T> __device__ __forceinline__ void ldn(const T* p, float (&v)[8], int n, bool vec) {
  if (vec && n == 8) { load8<T>(p, v); return; }
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (e < n) ? TT<T>::ld(p + e) : 0.f;
}
template <typename T> __device__ __forceinline__ void stn(T* p, const float (&v)[8], int n, bool vec) {
  if (vec && n == 8) { store8<T>(p, v); return; }
#pragma unroll
  for (int e = 0; e < 8; ++e) if (e < n) TT<T>::st(p + e, v[e]);
}

// gate non-linearities: exact-ish libm forms for the f32 path, v_exp_f32-based forms for the bf16 path
template <typename T> __device__ __forceinline__ float act_sigmoid(float x);
template <typename T> __device__ __forceinline__ float act_tanh(float x);
template <> __device__ __forceinline__ float act_sigmoid<float>(float x) { return 1.f / (1.f + expf(-x)); }
template <> __device__ __forceinline__ float act_tanh<float>(float x) { return tanhf(x); }
// bf16 storage mode: ONE v_rcp_f32 per non-linearity (1 ulp).  `a / b` and __fdividef compile to the IEEE division expansion
// (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup: ~10 VALU instructions) -- five of those per hidden unit made the
// gate epilogues VALU-bound.
template <> __device__ __forceinline__ float act_sigmoid<bf16_t>(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
template <> __device__ __forceinline__ float act_tanh<bf16_t>(float x) { return fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)), 1.f); }

__device__ __forceinline__ int xcd_remap_r(int bid, int n) {   // contiguous tile ranges per XCD (see gemm.hip)
  const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// ---------------------------------------------------------------------------------------------- inter-layer dropout
// keep / (1 - p) factor of element idx (see mvae_dropout_keep in include/mvae.h): injected byte mask when given, else the hash
struct DropArgs { float scale; uint32_t thresh, seed; };     // scale = 1 / (1 - p); thresh = (uint32)(p * 2^32)
__host__ __device__ __forceinline__ uint32_t drop_hash(uint32_t seed, uint32_t idx) {
  uint32_t h = idx * 0x9E3779B1u ^ seed;
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ float drop_factor(const DropArgs& da, const uint8_t* mask, uint32_t idx0, long off) {
  const bool keep = mask ? (mask[off] != 0) : (drop_hash(da.seed, idx0 + (uint32_t)off) >= da.thresh);
  return keep ? da.scale : 0.f;
}

// ---------------------------------------------------------------------------------------------- forward
struct StepTaskF {
  const void *A0, *A1, *W0, *W1;
  long lda0, lda1, ldw0, ldw1;
  int K0, K1;
  const float* add; long add_ld;      // fp32 row-wise pre-activation addend (layer 0) or nullptr
  const float* add_tbl;               // fp32 [rows][4H] table: row add_idx[b * add_idx_ld] is added on top (embedding folded into W_ih), or nullptr
  const int64_t* add_idx; long add_idx_ld; int add_rows;
  const float* bias;                  // fp32 [4H] or nullptr
  const float* c_prev; float* c_out;  // fp32 recurrent cell state (ping-pong)
  void* c_save;                       // [B,H] dtype: cell state saved for backward
  void* h_out; long ldh;
  void* g_out;                        // [B,4H] dtype: post-activation gates saved for backward
  const void* hprev_t0;               // GRU, t == 0 only: initial hidden state [B, ld = lda1] dtype (nullptr = zeros)
  int t;                              // time index of this cell (GRU length masking)
  void* h_drop;                       // [B, ldh] dtype: h * keep / (1 - p) for the layer above (nullptr: no dropout on this cell's output)
  const uint8_t* dmask;               // [B, H] injected keep mask of this cell (nullptr: hash)
  uint32_t didx0;                     // hash counter of element (b = 0, j = 0) of this cell
};
struct StepArgsF { StepTaskF t[MVAE_MAX_LAYERS]; const int* lengths; int ntask, B, H, tiles_m, tiles_j, dbg, vec, cell, gru3; DropArgs drop; };

// NBUF > 0: LDS-direct ring of that depth; NBUF == 0: generic register-staged loop (any shape)
// WS: 512 threads, waves 4-7 only load (tile_gemm_ws); everyone takes part in the epilogue
template <typename T, int BM, int BJ, int NBUF, bool WS = false>
__global__ __launch_bounds__(WS ? 512 : 256) void lstm_step_fwd_kernel(StepArgsF p) {
  constexpr bool PIPE = NBUF > 0;
  constexpr int NTHR = WS ? 512 : 256;
  static_assert(!WS || NBUF > 0, "wave specialisation needs the LDS-direct ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(BJ == 32 || BJ == 64, "hidden units per tile");
  constexpr int JS = BJ / 32;                  // 16-wide hidden sub-tiles per wave and gate
  constexpr int BN = 4 * BJ, WM = BM / 2, MI = WM / 16, NI = 4 * JS;     // acc[mi][g * JS + s]
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, wm = wave >> 1, wj = wave & 1;
  const int per_task = p.tiles_m * p.tiles_j;
  const int bid = xcd_remap_r(blockIdx.x, p.ntask * per_task);
  const int task = bid / per_task;
  const int rem = bid - task * per_task;
  const int tj = rem / p.tiles_m, tm = rem - tj * p.tiles_m;   // tiles sharing a weight panel are neighbours on one XCD
  const StepTaskF& q = p.t[task];
  const int m0 = tm * BM, j0 = tj * BJ, H = p.H, B = p.B;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int g = 0; g < NI; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifdef MVAE_TUNING
  const bool run_main = !(p.dbg & 2);          // diagnostic build: MVAE_DBG bit1 = epilogue only
#else
  constexpr bool run_main = true;
#endif
  if (run_main) {
  if constexpr (PIPE) {
    PipeSeg<BM, BN> s0, s1;
    const uint32_t sz = (uint32_t)sizeof(T);
    // GRU gate slots are [r | z | W_in x | W_hn h]: the x segment has no rows for slot 3, the h segment none for slot 2 (zero blocks in
    // the packed weights).  Those tile rows are given out-of-range offsets: the descriptor bound turns them into zeros without a fetch --
    // a quarter of the weight stream of a GRU step.
    const int skip0 = p.gru3 ? 3 : -1, skip1 = p.gru3 ? 2 : -1;
    {
      const uint32_t lda = (uint32_t)q.lda0 * sz, ldw = (uint32_t)q.ldw0 * sz;
      auto offA = [&](int r) -> uint32_t { const int gm = m0 + r; return gm < B ? (uint32_t)gm * lda : PIPE_OOB; };
      auto offB = [&](int r) -> uint32_t { const int g = r / BJ, j = j0 + (r - g * BJ); return (j < H && g != skip0) ? (uint32_t)(g * H + j) * ldw : PIPE_OOB; };
      pipe_seg_init<T, BM, BN>(s0, q.A0, (uint32_t)B * lda, q.W0, (uint32_t)(4 * H) * ldw, offA, offB, q.K0, tid & 255);
    }
    {
      const uint32_t lda = (uint32_t)q.lda1 * sz, ldw = (uint32_t)q.ldw1 * sz;
      auto offA = [&](int r) -> uint32_t { const int gm = m0 + r; return gm < B ? (uint32_t)gm * lda : PIPE_OOB; };
      auto offB = [&](int r) -> uint32_t { const int g = r / BJ, j = j0 + (r - g * BJ); return (j < H && g != skip1) ? (uint32_t)(g * H + j) * ldw : PIPE_OOB; };
      pipe_seg_init<T, BM, BN>(s1, q.A1, (uint32_t)B * lda, q.W1, (uint32_t)(4 * H) * ldw, offA, offB, q.K1, tid & 255);
    }
    if constexpr (WS) tile_gemm_ws<T, BM, BN, MI, NI, (NBUF > 0 ? NBUF : 3), JS, BJ>(smem, s0, s1, wm * WM, wj * (BJ / 2), acc, tid);
    else tile_gemm_pipe<T, BM, BN, MI, NI, (NBUF > 0 ? NBUF : 3), JS, BJ>(smem, s0, s1, wm * WM, wj * (BJ / 2), acc, tid);
  } else {
    int brow[NI];
#pragma unroll
    for (int g = 0; g < NI; ++g) brow[g] = (g / JS) * BJ + wj * (BJ / 2) + (g % JS) * 16;
    if (q.A0 != nullptr) {
      const T* A = reinterpret_cast<const T*>(q.A0);
      const T* W = reinterpret_cast<const T*>(q.W0);
      const long lda = q.lda0, ldw = q.ldw0;
      auto rowA = [&](int r) -> const T* { const int gm = m0 + r; return gm < B ? A + (long)gm * lda : nullptr; };
      auto rowB = [&](int r) -> const T* { const int g = r / BJ, j = j0 + (r - g * BJ); return (j < H && !(p.gru3 && g == 3)) ? W + ((long)g * H + j) * ldw : nullptr; };
      tile_gemm_segment<T, BM, BN, MI, NI>(smem, rowA, rowB, 0, q.K0, wm * WM, brow, acc, tid);
    }
    if (q.A1 != nullptr) {
      const T* A = reinterpret_cast<const T*>(q.A1);
      const T* W = reinterpret_cast<const T*>(q.W1);
      const long lda = q.lda1, ldw = q.ldw1;
      auto rowA = [&](int r) -> const T* { const int gm = m0 + r; return gm < B ? A + (long)gm * lda : nullptr; };
      auto rowB = [&](int r) -> const T* { const int g = r / BJ, j = j0 + (r - g * BJ); return (j < H && !(p.gru3 && g == 2)) ? W + ((long)g * H + j) * ldw : nullptr; };
      tile_gemm_segment<T, BM, BN, MI, NI>(smem, rowA, rowB, 0, q.K1, wm * WM, brow, acc, tid);
    }
  }
  }
#ifdef MVAE_TUNING       // diagnostic build only (build.sh tune -> libmvae_hip_tune.so): main loop without the epilogue
  if (p.dbg & 1) { if (acc[0][0][0] == 12345.678f) q.c_out[0] = 1.f; return; }
#endif

  // ---- epilogue: accumulators -> LDS [row][gate][BJ + 4] (fp32), then one thread per (row, 8 hidden units).
  // Everything the gate math reads from global memory (bias, previous state) is requested BEFORE the staging writes and the
  // barrier, for all of the thread's rows at once, so one load latency is exposed instead of one per row.
  constexpr int SJ = BJ + 4;
  constexpr int GPRW = BJ / 8;                 // 8-unit groups per tile row
  constexpr int NIT = (BM * GPRW + NTHR - 1) / NTHR;
  float* stg = reinterpret_cast<float*>(smem);
  const int g8 = tid % GPRW;                   // constant per thread
  const int j8 = j0 + g8 * 8;
  const int n = (H - j8 < 8) ? (H - j8) : 8;   // <= 0: this thread's units are outside the layer
  const bool vec = p.vec != 0;
  // FAST = vector path, whole 8-unit groups, batch tile entirely inside the batch: straight-line loads and stores.  Every `if (ptr) load`
  // (and every ldn / stn with its own vec test) is a branch, and at each join the compiler waits for all loads in flight -- four dependent
  // round trips for the bias and one per row for the previous state; in the fast form they all go out together: an absent bias is read
  // from c_out (valid fp32 memory, at least 4H long) and replaced by zero, the state loads sit under ONE uniform branch.
  const bool fast = vec && n == 8 && m0 + BM <= B && (BM * GPRW) % NTHR == 0;
  float bias[4][8], cpv[NIT][8];
  if (fast) {
    const float* bsrc = q.bias ? q.bias : q.c_out;
#pragma unroll
    for (int g = 0; g < 4; ++g) load8<float>(bsrc + g * H + j8, bias[g]);
    if (q.c_prev) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) load8<float>(q.c_prev + (long)(m0 + tid / GPRW + it * (NTHR / GPRW)) * H + j8, cpv[it]);
    } else if (q.hprev_t0) {
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        load8<T>(reinterpret_cast<const T*>(q.hprev_t0) + (long)(m0 + tid / GPRW + it * (NTHR / GPRW)) * q.lda1 + j8, cpv[it]);
    } else {
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) cpv[it][e] = 0.f;
    }
    if (!q.bias) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) bias[g][e] = 0.f;
    }
  } else {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (q.bias && n > 0) ldn<float>(q.bias + g * H + j8, bias[g], n, vec);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) bias[g][e] = 0.f;
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int lrow = tid / GPRW + it * (NTHR / GPRW);
    const int row = m0 + lrow;
    const bool live = n > 0 && lrow < BM && row < B;
    if (live && q.c_prev) ldn<float>(q.c_prev + (long)row * H + j8, cpv[it], n, vec);
    else if (live && q.hprev_t0) ldn<T>(reinterpret_cast<const T*>(q.hprev_t0) + (long)row * q.lda1 + j8, cpv[it], n, vec);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) cpv[it][e] = 0.f;
    }
  }
  }
  if (!WS || tid < 256) {
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wm * WM + i * 16 + lq * 4 + r;
          stg[(row * 4 + ni / JS) * SJ + wj * (BJ / 2) + (ni % JS) * 16 + lr] = acc[i][ni][r];
        }
  }
  __syncthreads();
  if (n <= 0) return;
  T* hout = reinterpret_cast<T*>(q.h_out);
  T* gout = reinterpret_cast<T*>(q.g_out);
  T* csave = reinterpret_cast<T*>(q.c_save);
  // the loop over the thread's rows, instantiated for (fast, has an addend): in the fast form without an addend -- every cell but layer 0 --
  // there is no load in the loop at all, so no wait between one row's stores and the next row's math
  auto rows = [&](auto fast_tag, auto add_tag) {
  constexpr bool FAST = decltype(fast_tag)::value, HAS_ADD = decltype(add_tag)::value;
  auto ld32 = [&](const float* src, float (&v)[8]) { if constexpr (FAST) load8<float>(src, v); else ldn<float>(src, v, n, vec); };
  auto st32 = [&](float* dst, const float (&v)[8]) { if constexpr (FAST) store8<float>(dst, v); else stn<float>(dst, v, n, vec); };
  auto stT = [&](T* dst, const float (&v)[8]) { if constexpr (FAST) store8<T>(dst, v); else stn<T>(dst, v, n, vec); };
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int lrow = tid / GPRW + it * (NTHR / GPRW);
    const int row = m0 + lrow;
    if constexpr (!FAST) { if (lrow >= BM || row >= B) continue; }
    float pre[4][8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float* s = stg + (lrow * 4 + g) * SJ + g8 * 8;
      const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
      pre[g][0] = a.x + bias[g][0]; pre[g][1] = a.y + bias[g][1]; pre[g][2] = a.z + bias[g][2]; pre[g][3] = a.w + bias[g][3];
      pre[g][4] = b.x + bias[g][4]; pre[g][5] = b.y + bias[g][5]; pre[g][6] = b.z + bias[g][6]; pre[g][7] = b.w + bias[g][7];
    }
    if constexpr (HAS_ADD) {
    if (q.add_tbl) {        // token table (a few hundred KB, L2-resident): the [T, B, 4H] gathered copy is never materialised
      long id = q.add_idx[(long)row * q.add_idx_ld];
      id = id < 0 ? 0 : (id >= q.add_rows ? q.add_rows - 1 : id);
      const float* tr = q.add_tbl + id * 4L * H + j8;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float a[8], b[8];
        ld32(tr + g * H, a);
        if (q.add) {        // (table row + per-sequence addend) first, as mvae_gather_rows_tb forms it: same bits as the gathered path
          ld32(q.add + (long)row * q.add_ld + g * H + j8, b);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += b[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) pre[g][e] += a[e];
      }
    } else if (q.add) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float a[8];
        ld32(q.add + (long)row * q.add_ld + g * H + j8, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) pre[g][e] += a[e];
      }
    }
    }
    const float (&cp)[8] = cpv[it];
    T* g4 = gout + (long)row * 4 * H + j8;
    if (p.cell == MVAE_CELL_LSTM) {
      float gi[8], gf[8], gg[8], go[8], c[8], h[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        gi[e] = act_sigmoid<T>(pre[0][e]); gf[e] = act_sigmoid<T>(pre[1][e]); gg[e] = act_tanh<T>(pre[2][e]); go[e] = act_sigmoid<T>(pre[3][e]);
        c[e] = gf[e] * cp[e] + gi[e] * gg[e];
        h[e] = go[e] * act_tanh<T>(c[e]);
      }
      st32(q.c_out + (long)row * H + j8, c);
      stT(hout + (long)row * q.ldh + j8, h);
      if (q.h_drop) {
        float hd[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) hd[e] = (e < n) ? h[e] * drop_factor(p.drop, q.dmask, q.didx0, (long)row * H + j8 + e) : 0.f;
        stT(reinterpret_cast<T*>(q.h_drop) + (long)row * q.ldh + j8, hd);
      }
      if (gout) {           // forward-only passes (no_grad) hand in no save buffers: 12 of the 16 bytes per (row, unit) are not written
        stT(csave + (long)row * H + j8, c);
        stT(g4, gi); stT(g4 + H, gf); stT(g4 + 2 * H, gg); stT(g4 + 3 * H, go);
      }
    } else {
      // GRU (torch.nn.GRU): slots = [r | z | W_in x + b_in | W_hn h + b_hn];  cp = h_{t-1} (fp32 recurrent state)
      const bool valid = p.lengths ? (q.t < p.lengths[row]) : true;
      float gr[8], gz[8], gn[8], hs_[8], hseq[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        gr[e] = act_sigmoid<T>(pre[0][e]); gz[e] = act_sigmoid<T>(pre[1][e]);
        gn[e] = act_tanh<T>(pre[2][e] + gr[e] * pre[3][e]);
        const float hn = (1.f - gz[e]) * gn[e] + gz[e] * cp[e];
        hs_[e] = valid ? hn : cp[e];          // a finished sequence keeps its last state (pack_sequence semantics) ...
        hseq[e] = valid ? hn : 0.f;           // ... and emits zeros (pad_packed_sequence)
      }
      st32(q.c_out + (long)row * H + j8, hs_);
      stT(hout + (long)row * q.ldh + j8, hseq);
      if (q.h_drop) {
        float hd[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) hd[e] = (e < n) ? hseq[e] * drop_factor(p.drop, q.dmask, q.didx0, (long)row * H + j8 + e) : 0.f;
        stT(reinterpret_cast<T*>(q.h_drop) + (long)row * q.ldh + j8, hd);
      }
      if (gout) { stT(g4, gr); stT(g4 + H, gz); stT(g4 + 2 * H, gn); stT(g4 + 3 * H, pre[3]); }
    }
  }
  };
  const bool has_add = q.add != nullptr || q.add_tbl != nullptr;
  if (fast) { if (has_add) rows(std::true_type{}, std::true_type{}); else rows(std::true_type{}, std::false_type{}); }
  else { if (has_add) rows(std::false_type{}, std::true_type{}); else rows(std::false_type{}, std::false_type{}); }
}

// ---------------------------------------------------------------------------------------------- forward, gate-major tile
// LSTM / bf16 / whole 64-unit tiles.  The roles of the MFMA operands are swapped with respect to lstm_step_fwd_kernel: the A operand
// is the WEIGHT tile (64 hidden units x 4 gates = 256 rows of [W_ih | W_hh]), the B operand the batch tile (BNB rows of x_t | h_{t-1}),
// so the accumulator tile is [gate rows][batch rows] and lane (q = lane >> 4, c = lane & 15) of a wave owns, for batch row c of each
// 16-row sub-tile, FOUR CONSECUTIVE W rows per fragment (rows 4q .. 4q+3).  The LDS image of the weight tile is filled in a permuted
// row order (the LDS-DMA source address is per lane, so this costs nothing):
//     LDS row r = 128 wm + 16 mi + j   <-   W row  g * H + j0 + 32 wm + 8 (j >> 2) + 4 p + (j & 3),   g = mi >> 1, p = mi & 1
// With it, fragments mi = 2g (p = 0) and 2g + 1 (p = 1) of a lane hold gate g of EIGHT consecutive hidden units 32 wm + 8 q .. + 7 of
// one batch row -- all four gates of those units sit in the lane's own registers.  The whole cell update (bias, non-linearities,
// c' = f c + i g, h' = o tanh c') therefore runs in registers: no LDS staging, no barrier, no bank conflicts, and every access of
// c / h / saved gates is a 16-byte (bf16) or 2 x 16-byte (fp32) vector per lane, 64 contiguous bytes per 4 lanes.
// 512 threads = 8 waves as 2 (weight rows) x 4 (batch rows), all of them loading AND computing (tile_gemm_pipe8); the 256-row batch
// tile needs 64 KB of operands per 64-deep K-step for 8.4 MFLOP (128 FLOP/B, against 85 for the 128-row tile), which is what counts for
// a loop bound by the per-CU L2 -> LDS intake.
// Small batches (B <= 256) take the same kernel with 32-unit tiles (BMW = 128 weight rows, 256 threads = 4 waves along the batch rows).
// WS (32-unit tiles only): 512 threads, waves 4-7 only issue the LDS-DMA pieces (tile_gemm_ws).  The small-batch tile's K-step is 16
// MFMAs per wave against 6 DMA pieces of 60-180 issue cycles each: with every wave doing both jobs the pieces, not the MFMAs, set the pace
// (~950 cycles per K-step for 256 cycles of MFMA at b = 128).
template <int BMW, int BNB, int NBUF, int MODE = 2, bool WS = false>
__global__ __launch_bounds__(WS ? 512 : BMW * 2) void lstm_step_fwd_gm_kernel(StepArgsF p) {
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(BMW == 256 || BMW == 128, "weight rows per tile: 64 or 32 units x 4 gates");
  static_assert(!WS || BMW == 128, "loader / consumer split: the 4-wave tile only");
  constexpr int NT = BMW * 2;                   // 8 waves as 2 (weight rows) x 4 (batch rows), or 4 waves as 1 x 4
  constexpr int BU = BMW / 4;                   // hidden units per tile
  constexpr int MI = 8, NI = BNB / 64;          // wave tile: 128 weight rows x (BNB / 4) batch rows
  const int tid = threadIdx.x, lane = tid & 63, wave = WS ? ((tid >> 6) & 3) : (tid >> 6), wm = wave >> 2, wn = wave & 3;
  const int per_task = p.tiles_m * p.tiles_j;
  const int bid = xcd_remap_r(blockIdx.x, p.ntask * per_task);
  const int task = bid / per_task;
  const int rem = bid - task * per_task;
  const int tj = rem / p.tiles_m, tm = rem - tj * p.tiles_m;   // tiles sharing a weight panel are neighbours on one XCD
  const StepTaskF& q = p.t[task];
  const int n0 = tm * BNB, j0 = tj * BU, H = p.H, B = p.B;

  // lane -> hidden units ub .. ub + 7 (all four gates), batch rows n0 + wn * 16 NI + 16 ni + lc  (see the epilogue)
  const int lq = lane >> 4, lc = lane & 15;
  const int ub = j0 + 32 * wm + 8 * lq;
  // The 256 x 256 tile STARTS its accumulators at the bias (loads in flight under the pipeline prologue; -0.1 ms / step at B = 1024, same-box
  // A/B); the small tiles add it in the epilogue: in front of their short main loop the same loads delay the first K-step (+0.5 ms / step at b = 128).
  constexpr bool BIAS_FIRST = (BMW == 256 && BNB == 256);
  auto load_bias = [&](float (&bias)[4][8]) {
    if (q.bias) {
#pragma unroll
      for (int g = 0; g < 4; ++g) load8<float>(q.bias + g * H + ub, bias[g]);
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) bias[g][e] = 0.f;
    }
  };
  f32x4 acc[MI][NI];
  if constexpr (BIAS_FIRST) {
    float bias[4][8];
    load_bias(bias);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[2 * g + (e >> 2)][ni][e & 3] = bias[g][e];
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int g = 0; g < NI; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // t == 0 has no previous cell state: the loads still go out (from this cell's own output buffer: valid memory of the same shape) and the
  // loop selects zero instead -- a branch around the loads would make the compiler wait for them at the join.  Rows past B read row B - 1.
  float cpa[NI][8];
  auto load_cprev_all = [&]() {
    const float* cpsrc = q.c_prev ? q.c_prev : q.c_out;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int row = n0 + wn * (16 * NI) + ni * 16 + lc;
      load8<float>(cpsrc + (long)(row < B ? row : B - 1) * H + ub, cpa[ni]);
    }
  };
  // 256 x 256 tile: requested before the main loop as well (the kernel has the registers: 249 VGPRs; -0.1 ... -0.25 ms / step at B = 1024)
  if constexpr (BIAS_FIRST) load_cprev_all();

#ifdef MVAE_TUNING
  const bool run_main = !(p.dbg & 2);
#else
  constexpr bool run_main = true;
#endif
  if (run_main) {
    PipeSeg<BMW, BNB, NT> s0, s1;
    auto wrow = [&](int r) -> int {             // LDS row -> W row (permutation above)
      const int mi = (r >> 4) & 7, j = r & 15;
      return (mi >> 1) * H + j0 + 32 * (r >> 7) + 8 * (j >> 2) + 4 * (mi & 1) + (j & 3);
    };
    {
      const uint32_t lda = (uint32_t)q.lda0 * 2u, ldw = (uint32_t)q.ldw0 * 2u;
      auto offW = [&](int r) -> uint32_t { return (uint32_t)wrow(r) * ldw; };
      auto offX = [&](int r) -> uint32_t { const int gn = n0 + r; return gn < B ? (uint32_t)gn * lda : PIPE_OOB; };
      pipe_seg_init<T, BMW, BNB, NT>(s0, q.W0, (uint32_t)(4 * H) * ldw, q.A0, (uint32_t)B * lda, offW, offX, q.K0, WS ? (tid & 255) : tid);
      if (q.A0 == nullptr) s0.nk = 0;
    }
    {
      const uint32_t lda = (uint32_t)q.lda1 * 2u, ldw = (uint32_t)q.ldw1 * 2u;
      auto offW = [&](int r) -> uint32_t { return (uint32_t)wrow(r) * ldw; };
      auto offX = [&](int r) -> uint32_t { const int gn = n0 + r; return gn < B ? (uint32_t)gn * lda : PIPE_OOB; };
      pipe_seg_init<T, BMW, BNB, NT>(s1, q.W1, (uint32_t)(4 * H) * ldw, q.A1, (uint32_t)B * lda, offW, offX, q.K1, WS ? (tid & 255) : tid);
      if (q.A1 == nullptr) s1.nk = 0;
    }
    if constexpr (WS) tile_gemm_ws<T, BMW, BNB, MI, NI, NBUF, NI, 0>(smem, s0, s1, 0, wn * (16 * NI), acc, tid);
    else tile_gemm_pipe_all<T, BMW, BNB, MI, NI, NBUF, NT, MODE>(smem, s0, s1, wm * 128, wn * (16 * NI), acc, tid);
  }
  if constexpr (WS) { if (tid >= 256) return; }          // loader waves own no accumulators
#ifdef MVAE_TUNING
  if (p.dbg & 1) { if (acc[0][0][0] == 12345.678f) q.c_out[0] = 1.f; return; }
#endif

  // ---- epilogue in registers.  The bias goes into the accumulators of every sub-tile first (one batch of loads under ONE uniform branch),
  // so that its 32 registers are free again before the previous cell states of all sub-tiles are requested.  (Measured: starting the
  // accumulators AT the bias, with its loads in front of the main loop, delays the first K-step instead: +0.7 ms / step at b = 128.)
  if constexpr (!BIAS_FIRST) {
    float bias[4][8];
    load_bias(bias);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[2 * g + (e >> 2)][ni][e & 3] += bias[g][e];
  }
  T* hout = reinterpret_cast<T*>(q.h_out);
  T* gout = reinterpret_cast<T*>(q.g_out);
  T* csave = reinterpret_cast<T*>(q.c_save);
  // Previous cell state of EVERY sub-tile is requested before the first store goes out, in one batch under ONE workgroup-uniform branch:
  // vmcnt retires loads and stores in issue order, so a load issued behind the stores of an earlier sub-tile cannot be waited for
  // without also waiting for those stores' write acknowledgements (~2 us each time), and a per-load branch makes the compiler wait after
  // every load.  With the loads in front the NI x 8 stores of a lane are fire-and-forget.  (Rows past B read row B - 1 and are not stored.)
  if constexpr (!BIAS_FIRST) load_cprev_all();
  const bool has_cprev = q.c_prev != nullptr;
  // the layer-0 cell (time-invariant input projection q.add) is its own instantiation of the loop: the other cells' loop then has no load
  // in it at all, and no wait between one sub-tile's stores and the next sub-tile's math
  // ... and so is the workgroup whose batch tile lies entirely inside the batch (no per-lane row test, hence no exec-mask branches
  // between the sub-tiles for the wait-count pass to be conservative about)
  auto epilogue = [&](auto has_add, auto full_tile) {
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int row = n0 + wn * (16 * NI) + ni * 16 + lc;
    if constexpr (!decltype(full_tile)::value) { if (row >= B) continue; }
    float cp[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cp[e] = has_cprev ? cpa[ni][e] : 0.f;
    float pre[4][8];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 8; ++e) pre[g][e] = acc[2 * g + (e >> 2)][ni][e & 3];
    if constexpr (decltype(has_add)::value) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float a[8];
        load8<float>(q.add + (long)row * q.add_ld + g * H + ub, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) pre[g][e] += a[e];
      }
    }
    float gi[8], gf[8], gg[8], go[8], c[8], h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      gi[e] = act_sigmoid<T>(pre[0][e]); gf[e] = act_sigmoid<T>(pre[1][e]); gg[e] = act_tanh<T>(pre[2][e]); go[e] = act_sigmoid<T>(pre[3][e]);
      c[e] = gf[e] * cp[e] + gi[e] * gg[e];
      h[e] = go[e] * act_tanh<T>(c[e]);
    }
#ifdef MVAE_TUNING
    // diagnostic builds only (results wrong): bit 2 = every store of the tile goes to its first 16 rows (same instructions, 1/16 of the
    // HBM write footprint: separates store ISSUE cost from write bandwidth), bit 3 = the saved gates are not stored (half the bytes)
    const int srow = (p.dbg & 4) ? n0 + (row & 15) : row;
    const bool sgate = !(p.dbg & 8);
#else
    const int srow = row;
    constexpr bool sgate = true;
#endif
    store8<float>(q.c_out + (long)srow * H + ub, c);
    store8<T>(hout + (long)srow * q.ldh + ub, h);
    if (gout) {
      T* g4 = gout + (long)srow * 4 * H + ub;
      store8<T>(csave + (long)srow * H + ub, c);
      if (sgate) { store8<T>(g4, gi); store8<T>(g4 + H, gf); store8<T>(g4 + 2 * H, gg); store8<T>(g4 + 3 * H, go); }
    }
  }
  };
  const bool full = n0 + BNB <= B;
  if (q.add) { if (full) epilogue(std::true_type{}, std::true_type{}); else epilogue(std::true_type{}, std::false_type{}); }
  else { if (full) epilogue(std::false_type{}, std::true_type{}); else epilogue(std::false_type{}, std::false_type{}); }
}

// ---------------------------------------------------------------------------------------------- backward
struct StepTaskB {
  const void *A0, *A1, *W0, *W1;
  long lda0, lda1, ldw0, ldw1;
  int K0, K1;
  const float* dy; long dy_ld;
  const void* gates; const void* c; const void* c_prev;     // dtype
  const float* dc_in; float* dc_out;                        // fp32 ping-pong
  void* dG; long ldg;
  const void* h_prev; long ldhp;      // GRU: h_{t-1} [B, ldhp] dtype (nullptr = zeros)
  float* dh0;                         // GRU pseudo-cell t = -1: gradient w.r.t. the initial hidden state (fp32 [B,H]); gates == nullptr
  int t;
  int seg1_full;                      // != 0: segment 1 is the output-gradient product dy_a . dy_w^T: contract all K1 columns (no GRU zero block)
  int drop;                           // != 0: the segment-1 contraction (gradient from the layer above) is multiplied by keep / (1 - p)
  const uint8_t* dmask;               // [B, H] injected keep mask of this cell's OUTPUT (nullptr: hash)
  uint32_t didx0;
};
struct StepArgsB { StepTaskB t[MVAE_MAX_LAYERS]; const int* lengths; float* partial; int ntask, B, H, tiles_m, tiles_n, dbg, vec, cell, split, gru3, tailpref; DropArgs drop; };

// Gate-derivative math of one (batch row, 8 hidden units) group, given dh = sum of the two contractions (fp32): the general form (any n,
// scalar or vector accesses, operands requested where they are used).  The element-wise launch of the split forms and ragged tiles use it.
template <typename T>
__device__ __forceinline__ void bwd_cell_group(const StepArgsB& p, const StepTaskB& q, int row, int j8, int n, bool vec, float (&dh)[8]) {
  const int H = p.H;
  const T* gates = reinterpret_cast<const T*>(q.gates);
  const T* cs = reinterpret_cast<const T*>(q.c);
  const T* csp = reinterpret_cast<const T*>(q.c_prev);
  T* dG = reinterpret_cast<T*>(q.dG);
  const long o = (long)row * H + j8;
  float dci[8];
  if (q.dc_in) ldn<float>(q.dc_in + o, dci, n, vec);
  else {
#pragma unroll
    for (int e = 0; e < 8; ++e) dci[e] = 0.f;
  }
  if (p.cell == MVAE_CELL_LSTM) {
    if (q.dy) {
      float a[8];
      ldn<float>(q.dy + (long)row * q.dy_ld + j8, a, n, vec);
#pragma unroll
      for (int e = 0; e < 8; ++e) dh[e] += a[e];
    }
    float gi[8], gf[8], gg[8], go[8], c[8], cp[8];
    const T* g4 = gates + (long)row * 4 * H + j8;
    ldn<T>(g4, gi, n, vec); ldn<T>(g4 + H, gf, n, vec); ldn<T>(g4 + 2 * H, gg, n, vec); ldn<T>(g4 + 3 * H, go, n, vec);
    ldn<T>(cs + o, c, n, vec);
    if (csp) ldn<T>(csp + o, cp, n, vec);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) cp[e] = 0.f;
    }
    float di[8], df[8], dg[8], dO[8], dco[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float tc = act_tanh<T>(c[e]);
      const float d_o = dh[e] * tc;
      const float dc = dh[e] * go[e] * (1.f - tc * tc) + dci[e];
      dco[e] = dc * gf[e];
      di[e] = dc * gg[e] * gi[e] * (1.f - gi[e]);
      df[e] = dc * cp[e] * gf[e] * (1.f - gf[e]);
      dg[e] = dc * gi[e] * (1.f - gg[e] * gg[e]);
      dO[e] = d_o * go[e] * (1.f - go[e]);
    }
    stn<float>(q.dc_out + o, dco, n, vec);
    T* d4 = dG + (long)row * q.ldg + j8;
    stn<T>(d4, di, n, vec); stn<T>(d4 + H, df, n, vec); stn<T>(d4 + 2 * H, dg, n, vec); stn<T>(d4 + 3 * H, dO, n, vec);
  } else {
    // GRU.  dh so far = dG^l_{t+1}[r,z,.,n*r] . W_hh + dG^{l+1}_t[r,z,n,.] . W_ih ; dci = the element-wise carry dh_{t+1} * z_{t+1}
#pragma unroll
    for (int e = 0; e < 8; ++e) dh[e] += dci[e];
    if (q.gates == nullptr) {                       // pseudo-cell t = -1: gradient w.r.t. the initial hidden state
      stn<float>(q.dh0 + o, dh, n, vec);
      return;
    }
    const bool valid = p.lengths ? (q.t < p.lengths[row]) : true;
    float dpr[8], dpz[8], dpn[8], dpnr[8], carry[8];
    if (valid) {
      if (q.dy) {
        float a[8];
        ldn<float>(q.dy + (long)row * q.dy_ld + j8, a, n, vec);
#pragma unroll
        for (int e = 0; e < 8; ++e) dh[e] += a[e];
      }
      float gr[8], gz[8], gn[8], nh[8], hp[8];
      const T* g4 = gates + (long)row * 4 * H + j8;
      ldn<T>(g4, gr, n, vec); ldn<T>(g4 + H, gz, n, vec); ldn<T>(g4 + 2 * H, gn, n, vec); ldn<T>(g4 + 3 * H, nh, n, vec);
      if (q.h_prev) ldn<T>(reinterpret_cast<const T*>(q.h_prev) + (long)row * q.ldhp + j8, hp, n, vec);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) hp[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dn = dh[e] * (1.f - gz[e]);
        const float dz = dh[e] * (hp[e] - gn[e]);
        dpn[e] = dn * (1.f - gn[e] * gn[e]);
        dpr[e] = dpn[e] * nh[e] * gr[e] * (1.f - gr[e]);
        dpz[e] = dz * gz[e] * (1.f - gz[e]);
        dpnr[e] = dpn[e] * gr[e];
        carry[e] = dh[e] * gz[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) { dpr[e] = 0.f; dpz[e] = 0.f; dpn[e] = 0.f; dpnr[e] = 0.f; carry[e] = dh[e]; }
    }
    stn<float>(q.dc_out + o, carry, n, vec);
    T* d4 = dG + (long)row * q.ldg + j8;
    stn<T>(d4, dpr, n, vec); stn<T>(d4 + H, dpz, n, vec); stn<T>(d4 + 2 * H, dpn, n, vec); stn<T>(d4 + 3 * H, dpnr, n, vec);
  }
}

// The same math in two phases for the fused kernel's whole tiles:
//   bwd_cell_load    requests everything the group reads from global memory (saved gates, cell states / h_{t-1}, the carried gradient, dy),
//   bwd_cell_finish  does the math and stores dG and the carried gradient,
// so that a caller can put the loads of its NEXT group in front of the stores of the current one: vmcnt retires loads and stores in issue
// order, and a load issued behind a store cannot be waited for without waiting for that store's write acknowledgement.
// FAST (vector path, whole 8-unit groups: what the host guarantees whenever H % 8 == 0 and everything is 16-byte aligned) is straight-line:
// every load goes out unconditionally -- an absent operand (no carried gradient at t = T-1, no dy below the top layer, no c_{t-1} at
// t = 0) is read from a buffer of the same shape that the kernel owns anyway and replaced by zero in bwd_cell_finish -- because each
// `if (ptr) load` is a branch, and at every join the compiler waits for all loads in flight: ten dependent round trips per group.
struct BwdOps { float dci[8], dy[8], g[4][8], c[8], cp[8]; };      // GRU: g = r, z, n, W_hn h + b_hn; cp = h_{t-1}; c unused

template <typename T, bool FAST>
__device__ __forceinline__ void bwd_cell_load(const StepArgsB& p, const StepTaskB& q, int row, int j8, int n, bool vec, BwdOps& r) {
  const int H = p.H;
  const T* gates = reinterpret_cast<const T*>(q.gates);
  const long o = (long)row * H + j8;
  if constexpr (FAST) {
    load8<float>((q.dc_in ? q.dc_in : q.dc_out) + o, r.dci);
    if (gates == nullptr) return;                   // GRU pseudo-cell t = -1 (workgroup-uniform): only the carry is read
    load8<float>(q.dy ? q.dy + (long)row * q.dy_ld + j8 : q.dc_out + o, r.dy);
    const T* g4 = gates + (long)row * 4 * H + j8;
    load8<T>(g4, r.g[0]); load8<T>(g4 + H, r.g[1]); load8<T>(g4 + 2 * H, r.g[2]); load8<T>(g4 + 3 * H, r.g[3]);
    if (p.cell == MVAE_CELL_LSTM) {
      const T* cs = reinterpret_cast<const T*>(q.c);
      const T* csp = reinterpret_cast<const T*>(q.c_prev);
      load8<T>(cs + o, r.c);
      load8<T>((csp ? csp : cs) + o, r.cp);
    } else {
      load8<T>(q.h_prev ? reinterpret_cast<const T*>(q.h_prev) + (long)row * q.ldhp + j8 : g4, r.cp);
    }
  } else {
    if (q.dc_in) ldn<float>(q.dc_in + o, r.dci, n, vec);
    if (gates == nullptr) return;
    if (q.dy) ldn<float>(q.dy + (long)row * q.dy_ld + j8, r.dy, n, vec);
    const T* g4 = gates + (long)row * 4 * H + j8;
    ldn<T>(g4, r.g[0], n, vec); ldn<T>(g4 + H, r.g[1], n, vec); ldn<T>(g4 + 2 * H, r.g[2], n, vec); ldn<T>(g4 + 3 * H, r.g[3], n, vec);
    if (p.cell == MVAE_CELL_LSTM) {
      ldn<T>(reinterpret_cast<const T*>(q.c) + o, r.c, n, vec);
      if (q.c_prev) ldn<T>(reinterpret_cast<const T*>(q.c_prev) + o, r.cp, n, vec);
    } else if (q.h_prev) ldn<T>(reinterpret_cast<const T*>(q.h_prev) + (long)row * q.ldhp + j8, r.cp, n, vec);
  }
}

template <typename T, bool FAST>
__device__ __forceinline__ void bwd_cell_finish(const StepArgsB& p, const StepTaskB& q, int row, int j8, int n, bool vec, const BwdOps& r,
                                                float (&dh)[8]) {
  const int H = p.H;
  T* dG = reinterpret_cast<T*>(q.dG);
  const long o = (long)row * H + j8;
  auto st32 = [&](float* dst, const float (&v)[8]) { if constexpr (FAST) store8<float>(dst, v); else stn<float>(dst, v, n, vec); };
  auto stT = [&](T* dst, const float (&v)[8]) { if constexpr (FAST) store8<T>(dst, v); else stn<T>(dst, v, n, vec); };
  const bool has_dci = q.dc_in != nullptr, has_dy = q.dy != nullptr;
  float dci[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) dci[e] = has_dci ? r.dci[e] : 0.f;
  if (p.cell == MVAE_CELL_LSTM) {
#pragma unroll
    for (int e = 0; e < 8; ++e) dh[e] += has_dy ? r.dy[e] : 0.f;
    const bool has_cp = q.c_prev != nullptr;
    float di[8], df[8], dg[8], dO[8], dco[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gi = r.g[0][e], gf = r.g[1][e], gg = r.g[2][e], go = r.g[3][e], cp = has_cp ? r.cp[e] : 0.f;
      const float tc = act_tanh<T>(r.c[e]);
      const float d_o = dh[e] * tc;
      const float dc = dh[e] * go * (1.f - tc * tc) + dci[e];
      dco[e] = dc * gf;
      di[e] = dc * gg * gi * (1.f - gi);
      df[e] = dc * cp * gf * (1.f - gf);
      dg[e] = dc * gi * (1.f - gg * gg);
      dO[e] = d_o * go * (1.f - go);
    }
    st32(q.dc_out + o, dco);
    T* d4 = dG + (long)row * q.ldg + j8;
    stT(d4, di); stT(d4 + H, df); stT(d4 + 2 * H, dg); stT(d4 + 3 * H, dO);
  } else {
    // GRU.  dh so far = dG^l_{t+1}[r,z,.,n*r] . W_hh + dG^{l+1}_t[r,z,n,.] . W_ih ; dci = the element-wise carry dh_{t+1} * z_{t+1}
#pragma unroll
    for (int e = 0; e < 8; ++e) dh[e] += dci[e];
    if (q.gates == nullptr) {                       // pseudo-cell t = -1: gradient w.r.t. the initial hidden state
      st32(q.dh0 + o, dh);
      return;
    }
    const bool valid = p.lengths ? (q.t < p.lengths[row]) : true;
    const bool has_hp = q.h_prev != nullptr;
    float dpr[8], dpz[8], dpn[8], dpnr[8], carry[8];
    if (valid) {
#pragma unroll
      for (int e = 0; e < 8; ++e) dh[e] += has_dy ? r.dy[e] : 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gr = r.g[0][e], gz = r.g[1][e], gn = r.g[2][e], nh = r.g[3][e], hp = has_hp ? r.cp[e] : 0.f;
        const float dn = dh[e] * (1.f - gz);
        const float dz = dh[e] * (hp - gn);
        dpn[e] = dn * (1.f - gn * gn);
        dpr[e] = dpn[e] * nh * gr * (1.f - gr);
        dpz[e] = dz * gz * (1.f - gz);
        dpnr[e] = dpn[e] * gr;
        carry[e] = dh[e] * gz;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) { dpr[e] = 0.f; dpz[e] = 0.f; dpn[e] = 0.f; dpnr[e] = 0.f; carry[e] = dh[e]; }
    }
    st32(q.dc_out + o, carry);
    T* d4 = dG + (long)row * q.ldg + j8;
    stT(d4, dpr); stT(d4 + H, dpz); stT(d4 + 2 * H, dpn); stT(d4 + 3 * H, dpnr);
  }
}

// Packed (unconverted) operands of one group: what a wave can hold across the whole main loop without touching them -- a bf16 -> fp32
// conversion at the load would make the wave wait for the data right there.  32 registers per group (40 with dy).
struct BwdRaw { float4 dci[2], dy[2]; uint4 g[4], c, cp; };
template <bool HAS_DY>
__device__ __forceinline__ void bwd_raw_load(const StepArgsB& p, const StepTaskB& q, int row, int j8, BwdRaw& r) {
  const int H = p.H;
  const bf16_t* gates = reinterpret_cast<const bf16_t*>(q.gates);
  const long o = (long)row * H + j8;
  const float* dsrc = (q.dc_in ? q.dc_in : q.dc_out) + o;
  r.dci[0] = *reinterpret_cast<const float4*>(dsrc); r.dci[1] = *reinterpret_cast<const float4*>(dsrc + 4);
  if (gates == nullptr) return;                   // GRU pseudo-cell t = -1 (workgroup-uniform): only the carry is read
  if constexpr (HAS_DY) {
    const float* ysrc = q.dy ? q.dy + (long)row * q.dy_ld + j8 : q.dc_out + o;
    r.dy[0] = *reinterpret_cast<const float4*>(ysrc); r.dy[1] = *reinterpret_cast<const float4*>(ysrc + 4);
  }
  const bf16_t* g4 = gates + (long)row * 4 * H + j8;
  r.g[0] = *reinterpret_cast<const uint4*>(g4); r.g[1] = *reinterpret_cast<const uint4*>(g4 + H);
  r.g[2] = *reinterpret_cast<const uint4*>(g4 + 2 * H); r.g[3] = *reinterpret_cast<const uint4*>(g4 + 3 * H);
  if (p.cell == MVAE_CELL_LSTM) {
    const bf16_t* cs = reinterpret_cast<const bf16_t*>(q.c);
    const bf16_t* csp = reinterpret_cast<const bf16_t*>(q.c_prev);
    r.c = *reinterpret_cast<const uint4*>(cs + o);
    r.cp = *reinterpret_cast<const uint4*>((csp ? csp : cs) + o);
  } else {
    r.cp = *reinterpret_cast<const uint4*>(q.h_prev ? reinterpret_cast<const bf16_t*>(q.h_prev) + (long)row * q.ldhp + j8 : g4);
  }
}
__device__ __forceinline__ void unpack8(const uint4& u, float (&v)[8]) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __builtin_bit_cast(float, w[i] << 16); v[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u); }
}
template <bool HAS_DY>
__device__ __forceinline__ void bwd_raw_unpack(const BwdRaw& r, BwdOps& o) {
  o.dci[0] = r.dci[0].x; o.dci[1] = r.dci[0].y; o.dci[2] = r.dci[0].z; o.dci[3] = r.dci[0].w;
  o.dci[4] = r.dci[1].x; o.dci[5] = r.dci[1].y; o.dci[6] = r.dci[1].z; o.dci[7] = r.dci[1].w;
  if constexpr (HAS_DY) {
    o.dy[0] = r.dy[0].x; o.dy[1] = r.dy[0].y; o.dy[2] = r.dy[0].z; o.dy[3] = r.dy[0].w;
    o.dy[4] = r.dy[1].x; o.dy[5] = r.dy[1].y; o.dy[6] = r.dy[1].z; o.dy[7] = r.dy[1].w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) o.dy[e] = 0.f;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) unpack8(r.g[g], o.g[g]);
  unpack8(r.c, o.c); unpack8(r.cp, o.cp);
}

// DROP: instantiations that can apply the inter-layer dropout factor inside the fused (single-launch) form -- a separate template
// flag so that the hot instantiations carry no trace of it (with both forms in one body hipcc moved the LDS-DMA descriptors to scratch).
// SPLITMODE: 1 = instantiation launched in split mode only (stores partial tiles; carries no gate-derivative epilogue), 0 = fused only
// (no partial-tile code), -1 = decided at run time.  The step kernels of the small batches run ~20 us: what an instantiation does not
// need is kept out of its code (the same kernel with both epilogues in it: +0.8 us per launch at b = 128).
// PREF (fused wave-specialised form only): 1 / 2 = the epilogue's operands are PREFETCHED (2: no cell of the stack has a dy tensor -- the
// output gradient enters as a product, so that load does not exist).  The loader waves request the saved gates / cell states / carried
// gradient of ALL their groups before the first LDS-DMA and hold them, packed, in registers they do not otherwise use (the counted vmcnt
// waits of the ring stay valid: loads retire in issue order and these are the oldest); the consumer waves request their first group there
// and the other three right after the main loop, in front of the staging barrier.  No load sits behind a store any more.
template <typename T, int BM, int BN, int NBUF, bool WS = false, bool DROP = false, int SPLITMODE = -1, int PREF = 0>
__global__ __launch_bounds__(WS ? 512 : 256) void lstm_step_bwd_kernel(StepArgsB p) {
  constexpr bool PIPE = NBUF > 0;
  constexpr int NTHR = WS ? 512 : 256;
  static_assert(!WS || NBUF > 0, "wave specialisation needs the LDS-direct ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 16, NI = WN / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, wm = wave >> 1, wn = wave & 1;
  const int per_task = p.tiles_m * p.tiles_n;
  const int nsp = p.split ? p.split : 1;         // split mode: 2 = one workgroup per K-segment, 4 = per half K-segment
  const int bid = xcd_remap_r(blockIdx.x, p.ntask * per_task * nsp);
  const int task2 = bid / per_task;
  const int task = task2 / nsp;
  const int sp = task2 - task * nsp;
  const int seg = (nsp == 4) ? (sp >> 1) : sp;   // which K-segment this workgroup contracts (split mode)
  const int half = (nsp == 4) ? (sp & 1) : 0;    // which half of it (4-way split)
  const int rem = bid - task2 * per_task;
  int tn = rem / p.tiles_m, tm = rem - tn * p.tiles_m;
  if (p.tiles_m >= 16 && (p.tiles_m & 7) == 0 && (p.tiles_n & 3) == 0) {
    // More row tiles than one round of an XCD's 32 CUs holds beside 4 column blocks (B = 2048: 16 x 8 tiles per cell, 64 per XCD): walk the
    // cell in blocks of 4 column blocks x 8 row tiles, row block fastest, so that the tiles resident on an XCD at one time share 4 weight
    // panels and 8 dG row panels (24 MB of operands per round instead of the 2 x 16 order's 36 MB) and the NEXT round re-uses the same weight
    // panels.  PMC, B = 2048 (profiles/r04_pmc_kernels_T16_B2048.json): 663 MB read per launch against 310 MB algorithmic in the old order.
    const int blk = rem >> 5, in = rem & 31, mb = p.tiles_m >> 3;
    const int tnb = blk / mb, tmb = blk - tnb * mb;
    tn = 4 * tnb + (in >> 3); tm = 8 * tmb + (in & 7);
  }
  const StepTaskB& q = p.t[task];
  const int m0 = tm * BM, n0 = tn * BN, H = p.H, B = p.B;
  if (p.split && (seg == 0 ? q.A0 == nullptr : q.A1 == nullptr)) return;   // this cell has no such segment: the epilogue kernel skips it too

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int n = 0; n < NI; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (PREF != 0) {
    static_assert(WS && NBUF > 0 && SPLITMODE == 0 && !DROP && sizeof(T) == 2, "prefetching form: fused, wave-specialised, bf16");
    constexpr bool HAS_DY = PREF == 1;
    constexpr int SN = BN + 4, GPR = BN / 8, RPI = NTHR / GPR, NIT = BM / RPI;
    static_assert(BM % RPI == 0, "whole row groups per thread");
    // host guarantees (rnn_bwd_impl, big_fused): B % BM == 0, H % BN == 0, 16-byte aligned everything -- every group is whole and in range
    PipeSeg<BM, BN> s0, s1;
#ifdef MVAE_TUNING
    // diagnostic build, MVAE_DBG bit 2: every workgroup contracts the SAME operand panels (tile 0 of its cell's matrices): all L2 hits after the
    // first touch -- separates the fabric (L2-miss) share of the main loop from the rest.  Results wrong.
    const int am0 = (p.dbg & 4) ? 0 : m0, an0 = (p.dbg & 4) ? 0 : n0;
#else
    const int am0 = m0, an0 = n0;
#endif
    {
      const uint32_t lda = (uint32_t)q.lda0 * 2u, ldw = (uint32_t)q.ldw0 * 2u;
      auto offA = [&](int r) -> uint32_t { return (uint32_t)(am0 + r) * lda; };
      auto offB = [&](int r) -> uint32_t { return (uint32_t)(an0 + r) * ldw; };
      pipe_seg_init<T, BM, BN>(s0, q.A0, (uint32_t)B * lda, q.W0, (uint32_t)H * ldw, offA, offB, p.gru3 ? 3 * H : q.K0, tid & 255);
      if (p.gru3) { s0.hole_st = (int)(2 * H * 2 / KB); s0.hole_bytes = (uint32_t)H * 2u; }
    }
    {
      const uint32_t lda = (uint32_t)q.lda1 * 2u, ldw = (uint32_t)q.ldw1 * 2u;
      auto offA = [&](int r) -> uint32_t { return (uint32_t)(am0 + r) * lda; };
      auto offB = [&](int r) -> uint32_t { return (uint32_t)(an0 + r) * ldw; };
      pipe_seg_init<T, BM, BN>(s1, q.A1, (uint32_t)B * lda, q.W1, (uint32_t)H * ldw, offA, offB, (p.gru3 && !q.seg1_full) ? 3 * H : q.K1, tid & 255);
    }
    const int nk = s0.nk + s1.nk;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g8 = tid % GPR, j8 = n0 + g8 * 8, row0 = m0 + tid / GPR;
    float* stg = reinterpret_cast<float*>(smem);
    BwdRaw raw[NIT];
    auto finish_all = [&]() {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int lrow = tid / GPR + it * RPI;
        const float* sp_ = stg + lrow * SN + g8 * 8;
        const float4 a = *reinterpret_cast<const float4*>(sp_), b = *reinterpret_cast<const float4*>(sp_ + 4);
        float dh[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        BwdOps o;
        bwd_raw_unpack<HAS_DY>(raw[it], o);
        bwd_cell_finish<T, true>(p, q, m0 + lrow, j8, 8, true, o, dh);
      }
    };
#ifdef MVAE_TUNING
    const bool run_main = !(p.dbg & 2);
#else
    constexpr bool run_main = true;
#endif
    if (wv >= 4) {
      // ---- loader waves: every group's operands first (oldest in the memory queue), then the ring
      // LSTM cells (a fixed number of operand loads per group): the loads go out BEHIND the last ring stage -- in front of the first they
      // delayed the whole main loop by their HBM latency (6 us of the launch, tune build) -- and land under the last three K-steps
      constexpr int NTAIL = NIT * (HAS_DY ? 10 : 8);
      const bool tail_form = p.cell == MVAE_CELL_LSTM && nk >= NBUF && run_main && p.tailpref;
      if (tail_form) {
        tile_gemm_ws_loader_tail<T, BM, BN, NBUF, NTAIL>(smem, s0, s1, wv - 4, [&]() {
#pragma unroll
          for (int it = 0; it < NIT; ++it) bwd_raw_load<HAS_DY>(p, q, row0 + it * RPI, j8, raw[it]);
        });
        ws_barrier();
      } else {
#ifdef MVAE_TUNING
      if (!(p.dbg & 64))                       // diagnostic build, MVAE_DBG bit 6: no operand prefetch (with bit 0: what the launch + barriers alone cost)
#endif
#pragma unroll
      for (int it = 0; it < NIT; ++it) bwd_raw_load<HAS_DY>(p, q, row0 + it * RPI, j8, raw[it]);
      if (nk > 0 && run_main) { tile_gemm_ws_loader<T, BM, BN, NBUF>(smem, s0, s1, wv - 4, p.dbg); ws_barrier(); }
      }
#ifdef MVAE_TUNING
      if (p.dbg & 1) return;
#endif
      __syncthreads();
      finish_all();
    } else {
#ifdef MVAE_TUNING
      if (!(p.dbg & 64))
#endif
      bwd_raw_load<HAS_DY>(p, q, row0, j8, raw[0]);
      if (nk > 0 && run_main) { tile_gemm_ws_consumer<T, BM, BN, MI, NI, NBUF, NI, 0, true>(smem, nk, wm * WM, wn * WN, acc, lane, p.dbg); ws_barrier(); }
#ifdef MVAE_TUNING
      if (p.dbg & 1) { if (acc[0][0][0] == 12345.678f) q.dc_out[0] = 1.f; return; }
#endif
      const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int n = 0; n < NI; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) stg[(wm * WM + i * 16 + lq * 4 + r) * SN + wn * WN + n * 16 + lr] = acc[i][n][r];
#pragma unroll
      for (int it = 1; it < NIT; ++it) bwd_raw_load<HAS_DY>(p, q, row0 + it * RPI, j8, raw[it]);
      __syncthreads();
      finish_all();
    }
    return;
  }
  const bool fused_drop = DROP && q.drop && !p.split && q.A1 != nullptr;      // wave-uniform; compiled out unless DROP
  auto scale_drop = [&]() {                     // accumulator layout: row = 4 (lane >> 4) + r, column = lane & 15 of each 16 x 16 sub-tile
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm * WM + i * 16 + (lane >> 4) * 4 + r, col = n0 + wn * WN + n * 16 + (lane & 15);
          if (row < B && col < H) acc[i][n][r] *= drop_factor(p.drop, q.dmask, q.didx0, (long)row * H + col);
        }
  };
#ifdef MVAE_TUNING
  const bool run_main = !(p.dbg & 2);
#else
  constexpr bool run_main = true;
#endif
  if (run_main) {
  if constexpr (PIPE) {
    PipeSeg<BM, BN> s0, s1;
    const uint32_t sz = (uint32_t)sizeof(T);
    {
      const uint32_t lda = (uint32_t)q.lda0 * sz, ldw = (uint32_t)q.ldw0 * sz;
      auto offA = [&](int r) -> uint32_t { const int gm = m0 + r; return gm < B ? (uint32_t)gm * lda : PIPE_OOB; };
      auto offB = [&](int r) -> uint32_t { const int gn = n0 + r; return gn < H ? (uint32_t)gn * ldw : PIPE_OOB; };
      // this workgroup's K range of segment 0 (elements): everything, or one half (4-way split).  GRU: the W_hh^T rows of gate slot 2
      // (k in [2H, 3H)) are a zero block -- skipped, as a hole in the K walk or by giving the second half only slot 3.
      int kh = (nsp == 4) ? q.K0 / 2 : q.K0, ko = half * kh * (int)sz;
      if (p.gru3) {
        if (nsp == 4) { kh = half ? H : 2 * H; ko = half ? 3 * H * (int)sz : 0; }
        else kh = 3 * H;
      }
      pipe_seg_init<T, BM, BN>(s0, q.A0 ? reinterpret_cast<const char*>(q.A0) + ko : nullptr, (uint32_t)B * lda - ko,
                               reinterpret_cast<const char*>(q.W0) + ko, (uint32_t)H * ldw - ko, offA, offB, kh, tid & 255);
      if (p.gru3 && nsp != 4) { s0.hole_st = (int)(2 * H * sz / KB); s0.hole_bytes = (uint32_t)H * sz; }
      if (p.split && seg == 1) s0.nk = 0;
    }
    {
      const uint32_t lda = (uint32_t)q.lda1 * sz, ldw = (uint32_t)q.ldw1 * sz;
      auto offA = [&](int r) -> uint32_t { const int gm = m0 + r; return gm < B ? (uint32_t)gm * lda : PIPE_OOB; };
      auto offB = [&](int r) -> uint32_t { const int gn = n0 + r; return gn < H ? (uint32_t)gn * ldw : PIPE_OOB; };
      const int k1 = (p.gru3 && !q.seg1_full) ? 3 * H : q.K1;   // GRU: the W_ih^T rows of gate slot 3 (k >= 3H) are a zero block
      const int kh = (nsp == 4) ? k1 / 2 : k1, ko = half * kh * (int)sz;
      pipe_seg_init<T, BM, BN>(s1, q.A1 ? reinterpret_cast<const char*>(q.A1) + ko : nullptr, (uint32_t)B * lda - ko,
                               reinterpret_cast<const char*>(q.W1) + ko, (uint32_t)H * ldw - ko, offA, offB, kh, tid & 255);
      if (p.split && seg == 0) s1.nk = 0;
    }
    if constexpr (DROP) {
      // inter-layer dropout, fused (single-launch) form: contract the segment that comes from the layer above FIRST (alone), multiply the
      // partial sums by keep / (1 - p) in the accumulator registers, then add the recurrent segment.  Straight-line: a cell without
      // dropout (top layer) runs the same two passes with a factor of one.
      const int nk0 = s0.nk;
      s0.nk = 0;
      tile_gemm_pipe<T, BM, BN, MI, NI, (NBUF > 0 ? NBUF : 3), NI, 0>(smem, s0, s1, wm * WM, wn * WN, acc, tid);
      if (fused_drop) scale_drop();
      s0.nk = nk0; s1.nk = 0;
      tile_gemm_pipe<T, BM, BN, MI, NI, (NBUF > 0 ? NBUF : 3), NI, 0>(smem, s0, s1, wm * WM, wn * WN, acc, tid);
    } else {
      if constexpr (WS) tile_gemm_ws<T, BM, BN, MI, NI, (NBUF > 0 ? NBUF : 3), NI, 0, (BM * BN <= 128 * 128)>(smem, s0, s1, wm * WM, wn * WN, acc, tid);   // interleaved fragment reads (the 256 x 128 tile has no registers to spare for them)
      else tile_gemm_pipe<T, BM, BN, MI, NI, (NBUF > 0 ? NBUF : 3), NI, 0>(smem, s0, s1, wm * WM, wn * WN, acc, tid);
    }
  } else {
    int brow[NI];
#pragma unroll
    for (int n = 0; n < NI; ++n) brow[n] = wn * WN + n * 16;
    if (fused_drop && q.A1 != nullptr) {                   // segment from the layer above first, then its dropout factor (see above)
      const T* A = reinterpret_cast<const T*>(q.A1);
      const T* W = reinterpret_cast<const T*>(q.W1);
      const long lda = q.lda1, ldw = q.ldw1;
      auto rowA = [&](int r) -> const T* { const int gm = m0 + r; return gm < B ? A + (long)gm * lda : nullptr; };
      auto rowB = [&](int r) -> const T* { const int gn = n0 + r; return gn < H ? W + (long)gn * ldw : nullptr; };
      tile_gemm_segment<T, BM, BN, MI, NI>(smem, rowA, rowB, 0, p.gru3 ? 3 * H : q.K1, wm * WM, brow, acc, tid);
      scale_drop();
    }
    if (q.A0 != nullptr) {
      const T* A = reinterpret_cast<const T*>(q.A0);
      const T* W = reinterpret_cast<const T*>(q.W0);
      const long lda = q.lda0, ldw = q.ldw0;
      auto rowA = [&](int r) -> const T* { const int gm = m0 + r; return gm < B ? A + (long)gm * lda : nullptr; };
      auto rowB = [&](int r) -> const T* { const int gn = n0 + r; return gn < H ? W + (long)gn * ldw : nullptr; };
      if (p.gru3) {                                       // skip the zero block k in [2H, 3H)
        tile_gemm_segment<T, BM, BN, MI, NI>(smem, rowA, rowB, 0, 2 * H, wm * WM, brow, acc, tid);
        tile_gemm_segment<T, BM, BN, MI, NI>(smem, rowA, rowB, 3 * H, 4 * H, wm * WM, brow, acc, tid);
      } else tile_gemm_segment<T, BM, BN, MI, NI>(smem, rowA, rowB, 0, q.K0, wm * WM, brow, acc, tid);
    }
    if (!fused_drop && q.A1 != nullptr) {
      const T* A = reinterpret_cast<const T*>(q.A1);
      const T* W = reinterpret_cast<const T*>(q.W1);
      const long lda = q.lda1, ldw = q.ldw1;
      auto rowA = [&](int r) -> const T* { const int gm = m0 + r; return gm < B ? A + (long)gm * lda : nullptr; };
      auto rowB = [&](int r) -> const T* { const int gn = n0 + r; return gn < H ? W + (long)gn * ldw : nullptr; };
      tile_gemm_segment<T, BM, BN, MI, NI>(smem, rowA, rowB, 0, p.gru3 ? 3 * H : q.K1, wm * WM, brow, acc, tid);
    }
  }
  }
#ifdef MVAE_TUNING
  if (p.dbg & 1) { if (acc[0][0][0] == 12345.678f) q.dc_out[0] = 1.f; return; }
#endif

  // ---- epilogue: dh tile -> LDS [row][BN + 4] (fp32), then one thread per (row, 8 hidden units)
  constexpr int SN = BN + 4;
  float* stg = reinterpret_cast<float*>(smem);
  if (!WS || tid < 256) {
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) stg[(wm * WM + i * 16 + lq * 4 + r) * SN + wn * WN + n * 16 + lr] = acc[i][n][r];
  }
  __syncthreads();
  constexpr int GPR = BN / 8;                  // 8-unit groups per tile row
  const int g8 = tid % GPR;
  const int j8 = n0 + g8 * 8;
  if (j8 >= H) return;
  const int n = (H - j8 < 8) ? (H - j8) : 8;
  const bool vec = p.vec != 0;
  if ((SPLITMODE == 1) || (SPLITMODE == -1 && p.split)) {
    // split mode (host guarantees whole tiles and the vector path): store this segment's fp32 partial tile; lstm_bwd_epi_kernel sums
    float* part = p.partial + ((long)(task * nsp + sp) * B) * H;
#pragma unroll
    for (int it = 0; it < (BM * GPR + NTHR - 1) / NTHR; ++it) {
      const int lrow = tid / GPR + it * (NTHR / GPR);
      const int row = m0 + lrow;
      if (lrow >= BM || row >= B) continue;
      const float* sp = stg + lrow * SN + g8 * 8;
      float* dst = part + (long)row * H + j8;
      *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(sp);
      *reinterpret_cast<float4*>(dst + 4) = *reinterpret_cast<const float4*>(sp + 4);
    }
    return;
  }
  if constexpr (SPLITMODE != 1) {
  constexpr int NIT = (BM * GPR + NTHR - 1) / NTHR;
  auto read_dh = [&](int lrow, float (&dh)[8]) {
    const float* s = stg + lrow * SN + g8 * 8;
    const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
    dh[0] = a.x; dh[1] = a.y; dh[2] = a.z; dh[3] = a.w; dh[4] = b.x; dh[5] = b.y; dh[6] = b.z; dh[7] = b.w;
  };
  // whole tile inside the batch, vector path, whole groups: the straight-line form, software-pipelined by one group -- the operands of
  // group it + 1 are requested before the stores of group it go out (bwd_cell_load / bwd_cell_finish)
  if (vec && n == 8 && m0 + BM <= B && (BM * GPR) % NTHR == 0) {
    BwdOps ops[2];
    bwd_cell_load<T, true>(p, q, m0 + tid / GPR, j8, 8, true, ops[0]);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int lrow = tid / GPR + it * (NTHR / GPR);
      if (it + 1 < NIT) bwd_cell_load<T, true>(p, q, m0 + lrow + NTHR / GPR, j8, 8, true, ops[(it + 1) & 1]);
      float dh[8];
      read_dh(lrow, dh);
      bwd_cell_finish<T, true>(p, q, m0 + lrow, j8, 8, true, ops[it & 1], dh);
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int lrow = tid / GPR + it * (NTHR / GPR);
    const int row = m0 + lrow;
    if (lrow >= BM || row >= B) continue;
    float dh[8];
    read_dh(lrow, dh);
    bwd_cell_group<T>(p, q, row, j8, n, vec, dh);
  }
  }
}

// Split mode, second launch: dh = partial(seg 0) + partial(seg 1), then the gate-derivative math.  One thread per (row, 8 units).
template <typename T>
__global__ __launch_bounds__(256) void lstm_bwd_epi_kernel(StepArgsB p) {
  const int H = p.H, B = p.B;
  const int gpr = H / 8;
  const long per_task = (long)B * gpr;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= per_task * p.ntask) return;
  const int task = (int)(gid / per_task);
  const long rem = gid - (long)task * per_task;
  const int row = (int)(rem / gpr), j8 = (int)(rem - (long)row * gpr) * 8;
  const StepTaskB& q = p.t[task];
  float dh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int nsp = p.split, per_seg = nsp >> 1;
  const float* p0 = p.partial + ((long)(task * nsp) * B + row) * H + j8;
  for (int s = 0; s < nsp; ++s) {                // fixed order: deterministic
    if ((s / per_seg == 0) ? (q.A0 == nullptr) : (q.A1 == nullptr)) continue;
    float a[8]; load8<float>(p0 + (long)s * B * H, a);
    if (q.drop && s / per_seg == 1) {             // gradient from the layer above: inter-layer dropout factor of this cell's output
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] *= drop_factor(p.drop, q.dmask, q.didx0, (long)row * H + j8 + e);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) dh[e] += a[e];
  }
  // One group per thread: the conditional form (operands requested one after the other, few registers, eight waves per SIMD to hide
  // it) is the faster one here -- the straight-line form costs 134 VGPRs = three waves per SIMD: 53.1 vs 50.7 us per launch pair at B = 512.
  bwd_cell_group<T>(p, q, row, j8, 8, true, dh);
}

// ---------------------------------------------------------------------------------------------- host drivers
#define MVAE_STEP_LAUNCH(KERN)                                                                                     \
  do {                                                                                                               \
    auto kern = KERN;                                                                                                \
    static bool attr_set = false;                                                                                    \
    if (!attr_set) { MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; } \
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);                                                               \
  } while (0)

// Schedule knobs (read per call; every setting computes the same results -- the GPU tests force each tile variant through them):
// MVAE_NBUF_FWD / MVAE_NBUF_BWD = LDS ring depth, MVAE_BM / MVAE_BJ = tile, MVAE_BWD_SPLIT, MVAE_ROWRES.  MVAE_DBG (skip parts of a
// kernel, results WRONG) exists only in the diagnostic build (-DMVAE_TUNING); the product library ignores it.
static int tune_int(const char* name, int dflt) {
  const char* v = mvae_knob(name);
  return v ? atoi(v) : dflt;
}

static inline const char* adv(const void* p, long elems, int dtype) {
  return reinterpret_cast<const char*>(p) + elems * (dtype == MVAE_BF16 ? 2 : 4);
}
static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int rnn_rowres_fwd(const mvae_rnn_fwd_desc* d, hipStream_t st);
int rnn_gru_rowres_fwd(const mvae_rnn_fwd_desc* d, hipStream_t st);

int rnn_fwd_impl(const mvae_rnn_fwd_desc* d, hipStream_t st) {
  if (!d) return MVAE_ERR_INVALID;
  if (d->cell != MVAE_CELL_LSTM && d->cell != MVAE_CELL_GRU) return MVAE_ERR_UNSUPPORTED;
  if (d->layers < 1 || d->layers > MVAE_MAX_LAYERS || d->T < 1 || d->B < 1 || d->H < 1) return MVAE_ERR_INVALID;
  if (d->dtype != MVAE_F32 && d->dtype != MVAE_BF16) return MVAE_ERR_INVALID;
  const int NL = d->layers, T = d->T, B = d->B, H = d->H, dt = d->dtype;
  const int epc = (dt == MVAE_BF16) ? 8 : 4;
  const bool gru = d->cell == MVAE_CELL_GRU;
  if (d->lengths && !gru) return MVAE_ERR_UNSUPPORTED;
  // forward-only (inference) call: no save buffers at all -- gates[l] == cs[l] == NULL for every layer
  const bool infer = !d->gates[0];
  for (int l = 0; l < NL; ++l) {
    if (infer ? (d->gates[l] || d->cs[l]) : ((!gru && !d->cs[l]) || !d->gates[l])) return MVAE_ERR_INVALID;
    if (!d->w_hh[l] || !d->hs[l] || !d->cstate[l]) return MVAE_ERR_INVALID;
    if (l > 0 && !d->w_ih[l]) return MVAE_ERR_INVALID;
    if (d->ldw_hh[l] % epc) return MVAE_ERR_INVALID;
    if (l > 0 && d->ldw_ih[l] % epc) return MVAE_ERR_INVALID;
  }
  if (d->x0 && (!d->w_ih[0] || d->in0 < 1)) return MVAE_ERR_INVALID;
  if (!d->x0 && !d->add0 && !d->add_table) return MVAE_ERR_INVALID;
  if (d->add_table && (!d->add_index || d->add_table_rows < 1 || d->add_index_ld < T)) return MVAE_ERR_INVALID;
  if (d->ldh % epc) return MVAE_ERR_INVALID;
  // inter-layer dropout: every layer but the last gets a dropped copy of its output, all or none
  bool drop = false;
  for (int l = 0; l < NL; ++l) drop = drop || d->hdrop[l] != nullptr;
  if (drop) {
    if (!(d->drop_p > 0.f) || d->drop_p >= 1.f) return MVAE_ERR_INVALID;
    for (int l = 0; l + 1 < NL; ++l) if (!d->hdrop[l]) return MVAE_ERR_INVALID;
  }
  // narrow f32 stacks (the encoder): row-resident schedule, one launch per layer instead of one per wavefront step
  if (!drop && tune_int("MVAE_ROWRES", 1)) {
    const int rc = rnn_rowres_fwd(d, st);
    if (rc != MVAE_ERR_UNSUPPORTED) return rc;
  }
  // one-layer bf16 GRU(256) over a token table (the MOSES encoder): row-resident as well, ONE launch for the whole sequence
  if (!drop && tune_int("MVAE_GRU_ROWRES", 1)) {
    const int rc = rnn_gru_rowres_fwd(d, st);
    if (rc != MVAE_ERR_UNSUPPORTED) return rc;
  }
  // row tile: the largest of 128 / 64 / 32 that still gives >= 256 workgroups per launch (the f32 MFMA rate is 1/16 of bf16, so
  // f32 stacks prefer many small tiles); MVAE_BM overrides
  auto nblk = [&](int bm) { return (long)((B + bm - 1) / bm) * ((H + 31) / 32) * NL; };
  int BM = (nblk(128) >= 256) ? 128 : (nblk(64) >= 256 || dt == MVAE_BF16) ? 64 : 32;
  if (dt == MVAE_BF16) BM = 64;     // measured (B=512, H=1024): 64-row tiles with a 3-deep ring keep two workgroups per CU -> 53 vs 64 us/launch
  BM = tune_int("MVAE_BM", BM);
  if (BM == 32 && dt == MVAE_BF16) BM = 64;
  const int sz = (dt == MVAE_BF16) ? 2 : 4, ke = KB / sz;
  // contraction length over the hidden axis: H, or H rounded up to whole K-steps when the caller guarantees zero padding
  int Hk = H;
  if (d->zero_padded_k && H % ke) {
    const int hp = (H + ke - 1) / ke * ke;
    bool ok = d->ldh >= hp && (!d->h0[0] || d->ldh0 >= hp);
    for (int l = 0; l < NL; ++l) ok = ok && d->ldw_hh[l] >= hp && (l == 0 || d->ldw_ih[l] >= hp);
    if (ok) Hk = hp;
  }
  // deep-pipelined LDS-direct main loop: whole K-steps, 16-byte aligned rows, operands < 2 GiB
  bool pipe = (Hk % ke == 0) && (d->ldh % (16 / sz) == 0) && ((long)B * d->ldh * sz < (1L << 31)) && (!d->x0 || (d->in0 % ke == 0 && d->x0_ld % (16 / sz) == 0)) &&
              (!d->h0[0] || d->ldh0 % (16 / sz) == 0);
  // 16-byte vector epilogue: 8-unit groups aligned in every array it touches
  bool vec = (H % 8 == 0) && (d->ldh % 8 == 0) && (!d->add0 || (al16(d->add0) && d->add0_tstride % 4 == 0)) && (!d->add_table || al16(d->add_table));
  for (int l = 0; l < NL; ++l) {
    if ((4L * H) * d->ldw_hh[l] * sz >= (1L << 31) || d->ldw_hh[l] % (16 / sz)) pipe = false;
    if ((l > 0 || d->x0) && ((4L * H) * d->ldw_ih[l] * sz >= (1L << 31) || d->ldw_ih[l] % (16 / sz))) pipe = false;
    if (!al16(d->hs[l]) || (!gru && !infer && !al16(d->cs[l])) || (!infer && !al16(d->gates[l])) || !al16(d->cstate[l]) || (d->bias[l] && !al16(d->bias[l]))) vec = false;
    if (d->h0[l] && (!al16(d->h0[l]) || d->ldh0 % 8)) vec = false;
  }
  StepArgsF a;
  a.lengths = d->lengths; a.cell = d->cell;
  // wide tile (128 rows x 64 units x 4 gates, one workgroup per CU): (128 + 256) operand rows per 32768 outputs -- half the L2->LDS
  // traffic per output of the 64 x 128 tile; used when it still gives >= 256 workgroups per full launch.  MVAE_BJ overrides.
  int BJ = (pipe && dt == MVAE_BF16 && (long)((B + 127) / 128) * ((H + 63) / 64) * NL >= 256) ? 64 : 32;
  BJ = tune_int("MVAE_BJ", BJ);
  if (BJ == 64 && (!pipe || dt != MVAE_BF16)) BJ = 32;
  if (BJ == 64) BM = 128;
  // gate-major tile (lstm_step_fwd_gm_kernel: LSTM, bf16, epilogue in registers): the largest of (64 units x 256 rows), (64 x 128),
  // (32 x 128), (32 x 64) that still gives about one workgroup per CU.  MVAE_FWD_GM: 0 = never, 1 = choose, 256256 / 256128 / 128128 /
  // 128064 = force that (weight rows, batch rows) tile (tests).
  const int gm_knob = tune_int("MVAE_FWD_GM", 1);
  int BMW = 0, BNB = 0;
  for (int l = 0; l < NL && drop; ++l) if (d->hdrop[l] && (!al16(d->hdrop[l]))) vec = false;
  if (gm_knob && !gru && !drop && !d->add_table && dt == MVAE_BF16 && pipe && vec && H % 64 == 0) {
    static const int cand[4][2] = {{256, 256}, {256, 128}, {128, 128}, {128, 64}};
    for (int c = 0; c < 4 && !BMW; ++c) {
      const long tiles = (long)((B + cand[c][1] - 1) / cand[c][1]) * (H / (cand[c][0] / 4)) * NL;
      // measured (4 x 1024, us / launch, this tile vs the 128 x (64 x 4) wave-specialised kernel): B=1024 67 vs 76, B=512 42.7 vs 42.9,
      // B=256 (32 x 128) 29.4 vs 28.5 -> not chosen, B=128 (32 x 64) 19.5 vs 21.0
      // in situ (whole training step, tests/ab_step.py): B=1024 -- this tile; B=512 (64 x 128) 18.66 vs 18.54 ms / step -> not chosen
      const bool chosen = gm_knob == 1 && tiles >= 192 && ((c == 0 && B >= 256) || (c == 3 && B >= 64 && B <= 128));
      if (gm_knob == cand[c][0] * 1000 + cand[c][1] || chosen) { BMW = cand[c][0]; BNB = cand[c][1]; }
    }
  }
  if (BMW) { BM = BNB; BJ = BMW / 4; }
  a.B = B; a.H = H; a.tiles_m = (B + BM - 1) / BM; a.tiles_j = (H + BJ - 1) / BJ; a.vec = vec ? 1 : 0;
  a.drop.scale = drop ? 1.f / (1.f - d->drop_p) : 1.f;
  a.drop.thresh = drop ? (uint32_t)((double)d->drop_p * 4294967296.0) : 0u;
  a.drop.seed = d->drop_seed;
  a.gru3 = (gru && tune_int("MVAE_GRU3", 1)) ? 1 : 0;       // GRU: never fetch the zero gate-slot block of either segment
#ifdef MVAE_TUNING
  a.dbg = tune_int("MVAE_DBG", 0);
#else
  a.dbg = 0;
#endif
  // 64 x (32 x 4) bf16 tile: a 3-deep ring (72 KB) lets two workgroups share a CU, a 2-deep one (48 KB) three.  The shallower ring wins exactly
  // when the third resident workgroup saves a round of workgroups: GRU 3 x 512 at B = 1024 is 768 tiles = 1.5 rounds of 512 but ONE round of
  // 768 (MOSES 7.28 -> 7.08 ms / step, models2d 10.18 -> 9.75); with 512 tiles (LSTM 4 x 1024 at B = 256) the deeper ring stays ahead.
  int nbuf_dflt = (BJ == 64) ? 3 : (dt == MVAE_BF16 && BM == 64) ? 3 : 4;
  if (dt == MVAE_BF16 && BM == 64 && BJ == 32 && !BMW) {
    const long nb = (long)((B + BM - 1) / BM) * ((H + BJ - 1) / BJ) * NL, cus = 256;
    if ((nb + 3 * cus - 1) / (3 * cus) < (nb + 2 * cus - 1) / (2 * cus)) nbuf_dflt = 2;
  }
  const int nbuf = pipe ? tune_int("MVAE_NBUF_FWD", nbuf_dflt) : 0;
  size_t lds = (size_t)(nbuf > 0 ? nbuf : 2) * (BM + 4 * BJ) * KB;
  const size_t stage_bytes = (size_t)BM * 4 * (BJ + 4) * sizeof(float);     // epilogue staging tile
  if (lds < stage_bytes) lds = stage_bytes;
  for (int dd = 0; dd < T + NL - 1; ++dd) {
    int n = 0;
    for (int l = 0; l < NL; ++l) {
      const int t = dd - l;
      if (t < 0 || t >= T) continue;
      StepTaskF& q = a.t[n++];
      if (l == 0) {
        q.A0 = d->x0 ? adv(d->x0, (long)t * B * d->x0_ld, dt) : nullptr;
        q.lda0 = d->x0_ld; q.K0 = d->in0; q.W0 = d->w_ih[0]; q.ldw0 = d->ldw_ih[0];
        q.add = d->add0 ? d->add0 + (long)t * d->add0_tstride : nullptr;
        q.add_ld = 4L * H;
        q.add_tbl = d->add_table; q.add_idx = d->add_table ? d->add_index + t : nullptr; q.add_idx_ld = d->add_index_ld; q.add_rows = d->add_table_rows;
      } else {
        q.A0 = adv(drop ? d->hdrop[l - 1] : d->hs[l - 1], (long)t * B * d->ldh, dt);     // the layer below, after its dropout
        q.lda0 = d->ldh; q.K0 = Hk; q.W0 = d->w_ih[l]; q.ldw0 = d->ldw_ih[l];
        q.add = nullptr; q.add_ld = 0;
        q.add_tbl = nullptr; q.add_idx = nullptr; q.add_idx_ld = 0; q.add_rows = 0;
      }
      q.A1 = (t > 0) ? adv(d->hs[l], (long)(t - 1) * B * d->ldh, dt) : d->h0[l];
      q.lda1 = (t > 0) ? d->ldh : d->ldh0;
      q.K1 = Hk; q.W1 = d->w_hh[l]; q.ldw1 = d->ldw_hh[l];
      q.bias = d->bias[l];
      q.c_prev = (t > 0) ? d->cstate[l] + (long)((t - 1) & 1) * B * H : nullptr;
      q.c_out = d->cstate[l] + (long)(t & 1) * B * H;
      q.c_save = (gru || infer) ? nullptr : const_cast<char*>(adv(d->cs[l], (long)t * B * H, dt));
      q.hprev_t0 = (gru && t == 0) ? d->h0[l] : nullptr;
      q.t = t;
      q.h_drop = (drop && l + 1 < NL) ? const_cast<char*>(adv(d->hdrop[l], (long)t * B * d->ldh, dt)) : nullptr;
      q.dmask = (drop && d->drop_mask[l]) ? d->drop_mask[l] + ((long)t * B) * H : nullptr;
      q.didx0 = (uint32_t)((((long)l * T + t) * B) * H);
      q.h_out = const_cast<char*>(adv(d->hs[l], (long)t * B * d->ldh, dt)); q.ldh = d->ldh;
      q.g_out = infer ? nullptr : const_cast<char*>(adv(d->gates[l], (long)t * B * 4 * H, dt));
    }
    a.ntask = n;
    dim3 grid(n * a.tiles_m * a.tiles_j), block(256);
    if (BMW) {
      block = dim3(BMW * 2);
      if (BMW == 256 && BNB == 256) {
        lds = 2 * (256 + 256) * KB;
#ifdef MVAE_TUNING
        const int mode = tune_int("MVAE_GM_MODE", 2);
        if (mode == 0) { MVAE_STEP_LAUNCH((lstm_step_fwd_gm_kernel<256, 256, 2, 0>)); continue; }
        if (mode == 1) { MVAE_STEP_LAUNCH((lstm_step_fwd_gm_kernel<256, 256, 2, 1>)); continue; }
        if (mode == 3) { MVAE_STEP_LAUNCH((lstm_step_fwd_gm_kernel<256, 256, 2, 3>)); continue; }
#endif
        MVAE_STEP_LAUNCH((lstm_step_fwd_gm_kernel<256, 256, 2>));
      }
      else if (BMW == 256) { lds = 3 * (256 + 128) * KB; MVAE_STEP_LAUNCH((lstm_step_fwd_gm_kernel<256, 128, 3>)); }
      else if (BNB == 128) { lds = 4 * (128 + 128) * KB; MVAE_STEP_LAUNCH((lstm_step_fwd_gm_kernel<128, 128, 4>)); }
      else {
        lds = 4 * (128 + 64) * KB;
        if (tune_int("MVAE_FWD_GM_WS", 1)) { block = dim3(512); MVAE_STEP_LAUNCH((lstm_step_fwd_gm_kernel<128, 64, 4, 2, true>)); }
        else MVAE_STEP_LAUNCH((lstm_step_fwd_gm_kernel<128, 64, 4>));
      }
      continue;
    }
#define FWD_CASE(TT_, BM_, NB_) if (BJ == 32 && BM == BM_ && nbuf == NB_) { MVAE_STEP_LAUNCH((lstm_step_fwd_kernel<TT_, BM_, 32, NB_>)); continue; }
    if (BJ == 64 && nbuf == 3) {
      block = dim3(512); MVAE_STEP_LAUNCH((lstm_step_fwd_kernel<bf16_t, 128, 64, 3, true>));      // loader / consumer wave specialisation
      continue;
    }
    if (dt == MVAE_BF16) {
      FWD_CASE(bf16_t, 128, 0) FWD_CASE(bf16_t, 128, 3) FWD_CASE(bf16_t, 128, 4) FWD_CASE(bf16_t, 128, 5)
      FWD_CASE(bf16_t, 64, 0) FWD_CASE(bf16_t, 64, 2) FWD_CASE(bf16_t, 64, 3) FWD_CASE(bf16_t, 64, 4) FWD_CASE(bf16_t, 64, 5)
    } else {
      FWD_CASE(float, 128, 0) FWD_CASE(float, 128, 4) FWD_CASE(float, 64, 0) FWD_CASE(float, 64, 4) FWD_CASE(float, 32, 0) FWD_CASE(float, 32, 4)
    }
#undef FWD_CASE
    return MVAE_ERR_UNSUPPORTED;
  }
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}

int rnn_rowres_bwd(const mvae_rnn_bwd_desc* d, hipStream_t st);
int rnn_gru_rowres_bwd(const mvae_rnn_bwd_desc* d, hipStream_t st);
size_t rnn_rowres_bwd_workspace(int layers, int T, int B, int H);

size_t rnn_bwd_workspace_bytes(const mvae_rnn_bwd_desc* d) {
  if (!d || d->layers < 1 || d->B < 1 || d->H < 1 || d->T < 1) return 0;
  const size_t split = (size_t)d->layers * 4 * d->B * d->H * sizeof(float);      // up to four fp32 partial dh tiles per cell of a wavefront launch
  const size_t rowres = rnn_rowres_bwd_workspace(d->layers, d->T, d->B, d->H);   // inter-layer input gradients of the row-resident schedule
  return split > rowres ? split : rowres;
}

int rnn_bwd_impl(const mvae_rnn_bwd_desc* d, hipStream_t st) {
  if (!d) return MVAE_ERR_INVALID;
  if (d->cell != MVAE_CELL_LSTM && d->cell != MVAE_CELL_GRU) return MVAE_ERR_UNSUPPORTED;
  if (d->layers < 1 || d->layers > MVAE_MAX_LAYERS || d->T < 1 || d->B < 1 || d->H < 1) return MVAE_ERR_INVALID;
  if (d->dtype != MVAE_F32 && d->dtype != MVAE_BF16) return MVAE_ERR_INVALID;
  const int NL = d->layers, T = d->T, B = d->B, H = d->H, dt = d->dtype;
  const int epc = (dt == MVAE_BF16) ? 8 : 4;
  const bool gru = d->cell == MVAE_CELL_GRU;
  if (!d->dy && !d->dy_a && !gru) return MVAE_ERR_INVALID;
  if (d->dy_a && (!d->dy_w || d->dy_k < 128 || d->dy_k % 128 || d->dy_a_ld < d->dy_k || d->dy_w_ld < d->dy_k || d->dy_a_ld % epc || d->dy_w_ld % epc ||
                  ((reinterpret_cast<uintptr_t>(d->dy_a) | reinterpret_cast<uintptr_t>(d->dy_w)) & 15)))
    return MVAE_ERR_INVALID;
  if (d->lengths && !gru) return MVAE_ERR_UNSUPPORTED;
  for (int l = 0; l < NL; ++l) {
    if (!d->w_hhT[l] || (!gru && !d->cs[l]) || !d->gates[l] || !d->dG[l] || !d->dstate[l]) return MVAE_ERR_INVALID;
    if (gru && !d->hs[l]) return MVAE_ERR_INVALID;
    if (l > 0 && !d->w_ihT[l]) return MVAE_ERR_INVALID;
    if (d->ldw_hhT[l] % epc || (l > 0 && d->ldw_ihT[l] % epc)) return MVAE_ERR_INVALID;
  }
  const long ldg = d->ldg;
  if (ldg < 4L * H || ldg % epc) return MVAE_ERR_INVALID;
  const bool drop = d->drop_p > 0.f;
  if (drop && d->drop_p >= 1.f) return MVAE_ERR_INVALID;
  if (!drop && tune_int("MVAE_ROWRES", 1)) {
    const int rc = rnn_rowres_bwd(d, st);
    if (rc != MVAE_ERR_UNSUPPORTED) return rc;
  }
  if (!drop && tune_int("MVAE_GRU_ROWRES", 1)) {             // one-layer bf16 GRU(256): row-resident, one launch (rnn_rowres.hip)
    const int rc = rnn_gru_rowres_bwd(d, st);
    if (rc != MVAE_ERR_UNSUPPORTED) return rc;
  }
  // bf16: 64-row tiles (two workgroups per CU: one's epilogue runs under the other's main loop); f32: 32 x 32 tiles when the
  // stack is small (MFMA-f32 bound per workgroup).  MVAE_BM overrides.
  int BM = 64, BN = 64;
  if (dt == MVAE_F32 && (long)((B + 63) / 64) * ((H + 63) / 64) * NL < 256) { BM = 32; BN = 32; }
  if (!drop) BM = tune_int("MVAE_BM", BM);
  if (BM == 32 && dt == MVAE_BF16) BM = 64;
  if (BM != 32) BN = 64;
  const int sz = (dt == MVAE_BF16) ? 2 : 4, ke = KB / sz;
  bool pipe = ((4 * H) % ke == 0) && ((long)B * ldg * sz < (1L << 31)) && (ldg % (16 / sz) == 0);
  bool vec = (H % 8 == 0) && (ldg % 8 == 0) && (!d->dy || (al16(d->dy) && (d->dy_ld % 4 == 0)));
  for (int l = 0; l < NL; ++l) {
    if ((long)H * d->ldw_hhT[l] * sz >= (1L << 31) || d->ldw_hhT[l] % (16 / sz)) pipe = false;
    if (l > 0 && ((long)H * d->ldw_ihT[l] * sz >= (1L << 31) || d->ldw_ihT[l] % (16 / sz))) pipe = false;
    if ((!gru && !al16(d->cs[l])) || !al16(d->gates[l]) || !al16(d->dG[l]) || !al16(d->dstate[l])) vec = false;
    if (gru && (!al16(d->hs[l]) || d->ldh % 8 || (d->h0[l] && (!al16(d->h0[l]) || d->ldh0 % 8)))) vec = false;
    if (d->dh_last[l] && !al16(d->dh_last[l])) vec = false;
    if (d->dh0[l] && !al16(d->dh0[l])) vec = false;
  }
  // Split mode (bf16, whole 128 x 128 tiles): the hidden axis is only H wide, so fused tiles must be small (64 x 64) to fill the chip and
  // each CU then streams (64+64) operand rows per 4096 outputs.  Splitting the contraction by SEGMENT across workgroups doubles the
  // parallelism instead: 128 x 128 tiles move half the bytes per output; the fp32 partial tiles (2 x B x H per cell, L2-resident) are summed
  // by a second, fully parallel element-wise launch that also does the gate-derivative math.
  // Tile / split choice (largest tile that still gives about one workgroup per CU): (256 x 128, 2 segments) for B >= 1024 at 4 x 1024,
  // (128 x 128, 2) at B = 512, (128 x 128, 4 = half segments) at B = 256, (128 x 64, 4) at B = 128.
  // MVAE_BWD_SPLIT: 0 = never, 1 = choose, 2 = (128 x 128, 2) whenever the shape allows, 2562 / 1284 / 644 = force (tests).
  if (d->dy_a && (!pipe || dt != MVAE_BF16 || (long)B * d->dy_a_ld * sz >= (1L << 31) || (long)H * d->dy_w_ld * sz >= (1L << 31))) return MVAE_ERR_UNSUPPORTED;
  const int split_knob = tune_int("MVAE_BWD_SPLIT", 1);
  int nsplit = 0;
  if (split_knob && dt == MVAE_BF16 && pipe && vec && B % 128 == 0 && H % 128 == 0 && d->split_ws) {
    struct Cand { int bm, bn, ns, key; };
    // (128 x 128, 1) = unsplit, gate-derivative epilogue fused into the wave-specialised GEMM kernel: no partial tiles at all; wins once it
    // fills the chip by itself (B = 1024: 31.7 vs 32.4 ms / step in situ against (256 x 128, 2); at B = 512 it has 128 workgroups: 70 vs 49 us)
    // (128 x 64, 1) = the same fused kernel on half-width tiles: B = 512 at 4 x 1024 is 256 of them -- one launch of 3 MB per CU instead of the
    // (128 x 128, 2) GEMM + element-wise pair (MVAE_BWD_SPLIT=641 forces it)
    // (256 x 128, 1) = the fused kernel on 256-row tiles (a quarter fewer operand bytes and LDS reads per FLOP than 128 x 128): only where it
    // still gives a workgroup per CU -- B >= 2048 at 4 x 1024 (BASELINE configs[4]); MVAE_BWD_SPLIT=2561 forces it (tests), MVAE_BWD_256=0 keeps it out
    static const Cand cand[7] = {{256, 128, 1, 2561}, {128, 128, 1, 1281}, {128, 64, 1, 641}, {256, 128, 2, 2562}, {128, 128, 2, 2}, {128, 128, 4, 1284}, {128, 64, 4, 644}};
    for (int c = 0; c < 7 && !nsplit; ++c) {
      if (B % cand[c].bm || (cand[c].ns == 4 && (4 * H) % (2 * ke))) continue;
      if (drop && cand[c].ns == 1) continue;      // the unsplit wave-specialised instantiation carries no dropout factor (DROP = false): never with a mask
      if (cand[c].key == 2561 && split_knob != 2561 && !tune_int("MVAE_BWD_256", 1)) continue;
      const long wgs = (long)(B / cand[c].bm) * (H / cand[c].bn) * cand[c].ns * NL;
      if (split_knob == cand[c].key || (split_knob == 1 && wgs >= (cand[c].ns == 1 ? 256 : 192))) { BM = cand[c].bm; BN = cand[c].bn; nsplit = cand[c].ns; }
    }
    // small stacks (GRU 3 x 512 at B = 128: 96 workgroups of (128 x 64, 4)) still beat the fused 64 x 64 tiles: 15 vs 25 us / launch
    if (!nsplit && split_knob == 1 && (4 * H) % (2 * ke) == 0 && (long)(B / 128) * (H / 64) * 4 * NL >= 64) { BM = 128; BN = 64; nsplit = 4; }
    if (nsplit > 1 && d->split_ws_bytes < (size_t)NL * nsplit * B * H * sizeof(float)) { nsplit = 0; BM = 64; BN = 64; }   // caller's scratch too small: the fused 64 x 64 tile
    // Short contractions (GRU 3 x 512: 6H = 3072 columns): a fused 64 x 64 tile streams under 1 MB, and with the pipelined straight-line
    // epilogue one fused launch beats the split GEMM + element-wise pair once it has about a workgroup per CU: MOSES B = 1024 6.99 -> 6.64,
    // B = 512 5.58 -> 5.10 ms / step (B <= 256: 4.43 vs 4.48, the split pair stays); 4 x 1024 (8H = 8192 columns, 2 MB per tile) never.
    if (split_knob == 1 && nsplit > 1) {
      const long fused_tiles = (long)((B + 63) / 64) * ((H + 63) / 64) * NL;
      const long kcols = (gru && tune_int("MVAE_GRU3", 1) ? 6L : 8L) * H;
      if (fused_tiles >= 192 && 128 * kcols * sz <= (1L << 20)) { nsplit = 0; BM = 64; BN = 64; }
      // A ONE-layer stack (the MOSES encoder GRU(256)): a cell is a few hundred KB of operands either way, and every time step of the
      // split form is two dependent launches (GEMM + element-wise) of ~7 us each, i.e. launch latency twice: the fused single launch wins
      // (MOSES B = 1024: 6.56 -> 6.26 ms / step, in-situ A/B)
      if (NL == 1 && 128 * kcols * sz <= (1L << 20)) { nsplit = 0; BM = 64; BN = 64; }
    }
  }
  const bool big_fused = nsplit == 1;            // (128 x 128, unsplit): the wave-specialised kernel with the gate-derivative epilogue fused
  if (big_fused) nsplit = 0;
  const bool split = nsplit != 0;
  StepArgsB a;
  a.lengths = d->lengths; a.cell = d->cell; a.split = nsplit; a.partial = split ? reinterpret_cast<float*>(d->split_ws) : nullptr;
  a.drop.scale = drop ? 1.f / (1.f - d->drop_p) : 1.f;
  a.drop.thresh = drop ? (uint32_t)((double)d->drop_p * 4294967296.0) : 0u;
  a.drop.seed = d->drop_seed;
  // GRU: skip the zero gate-slot block of each contraction (k in [2H,3H) of dG . W_hh^T, k >= 3H of dG_up . W_ih^T); the skipped ranges
  // must start on whole K-steps (and the 4-way split's half of 3H as well)
  a.tailpref = tune_int("MVAE_BWD_TAILPREF", 1);      // fused prefetching form: the loader waves' operand loads behind the last ring stage (A/B knob)
  a.gru3 = (gru && tune_int("MVAE_GRU3", 1) && H % ke == 0 && (nsplit != 4 || (3 * H / 2) % ke == 0)) ? 1 : 0;
  a.B = B; a.H = H; a.tiles_m = (B + BM - 1) / BM; a.tiles_n = (H + BN - 1) / BN; a.vec = vec ? 1 : 0;
#ifdef MVAE_TUNING
  a.dbg = tune_int("MVAE_DBG", 0);
#else
  a.dbg = 0;
#endif
  const int nbuf = pipe ? (drop ? 4 : tune_int("MVAE_NBUF_BWD", 4)) : 0;
  const bool ws = tune_int("MVAE_WS_BWD", 1) != 0;      // loader / consumer wave specialisation of the split-mode GEMM kernel
  size_t lds = (size_t)(split ? (BM == 256 ? 3 : 4) : big_fused ? (BM == 256 ? 3 : 4) : (nbuf > 0 ? nbuf : 2)) * (BM + BN) * KB;
  const size_t stage_bytes = (size_t)BM * (BN + 4) * sizeof(float);
  if (lds < stage_bytes) lds = stage_bytes;
  bool want_dh0 = false;
  for (int l = 0; l < NL; ++l) want_dh0 = want_dh0 || (gru && d->dh0[l]);
  for (int e = T + NL - 2; e >= (want_dh0 ? -1 : 0); --e) {
    int n = 0;
    if (e < 0) {
      // GRU pseudo-cells t = -1: dh0[l] = dG^l_0[r,z,.,n*r] . W_hh^l + (element-wise carry of t = 0)
      for (int l = 0; l < NL; ++l) {
        if (!d->dh0[l]) continue;
        StepTaskB& q = a.t[n++];
        q.A0 = d->dG[l]; q.lda0 = ldg; q.K0 = 4 * H; q.W0 = d->w_hhT[l]; q.ldw0 = d->ldw_hhT[l];
        q.A1 = nullptr; q.lda1 = ldg; q.K1 = 4 * H; q.W1 = nullptr; q.ldw1 = 0;
        q.dy = nullptr; q.dy_ld = 0; q.gates = nullptr; q.c = nullptr; q.c_prev = nullptr;
        q.dc_in = d->dstate[l]; q.dc_out = nullptr; q.dG = nullptr; q.ldg = ldg; q.h_prev = nullptr; q.ldhp = 0;
        q.dh0 = d->dh0[l]; q.t = -1; q.drop = 0; q.dmask = nullptr; q.didx0 = 0; q.seg1_full = 0;
      }
      if (n == 0) break;
    }
    for (int l = 0; l < NL && e >= 0; ++l) {
      const int t = e - l;
      if (t < 0 || t >= T) continue;
      StepTaskB& q = a.t[n++];
      q.A0 = (t < T - 1) ? adv(d->dG[l], (long)(t + 1) * B * ldg, dt) : nullptr;
      q.lda0 = ldg; q.K0 = 4 * H; q.W0 = d->w_hhT[l]; q.ldw0 = d->ldw_hhT[l];
      q.A1 = (l < NL - 1) ? adv(d->dG[l + 1], (long)t * B * ldg, dt) : nullptr;
      q.lda1 = ldg; q.K1 = 4 * H; q.W1 = (l < NL - 1) ? d->w_ihT[l + 1] : nullptr; q.ldw1 = (l < NL - 1) ? d->ldw_ihT[l + 1] : 0;
      q.seg1_full = 0;
      if (l == NL - 1 && d->dy_a) {       // the top cell has no layer above: its second K-segment contracts dy_t = dy_a[t] . dy_w^T instead
        q.A1 = adv(d->dy_a, (long)t * B * d->dy_a_ld, dt); q.lda1 = d->dy_a_ld; q.K1 = d->dy_k; q.W1 = d->dy_w; q.ldw1 = d->dy_w_ld;
        q.seg1_full = 1;
      }
      q.dy = (l == NL - 1 && d->dy) ? d->dy + (long)t * B * d->dy_ld : nullptr; q.dy_ld = d->dy_ld;
      q.gates = adv(d->gates[l], (long)t * B * 4 * H, dt);
      q.c = gru ? nullptr : adv(d->cs[l], (long)t * B * H, dt);
      q.c_prev = (!gru && t > 0) ? adv(d->cs[l], (long)(t - 1) * B * H, dt) : nullptr;
      q.dc_in = (t < T - 1) ? d->dstate[l] + (long)((t + 1) & 1) * B * H : (gru ? d->dh_last[l] : nullptr);
      q.h_prev = gru ? ((t > 0) ? adv(d->hs[l], (long)(t - 1) * B * d->ldh, dt) : d->h0[l]) : nullptr;
      q.ldhp = (t > 0) ? d->ldh : d->ldh0;
      q.dh0 = nullptr; q.t = t;
      q.drop = (drop && l < NL - 1) ? 1 : 0;
      q.dmask = (q.drop && d->drop_mask[l]) ? d->drop_mask[l] + ((long)t * B) * H : nullptr;
      q.didx0 = (uint32_t)((((long)l * T + t) * B) * H);
      q.dc_out = d->dstate[l] + (long)(t & 1) * B * H;
      q.dG = const_cast<char*>(adv(d->dG[l], (long)t * B * ldg, dt)); q.ldg = ldg;
    }
    a.ntask = n;
    dim3 grid(n * a.tiles_m * a.tiles_n * (split ? nsplit : 1)), block(256);
    if (big_fused && drop) return MVAE_ERR_UNSUPPORTED;      // (excluded above; an unhandled combination must never run silently)
    if (big_fused) {
      block = dim3(512);
      const int pref = tune_int("MVAE_BWD_PREF", 1);           // 0: the round-2 form (operands requested inside the epilogue), A/B knob
      if (BM == 256) {              // (the round-2 form of the fused epilogue: the prefetching forms are written out for a ring of four)
        MVAE_STEP_LAUNCH((lstm_step_bwd_kernel<bf16_t, 256, 128, 3, true, false, 0>));
        continue;
      }
      if (BN == 64) {
        if (d->dy) MVAE_STEP_LAUNCH((lstm_step_bwd_kernel<bf16_t, 128, 64, 4, true, false, 0, 1>));
        else MVAE_STEP_LAUNCH((lstm_step_bwd_kernel<bf16_t, 128, 64, 4, true, false, 0, 2>));
        continue;
      }
      if (!pref) MVAE_STEP_LAUNCH((lstm_step_bwd_kernel<bf16_t, 128, 128, 4, true, false, 0>));
      else if (d->dy) MVAE_STEP_LAUNCH((lstm_step_bwd_kernel<bf16_t, 128, 128, 4, true, false, 0, 1>));
      else MVAE_STEP_LAUNCH((lstm_step_bwd_kernel<bf16_t, 128, 128, 4, true, false, 0, 2>));
      continue;
    }
    if (split) {
      if (BM == 256) { block = dim3(512); MVAE_STEP_LAUNCH((lstm_step_bwd_kernel<bf16_t, 256, 128, 3, true, false, 1>)); block = dim3(256); }
      else if (BN == 64) { block = dim3(512); MVAE_STEP_LAUNCH((lstm_step_bwd_kernel<bf16_t, 128, 64, 4, true, false, 1>)); block = dim3(256); }
      else if (ws) { block = dim3(512); MVAE_STEP_LAUNCH((lstm_step_bwd_kernel<bf16_t, 128, 128, 4, true, false, 1>)); block = dim3(256); }
      else MVAE_STEP_LAUNCH((lstm_step_bwd_kernel<bf16_t, 128, 128, 4, false, false, 1>));
      const long groups = (long)n * B * (H / 8);
      hipLaunchKernelGGL((lstm_bwd_epi_kernel<bf16_t>), dim3((unsigned)((groups + 255) / 256)), block, 0, st, a);
      continue;
    }
#define BWD_DROP(TT_, BM_, NB_) if (BM == BM_ && nbuf == NB_) { MVAE_STEP_LAUNCH((lstm_step_bwd_kernel<TT_, BM_, (BM_ == 32 ? 32 : 64), NB_, false, true>)); continue; }
    if (drop) {                                   // fused form with the dropout factor (tile knobs are ignored: see the host choice above)
      if (dt == MVAE_BF16) { BWD_DROP(bf16_t, 64, 4) BWD_DROP(bf16_t, 64, 0) }
      else { BWD_DROP(float, 64, 4) BWD_DROP(float, 64, 0) BWD_DROP(float, 32, 4) BWD_DROP(float, 32, 0) }
      return MVAE_ERR_UNSUPPORTED;
    }
#undef BWD_DROP
#define BWD_CASE(TT_, BM_, NB_) if (BM == BM_ && nbuf == NB_) { MVAE_STEP_LAUNCH((lstm_step_bwd_kernel<TT_, BM_, (BM_ == 32 ? 32 : 64), NB_>)); continue; }
    if (dt == MVAE_BF16) {
      BWD_CASE(bf16_t, 128, 0) BWD_CASE(bf16_t, 128, 3) BWD_CASE(bf16_t, 128, 4) BWD_CASE(bf16_t, 128, 5) BWD_CASE(bf16_t, 128, 6)
      BWD_CASE(bf16_t, 64, 0) BWD_CASE(bf16_t, 64, 3) BWD_CASE(bf16_t, 64, 4) BWD_CASE(bf16_t, 64, 5) BWD_CASE(bf16_t, 64, 6)
    } else {
      BWD_CASE(float, 128, 0) BWD_CASE(float, 128, 4) BWD_CASE(float, 64, 0) BWD_CASE(float, 64, 4) BWD_CASE(float, 32, 0) BWD_CASE(float, 32, 4)
    }
#undef BWD_CASE
    return MVAE_ERR_UNSUPPORTED;
  }
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}

