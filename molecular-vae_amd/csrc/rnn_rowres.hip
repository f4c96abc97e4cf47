// Row-resident LSTM schedule for narrow f32 stacks (the reference's encoder: LSTM(30 -> 72, 3 layers), models.py:117,128).
//
// The layer-wavefront schedule (rnn.hip) pays one kernel launch per wavefront step; for H = 72 a step is ~0.1 GFLOP, so the
// 122 dependent launches of a pass are pure launch + cold-cache latency (12 us each, GPU idle).  An LSTM's recurrence only
// couples the hidden units of ONE batch row, so here a workgroup owns 4 batch rows of one layer and walks the whole time axis
// by itself: no inter-workgroup communication, one launch per layer (3 per pass), 128 workgroups at B = 512.
//   * weights live in REGISTERS for the whole pass: wave w holds the K-slice [K/4 * w, K/4 * (w+1)) of [W_ih | W_hh] as the B
//     operands of v_mfma_f32_4x4x1_16B_f32 (one VGPR per (k, 64 gate columns));
//   * that MFMA with cbsz = 4 broadcasts the 4-row A block `abid` to all 16 column blocks, i.e. it is a 4 x 64 x 1 outer
//     product: one VGPR (lane 4b + i <- A[row i][k0 + b]) feeds 16 consecutive k;
//   * per step: MFMA phase (each wave: its K-slice, all 4H columns) -> partial sums to LDS -> barrier -> gate phase (thread
//     per (row, unit): sum 4 partials, non-linearities, c/h update in registers, h into the next step's A buffer) -> barrier.
// Exact f32 (v_mfma_f32_4x4x1 == fmaf chain).  Backward: the same structure over dG . [W_ih | W_hh] (rnn_rowres_bwd).
#include <cstdlib>
#include "common.hpp"
#include "kernels.hpp"
#include "tile_pipe.hpp"
#include <stdlib.h>

namespace {

constexpr int RR_ROWS = 4;
// Barrier of the time loops: this wave's LDS traffic has completed (lgkmcnt), then s_barrier -- WITHOUT the vmcnt(0) that __syncthreads()
// carries (its workgroup-scope fence): nothing a step writes to global memory is read by another wave of the pass, and with it every barrier
// waited for the write acknowledgements of the step's hs / cs / gates (dG / dx) stores, twice per time step.
__device__ __forceinline__ void rr_barrier() { wait_lgkmcnt<0>(); ws_barrier(); }

template <int KK, int KW, int NG> struct RowMma {
  static __device__ __forceinline__ void run(const float (&areg)[(KW + 15) / 16], const float (&W)[NG][KW], f32x4 (&C)[NG]) {
    const float a = areg[KK / 16];
#pragma unroll
    for (int g = 0; g < NG; ++g) C[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, W[g][KK], C[g], 4, KK % 16, 0);
    RowMma<KK + 1, KW, NG>::run(areg, W, C);
  }
};
template <int KW, int NG> struct RowMma<KW, KW, NG> {
  static __device__ __forceinline__ void run(const float (&)[(KW + 15) / 16], const float (&)[NG][KW], f32x4 (&)[NG]) {}
};

// v_exp_f32 / v_rcp_f32 forms: absolute error ~1e-7, far inside the 1e-4 the encoder outputs are held to (tests: 1e-5)
// (__builtin_amdgcn_rcpf, not `/` or __fdividef: those expand to the ~10-instruction IEEE division sequence, and the gate phase is a
// dependent chain that every one of the 120 steps waits for)
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) {
  const float t = __expf(-2.f * fabsf(x));
  return copysignf((1.f - t) * __builtin_amdgcn_rcpf(1.f + t), x);
}
// NTHR = 320: waves 0-3 run the MFMA phase (one K-slice each), all 5 waves share the gate phase (one (row, unit) item per thread).
// NTHR = 256 where the weight fragments need more than 256 VGPRs per wave (backward with an input gradient: 216 + state).

struct RowResF {
  const float* x; long ldx;          // layer input [T][B][ldx] (nullptr: layer 0, pre-activations come from `add`)
  const float* add; long add_ts;     // [T][B][4H] fp32 pre-activation addend, row stride 4H, time stride add_ts (or nullptr)
  const float* w_ih; long ldw_ih;    // [4H][ldw_ih]
  const float* w_hh; long ldw_hh;    // [4H][ldw_hh]
  const float* bias;                 // [4H] or nullptr
  float* hs; long ldh;               // [T][B][ldh]
  float* cs;                         // [T][B][H]
  float* gates;                      // [T][B][4H] post-activation i, f, g, o
  int T, B;
  const float* tbl; int tbl_rows;    // TBL form (layer 0): pre-activation addend = tbl[idx[b][t]] ([rows][4H] fp32: the embedding folded into
  const int64_t* idx; long idx_ld;   //   W_ih0, bias included), ids [B][idx_ld] -- the table and this workgroup's ids live in LDS
};

constexpr int RR_TBL_ROWS = 64, RR_TBL_T = 256;      // LDS table / id capacity of the TBL form

// Pipeline context of the layer-concurrent forms: `above` = progress word of the workgroup that produces our input (same batch rows, the
// neighbouring layer; nullptr: the input is complete), `mine` = our own progress word (nullptr: nobody consumes our output inside the launch).
// Forward: progress = number of completed time steps; hs[t] is visible once progress >= t + 1.
struct RowPipe { const uint32_t* above; uint32_t* mine; uint32_t* status; uint32_t spin_limit; float* poison; };

template <int H, bool HASX, int NTHR, bool TBL = false, bool PIPE = false>
__device__ __forceinline__ void lstm_rowres_fwd_body(const RowResF& p, int rowgroup, RowPipe pp) {
  constexpr int G4 = 4 * H, K = (HASX ? 2 * H : H), KW = K / 4, NG = (G4 + 63) / 64, NA = (KW + 15) / 16;
  constexpr int KPAD = 2 * H + 8, NPAD = NG * 64;
  static_assert(K % 4 == 0 && (KPAD % 32) == 24, "K-slices per wave; A-buffer rows on disjoint banks");
  __shared__ float abuf[2][RR_ROWS][KPAD];            // [x_t | h_{t-1}] of the 4 rows, double-buffered
  __shared__ float red[4][RR_ROWS][NPAD];             // per-wave partial pre-activations
  // TBL: the [rows, 4H] token table (35 x 288 fp32 = 40 KB) and the ids of this workgroup's rows, copied once: the gathered [T, B, 4H]
  // sequence (141 MB at B = 1024) is never written or read (models.py:127 nn.Embedding + the layer-0 input projection)
  __shared__ float tblS[TBL ? RR_TBL_ROWS * G4 : 1];
  __shared__ int idS[TBL ? RR_ROWS * RR_TBL_T : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = rowgroup * RR_ROWS, B = p.B, T = p.T;
  uint32_t known = 0;
  auto wait_above = [&](uint32_t need) {          // PIPE: bounded poll of the producer's progress word (see rowres_bwd_layer)
    if (!PIPE || pp.above == nullptr || known >= need) return;
    for (uint32_t it = 0; it < pp.spin_limit; ++it) {
      known = __hip_atomic_load(pp.above, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (known >= need) return;
      __builtin_amdgcn_s_sleep(4);
    }
    if (pp.status && atomicCAS(pp.status, 0u, 3u) == 0u && pp.poison) *pp.poison = __builtin_nanf("");
    known = 0xffffffffu;
  };
  auto load_x = [&](const float* src) -> float4 {  // the layer below's h: agent-scope loads in the pipelined form (not through this CU's L1)
    if (!PIPE) return *reinterpret_cast<const float4*>(src);
    float4 v;
    v.x = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v.y = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.z = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v.w = __hip_atomic_load(src + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
  };
  if (TBL) {
    for (int i = tid; i < p.tbl_rows * G4; i += NTHR) tblS[i] = p.tbl[i];
    for (int i = tid; i < RR_ROWS * T; i += NTHR) {
      const int row = i / T, t = i - row * T;
      long id = (r0 + row < B) ? p.idx[(long)(r0 + row) * p.idx_ld + t] : 0;
      idS[row * RR_TBL_T + t] = (int)(id < 0 ? 0 : (id >= p.tbl_rows ? p.tbl_rows - 1 : id));
    }
  }

  // ---- weights -> registers (once): W[g][kk] = Wcat[n = 64 g + lane][k = KW * wave + kk]
  float W[NG][KW];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int n = 64 * g + lane;
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
      const int k = KW * (wave & 3) + kk;
      float v = 0.f;
      if (n < G4 && wave < 4) {
        if (HASX && k < H) v = p.w_ih[(long)n * p.ldw_ih + k];
        else v = p.w_hh[(long)n * p.ldw_hh + (HASX ? k - H : k)];
      }
      W[g][kk] = v;
    }
  }
  // The weight registers are complete BEFORE the time loop is entered.  Without this the wait-count pass sees them as possibly still in
  // flight at the loop header (pending on the entry edge, complete on the back edge) and puts vmcnt(0) in front of the first MFMA of
  // EVERY step -- which also drains the prefetch loads issued a few instructions earlier and the previous step's stores.
  __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0), expcnt / lgkmcnt untouched
  // ---- per-thread items of the gate phase: (row, unit) = (id / H, id % H), id = tid (+ 256)
  constexpr int NITEM = RR_ROWS * H, NIT = (NITEM + NTHR - 1) / NTHR;
  float c_prev[NIT], bia[NIT][4], addv[NIT][4];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int id = tid + it * NTHR, u = id % H;
    c_prev[it] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) { bia[it][g] = (p.bias && id < NITEM) ? p.bias[g * H + u] : 0.f; addv[it][g] = 0.f; }
  }
  // A buffer of step 0: x_0 | zeros
  for (int i = tid; i < 2 * RR_ROWS * KPAD; i += NTHR) (&abuf[0][0][0])[i] = 0.f;
  __syncthreads();
  const int xr = tid / (H / 4), xc = (tid % (H / 4)) * 4;       // x prefetch: thread -> (row, 4 columns), tid < 4 * H / 4
  const bool xload = HASX && tid < RR_ROWS * (H / 4) && r0 + xr < B;
  float4 xpre = make_float4(0.f, 0.f, 0.f, 0.f);
  if (HASX) wait_above(1u);                       // x_0 = the layer below's step 0
  if (xload) xpre = load_x(p.x + ((long)(r0 + xr)) * p.ldx + xc);
  if (HASX && tid < RR_ROWS * (H / 4)) *reinterpret_cast<float4*>(&abuf[0][xr][xc]) = xpre;
  auto load_add = [&](int t) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int id = tid + it * NTHR, row = id / H, u = id % H;
      if (id < NITEM && r0 + row < B) {
        if (TBL) {
          const float* tr = tblS + idS[row * RR_TBL_T + t] * G4 + u;
#pragma unroll
          for (int g = 0; g < 4; ++g) addv[it][g] = tr[g * H];
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) addv[it][g] = p.add[(long)t * p.add_ts + (long)(r0 + row) * G4 + g * H + u];
        }
      }
    }
  };
  if (TBL) __syncthreads();                       // table and ids are in LDS
  if (TBL || p.add) load_add(0);
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    const int cur = t & 1, nxt = cur ^ 1;
    // prefetch x_{t+1} (the gate phase stores it into the next A buffer)
    if (HASX && t + 1 < T) wait_above((uint32_t)(t + 2));
    if (xload && t + 1 < T) xpre = load_x(p.x + ((long)(t + 1) * B + r0 + xr) * p.ldx + xc);
    // ---- MFMA phase
    if (wave < 4) {
      float areg[NA];
#pragma unroll
      for (int v = 0; v < NA; ++v) {
        const int kk = 16 * v + (lane >> 2);
        areg[v] = (kk < KW) ? abuf[cur][lane & 3][KW * wave + kk] : 0.f;
      }
      f32x4 C[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) C[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      RowMma<0, KW, NG>::run(areg, W, C);
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < RR_ROWS; ++i) red[wave][i][64 * g + lane] = C[g][i];
    }
    if (PIPE && pp.mine) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the previous step's h stores (a whole MFMA phase old) are out
    rr_barrier();
    if (PIPE && pp.mine && tid == 0 && t > 0) __hip_atomic_store(pp.mine, (uint32_t)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // steps 0 .. t - 1 complete
    // ---- gate phase.  x_{t+1} (requested before the MFMA phase) goes into the next A buffer first: behind this step's stores the wait
    // for it would also wait for their write acknowledgements
    if (HASX && tid < RR_ROWS * (H / 4)) *reinterpret_cast<float4*>(&abuf[nxt][xr][xc]) = xpre;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int id = tid + it * NTHR, row = id / H, u = id % H;
      if (id >= NITEM) continue;
      float pre[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = g * H + u;
        pre[g] = ((red[0][row][n] + red[1][row][n]) + (red[2][row][n] + red[3][row][n])) + bia[it][g] + addv[it][g];
      }
      const float gi = sigm(pre[0]), gf = sigm(pre[1]), gg = tanh_fast(pre[2]), go = sigm(pre[3]);
      const float c = gf * c_prev[it] + gi * gg;
      const float h = go * tanh_fast(c);
      c_prev[it] = c;
      abuf[nxt][row][(HASX ? H : 0) + u] = h;
      if (r0 + row < B) {
        const long o = (long)t * B + r0 + row;
        if (PIPE) __hip_atomic_store(p.hs + o * p.ldh + u, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // write-through: the layer above reads it inside this launch
        else p.hs[o * p.ldh + u] = h;
        p.cs[o * H + u] = c;
        float* g4 = p.gates + o * G4 + u;
        g4[0] = gi; g4[H] = gf; g4[2 * H] = gg; g4[3 * H] = go;
      }
    }
    if ((TBL || p.add) && t + 1 < T) load_add(t + 1);
    rr_barrier();
  }
  if (PIPE && pp.mine) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(pp.mine, (uint32_t)T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int H, bool HASX, int NTHR, bool TBL = false>
__global__ __launch_bounds__(NTHR) void lstm_rowres_fwd_kernel(RowResF p) {
  lstm_rowres_fwd_body<H, HASX, NTHR, TBL, false>(p, blockIdx.x, RowPipe{nullptr, nullptr, nullptr, 0});
}

// All layers as CONCURRENT workgroups (round 4; see lstm_rowres_bwd_pipe_kernel): workgroup (l, g) runs layer l for batch rows 4 g .. 4 g + 3
// and reads the h_t that workgroup (l - 1, g) stored a step or two earlier.  Layer 0 (token table in LDS) first in the grid.  The launcher
// uses this form only when every workgroup of the grid is resident at once.
struct RowResFAll { RowResF l[MVAE_MAX_LAYERS]; int nl; };
template <int H, bool TBL0>
__global__ __launch_bounds__(320) void lstm_rowres_fwd_pipe_kernel(RowResFAll a, uint32_t* flags, uint32_t* status, uint32_t spin_limit, float* poison) {
  const int nblk = gridDim.x / a.nl;
  const int l = (int)(blockIdx.x / nblk), g = (int)(blockIdx.x % nblk);
  RowPipe pp;
  pp.above = (l > 0) ? flags + (l - 1) * nblk + g : nullptr;
  pp.mine = (l + 1 < a.nl) ? flags + l * nblk + g : nullptr;
  pp.status = status; pp.spin_limit = spin_limit; pp.poison = poison;
  if (l == 0) lstm_rowres_fwd_body<H, false, 320, TBL0, true>(a.l[0], g, pp);
  else lstm_rowres_fwd_body<H, true, 320, false, true>(a.l[l], g, pp);
}

// ---------------------------------------------------------------------------------------------------------- backward
struct RowResB {
  const float* dy; long dy_ld;       // gradient w.r.t. this layer's outputs: row (t * B + b) at dy + (t * B + b) * dy_ld
  const float* gates; const float* cs;
  const float* w_ihT; long ldw_ihT;  // [H][ldw] = W_ih^T of THIS layer (nullptr: layer 0, no input gradient)
  const float* w_hhT; long ldw_hhT;  // [H][ldw] = W_hh^T
  float* dG; long ldg;               // [T][B][ldg] pre-activation gradients (i, f, g, o)
  float* dx;                         // [T][B][H] gradient w.r.t. the layer input (HASX only)
  int T, B;
};
struct RowResBAll { RowResB l[MVAE_MAX_LAYERS]; int nl; };

// One step: gate phase (dG_t from dh_t, dc carry) -> barrier -> MFMA phase: [dx_t | dh_rec_{t-1}] = dG_t . [W_ih | W_hh], wave w
// contracts gate w's 72 columns -> partial sums to LDS -> barrier.  The partials are summed by the NEXT gate phase.
// ALL layers of the stack run inside ONE launch, top layer first: layer l's input gradient dx (scratch) is layer l-1's dy, and a
// workgroup only ever reads the dx rows it wrote itself, so an agent-scope fence + barrier between layers is all the ordering needed.
// One launch keeps the 128 workgroups resident for the whole pass: they do not have to win their CUs back from concurrently
// running GEMM workgroups at every layer boundary.  Layer 0 has no input gradient (hasx = false: zero W_ih fragments, no dx stores).
// HASX = false (layer 0: no input gradient): the MFMA output is dh_rec alone -- H columns = 2 groups of 64 instead of 3, a third fewer MFMAs.
// (RowPipe, backward: `above` = progress word of the workgroup that produces our dy -- the layer above, same batch rows --, progress = number
// of completed time steps, T + 1 once dx[0] is out: dx[s] is visible once progress >= T - s + 1)
template <int H, int NTHR, bool HASX, bool PIPE = false>
__device__ __forceinline__ void rowres_bwd_layer(const RowResB& p, float (&gbuf)[RR_ROWS][4 * H + 24], float (&red)[4][RR_ROWS][192], int rowgroup,
                                                 RowPipe pp = RowPipe{nullptr, nullptr, nullptr, 0}) {
  constexpr int G4 = 4 * H, KW = H, NOUT = HASX ? 2 * H : H, NG = (NOUT + 63) / 64, NA = (KW + 15) / 16;
  constexpr int GPAD = G4 + 24, NPAD = NG * 64, RO = HASX ? H : 0;       // RO: column of dh_rec inside the MFMA output
  static_assert((GPAD % 32) == 24 && NPAD <= 192, "A-buffer rows on disjoint banks");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = rowgroup * RR_ROWS, B = p.B, T = p.T;
  // PIPE: wait (bounded) until the producer of our dy has completed `need` steps; every thread polls for itself (it then loads its own dy
  // elements with agent-scope loads), and only when its last look at the word does not cover the request -- a consumer that has fallen a few
  // steps behind polls once for several steps
  uint32_t known = 0;
  auto wait_above = [&](uint32_t need) {
    if (!PIPE || pp.above == nullptr || known >= need) return;
    for (uint32_t it = 0; it < pp.spin_limit; ++it) {
      known = __hip_atomic_load(pp.above, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (known >= need) return;
      __builtin_amdgcn_s_sleep(4);
    }
    if (pp.status && atomicCAS(pp.status, 0u, 2u) == 0u && pp.poison) *pp.poison = __builtin_nanf("");          // gave up: the results are invalid, the launch still ends
    known = 0xffffffffu;
  };
  float W[NG][KW];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int n = 64 * g + lane;
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
      const int k = KW * (wave & 3) + kk;
      float v = 0.f;
      if (n < NOUT && wave < 4) {
        if (HASX && n < H) v = p.w_ihT[(long)n * p.ldw_ihT + k];
        else v = p.w_hhT[(long)(n - RO) * p.ldw_hhT + k];
      }
      W[g][kk] = v;
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);           // weight registers complete before the loop (see the forward kernel)
  constexpr int NITEM = RR_ROWS * H, NIT = (NITEM + NTHR - 1) / NTHR;
  float dc_carry[NIT], c_cur[NIT], c_prev[NIT], gt[NIT][4], dyv[NIT];
  auto prefetch = [&](int t) {       // gates[t], c[t-1], dy[t]
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int id = tid + it * NTHR, row = id / H, u = id % H;
      if (id < NITEM && r0 + row < B) {
        const long o = (long)t * B + r0 + row;
        const float* g4 = p.gates + o * G4 + u;
        gt[it][0] = g4[0]; gt[it][1] = g4[H]; gt[it][2] = g4[2 * H]; gt[it][3] = g4[3 * H];
        c_prev[it] = (t > 0) ? p.cs[(o - B) * H + u] : 0.f;
        if (PIPE) dyv[it] = __hip_atomic_load(p.dy + o * p.dy_ld + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // sc1: not through this CU's L1
        else dyv[it] = p.dy[o * p.dy_ld + u];
      }
    }
  };
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int id = tid + it * NTHR, row = id / H, u = id % H;
    dc_carry[it] = 0.f; c_prev[it] = 0.f; dyv[it] = 0.f; c_cur[it] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) gt[it][g] = 0.f;
    if (id < NITEM && r0 + row < B) c_cur[it] = p.cs[((long)(T - 1) * B + r0 + row) * H + u];
  }
  wait_above(2u);                                                    // dy[T - 1] = the producer's dx[T - 1]: out after its first two steps
  prefetch(T - 1);
  for (int t = T - 1; t >= 0; --t) {
    // ---- gate phase
    float gi[NIT], gf[NIT], gg[NIT], go[NIT], cp[NIT], cc[NIT], dyt[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) { gi[it] = gt[it][0]; gf[it] = gt[it][1]; gg[it] = gt[it][2]; go[it] = gt[it][3]; cp[it] = c_prev[it]; cc[it] = c_cur[it]; dyt[it] = dyv[it]; }
    if (t > 0) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) c_cur[it] = c_prev[it];      // c_{t-1} is the next step's cell state
      wait_above((uint32_t)(T - t + 2));                             // dy[t - 1]
      prefetch(t - 1);                                               // in flight under this step's math and MFMAs
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int id = tid + it * NTHR, row = id / H, u = id % H;
      if (id >= NITEM) continue;
      float dh = dyt[it];
      if (t < T - 1) {
        dh += (red[0][row][RO + u] + red[1][row][RO + u]) + (red[2][row][RO + u] + red[3][row][RO + u]);
        if (HASX && r0 + row < B) {
          const float v = (red[0][row][u] + red[1][row][u]) + (red[2][row][u] + red[3][row][u]);
          if (PIPE) __hip_atomic_store(p.dx + ((long)(t + 1) * B + r0 + row) * H + u, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
          else p.dx[((long)(t + 1) * B + r0 + row) * H + u] = v;
        }
      }
      const float tc = tanh_fast(cc[it]);
      const float d_o = dh * tc;
      const float dc = dh * go[it] * (1.f - tc * tc) + dc_carry[it];
      dc_carry[it] = dc * gf[it];
      const float di = dc * gg[it] * gi[it] * (1.f - gi[it]);
      const float df = dc * cp[it] * gf[it] * (1.f - gf[it]);
      const float dg = dc * gi[it] * (1.f - gg[it] * gg[it]);
      const float dO = d_o * go[it] * (1.f - go[it]);
      gbuf[row][u] = di; gbuf[row][H + u] = df; gbuf[row][2 * H + u] = dg; gbuf[row][3 * H + u] = dO;
      if (r0 + row < B) {
        float* d4 = p.dG + ((long)t * B + r0 + row) * p.ldg + u;
        d4[0] = di; d4[H] = df; d4[2 * H] = dg; d4[3 * H] = dO;
      }
    }
    rr_barrier();
    // ---- MFMA phase: wave w contracts gate w
    if (wave < 4) {
      float areg[NA];
#pragma unroll
      for (int v = 0; v < NA; ++v) {
        const int kk = 16 * v + (lane >> 2);
        areg[v] = (kk < KW) ? gbuf[lane & 3][KW * wave + kk] : 0.f;
      }
      f32x4 C[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) C[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      RowMma<0, KW, NG>::run(areg, W, C);
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < RR_ROWS; ++i) red[wave][i][64 * g + lane] = C[g][i];
    }
    if (PIPE && pp.mine) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this step's dx stores (issued a whole MFMA phase ago) are out
    rr_barrier();
    if (PIPE && pp.mine && tid == 0) __hip_atomic_store(pp.mine, (uint32_t)(T - t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (HASX) {                                                       // input gradient of step 0
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int id = tid + it * NTHR, row = id / H, u = id % H;
      if (id < NITEM && r0 + row < B) {
        const float v = (red[0][row][u] + red[1][row][u]) + (red[2][row][u] + red[3][row][u]);
        if (PIPE) __hip_atomic_store(p.dx + ((long)(r0 + row)) * H + u, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else p.dx[((long)(r0 + row)) * H + u] = v;
      }
    }
    if (PIPE && pp.mine) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(pp.mine, (uint32_t)(T + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}


template <int H>
__global__ __launch_bounds__(256) void lstm_rowres_bwd_all_kernel(RowResBAll a) {
  __shared__ float gbuf[RR_ROWS][4 * H + 24];         // dG_t of the 4 rows (A operand)
  __shared__ float red[4][RR_ROWS][192];
  for (int l = a.nl - 1; l >= 0; --l) {
    if (l > 0) rowres_bwd_layer<H, 256, true>(a.l[l], gbuf, red, blockIdx.x);
    else rowres_bwd_layer<H, 256, false>(a.l[l], gbuf, red, blockIdx.x);
    if (l > 0) { __threadfence(); __syncthreads(); }  // our dx rows are visible to the loads of the next layer; gbuf / red are free again
  }
}

// The same layers as CONCURRENT workgroups (round 4): workgroup (l, g) runs layer l for batch rows 4 g .. 4 g + 3 and consumes the input
// gradient dx[t] that workgroup (l + 1, g) produced a couple of steps earlier -- a pipeline over the layers with one progress word per
// workgroup (write-through dx stores, drained before the step's second barrier; agent-scope dy loads; bounded polls) instead of one layer
// after the other: the pass takes ~T + 2 (layers - 1) steps instead of layers x T.  The top layer comes first in the grid (dispatch order is
// not promised: the launcher only uses this form when EVERY workgroup of the grid is resident at once, so nobody waits for a queued one).
template <int H>
__global__ __launch_bounds__(256) void lstm_rowres_bwd_pipe_kernel(RowResBAll a, uint32_t* flags, uint32_t* status, uint32_t spin_limit, float* poison) {
  __shared__ float gbuf[RR_ROWS][4 * H + 24];
  __shared__ float red[4][RR_ROWS][192];
  const int nblk = gridDim.x / a.nl;
  const int l = a.nl - 1 - (int)(blockIdx.x / nblk), g = (int)(blockIdx.x % nblk);
  RowPipe pp;
  pp.above = (l + 1 < a.nl) ? flags + (l + 1) * nblk + g : nullptr;
  pp.mine = (l > 0) ? flags + l * nblk + g : nullptr;
  pp.status = status; pp.spin_limit = spin_limit; pp.poison = poison;
  if (l > 0) rowres_bwd_layer<H, 256, true, true>(a.l[l], gbuf, red, g, pp);
  else rowres_bwd_layer<H, 256, false, true>(a.l[l], gbuf, red, g, pp);
}

// =====================================================================================================================
// Row-resident GRU for ONE-layer bf16 stacks of H = 256 whose input projection is a token table (the MOSES encoder: GRU(V -> 256),
// mosesvae.py:142-156 with the embedding folded into tbl[V][4H]).  The wavefront schedule pays one launch per time step for
// 0.4 GFLOP (8 us each, 59 + 59 launches of a 6 ms step).  Here a workgroup of 8 waves owns 4 batch rows (one per accumulator-row group of the 16-row MFMA tile; the other rows
// are zero padding: the MFMA rate is not what a step waits for, registers per lane are) for the whole sequence:
//   * W_hh (768 x 256 bf16 = 393 KB) lives in REGISTERS for the whole pass: wave w holds, for hidden units 32 w .. 32 w + 31, the rows of
//     its r and z gates as B fragments of v_mfma_f32_16x16x32_bf16 (2 gates x 2 unit tiles x 8 K-blocks x 4 VGPRs = 128 VGPRs); the n gate's
//     fragments (W_hn, 128 KB) sit in LDS in fragment order and stream through registers with the A fragments;
//   * h_{t-1} of the 16 rows sits in LDS as bf16 (the A operand, 8 ds_read_b128 per wave and step), double-buffered: ONE barrier per step;
//   * the accumulator layout gives every lane rows 4 q .. 4 q + 3 of ONE hidden unit for all three gates, so the whole cell update
//     (table row + bias, sigmoid / tanh, length mask, fp32 state) stays in the lane's registers.
// Gate slots, masks and saved tensors exactly as lstm_step_fwd_kernel's GRU branch (rnn.hip): slots [r | z | W_in x + b_in | W_hn h + b_hn],
// a finished row keeps its state and emits zeros.
constexpr int GR_ROWS = 16;        // MFMA rows of the A tile
constexpr int GR_RPL = 1;          // real batch rows per lane (of its 4 accumulator rows): 4 batch rows per workgroup (B = 1024: one workgroup per CU)
__device__ __forceinline__ bf16_t f2bf_t(float f) { bf16_t r; r.x = f2bf(f); return r; }
// the bf16 step kernels' activation forms (rnn.hip act_sigmoid<bf16_t> / act_tanh<bf16_t>): same numbers on either schedule
__device__ __forceinline__ float act_sigmoid_bf(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float act_tanh_bf(float x) { return fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)), 1.f); }
struct GruRowF {
  const bf16_t* w_hh; long ldw;      // [4H][ldw] packed 4-slot (slot 2 rows are a zero block and never read)
  const float* bias;                 // [4H] or nullptr
  const float* tbl; const long* idx; long idx_ld; int tbl_rows;      // addend of step t, row b: tbl[idx[b * idx_ld + t]][4H] (slots 0-2)
  const int* lengths;                // [B] or nullptr
  const bf16_t* h0; long ldh0;       // [B][ldh0] or nullptr (zeros)
  bf16_t* hs; long ldh;              // out [T][B][ldh]
  bf16_t* gates;                     // out [T][B][4H] (r, z, n, W_hn h + b_hn) or nullptr
  float* hstate;                     // out [2][B][H]: slot (T - 1) & 1 receives the final state
  int T, B;
  int safe;                          // != 0 (MVAE_GRU_ROWRES_SAFE, tests): every counted wait becomes vmcnt(0) -- the reference form the counted one must equal bit for bit
};

// Loads of the time loops as inline asm: the compiler's wait-count pass loses the exact queue state at a loop's back edge and drains the
// whole queue -- this step's table rows AND the previous step's stores (their acknowledgements: ~1 us) -- where a counted wait for the loads
// alone is enough.  The memory queue is in order, so "everything older than the newest N operations has completed" is exact once the number
// of stores per step is fixed (host: B a multiple of the rows per workgroup; template GATES).
// The destination is a read-write operand: the variable keeps ONE register across the asm, so the compiler never has a reason to copy a value
// that has not arrived yet (as the phi of a conditional or loop-carried "=v" result would: measured the hard way on the f32 LSTM loops).
template <int OFF> __device__ __forceinline__ void gload_f32(float& dst, const float* p) {
  asm volatile("global_load_dword %0, %1, off offset:%2" : "+v"(dst) : "v"(p), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void gload_u16(uint32_t& dst, const bf16_t* p) {
  asm volatile("global_load_ushort %0, %1, off offset:%2" : "+v"(dst) : "v"(p), "n"(OFF) : "memory");
}
template <int NL, int NS> __device__ __forceinline__ void wait_vm_case(bool next_issued, bool prev_stores) {
  // operands of the current step are older than: this step's NL loads (if issued) and the previous step's NS stores (if any)
  __builtin_amdgcn_sched_barrier(0);
  if (next_issued) { if (prev_stores) wait_vmcnt<NL + NS>(); else wait_vmcnt<NL>(); }
  else { if (prev_stores) wait_vmcnt<NS>(); else wait_vmcnt<0>(); }
  __builtin_amdgcn_sched_barrier(0);
}
// K loop of one step: fragment set kb & 1 holds the A fragment (h_{t-1}) and the W_hn fragments of K-block kb; the reads of block kb + 1
// are in flight under the MFMAs of block kb (three ds_read_b128 per block: the counted wait lets exactly those three stay outstanding)
template <int KB_, int KBLK, int UT> struct GruFrag {
  static __device__ __forceinline__ void run(u32x4 (&af)[2], u32x4 (&wn)[2][UT], uint32_t ab, uint32_t wnb, const uint4 (&wf)[2][UT][KBLK], f32x4 (&acc)[3][UT]) {
    constexpr int c = KB_ & 1, n = c ^ 1;
    if constexpr (KB_ + 1 < KBLK) {
      af[n] = lds_read128<(KB_ + 1) * 64>(ab);
#pragma unroll
      for (int j = 0; j < UT; ++j) wn[n][j] = lds_read128<0>(wnb + (uint32_t)((j * KBLK + KB_ + 1) * 1024));
      wait_lgkmcnt<1 + UT>();
    } else wait_lgkmcnt<0>();
#pragma unroll
    for (int j = 0; j < UT; ++j) {
      mma16<bf16_t>(__builtin_bit_cast(uint4, af[c]), wf[0][j][KB_], acc[0][j]);
      mma16<bf16_t>(__builtin_bit_cast(uint4, af[c]), wf[1][j][KB_], acc[1][j]);
      mma16<bf16_t>(__builtin_bit_cast(uint4, af[c]), __builtin_bit_cast(uint4, wn[c][j]), acc[2][j]);
    }
    GruFrag<KB_ + 1, KBLK, UT>::run(af, wn, ab, wnb, wf, acc);
  }
};
template <int KBLK, int UT> struct GruFrag<KBLK, KBLK, UT> {
  static __device__ __forceinline__ void run(u32x4 (&)[2], u32x4 (&)[2][UT], uint32_t, uint32_t, const uint4 (&)[2][UT][KBLK], f32x4 (&)[3][UT]) {}
};
template <int KB0, int KBLK, int UT>
__device__ __forceinline__ void gru_frag_step(u32x4 (&af)[2], u32x4 (&wn)[2][UT], uint32_t ab, uint32_t wnb, const uint4 (&wf)[2][UT][KBLK], f32x4 (&acc)[3][UT]) {
  af[0] = lds_read128<0>(ab);
#pragma unroll
  for (int j = 0; j < UT; ++j) wn[0][j] = lds_read128<0>(wnb + (uint32_t)(j * KBLK * 1024));
  GruFrag<0, KBLK, UT>::run(af, wn, ab, wnb, wf, acc);
}

template <int H, bool GATES>
__global__ __launch_bounds__(512) void gru_rowres_fwd_kernel(GruRowF p) {
  static_assert(H == 256, "8 waves x 32 hidden units");
  constexpr int KBLK = H / 32, LDA = H + 8, UT = 2, RPL = GR_RPL;          // A rows padded by 16 bytes: the 16 rows of a fragment read start on different banks
  constexpr int WN_BYTES = (H / 16) * KBLK * 1024;           // the n gate's W_hn fragments: [unit tile][K-block][lane] x 16 bytes = 128 KB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t (*abuf)[GR_ROWS][LDA] = reinterpret_cast<bf16_t (*)[GR_ROWS][LDA]>(smem + WN_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lc = lane & 15, lq = lane >> 4;
  const int r0 = blockIdx.x * (4 * RPL), B = p.B, T = p.T;   // RPL of the 4 accumulator rows of a lane are real batch rows (registers: gx / state per row)
  const int u0 = 32 * wave + lc;                              // this lane's hidden units: u0 and u0 + 16; its batch row: r0 + lq
  // ---- weights: fragment (gate s, unit tile j, K-block kb) = W[slot_s * H + u0 + 16 j][32 kb + 8 lq .. + 8].
  // r and z stay in registers (128 VGPRs); the n gate's go to LDS in fragment order (a wave reads back only what it wrote itself)
  uint4 wf[2][UT][KBLK];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int j = 0; j < UT; ++j) {
      const bf16_t* wr = p.w_hh + (long)((s == 2 ? 3 : s) * H + u0 + 16 * j) * p.ldw + 8 * lq;
#pragma unroll
      for (int kb = 0; kb < KBLK; ++kb) {
        const uint4 v = *reinterpret_cast<const uint4*>(wr + 32 * kb);
        if (s < 2) wf[s][j][kb] = v;
        else *reinterpret_cast<uint4*>(smem + (((2 * wave + j) * KBLK + kb) * 64 + lane) * 16) = v;
      }
    }
  float* bias_s = reinterpret_cast<float*>(smem + WN_BYTES + 2 * GR_ROWS * LDA * 2);      // [4][H] in LDS: read back per step (registers are short)
  for (int i = tid; i < 4 * H; i += 512) bias_s[i] = p.bias ? p.bias[i] : 0.f;
  int len[RPL]; float hst[RPL][UT];
  const int rbase = r0 + RPL * lq;                            // rows rbase .. rbase + RPL - 1 (clamped to B - 1 for loads, never stored beyond B)
#pragma unroll
  for (int i = 0; i < RPL; ++i) {
    const int rc = (rbase + i < B) ? rbase + i : B - 1;
    len[i] = p.lengths ? p.lengths[rc] : T;
#pragma unroll
    for (int j = 0; j < UT; ++j) hst[i][j] = p.h0 ? bf2f(p.h0[(long)rc * p.ldh0 + u0 + 16 * j].x) : 0.f;
  }
  // token ids of the workgroup's rows for every step, clamped, in LDS (a global load inside the time loop would sit in the same in-order
  // memory queue as the step's stores: waiting for it waits for their acknowledgements)
  int* tok_s = reinterpret_cast<int*>(smem + WN_BYTES + 2 * GR_ROWS * LDA * 2 + 4 * H * 4);      // [4 RPL][T]
  for (int i = tid; i < 4 * RPL * T; i += 512) {
    const int r = i / T, t = i - r * T;
    const int rc = (r0 + r < B) ? r0 + r : B - 1;
    const long id = p.idx[(long)rc * p.idx_ld + t];
    tok_s[i] = (int)(id < 0 ? 0 : (id >= p.tbl_rows ? p.tbl_rows - 1 : id));
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0): the weight registers are complete before the time loop (see lstm_rowres_fwd_kernel)
  for (int i = tid; i < 2 * GR_ROWS * LDA / 2; i += 512) reinterpret_cast<uint32_t*>(&abuf[0][0][0])[i] = 0u;      // padding rows stay zero
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RPL; ++i)
#pragma unroll
    for (int j = 0; j < UT; ++j) abuf[0][4 * lq + i][u0 + 16 * j] = f2bf_t(hst[i][j]);      // A buffer of step 0: h_0
  typedef float Gx[RPL][3][UT];
  static_assert(RPL == 1 && UT == 2, "load / store counts of the counted waits below");
  constexpr int NLD = 6, NST = GATES ? 10 : 2;                // vector loads / stores of one step
  auto load_gx = [&](int t, Gx& g) {                          // table rows of step t for the lane's row / units (asm: see gload_f32)
    const float* tr = p.tbl + (long)tok_s[lq * T + t] * 4 * H + u0;
    gload_f32<0>(g[0][0][0], tr); gload_f32<64>(g[0][0][1], tr);
    gload_f32<4 * H>(g[0][1][0], tr); gload_f32<4 * H + 64>(g[0][1][1], tr);
    gload_f32<8 * H>(g[0][2][0], tr); gload_f32<8 * H + 64>(g[0][2][1], tr);
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const uint32_t a_lane = lds0 + (uint32_t)WN_BYTES + (uint32_t)(lc * LDA * 2 + lq * 16);
  const uint32_t wn_lane = lds0 + (uint32_t)((2 * wave * KBLK * 64 + lane) * 16);
  const uint32_t hs_ts = (uint32_t)B * (uint32_t)p.ldh, g_ts = (uint32_t)B * 4u * H;      // elements per time step
  // one time step: `gx` holds this step's table rows (requested a step ago), `gxn` receives the next step's -- requested FIRST, in front of
  // this step's stores, so that the wait for them next step never includes a store acknowledgement
  auto step = [&](int t, Gx& gx, Gx& gxn) {
    const int cur = t & 1, nxt = cur ^ 1;
    if (t + 1 < T) load_gx(t + 1, gxn);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[3][UT];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int j = 0; j < UT; ++j) acc[s][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t ab = a_lane + (uint32_t)(cur * GR_ROWS * LDA * 2);
    u32x4 af[2], wn[2][UT];
    gru_frag_step<0, KBLK, UT>(af, wn, ab, wn_lane, wf, acc);
    if (p.safe) wait_vmcnt<0>(); else
    wait_vm_case<NLD, NST>(t + 1 < T, t > 0);                 // this step's table rows have arrived
#pragma unroll
    for (int i = 0; i < RPL; ++i) asm volatile("" : "+v"(gx[i][0][0]), "+v"(gx[i][0][1]), "+v"(gx[i][1][0]), "+v"(gx[i][1][1]), "+v"(gx[i][2][0]), "+v"(gx[i][2][1]));
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
      const int row = rbase + i;
      const bool valid = t < len[i];
#pragma unroll
      for (int j = 0; j < UT; ++j) {
        const int u = u0 + 16 * j;
        const float pr = (acc[0][j][i] + bias_s[u]) + gx[i][0][j];
        const float pz = (acc[1][j][i] + bias_s[H + u]) + gx[i][1][j];
        const float pn = (0.f + bias_s[2 * H + u]) + gx[i][2][j];
        const float ph = acc[2][j][i] + bias_s[3 * H + u];
        const float gr = act_sigmoid_bf(pr), gz = act_sigmoid_bf(pz);
        const float gn = act_tanh_bf(pn + gr * ph);
        const float hn = (1.f - gz) * gn + gz * hst[i][j];
        hst[i][j] = valid ? hn : hst[i][j];
        const float hseq = valid ? hn : 0.f;
        abuf[nxt][4 * lq + i][u] = f2bf_t(hst[i][j]);
        // (host: B is a multiple of the 4 rows of a workgroup -- every row is real, every store happens: the counted waits rely on it)
        // 32-bit element offsets from uniform bases (host: both tensors < 2^31 elements): scalar base + one offset register per store
        p.hs[(uint32_t)t * hs_ts + (uint32_t)row * (uint32_t)p.ldh + (uint32_t)u] = f2bf_t(hseq);
        if constexpr (GATES) {
          const uint32_t go = (uint32_t)t * g_ts + (uint32_t)row * 4u * H + (uint32_t)u;
          p.gates[go] = f2bf_t(gr); p.gates[go + H] = f2bf_t(gz); p.gates[go + 2 * H] = f2bf_t(gn); p.gates[go + 3 * H] = f2bf_t(ph);
        }
      }
    }
    wait_lgkmcnt<0>();                                        // my LDS writes of h_t are done
    ws_barrier();                                             // (no vmcnt wait: the stores of this step stay in flight)
  };
  __syncthreads();
  Gx gxa = {}, gxb = {};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the counted waits start from an empty queue
  load_gx(0, gxa);
  for (int t = 0; t < T; t += 2) {                            // two steps per iteration: the register sets swap roles, nothing is copied
    step(t, gxa, gxb);
    if (t + 1 < T) step(t + 1, gxb, gxa);
  }
#pragma unroll
  for (int i = 0; i < RPL; ++i) {
    const int row = rbase + i;
#pragma unroll
    for (int j = 0; j < UT; ++j)
      if (row < B) p.hstate[((long)((T - 1) & 1) * B + row) * H + u0 + 16 * j] = hst[i][j];
  }
}

// ---- backward of the same stack.  Walking t = T-1 .. 0 a workgroup keeps, for its 4 batch rows: the carried gradient dh in registers (fp32),
// dG_{t+1}[r | z | n*r] as bf16 in LDS (the A operand of  dh_t += dG_{t+1} . W_hh, K = 3H: the zero gate-slot block is never stored), W_hh^T
// for the wave's 32 output units as B fragments -- K-blocks of the r and z slots in registers (128 VGPRs), of the n*r slot in LDS (128 KB, fragment
// order).  Gate-derivative math and masks exactly as bwd_cell_finish's GRU branch (rnn.hip); dG[t] goes to global for the weight-gradient GEMMs.
struct GruRowB {
  const bf16_t* w_hhT; long ldwT;    // [H][ldwT]: W_hh^T, columns = 4 gate slots x H
  const bf16_t* gates;               // [T][B][4H] saved r, z, n, W_hn h + b_hn
  const bf16_t* hs; long ldh;        // [T][B][ldh] forward outputs (h_{t-1} of step t = hs[t-1])
  const bf16_t* h0; long ldh0;       // [B][ldh0] or nullptr
  const float* dh_last;              // [B][H] gradient w.r.t. the final state, or nullptr
  const int* lengths;
  bf16_t* dG; long ldg;              // out [T][B][ldg], 4 slots
  int T, B;
  int safe;                          // as GruRowF::safe
};
constexpr int GRB_LDG = 3 * 256 + 8;

template <int KB_, int NREG, int NK, int UT> struct GruFragB {
  // K-blocks 0 .. NREG-1: W fragments in registers; NREG .. NK-1: streamed from LDS beside the A fragment
  static __device__ __forceinline__ void run(u32x4 (&af)[2], u32x4 (&wl)[2][UT], uint32_t ab, uint32_t wlb, const uint4 (&wf)[UT][NREG], f32x4 (&acc)[UT]) {
    constexpr int c = KB_ & 1, n = c ^ 1;
    if constexpr (KB_ + 1 < NK) {
      af[n] = lds_read128<(KB_ + 1) * 64>(ab);
      if constexpr (KB_ + 1 >= NREG) {
#pragma unroll
        for (int j = 0; j < UT; ++j) wl[n][j] = lds_read128<0>(wlb + (uint32_t)((j * (NK - NREG) + KB_ + 1 - NREG) * 1024));
        wait_lgkmcnt<1 + UT>();
      } else wait_lgkmcnt<1>();
    } else wait_lgkmcnt<0>();
#pragma unroll
    for (int j = 0; j < UT; ++j) {
      if constexpr (KB_ < NREG) mma16<bf16_t>(__builtin_bit_cast(uint4, af[c]), wf[j][KB_ < NREG ? KB_ : 0], acc[j]);
      else mma16<bf16_t>(__builtin_bit_cast(uint4, af[c]), __builtin_bit_cast(uint4, wl[c][j]), acc[j]);
    }
    GruFragB<KB_ + 1, NREG, NK, UT>::run(af, wl, ab, wlb, wf, acc);
  }
};
template <int NREG, int NK, int UT> struct GruFragB<NK, NREG, NK, UT> {
  static __device__ __forceinline__ void run(u32x4 (&)[2], u32x4 (&)[2][UT], uint32_t, uint32_t, const uint4 (&)[UT][NREG], f32x4 (&)[UT]) {}
};

template <int H>
__global__ __launch_bounds__(512) void gru_rowres_bwd_kernel(GruRowB p) {
  static_assert(H == 256, "8 waves x 32 hidden units");
  constexpr int KBH = H / 32, NK = 3 * KBH, NREG = 2 * KBH, UT = 2, RPL = GR_RPL, NROW = 4 * RPL, LDG = GRB_LDG;
  constexpr int WL_BYTES = (H / 16) * KBH * 1024;             // the n*r slot's W fragments: [unit tile][K-block][lane] x 16 bytes = 128 KB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t (*gbuf)[NROW][LDG] = reinterpret_cast<bf16_t (*)[NROW][LDG]>(smem + WL_BYTES);      // dG_{t+1}[r | z | n*r] of the 8 rows, double-buffered
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lc = lane & 15, lq = lane >> 4;
  const int r0 = blockIdx.x * NROW, B = p.B, T = p.T;
  const int u0 = 32 * wave + lc;
  // ---- W_hh^T fragments: (unit tile j, K-block kb) = W_hhT[u0 + 16 j][kcol(kb) + 8 lq .. + 8], kcol = slot r / z / n*r (gate slots 0, 1, 3)
  uint4 wf[UT][NREG];
#pragma unroll
  for (int j = 0; j < UT; ++j) {
    const bf16_t* wr = p.w_hhT + (long)(u0 + 16 * j) * p.ldwT + 8 * lq;
#pragma unroll
    for (int kb = 0; kb < NK; ++kb) {
      const int kcol = (kb < KBH ? 0 : kb < 2 * KBH ? H : 3 * H) + 32 * (kb % KBH);
      const uint4 v = *reinterpret_cast<const uint4*>(wr + kcol);
      if (kb < NREG) wf[j][kb < NREG ? kb : 0] = v;
      else *reinterpret_cast<uint4*>(smem + (((2 * wave + j) * KBH + (kb - NREG)) * 64 + lane) * 16) = v;
    }
  }
  int len[RPL]; float dh[RPL][UT];
  const int rbase = r0 + RPL * lq;
#pragma unroll
  for (int i = 0; i < RPL; ++i) {
    const int rc = (rbase + i < B) ? rbase + i : B - 1;
    len[i] = p.lengths ? p.lengths[rc] : T;
#pragma unroll
    for (int j = 0; j < UT; ++j) dh[i][j] = p.dh_last ? p.dh_last[(long)rc * H + u0 + 16 * j] : 0.f;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0): the weight registers are complete before the time loop
  for (int i = tid; i < 2 * NROW * LDG / 2; i += 512) reinterpret_cast<uint32_t*>(&gbuf[0][0][0])[i] = 0u;      // dG_T = 0
  // operands of step t: saved gates of the lane's row / two unit tiles and h_{t-1}.  Ten asm loads per step, always (see gload_f32): with no
  // initial state the t = 0 read goes to the gates row instead and is replaced by zero
  static_assert(RPL == 1 && UT == 2, "load / store counts of the counted waits below");
  constexpr int NLD = 10, NST = 8;
  struct Ops { uint32_t g[UT][4]; uint32_t hp[UT]; };
  const int rc0 = (rbase < B) ? rbase : B - 1;
  auto load_ops = [&](int t, Ops& o) {
    const bf16_t* g4 = p.gates + ((long)t * B + rc0) * 4 * H + u0;
    const bf16_t* hp = (t > 0) ? p.hs + ((long)(t - 1) * B + rc0) * p.ldh + u0 : (p.h0 ? p.h0 + (long)rc0 * p.ldh0 + u0 : g4);
    gload_u16<0>(o.g[0][0], g4); gload_u16<32>(o.g[1][0], g4);
    gload_u16<2 * H>(o.g[0][1], g4); gload_u16<2 * H + 32>(o.g[1][1], g4);
    gload_u16<4 * H>(o.g[0][2], g4); gload_u16<4 * H + 32>(o.g[1][2], g4);
    gload_u16<6 * H>(o.g[0][3], g4); gload_u16<6 * H + 32>(o.g[1][3], g4);
    gload_u16<0>(o.hp[0], hp); gload_u16<32>(o.hp[1], hp);
  };
  __syncthreads();
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const int rr = ((lc & 3) < RPL) ? (lc >> 2) * RPL + (lc & 3) : 0;      // MFMA tile row lc -> real row (padding rows alias row 0: their results are unused)
  const uint32_t a_lane = lds0 + (uint32_t)WL_BYTES + (uint32_t)(rr * LDG * 2 + lq * 16);
  const uint32_t wl_lane = lds0 + (uint32_t)((2 * wave * KBH * 64 + lane) * 16);
  // one time step: `o` holds this step's operands (requested a step ago), `on` receives the next step's -- requested first, in front of this
  // step's stores (the memory queue is in order: a load behind a store cannot be waited for without the store's acknowledgement)
  auto step = [&](int t, int k, Ops& o, Ops& on) {
    const int cur = k & 1, nxt = cur ^ 1;
    if (t > 0) load_ops(t - 1, on);
    __builtin_amdgcn_sched_barrier(0);
    // ---- dh_t += dG_{t+1}[r | z | n*r] . W_hh  (zero at t = T - 1)
    f32x4 acc[UT];
#pragma unroll
    for (int j = 0; j < UT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t ab = a_lane + (uint32_t)(cur * NROW * LDG * 2);
    u32x4 af[2], wl[2][UT];
    af[0] = lds_read128<0>(ab);
    GruFragB<0, NREG, NK, UT>::run(af, wl, ab, wl_lane, wf, acc);
    if (p.safe) wait_vmcnt<0>(); else
    wait_vm_case<NLD, NST>(t > 0, k > 0);                     // this step's operands have arrived
#pragma unroll
    for (int j = 0; j < UT; ++j) asm volatile("" : "+v"(o.g[j][0]), "+v"(o.g[j][1]), "+v"(o.g[j][2]), "+v"(o.g[j][3]), "+v"(o.hp[j]));
    const bool has_hp = t > 0 || p.h0 != nullptr;
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
      const int row = rbase + i;
      const bool valid = t < len[i];
#pragma unroll
      for (int j = 0; j < UT; ++j) {
        const int u = u0 + 16 * j;
        const float d = dh[i][j] + acc[j][i];
        float dpr = 0.f, dpz = 0.f, dpn = 0.f, dpnr = 0.f, carry = d;
        if (valid) {
          const float gr = bf2f((uint16_t)o.g[j][0]), gz = bf2f((uint16_t)o.g[j][1]), gn = bf2f((uint16_t)o.g[j][2]), nh = bf2f((uint16_t)o.g[j][3]);
          const float hp = has_hp ? bf2f((uint16_t)o.hp[j]) : 0.f;
          const float dn = d * (1.f - gz);
          const float dz = d * (hp - gn);
          dpn = dn * (1.f - gn * gn);
          dpr = dpn * nh * gr * (1.f - gr);
          dpz = dz * gz * (1.f - gz);
          dpnr = dpn * gr;
          carry = d * gz;
        }
        dh[i][j] = carry;
        const bf16_t br = f2bf_t(dpr), bz = f2bf_t(dpz), bnr = f2bf_t(dpnr);
        bf16_t* gl = &gbuf[nxt][RPL * lq + i][u];
        gl[0] = br; gl[H] = bz; gl[2 * H] = bnr;
        {     // (host: B is a multiple of the 4 rows of a workgroup: every store happens -- the counted waits rely on it)
          bf16_t* d4 = p.dG + ((long)t * B + row) * p.ldg + u;
          d4[0] = br; d4[H] = bz; d4[2 * H] = f2bf_t(dpn); d4[3 * H] = bnr;
        }
      }
    }
    wait_lgkmcnt<0>();
    ws_barrier();
  };
  Ops oa = {}, ob = {};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the counted waits start from an empty queue
  load_ops(T - 1, oa);
  for (int t = T - 1, k = 0; t >= 0; t -= 2, k += 2) {        // two steps per iteration: the operand sets swap roles, nothing is copied
    step(t, k, oa, ob);
    if (t >= 1) step(t - 1, k + 1, ob, oa);
  }
}

}  // namespace

// layers of one pass, sequentially; returns MVAE_ERR_UNSUPPORTED when the shape is not the one this schedule is built for
// scratch of the layer-concurrent forward (0: the shape / device is not served by it): [status 16 words | layers x ceil(B / 4) progress words]
size_t rnn_rowres_fwd_pipe_workspace(const mvae_rnn_fwd_desc* d) {
  if (d->cell != MVAE_CELL_LSTM || d->dtype != MVAE_F32 || d->H != 72 || d->x0 || d->lengths || d->layers < 2 || !d->gates[0]) return 0;
  const int nblk = (d->B + RR_ROWS - 1) / RR_ROWS;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || d->layers * nblk > cus) return 0;
  return ((size_t)d->layers * nblk + 16) * sizeof(uint32_t);
}

int rnn_rowres_fwd(const mvae_rnn_fwd_desc* d, hipStream_t st) {
  constexpr int H = 72;
  if (d->cell != MVAE_CELL_LSTM || d->dtype != MVAE_F32 || d->H != H || d->x0 || d->lengths) return MVAE_ERR_UNSUPPORTED;
  // layer 0's pre-activation addend: a gathered sequence add0 [T, B, 4H], or the token table + ids (kept in LDS; add0 must then be absent)
  const bool tbl = d->add_table != nullptr;
  if (tbl ? (d->add0 || !d->add_index || d->add_table_rows < 1 || d->add_table_rows > RR_TBL_ROWS || d->T > RR_TBL_T) : !d->add0) return MVAE_ERR_UNSUPPORTED;
  if (!d->gates[0]) return MVAE_ERR_UNSUPPORTED;      // forward-only calls (no save buffers) take the wavefront schedule
  for (int l = 0; l < d->layers; ++l)
    if (d->h0[l]) return MVAE_ERR_UNSUPPORTED;
  if (d->ldh % 4 || (reinterpret_cast<uintptr_t>(d->hs[0]) & 15)) return MVAE_ERR_UNSUPPORTED;
  const int T = d->T, B = d->B;
  // hoisting the upper layers' input projection pays while its GEMM is small (measured, encoder forward per step: b = 128: 0.478 -> 0.427 ms;
  // B = 1024: 0.575 -> 0.790 ms, the two [T*B, 72] x [72, 288] GEMMs then cost more than the halved contraction saves)
  const char* hv = mvae_knob("MVAE_ROWRES_HOIST");
  const bool hoist = hv ? atoi(hv) != 0 : (long)T * B <= 16384;
  // Layer-concurrent form (one launch, the layers as a pipeline over progress words): when every workgroup of the grid is resident at once
  // (one per CU is what is assumed) and the caller gave scratch for the progress words (persist_ws: layers x ceil(B / 4) + 16 words).  It
  // replaces the hoisted form too: 0.427 ms (hoisted, layer after layer) -> see DESIGN.md at b = 128.
  {
    const int nblk = (B + RR_ROWS - 1) / RR_ROWS, NL = d->layers;
    int dev = 0, cus = 0;
    const char* pk = mvae_knob("MVAE_ROWRES_PIPE");
    bool pipe = (pk ? atoi(pk) != 0 : true) && !d->no_spin && NL >= 2 && d->persist_ws && d->persist_ws_bytes >= ((size_t)NL * nblk + 16) * sizeof(uint32_t);
    if (pipe && (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || NL * nblk > cus)) pipe = false;
    for (int l = 1; l < NL && pipe; ++l) if (reinterpret_cast<uintptr_t>(d->hs[l - 1]) & 15) pipe = false;
    if (pipe) {
      RowResFAll all;
      all.nl = NL;
      for (int l = 0; l < NL; ++l) {
        RowResF& a = all.l[l];
        a.x = l ? reinterpret_cast<const float*>(d->hs[l - 1]) : nullptr; a.ldx = d->ldh;
        a.add = l ? nullptr : d->add0; a.add_ts = d->add0_tstride;
        a.tbl = d->add_table; a.tbl_rows = d->add_table_rows; a.idx = d->add_index; a.idx_ld = d->add_index_ld;
        a.w_ih = l ? reinterpret_cast<const float*>(d->w_ih[l]) : nullptr; a.ldw_ih = d->ldw_ih[l];
        a.w_hh = reinterpret_cast<const float*>(d->w_hh[l]); a.ldw_hh = d->ldw_hh[l];
        a.bias = d->bias[l];
        a.hs = reinterpret_cast<float*>(d->hs[l]); a.ldh = d->ldh;
        a.cs = reinterpret_cast<float*>(d->cs[l]); a.gates = reinterpret_cast<float*>(d->gates[l]);
        a.T = T; a.B = B;
      }
      uint32_t* status = reinterpret_cast<uint32_t*>(d->persist_ws);       // [status 16 words | progress words]: the host reads the first 16 bytes
      uint32_t* flags = status + 16;
      MVAE_CHECK_HIP(hipMemsetAsync(status, 0, ((size_t)NL * nblk + 16) * sizeof(uint32_t), st));
      const char* sp = mvae_knob("MVAE_ROWRES_SPIN");          // tests: 1 = give up at the first hand-off that is not there yet
      const uint32_t spin = sp ? (uint32_t)atoi(sp) : (1u << 17);
      if (tbl) hipLaunchKernelGGL((lstm_rowres_fwd_pipe_kernel<H, true>), dim3(NL * nblk), dim3(320), 0, st, all, flags, status, spin, d->poison);
      else hipLaunchKernelGGL((lstm_rowres_fwd_pipe_kernel<H, false>), dim3(NL * nblk), dim3(320), 0, st, all, flags, status, spin, d->poison);
      mvae_tls_status = status;
      MVAE_CHECK_HIP(hipGetLastError());
      return MVAE_OK;
    }
  }
  for (int l = 0; l < d->layers; ++l) {
    RowResF a;
    a.x = l ? reinterpret_cast<const float*>(d->hs[l - 1]) : nullptr; a.ldx = d->ldh;
    a.add = l ? nullptr : d->add0; a.add_ts = d->add0_tstride;
    a.tbl = d->add_table; a.tbl_rows = d->add_table_rows; a.idx = d->add_index; a.idx_ld = d->add_index_ld;
    a.w_ih = l ? reinterpret_cast<const float*>(d->w_ih[l]) : nullptr; a.ldw_ih = d->ldw_ih[l];
    a.w_hh = reinterpret_cast<const float*>(d->w_hh[l]); a.ldw_hh = d->ldw_hh[l];
    a.bias = d->bias[l];
    a.hs = reinterpret_cast<float*>(d->hs[l]); a.ldh = d->ldh;
    a.cs = reinterpret_cast<float*>(d->cs[l]); a.gates = reinterpret_cast<float*>(d->gates[l]);
    a.T = T; a.B = B;
    if (l && (reinterpret_cast<uintptr_t>(d->hs[l - 1]) & 15)) return MVAE_ERR_UNSUPPORTED;
    dim3 grid((B + RR_ROWS - 1) / RR_ROWS);
    if (l && hoist) {
      // The input part of an upper layer does not depend on its own recurrence: x_t . W_ih^T for ALL t is one exact-f32 GEMM over the
      // layer below's finished outputs, written into this layer's `gates` buffer; the recurrent kernel then takes it as its addend
      // (reads [t+1] one step ahead, overwrites [t] with the saved gates: in place) and contracts over K = H instead of 2H per step.
      const int rc = launch_gemm_nt(MVAE_F32, T * B, 4 * H, H, d->hs[l - 1], d->ldh, d->w_ih[l], d->ldw_ih[l], d->gates[l], 4L * H, MVAE_F32,
                                    nullptr, MVAE_ACT_NONE, 0, nullptr, 0, st);
      if (rc != MVAE_OK) return rc;
      a.x = nullptr; a.w_ih = nullptr;
      a.add = reinterpret_cast<const float*>(d->gates[l]); a.add_ts = (long)B * 4 * H;
      hipLaunchKernelGGL((lstm_rowres_fwd_kernel<H, false, 320>), grid, dim3(320), 0, st, a);
    } else if (l) hipLaunchKernelGGL((lstm_rowres_fwd_kernel<H, true, 320>), grid, dim3(320), 0, st, a);
    else if (tbl) hipLaunchKernelGGL((lstm_rowres_fwd_kernel<H, false, 320, true>), grid, dim3(320), 0, st, a);
    else hipLaunchKernelGGL((lstm_rowres_fwd_kernel<H, false, 320>), grid, dim3(320), 0, st, a);
  }
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}

// [dx ping | dx pong | progress words layers x ceil(B / 4) + status (pipelined form)]
size_t rnn_rowres_bwd_workspace(int layers, int T, int B, int H) {
  return (H == 72 && layers > 1) ? (size_t)2 * T * B * H * sizeof(float) + ((size_t)layers * ((B + RR_ROWS - 1) / RR_ROWS) + 16) * sizeof(uint32_t) : 0;
}

// top layer first; layer l's input gradient (dx, scratch ping-pong) is layer l-1's dy
int rnn_rowres_bwd(const mvae_rnn_bwd_desc* d, hipStream_t st) {
  constexpr int H = 72;
  if (d->cell != MVAE_CELL_LSTM || d->dtype != MVAE_F32 || d->H != H || !d->dy || d->dy_a || d->lengths) return MVAE_ERR_UNSUPPORTED;
  const int NL = d->layers, T = d->T, B = d->B;
  for (int l = 0; l < NL; ++l)
    if (d->dh_last[l] || d->dh0[l] || d->h0[l]) return MVAE_ERR_UNSUPPORTED;
  const size_t need = rnn_rowres_bwd_workspace(NL, T, B, H);
  if (need && (!d->split_ws || d->split_ws_bytes < need)) return MVAE_ERR_UNSUPPORTED;
  float* dxbuf[2] = {reinterpret_cast<float*>(d->split_ws), reinterpret_cast<float*>(d->split_ws) + (size_t)T * B * H};
  const float* dy = d->dy; long dy_ld = d->dy_ld;
  RowResBAll all;
  all.nl = NL;
  for (int l = NL - 1; l >= 0; --l) {
    RowResB& a = all.l[l];
    a.dy = dy; a.dy_ld = dy_ld;
    a.gates = reinterpret_cast<const float*>(d->gates[l]); a.cs = reinterpret_cast<const float*>(d->cs[l]);
    a.w_ihT = l ? reinterpret_cast<const float*>(d->w_ihT[l]) : nullptr; a.ldw_ihT = d->ldw_ihT[l];
    a.w_hhT = reinterpret_cast<const float*>(d->w_hhT[l]); a.ldw_hhT = d->ldw_hhT[l];
    a.dG = reinterpret_cast<float*>(d->dG[l]); a.ldg = d->ldg;
    a.dx = l ? dxbuf[l & 1] : nullptr;
    a.T = T; a.B = B;
    dy = a.dx; dy_ld = H;
  }
  // layer-concurrent form: every workgroup of the grid must be resident at once (one per CU is what is assumed), 2 or 3 layers (the dx
  // ping-pong has two buffers: layer l writes l & 1 while layer l + 2 writes the same one only when there are four)
  const int nblk = (B + RR_ROWS - 1) / RR_ROWS;
  int dev = 0, cus = 0;
  const char* pk = mvae_knob("MVAE_ROWRES_PIPE");
  bool pipe = (pk ? atoi(pk) != 0 : true) && !d->no_spin && NL >= 2 && NL <= 3;
  if (pipe && (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || NL * nblk > cus)) pipe = false;
  uint32_t* flags = need ? reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(d->split_ws) + (size_t)2 * T * B * H * sizeof(float)) : nullptr;
  uint32_t* status = flags ? flags + (size_t)NL * nblk : nullptr;
  if (pipe && flags) {
    MVAE_CHECK_HIP(hipMemsetAsync(flags, 0, ((size_t)NL * nblk + 16) * sizeof(uint32_t), st));     // progress words + status
    const char* sp = mvae_knob("MVAE_ROWRES_SPIN");
    hipLaunchKernelGGL((lstm_rowres_bwd_pipe_kernel<H>), dim3(NL * nblk), dim3(256), 0, st, all, flags, status, sp ? (uint32_t)atoi(sp) : (1u << 17), d->poison);
    mvae_tls_status = status;              // (the layer-by-layer form has no spins: no status to report, no memset)
  } else {
    hipLaunchKernelGGL((lstm_rowres_bwd_all_kernel<H>), dim3(nblk), dim3(256), 0, st, all);
  }
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}

// One-layer bf16 GRU(256) with a token-table addend (the MOSES encoder): row-resident forward, one launch for the whole sequence
int rnn_gru_rowres_fwd(const mvae_rnn_fwd_desc* d, hipStream_t st) {
  constexpr int H = 256;
  if (d->cell != MVAE_CELL_GRU || d->dtype != MVAE_BF16 || d->layers != 1 || d->H != H || d->x0 || d->add0 || !d->add_table || !d->add_index)
    return MVAE_ERR_UNSUPPORTED;
  if (d->drop_p > 0.f || !d->hs[0] || !d->cstate[0] || d->T < 1 || d->B % (4 * GR_RPL)) return MVAE_ERR_UNSUPPORTED;      // whole workgroups of rows (ragged batches: the wavefront schedule)
  if ((long)d->T * d->B * 4 * H >= (1L << 31) || (long)d->T * d->B * d->ldh >= (1L << 31)) return MVAE_ERR_UNSUPPORTED;      // 32-bit element offsets in the kernel
  if (d->ldw_hh[0] % 8 || (reinterpret_cast<uintptr_t>(d->w_hh[0]) & 15)) return MVAE_ERR_UNSUPPORTED;
  GruRowF a;
  { const char* sf = mvae_knob("MVAE_GRU_ROWRES_SAFE"); a.safe = (sf && atoi(sf) != 0) ? 1 : 0; }
  a.w_hh = reinterpret_cast<const bf16_t*>(d->w_hh[0]); a.ldw = d->ldw_hh[0];
  a.bias = d->bias[0];
  a.tbl = d->add_table; a.idx = reinterpret_cast<const long*>(d->add_index); a.idx_ld = d->add_index_ld; a.tbl_rows = d->add_table_rows;
  a.lengths = d->lengths;
  a.h0 = reinterpret_cast<const bf16_t*>(d->h0[0]); a.ldh0 = d->ldh0;
  a.hs = reinterpret_cast<bf16_t*>(d->hs[0]); a.ldh = d->ldh;
  a.gates = reinterpret_cast<bf16_t*>(d->gates[0]);
  a.hstate = d->cstate[0];
  a.T = d->T; a.B = d->B;
  const size_t lds = (size_t)(H / 16) * (H / 32) * 1024 + 2 * GR_ROWS * (H + 8) * sizeof(bf16_t) + 4 * H * sizeof(float) + (size_t)4 * GR_RPL * d->T * sizeof(int);      // W_hn fragments + the double-buffered A rows + bias + token ids
  if (lds > 160 * 1024) return MVAE_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_rowres_fwd_kernel<H, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_rowres_fwd_kernel<H, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  if (a.gates) hipLaunchKernelGGL((gru_rowres_fwd_kernel<H, true>), dim3(d->B / (4 * GR_RPL)), dim3(512), lds, st, a);
  else hipLaunchKernelGGL((gru_rowres_fwd_kernel<H, false>), dim3(d->B / (4 * GR_RPL)), dim3(512), lds, st, a);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}

// backward of the same stack: one launch; the weight / bias / table gradients stay with the caller's GEMMs over dG
int rnn_gru_rowres_bwd(const mvae_rnn_bwd_desc* d, hipStream_t st) {
  constexpr int H = 256;
  if (d->cell != MVAE_CELL_GRU || d->dtype != MVAE_BF16 || d->layers != 1 || d->H != H || d->dy || d->dy_a || d->drop_p > 0.f) return MVAE_ERR_UNSUPPORTED;      // (a gradient through the output sequence: the wavefront schedule)
  if (d->dh0[0] || !d->gates[0] || !d->hs[0] || !d->dG[0] || !d->w_hhT[0] || d->T < 1 || d->B % (4 * GR_RPL)) return MVAE_ERR_UNSUPPORTED;
  if (d->ldw_hhT[0] % 8 || (reinterpret_cast<uintptr_t>(d->w_hhT[0]) & 15) || d->ldg < 4L * H) return MVAE_ERR_UNSUPPORTED;
  GruRowB a;
  { const char* sf = mvae_knob("MVAE_GRU_ROWRES_SAFE"); a.safe = (sf && atoi(sf) != 0) ? 1 : 0; }
  a.w_hhT = reinterpret_cast<const bf16_t*>(d->w_hhT[0]); a.ldwT = d->ldw_hhT[0];
  a.gates = reinterpret_cast<const bf16_t*>(d->gates[0]);
  a.hs = reinterpret_cast<const bf16_t*>(d->hs[0]); a.ldh = d->ldh;
  a.h0 = reinterpret_cast<const bf16_t*>(d->h0[0]); a.ldh0 = d->ldh0;
  a.dh_last = d->dh_last[0];
  a.lengths = d->lengths;
  a.dG = reinterpret_cast<bf16_t*>(d->dG[0]); a.ldg = d->ldg;
  a.T = d->T; a.B = d->B;
  constexpr size_t lds = (size_t)(H / 16) * (H / 32) * 1024 + 2 * (4 * GR_RPL) * GRB_LDG * sizeof(bf16_t);
  static bool attr_set = false;
  if (!attr_set) { MVAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_rowres_bwd_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; }
  hipLaunchKernelGGL((gru_rowres_bwd_kernel<H>), dim3(d->B / (4 * GR_RPL)), dim3(512), lds, st, a);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}
