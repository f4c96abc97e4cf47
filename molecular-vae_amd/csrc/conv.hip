// Conv1d(k) + bias + SELU, valid, stride 1 (models.py:71-77 ConvSELU, :118-120, :129-131) as a SLIDING-WINDOW GEMM.
// Activations are channels-last: x[b][w][c] at x + b * x_bs + w * ldx + c, with the pad channels [Cin, ldx) held at zero.
// Then the im2col row of output position (b, w) is the CONTIGUOUS range x[b, w : w + k, :] of k * ldx floats, i.e. the
// im2col matrix is a view of x with overlapping rows (row stride ldx, row length k * ldx) -- it is never written anywhere.
//   forward : y[(b, w), o]  = selu(sum_kc window(b, w)[kc] * wp[o][kc] + bias[o])        NT GEMM, grouped A rows
//   dX      : dx[(b, w), c] = sum_ko windowz(b, w)[ko] * wq[c][ko]                       the same GEMM on the zero-padded dz
//   dW      : dwp[o][kc]    = sum_(b, w) dz[(b, w), o] * window(b, w)[kc]                f32 TN GEMM, both operands grouped
// with the weights repacked per step (tiny): wp[o][j * ldx + c] = w[o][c][j],  wq[c][j * ldo + o] = w[o][c][k - 1 - j].
#include "common.hpp"
#include "kernels.hpp"

static inline int cgrid(long n, int cap = 4096) {
  long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// wp [Cout][k * ldx], wq [Cin][k * ldo] (either may be null); pad entries are written as zero.
__global__ __launch_bounds__(256) void conv_pack_w_kernel(int Cin, int Cout, int k, const float* w, int ldx, float* wp, int ldo, float* wq) {
  const long np = wp ? (long)Cout * k * ldx : 0, nq = wq ? (long)Cin * k * ldo : 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < np + nq; i += (long)gridDim.x * 256) {
    if (i < np) {
      const int c = (int)(i % ldx), j = (int)((i / ldx) % k), o = (int)(i / ((long)ldx * k));
      wp[i] = (c < Cin) ? w[((long)o * Cin + c) * k + j] : 0.f;
    } else {
      const long q = i - np;
      const int o = (int)(q % ldo), j = (int)((q / ldo) % k), c = (int)(q / ((long)ldo * k));
      wq[q] = (o < Cout) ? w[((long)o * Cin + c) * k + (k - 1 - j)] : 0.f;
    }
  }
}

// dw[o][c][j] = dwp[o][j * ldx + c]
__global__ __launch_bounds__(256) void conv_unpack_dw_kernel(int Cin, int Cout, int k, const float* dwp, int ldx, float* dw) {
  const long n = (long)Cout * Cin * k;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int j = (int)(i % k), c = (int)((i / k) % Cin), o = (int)(i / ((long)Cin * k));
    dw[i] = dwp[((long)o * k + j) * ldx + c];
  }
}

// dzp [B][Wout + 2(k-1)][ldo]: interior rows = dy * act'(y) (columns < Cout; derivative expressed through the OUTPUT y), everything else zero.
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
  return act == MVAE_ACT_SELU ? selu_grad_from_out(y) : act == MVAE_ACT_RELU ? (y > 0.f ? 1.f : 0.f) : 1.f;
}
__global__ __launch_bounds__(256) void conv_dz_pad_kernel(int B, int Wout, int Cout, int ldo, int k, const float* dy, const float* y, float* dzp, int act) {
  const int Wp = Wout + 2 * (k - 1), c4n = ldo / 4;
  const long n = (long)B * Wp * c4n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % c4n) * 4, wp = (int)((i / c4n) % Wp), b = (int)(i / ((long)c4n * Wp));
    const int w = wp - (k - 1);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (w >= 0 && w < Wout) {
      const long src = ((long)b * Wout + w) * ldo + c4;
      const float4 g = *reinterpret_cast<const float4*>(dy + src), o = *reinterpret_cast<const float4*>(y + src);
      if (c4 < Cout) v.x = g.x * act_grad_from_out(o.x, act);
      if (c4 + 1 < Cout) v.y = g.y * act_grad_from_out(o.y, act);
      if (c4 + 2 < Cout) v.z = g.z * act_grad_from_out(o.z, act);
      if (c4 + 3 < Cout) v.w = g.w * act_grad_from_out(o.w, act);
    }
    *reinterpret_cast<float4*>(dzp + i * 4) = v;
  }
}

extern "C" {

int mvae_conv1d_pack_weights(int Cin, int Cout, int k, const float* w, int ldx, float* wp, int ldo, float* wq, void* stream) {
  if (!w || Cin < 1 || Cout < 1 || k < 1 || (wp && ldx < Cin) || (wq && ldo < Cout) || (!wp && !wq)) return MVAE_ERR_INVALID;
  const long n = (wp ? (long)Cout * k * ldx : 0) + (wq ? (long)Cin * k * ldo : 0);
  hipLaunchKernelGGL(conv_pack_w_kernel, dim3(cgrid(n)), dim3(256), 0, (hipStream_t)stream, Cin, Cout, k, w, ldx, wp, ldo, wq);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}

size_t mvae_conv1d_selu_fwd_workspace(int B, int W, int ldx, int Cout, int k) {
  return gemm_nt_workspace_bytes(B * (W - k + 1), Cout, k * ldx, MVAE_F32);
}

int mvae_conv1d_act_fwd(int act, int B, int W, int ldx, int64_t x_bs, int Cout, int k, const float* x, const float* wp, const float* bias,
                        float* y, int64_t ldy, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !wp || !y || B < 1 || Cout < 1 || k < 1 || W < k || ldx < 1 || ldy < Cout || x_bs < (int64_t)W * ldx) return MVAE_ERR_INVALID;
  if (act != MVAE_ACT_NONE && act != MVAE_ACT_SELU && act != MVAE_ACT_RELU) return MVAE_ERR_INVALID;
  const int Wout = W - k + 1;
  return launch_gemm_nt_grouped(MVAE_F32, B * Wout, Cout, k * ldx, x, ldx, Wout, x_bs, (long)B * x_bs, wp, (long)k * ldx, y, ldy, MVAE_F32, bias,
                                act, 0, ws, ws_bytes, (hipStream_t)stream);
}
int mvae_conv1d_selu_fwd(int B, int W, int ldx, int64_t x_bs, int Cout, int k, const float* x, const float* wp, const float* bias,
                         float* y, int64_t ldy, void* ws, size_t ws_bytes, void* stream) {
  return mvae_conv1d_act_fwd(MVAE_ACT_SELU, B, W, ldx, x_bs, Cout, k, x, wp, bias, y, ldy, ws, ws_bytes, stream);
}

static size_t conv_bwd_parts(int B, int W, int Cin, int ldx, int Cout, int ldo, int k, size_t* dwp_bytes, size_t* g_bytes) {
  const int Wout = W - k + 1;
  *dwp_bytes = (size_t)Cout * k * ldx * sizeof(float);
  size_t g = gemm_tn_f32_workspace_bytes(Cout, k * ldx, B * Wout);
  const size_t g2 = gemm_nt_workspace_bytes(B * W, Cin, k * ldo, MVAE_F32);
  const size_t g3 = colsum_workspace_bytes(B * (Wout + 2 * (k - 1)), Cout);
  if (g2 > g) g = g2;
  if (g3 > g) g = g3;
  *g_bytes = g;
  return *dwp_bytes + g + 256;
}
size_t mvae_conv1d_selu_bwd_workspace(int B, int W, int Cin, int ldx, int Cout, int ldo, int k) {
  size_t a, b;
  return conv_bwd_parts(B, W, Cin, ldx, Cout, ldo, k, &a, &b);
}

int mvae_conv1d_selu_bwd(int B, int W, int Cin, int ldx, int64_t x_bs, int Cout, int ldo, int k, const float* dy, const float* y,
                         const float* x, const float* wq, float* dzp, float* dw, float* db, float* dx, int64_t lddx,
                         void* ws, size_t ws_bytes, void* stream) {
  return mvae_conv1d_act_bwd(MVAE_ACT_SELU, B, W, Cin, ldx, x_bs, Cout, ldo, k, dy, y, x, wq, dzp, dw, db, dx, lddx, ws, ws_bytes, stream);
}
int mvae_conv1d_act_bwd(int act, int B, int W, int Cin, int ldx, int64_t x_bs, int Cout, int ldo, int k, const float* dy, const float* y,
                        const float* x, const float* wq, float* dzp, float* dw, float* db, float* dx, int64_t lddx,
                        void* ws, size_t ws_bytes, void* stream) {
  if (!dy || !y || !x || !dzp || (dw && !db) || !ws) return MVAE_ERR_INVALID;      // dw == NULL: the weight gradient is left to the caller (mvae_conv1d_dw_problem)
  const bool x3 = (act & MVAE_CONV_BWD_X3) != 0;
  act &= ~MVAE_CONV_BWD_X3;
  if (act != MVAE_ACT_NONE && act != MVAE_ACT_SELU && act != MVAE_ACT_RELU) return MVAE_ERR_INVALID;
  if (B < 1 || Cin < 1 || Cout < 1 || k < 1 || W < k || ldx < Cin || ldo < Cout || (ldx & 3) || (ldo & 3) || (x_bs & 3) ||
      x_bs < (int64_t)W * ldx) return MVAE_ERR_INVALID;
  if (dx && (!wq || lddx < Cin)) return MVAE_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dzp)) & 15) return MVAE_ERR_INVALID;
  size_t dwp_bytes, g_bytes;
  if (ws_bytes < conv_bwd_parts(B, W, Cin, ldx, Cout, ldo, k, &dwp_bytes, &g_bytes)) return MVAE_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int Wout = W - k + 1, Wp = Wout + 2 * (k - 1);
  char* wsp = reinterpret_cast<char*>(ws);
  float* dwp = reinterpret_cast<float*>(wsp); wsp += (dwp_bytes + 255) & ~(size_t)255;
  void* gws = wsp; const size_t gws_bytes = ws_bytes - (size_t)(wsp - reinterpret_cast<char*>(ws));
  int rc;
  hipLaunchKernelGGL(conv_dz_pad_kernel, dim3(cgrid((long)B * Wp * (ldo / 4))), dim3(256), 0, st, B, Wout, Cout, ldo, k, dy, y, dzp, act);
  MVAE_CHECK_HIP(hipGetLastError());
  // dwp[o][kc] = sum over output positions of dz[(b,w)][o] * window(b,w)[kc]; the bias gradient db[o] = sum over positions of dz[..][o] rides
  // along as the GEMM's virtual ones column (the x3 form: ~16 mantissa bits, like the rest of that GEMM) -- no separate column-sum launches
  if (dw) {
    const float* dz0 = dzp + (long)(k - 1) * ldo;
    if ((rc = launch_gemm_tn_f32(Cout, k * ldx, B * Wout, dz0, ldo, Wout, (long)Wp * ldo, x, ldx, Wout, x_bs, dwp, (long)k * ldx, 0, gws, gws_bytes, st, x3, db, 0))) return rc;
    hipLaunchKernelGGL(conv_unpack_dw_kernel, dim3(cgrid((long)Cout * Cin * k)), dim3(256), 0, st, Cin, Cout, k, dwp, ldx, dw);
    MVAE_CHECK_HIP(hipGetLastError());
  }
  if (dx) {
    // dx[(b,w)][c] = sum_{j,o} dzp[b][w + j][o] * w[o][c][k-1-j]: full correlation = the forward GEMM on the padded dz
    if ((rc = launch_gemm_nt_grouped(x3 ? MVAE_F32X3 : MVAE_F32, B * W, Cin, k * ldo, dzp, ldo, W, (long)Wp * ldo, (long)B * Wp * ldo, wq, (long)k * ldo, dx, lddx,
                                     MVAE_F32, nullptr, MVAE_ACT_NONE, 0, gws, gws_bytes, st))) return rc;
  }
  return MVAE_OK;
}

// The weight / bias gradient of the same layer as a PROBLEM of mvae_gemm_tn_f32_multi: dwp [Cout][k * ldx] (packed like wp) = dz^T . windows(x),
// db = column sums of dz, from the dzp buffer a previous mvae_conv1d_act_bwd(..., dw = NULL, ...) call filled.  mvae_conv1d_unpack_dw turns dwp
// into the parameter layout dw [Cout][Cin][k] afterwards.
int mvae_conv1d_dw_problem(int B, int W, int Cin, int ldx, int64_t x_bs, int Cout, int ldo, int k, const float* dzp, const float* x, float* dwp,
                           float* db, int x3, mvae_gemm_tn_f32_problem* out) {
  if (!dzp || !x || !dwp || !db || !out || B < 1 || Cin < 1 || Cout < 1 || k < 1 || W < k || ldx < Cin || ldo < Cout || (ldx & 3) || (ldo & 3) ||
      (x_bs & 3) || x_bs < (int64_t)W * ldx) return MVAE_ERR_INVALID;
  const int Wout = W - k + 1, Wp = Wout + 2 * (k - 1);
  out->M = Cout; out->N = k * ldx; out->K = (int64_t)B * Wout;
  out->A = dzp + (long)(k - 1) * ldo; out->lda = ldo; out->a_group = Wout; out->a_gstride = (int64_t)Wp * ldo;
  out->B = x; out->ldb = ldx; out->b_group = Wout; out->b_gstride = x_bs;
  out->C = dwp; out->ldc = (int64_t)k * ldx; out->accumulate = 0;
  out->colsum_out = db; out->colsum_accumulate = 0; out->x3 = x3 ? 1 : 0;
  return MVAE_OK;
}
int mvae_conv1d_unpack_dw(int Cin, int Cout, int k, const float* dwp, int ldx, float* dw, void* stream) {
  if (!dwp || !dw || Cin < 1 || Cout < 1 || k < 1 || ldx < Cin) return MVAE_ERR_INVALID;
  hipLaunchKernelGGL(conv_unpack_dw_kernel, dim3(cgrid((long)Cout * Cin * k)), dim3(256), 0, (hipStream_t)stream, Cin, Cout, k, dwp, ldx, dw);
  MVAE_CHECK_HIP(hipGetLastError());
  return MVAE_OK;
}

}  // extern "C"
