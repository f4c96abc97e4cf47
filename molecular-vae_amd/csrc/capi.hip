// extern "C" entry points that are thin wrappers over the launchers in gemm.hip / rnn.hip.
#include "common.hpp"
#include "kernels.hpp"

int rnn_fwd_impl(const mvae_rnn_fwd_desc* d, hipStream_t st);
int rnn_bwd_impl(const mvae_rnn_bwd_desc* d, hipStream_t st);
size_t rnn_bwd_workspace_bytes(const mvae_rnn_bwd_desc* d);

thread_local const void* mvae_tls_status = nullptr;

extern "C" {

int mvae_abi_version(void) { return MVAE_ABI_VERSION; }
size_t mvae_struct_size(int which) {
  switch (which) {
    case 0: return sizeof(mvae_rnn_fwd_desc);
    case 1: return sizeof(mvae_rnn_bwd_desc);
    case 2: return sizeof(mvae_gemm_tn_problem);
    case 3: return sizeof(mvae_pack_job);
    case 4: return sizeof(mvae_gemm_tn_f32_problem);
    default: return 0;
  }
}

int mvae_knob_int(const char* name, int dflt) {
  const char* v = name ? mvae_knob(name) : nullptr;
  return v ? atoi(v) : dflt;
}

const char* mvae_status_string(int status) {
  switch (status) {
    case MVAE_OK: return "ok";
    case MVAE_ERR_INVALID: return "invalid argument";
    case MVAE_ERR_WORKSPACE: return "workspace too small";
    case MVAE_ERR_UNSUPPORTED: return "unsupported combination";
    default: return status > 0 ? hipGetErrorString((hipError_t)status) : "unknown status";
  }
}

size_t mvae_gemm_nt_workspace(int M, int N, int K, int dtype_ab) { return gemm_nt_workspace_bytes(M, N, K, dtype_ab); }

int mvae_gemm_nt(int dtype_ab, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                 int dtype_c, const float* bias, int act, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  return launch_gemm_nt(dtype_ab, M, N, K, A, lda, B, ldb, C, ldc, dtype_c, bias, act, accumulate, ws, ws_bytes, (hipStream_t)stream);
}

size_t mvae_gemm_tn_workspace(int M, int N, int K, int dtype_ab) {
  return dtype_ab == MVAE_BF16 ? gemm_tn_workspace_bytes(M, N, K) : dtype_ab == MVAE_F32 ? gemm_tn_f32_workspace_bytes(M, N, K) : 0;
}

int mvae_gemm_tn(int dtype_ab, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                 int dtype_c, const float* bias, int act, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (dtype_ab == MVAE_F32) {                                // exact-f32 kernel: plain fp32 result, no fused bias / activation
    if (dtype_c != MVAE_F32 || bias || act != MVAE_ACT_NONE) return MVAE_ERR_UNSUPPORTED;
    return launch_gemm_tn_f32(M, N, K, (const float*)A, lda, 0, 0, (const float*)B, ldb, 0, 0, (float*)C, ldc, accumulate, ws, ws_bytes, (hipStream_t)stream);
  }
  if (dtype_ab != MVAE_BF16) return MVAE_ERR_UNSUPPORTED;
  return launch_gemm_tn_bf16(M, N, K, A, lda, B, ldb, C, ldc, dtype_c, bias, act, accumulate, ws, ws_bytes, (hipStream_t)stream);
}

int mvae_gemm_tn_f32_colsum(int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int accumulate,
                            float* colsum_out, int colsum_accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (!colsum_out) return MVAE_ERR_INVALID;
  return launch_gemm_tn_f32(M, N, K, A, lda, 0, 0, B, ldb, 0, 0, C, ldc, accumulate, ws, ws_bytes, (hipStream_t)stream, false, colsum_out, colsum_accumulate);
}

size_t mvae_gemm_tn_colsum_workspace(int M, int N, int K) { return gemm_tn_colsum_supported(M, N, K) ? gemm_tn_colsum_workspace_bytes(M, N, K) : 0; }
int mvae_gemm_tn_colsum_supported(int M, int N, int K) { return gemm_tn_colsum_supported(M, N, K) ? 1 : 0; }
int mvae_gemm_tn_colsum(int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int accumulate,
                        float* colsum_out, int colsum_accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (!colsum_out) return MVAE_ERR_INVALID;
  return launch_gemm_tn_bf16_colsum(M, N, K, A, lda, B, ldb, C, ldc, MVAE_F32, nullptr, MVAE_ACT_NONE, accumulate, colsum_out, colsum_accumulate,
                                    ws, ws_bytes, (hipStream_t)stream);
}

int mvae_dropout_keep(uint32_t seed, uint32_t idx, float p) {
  uint32_t h = idx * 0x9E3779B1u ^ seed;
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h >= (uint32_t)((double)p * 4294967296.0) ? 1 : 0;
}

int mvae_rnn_fwd(const mvae_rnn_fwd_desc* d, void* stream, const void** status_out) {
  mvae_tls_status = nullptr;
  const int rc = (d && d->persist_ws && !d->no_spin && rnn_persist_fwd_supported(d))
                     ? rnn_persist_fwd(d, d->persist_ws, d->persist_ws_bytes, (hipStream_t)stream)
                     : rnn_fwd_impl(d, (hipStream_t)stream);      // (the row-resident f32 stacks use persist_ws for their layer-concurrent form)
  if (status_out) *status_out = rc == MVAE_OK ? mvae_tls_status : nullptr;
  return rc;
}
size_t mvae_rnn_fwd_persist_workspace(const mvae_rnn_fwd_desc* d) {
  if (!d) return 0;
  if (rnn_persist_fwd_supported(d)) return rnn_persist_fwd_workspace_bytes(d->T);
  return rnn_rowres_fwd_pipe_workspace(d);
}
int mvae_rnn_bwd(const mvae_rnn_bwd_desc* d, void* stream, const void** status_out) {
  mvae_tls_status = nullptr;
  const int rc = (d && d->persist_ws && !d->no_spin && rnn_persist_bwd_supported(d))
                     ? rnn_persist_bwd(d, d->persist_ws, d->persist_ws_bytes, (hipStream_t)stream)
                     : rnn_bwd_impl(d, (hipStream_t)stream);
  if (status_out) *status_out = rc == MVAE_OK ? mvae_tls_status : nullptr;
  return rc;
}
size_t mvae_rnn_bwd_persist_workspace(const mvae_rnn_bwd_desc* d) { return (d && rnn_persist_bwd_supported(d)) ? rnn_persist_bwd_workspace_bytes(d->T) : 0; }
size_t mvae_rnn_bwd_workspace(const mvae_rnn_bwd_desc* d) { return rnn_bwd_workspace_bytes(d); }

}  // extern "C"
