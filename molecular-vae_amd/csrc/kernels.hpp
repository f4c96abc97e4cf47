// Internal launcher declarations (host side).  The extern "C" surface is include/mvae.h.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "../../include/mvae.h"

size_t gemm_nt_workspace_bytes(int M, int N, int K, int dtype);
int launch_gemm_nt(int dtype, int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                   int c_dtype, const float* bias, int act, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
int launch_gemm_nt_grouped(int dtype, int M, int N, int K, const void* A, long lda, int a_group, long a_gstride, long a_total,
                           const void* B, long ldb, void* C, long ldc, int c_dtype, const float* bias, int act, int accumulate,
                           void* ws, size_t ws_bytes, hipStream_t st);
size_t gemm_tn_f32_workspace_bytes(int M, int N, int R);
int launch_gemm_tn_f32(int M, int N, int R, const float* A, long lda, int a_group, long a_gstride, const float* B, long ldb, int b_group,
                       long b_gstride, float* C, long ldc, int accumulate, void* ws, size_t ws_bytes, hipStream_t st, bool x3 = false,
                       float* colsum_out = nullptr, int colsum_acc = 0);
size_t gemm_tn_workspace_bytes(int M, int N, int K);
int launch_gemm_tn_bf16(int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc, int c_dtype,
                        const float* bias, int act, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
bool gemm_tn_colsum_supported(int M, int N, int K);
size_t gemm_tn_colsum_workspace_bytes(int M, int N, int K);
int launch_gemm_tn_bf16_colsum(int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc, int c_dtype,
                               const float* bias, int act, int accumulate, float* colsum_out, int colsum_acc, void* ws, size_t ws_bytes,
                               hipStream_t st);
int launch_cast_transpose(int dsrc, int ddst, int R, int C, const void* src, long lds_, void* dst, long ldd, void* dstT,
                          long ldt, hipStream_t st);
size_t colsum_workspace_bytes(int M, int N);
int launch_colsum(int M, int N, const float* X, long ldx, float* out, void* ws, size_t ws_bytes, hipStream_t st);
int launch_selu_bwd(long n, float* dy, const float* y, hipStream_t st);

// Where the launches of the current mvae_rnn_fwd / mvae_rnn_bwd call report a bounded spin that ran out (capi.hip clears it before dispatching
// and hands it to the caller as *status_out): set by every launcher that enqueues a kernel with bounded spins, left alone by the others.
extern thread_local const void* mvae_tls_status;
constexpr int MVAE_MAX_DEVICES = 64;

// rnn_persist.hip: weights-resident dataflow LSTM forward (b = 128, 4 x 1024, bf16)
bool rnn_persist_fwd_supported(const mvae_rnn_fwd_desc* d);
size_t rnn_persist_fwd_workspace_bytes(int T);
int rnn_persist_fwd(const mvae_rnn_fwd_desc* d, void* ws, size_t ws_bytes, hipStream_t st);
// rnn_persist_bwd.hip: weights-resident dataflow LSTM backward (same shape)
bool rnn_persist_bwd_supported(const mvae_rnn_bwd_desc* d);
size_t rnn_persist_bwd_workspace_bytes(int T);
int rnn_persist_bwd(const mvae_rnn_bwd_desc* d, void* ws, size_t ws_bytes, hipStream_t st);
size_t rnn_rowres_fwd_pipe_workspace(const mvae_rnn_fwd_desc* d);     // rnn_rowres.hip: layer-concurrent row-resident LSTM forward
