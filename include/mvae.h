/* mvae.h -- C ABI of libmvae_hip.so: the MI355X (gfx950) kernels behind the SMILES-VAE training hot path.
 *
 * The reference (aclyde11/molecular-VAE) has no FFI or plugin layer: its hot path is the Python nn.Module
 * surface of models.py / mosesvae.py and the loop body of train.py:94-104, and all arithmetic is delegated
 * to torch.nn primitives.  This header is therefore the boundary the build defines *beneath* that surface
 * (SURVEY.md section 8b): one entry point per fused op x {fwd,bwd}; each comment names the reference call
 * site (file:line under /root/reference) whose torch primitive it replaces.
 *
 * Conventions
 *  - raw DEVICE pointers, caller-owned (the library never allocates or frees caller memory);
 *  - an explicit stream (hipStream_t passed as void*); every call is asynchronous on it and graph-capturable
 *    (no allocation, no synchronisation inside);
 *  - returns 0 on success, a negative MVAE_ERR_* for bad arguments, a positive hipError_t otherwise;
 *    never throws, never aborts;
 *  - no hidden RNG: noise (eps) is an input pointer;
 *  - matrices are row-major with explicit leading dimensions (in elements).  GEMM operands are read in
 *    16-byte chunks along K: K-extents that are not a multiple of the chunk (4 f32 / 8 bf16) require the
 *    rows to be zero-padded up to the next chunk boundary (the library's own producers do this);
 *  - dtype codes select the storage type of activations/weights inside the recurrent and GEMM kernels;
 *    accumulation, cell state, reductions, loss and optimiser state are always fp32.
 */
#ifndef MVAE_H_
#define MVAE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVAE_ABI_VERSION 10

#define MVAE_OK 0
#define MVAE_ERR_INVALID (-1)     /* bad argument (null pointer, bad size, misaligned leading dimension) */
#define MVAE_ERR_WORKSPACE (-2)   /* workspace too small */
#define MVAE_ERR_UNSUPPORTED (-3) /* combination not implemented */

#define MVAE_F32 0
#define MVAE_BF16 1
#define MVAE_F32X3 2   /* mvae_gemm_nt only: fp32 operands in memory, products on the bf16 MFMA as hi.hi + hi.lo + lo.hi (x = hi + lo in bf16):
                        ~16 mantissa bits per product, fp32 accumulation; shapes the LDS-direct loop does not serve run in exact fp32 */

#define MVAE_ACT_NONE 0
#define MVAE_ACT_SELU 1           /* models.py:58-68 */
#define MVAE_ACT_RELU 2           /* mosesvae.py:66-67 */

#define MVAE_CELL_LSTM 0          /* torch.nn.LSTM, gate rows [i;f;g;o]  (models.py:117,156) */
#define MVAE_CELL_GRU 1           /* torch.nn.GRU,  gate rows [r;z;n]    (mosesvae.py:54-61,73-79) */

#define MVAE_MAX_LAYERS 8

int mvae_abi_version(void);
/* sizeof of the descriptor structs as THIS library was compiled (0: mvae_rnn_fwd_desc, 1: mvae_rnn_bwd_desc, 2: mvae_gemm_tn_problem, 3: mvae_pack_job, 4: mvae_gemm_tn_f32_problem;
 * anything else: 0) -- a binding checks its own mirror of the structs against it at load time. */
size_t mvae_struct_size(int which);
const char* mvae_status_string(int status);
/* Schedule knobs (MVAE_BM, MVAE_BWD_SPLIT, ... : tile / split choices, every setting computes the same results) are read from the environment
 * ONLY while MVAE_TUNING=1 is set too; otherwise stray MVAE_* variables are ignored.  Returns the integer value of knob `name` as the library
 * would read it now, or `dflt`. */
int mvae_knob_int(const char* name, int dflt);

/* ---------------------------------------------------------------------------------------------------------
 * Dense contraction  C[M,N] = act( A[M,K] . B[N,K]^T + bias[N] )          (MFMA, fp32 accumulate)
 * Replaces nn.Linear at models.py:122,87-88,153,157 and mosesvae.py:66-67,81-82, and is the GEMM under the
 * conv / weight-gradient ops below.  dtype_ab: storage of A and B; dtype_c: storage of C.
 * accumulate != 0: C += result (fp32 C only).  ws: scratch for split-K partials (may be NULL when
 * mvae_gemm_nt_workspace() returns 0).
 */
size_t mvae_gemm_nt_workspace(int M, int N, int K, int dtype_ab);
int mvae_gemm_nt(int dtype_ab, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb,
                 void* C, int64_t ldc, int dtype_c, const float* bias, int act, int accumulate,
                 void* ws, size_t ws_bytes, void* stream);

/* C[M,N] = act( A^T . B + bias[N] ) with A [K, lda] and B [K, ldb] both K-major (row = k): the weight-gradient contraction
 * dW = dG^T . X reading dG [T*B, 4H] and X [T*B, H] as the recurrent kernels wrote them (hardware-transposed LDS reads, no
 * transposed copies).  bf16 operands: lda, ldb multiples of 8.  f32 operands (exact v_mfma_f32_16x16x4_f32 kernel): lda, ldb
 * multiples of 4, 16-byte aligned bases, fp32 C, no bias / activation.  Rows k >= K are never read.  Replaces autograd's weight-gradient GEMMs of models.py:128,164,157. */
size_t mvae_gemm_tn_workspace(int M, int N, int K, int dtype_ab);
int mvae_gemm_tn(int dtype_ab, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb,
                 void* C, int64_t ldc, int dtype_c, const float* bias, int act, int accumulate,
                 void* ws, size_t ws_bytes, void* stream);

/* fp32 (exact) form with the bias gradient:  C[M,N] (+)= A^T . B  and  colsum_out[m] (+)= sum_k A[k][m]  from the same MFMAs -- B gets a virtual
 * column of ones at index N, in the slack of the last 64-wide tile (a whole extra tile column only when N % 64 == 0).  Replaces the
 * mvae_gemm_tn + mvae_colsum pair behind every nn.Linear / nn.LSTM bias of the fp32 encoder.  ws >= mvae_gemm_tn_workspace(MVAE_F32). */
int mvae_gemm_tn_f32_colsum(int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int accumulate,
                            float* colsum_out, int colsum_accumulate, void* ws, size_t ws_bytes, void* stream);

/* bf16 weight gradient AND bias gradient from one pass over dG:  C[M,N] (+)= A^T . B  and  colsum_out[m] (+)= sum_k A[k][m].
 * The column sums come from ONE extra MFMA per wave and K-step (an A fragment against a fragment of ones) inside the 256 x 256-tile
 * kernel, shared out over the 16 waves that hold the same A rows, so they cost no extra HBM traffic (a separate mvae_colsum_t re-reads
 * all of dG).  Only for shapes that kernel serves with 4 tile columns and split-K (mvae_gemm_tn_colsum_supported: N == 1024, M >= 2048,
 * K >= 4096); otherwise MVAE_ERR_UNSUPPORTED.  fp32 C, no bias / activation. */
int mvae_gemm_tn_colsum_supported(int M, int N, int K);
/* Grouped form: up to 2 * MVAE_MAX_LAYERS contractions of that kind in ONE launch, each 256 x 256 output tile accumulated in registers over
 * its problem's full K -- no split-K slabs, no reduction launch (the decoder's 7 dW_ih / dW_hh GEMMs are 7 x 64 tiles >= 256 CUs).
 * colsum_out (optional, needs N == 1024) as above; ws >= mvae_gemm_tn_grouped_workspace bytes (two K-half column-sum partials per
 * problem that asks for them, summed in a fixed order by a tiny second launch).  Problems must pass mvae_gemm_tn_grouped_supported. */
typedef struct {
  int M, N; int64_t K;
  const void* A; int64_t lda;          /* [K, lda] bf16, K-major */
  const void* B; int64_t ldb;          /* [K, ldb] bf16 */
  float* C; int64_t ldc; int accumulate;
  float* colsum_out; int colsum_accumulate;
} mvae_gemm_tn_problem;
int mvae_gemm_tn_grouped_supported(int M, int N, int64_t K, int64_t lda, int64_t ldb);
size_t mvae_gemm_tn_grouped_workspace(int n, const mvae_gemm_tn_problem* problems);
int mvae_gemm_tn_grouped(int n, const mvae_gemm_tn_problem* problems, void* ws, size_t ws_bytes, void* stream);
/* The same launch with at most `max_workgroups` workgroups (0: one per tile), each looping over tiles: a throughput-bound group of GEMMs on a
 * side stream that must leave compute units to the latency-bound launches of the main stream (the per-rank batch of a data-parallel job:
 * K = T * b is short, the encoder's backward runs beside it). */
int mvae_gemm_tn_grouped_capped(int n, const mvae_gemm_tn_problem* problems, int max_workgroups, void* ws, size_t ws_bytes, void* stream);
/* Exact-f32 TN contractions of DIFFERENT shapes in ONE launch (+ one reduction launch for all of their split-K slabs): the parameter-gradient
 * GEMMs of the encoder's backward pass -- Lambda heads, dense_1, the three Conv1d weight gradients, the LSTM(72) stack's dW_ih / dW_hh
 * (models.py:87-88,122,118-120,117) -- are a dozen small dependent-free products whose only consumer is the optimiser; launched one by one
 * each leaves most of the chip idle.  Problem i: C[M,N] (+)= A^T . B over K rows, A / B row-major fp32 with optional row groups (row r starts
 * at (r / group) * gstride + (r % group) * ld when group > 0: the overlapping windows of a channels-last Conv1d), colsum_out (optional):
 * column sums of A through a virtual ones column (as mvae_gemm_tn_f32_colsum), x3 != 0: products as 3 x bf16 (MVAE_F32X3).  lda, ldb and the
 * group strides multiples of 4, bases 16-byte aligned.  At most MVAE_TN_F32_MULTI_MAX problems; outputs must not overlap. */
#define MVAE_TN_F32_MULTI_MAX 16
typedef struct {
  int M, N; int64_t K;
  const float* A; int64_t lda; int a_group; int64_t a_gstride;
  const float* B; int64_t ldb; int b_group; int64_t b_gstride;
  float* C; int64_t ldc; int accumulate;
  float* colsum_out; int colsum_accumulate;
  int x3;
} mvae_gemm_tn_f32_problem;
size_t mvae_gemm_tn_f32_multi_workspace(int n, const mvae_gemm_tn_f32_problem* problems);
int mvae_gemm_tn_f32_multi(int n, const mvae_gemm_tn_f32_problem* problems, void* ws, size_t ws_bytes, void* stream);
size_t mvae_gemm_tn_colsum_workspace(int M, int N, int K);
int mvae_gemm_tn_colsum(int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int accumulate,
                        float* colsum_out, int colsum_accumulate, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Data movement helpers.
 */
/* dst[r, c] = (Td) src[r, c] for r<R, c<C; columns C..ldd-1 of dst are zeroed.  When dstT != NULL also
 * dstT[c, r] = (Td) src[r, c] with columns R..ldt-1 zeroed.  Weight packing (fp32 master -> bf16 shadow and
 * its transpose) and activation transposes. */
int mvae_cast_transpose(int dtype_src, int dtype_dst, int R, int C, const void* src, int64_t lds_,
                        void* dst, int64_t ldd, void* dstT, int64_t ldt, void* stream);

/* Multi-tensor pack: a LIST of small packing jobs in one launch (a model refreshes dozens of weight shadows after every optimiser step:
 * fp32 master -> bf16 / zero-padded / transposed copies, bias sums, block copies -- each a few-microsecond launch of its own otherwise).
 *   kind 0: dst[r, c] and / or dstT[c, r] = (dst_dtype) src[r, c] for r < R, c < C only (no zero fill of padding: allocate destinations zeroed);
 *   kind 1: dst[i] = src[i] + src2[i], fp32, R * C elements (b_ih + b_hh);   kind 2: dst[r, c] = src[r, c], fp32, leading dimensions lds / ldd.
 * `jobs_device` is an array in DEVICE memory (build it once: the pointers do not change from step to step); block0 = index of the job's
 * first block = sum of mvae_pack_job_blocks() of the jobs before it; total_blocks = the sum over all jobs.  Jobs must not overlap. */
typedef struct {
  int kind, src_dtype, dst_dtype, R, C, block0;
  const void* src; int64_t lds;
  void* dst; int64_t ldd;
  void* dstT; int64_t ldt;
  const void* src2;
} mvae_pack_job;
int mvae_pack_job_blocks(const mvae_pack_job* job);
int mvae_pack_multi(int njobs, const mvae_pack_job* jobs_device, int total_blocks, void* stream);

/* out[n, b, a] = in[n, a, b] (fp32): the channel-major Flatten of models.py:6-10 <-> the conv GEMM's row order. */
int mvae_permute021(int N, int A, int Bd, const float* in, float* out, void* stream);

/* out[(t*B + b), :] = table[idx[b*L + t], :]   (fp32, W columns).  models.py:127 nn.Embedding, fused with the
 * layer-0 input projection: table = E . W_ih0^T + b. */
int mvae_gather_rows_tb(const int64_t* idx, int B, int L, int nrows, const float* table, int W, const float* base /* optional [B, W] per-sequence addend */,
                        float* out, void* stream);
/* dtable[c, :] = sum over (t,b) with idx[b*L+t]==c of d[(t*B+b), :]   (deterministic).  d has dtype `dtype`. */
int mvae_scatter_rows_tb(int dtype, const int64_t* idx, int B, int L, int nrows, const void* d, int64_t ldd, int W,
                         float* dtable, void* ws, size_t ws_bytes, void* stream);
size_t mvae_scatter_rows_tb_workspace(int B, int L, int nrows, int W);
/* out[(t*B + b), c] = (idx[b*L + t] == c) ? 1 : 0  for c < ld (bf16 [L*B, ld], ld >= nrows, ld % 8 == 0; ids clamped as in the gather).
 * With it the scatter above is the TN contraction  dtable = out^T . d  (mvae_gemm_tn, M = nrows): the bf16 gradient sequence is read
 * once at GEMM streaming rate and summed in fp32 by the MFMAs. */
int mvae_onehot_tb(const int64_t* idx, int B, int L, int nrows, void* out, int64_t ld, void* stream);
/* fp32 form, rows in the order of idx itself: out[r, c] = (idx[r] == c) for r < n, c < ld (ld >= nrows, ld % 4 == 0).  The exact-f32 table gradient
 * of the encoder (models.py:116 Embedding folded into the LSTM's layer-0 projection) is then one problem of mvae_gemm_tn_f32_multi: A = out,
 * B = the layer-0 pre-activation gradients addressed through row groups (row b * L + t of B = dG[t][b]). */
int mvae_onehot_f32(const int64_t* idx, int64_t n, int nrows, float* out, int64_t ld, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Recurrent stack (K2, K7, K10, K12 of SURVEY.md): torch.nn.LSTM / nn.GRU, batch-major module semantics,
 * zero (LSTM) or given (GRU) initial state, time-major storage here.  Replaces models.py:128, models.py:164,
 * mosesvae.py:153, mosesvae.py:188.
 *
 * Schedule: layer-wavefront.  Launch d updates every cell (layer l, time t = d - l); one launch computes, for
 * up to `layers` cells,   pre = x_t . W_ih^T + h_{t-1} . W_hh^T + bias   (two K-segments of one MFMA tile
 * loop) and applies the gate non-linearities and the state update in the epilogue.
 *
 * Layer 0 input is either a real sequence x0 [T*B, in0] (dtype) or a precomputed fp32 pre-activation
 * addend add0 (row (t*B+b) at add0 + t*add0_tstride + b*G*H; tstride 0 = time-invariant input, models.py:163
 * Repeat), or both.  A token input that was folded into W_ih (embedding . W_ih^T = a [rows, G*H] fp32 table, mosesvae.py:150-153,
 * 176-188) is given as add_table + add_index: cell (0, t) adds table row add_index[b * add_index_ld + t] to row b in its epilogue,
 * on top of add0 -- the gathered [T, B, G*H] sequence is never written.  bias[l] (fp32 [G*H]) is added for layers whose input is a
 * real sequence.
 *
 * Saved for backward (all caller-allocated):
 *   hs[l]     [T][B][ldh]   dtype   layer outputs h_t          (hs[layers-1] is the stack output)
 *   cs[l]     [T][B][H]     dtype   cell states as the backward pass reads them (LSTM only)
 *   gates[l]  [T][B][G*H]   dtype   post-activation gates i,f,g,o (LSTM) / r,z,n,(W_hn h + b_hn) (GRU)
 *   A forward-only call (evaluation under no_grad, train.py:120-153) passes gates[l] == cs[l] == NULL for EVERY layer: the
 *   kernels then skip the saved-state stores (12 of the 16 bytes written per (row, unit, step)); hs / cstate are still written.
 *   cstate[l] [2][B][H]     fp32    scratch: the recurrent cell state itself stays fp32 (ping-pong over t)
 * lengths (GRU path): per-sequence valid length, sorted descending (pack_sequence semantics: a finished
 * sequence keeps its last state and emits zeros); NULL = all T.
 */
typedef struct {
  int cell, dtype, layers, T, B, H, in0;
  const void* x0; int64_t x0_ld;
  const float* add0; int64_t add0_tstride;
  const float* add_table; const int64_t* add_index; int64_t add_index_ld; int add_table_rows;   /* optional (NULL): see above; ids are clamped to the table */
  const void* w_ih[MVAE_MAX_LAYERS]; int64_t ldw_ih[MVAE_MAX_LAYERS];   /* [G*H, in] (w_ih[0] unused when x0 == NULL) */
  const void* w_hh[MVAE_MAX_LAYERS]; int64_t ldw_hh[MVAE_MAX_LAYERS];   /* [G*H, H] */
  const float* bias[MVAE_MAX_LAYERS];                                    /* LSTM: b_ih+b_hh [4H]; GRU: [b_ir+b_hr; b_iz+b_hz; b_in; b_hn] [4H] */
  const void* h0[MVAE_MAX_LAYERS]; int64_t ldh0;                         /* optional initial hidden state [B, ldh0] dtype (NULL = zeros) */
  const int32_t* lengths;
  void* hs[MVAE_MAX_LAYERS]; int64_t ldh;
  void* cs[MVAE_MAX_LAYERS];
  void* gates[MVAE_MAX_LAYERS];
  float* cstate[MVAE_MAX_LAYERS];
  int zero_padded_k;   /* != 0: rows of hs, h0, w_ih (l>0) and w_hh are allocated AND zero up to the next multiple of 128 bytes
                          past H, so the contraction may run over whole K-steps (enables the LDS-direct main loop for e.g. H = 72) */
  /* Inter-layer dropout (torch.nn.GRU(dropout=p), train mode; mosesvae.py:73-79): layer l+1 reads  hdrop[l] = hs[l] * keep / (1 - p)
   * instead of hs[l] (the recurrence of layer l itself still reads hs[l]).  hdrop[l] [T][B][ldh] dtype for l < layers-1, all NULL = no
   * dropout.  keep(l,t,b,j) comes from drop_mask[l] ([T][B][H] bytes, 1 = keep) when given (parity tests inject the reference's draw),
   * else from the counter-based hash  mvae_dropout_keep(drop_seed, ((l*T + t)*B + b)*H + j, drop_p)  -- no hidden RNG state. */
  void* hdrop[MVAE_MAX_LAYERS];
  const uint8_t* drop_mask[MVAE_MAX_LAYERS];
  float drop_p; uint32_t drop_seed;
  /* Optional second schedule, the WEIGHTS-RESIDENT DATAFLOW pass (rnn_persist.hip): ONE persistent launch of 256 workgroups in which every
   * workgroup keeps its 256 KB slice of [W_ih | W_hh] in registers for all T steps and the h_t tiles travel between workgroups through
   * write-through stores + flag words (no launch per diagonal).  Serves the per-rank shape of the 8-GPU configuration only (LSTM, bf16,
   * 4 layers, H = 1024, B = 128, zero initial state, time-invariant layer-0 input, ldh = H + 64, a 256-CU device with nothing else running
   * beside it): mvae_rnn_fwd_persist_workspace() returns 0 for everything else.  persist_ws != NULL (that many bytes of device memory,
   * 16-byte aligned) selects it; the first 16 bytes are a status record the launch leaves behind -- word 0 != 0: a bounded spin ran out
   * (a workgroup was not resident, or a producer died), the outputs are invalid; the launch itself always ends.
   * The narrow exact-f32 stacks (LSTM, H = 72: the encoder, models.py:117,128) use the same scratch for their LAYER-CONCURRENT row-resident form
   * (rnn_rowres.hip: all layers in one launch as a pipeline over per-workgroup progress words, when layers x ceil(B / 4) workgroups are resident
   * at once); same status convention.
   * poison (optional, device fp32 word): a launch of this call that gives up ALSO stores a quiet NaN there -- the caller points it at the
   * spare slot behind its flat gradient buffer, so that mvae_sumsq / mvae_clip_adam skip the whole update of a step whose forward or
   * backward pass produced garbage (and, in data parallel, every rank skips it: the slot travels with the gradient all-reduce).  Nothing
   * is ever written there by a launch that succeeds. */
  void* persist_ws; size_t persist_ws_bytes;
  float* poison;
  int no_spin;         /* != 0: never take a schedule that contains bounded spins (the caller's fallback after a launch that gave up) */
} mvae_rnn_fwd_desc;

/* keep decision of the device-generated dropout mask (host-callable restatement: the oracle and the tests use the same hash):
 *   h = idx * 0x9E3779B1 ^ seed; h ^= h >> 16; h *= 0x85EBCA6B; h ^= h >> 13; h *= 0xC2B2AE35; h ^= h >> 16;  keep = h >= (uint32)(p * 2^32) */
int mvae_dropout_keep(uint32_t seed, uint32_t idx, float p);

/* status_out (optional): receives the device address of the 16-byte status record the launches THIS call enqueued will leave behind (word 0
 * != 0: a bounded spin ran out, the outputs are invalid; words 1-2: who), or NULL when the schedule it took contains no bounded spin (the
 * wavefront and layer-by-layer schedules: nothing to check).  The library, not the caller, knows which schedule ran. */
int mvae_rnn_fwd(const mvae_rnn_fwd_desc* d, void* stream, const void** status_out);
size_t mvae_rnn_fwd_persist_workspace(const mvae_rnn_fwd_desc* d);   /* 0: this shape / device is not served by the persistent schedule */

/* Backward through time of the same stack (reverse wavefront).  One launch computes, per cell,
 *   dh_t = dG^{l}_{t+1} . W_hh + dG^{l+1}_t . W_ih^{l+1} (+ dy_t for the top layer)
 * and, in the epilogue, the gate derivative dG^{l}_t (pre-activation gradient) and dc_{t-1}.
 *   w_hhT[l] [H, G*H], w_ihT[l] [in, G*H]: TRANSPOSED weights (K-contiguous for this contraction).
 *   dy       [T][B][H] fp32 gradient w.r.t. the stack output (row stride dy_ld).
 *   dy_a / dy_w / dy_k (optional, dtype bf16; GRU: rows of finished sequences must be zero in dy_a): the same gradient given as a product  dy_t = dy_a[t] . dy_w^T  with
 *            dy_a [T][B][dy_a_ld] dtype (e.g. the logit gradients of TimeDistributed(Linear), models.py:157) and dy_w [H][dy_w_ld] dtype
 *            (the Linear's weight, transposed), both zero-padded to dy_k columns, dy_k a multiple of 128: the TOP layer's cell contracts it
 *            as its second K-segment (that cell has no layer above), so the [T, B, H] fp32 tensor is never written or read.
 *            dy and dy_a may both be given (they add).
 *   dG[l]    [T][B][ldg]  dtype  (out)  pre-activation gradients (G*H valid columns); dG[0] is also the gradient of add0.
 *   dstate[l] [2][B][H] fp32 scratch (ping-pong dc).
 * Weight / bias / input gradients are then mvae_gemm_tn(dG[l], hs[l-1] / hs[l] shifted by one step) and mvae_colsum_t(dG[l]).
 */
typedef struct {
  int cell, dtype, layers, T, B, H;
  const void* w_hhT[MVAE_MAX_LAYERS]; int64_t ldw_hhT[MVAE_MAX_LAYERS];
  const void* w_ihT[MVAE_MAX_LAYERS]; int64_t ldw_ihT[MVAE_MAX_LAYERS];   /* w_ihT[0] unused */
  const int32_t* lengths;
  const float* dy; int64_t dy_ld;
  const void* dy_a; int64_t dy_a_ld; const void* dy_w; int64_t dy_w_ld; int dy_k;   /* optional (NULL): see above */
  const float* dh_last[MVAE_MAX_LAYERS];                                  /* optional fp32 [B,H] gradient w.r.t. the final hidden state */
  const void* hs[MVAE_MAX_LAYERS]; int64_t ldh;
  const void* h0[MVAE_MAX_LAYERS]; int64_t ldh0;
  const void* cs[MVAE_MAX_LAYERS];
  const void* gates[MVAE_MAX_LAYERS];
  void* dG[MVAE_MAX_LAYERS]; int64_t ldg;                                  /* row stride of dG (>= G*H; pad it off powers of two) */
  void* dGh[MVAE_MAX_LAYERS];                                              /* GRU only: W_hh-side gradient rows */
  float* dstate[MVAE_MAX_LAYERS];                                         /* fp32 [2][B][H] ping-pong: LSTM dc, GRU dh carry */
  float* dh0[MVAE_MAX_LAYERS];                                            /* optional out: gradient w.r.t. h0 (GRU decoder_lat path) */
  void* split_ws; size_t split_ws_bytes;                                  /* optional scratch of mvae_rnn_bwd_workspace(d) bytes: enables the split-K
                                                                             schedules (fp32 partial tiles of dh summed across workgroups) */
  /* inter-layer dropout of the forward pass (same mask / seed / p): the gradient that layer l receives from layer l+1
   * (dG^{l+1}_t . W_ih^{l+1}) is multiplied by keep(l,t,b,j) / (1 - p).  drop_p == 0: no dropout. */
  const uint8_t* drop_mask[MVAE_MAX_LAYERS];
  float drop_p; uint32_t drop_seed;
  /* Optional second schedule, the WEIGHTS-RESIDENT DATAFLOW backward (rnn_persist_bwd.hip), the counterpart of mvae_rnn_fwd_desc.persist_ws: ONE
   * persistent launch of 256 workgroups, each keeping a (64 hidden units x one gate's K-quarter) slice of [W_hh^T | W_ih^T] in registers for
   * all T steps; the dG tiles and the K-quarter partial sums travel between workgroups through write-through stores + flag words.  Serves
   * LSTM, bf16, 4 layers, H = 1024, B = 128 (256: two passes), ldg = 4H + 64, the output gradient given as dy (fp32, dy_ld = H; not dy_a), no
   * dh_last / dropout, on a 256-CU device with nothing else running beside it: mvae_rnn_bwd_persist_workspace() returns 0 for everything else.
   * persist_ws != NULL (that many bytes, 16-byte aligned) selects it; its first 16 bytes are the status record (word 0 != 0: a bounded spin
   * ran out, dG is invalid).  dstate is not written by this schedule (the carried dc never leaves the registers).
   * poison: as in mvae_rnn_fwd_desc (also honoured by the layer-concurrent row-resident backward, whose status record lives inside split_ws). */
  void* persist_ws; size_t persist_ws_bytes;
  float* poison;
  int no_spin;         /* as in mvae_rnn_fwd_desc */
} mvae_rnn_bwd_desc;

int mvae_rnn_bwd(const mvae_rnn_bwd_desc* d, void* stream, const void** status_out);     /* status_out: see mvae_rnn_fwd */
size_t mvae_rnn_bwd_persist_workspace(const mvae_rnn_bwd_desc* d);   /* 0: this shape / device is not served by the persistent schedule */
/* bytes of split_ws that let mvae_rnn_bwd pick any of its schedules for this shape (reads layers, T, B, H only). */
size_t mvae_rnn_bwd_workspace(const mvae_rnn_bwd_desc* d);

/* out[r] = sum_c X[r, c]  (X dtype, fp32 out; one wave per row, fixed order): bias gradients from dGT. */
int mvae_rowsum(int dtype, int R, int C, const void* X, int64_t ldx, float* out, int accumulate, void* stream);
/* out[b, :] = sum_t X[t, b, :]  (fp32 out): gradient of a time-invariant layer-0 input (models.py:163). */
int mvae_timesum(int dtype, int T, int B, int W, const void* X, float* out, void* stream);
/* out[n] = sum_m X[m, n] (fp32 in/out, deterministic two-stage): bias gradients of the dense / conv layers. */
size_t mvae_colsum_workspace(int M, int N);
int mvae_colsum(int M, int N, const float* X, int64_t ldx, float* out, void* ws, size_t ws_bytes, void* stream);
/* same for X of `dtype` read in 16-byte vectors (ldx multiple of 8, X 16-byte aligned): bias gradients from dG [T*B, G*H]. */
size_t mvae_colsum_t_workspace(int M, int N);
int mvae_colsum_t(int dtype, int M, int N, const void* X, int64_t ldx, float* out, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Conv1d(k) + bias + SELU, valid, stride 1 (K3; models.py:71-77 ConvSELU, :118-120, :129-131) as a
 * sliding-window MFMA GEMM (exact f32).  Activations are CHANNELS-LAST with padded channel count:
 *   x[b, w, c] = x[b * x_bs + w * ldx + c],  c < Cin valid, channels [Cin, ldx) must be zero;
 *   y[b, w, o] = y[(b * Wout + w) * ldy + o], Wout = W - k + 1 (columns [Cout, ldy) are not written).
 * The im2col row of (b, w) is the contiguous range x[b, w : w+k, :], so no im2col matrix exists; weights are
 * repacked per step by mvae_conv1d_pack_weights:
 *   wp [Cout][k * ldx]: wp[o][j * ldx + c] = w[o][c][j]          (forward / weight gradient; pads zero)
 *   wq [Cin][k * ldo] : wq[c][j * ldo + o] = w[o][c][k - 1 - j]  (input gradient; pads zero)      either may be NULL.
 * The LDS-direct main loop needs k * ldx (forward) / k * ldo (input gradient) to be a multiple of 32; other
 * sizes run on the generic loop.
 */
int mvae_conv1d_pack_weights(int Cin, int Cout, int k, const float* w /* [Cout, Cin, k] */, int ldx, float* wp, int ldo, float* wq,
                             void* stream);
size_t mvae_conv1d_selu_fwd_workspace(int B, int W, int ldx, int Cout, int k);
int mvae_conv1d_selu_fwd(int B, int W, int ldx, int64_t x_bs, int Cout, int k, const float* x, const float* wp, const float* bias,
                         float* y, int64_t ldy, void* ws, size_t ws_bytes, void* stream);
/* dy, y: [B * Wout, ldo] (gradient w.r.t. y / forward output; ldo % 4 == 0), x: the forward input (ldx % 4 == 0, x_bs % 4 == 0).
 * dzp: scratch [B, Wout + 2(k-1), ldo] -- receives the zero-padded pre-activation gradient.
 * dw [Cout, Cin, k] and db [Cout] in the reference's parameter layout; dx [B * W, lddx] channels-last (NULL: not needed; else wq required).
 * All pointers 16-byte aligned.  ws: scratch >= mvae_conv1d_selu_bwd_workspace bytes. */
size_t mvae_conv1d_selu_bwd_workspace(int B, int W, int Cin, int ldx, int Cout, int ldo, int k);
/* The same pair with the activation as a parameter (MVAE_ACT_NONE / _SELU / _RELU; same workspaces): Conv1d + ReLU of the
 * models2d.VAE encoder (models2d.py:12-14,24-27). */
int mvae_conv1d_act_fwd(int act, int B, int W, int ldx, int64_t x_bs, int Cout, int k, const float* x, const float* wp, const float* bias,
                        float* y, int64_t ldy, void* ws, size_t ws_bytes, void* stream);
/* OR-ed into `act` of mvae_conv1d_act_bwd: the input-gradient AND weight-gradient GEMMs multiply in the MVAE_F32X3 form (gradients only: the bf16 training mode,
 * whose decoder gradients are bf16-accurate anyway; the forward conv and the exact-fp32 mode never use it). */
#define MVAE_CONV_BWD_X3 0x100
int mvae_conv1d_act_bwd(int act, int B, int W, int Cin, int ldx, int64_t x_bs, int Cout, int ldo, int k, const float* dy, const float* y,
                        const float* x, const float* wq, float* dzp, float* dw, float* db, float* dx, int64_t lddx,
                        void* ws, size_t ws_bytes, void* stream);
/* dw == NULL in mvae_conv1d_*_bwd: only dzp and dx are produced; the layer's weight / bias gradient is then described by mvae_conv1d_dw_problem
 * as one problem of mvae_gemm_tn_f32_multi (dwp [Cout][k * ldx]: the gradient in the packed layout of wp; db: bias gradient) and brought into
 * the parameter layout dw [Cout][Cin][k] by mvae_conv1d_unpack_dw -- so that ALL parameter gradients of an encoder share one launch. */
int mvae_conv1d_dw_problem(int B, int W, int Cin, int ldx, int64_t x_bs, int Cout, int ldo, int k, const float* dzp, const float* x, float* dwp,
                           float* db, int x3, mvae_gemm_tn_f32_problem* out);
int mvae_conv1d_unpack_dw(int Cin, int Cout, int k, const float* dwp, int ldx, float* dw, void* stream);
int mvae_conv1d_selu_bwd(int B, int W, int Cin, int ldx, int64_t x_bs, int Cout, int ldo, int k, const float* dy, const float* y,
                         const float* x, const float* wq, float* dzp, float* dw, float* db, float* dx, int64_t lddx,
                         void* ws, size_t ws_bytes, void* stream);

/* dpre = dy * SELU'(pre) expressed through the forward OUTPUT y (in place on dy).  models.py:58-68. */
int mvae_selu_bwd(int64_t n, float* dy, const float* y, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Lambda / reparameterisation (K5; models.py:80-94).  mulv [B, 2*o]: mu | log_var (one stacked GEMM).
 * z = mu + exp(log_var/2) * eps        eps [B,o] is the already scaled noise (reference: 1e-2 * randn).
 * Noise source (models.py:92 draws it on the CPU default generator; SURVEY 8b: "eps|null, seed"):
 *   eps != NULL  the caller's noise is used as it is (parity tests, the reference's CPU RNG stream); scale / seed / offset ignored.  The pointer
 *                may be device memory or device-mapped PINNED HOST memory (hipHostMalloc / torch pin_memory), which the launch reads in place --
 *                no copy command in the stream; eps_out (optional) then receives a device copy for the backward pass;
 *   eps == NULL  the draw happens INSIDE this launch: element i of the [B, o] block is scale * n(seed, offset + i), n = a standard normal
 *                from the counter hash below (no generator state anywhere: the same (seed, offset) always yields the same block), and is
 *                written to eps_out [B, o] (required then), which is what mvae_lambda_bwd takes as eps.
 * Backward: dmulv[:, :o] = dmu + dz ; dmulv[:, o:] = dlogvar + dz * eps * 0.5 * exp(log_var/2).
 *
 * The counter normal n(seed, c), c a 64-bit element counter: s = H(seed, (c >> 31) ^ 0x6A09E667), w1 = H(s, 2c mod 2^32),
 * w2 = H(s ^ 0xBB67AE85, (2c mod 2^32) | 1) with H the counter hash of mvae_dropout_keep; u1 = ((w1 >> 8) + 0.5) / 2^24, u2 = (w2 >> 8) / 2^24,
 * n = sqrt(-2 ln u1) * cos(2 pi u2).  mvae_normal_words returns (w1, w2) computed on the HOST by the same function the kernels compile
 * (tests restate the draw from them); mvae_normal_fill writes out[i] = scale * n(seed, offset + i), i < n (mosesvae.py:159 / models2d.py:34
 * randn_like as a library op).
 */
int mvae_lambda_fwd(int B, int o, const float* mulv, const float* eps, float scale, uint32_t seed, uint64_t offset, float* eps_out,
                    float* z, float* mu, float* logvar, void* stream);
int mvae_normal_fill(int64_t n, float scale, uint32_t seed, uint64_t offset, float* out, void* stream);
void mvae_normal_words(uint32_t seed, uint64_t counter, uint32_t* words2);
int mvae_lambda_bwd(int B, int o, const float* mulv, const float* eps, const float* dz, const float* dmu,
                    const float* dlogvar, float* dmulv, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Output head (K8; models.py:157 Linear + nn.Softmax() over the class axis of the [B*L, C] view,
 * models.py:43-50).  logits [(t*B+b), ldl] fp32 -> recon [B, L, C] fp32 probabilities.
 * Backward: dlogits = p * (drecon - sum_c drecon*p), written in dtype as dl [(t*B+b), ldd] and, when dlT != NULL, transposed
 * dlT [C][ldT] (column t*B+b).  Columns C .. C8-1 (C8 = C rounded up to 8) of dl are written as zero; the K-padding columns C8 .. ldd-1
 * are the CALLER's to keep zero (allocate the buffer zeroed: this call may or may not touch them -- the tiled bf16 form does not).
 */
int mvae_softmax_tb_fwd(int B, int L, int C, const float* logits, int64_t ldl, float* recon, void* stream);
int mvae_softmax_tb_bwd(int dtype, int B, int L, int C, const float* recon, const float* drecon,
                        void* dl, int64_t ldd, void* dlT, int64_t ldT, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * ELBO of train.py:31-38, verbatim: max_len * BCELoss(mean)(recon, x) - 0.5*mean(1 + mu - logvar^2 - exp(mu))
 * (binary CE on the softmax outputs with the log clamp at -100; mu / logvar swapped in the KL term, as the
 * reference computes it).  loss_out[0] = total, [1] = xent term, [2] = KL term.  ONE launch: per-block partial sums, a ticket counter, and
 * the block drawing the last ticket sums the partials in a fixed order (deterministic).  ws >= mvae_bce_kl_loss_workspace() bytes, 4-byte
 * aligned, private to this call site; its LAST 16 BYTES (the ticket) must be zero before the first call -- every call leaves them zero again.
 * Backward: drecon = g*(max_len/n)*(p - t)/max(p(1-p),1e-12); dmu = g*(-0.5/m)(1 - exp(mu));
 * dlogvar = g*(0.5/m)*2*logvar, with g = *grad_out (device scalar) or 1 when NULL.
 */
size_t mvae_bce_kl_loss_workspace(int64_t n_recon, int64_t n_latent);
int mvae_bce_kl_loss_fwd(int64_t n_recon, const float* recon, const float* target, int64_t n_latent, const float* mu,
                         const float* logvar, float max_len, float* loss_out, void* ws, size_t ws_bytes, void* stream);
int mvae_bce_kl_loss_bwd(int64_t n_recon, const float* recon, const float* target, int64_t n_latent, const float* mu,
                         const float* logvar, float max_len, const float* grad_out, float* drecon, float* dmu,
                         float* dlogvar, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Device-side input pipeline: the encoded data set lives in HBM as uint8 indices [N, L]; a batch is the
 * rows `rows[0..B)` expanded to what MoleLoader.__getitem__ + default collate produce (data_loader.py:26-31): int64 indices
 * [B, L] and, when ohe != NULL, the float one-hot [B, L, C].
 */
int mvae_expand_indices(const uint8_t* store, const int64_t* rows, int B, int L, int C, int64_t* idx, float* ohe, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * MOSES GRU path (mosesvae.py:126-199).
 *   mvae_moses_latent_*: z = mu + exp(logvar/2)*eps and kl = 0.5*mean_b sum_j(exp(logvar) + mu^2 - 1 - logvar)  (mosesvae.py:158-162);
 *     backward adds dkl * d(kl) and an optional external dlogvar (logvar is a return value of VAE.forward).
 *   mvae_ce_loss_*: F.cross_entropy(y[:, :-1], x[:, 1:], ignore_index=pad) (mosesvae.py:193-197) on TIME-MAJOR logits
 *     (row t*B+b, leading dimension ldl); loss2[0] = loss, loss2[1] = number of counted tokens.  Backward writes dlogits in
 *     `dtype` with zero-padded columns V..ldd-1 (the next GEMM's K), adding an optional external dy given in [B,T,V] layout.
 *   mvae_permute102: [T,B,V] -> [B,T,V] (the `y` return value); mvae_relu_bwd: dy *= (y > 0) in place.
 *   mvae_mask_rows_tb: rows (t*B + b) with t >= lengths[b] of a time-major [T*B, ld] buffer := 0 (ld * sizeof(dtype) a multiple of 16).
 *     pad_packed_sequence (mosesvae.py:189) emits zeros at finished positions, so no gradient reaches h there: when an external gradient
 *     w.r.t. the returned logits was added to dl, those rows are cleared before mvae_rnn_bwd contracts dl as dy_a (the decoder_fc weight /
 *     bias gradients are taken from the unmasked dl first).
 */
size_t mvae_moses_latent_workspace(int B);     /* one float per sequence: per-row KL sums, added up in a fixed order */
/* eps == NULL: eps[i] = n(seed, offset + i) is drawn in the launch and written to eps_out (see mvae_lambda_fwd). */
int mvae_moses_latent_fwd(int B, int dz, const float* mu, const float* logvar, const float* eps, uint32_t seed, uint64_t offset, float* eps_out,
                          float* z, float* kl_out, void* ws, size_t ws_bytes, void* stream);
int mvae_moses_latent_bwd(int B, int dz, const float* mu, const float* logvar, const float* eps, const float* dz_in, const float* dkl,
                          const float* dlogvar_ext, float* dmu, float* dlogvar, void* stream);
size_t mvae_ce_loss_workspace(int B, int T);
int mvae_ce_loss_fwd(int B, int T, int V, const float* logits, int64_t ldl, const int64_t* x, int pad, float* loss2, void* ws, size_t ws_bytes,
                     void* stream);
int mvae_ce_loss_bwd(int dtype, int B, int T, int V, const float* logits, int64_t ldl, const int64_t* x, int pad, const float* loss2,
                     const float* grad_out, const float* dy_ext, void* dl, int64_t ldd, void* stream);
int mvae_permute102(int T, int B, int V, const float* in, float* out, void* stream);
/* One autoregressive sampling step behind the GRU step kernels (mosesvae.py:236-253): y = decoder_fc(h_top) (w_fc [V, ldw] dtype, V <= 64),
 * p = softmax(y / temp), w ~ multinomial(p, 1) with EXPLICIT randomness -- u(b) = hash(seed, step * B + b) / 2^32 with the counter hash of
 * mvae_dropout_keep; the sample is the first class k with cumsum_k(p) > u * sum(p) --, then the reference's bookkeeping
 * (x[b, step] = w unless the sequence has ended; a first <eos> sets end_pads[b] = step + 1 and eos_mask[b]) and the NEXT step's layer-0
 * addend row add_out[b] = table[w_b] + base[b] (fp32 [*, W], the embedding folded into the input projection).  w_out [B]: the sampled ids. */
int mvae_moses_sample_step(int dtype, int B, int V, int H, const void* h_top, int64_t ldh, const void* w_fc, int64_t ldw, const float* bias, float temp,
                           uint32_t seed, int step, int eos_id, const float* table, int W, const float* base, float* add_out, int64_t* x, int64_t x_ld,
                           int64_t* end_pads, uint8_t* eos_mask, int64_t* w_out, void* stream);
int mvae_relu_bwd(int64_t n, float* dy, const float* y, void* stream);
int mvae_mask_rows_tb(int dtype, int T, int B, int64_t ld, const int32_t* lengths, void* buf, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Optimiser surface (K14 + K15): torch.nn.utils.clip_grad_norm_(params, max_norm) (train.py:102) followed by
 * torch.optim.Adam.step() (train.py:81,104) on a FLAT fp32 parameter / gradient / m / v buffer.
 *   mvae_sumsq: partial[i] = sum of squares of chunk i (deterministic); norm_out[0] = sqrt(total) is
 *   produced on device by mvae_clip_adam itself from `partial` (no host synchronisation).
 *   coef = min(1, max_norm / (norm + 1e-6)) (max_norm <= 0: no clipping); grads are scaled by
 *   grad_scale first (1/world_size after an all-reduce SUM).
 *   A step whose global norm is NOT FINITE is skipped as a whole: p, m, v stay as they are and norm_out[1] (when norm_out has room for it:
 *   norm_out_len >= 2) is incremented by one.  That is how a failed persistent launch (mvae_rnn_*_desc.poison -> a NaN in a slot of g that
 *   mvae_sumsq covers) keeps the weights intact without any host synchronisation; the reference would turn every parameter into NaN at
 *   such a step (clip_grad_norm_ scales by NaN).  poison_reset (optional): that slot, set back to 0 by the launch for the next step.
 */
size_t mvae_sumsq_workspace(int64_t n);
int mvae_sumsq(int64_t n, const float* g, float* partial, void* stream);
int mvae_clip_adam(int64_t n, float* p, const float* g, float* m, float* v, const float* partial, int64_t npartial,
                   float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps, int step,
                   float* norm_out, int norm_out_len, float* poison_reset, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MVAE_H_ */
