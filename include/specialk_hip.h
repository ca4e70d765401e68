/*
 * specialk_hip.h — C ABI of libspecialk_hip.so, the MI355X (gfx950) kernels behind the
 * Transformer encoder-decoder hot path of thien/specialk.
 *
 * The reference has no FFI: its "operator layer" is stock PyTorch (torch.bmm, nn.Linear,
 * nn.LayerNorm, F.log_softmax, optim.Adam ...) called from the files under base/lib/transformer/.  Each
 * entry point below therefore cites the reference call site(s) whose arithmetic it replaces
 * (paths relative to the reference's base/ directory).
 *
 * Conventions
 *   - every function returns 0 on success, a negative SK_ERR_* otherwise;
 *     skhip_last_error() returns a thread-local message for the last failure;
 *   - all pointers are DEVICE pointers unless named host_*; the caller owns every buffer;
 *   - nothing here allocates, frees or synchronises; work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream);
 *   - matrices are row-major with explicit leading dimensions in ELEMENTS;
 *   - `dtype` is SK_F32 or SK_BF16 and names the storage type of activations / weights;
 *     accumulation, softmax and LayerNorm statistics are always fp32.
 */
#ifndef SPECIALK_HIP_H
#define SPECIALK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SK_F32 0
#define SK_BF16 1

#define SK_OK 0
#define SK_ERR_ARG (-1)     /* bad shape / alignment / unsupported combination */
#define SK_ERR_LAUNCH (-2)  /* hipLaunch / runtime error */
#define SK_ERR_WORKSPACE (-3)

/* GEMM operand layouts: C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] (+ epilogue)           */
#define SK_GEMM_NT 0 /* A stored [M][K], B stored [N][K]   (nn.Linear forward)            */
#define SK_GEMM_NN 1 /* A stored [M][K], B stored [K][N]   (input gradient)               */
#define SK_GEMM_TN 2 /* A stored [K][M], B stored [K][N]   (weight gradient)              */

/* GEMM epilogue flags (bit-or) */
#define SK_EPI_BIAS 1      /* + bias[n]                                                   */
#define SK_EPI_RELU 2      /* max(., 0)                                                   */
#define SK_EPI_RELU_MASK 4 /* multiply by (aux[m][n] > 0): backward of ReLU               */
#define SK_EPI_ACCUM 8     /* C = result + C_old                                          */
#define SK_EPI_RELU_BITS_OUT 16  /* with RELU: also write the sign pattern, one bit per element (bit n%16 of
                                    uint16 word (m*N + n)/16), to aux                           */
#define SK_EPI_RELU_MASK_BITS 32 /* as RELU_MASK, with aux holding such a bit pattern (1/16 of the bytes) */
#define SK_EPI_ROWSTATS 64  /* plain bf16 NT product with fp32 output only: aux (fp32 [M][ceil(N/64)][2]) receives, per
                              row and 64-column chunk, the maximum and the sum of exp(x - maximum): what a log-softmax
                              and a top-k over the rows need to know about the chunks they do not read */

const char* skhip_version(void);
const char* skhip_last_error(void);

/* ---- init ---------------------------------------------------------------------------
 * Optional: queries the current device once (CU count) so that launch heuristics do not
 * touch the runtime on the hot path.  Every entry point calls it lazily.                */
int skhip_init(void);

/* ---- CUs held by other work --------------------------------------------------------------
 * No counterpart in the reference (nn.DataParallel, train.py:385-388, has no notion of it): the large-tile GEMMs
 * run one workgroup per CU and size their tiles so that a launch fills whole rounds of the chip.  When another
 * kernel holds CUs for as long as they run — RCCL's all-reduce kernels during the backward pass of a data-parallel
 * step — a launch planned for all CUs needs a second, almost empty round.  n = the CUs to leave out of that
 * planning (0 = none, the default; SKHIP_RESERVE_CUS sets the start value).  Tile shapes, K-split counts and the
 * chunking of column sums and LayerNorm-gradient partials follow the planned CU count, so results with and without
 * a reserve agree to fp32 summation order (bf16 outputs: to one rounding step), not bit for bit; with a fixed
 * reserve every call is bit-reproducible.  Not thread-safe against concurrent GEMM calls.                      */
int skhip_set_reserved_cus(int n);

/* ---- GEMM ---------------------------------------------------------------------------
 * Replaces: nn.Linear w_qs/w_ks/w_vs/fc (lib/transformer/SubLayers.py:46-48,62), the
 * Conv1d(k=1) pair of the FFN (SubLayers.py:73-74,81), the generator projection
 * (lib/transformer/Models.py:252,278) and their autograd backward.
 * in_dtype: storage of A and B; out_dtype: storage of C (bf16 inputs may write fp32, used
 * for weight gradients).  bias is always fp32 [N] (the master parameter).  aux (RELU_MASK)
 * has C's shape, ldc and in_dtype; the two *_BITS forms (bf16, NT, N % 64 == 0, 16-byte aligned
 * operands) use a dense [M][N/16] uint16 pattern instead.
 * workspace: optional fp32 scratch for deterministic split-K (may be NULL: no split).     */
int skhip_gemm(int layout, const void* A, const void* B, void* C, const void* bias, const void* aux,
               int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
               float alpha, int epilogue, int in_dtype, int out_dtype,
               void* workspace, int64_t workspace_bytes, void* stream);

/* Grouped weight gradients: n independent products  C_i (+)= A_i^T B_i  in one pass over the chip —
 * the autograd backward of every nn.Linear / Conv1d(k=1) weight of the layers (SubLayers.py:46-48,62,
 * 73-74), which the reference's autograd engine runs as one small GEMM per weight.  A_i (K_i x M_i,
 * row stride lda_i) and B_i (K_i x N_i, ldb_i) are bf16 with rows contiguous, C_i (M_i x N_i, ldc_i) is
 * fp32.  problems: HOST array of n x 12 int64 {A, B, C, M, N, K, lda, ldb, ldc, accumulate, colsum,
 * colsum_accumulate}; operand addresses, M, N, lda, ldb must be multiples of 16 bytes / 8 elements.  colsum
 * (optional, 0 = none) is an fp32 [M_i] vector that receives (or, with colsum_accumulate, is added) the column
 * sums of A_i: with A_i = dy these are the bias gradient of the same layer (the autograd backward of the
 * `+ bias` in nn.Linear), computed on the matrix pipe beside the weight gradient instead of in a pass of its
 * own over dy.  workspace: fp32 scratch for the K-split slabs of the tiles that do not fill a whole round of
 * the chip (NULL: no split).  Deterministic (fixed summation order).                                  */
int skhip_gemm_grouped_tn(const int64_t* problems, int n, void* workspace, int64_t workspace_bytes, void* stream);

/* Column sums: out[n] (+)= sum_m X[m][n]  (bias gradients, the autograd backward of the
 * `+ bias` in nn.Linear, SubLayers.py:46-48,62).  out is fp32.  workspace: optional fp32
 * scratch (>= 64*N floats lets the rows be summed in parallel chunks, combined in a fixed
 * order); required when accumulate != 0.                                                  */
int skhip_colsum(const void* X, float* out, int64_t M, int64_t N, int64_t ldx, int accumulate,
                 int dtype, void* workspace, int64_t workspace_bytes, void* stream);

/* skhip_colsum without a workspace argument and without any allocation.  No counterpart in the
 * reference: it exists for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  The chunk partials go to a 2 MiB array built
 * into the library's code object (one copy per GPU, held until the process ends), so N is limited to
 * SK_COLSUM_NOALLOC_MAX_N; larger N returns SK_ERR_ARG.  Results are bit-identical to skhip_colsum
 * called with a 2 MiB workspace.  All calls on one GPU share the array: calls that overlap in time on
 * one GPU (different streams) race; calls on one stream do not.                                   */
#define SK_COLSUM_NOALLOC_MAX_N 65536
int skhip_colsum_noalloc(const void* X, float* out, int64_t M, int64_t N, int64_t ldx, int accumulate,
                         int dtype, void* stream);

/* ---- attention ----------------------------------------------------------------------
 * Replaces ScaledDotProductAttention.forward (lib/transformer/Modules.py:17-28) together
 * with the head split / merge copies and the h-fold mask repeat of MultiHeadAttention
 * (lib/transformer/SubLayers.py:50-60), and their backward.
 * Q is read as Q[b][i][head*dk + c] = Q[(b*Lq + i)*ldq + head*dk + c]; same for K, V, O
 * (so a fused [q|k|v] projection output is consumed in place by pointing Q, K, V into it).
 * Masking (lib/transformer/Models.py:49-69,173-177): key j of batch b is masked when
 * key_pad[b*Lk + j] != 0, and, if causal, when j > i.  dense_mask (optional, uint8
 * [B][Lq][Lk], nonzero = masked) is OR-ed in for callers that hold an arbitrary mask.
 * Fully masked rows produce NaN like the reference's softmax.
 * p_drop > 0 applies dropout to the probabilities (Modules.py:25) with a counter-based
 * generator keyed by (seed, b, head, i, j); backward regenerates the same mask.
 * LSE [B][h][Lq] fp32 receives the row log-sum-exp (of the scaled, masked scores), which
 * backward needs.  P_out (optional, in `dtype`, [h][B][Lq][Lk] head-major like the
 * reference's returned attn, after dropout) is only written when non-NULL.              */
int skhip_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, void* P_out,
                   const uint8_t* key_pad, const uint8_t* dense_mask,
                   int B, int n_head, int Lq, int Lk, int dk, int dv,
                   int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                   float scale, int causal, float p_drop, uint64_t seed,
                   int dtype, void* stream);

/* dQ/dK/dV are written (not accumulated) with the same addressing as Q/K/V.
 * delta_ws: fp32 scratch [B][h][Lq].                                                     */
int skhip_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                   const float* LSE, void* dQ, void* dK, void* dV, float* delta_ws,
                   const uint8_t* key_pad, const uint8_t* dense_mask,
                   int B, int n_head, int Lq, int Lk, int dk, int dv,
                   int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                   int64_t lddq, int64_t lddk, int64_t lddv,
                   float scale, int causal, float p_drop, uint64_t seed,
                   int dtype, void* stream);

/* Variable-length ("packed") batches: the rows of batch b are rows q_offsets[b] ..
 * q_offsets[b+1]-1 of Q / O / dQ and k_offsets[b] .. k_offsets[b+1]-1 of K / V / dK / dV
 * (int32 [B+1], device memory), with no padding rows between sentences, so key padding
 * (lib/transformer/Models.py:49-57) is implied by the lengths; max_Lq / max_Lk are the longest
 * sequences of the batch.  LSE and delta_ws stay [B][h][max_Lq]; entries of rows beyond a
 * sentence's length are not written.  Otherwise as skhip_attn_fwd / skhip_attn_bwd (which have
 * no counterpart of their own in the reference either: the reference pads).              */
int skhip_attn_varlen_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE,
                          const int32_t* q_offsets, const int32_t* k_offsets,
                          int B, int n_head, int max_Lq, int max_Lk, int dk, int dv,
                          int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                          float scale, int causal, float p_drop, uint64_t seed,
                          int dtype, void* stream);

int skhip_attn_varlen_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                          const float* LSE, void* dQ, void* dK, void* dV, float* delta_ws,
                          const int32_t* q_offsets, const int32_t* k_offsets,
                          int B, int n_head, int max_Lq, int max_Lk, int dk, int dv,
                          int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                          int64_t lddq, int64_t lddk, int64_t lddv,
                          float scale, int causal, float p_drop, uint64_t seed,
                          int dtype, void* stream);

/* ---- residual + dropout + LayerNorm + pad-row zeroing --------------------------------
 * y = LayerNorm(res + dropout_p(x)) * gamma + beta, rows with row_pad != 0 set to 0.
 * Replaces `self.layer_norm(self.dropout(out) + residual)` (SubLayers.py:62-63,83-84) and
 * the `*= non_pad_mask` that follows every sublayer (lib/transformer/Layers.py:22-58).
 * gamma/beta fp32; mean/rstd fp32 [T] saved for backward; row_pad may be NULL.            */
int skhip_add_ln_fwd(const void* x, const void* res, const float* gamma, const float* beta,
                     const uint8_t* row_pad, void* y, float* mean, float* rstd,
                     int64_t T, int d, float eps, float p_drop, uint64_t seed,
                     int dtype, void* stream);

/* dx = grad wrt x (through dropout), dres = grad wrt res (may alias dx when p_drop == 0).
 * dgamma/dbeta fp32 [d] are ACCUMULATED when accumulate != 0, else overwritten.
 * dx_colsum (optional fp32 [d]) receives sum_rows dx, i.e. the bias gradient of the linear
 * layer whose output x is (fc / w_2, SubLayers.py:62,81), saving a separate reduction pass.
 * partial_ws: fp32 scratch of at least skhip_add_ln_bwd_ws_floats(d) floats.              */
int64_t skhip_add_ln_bwd_ws_floats(int d);
int skhip_add_ln_bwd(const void* dy, const void* x, const void* res, const float* gamma,
                     const uint8_t* row_pad, const float* mean, const float* rstd,
                     void* dx, void* dres, float* dgamma, float* dbeta, float* dx_colsum,
                     float* partial_ws, int64_t T, int d, float p_drop, uint64_t seed, int accumulate,
                     int dtype, void* stream);

/* ---- row-panel products (d_model = 512, bf16) -----------------------------------------------
 * The activation-sized products of the layers with the token ROWS as the unit of work: a workgroup keeps
 * SK_ROWPANEL_ROWS rows in LDS, every wave owns 64 of 512 output columns and streams its slice of the weights
 * (stored FRAGMENT-MAJOR by skhip_cast2d_multi, meta flag bits 2-3 = 2) from the L2 straight into the MFMA operand
 * registers.  25.6 k rows are 247 workgroups: one round of the chip.  Results equal skhip_gemm's (+ skhip_add_ln_fwd's)
 * up to fp32 summation order.  All operands 16-byte aligned, row strides multiples of 8 elements; any T (the callers
 * use these entry points when T / SK_ROWPANEL_ROWS fills the chip, see ops.py).
 *
 * skhip_rowpanel_gemm: c [T][N] = epilogue(a [T][512] w^T), w [N][512] fragment-major, N a multiple of 512 — nn.Linear /
 * Conv1d(k=1) forward and input gradient with a 512-deep contraction (SubLayers.py:46-48,62,73,80 and their autograd
 * backward).  epilogue: 0, SK_EPI_BIAS, SK_EPI_BIAS | SK_EPI_RELU | SK_EPI_RELU_BITS_OUT (signs receives the pattern),
 * SK_EPI_RELU_MASK_BITS (signs is read), SK_EPI_ACCUM (c += ...).  The pattern has the row-panel kernels' OWN layout
 * (two bits per element in lane order, skhip_rowpanel_pattern_bytes(T, N) bytes, not skhip_gemm's): it is only read
 * back by this entry point, in a call with the same T and N.
 * skhip_rowpanel_gemm_ln: N = 512 and the sub-layer tail as the epilogue — replaces `self.layer_norm(self.dropout(
 * self.fc(output)) + residual)` (SubLayers.py:62-63) = skhip_gemm + skhip_add_ln_fwd:  sub_out = a w^T + bias [T][512]
 * (stored: the backward pass reads it), y = LayerNorm(res + dropout_p(sub_out)) gamma + beta with pad rows zeroed,
 * mean / rstd [T] as skhip_add_ln_fwd leaves them.                                                           */
#define SK_ROWPANEL_ROWS 104
int skhip_rowpanel_rows(void);
int64_t skhip_rowpanel_pattern_bytes(int64_t T, int64_t N);
/* Diagnostics (no counterpart in the reference): with a device buffer of ceil(T / SK_ROWPANEL_ROWS) * 8 * 32 uint64 set,
 * every wave of the row-panel kernels leaves shader-clock readings at its phase boundaries there; NULL (the start
 * value) turns it off.  Process-wide, not thread-safe.                                                         */
int skhip_rowpanel_set_stamps(void* buffer);
int skhip_rowpanel_gemm(const void* a, int64_t lda, const void* w_fm, const float* bias, void* c, int64_t ldc,
                        void* signs, int64_t T, int64_t N, int epilogue, void* stream);
int skhip_rowpanel_gemm_ln(const void* a, int64_t lda, const void* w_fm, const float* bias, const void* res,
                           int64_t ldres, void* sub_out, const float* gamma, const float* beta,
                           const uint8_t* row_pad, void* y, float* mean, float* rstd, int64_t T, float eps,
                           float p_drop, uint64_t seed, void* stream);

/* ---- embeddings ----------------------------------------------------------------------
 * out[t] = word_emb[seq[t]] + pos_table[pos[t]]   (lib/transformer/Models.py:117-118,
 * 181-182; no sqrt(d) scaling).  Tables are fp32 (master weights); out is `dtype`.        */
int skhip_embed_fwd(const int64_t* seq, const int64_t* pos, const float* word_emb,
                    const float* pos_table, void* out, int64_t T, int d, int64_t n_vocab,
                    int64_t n_pos, int dtype, void* stream);
/* dword_emb[seq[t]] += dout[t] for seq[t] != pad_id (nn.Embedding(padding_idx=PAD));
 * dword_emb fp32, caller zeroes it (or lets several calls accumulate for tied tables).    */
int skhip_embed_bwd(const int64_t* seq, const void* dout, float* dword_emb, int64_t T, int d,
                    int64_t n_vocab, int64_t pad_id, int dtype, void* stream);

/* Row bookkeeping of a right-padded id batch seq [B][L] (the output of collate_fn,
 * core/dataset.py:21-46) that is to be processed without its padding rows — the information the
 * reference carries as masks (get_non_pad_mask / get_attn_key_pad_mask, lib/transformer/Models.py:21-57):
 * lens[b] = max(#non-PAD ids of row b - trim, 0); off[B+1] = exclusive prefix sums (off[B] = n rows);
 * idx[r] (capacity B*L, first n written) = b*L + i for packed row r = off[b] + i;
 * facts[4] = {1 if every row is right-padded, min_b(#non-PAD - trim), max_b lens, n}.           */
int skhip_pack_plan(const int64_t* seq, int B, int L, int trim, int64_t pad_id, int32_t* lens,
                    int32_t* off, int64_t* idx, int64_t* facts, void* stream);

/* skhip_pack_plan for up to 4 batches in one pair of launches (a step's source and target batch), with the range
 * check that nn.Embedding makes on the host (lib/transformer/Models.py:117-118,181-183: an id or position outside its
 * table raises IndexError there).  desc: HOST table of 12 int64 per batch: {seq address, pos address or 0, row stride
 * of seq, row stride of pos (elements), B, L, trim, n_vocab (<= 0: ids not checked), n_pos, lens address, off
 * address, idx address}; lens / off / idx as in skhip_pack_plan.  B, L >= 1.
 * facts [n][6] (device) = {right-padded, min_b(#non-PAD - trim), max_b lens, n rows, every id in [0, n_vocab),
 * every position in [0, n_pos)}.                                                                              */
int skhip_pack_plans(const int64_t* desc, int n, int64_t pad_id, int64_t* facts, void* stream);

/* Batch collation on the device (collate_fn, core/dataset.py:21-46): the B sentences of a batch arrive as one
 * flat id array (sentence b = flat[off[b] .. off[b+1]-1], off[B+1] int64) and are written right-padded with
 * pad_id into seq [B][L] (L >= the longest sentence; longer sentences are cut at L), with pos [B][L] =
 * index + 1 on real ids and 0 on padding.                                                     */
int skhip_collate(const int64_t* flat, const int64_t* off, int B, int L, int64_t pad_id, int64_t* seq,
                  int64_t* pos, void* stream);

/* ---- label-smoothed softmax cross-entropy, forward + backward + accuracy -------------
 * Replaces calculate_loss / performance (lib/TransformerModel.py:340-375).
 * logits [T][V] (ld = ldl) are multiplied by `logit_scale` first (x_logit_scale,
 * lib/transformer/Models.py:262-264,278).  smoothing_eps = 0 gives plain cross_entropy with
 * ignore_index = pad_id, reduction = "sum".
 * out_stats (double[3], caller zeroes): [0] += loss sum, [1] += n_correct, [2] += n_non_pad.
 * dlogits (may alias logits; may be NULL for evaluation) receives
 * grad_scale * logit_scale * (softmax - q) on non-pad rows and 0 on pad rows; when
 * grad_scale_dev (one float in device memory, may be NULL) is given, the gradient is multiplied by
 * it as well — the classifier loss that multiplies the reconstruction loss in
 * train_decoder.py:471, without a host read of it.                                        */
int skhip_xent_fwd_bwd(const void* logits, const int64_t* gold, void* dlogits, double* out_stats,
                       int64_t T, int64_t V, int64_t ldl, float logit_scale, float smoothing_eps,
                       int64_t pad_id, float grad_scale, const float* grad_scale_dev, int dtype,
                       void* stream);

/* ---- style-transfer classifier path -------------------------------------------------
 * The pieces of classifier_loss (train_decoder.py:385-424) and ConvNet.forward
 * (classifier/onmt/CNNModels.py:48-65) that are not matrix products (those go through skhip_gemm).
 *
 * skhip_softmax_rows: y[r][:] = softmax(x[r][:V]) (nn.Softmax(dim=1), train_decoder.py:392,404);
 * y may alias x.                                                                          */
int skhip_softmax_rows(const void* x, void* y, int64_t rows, int64_t V, int64_t ldx, int64_t ldy,
                       int dtype, void* stream);

/* The classifier's input window and its convolution windows in one pass (train_decoder.py:408-418:
 * time-major, zero-padded or trimmed to S = sequence_length steps; CNNModels.py:53-58: Conv2d with
 * kernel (filter_size, 1) over all E channels).  emb [emb_rows][E] (ld = lde) holds the embedded
 * probabilities of the target positions: row b*stride_b + i*stride_t of a padded batch when off is
 * NULL (batch-major: L, 1; time-major as ConvNet.forward receives it: 1, B), or rows
 * off[b] .. off[b+1]-1 of sentence b when the batch is stored without padding; positions past a
 * sentence's rows and below min(L, S) then read row `pad_row` (what a zeroed decoder output embeds
 * to), positions from L to S are zero.  cols [B*(S-filter_size+1)][filter_size*E] (ld = ldcols):
 * cols[b*P + t][j*E + e] = step t+j, channel e of sentence b, so the convolution is
 * cols x W^T with W[f][j*E + e] = conv1.weight[f][e][j][0].                               */
int skhip_conv_windows(const void* emb, const int32_t* off, void* cols, int B, int L, int S,
                       int filter_size, int E, int64_t lde, int64_t ldcols, int64_t emb_rows,
                       int64_t pad_row, int64_t stride_b, int64_t stride_t, int dtype, void* stream);

/* ReLU'd convolution outputs h [B*P][F] (ld = ldh) -> max over the P positions (MaxPool3d,
 * CNNModels.py:36-38,60) -> Linear(F, 1) with fp32 w[F], bias[1] -> sigmoid = y[B] (fp32) ->
 * nn.BCELoss against `target_label` (train_decoder.py:424; logarithms clamped at -100):
 * terms[B] (fp32 scratch) and their mean in loss[0].  Dropout is the identity (eval mode,
 * train_decoder.py:350).                                                                  */
int skhip_cnn_head(const void* h, const float* w, const float* bias, float* y, float* terms,
                   float* loss, int B, int P, int F, int64_t ldh, int target_label, int dtype,
                   void* stream);

/* ---- Adam ---------------------------------------------------------------------------
 * One launch over n tensors: torch.optim.Adam(betas, eps, no weight decay) as configured in
 * lib/TransformerModel.py:159-165, with the learning rate of lib/transformer/Optim.py:24-36
 * computed by the caller.  The five tables are DEVICE arrays of n entries: fp32 param,
 * fp32 grad, fp32 exp_avg, fp32 exp_avg_sq pointers, element counts; bf16_copy[i] (table may
 * be NULL, entries may be 0) receives the updated parameter rounded to bf16.
 * grad_scale multiplies every gradient first (1.0 for the reference's behaviour).        */
int skhip_adam_multi(const uint64_t* param_ptrs, const uint64_t* grad_ptrs, const uint64_t* m_ptrs,
                     const uint64_t* v_ptrs, const uint64_t* bf16_copy_ptrs, const int64_t* counts,
                     int n, int64_t max_count, float lr, float beta1, float beta2, float eps,
                     int step, float grad_scale, void* stream);

/* dst = (bf16) src for n tensors (same table convention), used once after loading weights. */
int skhip_cast_multi(const uint64_t* src_f32_ptrs, const uint64_t* dst_bf16_ptrs,
                     const int64_t* counts, int n, int64_t max_count, void* stream);

/* Derived working copies of weight matrices, refreshed in one launch after the optimiser step:
 * entry t reads the fp32 matrix src[t] ([R][C], contiguous) and writes it, cast to bf16 (or kept
 * fp32), into dst[t] with row stride ld either as it is or transposed.  meta holds four int64 per
 * entry: R, C, ld, flags (bit 0: transpose, bit 1: fp32 destination).  Used to keep [Wq;Wk;Wv]
 * row-concatenations (one fused projection GEMM instead of SubLayers.py:46-48's three) and W^T
 * copies (input-gradient GEMMs with a K-contiguous weight operand).  max_tiles = the largest
 * ceil(R/32)*ceil(C/32) over the entries.
 * flags bits 2-3 = t in {1, 2}: the destination (bf16) is FRAGMENT-MAJOR with T = 2 t tiles per group (the row-panel
 * kernels above read T = 4): the [N][K] matrix (after the optional transpose; N % (16 T) == 0,
 * K % 32 == 0) is stored as [N / (16 T)][K' / 32][T][64][8] with K' = ld if ld > 0 (the entry is then a K slice of a
 * wider destination and dst points at its first K-step), else K, where tile ti of a group holds as its row
 * i the group's row 4 T (i / 4) + 4 ti + i % 4, and lane i + 16 g of a tile the elements k = 8 g .. 8 g + 7.   */
int skhip_cast2d_multi(const uint64_t* src_f32_ptrs, const uint64_t* dst_ptrs, const int64_t* meta,
                       int n, int64_t max_tiles, void* stream);
/* dst[i] = convert(src[i]); dtypes SK_F32 / SK_BF16.                                      */
int skhip_convert(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream);

/* ---- beam-search decode ---------------------------------------------------------------
 * Replace Translator.translate_batch's per-step work (lib/transformer/Translator.py:103-168)
 * and Beam.advance (lib/transformer/Beam.py:46-79).                                        */

/* Single-query attention over a KV cache (incremental replacement for re-running the
 * decoder on the whole prefix, Translator.py:130-140).  q [R][h*dk]; K/V cache
 * [R][Lmax][h*dk|h*dv] of which the first kv_len[r] (or `len` if kv_len NULL) entries are
 * valid; key_pad optional [R][Lmax].  out [R][h*dv].
 * kv_row_div >= 1: query row r reads cache row r / kv_row_div (and key_pad / kv_len row
 * r / kv_row_div), so the beams of one sentence share a single encoder-side K/V copy.
 * hist_row (optional, [R][Lmax] int32): position j of query row r's history lives in cache row
 * hist_row[r][j] (K, V and key_pad alike) — beam search re-parents hypotheses every step
 * (Translator.py:65-101 re-orders whole tensors); re-ordering this small table instead leaves
 * the caches in place.
 * new_k / new_v (optional, [R][ld_new] rows, heads at h*dk / h*dv): key and value of the newest
 * position (kv_len[r]-1, or len-1 when kv_len is NULL), taken from there and WRITTEN into the caches
 * (row hist_row[r][that position], or r) by this call — the cache append of the k / v projections
 * without launches of their own.  With kv_len given, `len` is only the upper bound of the history
 * (the launch then does not depend on the step number: a captured HIP graph can replay it).      */
int skhip_decode_attn(const void* q, void* Kc, void* Vc, void* out,
                      const int32_t* kv_len, const uint8_t* key_pad, const int32_t* hist_row,
                      const void* new_k, const void* new_v, int64_t ld_new, int R, int n_head, int len,
                      int Lmax, int dk, int dv, int64_t ldq, int64_t ldo, float scale,
                      int kv_row_div, int dtype, void* stream);

/* log_softmax over V + beam expansion + top-`bm` per sentence.
 * logits [n_sent*bm][V] fp32 (ld = V).  scores_in [n_sent][bm] fp32.
 * normalize != 0: rows are raw generator outputs and log_softmax is applied here;
 * normalize == 0: rows already hold log-probabilities (the argument of Beam.advance).
 * normalize == 2: as 1, with the hardware exponential (~1 ulp) in the normaliser instead of expf (about a third
 * faster; scores agree to ~1e-6, so near-ties may resolve differently than with normalize == 1).
 * first_step != 0: only beam 0 of each sentence is considered (Beam.py:51-54).
 * Outputs per sentence, sorted by descending score, ties to the lower flat index:
 * scores_out [n_sent][bm], prev_out [n_sent][bm] (source beam = flat_id / V),
 * tok_out [n_sent][bm] (flat_id - prev*V).  ws: fp32/int scratch of
 * skhip_beam_topk_ws_bytes(n_sent, bm, V) bytes.                                           */
int64_t skhip_beam_topk_ws_bytes(int n_sent, int bm, int64_t V);
int skhip_beam_topk(const float* logits, const float* scores_in, float* scores_out,
                    int32_t* prev_out, int32_t* tok_out, void* ws, int n_sent, int bm, int64_t V,
                    int first_step, int normalize, void* stream);

/* skhip_beam_topk (normalize = 1) for rows whose generator product was run with SK_EPI_ROWSTATS: `stats` is that
 * call's aux output (fp32 [n_sent*bm][ceil(V/chunk_cols)][2], chunk_cols = 64), or come from
 * skhip_generator_rowstats (chunk_cols = skhip_generator_chunk_cols()).  The normaliser comes from the statistics and
 * only the chunks that can hold one of a row's best bm tokens are read, instead of every row twice.  Same candidates,
 * same order and tie rule; scores agree with skhip_beam_topk to fp32 rounding (the exponentials are summed chunk by
 * chunk, with the hardware exponential).  chunk_cols in 1..64, V <= 1024 chunk_cols.  ws as for skhip_beam_topk.   */
int skhip_beam_topk_stats(const float* logits, const float* stats, const float* scores_in, float* scores_out,
                          int32_t* prev_out, int32_t* tok_out, void* ws, int n_sent, int bm, int64_t V,
                          int first_step, int chunk_cols, void* stream);

/* The generator product of a decode step with its chunk statistics, for d_model = K = 512 (bf16): logits [R][V] =
 * x [R][K] (row stride ldx) times w [V][K] transposed, fp32 — `self.model.generator(dec_output)` of
 * Translator.predict_word (lib/transformer/Translator.py:136-137; the log_softmax of :137 is taken by
 * skhip_beam_topk_stats / skhip_beam_step from the statistics) — and stats [R][ceil(V/c)][2], c =
 * skhip_generator_chunk_cols(): per row and c-column chunk the maximum and the sum of exponentials against it.  Same
 * results as skhip_gemm(SK_GEMM_NT, SK_EPI_ROWSTATS) up to the chunk width; a workgroup owns 144 vocabulary entries
 * and every row, so the weight is read once and the launch is one round of the chip for V up to 36 864.
 * x_layout: SK_ROWS_PLAIN (row r at x + r * ldx) or SK_ROWS_FRAGMENT_MAJOR, the order the kernel's MFMA operands are
 * read in, which skhip_decode_stack can write (ldx ignored): 16-byte piece c (eight elements) of row r at element
 * offset (((r / 16) * 16 + c / 4) * 64 + 16 * (c % 4) + r % 16) * 8, in a buffer of ceil(R / 16) * 16 rows — a load
 * instruction of a wave then fetches 1 KB of consecutive bytes instead of 64 bytes from each of 16 rows 1 KB apart,
 * which costs the plain layout about a third more time.                                                             */
#define SK_ROWS_PLAIN 0
#define SK_ROWS_FRAGMENT_MAJOR 1
int skhip_generator_chunk_cols(void);
int skhip_generator_rowstats(const void* x, int64_t ldx, int x_layout, const void* w, float* logits, float* stats,
                             int64_t R, int64_t V, int64_t K, void* stream);

/* One decode step of the WHOLE decoder stack in one launch (bf16; d_model 512, 8 heads of 64, d_inner 2048 — other
 * shapes: SK_ERR_ARG, use the entry points above op by op).  Replaces, for the newest position of every beam row,
 * Decoder.forward (lib/transformer/Models.py:173-196) as Translator.predict_word calls it once per step
 * (lib/transformer/Translator.py:121-140): embedding, and per layer self-attention over the cached prefix
 * (appending the new key / value to the caches), encoder-decoder attention, feed-forward, with their residuals and
 * LayerNorms.  Teams of 8 workgroups own 32 rows each for the whole stack and exchange slices through the
 * workspace; no other synchronisation, fixed summation orders.
 * layers: HOST table, n_layers x 22 device addresses per layer (consumed during the call):
 *   {Wqkv bf16 [1536][512] (rows q | k | v), bqkv f32 [1536], Wo bf16 [512][512], bo, ln1 gamma, ln1 beta,
 *    Wq_enc bf16 [512][512], bq_enc, Wo_enc bf16 [512][512], bo_enc, ln2 gamma, ln2 beta,
 *    W1 bf16 [2048][512], b1, W2 bf16 [512][2048], b2, ln3 gamma, ln3 beta,
 *    — the six weight MATRICES not in their [rows][K] layout but fragment-major, [rows / 16][K / 32][64][8]: element
 *    (row 16 a + r, column 32 b + 8 g + j) at (((a * K / 32 + b) * 64 + r + 16 g) * 8 + j), which is the order the
 *    kernel's matrix instructions consume them in (whole 128-byte lines per load instruction) —
 *    Kcache, Vcache bf16 [R][Tc][512], encK, encV bf16 [n_sent][Ls][512]}   (all 16-byte aligned).
 * tok, pos int64 [R]; cur_len int32 [R] (history length of row r including the new position); row_pad uint8 [R];
 * pad_hist uint8 [R][Tc]; hist int32 [R][Tc] (history-row table as for skhip_decode_attn); src_key_pad uint8
 * [n_sent][Ls]; dec_out bf16 [R][512] (out_layout SK_ROWS_PLAIN) or, with SK_ROWS_FRAGMENT_MAJOR, the same rows in the
 * order skhip_generator_rowstats reads them in (see there), in a buffer of ceil(R / 16) * 16 rows of which only rows
 * < R are written.  R <= 32 x min(32, CUs / 8) rows, sentence of row r = r / bm.
 * workspace: skhip_decode_stack_ws_bytes(R) bytes, 256-byte aligned, which the caller zeroes ONCE before the first
 * call of a decode run (the team barriers' words: the first 1024 bytes and the last 4096); parity: 0 / 1, alternating
 * from call to call (the launch of one parity resets the barrier words and XCC masks of the other).  The first 1024 bytes hold, as int32 words: [0, 64) team
 * counters, word 64 (byte 256) the ERROR WORD, [192, 256) per-team masks of the XCCs the team's workgroups ran on.
 * Error word after a step: 0 = fine; 1 = a team member stopped waiting for its team; 2 = the workgroups of a team
 * ran on more than one XCC (the exchange through their common L2 is then not coherent).  Non-zero: the results of
 * the step, and of every later step built on it, are invalid — run the batch again through the op-by-op entry
 * points (specialk_amd.decode.BeamDecoder.run does).                                                             */
int64_t skhip_decode_stack_ws_bytes(int R);
int skhip_decode_stack(const int64_t* layers, int n_layers, const int64_t* tok, const int64_t* pos,
                       const float* word_emb, const float* pos_tab, const int32_t* cur_len, const uint8_t* row_pad,
                       const uint8_t* pad_hist, const int32_t* hist, const uint8_t* src_key_pad, void* dec_out,
                       void* workspace, int64_t workspace_bytes, int R, int bm, int Tc, int Ls, int d_model, int n_head,
                       int d_head, int d_inner, float scale, float eps, int parity, int out_layout, void* stream);

/* The bookkeeping between two decoder steps of the beam search, for all sentences, in one launch:
 * Beam.advance's state update (lib/transformer/Beam.py:64-79: scores, prev_ks, next_ys, the EOS test on the
 * best beam), the active-sentence bookkeeping of Translator.py:142-168 and the next step's decoder inputs
 * (Translator.py:113-128).  R = n_sent*bm rows.  Step number t is read per row from cur_len[r] (and advanced),
 * so that a captured step can be replayed.  new_scores / prev / new_tok: outputs of skhip_beam_topk.
 * scores [n_sent][bm] is overwritten for sentences that had not finished; ys [max_len+1][R], ks [max_len][R]
 * record tokens and back-pointers; done / length [n_sent] int32; *n_done counts finished sentences.
 * hist_in / hist_out [R][Tc]: the history-row table of skhip_decode_attn before / after re-parenting.
 * tok, pos (int64 [R]), row_pad (uint8 [R]) and pad_hist[r][t] are the inputs of step t+1.            */
int skhip_beam_update(const float* new_scores, const int32_t* prev, const int32_t* new_tok, float* scores,
                      int32_t* ys, int32_t* ks, int32_t* done, int32_t* length, int32_t* n_done,
                      int32_t* cur_len, int64_t* tok, int64_t* pos, uint8_t* row_pad, uint8_t* pad_hist,
                      const int32_t* hist_in, int32_t* hist_out, int n_sent, int bm, int Tc, int max_len,
                      int pad_id, int eos_id, void* stream);

/* skhip_beam_topk_stats followed by skhip_beam_update in ONE launch (a workgroup per sentence, a wave per beam row):
 * Beam.advance and the bookkeeping around it as cited at those two (lib/transformer/Beam.py:46-79,
 * lib/transformer/Translator.py:113-168), with the same arguments (chunk_cols as there), bytes and results — new_scores / prev / new_tok
 * [n_sent][bm] are still written — and `scores` both the beams' scores going in and the updated state.  Inside a
 * replayed graph a step then crosses one kernel boundary for this work instead of three.                          */
int skhip_beam_step(const float* logits, const float* stats, float* new_scores, int32_t* prev, int32_t* new_tok,
                    void* ws, float* scores, int32_t* ys, int32_t* ks, int32_t* done, int32_t* length,
                    int32_t* n_done, int32_t* cur_len, int64_t* tok, int64_t* pos, uint8_t* row_pad,
                    uint8_t* pad_hist, const int32_t* hist_in, int32_t* hist_out, int n_sent, int bm, int64_t V,
                    int first_step, int Tc, int max_len, int pad_id, int eos_id, int chunk_cols, void* stream);

/* The hypotheses at the end of a search, for all sentences in one launch: Beam.get_hypothesis
 * (lib/transformer/Beam.py:103-110) walks the back-pointers of one beam on the host.  ys [steps+1][R] and ks [steps][R]
 * as skhip_beam_update records them (R = n_sent*bm), length [n_sent].  out[s][r][j] (int32, row stride out_ld >= steps)
 * = token j of the hypothesis of sentence s that ended as rank r < n_best, for j < min(length[s], steps); the rest of
 * a row is left as it was.  2 * steps * bm * 4 bytes of LDS per sentence: at most 64 KB (SK_ERR_ARG beyond).        */
int skhip_beam_backtrack(const int32_t* ys, const int32_t* ks, const int32_t* length, int32_t* out, int n_sent, int bm,
                         int n_best, int steps, int64_t out_ld, void* stream);

/* Reorder a cache after a beam step: dst[r] = src[src_row[r]] for rows of `row_elems`
 * elements (replaces re-building dec_seq from back-pointers every step, Beam.py:90-110,
 * and the index_select compaction, Translator.py:65-101).                                  */
int skhip_gather_rows(const void* src, void* dst, const int32_t* src_row, int64_t n_rows,
                      int64_t row_elems, int64_t src_stride, int64_t dst_stride, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPECIALK_HIP_H */
