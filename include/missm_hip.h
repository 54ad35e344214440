/* missm_hip.h - C ABI of the MI355X (gfx950) kernels behind the MissM-Benchmark forward/backward hot path.
 *
 * The reference (Fieldhunter/MissM-Benchmark) has no native code and no FFI: its hot path is PyTorch modules
 * (`languagebind.LanguageBind`, `src.model.baseline.finetune_model`) whose arithmetic PyTorch dispatches to
 * vendor kernels.  This library is what a binding for that path would bind instead; every entry point names the
 * reference call site whose dispatched kernels it replaces.  The Python host side (missm_benchmark_amd/_lib.py)
 * binds it with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP), plain `void*` / `float*` / `int*`; no framework types
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued, nothing syncs
 *   - `dtype`: 0 = fp32, 1 = bf16 (the activation / GEMM-operand type "T"); the residual stream, LayerNorm
 *     statistics, all parameters' master copies, gradients of parameters and the fusion tail are always fp32
 *   - return value: MISSM_OK (0) or a negative MISSM_ERR_* code; missm_last_error() returns a thread-local,
 *     NUL-terminated description of the last failure on the calling thread
 *   - kernels borrow the buffers for the duration of the enqueued work only (the caller owns all memory)
 */
#ifndef MISSM_HIP_H
#define MISSM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MISSM_OK 0
#define MISSM_ERR_INVALID (-1)
#define MISSM_ERR_LAUNCH (-2)
#define MISSM_ERR_NO_DEVICE (-3)

#define MISSM_F32 0
#define MISSM_BF16 1

/* epilogue activations of missm_gemm_nt */
#define MISSM_ACT_NONE 0
#define MISSM_ACT_QGELU 1   /* x * sigmoid(1.702 x)  (hidden_act default, configuration_image.py:191) */
#define MISSM_ACT_GELU 2    /* erf gelu */
#define MISSM_ACT_DQGELU 3  /* multiply by quick_gelu'(aux_in) */
#define MISSM_ACT_DGELU 4   /* multiply by gelu'(aux_in) */
#define MISSM_ACT_RELU 5

const char* missm_last_error(void);
/* ABI version of this header (bumped on any signature change). */
int missm_abi_version(void);
/* Number of visible HIP devices (0 when none); never initialises a device context. */
int missm_device_count(void);

/* `ngroups` (1..8) problems of ONE shape in one call: the same linear (or its weight gradient) of several shape-identical towers -
 * image / audio / depth / thermal (languagebind/__init__.py:75-85 encodes them one after the other).  Every operand of missm_gemm
 * becomes an array of `ngroups` pointers (optional ones: null array, or all entries set).  Where a grouped tile kernel covers the
 * shape (bf16; forward / dX form with a 256-row tile grid that fills the chip, weight-gradient form) the problems share ONE grid,
 * otherwise they are launched one after the other: results do not depend on which. */
int missm_gemm_grouped(int ngroups, const void* const* A, const void* const* B, void* const* C, int M, int N, int K, int lda, int ldb,
                       int ldc, int trans_a, int trans_b, float alpha, const float* const* bias, const float* const* resid,
                       const void* const* aux_in, void* const* aux_out, int ldaux, int act, int out_f32, int accumulate, int splitk,
                       float* const* colsum_a, int dtype, void* stream);
/* C[M,N] = alpha * A[M,K] . B[N,K]^T (+bias[N]) -> act -> (+resid) ; A, B of type `dtype`, fp32 accumulate (MFMA).
 * Output is `dtype` unless out_f32.  aux_out receives the pre-activation (for the backward), aux_in supplies it.
 * Replaces: nn.Linear inside CLIPAttention q/k/v/out_proj and CLIPMLP fc1/fc2
 * (languagebind/image/modeling_image.py:69,71,140-151), the patch-embed conv as a GEMM over unfolded patches
 * (languagebind/video/modeling_video.py:29-35,45-46) and all their autograd GEMMs. */
int missm_gemm_nt(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, float alpha,
                  const float* bias, const float* resid, const void* aux_in, void* aux_out, int ldaux, int act, int out_f32,
                  int accumulate, int dtype, void* stream);

/* General form: C[M,N] = alpha * op(A) . op(B) with the reduction index k,
 *   trans_a == 0: A is [M,K] (k contiguous)      trans_a == 1: A is [K,M] (stored as produced, e.g. dY for dW = dY^T X)
 *   trans_b == 0: B is [N,K] (k contiguous)      trans_b == 1: B is [K,N] (e.g. the weight itself for dX = dY W)
 * so weight and input gradients read activations / weights where they lie - no transposed copies.  (trans_a, trans_b) in
 * {(0,0), (0,1), (1,1)}.  splitk: 1 = off, 0 = auto (fp32 outputs whose tile grid cannot fill the chip: weight gradients -
 * tiny output, huge K), > 1 = number of K slices.  The slices park their partial tiles in a per-stream workspace the library
 * owns and a second kernel sums them in slice order (no atomics, bit-reproducible; alpha / bias / accumulate apply there);
 * the workspace is allocated / grown on first use, which synchronises `stream` once.  colsum_a (trans_a only, optional): colsum_a[m] +=
 * sum_k A[k][m] - the bias gradient rides in the weight-gradient GEMM as a ones-column (caller zeroes it). */
int missm_gemm(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int trans_a, int trans_b,
               float alpha, const float* bias, const float* resid, const void* aux_in, void* aux_out, int ldaux, int act,
               int out_f32, int accumulate, int splitk, float* colsum_a, int dtype, void* stream);

/* Diagnostic only: when set (device pointer to 8 x uint64 per workgroup), the next GEMM launches record per-workgroup
 * {start, first tile landed, main loop end, end, HW_ID, stores issued} stamps of the 100 MHz clock; NULL switches it off. */
void missm_gemm_set_debug_buffer(void* stamps);
/* Frees every split-K workspace (device-synchronises); they are re-created on demand. */
void missm_gemm_release_workspaces(void);

/* out[C, ldo] = in[R, C]^T zero-padded to ldo columns; optional colsum[C] += column sums (bias gradient). */
int missm_transpose_pad(const void* in, void* out, int R, int C, int ld, int ldo, float* colsum, int dtype, void* stream);
/* out[(row / div) % mod][c] += in[row][c]  (position / temporal embedding and bias gradients). */
int missm_colsum(const void* in, float* out, int R, int C, int ld, int div, int mod, int dtype, void* stream);
/* fp32 master weight [R,C] -> `dtype` shadow dst[R,C] and/or transposed shadow dst_t[C,R] (either may be NULL). */
int missm_cast_weight(const float* src, void* dst, void* dst_t, int R, int C, int dtype, void* stream);

/* The same for every weight block of a tower in one launch: `tiles` is a device array of ntiles records
 * { const float* src; void* dst; void* dst_t; int R, C, r0, c0; } (40 bytes, one 64x64 tile each). */
int missm_cast_weights_batched(const void* tiles, int ntiles, int dtype, void* stream);

/* LayerNorm over fp32 rows.  Input row = x[row * in_mul + in_off[row]] (row gather for CLS / EOT pooling);
 * if `add` is given, add[(row / add_div) % add_mod] is added first and the sum written back to x_wb
 * (temporal_embedding).  Saves mean / rstd.  Replaces nn.LayerNorm at image/modeling_image.py:70,72,82,465,604,606,
 * src/model/baseline.py:48. */
int missm_layernorm_fwd(const float* x, float* x_wb, const float* add, int add_div, int add_mod, int in_mul, const int* in_off,
                        const float* gamma, const float* beta, void* y, float* mean, float* rstd, int rows, int cols, float eps,
                        int out_dtype, void* stream);
/* dx (fp32, same row mapping as x; += when accumulate) ; dgamma/dbeta atomically accumulated (caller zeroes).
 * dy row = row / dy_div, scaled by dy_scale (mean over frames).  dx_cast (optional) receives the updated dx rows
 * converted to dy's dtype: the GEMM operand of the next backward block. */
int missm_layernorm_bwd(const void* dy, int dy_div, float dy_scale, const float* x, int in_mul, const int* in_off,
                        const float* mean, const float* rstd, const float* gamma, float* dx, int accumulate, float* dgamma,
                        float* dbeta, void* dx_cast, int rows, int cols, int dy_dtype, void* stream);
/* missm_layernorm_bwd (no row gather, dy_div 1) that also accumulates GROUP SUMS of the updated dx rows:
 * gsum[(row / gs_div) % gs_mod][:] += dx[row][:] (atomics; caller zeroes), rows = frames x gs_div.  Replaces the separate column-sum
 * pass of the temporal-embedding gradient: `hidden_states + self.temporal_embedding[:, :t, :]` (image/modeling_image.py:114-116) puts
 * d temporal_embedding[t] = sum over (b, n) of the residual gradient behind temporal_layer_norm1 (gs_div = S tokens, gs_mod = T).
 * gs_div = 0 selects group = row % gs_mod: d position_embedding[s] = sum over frames of the gradient behind pre_layrnorm (gs_mod = S;
 * `embeddings = embeddings + self.position_embedding(self.position_ids)`, third-party CLIPVisionEmbeddings / video/modeling_video.py:19-51). */
int missm_layernorm_bwd_groupsum(const void* dy, float dy_scale, const float* x, const float* mean, const float* rstd,
                                 const float* gamma, float* dx, int accumulate, float* dgamma, float* dbeta, void* dx_cast,
                                 float* gsum, int gs_div, int gs_mod, int rows, int cols, int dy_dtype, void* stream);
/* The same two kernels for `ngroups` <= 8 shape-identical towers that run in lock-step (languagebind/__init__.py:75-85 encodes the
 * modalities one after the other; missm_gemm_grouped): one launch, every operand an array of per-tower pointers.  Residual-stream
 * form only: no row gather / additive vector; the backward accumulates into dx (+=) and writes the dy-dtype copy dx_cast; dgamma /
 * dbeta (arrays or entries) may be null (frozen LayerNorms of a LoRA tower). */
int missm_layernorm_fwd_grouped(int ngroups, const float* const* x, const float* const* gamma, const float* const* beta, void* const* y,
                                float* const* mean, float* const* rstd, int rows, int cols, float eps, int out_dtype, void* stream);
int missm_layernorm_bwd_grouped(int ngroups, const void* const* dy, const float* const* x, const float* const* mean, const float* const* rstd,
                                const float* const* gamma, float* const* dx, float* const* dgamma, float* const* dbeta, void* const* dx_cast,
                                int rows, int cols, int dy_dtype, void* stream);
/* out[b] = mean_t in[b*T + t]   (pooled_output.reshape(B, T, -1).mean(1), image/modeling_image.py:662). */
int missm_mean_rows(const float* in, float* out, int B, int T, int cols, void* stream);

/* Fused softmax(Q K^T * scale + masks) V for all heads of all sequences, reading the fused [rows, 3*H*hd] QKV matrix.
 * Token j of sequence q is row (q / seq_div) * seq_outer + (q % seq_div) * seq_inner + j * tok_stride.
 * causal != 0 masks keys j > i; key_mask[nseq, L] (0 = masked) is the padding mask.  lse[nseq, H, L] is saved.
 * Replaces CLIPAttention's bmm/softmax/bmm (third-party; called at image/modeling_image.py:121-126,140-145). */
int missm_attention_fwd(const void* qkv, void* out, float* lse, int nseq, int L, int H, int head_dim, int ld, int ldo,
                        int seq_div, int seq_outer, int seq_inner, int tok_stride, int causal, const int* key_mask, float scale,
                        int dtype, void* stream);
/* Backward: dqkv[rows, 3*H*hd] from the saved forward tensors.  `out` is the forward output (ld = ldo, like dout): the
 * softmax-gradient row term D = rowsum(dO . O) is taken from it, so no [L, L] tile is held while a row sum is pending. */
int missm_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int nseq, int L, int H,
                        int head_dim, int ld, int ldo, int seq_div, int seq_outer, int seq_inner, int tok_stride, int causal,
                        const int* key_mask, float scale, int dtype, void* stream);

/* Patch unfold: pixels fp32 (frame n = (b, t): base b*stride_b + t*stride_t, channel stride_c, row-major HxW) ->
 * A[n*P + p, c*ps*ps + ky*ps + kx] of type `dtype` (the conv weight's .view(d, -1) order).
 * Rows hold Kp = C*ps*ps rounded up to a multiple of 8 elements (zero columns behind K: a 14-pixel patch has K = 588). */
int missm_unfold_patches(const float* pixels, void* out, int B, int T, int C, int H, int W, int ps, long stride_b,
                         long stride_t, long stride_c, int dtype, void* stream);
/* out[r] = in[rdiv > 0 ? r + r / rdiv + roff : r] converted fp32 -> `dtype` (residual-gradient stream -> GEMM operand;
 * rdiv = S-1, roff = 1 drops the CLS rows for the patch-embedding weight gradient). */
int missm_cast_rows(const float* in, void* out, long R, int C, int rdiv, int roff, int dtype, void* stream);
/* x[n, s, :] = (s == 0 ? cls : patches[n*(S-1) + s-1]) + pos[s]   (CLIPVisionEmbeddings, video/modeling_video.py:48-50). */
int missm_embed_assemble(const void* patches, const float* cls, const float* pos, float* x, int N, int S, int d, int dtype,
                         void* stream);
/* h[b, s, :] = tok[ids[b, s]] + pos[s]  (CLIPTextEmbeddings, third-party; image/modeling_image.py:494).  An id outside
 * [0, vocab) never reaches memory: its row is filled with NaN (the reference raises IndexError / a device-side assert). */
int missm_token_embed_fwd(const long* ids, const float* tok, const float* pos, float* h, int B, int S, int d, int vocab,
                          void* stream);
/* dtok[ids[b,s]] += dh[b,s] (atomic), dpos[s] += sum_b dh[b,s]; rows whose id is outside [0, vocab) are skipped. */
int missm_token_embed_bwd(const long* ids, const float* dh, float* dtok, float* dpos, int B, int S, int d, int vocab,
                          void* stream);
/* eot[b] = argmax_s ids[b, s] (first occurrence), image/modeling_image.py:519-522. */
int missm_argmax_rows(const long* ids, int* out, int B, int S, void* stream);

/* y[b, o] = sum_i x[b, i] w[o, i] (+ bias[o]) -> optional ReLU ; all fp32 (projection / fusion tail).  y has row stride
 * ldy >= O, so a head can write its slice of a concatenated feature row (torch.cat, src/model/baseline.py:84,176).
 * row_code/code: rows with row_code[b] == code produce 0 (modality missing, src/model/baseline.py:57) - or, when x_sub[I]
 * is given, are computed from that row instead of x[b] (zero / mean / median imputation, src/model/baseline.py:80-82);
 * select != 0 inverts the role of the code: ONLY rows with row_code[b] == code are computed and written, the others are left
 * untouched (a dedicated network overwriting the rows of its missing-modality case, src/model/baseline.py:349-351);
 * alpha scales (x W^T + bias) before the ReLU (the mean over source modalities of the cross-modal regressors, :139-141);
 * accumulate: y += (sum over modalities). */
int missm_small_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int I, int O, int ldy, int relu,
                           const long* row_code, long code, const float* x_sub, int select, float alpha, int accumulate,
                           void* stream);
/* dx[b,i] (= or +=), dw[o,i] (=), dbias[o] (=) for the layer above; dy (row stride lddy) is masked by row_code/code and,
 * when relu_y is given, by relu_y > 0.  With x_sub the masked rows still feed dw / dbias (their input was x_sub); only
 * their dx is zero.  accumulate_dw: dw / dbias += (a layer applied to several modalities). */
int missm_small_linear_bwd(const float* dy, int lddy, const float* x, const float* w, const float* relu_y, float* dx, float* dw,
                           float* dbias, int B, int I, int O, const long* row_code, long code, const float* x_sub,
                           int select, float alpha, int accumulate_dx, int accumulate_dw, void* stream);
/* dst[b, 0:W] += src[b, 0:W] with row strides lddst / ldsrc (gradient slices of a concatenated feature row meeting again). */
int missm_add_block(float* dst, int lddst, const float* src, int ldsrc, int B, int W, void* stream);
/* dst[b, 0:W] = row_code[b] == code ? 0 : src[b, 0:W]: one modality's block of the concatenated feature row with its missing
 * rows zeroed (the distillation heads, src/model/baseline.py:370-376); also its own backward.  keep_matching != 0 inverts the
 * selection (only the rows that carry the code are copied: the unified graph head's "fill the missing modality", :311). */
int missm_masked_copy_block(float* dst, int lddst, const float* src, int ldsrc, int B, int W, const long* row_code, long code,
                            int keep_matching, void* stream);
/* Channel-attention gate of the intra-modality attention head (src/model/baseline.py:198-201):
 * y[b,f] (= or +=) row b missing ? 0 : d[b,f] * sigmoid(pre[b,f]); d has row stride ldd.  Backward: dd (stride lddd, = or +=)
 * and dpre, both zero for missing rows. */
int missm_gate_fwd(const float* d, int ldd, const float* pre, float* y, int B, int F, const long* row_code, long code,
                   int accumulate, void* stream);
int missm_gate_bwd(const float* dy, const float* d, int ldd, const float* pre, float* dd, int lddd, float* dpre, int B, int F,
                   const long* row_code, long code, int accumulate_dd, void* stream);
/* y = x / ||x||_2 * scale   (languagebind/__init__.py:80-83) and its backward. */
int missm_l2norm_scale_fwd(const float* x, float* y, int B, int D, float scale, void* stream);
int missm_l2norm_scale_bwd(const float* dy, const float* x, float* dx, int B, int D, float scale, void* stream);
/* loss = mean_b CE(logits[b], labels[b]) ; dlogits = (softmax - onehot) / B   (nn.CrossEntropyLoss, train_ddp.py:88,250). */
int missm_cross_entropy(const float* logits, const long* labels, float* loss, float* dlogits, int B, int C, void* stream);
/* KL_loss (train_ddp.py:70-79): kl_div(log_softmax(student / T), softmax(teacher / T), 'batchmean') over the rows whose row_mask byte
 * is non-zero (all rows when row_mask is null - the reference gathers them by boolean indexing, train_ddp.py:238-240);
 * dstudent (optional) = d loss / d student (zero on unselected rows); the teacher side is detached, as in the reference. */
int missm_kl_loss(const float* student, const float* teacher, const unsigned char* row_mask, float* loss, float* dstudent, int B, int C,
                  float temperature, void* stream);
/* nn.MSELoss() of the MTD student mode (train_ddp.py:84): mean((a - b)^2); da (optional) = 2 (a - b) / n. */
int missm_mse_loss(const float* a, const float* b, float* loss, float* da, long n, void* stream);
/* teacher EMA of the MTD student mode (train_ddp.py:256-259): teacher = decay * teacher + (1 - decay) * student. */
int missm_ema_update(float* teacher, const float* student, long n, float decay, void* stream);
/* GPU-side preprocessing of ONE decoded image (reference image/processing_image.py:18-28, thermal/processing_thermal.py:18-28:
 * ToTensor, Resize(S, bicubic) of the shorter edge, CenterCrop(S), Normalize; depth/processing_depth.py:21-55: DepthNorm first).
 * src: uint8 or float32 (src_u8), [C,H,W] (chw) or [H,W,C], C = 1 (replicated to 3 channels) or 3, DEVICE memory (pinned-staged
 * by the caller).  v = clip(v * pre_scale, pre_min, pre_max) / pre_div before resampling (pre_max <= 0: no upper clip):
 * images 1/255, -inf.., 1 ; depth 1/1000, 0.01, max_depth, max_depth.  dst: fp32 [3, S, S].  mean3 / std3: HOST arrays. */
int missm_preprocess_image(const void* src, int src_u8, int chw, int H, int W, int C, float* dst, int S, float pre_scale, float pre_min,
                           float pre_max, float pre_div, const float* mean3, const float* std3, void* stream);
/* SuperGAT attention over per-sample modality graphs (fusion_gcn of the graph heads, src/model/baseline.py:11-24,240-331:
 * torch_geometric SuperGATConv, 'MX' attention, self loops, negative_slope 0.2).  xp = lin(x) [B, M, H, C], M <= 8 nodes; an edge
 * j -> i exists iff i == j or node_ok[b,i] && node_ok[b,j].  out [B, M, H, C] (before the head concat / mean and the bias), alpha
 * [B, H, M, M] saved for the backward.  PARITY UNPINNED: torch_geometric is absent and unpinned upstream; restated from the paper. */
int missm_sgat_fwd(const float* xp, const float* att_l, const float* att_r, const unsigned char* node_ok, const float* bias, float* out,
                   float* out_gelu, float* alpha, int B, int M, int H, int C, void* stream);
/* dx = dy * gelu'(pre) (exact erf form): the nn.GELU between the two SuperGAT layers (src/model/baseline.py:16,21); missm_sgat_fwd
 * writes gelu(out) to out_gelu when that pointer is given.  bias [H*C] (optional) is added to out. */
int missm_gelu_bwd(const float* dy, const float* pre, float* dx, long n, void* stream);
/* backward: dxp [B, M, H, C]; per-sample partial sums of the attention-vector gradients [B, H, C] (summed over the batch afterwards). */
int missm_sgat_bwd(const float* xp, const float* att_l, const float* att_r, const unsigned char* node_ok, const float* alpha, const float* dout,
                   float* dxp, float* datt_l_part, float* datt_r_part, int B, int M, int H, int C, void* stream);
/* inverted dropout with a counter-based generator: y = x * mask / (1-p); mask saved as bytes (src/model/baseline.py:34). */
int missm_dropout_fwd(const float* x, float* y, unsigned char* mask, long n, float p, unsigned long long seed, void* stream);
int missm_dropout_bwd(const float* dy, const unsigned char* mask, float* dx, long n, float p, void* stream);

/* Fused Adam over a flat fp32 parameter buffer (torch.optim.Adam semantics, train_ddp.py:205,254):
 * reads p, g, m, v; writes p, m, v. grad_scale multiplies g first (1/world_size for an all-reduce SUM). */
int missm_adam_step(float* p, const float* g, float* m, float* v, long n, int step, float lr, float beta1, float beta2, float eps,
                    float weight_decay, float grad_scale, void* stream);
/* The same step for the weight matrices listed in a missm_cast_weights_batched tile table, fused with the refresh of their
 * `dtype` copies (W and W^T): g, m, v are addressed at g_off / m_off / v_off FLOATS from each tile's master pointer (flat
 * buffers parallel to the master buffer).  Saves re-reading the updated weights for the separate refresh. */
int missm_adam_cast_batched(const void* tiles, int ntiles, long g_off, long m_off, long v_off, int step, float lr, float beta1,
                            float beta2, float eps, float weight_decay, float grad_scale, int dtype, void* stream);

/* LoRA adapters of the vision encoders (languagebind/image/modeling_image.py:775-793: peft's get_peft_model over vision_model.encoder,
 * targets q/k/v/out_proj - or temporal_attn.* and temporal_mlp.fc1/fc2 with add_time_attn; lora_dropout 0, configuration_image.py:200-202).
 * merge: W[n_out, k_in] (fp32, leading dimension ldw) += scale * B[n_out, r] A[r, k_in] - the weight the unmerged peft forward
 * x W^T + scale (x A^T) B^T is the linear of (scale = lora_alpha / r); the compute-dtype weight copies are cast from it.
 * grad : dB[n_out, r] += scale * G A^T, dA[r, k_in] += scale * B^T G from the full weight gradient G = dY^T X [n_out, k_in] that the
 * weight-gradient GEMM produces; base weights stay frozen (adapter-only training, as in the reference). */
int missm_lora_merge(float* W, int ldw, const float* A, const float* B, int n_out, int k_in, int r, float scale, void* stream);
int missm_lora_grad(const float* G, int ldg, const float* A, const float* B, float* dA, float* dB, int n_out, int k_in, int r, float scale,
                    void* stream);

/* Audio front end (languagebind/audio/processing_audio.py:31-111).  All buffers are device fp32.
 * buffer_mean   : out[0] = mean(x[0..n))  (`audio_data -= audio_data.mean()`, :96; fixed summation order)
 * fbank_frames  : number of frames of an n-sample clip (snip_edges): 1 + (n - window) / shift, 0 if shorter than one window (host only)
 * kaldi_fbank   : torchaudio.compliance.kaldi.fbank(wave - *global_mean, htk_compat=True, use_energy=False, window_type="hanning",
 *                 dither=0, num_mel_bins, frame_length, frame_shift, sample_frequency; defaults: remove_dc_offset, preemphasis 0.97,
 *                 round_to_power_of_two, power spectrum, log mel energies floored at float eps, low_freq 20, high_freq 0 = Nyquist)
 *                 -> out [frames, num_mel_bins]  (:97-107)
 * mel_assemble  : the three chunks starting at start0/1/2 (clip longer than target_length) or the clip tiled up to target_length
 *                 (shorter / equal), transposed to out [3, num_mel_bins, target_length] and normalised (x - mean) / (2 std)  (:54-93)
 * sinc_resample : torchaudio.functional.resample's strided convolution (:44-46): out[q * new + p] = sum_j kernels[p, j] *
 *                 padded[q * orig + j]; `kernels` [new_freq, 2 width + orig_freq] is the windowed-sinc table (host-built, see
 *                 missm_benchmark_amd/processing.py), orig_freq / new_freq already divided by their gcd. */
int missm_buffer_mean(const float* x, long n, float* out, void* stream);
int missm_fbank_frames(long n, float sample_rate, float frame_length_ms, float frame_shift_ms);
int missm_kaldi_fbank(const float* wave, long n, const float* global_mean, float* out, int num_mel_bins, float sample_rate,
                      float frame_length_ms, float frame_shift_ms, float low_freq, float high_freq, float preemphasis, void* stream);
int missm_mel_assemble(const float* mel, int frames, int num_mel_bins, float* out, int target_length, int start0, int start1, int start2,
                       float mean, float std, void* stream);
int missm_sinc_resample(const float* wave, long n, const float* kernels, int kernel_len, int orig_freq, int new_freq, int width, float* out,
                        long n_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MISSM_HIP_H */
