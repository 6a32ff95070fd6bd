/*
 * dfdclip.h — C ABI of libdfdclip_hip.so: hand-written gfx950 (MI355X) kernels for DFD-CLIP's
 * hot path (per-clip CLIP-ViT K/V extraction + temporal cross-attention decoder).
 *
 * The reference is pure Python on PyTorch ATen ops and has no FFI of its own;
 * each entry point below replaces the ATen op sequence of the cited reference lines
 * (paths under the reference tree).  All pointers are DEVICE pointers unless noted.  Every
 * function enqueues on `stream` (a hipStream_t passed as void*; NULL = default stream) and
 * returns without synchronising: 0 on success, DFD_ERR_* (<0) on failure with a message
 * available from dfd_last_error().  No entry point allocates device memory; workspaces are
 * passed in.  Thread-safe for distinct streams (the error string is thread-local).
 */
#ifndef DFDCLIP_H
#define DFDCLIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFD_ABI_VERSION 15

enum { DFD_F32 = 0, DFD_BF16 = 1, DFD_FP8 = 2 /* OCP e4m3 ("e4m3fn"), one byte per element */ };

enum {
  DFD_OK = 0,
  DFD_ERR_INVALID_ARG = -1, /* null pointer, unsupported shape/dtype, misaligned leading dimension */
  DFD_ERR_LAUNCH = -2,      /* hipLaunchKernel / hipGetLastError reported a failure */
  DFD_ERR_NO_DEVICE = -3    /* no gfx950 device visible */
};

/* Epilogues of dfd_gemm (applied to acc = A · Wᵀ, fp32 accumulator). */
enum {
  DFD_EPI_BIAS = 0,        /* C = acc + bias                      — nn.Linear                              */
  DFD_EPI_BIAS_QUICKGELU,  /* C = g(acc + bias), g(u)=u·σ(1.702u) — c_fc + QuickGELU  (clip/model.py:166-168, :208-212) */
  DFD_EPI_BIAS_RESIDUAL,   /* C(f32) += acc + bias                — out_proj / c_proj + residual (clip/model.py:222-223) */
  DFD_EPI_PATCH_EMBED,     /* conv1-as-GEMM: scatter to token rows, + positional embedding, CLS row
                              (clip/model.py:277-291)                                              */
  DFD_EPI_QKV_EXPORT,      /* C = acc + bias AND export of the K / V column blocks, CLS row dropped,
                              temporal positional embedding added (clip/model.py:186-199,
                              models.py:505-509, :326-334)                                        */
  DFD_EPI_RESIDUAL_POS     /* C(c_dtype) += acc + pos[(row / (tokens-1)) % T]: the adapter's second Linear,
                              its residual and the decoder's positional add in one pass
                              (models.py:935-937, :326-329); extra.pos may be NULL, extra.tokens-1 =
                              rows (patches) per frame                                            */
};

/* Train-mode dropout (models.py:163, :294, :304; adapter :804-912).  Masks are counter-based (Philox4x32-10 on
 * (seed, step, site, element index)), generated inside the kernels and regenerated in the backward:
 * rng_state is a DEVICE pointer to {seed, step} (uint64[2]; read at execution time, so a captured graph draws
 * new masks when the host bumps step), site distinguishes the dropout layers of one step, p = drop probability
 * (0 = identity, bit for bit).  Element e of a site is kept iff draw16(e) >= round(p*65536); kept values are
 * scaled by 65536/(65536 - round(p*65536)).  Exact definition: dfd-clip_amd/csrc/dropout.hpp. */
typedef struct dfd_dropout_t {
  const uint64_t* rng_state;
  uint32_t site;
  float p;
} dfd_dropout_t;

typedef struct dfd_gemm_extra {
  /* PATCH_EMBED: encoder positional_embedding [tokens, N] f32.
     QKV_EXPORT : decoder temporal positional embedding viewed [frames_per_clip, N/3] f32, or NULL. */
  const float* pos;
  const float* cls;        /* PATCH_EMBED: class_embedding [N] f32 */
  void* k_export;          /* QKV_EXPORT: [n_frames*(tokens-1), N/3] in c_dtype, or NULL (no export) */
  void* v_export;
  int32_t tokens;          /* tokens per frame incl. CLS (197 for ViT-B/16) */
  int32_t frames_per_clip; /* T */
  const void* residual;    /* RESIDUAL_POS: residual source in c_dtype with C's leading dimension; NULL = C (in place) */
  int32_t qkv_first;       /* QKV_EXPORT: 0 = W / C columns are [q | k | v] (N = 3D); 1 = [k | v] only (N = 2D): the last
                              tapped layer needs no queries (nothing after its K/V export is read) */
  const uint64_t* drop_rng; /* RESIDUAL_POS: dropout on acc before the residual add (the adapter's last nn.Dropout,   */
  uint32_t drop_site;       /* models.py:807 etc.): C = residual + dropout(acc) + pos; element index = row*N + col;  */
  float drop_p;             /* drop_rng NULL or drop_p 0 = none                                                      */
  uint32_t flags;           /* DFD_GEMM_* bits */
} dfd_gemm_extra;

/* dfd_gemm_extra.flags */
enum {
  DFD_GEMM_STREAM_OUT = 1,  /* C (and the K/V export) is written once and not re-read soon by this GPU's caches' standards:
                               store it non-temporally so that it does not evict the operand panels from L2 (tuned bf16
                               kernels; ignored elsewhere) */
  DFD_GEMM_SPARE_IF_FREE = 2, /* the spare compute units (bits 8..15) are left free only where that costs no extra round of
                               tiles for this shape (ViT-B/16's c_proj: 5 rounds on 224 CUs as on 256; ViT-L/14's: 5 instead
                               of 4, so there the kernel takes every CU).  Without the bit the request is binding (the
                               window a collective needs beside the encoder) */
  DFD_GEMM_TILE_BLOCKS_SHIFT = 16, /* bits 16..19: 0 = the persistent kernel chooses its tile height; 7 / 8 = force 224- /
                               256-row tiles (tests, tuning) */
  DFD_GEMM_SPARE_CUS_SHIFT = 8 /* bits 8..15: compute units the persistent kernel leaves free (its grid is one workgroup
                               per remaining CU), so that small latency-bound kernels of ANOTHER stream — the decoder's
                               backward and the optimizer while the next batch's encoder pass runs — find a CU at once */
};

const char* dfd_last_error(void);          /* host pointer, valid until the thread's next failing call */
int dfd_abi_version(void);
int dfd_device_check(void);                /* DFD_OK iff device 0.. current is gfx950 */

/* LayerNorm over the last dim, fp32 statistics, eps inside the sqrt
 * (clip/model.py:157-163, models.py:58-68).  x f32 [rows, cols] (row stride ldx);
 * y in y_dtype (row stride ldy); y may alias x when y_dtype == DFD_F32.
 * cols % 4 == 0, cols <= 4096. */
int dfd_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy,
                  int y_dtype, int64_t rows, int cols, float eps, float y_inv_scale, void* stream);

/* Two LayerNorms back to back in one pass over the rows: x <- LayerNorm_a(x) (f32, in place), y = LayerNorm_b(x) — the
 * encoder's ln_pre followed by the first block's ln_1 (clip/model.py:292, :221).  The same arithmetic, in the same order, as
 * two dfd_layernorm calls (bit-identical); x is read once.  cols <= 2048; y must not alias x. */
int dfd_layernorm2(float* x, int64_t ldx, const float* gamma_a, const float* beta_a, const float* gamma_b, const float* beta_b,
                   void* y, int64_t ldy, int y_dtype, int64_t rows, int cols, float eps, float y_inv_scale, void* stream);
/* y_dtype DFD_FP8 (both LayerNorm entry points): y = e4m3(LayerNorm(..) * y_inv_scale), saturated at +-448 — the A
 * operand of dfd_gemm_fp8; y_inv_scale is ignored for the other output types. */

/* Residual add fused with the LayerNorm that follows it: v = (x[rows, cols] + delta) [+ delta2], deltas in
 * delta_dtype with one leading dimension ldd; y = LayerNorm(v) as dfd_layernorm; x (f32) is overwritten with v
 * when store_x != 0.  The bf16 encoder path has out_proj / c_proj write their output (bias included) as bf16
 * deltas and folds `x = x + attn(...)`, `x = x + mlp(...)` (clip/model.py:222-223) into the next ln_2 / ln_1 —
 * torch autocast's dataflow.  ln_2 reads x + delta_attn without storing it; the next ln_1 adds both deltas and
 * stores x once per block.  delta2 may be NULL.  y must not alias x or a delta.  cols % 4 == 0, cols <= 2048. */
int dfd_add_layernorm(float* x, int64_t ldx, const void* delta, const void* delta2, int64_t ldd, int delta_dtype,
                      int store_x, const float* gamma, const float* beta, void* y, int64_t ldy, int y_dtype, int64_t rows,
                      int cols, float eps, float y_inv_scale, void* stream);

/* Frames [n_frames, 3, res, res] (f32) -> patch rows [n_frames*P, kpad] in out_dtype, column
 * k = c*patch*patch + i*patch + j (the flatten order of conv1.weight [D,3,patch,patch]), columns
 * >= 3*patch*patch zero-filled.  The patch conv (clip/model.py:264, :277) then is a plain GEMM. */
int dfd_patchify(const float* frames, void* patches, int out_dtype, int n_frames, int res, int patch, int kpad,
                 void* stream);

/* Device-side `Detector._transform` for uint8 frames (src/models.py:756-768: Resize(res, BICUBIC)
 * -> CenterCrop(res) -> ConvertImageDtype(float32) -> Normalize(mean, std)), fused with the patch
 * extraction of dfd_patchify.  frames: [n_frames, 3, in_h, in_w] uint8 (device).  The shorter
 * side is resized to `res` (longer side int(res*long/short), as torchvision), the resized image is
 * rounded back to the uint8 grid and clamped (skipped when no resize is needed), centre-cropped
 * (offset round-half-even((size-res)/2)), divided by 255 and normalised per channel.
 * antialias 0: ATen upsample_bicubic2d (A=-0.75); 1: _upsample_bicubic2d_aa (A=-0.5).
 * mean3 / std3: HOST pointers to 3 floats.  layout 0: out = frames [n,3,res,res]; layout 1: out =
 * patch rows [n*P, kpad] exactly as dfd_patchify writes them.  out_dtype DFD_F32 or DFD_BF16. */
int dfd_preprocess_u8(const uint8_t* frames, int n_frames, int in_h, int in_w, int res, int patch, int antialias,
                      const float* mean3, const float* std3, void* out, int out_dtype, int layout, int kpad,
                      void* stream);
/* The geometry dfd_preprocess_u8 uses: resized size and crop origin (host-only helper). */
int dfd_preprocess_geometry(int in_h, int in_w, int res, int* rs_h, int* rs_w, int* top, int* left);

/* C = epilogue(A[M,K] · W[N,K]ᵀ): A and W row-major in ab_dtype with leading dimensions lda / ldw
 * (elements), fp32 accumulation on the matrix cores (bf16: v_mfma_f32_32x32x16_bf16 /
 * 16x16x32; f32: v_mfma_f32_32x32x2_f32, exact fp32).  K % 32 == 0; 16-byte aligned rows.
 * C row-major [*, ldc] in c_dtype (see the epilogue list for what "row" means). */
int dfd_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, int ab_dtype, void* C, int64_t ldc,
             int c_dtype, const float* bias, int epilogue, const dfd_gemm_extra* extra, int64_t M, int N, int K,
             void* stream);
/* The same product on OCP e4m3 ("fp8") operands, on the block-scaled matrix cores (v_mfma_scale_f32_16x16x128_f8f6f4,
 * unit block scales; twice the bf16 rate): C = epilogue((A[M,K] · W[N,K]ᵀ) · col_scale[n] + bias[n]).  A, W: one byte per
 * element, row-major (lda, ldw in elements); col_scale[n] = (scale the activations were divided by) x (scale row n of W was
 * divided by); C bf16, or — plain and QuickGELU epilogues — e4m3 of result * out_inv_scale, saturated at +-448 (the c_fc
 * output feeding the next fp8 GEMM).  Epilogues: BIAS, BIAS_QUICKGELU, QKV_EXPORT (exports bf16).  Served shapes:
 * M >= 1024, N % 256 == 0, K % 128 == 0, K >= 256, 16-byte aligned rows; anything else is DFD_ERR_INVALID_ARG (there is
 * no second fp8 kernel).  BASELINE configs[4]; quantisation policy: dfd-clip_amd/encoder.py. */
int dfd_gemm_fp8(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int c_dtype,
                 const float* col_scale, const float* bias, float out_inv_scale, int epilogue,
                 const dfd_gemm_extra* extra, int64_t M, int N, int K, void* stream);

/* Which kernel served this thread's last successful dfd_gemm: 257 = the persistent 256x256 kernel with the ping-pong
 * K loop (M >= 1024, N % 256 == 0, K % 128 == 0, K >= 384), 256 = the other tuned 256x256 bf16 kernels (K % 64 == 0,
 * K >= 128), 128 = the general 128x128 kernel, 0 = none yet.  For tests and profilers. */
int dfd_gemm_last_path(void);

/* Tests and A/B measurements: 0 (default) = every kernel eligible; 1 = skip the ping-pong kernel, so that the round-2
 * persistent kernel serves the shapes both can (their results are bit-identical: tests/test_hip_kernels.py); 3 = the
 * ping-pong kernel hands out the tiles after a workgroup's first dynamically (per-XCD counters) instead of dealing them
 * statically — for a chip shared with kernels of other streams that hold compute units unpredictably; measured 0.4 %
 * slower than the static order when nothing of the kind happens, so not the default.  Per thread; returns the previous
 * value. */
int dfd_gemm_set_variant(int variant);

/* C[Ma, Nb] (f32) = Aᵀ · B for tall row-major operands A [R, Ma], B [R, Nb] in `dtype` — the weight
 * gradient of a Linear applied to R rows (adapter training: R = B·T·patches).  Internally: zero-padded
 * transposes, a split-K pass of the general MFMA kernel into f32 slabs, fixed-order slab reduction
 * (deterministic).  workspace >= dfd_gemm_at_b_workspace(...) bytes, 256-byte aligned. */
size_t dfd_gemm_at_b_workspace(int64_t R, int Ma, int Nb, int dtype);
int dfd_gemm_at_b(const void* A, int64_t lda, const void* B, int64_t ldb, int dtype, float* C, int64_t R, int Ma, int Nb,
                  void* workspace, void* stream);

/* Encoder self-attention over the packed projection qkv [n_frames*tokens, 3*heads*head_dim]
 * (column blocks q | k | v; head h = columns h*head_dim ..): out = softmax(q kᵀ · scale) v,
 * out [n_frames*tokens, heads*head_dim] (clip/model.py:188-195).  head_dim == 64. */
int dfd_attention_fwd(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, int dtype, int n_frames,
                      int tokens, int heads, int head_dim, float scale, void* stream);

/* y[B,N] = epilogue(x[B,K] · W[N,K]ᵀ + bias), all f32 — the decoder's one-token-per-clip linears
 * (models.py:130-131, :160-165).  epilogue: DFD_EPI_BIAS, _BIAS_QUICKGELU or _BIAS_RESIDUAL
 * (y += ...).  B <= 64, K % 4 == 0. */
int dfd_linear_rows(const float* x, int64_t ldx, const float* W, const float* bias, float* y, int64_t ldy,
                    int epilogue, int B, int N, int K, void* stream);

/* Same product from the TRANSPOSED weight Wt [K, N] (row-major): y[B,N] = epilogue(x[B,K] · Wt + bias).
 * Rows of Wt stream fully coalesced and the activations are wave-uniform scalars, so this is the fast
 * form for both the forward (Wt = weightᵀ, kept by the host and refreshed after each optimizer step)
 * and the data gradient (dx = dy · W: pass the weight itself as "Wt").  Deterministic (fixed-order
 * slab reduction).  N % 4 == 0; workspace >= dfd_linear_rows_t_workspace(B, N, K) bytes.
 * DFD_EPI_BIAS_RESIDUAL adds `residual` [B, N] (row stride ldr, 16-byte aligned rows); residual == NULL adds the
 * previous contents of y (in place).  The other epilogues ignore it. */
size_t dfd_linear_rows_t_workspace(int B, int N, int K);
int dfd_linear_rows_t(const float* x, int64_t ldx, const float* Wt, const float* bias, const float* residual, int64_t ldr,
                      float* y, int64_t ldy, int epilogue, int B, int N, int K, void* workspace, void* stream);

/* Where the decoder's keys / values live.  NULL = the dense export: [B, T*patches, heads*d] rows of heads*d elements.
 * Otherwise element (frame f = clip*T + t, patch p, channel c) is read at base + f*frame_stride + p*row_stride + c
 * (strides in ELEMENTS; rows 16-byte aligned), and `pos` [T, heads*d] f32 (may be NULL) is added to every key AND
 * value row of frame t as it is read — models.py:326-329's `kv + temporal positional embedding` done on the fly.
 * This is the layout of the encoder's q|k|v activation [frames, tokens, 3*heads*d] read in place with the CLS row
 * skipped (k = qkv + 3*D + D, v = qkv + 3*D + 2*D, row_stride = 3*D, frame_stride = tokens*3*D): the encoder then
 * writes no export at all. */
typedef struct {
  int64_t row_stride, frame_stride;
  const float* pos;
} dfd_kv_layout_t;

/* Decoder cross-attention of ONE query per clip over S = T*P exported keys/values, two branches
 * averaged (models.py:136-146): softmax(q_s·k/√d) and tanh(q_c·k/√d)·2σ(−‖q_c−k‖₁/√d); keys of padded
 * frames (frame_mask[b,t] == 0) get weight 0 in both (models.py:104, :124).
 *   q          f32 [B, heads, 2*d]  (in_proj output: per head softmax query then CoDA query)
 *   k, v       kv_dtype [B, S, heads*d]
 *   frame_mask u8 [B, T], S == T * patches
 *   ext_weights f32 [B, heads, S] or NULL: softmax-branch weights computed elsewhere (attn_mode, below)
 *   mix        f32 [B, heads*d]
 *   mix_softmax f32 [B, heads*d] or NULL: the softmax branch alone (Σ a_j v_j), kept for backward
 *   stats      f32 [B, heads, 2] = (row max, sum of exp) of the softmax branch, kept for backward
 *   workspace  f32, at least dfd_decoder_attn_workspace(B, heads, d, splits) bytes. */
size_t dfd_decoder_attn_workspace(int B, int heads, int d, int splits);
int dfd_decoder_attn_fwd(const float* q, const void* k, const void* v, int kv_dtype, const dfd_kv_layout_t* layout,
                         const uint8_t* frame_mask, const float* ext_weights, float* mix, float* mix_softmax, float* stats,
                         void* workspace, int splits, int B, int T, int patches, int heads, int d, void* stream);

/* op_mode.attn_mode (models.py:107-115): the softmax branch becomes a sum of grouped softmaxes of the
 * scores viewed [T, patches] — bit 0 of `modes` = "frame" (over the patches of each frame), bit 1 =
 * "temporal" (over the frames at each patch position).  Writes scores [B, heads, S] = q_s·k/√d (-inf on
 * padded frames) and weights [B, heads, S] = Σ_modes softmax_mode(scores); pass `weights` to
 * dfd_decoder_attn_fwd as ext_weights (mix_softmax / stats are then not meaningful).  A group with
 * every key padded yields NaN, as in the reference. */
int dfd_decoder_attn_modes_fwd(const float* q, const void* k, int kv_dtype, const dfd_kv_layout_t* layout,
                               const uint8_t* frame_mask, int modes, float* scores, float* weights, int B, int T, int patches,
                               int heads, int d, void* stream);

/* Head: video_feature = LayerNorm(x) (ln_post), z = video_feature @ proj [D, out_dim],
 * logits = 5 z / (‖z‖₂ + 1e-10)  (models.py:342-343, :359, :551-553).  All f32. */
int dfd_head_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, const float* proj,
                 float* video_feature, float* raw_logits, float* logits, int B, int D, int out_dim, float eps,
                 const dfd_dropout_t* drop_post, void* stream);
/* drop_post (may be NULL): the decoder's drop_post between ln_post and the projection (models.py:342):
 * video_feature holds the DROPPED features, element index = clip*D + channel. */

/* out = dropout(in) over n elements (element index = position), any mix of f32 / bf16; in place allowed.
 * Also the backward of a dropped tensor (dx = mask·scale·dy with the same descriptor). */
int dfd_dropout(const void* in, int in_dtype, void* out, int out_dtype, int64_t n, const dfd_dropout_t* drop,
                void* stream);

/* CompInvAdapter middle stage (models.py:823-875): y = GELU_erf(LayerNorm(a)) on a [frames, patches, x]
 * tensor in `dtype`.  joint == 1 ("nln"): statistics over the whole (patches, x) slab of a frame, affine
 * weight/bias [patches, x]; joint == 0 ("ln"/"z0"): statistics per row of x, affine [x]; joint == 2
 * ("768-x-768" / "legacy-768-x-768", models.py:795-821): y = LayerNorm(GELU(a)) per row of x — the same two
 * stages in the other order.  `a` is in a_dtype = dtype, or f32 with a bf16 y (the GELU-first structs keep
 * the projection output in f32: LayerNorm after the non-linearity amplifies its rounding).  y may alias a
 * when the dtypes agree.  The backward takes the same mode and a_dtype. */
int dfd_adapter_norm_gelu(const void* a, int a_dtype, void* y, int dtype, const float* weight, const float* bias, int frames,
                          int patches, int x, int joint, float eps, void* stream);

/* Backward of dfd_adapter_norm_gelu: a = the forward's input, dy = dL/dy; writes da = dL/da (same dtype;
 * must NOT alias dy: the affine pass re-reads dy) and the affine gradients dweight / dbias (shapes of weight / bias), summed over frames in
 * a fixed order.  workspace >= dfd_adapter_norm_gelu_bwd_workspace(...) bytes. */
size_t dfd_adapter_norm_gelu_bwd_workspace(int frames, int patches, int x, int joint);
int dfd_adapter_norm_gelu_bwd(const void* a, int a_dtype, const void* dy, void* da, int dtype, const float* weight, const float* bias,
                              float* dweight, float* dbias, void* workspace, int frames, int patches, int x, int joint,
                              float eps, void* stream);

/* ---- decoder backward (the encoder is frozen: reference models.py:440, :501; these are the
 *      gradients `accelerator.backward` produces in the reference train step, trainer.py:157-165) ---- */

/* Gradient of dfd_decoder_attn_fwd w.r.t. its query and, because the exported K/V are
 * `encoder_kv + temporal positional embedding` (models.py:326-329), w.r.t. that embedding:
 *   dq   f32 [B, heads, 2*d];  dpos f32 [T, heads*d] = Σ_{clip, patch} (dK + dV), or NULL;
 *   dk, dv [B, S, heads*d] in dkv_dtype, or NULL: the full key/value gradients (adapter training only);
 *   dmix, mix_softmax [B, heads*d], stats [B, heads, 2] from the forward;
 *   workspace >= dfd_decoder_attn_bwd_workspace(B, T, heads, d) bytes. */
size_t dfd_decoder_attn_bwd_workspace(int B, int T, int heads, int d);
int dfd_decoder_attn_bwd(const float* q, const void* k, const void* v, int kv_dtype, const dfd_kv_layout_t* layout,
                         const uint8_t* frame_mask, const float* dmix, const float* mix_softmax, const float* stats, const float* ext_weights,
                         const float* ext_dscores, float* dq, float* dpos, void* dk, void* dv, int dkv_dtype,
                         void* workspace, int B, int T, int patches, int heads, int d, void* stream);

/* attn_mode backward, first half: from the forward's scores and dmix, dscores [B, heads, S] = dL/d(scores)
 * through the grouped softmaxes.  dwv_workspace: f32 [B, heads, S].  Then call dfd_decoder_attn_bwd with
 * ext_weights = the forward's weights and ext_dscores = dscores (mix_softmax / stats may be NULL). */
int dfd_decoder_attn_modes_bwd(const float* scores, const void* v, int kv_dtype, const dfd_kv_layout_t* layout,
                               const float* dmix, int modes, float* dwv_workspace, float* dscores, int B, int T, int patches,
                               int heads, int d, void* stream);

/* dW[N,K] = dyᵀ x, db[N] = Σ_b dy (db may be NULL): weight gradient of dfd_linear_rows. */
int dfd_linear_rows_bwd_weight(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dW, float* db, int B,
                               int N, int K, void* stream);

/* dst[cols, rows] = src[rows, cols]ᵀ — the data gradient of dfd_linear_rows is dfd_linear_rows on Wᵀ. */
int dfd_transpose_f32(const float* src, float* dst, int rows, int cols, void* stream);

/* One fused step of SGD with momentum and weight decay over a list of f32 parameters — `Detector.configure_optimizers`
 * (reference src/models.py:740-754, stepped once per batch by src/trainer.py:157-177) — in torch.optim.SGD's arithmetic:
 * g = grad + wd p; buf = first_step ? g : momentum buf + g; p -= lr buf.  An entry with `mirror` != NULL is a [rows, cols]
 * weight whose transposed copy [cols, rows] (what dfd_linear_rows_t reads) is rewritten by the same launch.  The table
 * lives in DEVICE memory (the pointers of a model do not change from step to step; lr and momentum do, every step, under
 * OneCycleLR); `first_block` of entry i = sum of dfd_sgd_blocks(...) of the entries before it, `total_blocks` the sum over
 * all. */
typedef struct dfd_sgd_param {
  float* p;
  const float* g;
  float* buf;
  float* mirror;
  int64_t numel;
  int32_t rows, cols;
  int64_t first_block;
} dfd_sgd_param;
int64_t dfd_sgd_blocks(int64_t numel, int rows, int cols, int mirrored);
int dfd_sgd_step(const dfd_sgd_param* table_dev, int n, int64_t total_blocks, float lr, float momentum, float weight_decay,
                 int first_step, void* stream);

/* LayerNorm backward over `rows` rows: dx = [dx +] ∂L/∂x (accumulate_dx != 0 adds into dx: the
 * residual branch), dgamma/dbeta [cols] summed over rows; xhat_ws: rows*cols floats of scratch. */
int dfd_layernorm_bwd(const float* x, int64_t ldx, const float* gamma, const float* dy, int64_t lddy, float* dx,
                      int64_t lddx, float* dgamma, float* dbeta, float* xhat_ws, int rows, int cols, float eps,
                      int accumulate_dx, void* stream);

/* QuickGELU on n elements: du == NULL: out = drop(u·σ(1.702u)); otherwise out = drop'(du) · d/du[u·σ(1.702u)].
 * drop (may be NULL) = the decoder MLP's nn.Dropout after the activation (models.py:163), same mask both ways. */
int dfd_quickgelu(const float* u, const float* du, float* out, int64_t n, const dfd_dropout_t* drop, void* stream);

/* Backward of dfd_head_fwd after its LayerNorm: dz = ∂L/∂(raw logits) through 5z/(‖z‖+1e-10),
 * dfeat = dz·projᵀ (+ dfeat_ext if not NULL), dproj [D, out_dim] = featᵀ dz. */
int dfd_head_bwd(const float* raw_logits, const float* dlogits, const float* proj, const float* feat,
                 const float* dfeat_ext, float* dz, float* dfeat, float* dproj, int B, int D, int out_dim,
                 void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DFDCLIP_H */
