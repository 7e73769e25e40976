/*
 * pigeon_hip.h — C ABI of libpigeon_hip.so (gfx950 / MI355X only).
 *
 * Drop-in boundary for the image-generation hot path of YiyanXu/Pigeon
 * (Pigeon/models/lavit_for_pigeon.py:674-789).  The reference has no FFI: every
 * kernel on its path lives in third-party wheels (diffusers 0.27.2, xformers
 * 0.0.27.dev792, torch 2.3).  Each entry point below therefore cites the
 * reference call site (or the third-party op reached from it) that it replaces.
 *
 * Conventions
 *   - plain pointers are DEVICE pointers unless the name says `host`;
 *   - nothing here allocates, frees or synchronises: work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the legacy default stream);
 *   - return value: 0 = ok, <0 = PGN_ERR_* (argument / shape refused before any
 *     launch), >0 = a hipError_t from the launch;
 *   - re-entrant per stream; no process-wide switches (no environment variable is read), no mutable global
 *     state except pgn_groupnorm_silu_nhwc_static's per-device scratch array, which its comment describes;
 *   - bf16 / f16 tensors are passed as raw 16-bit storage.
 */
#ifndef PIGEON_HIP_H
#define PIGEON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGN_VERSION 112 /* 0.1.3: + pgn_bias_residual_add_stats, pgn_groupnorm_silu_nhwc_pre (GroupNorm statistics emitted by the launch that produces the tensor); 0.1.2: the channels-last GroupNorm entries take their choice of form as bits of `act` (the library reads no environment); 0.1.1: + pgn_token_embed, pgn_attention_causal_fwd, pgn_groupnorm_silu_nhwc_static */

#define PGN_OK 0
#define PGN_ERR_INVALID (-1)     /* null pointer, non-positive size, bad enum, misalignment */
#define PGN_ERR_UNSUPPORTED (-2) /* valid request the gfx950 kernels do not cover (e.g. head_dim) */
#define PGN_ERR_SCRATCH (-3)     /* pgn_groupnorm_silu_nhwc_static: the shape needs more than PGN_GN_STATIC_SCRATCH_BYTES */

typedef enum { PGN_F32 = 0, PGN_BF16 = 1, PGN_F16 = 2 } pgn_dtype;

/* prediction types of diffusers schedulers (DDIMScheduler.step / EulerDiscreteScheduler.step) */
typedef enum { PGN_PRED_EPSILON = 0, PGN_PRED_SAMPLE = 1, PGN_PRED_V = 2 } pgn_pred_type;

/* flags for the fused step kernels */
#define PGN_STEP_FUSED_FP32 1u /* keep every intermediate in fp32 (FMA allowed). Default (flag clear)
                                  reproduces the reference's rounding points: CFG arithmetic and the
                                  coefficient*eps products are rounded to the eps dtype, exactly as
                                  torch does for `bf16_tensor op scalar` (lavit_for_pigeon.py:761,764). */
#define PGN_STEP_CLIP_SAMPLE 2u        /* DDIM `clip_sample`: clamp pred_original_sample to +-clip_range */
#define PGN_STEP_USE_CLIPPED_OUTPUT 4u /* DDIM `use_clipped_model_output`: re-derive eps from the clipped x0 */

int pgn_version(void);
const char* pgn_error_string(int code);

/* Per-op workspace query kept for ABI stability: 0 for every op (the one shape-dependent workspace, GroupNorm's
 * split path, has its own query: pgn_groupnorm_workspace_bytes). */
typedef enum { PGN_OP_GATHER = 0, PGN_OP_CFG_STEP = 1, PGN_OP_GROUPNORM = 2, PGN_OP_ATTENTION = 3 } pgn_op;
size_t pgn_workspace_bytes(int op);

/* ---------------------------------------------------------------------------------------------
 * Token -> codebook-embedding gather.
 * Replaces LaVITforPigeon.generate_image_embeds, Pigeon/models/lavit_for_pigeon.py:897-918
 * (id - 32002, keep >= 0, truncate to 256, F.embedding on the [16384,32] fp32 codebook of
 * Pigeon/models/modeling_visual_tokenzier.py:54-64, cast to the model dtype) and the pad/mask loop
 * of HighresVQDecoder.forward, Pigeon/models/modeling_decoder.py:310-317.
 *
 * ids        int64 [offsets[B]]   concatenated raw LLM ids of B samples (ragged)
 * offsets    int32 [B+1]          sample b owns ids[offsets[b] .. offsets[b+1])
 * codebook   fp32  [num_codes, dim]   dim % 4 == 0
 * token_nums int64 [B]            OUT: kept tokens per sample (<= max_tokens)
 * out_packed out_dtype [sum(token_nums), dim]  OUT, may be NULL; caller sizes it for the worst case
 *                                 min(len_b, max_tokens) summed over b
 * out_padded out_dtype [B, pad_len, dim] OUT, may be NULL: right-padded with zeros
 * out_mask   out_dtype [B, pad_len]      OUT, may be NULL: 1 for kept tokens else 0
 * status     int32 [1]            OUT, may be NULL: set to 1 if a kept id is >= num_codes (F.embedding
 *                                 would raise); such rows are written as zeros.  Caller zeroes it first.
 * Bit-exact with the reference (integer index math + exact fp32->bf16 RNE cast).
 */
int pgn_token_gather(const int64_t* ids, const int32_t* offsets, int B, const float* codebook, int num_codes,
                     int dim, int64_t id_offset, int max_tokens, int out_dtype, int64_t* token_nums,
                     void* out_packed, void* out_padded, void* out_mask, int pad_len, int32_t* status,
                     void* stream);

/* ---------------------------------------------------------------------------------------------
 * Token gather carried on into the token decoder's first layer: the rows pgn_token_gather would write as
 * out_padded go through `in_proj`, padding rows are zeroed, `pos_embed` is added, and the additive key
 * bias of the decoder's cross-attention is emitted - two launches for what HighresVQDecoder.forward /
 * VQDecoder.forward do in Pigeon/models/modeling_decoder.py:310-319 (pad loop, `self.in_proj(x) * mask`,
 * `+ self.pos_embed`) and :100-101 (`(1 - mask) * -10000`).
 *
 * ids / offsets / codebook / id_offset / max_tokens / token_nums / status   as for pgn_token_gather; dim <= 64
 * dtype           PGN_BF16 or PGN_F16: type of in_proj_weight, in_proj_bias, pos_embed and out
 * in_proj_weight  dtype [E, dim]   (nn.Linear layout)          in_proj_bias  dtype [E] or NULL
 * pos_embed       dtype [>= pad_len, E]
 * out             dtype [B, pad_len, E]   OUT: round(round(x_row . W^T + b) * mask + pos_embed)
 * key_bias        fp32  [B, pad_len]      OUT, may be NULL: 0 for kept tokens, -10000 for padding
 * Rounding points are the reference's (16-bit rows, fp32 dot + bias, 16-bit GEMM output, 16-bit sum); only the
 * order of the fp32 dot differs from a library GEMM's, so results agree to 1 ulp of the 16-bit type.
 */
int pgn_token_embed(const int64_t* ids, const int32_t* offsets, int B, const float* codebook, int num_codes,
                    int dim, int64_t id_offset, int max_tokens, int dtype, const void* in_proj_weight,
                    const void* in_proj_bias, const void* pos_embed, int E, int64_t* token_nums, void* out,
                    float* key_bias, int pad_len, int32_t* status, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused classifier-free-guidance combine + DDIM step (eta = 0).
 * Replaces lavit_for_pigeon.py:760-764: `chunk(2)`, `eps_u + g*(eps_t - eps_u)` and
 * diffusers 0.27.2 DDIMScheduler.step(...).prev_sample.
 *
 *   eps   = eps_u + g*(eps_t - eps_u)
 *   epsilon:      x0 = (x - sqrt_beta_t*eps)/sqrt_alpha_t            ; e = eps
 *   sample:       x0 = eps                                            ; e = (x - sqrt_alpha_t*x0)/sqrt_beta_t
 *   v_prediction: x0 = sqrt_alpha_t*x - sqrt_beta_t*eps               ; e = sqrt_alpha_t*eps + sqrt_beta_t*x
 *   (PGN_STEP_CLIP_SAMPLE clamps x0; PGN_STEP_USE_CLIPPED_OUTPUT then re-derives e from it)
 *   prev  = sqrt_alpha_prev*x0 + dir_coef*e        with dir_coef = sqrt(1 - alpha_prev)
 *
 * eps_uncond/eps_text: eps_dtype [numel] (eps_uncond may be NULL: no guidance, eps = eps_text);
 * sample/prev_sample: sample_dtype [numel] (fp32, or the same 16-bit type as eps);
 * pred_x0 (sample_dtype, may be NULL) receives pred_original_sample.  Scalars are the fp32 values
 * the scheduler computes on the host.  12 algorithmic bytes per element (bf16 eps, fp32 latents).
 */
int pgn_cfg_ddim_step(const void* eps_uncond, const void* eps_text, int eps_dtype, const void* sample,
                      void* prev_sample, void* pred_x0, int sample_dtype, int64_t numel, float guidance,
                      float sqrt_alpha_t, float sqrt_beta_t, float sqrt_alpha_prev, float dir_coef,
                      int prediction_type, unsigned flags, float clip_range, void* stream);

/* Fused CFG combine + Euler step (diffusers 0.27.2 EulerDiscreteScheduler.step with gamma = 0):
 *   eps = eps_u + g*(eps_t - eps_u);  x0 per prediction type with sigma;  d = (x - x0)/sigma;
 *   prev = x + d*(sigma_next - sigma)
 * `sample` (sample_dtype) is the UNSCALED latent and is upcast to fp32 as diffusers does; prev_sample
 * and pred_x0 are written in out_dtype (diffusers casts back to the eps dtype; PGN_F32 keeps fp32). */
int pgn_cfg_euler_step(const void* eps_uncond, const void* eps_text, int eps_dtype, const void* sample,
                       int sample_dtype, void* prev_sample, void* pred_x0, int out_dtype, int64_t numel,
                       float guidance, float sigma, float sigma_next, int prediction_type, unsigned flags,
                       void* stream);

/* `torch.cat([latents]*2)` + scheduler.scale_model_input + the cast conv_in's autocast performs
 * (lavit_for_pigeon.py:747-748): out[0:numel] = out[numel:2*numel] = cast(latents / divisor).
 * divisor = 1 for DDIM, sqrt(sigma^2+1) for Euler. */
int pgn_cfg_expand_scale(const void* latents, int in_dtype, void* out, int out_dtype, int64_t numel,
                         float divisor, void* stream);

/* ---------------------------------------------------------------------------------------------
 * GroupNorm (+ optional SiLU), NCHW contiguous.  Replaces the GroupNorm -> SiLU pairs inside
 * diffusers ResnetBlock2D / conv_norm_out and the plain GroupNorm of Transformer2DModel.norm,
 * reached through `self.unet(...)` at lavit_for_pigeon.py:752-757 (torch.nn.functional.group_norm
 * run in fp32 by autocast, then F.silu, then the bf16 cast of the following conv).
 * x, y: [N, C, HW]; gamma/beta: affine_dtype [C] (may both be NULL = no affine); C % G == 0.
 * Statistics and arithmetic in fp32, one rounding to y_dtype at the end.
 * act: 0 = none, 1 = SiLU.  Algorithmic bytes: numel*(sizeof x + sizeof y).
 */
int pgn_groupnorm_silu(const void* x, int x_dtype, void* y, int y_dtype, const void* gamma, const void* beta,
                       int affine_dtype, int N, int C, int HW, int G, float eps, int act, void* stream);

/* The same op with two optional extensions (either may be absent):
 *  add_nc  [N, C] of add_dtype, or NULL: y = act(GN(round_x(x + add_nc[n, c]))).  Replaces the
 *          `hidden_states = hidden_states + temb[:, :, None, None]` add that precedes norm2 in diffusers 0.27.2
 *          ResnetBlock2D.forward (a bf16 tensor add, hence the rounding to x's type before the statistics).
 *  workspace  caller-provided device memory for shapes with few (sample, group) pairs and large groups (the VAE
 *          decoder's GroupNorms at batch 1, diffusers AutoencoderKL.decode reached from lavit_for_pigeon.py:774: 32
 *          pairs of up to 4 Mi elements; the UNet at batch 1): one workgroup per pair would leave the chip idle, so
 *          each group is cut into slices — per-slice partial statistics go to the workspace, a second launch reduces
 *          them in a fixed order and normalises.  pgn_groupnorm_workspace_bytes returns the bytes the shape needs
 *          (0: the one-launch path is used and `workspace` may be NULL); 8-byte aligned, contents irrelevant on
 *          entry.  With add_nc the one-launch path is always used. */
size_t pgn_groupnorm_workspace_bytes(int N, int C, int HW, int G);
int pgn_groupnorm_silu_ex(const void* x, int x_dtype, void* y, int y_dtype, const void* gamma, const void* beta,
                          int affine_dtype, const void* add_nc, int add_dtype, int N, int C, int HW, int G, float eps,
                          int act, void* workspace, size_t workspace_bytes, void* stream);

/* GroupNorm (+SiLU) for CHANNELS-LAST activations: x, y are [N, HW, C] (torch memory_format channels_last of the
 * logical [N, C, H, W]), the layout the MIOpen / CK convolution kernels compute in — keeping the UNet / VAE in it
 * removes the NCHW<->NHWC transposes MIOpen otherwise runs around every convolution and makes the [B, HW, C] view of
 * diffusers Transformer2DModel free.  Same reference call sites and arithmetic as pgn_groupnorm_silu_ex (add_nc as
 * there).  Three launches (per-chunk per-channel partial statistics -> per-group mean / rstd -> apply: 2 reads + 1
 * write), or, for tensors of up to 32 MiB whose y does not overlap x (C <= 4096, G <= 64), ONE launch behind a small
 * memset that reads and writes the tensor once: each workgroup keeps its rows in registers and the workgroups of a
 * sample exchange per-group partial sums inside the launch (agent-scope hand-off, bounded wait with a recompute safety
 * net: no dependence on dispatch order).  Both forms are deterministic (run-to-run bit-identical, no float atomics);
 * they differ from each other in the last bit of some outputs (two summation orders).  `act` is a bit set: bit 0
 * PGN_GN_ACT_SILU, and the caller's choice of form - PGN_GN_ONEPASS_NEVER keeps the three launches for every shape,
 * PGN_GN_ONEPASS_ALWAYS takes the one-launch form wherever the shape allows (it is slower than three passes on large
 * tensors: DESIGN.md section 4), PGN_GN_ONEPASS_NO_WAIT sets its wait bound to zero so that every workgroup takes the
 * recompute safety net (tests); neither NEVER nor ALWAYS: the 32 MiB rule.  act = 0 / 1 mean what they always meant.
 * The workspace (pgn_groupnorm_nhwc_workspace_bytes, 8-byte aligned, contents irrelevant) is mandatory and must not be
 * shared by calls that may run concurrently.
 * bf16 / f16, C % 8 == 0, C <= 8192, C / G <= 128.  Algorithmic bytes as pgn_groupnorm_silu. */
#define PGN_GN_ACT_SILU 1
#define PGN_GN_ONEPASS_NEVER 2
#define PGN_GN_ONEPASS_ALWAYS 4
#define PGN_GN_ONEPASS_NO_WAIT 8
#define PGN_GN_PIPELINE_NEVER 16  /* accepted and ignored by the shipped library (see the next line) */
#define PGN_GN_PIPELINE_ALWAYS 32 /* measurement build only: the pipelined persistent one-pass form (slower than the three
                                     launches); the shipped library accepts the bit and ignores it */
size_t pgn_groupnorm_nhwc_workspace_bytes(int N, int C, int HW, int G);
int pgn_groupnorm_silu_nhwc(const void* x, void* y, int dtype, const void* gamma, const void* beta, int affine_dtype,
                            const void* add_nc, int add_dtype, int N, int C, int HW, int G, float eps, int act,
                            void* workspace, size_t workspace_bytes, void* stream);

/* The same GroupNorm (+SiLU) over the channel concatenation [x1 | x2] read in place: x1 is [N, HW, C1], x2 is
 * [N, HW, C2], y is [N, HW, C1 + C2].  Replaces `hidden_states = torch.cat([hidden_states, res_hidden_states], dim=1)`
 * of diffusers CrossAttnUpBlock2D.forward / UpBlock2D.forward (unet_2d_blocks.py, reached from
 * lavit_for_pigeon.py:752-757) together with the `norm1` + SiLU of the ResnetBlock2D that consumes it: the
 * concatenated tensor is never written.  Bit-identical to pgn_groupnorm_silu_nhwc on the materialised concatenation
 * (same partial sums in the same order, same choice of form); groups may straddle the boundary.  C1 % 8 == 0, C2 % 8 == 0,
 * other limits and the workspace (query with C = C1 + C2) as above.  Algorithmic bytes: 2 * N*HW*(C1+C2) * 2,
 * against 4 * ... for cat + GroupNorm. */
int pgn_groupnorm_silu_nhwc_cat(const void* x1, int C1, const void* x2, int C2, void* y, int dtype, const void* gamma,
                                const void* beta, int affine_dtype, const void* add_nc, int add_dtype, int N, int HW,
                                int G, float eps, int act, void* workspace, size_t workspace_bytes, void* stream);

/* pgn_groupnorm_silu_nhwc / _cat for inputs that come with the partial statistics of their producer
 * (pgn_bias_residual_add_stats): part1 belongs to x1 [N, HW, C1], part2 to x2 [N, HW, C2] (x2 = part2 = NULL and
 * C2 = 0 for a single input; with two inputs both must bring partials).  Two launches (per-group mean / rstd from the
 * partials, apply): the tensor is read once and written once.  For a single input the result is bit-identical to
 * pgn_groupnorm_silu_nhwc's three-launch form (same words, same order); for a concatenation the halves' chunkings
 * differ from the concatenation's own, so mean / rstd agree to fp32 rounding and outputs to <= 1 unit in the last
 * place of the 16-bit type.  No add_nc (the partials describe the tensor as stored).  act: 0 or PGN_GN_ACT_SILU.
 * workspace: at least N * G * 8 bytes, 8-byte aligned.  Same reference call sites as pgn_groupnorm_silu_nhwc / _cat. */
int pgn_groupnorm_silu_nhwc_pre(const void* x1, int C1, const void* part1, const void* x2, int C2, const void* part2,
                                void* y, int dtype, const void* gamma, const void* beta, int affine_dtype, int N, int HW,
                                int G, float eps, int act, void* workspace, size_t workspace_bytes, void* stream);

/* Allocation-free variant of pgn_groupnorm_silu_nhwc.  NO COUNTERPART IN THE REFERENCE: it exists for callers that
 * cannot allocate device memory around the call — code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  Same kernels, arithmetic and results (bit-identical) as pgn_groupnorm_silu_nhwc; the
 * partial-statistics workspace is a static device array inside the library (PGN_GN_STATIC_SCRATCH_BYTES per GPU, held
 * until the process ends) instead of caller memory, and the call path contains no hipMalloc / hipFree / allocating
 * library routine.  The first call on a device resolves that device's copy of the array (hipGetSymbolAddress; make
 * that call outside graph capture).  Shapes whose workspace (pgn_groupnorm_nhwc_workspace_bytes) exceeds
 * PGN_GN_STATIC_SCRATCH_BYTES return PGN_ERR_SCRATCH before any launch (for N <= 768 the workspace is about
 * 768 * C * 8 bytes whatever the batch: every shape with C <= 2560, i.e. the whole SDXL UNet and the VAE, fits; beyond
 * 768 samples it is N * C * 8 and grows with N).  `stream` must belong to the CURRENT device (the scratch array is
 * resolved for the current device; a stream of another GPU is refused with PGN_ERR_INVALID).  All calls on one GPU share the array: calls that overlap on one GPU (two streams, two
 * threads) race — serialise them on one stream.  Nothing in it has to start at zero. */
#define PGN_GN_STATIC_SCRATCH_BYTES (16u << 20)
int pgn_groupnorm_silu_nhwc_static(const void* x, void* y, int dtype, const void* gamma, const void* beta,
                                   int affine_dtype, const void* add_nc, int add_dtype, int N, int C, int HW, int G,
                                   float eps, int act, void* stream);

/* Nearest-neighbour 2x upsampling of a CHANNELS-LAST activation: x [N, H, W, C] -> y [N, 2H, 2W, C],
 * y[n, 2h+dy, 2w+dx, :] = x[n, h, w, :].  Replaces `F.interpolate(hidden_states, scale_factor=2.0, mode="nearest")` of
 * diffusers Upsample2D.forward (models/upsampling.py), reached from `self.unet(...)` lavit_for_pigeon.py:752-757 (two
 * per forward) and `self.vae.decode(...)` :774 (three per image).  Pure data movement (bit-exact); bf16 / f16,
 * C % 8 == 0, x != y.  Algorithmic bytes: N*H*W*C * 2 * (1 + 4). */
int pgn_upsample_nearest2x_nhwc(const void* x, void* y, int dtype, int N, int H, int W, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-head attention forward, softmax(q k^T * scale + key_bias) v, flash-style on MFMA.
 * Replaces xformers.ops.memory_efficient_attention(q, k, v, attn_bias, scale) as called by diffusers
 * XFormersAttnProcessor (enabled at lavit_for_pigeon.py:202) — self- and cross-attention.
 *
 * q [B, Nq, H, D], k/v [B, Nk, H, D], out [B, Nq, H, D]; last dim contiguous, the other three given
 * as ELEMENT strides (batch, sequence, head), so the [B, N, H*D] output of to_q/to_k/to_v is read in
 * place (no head_to_batch_dim copy).  dtype: PGN_BF16 or PGN_F16 storage, fp32 softmax/accumulate.
 * key_bias: fp32 [B, Nk] additive bias broadcast over heads and queries (key padding:
 * lavit_for_pigeon.py:873-874 lowres path; modeling_decoder.py:100-102), or NULL.
 * Any Nq, Nk >= 1 (ragged tails handled).  D: any multiple of 8 up to 256 — 64 is the SDXL UNet (every level),
 * 88 the token decoder (modeling_decoder.py:44-109, 16 heads x 88), 40 / 80 / 160 the SD-1.5 lowres UNet; head dims
 * other than 64 / 128 / 256 are zero-filled to the next of the three inside the kernel (no padded copies); above 128
 * the kernel runs one wavefront per SIMD (correct, not tuned).  D > 256 (the VAE's single 512-wide head) ->
 * PGN_ERR_UNSUPPORTED.
 * Algorithmic work: 4*B*H*Nq*Nk*D flop.
 */
int pgn_attention_fwd(const void* q, const void* k, const void* v, void* out, int dtype, int B, int H, int Nq,
                      int Nk, int D, int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn,
                      int64_t k_sh, int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn,
                      int64_t o_sh, const float* key_bias, int64_t bias_sb, float scale, void* stream);

/* Which kernel class pgn_attention_fwd launches for a shape (host arithmetic, no launch): the 64-rows-per-wave ring
 * kernel (head dim 64, no key bias, >= 256 keys, enough workgroups to fill the chip: DESIGN.md section 5) or the
 * 32-rows-per-wave kernel.  No counterpart in the reference; it exists so that callers that time launches per kernel
 * (bench.py's `roofline` / `kernels`) label them by the library's own rule instead of a copy of it. */
#define PGN_ATT_KERNEL_ROWS32 0
#define PGN_ATT_KERNEL_RING64 1
int pgn_attention_kernel_class(int B, int H, int Nq, int Nk, int D, int has_key_bias);

/* ---------------------------------------------------------------------------------------------
 * Causal multi-head self-attention over a short sequence: out[i] = softmax_{j <= i}(q_i . k_j * scale) v_j.
 * Replaces the `attn_mask` form of nn.MultiheadAttention inside open_clip's text Transformer, reached from
 * `self.model.encode_text(texts)` at Evaluation/eval_utils.py:126 (CLIPScore.calculate_clip_score; ViT-H-14 text tower:
 * 77 tokens, 16 heads x 64, upper-triangular -inf mask).  Layout and strides as for pgn_attention_fwd
 * (q / k / v / out [B, N, H, D], element strides, last dim contiguous).  N <= 128, D even and <= 128, bf16 / f16;
 * otherwise PGN_ERR_UNSUPPORTED.  Latency-bound (0.4 GFLOP per call at batch 16): one workgroup per (batch, head),
 * fp32 dot products and probabilities, output rounded once.
 */
int pgn_attention_causal_fwd(const void* q, const void* k, const void* v, void* out, int dtype, int B, int H, int N,
                             int D, int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn,
                             int64_t k_sh, int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn,
                             int64_t o_sh, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Row softmax of materialised attention scores:  probs[m, :] = softmax(scale * scores[m, :]).
 * Middle step of the VAE decoder's single 512-wide attention head (diffusers AutoencoderKL mid-block Attention, reached
 * through `self.vae.decode(...)` at lavit_for_pigeon.py:774; the reference leaves it on diffusers' default processor:
 * only self.unet is switched to xformers at :202): scores = q k^T from a 16-bit GEMM with fp32 output, probs go into the
 * second GEMM in the activation type.  fp32 statistics; same rounding points as the fused kernel (fp32 softmax, 16-bit
 * P).  scores: fp32 [M, N] rows of ld_scores elements (16-byte aligned, ld % 4 == 0); probs: bf16 / f16 [M, N] rows of
 * ld_probs elements (ld % 4 == 0); scale > 0.
 * Algorithmic bytes: M*N*(4 + sizeof(elem)).
 */
int pgn_softmax_rows(const float* scores, int64_t ld_scores, void* probs, int probs_dtype, int64_t ld_probs, int64_t M,
                     int64_t N, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * GEGLU gate of the transformer feed-forward: y[m, j] = h[m, j] * gelu(h[m, D + j]), exact (erf) GELU.
 * Replaces diffusers 0.27.2 GEGLU.forward (`hidden, gate = proj(x).chunk(2, dim=-1); hidden * F.gelu(gate)`)
 * inside BasicTransformerBlock.ff, reached through `self.unet(...)` at lavit_for_pigeon.py:752-757 — in torch
 * a strided gelu launch plus a multiply launch.  Both torch ops round to the io type, and so does this
 * kernel (gelu in fp32 -> round -> product in fp32 -> round): bit-exact with the two-op form.
 * h: [M, 2*D] rows of ld_h elements; y: [M, D] rows of ld_y elements; D % 8 == 0; bf16 / f16.
 * Algorithmic bytes: M*D*3*sizeof(elem).
 */
int pgn_geglu(const void* h, void* y, int dtype, int64_t M, int D, int64_t ld_h, int64_t ld_y, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Scoring stage: per-image target score and history score from CLIP features.
 *   tgt_score[m]  = 100 * cos(img[m], txt[m])                          Evaluation/eval_utils.py:121-129
 *                   (CLIPScore.calculate_clip_score, called from cal_scores.py:160-165)
 *   hist_score[m] = mean_k 100 * cos(img[m], hist[m, k])               eval_utils.py:540-548 + the mean over the
 *                   5 history images of cal_scores.py:178-179
 * img_feat: [M, D] rows of img_ld elements (f32 / bf16 / f16: the image encoder's output); txt_feat: [M, D] rows of
 * txt_ld elements or NULL (then tgt_score is not written); hist_feat: fp32 [M, n_hist, D], sample stride hist_ld, or
 * n_hist == 0.  n_hist <= 16.  All arithmetic fp32; cos(a, b) = a.b / max(|a||b|, 1e-8) as F.cosine_similarity.
 * These are the two per-image floats the multi-GPU all-gather carries.
 * Algorithmic bytes: M * D * (sizeof(img) + sizeof(txt) + 4 * n_hist).
 */
int pgn_clip_scores(const void* img_feat, int img_dtype, int64_t img_ld, const void* txt_feat, int txt_dtype,
                    int64_t txt_ld, const float* hist_feat, int64_t hist_ld, int n_hist, float* tgt_score,
                    float* hist_score, int64_t M, int D, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Convolution bias add, optionally fused with a residual add:  y = round(round(x + bias[c]) + residual).
 * Replaces the broadcast bias add that follows every bias-less MIOpen convolution of diffusers 0.27.2
 * ResnetBlock2D / Downsample2D / Upsample2D / conv_in (`self.conv2(hidden_states)`), and ResnetBlock2D.forward's
 * `output_tensor = input_tensor + hidden_states` (reached through `self.unet(...)` at lavit_for_pigeon.py:752-757 and
 * `self.vae.decode(...)` at :774): two torch launches (a strided broadcast add_ and an add, 5 tensor passes) become
 * one (3 passes).  Same two roundings as the torch pair.
 * x, residual (or NULL), y: [rows][C][inner] in memory order, bf16 / f16, 16-byte aligned; inner = 1 for channels-last
 * activations (C % 8 == 0, bias 16-byte aligned), inner = H*W for NCHW (inner % 8 == 0).  bias: [C] of bias_dtype.
 * y may alias x.
 * Algorithmic bytes: rows*C*inner*sizeof(elem) * (3 with residual, 2 without).
 */
int pgn_bias_residual_add(const void* x, const void* residual, const void* bias, int bias_dtype, void* y, int dtype,
                          int64_t rows, int C, int64_t inner, void* stream);

/* The same add for a CHANNELS-LAST activation [N, HW, C] whose consumer is a channels-last GroupNorm, emitting that
 * GroupNorm's partial statistics in the same pass: y as pgn_bias_residual_add writes it (bit-identical), and `part`
 * [N][S][C] pairs of floats — per (sample, row chunk, channel) the pivot-shifted sum and sum of squares of y that the
 * first of pgn_groupnorm_silu_nhwc's three launches would otherwise compute by reading y back (bit-identical words:
 * same chunking, a function of (N, C, HW) alone, same pivots, same order).  Same reference lines as
 * pgn_bias_residual_add; what it additionally replaces is the statistics read of `self.norm1(hidden_states)` /
 * Transformer2DModel's `self.norm(hidden_states)` in the module that consumes y (diffusers 0.27.2 ResnetBlock2D.forward,
 * Transformer2DModel.forward, reached from lavit_for_pigeon.py:752-757): GroupNorm's traffic on such a tensor falls
 * from 2 reads + 1 write to 1 + 1.
 * pgn_bias_residual_add_stats_bytes: bytes of `part` for the shape, or 0 when the form does not apply (C % 8 != 0 or
 * C > 8192: use pgn_bias_residual_add).  It pays for tensors above 32 MiB; smaller ones the consuming GroupNorm takes
 * in ONE launch without partials (that rule is the caller's: pigeon_amd/ops.py GN_STATS_MIN_BYTES).  y must overlap neither x nor residual (every workgroup of a sample reads three pivot
 * rows of the inputs); bias 16-byte aligned; `part` 8-byte aligned, contents irrelevant on entry.  bf16 / f16.
 * Algorithmic bytes as pgn_bias_residual_add (+ N*S*C*8 for the partials, < 1 % of the tensor). */
size_t pgn_bias_residual_add_stats_bytes(int N, int C, int HW);
int pgn_bias_residual_add_stats(const void* x, const void* residual, const void* bias, int bias_dtype, void* y,
                                int dtype, int N, int C, int HW, void* part, size_t part_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (residual add +) LayerNorm over the last dimension, one pass.
 * Replaces the `attn(...) + hidden_states` / `ff(...) + hidden_states` adds of diffusers 0.27.2
 * BasicTransformerBlock.forward together with the norm1/norm2/norm3 LayerNorm that consumes each sum
 * (torch.nn.functional.layer_norm, run in fp32 by autocast; reached from lavit_for_pigeon.py:752-757):
 *   s = x + residual (rounded to the io type, written to sum_out)      [skipped when residual == NULL]
 *   y = (s - mean(s)) * rsqrt(var(s) + eps) * gamma + beta             fp32 statistics, one rounding
 * x, residual, sum_out, y: [M, C] contiguous, bf16 / f16; gamma/beta: affine_dtype [C] or both NULL.
 * residual and sum_out are given together or both NULL.  C % 8 == 0, C <= 4096.
 * Algorithmic bytes: M*C*sizeof(elem) * (2 without residual, 4 with).
 */
int pgn_add_layernorm(const void* x, const void* residual, void* sum_out, void* y, const void* gamma,
                      const void* beta, int dtype, int affine_dtype, int64_t M, int C, float eps, void* stream);

/* Same, with a pending per-channel bias: s = x + col_bias[c] (+ residual).  Lets the attention output projections run
 * as bias-free GEMMs that accumulate straight into the residual buffer (beta = 1): the bias they owe travels as an
 * fp32 [C] vector into the next LayerNorm instead of costing a pass over the activations, and that LayerNorm reads one
 * tensor and writes one (2 passes) where add + norm took 4.  col_bias: fp32 [C], 16-byte aligned, or NULL. */
int pgn_add_layernorm_cb(const void* x, const void* residual, void* sum_out, void* y, const void* gamma,
                         const void* beta, int dtype, int affine_dtype, const float* col_bias, int64_t M, int C,
                         float eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * VAE output -> displayable bytes in one pass: y[n, h, w, c] = uint8(rint(clamp(float(x[n, c, h, w]) / 2 + 0.5, 0, 1)
 * * 255)), rint = round half to even.  Replaces lavit_for_pigeon.py:780-782 (`image.float()`, `/ 2 + 0.5`,
 * `.clamp(0, 1)`, `.permute(0, 2, 3, 1)`) and numpy_to_pil :929 (`(images * 255).round().astype("uint8")`), which the
 * reference runs as fp32 tensor ops plus a 4x larger device->host copy.  Each fp32 operation is rounded separately, so
 * the bytes are identical to the reference's.  x: [N, C, H, W] of x_dtype, C <= 4; y: [N, H, W, C] uint8.
 */
int pgn_image_to_uint8(const void* x, int x_dtype, void* y, int N, int C, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PIGEON_HIP_H */
