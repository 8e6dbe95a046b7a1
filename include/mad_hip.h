/*
 * mad_hip.h -- C ABI of libmad_hip.so: the MI355X (gfx950) kernels of the MAD denoising path.
 *
 * The reference has no FFI of its own for this path: it is pure Python calling torch operators
 * (cuDNN / cuBLAS / xformers behind them).  Each entry point below therefore names the reference
 * operator site (file:line under /root/reference) whose arithmetic it replaces; a binding for it
 * is a few lines of ctypes (mad_amd/_lib.py holds the full signature table).
 *
 * Conventions
 *   - every pointer is a device pointer owned by the caller; nothing here allocates or frees;
 *   - every call only enqueues work on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) and returns immediately;
 *   - return value 0 = ok, non-zero = rejected before launch; mad_last_error() gives the reason
 *     (thread-local string);
 *   - activations inside the denoiser are NHWC ("channels last"): element (b, y, x, c) at
 *     ((b*H + y)*W + x)*C + c; dtype is MAD_BF16 (bf16 storage, fp32 accumulate) or MAD_F32;
 *   - tensors at the sampler level are NCHW fp32, as in the reference.
 */
#ifndef MAD_HIP_H
#define MAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAD_BF16 0
#define MAD_F32 1

int mad_version(void);
const char* mad_last_error(void);
/* Hash of the native sources this library was compiled from (mad_amd/buildid.py computes the same hash from the
 * files): lets a caller refuse a binary that does not belong to its source tree.  No counterpart in the reference,
 * which has no native code. */
const char* mad_build_id(void);

/* ------------------------------------------------------------------ sampler (NCHW fp32) ---- */

#define MAD_ALGO_DDPM 0
#define MAD_ALGO_DDIM 1
#define MAD_PRED_EPSILON 0
#define MAD_PRED_SAMPLE 1
#define MAD_PRED_V 2
#define MAD_OP_STEP 0        /* x_t -> x_{t-1}                         */
#define MAD_OP_COMPUTE_EPS 1 /* noise that maps x_t to a given x_{t-1} */
#define MAD_OP_INVERSE 2     /* deterministic DDIM inversion step      */

/* Scalars of one sampler update, computed on the host in fp32 exactly as the reference computes
 * them from alphas_cumprod (scheduler.py:341-346,374-380 DDPM; :45-50,89,106 DDIM). */
typedef struct mad_sampler_coef {
    int32_t algo, pred_type, op;
    int32_t clip;             /* clamp predicted x0 to +-clip_range (scheduler.py:367-370)      */
    int32_t use_clipped;      /* DDIM: re-derive eps from the clamped x0 (scheduler.py:81-85)   */
    int32_t add_variance;     /* step: add std*noise (t > 0, and eta > 0 for DDIM)              */
    int32_t known_noise;      /* step, masked: the known part gets sqrt(1-a_prev)*noise (t > 0) */
    float sqrt_a_t;           /* alpha_prod_t ** 0.5                                            */
    float sqrt_b_t;           /* (1 - alpha_prod_t) ** 0.5                                      */
    float c0, c1;             /* DDPM: x0 and x_t coefficients of the posterior mean            */
    float std;                /* sigma_t                                                        */
    float sqrt_a_prev;        /* alpha_prod_t_prev ** 0.5                                       */
    float sqrt_b_prev;        /* (1 - alpha_prod_t_prev) ** 0.5                                 */
    float dir_coef;           /* DDIM: (1 - a_prev - std^2) ** 0.5                              */
    float clip_range;
    float eps_div;            /* compute_eps divisor (std, or 1 when std/eta is 0)              */
} mad_sampler_coef;

/* One coalesced elementwise pass over (B,C,H,W) fp32.
 * Replaces CustomDDPMScheduler.step (scheduler.py:293-423), .compute_eps (:425-493),
 * .ddim_inverse_step (:526-585) and the CustomDDIMScheduler counterparts (:12-131, :133-197,
 * :230-289).  `noise`, `original`, `mask`, `next_sample`, `x0_out` may be NULL where the
 * operation does not use them.  mask is (mask_batch,1,H,W) with mask_batch 1 or B. */
int mad_sampler_update(const mad_sampler_coef* coef, const float* model_output, const float* sample,
                       const float* noise, const float* original, const float* mask,
                       const float* next_sample, float* out, float* x0_out, int B, int C, int H,
                       int W, int mask_batch, void* stream);

/* out = sa*x0 + sb*noise; with mask: out = mask*(sa*x0 + sb*noise) + (1-mask)*x0.
 * Replaces scheduler.add_noise as used by sample_xt (translation.py:699-706, 781-789). */
int mad_add_noise(const float* x0, const float* noise, const float* mask, float* out, float sa,
                  float sb, int B, int C, int H, int W, int mask_batch, void* stream);

/* out = clamp(x,-1,1) [/2 + 0.5 when normalize] (translation.py:870-873). */
int mad_finalize_image(const float* x, float* out, int normalize, size_t n, void* stream);

/* dst[b,c,y,x] = src[b,c,y,x] where mask[(b),0,y,x] > 0 (translation.py:820-821). */
int mad_masked_copy(float* dst, const float* src, const float* mask, int B, int C, int H, int W,
                    int mask_batch, void* stream);

/* Label map (H,W) uint8 -> keep-mask (H,W) float, 1 = keep the source pixel.  `comps` lists the
 * active component values already sorted by priority; `dilate` enables the 21x21 eye ring.
 * Replaces the tensor branch of process_mask + tensor_dilate (translation.py:54-58, 234-273). */
int mad_keep_mask(const uint8_t* label_map, float* keep, int H, int W, const int32_t* comps,
                  int ncomps, int dilate, void* stream);
/* The same for a stack of B label maps (B,H,W) -> (B,H,W) in one launch: the batched form of the per-image call
 * the reference makes (generate_transfer.py:33-91 translates one pair per call; a batch needs one mask per image). */
int mad_keep_mask_batch(const uint8_t* label_maps, float* keep, int B, int H, int W, const int32_t* comps,
                        int ncomps, int dilate, void* stream);

/* ------------------------------------------------------------------ denoiser pieces -------- */

/* y = act_out(act_in(x) @ W^T + bias) in fp32; x (rows,K), W (N,K) row-major, bias may be NULL.
 * act: 0 none, 1 SiLU.  `accumulate` adds into y instead of overwriting (the class-label term).
 * Replaces Linear.forward (model.py:60-64) for map_layer0/1, map_label and every block's affine. */
int mad_linear_f32(const float* x, const float* W, const float* bias, float* y, int rows, int K,
                   int N, int act_in, int act_out, int accumulate, void* stream);

/* emb[b, :half] = cos(t[b]*freqs), emb[b, half:] = sin(t[b]*freqs)  (model.py:392-400).
 * t has t_count (1 or rows) entries; freqs is the fp32 table (1/10000)**(i/half). */
int mad_pos_embedding(const float* t, int t_count, const float* freqs, float* emb, int rows,
                      int half, void* stream);

/* First convolution: x NCHW fp32 (B,Cin<=4,H,W) -> y NHWC (B,H,W,Cout), 3x3 pad 1, fp32 weights
 * (Cout,Cin,3,3) and bias.  Replaces enc.{R}x{R}_conv (model.py:506-508, 580-582). */
int mad_conv_in(const float* x, const float* w, const float* bias, void* y, int dtype, int B,
                int Cin, int H, int W, int Cout, void* stream);

/* mad_conv_in that also writes the GroupNorm partial sums of its output: stats fp32 (B, (H/8)*(W/16), Cout, 2),
 * per-channel sum / sum of squares of every 8x16-pixel tile (tile = ty*(W/16)+tx), as mad_gn_finalize consumes
 * them with nsplit = (H/8)*(W/16).  Only for shapes mad_conv_in_stats_ok accepts (H % 8 == 0, W % 16 == 0,
 * Cout in {32, 64, 96, 128, 192, 256}).  Replaces the statistics pass of the first block's GroupNorm
 * (model.py:182-190). */
int mad_conv_in_stats_ok(int dtype, int H, int W, int Cout);
int mad_conv_in_stats(const float* x, const float* w, const float* bias, void* y, float* stats, int dtype,
                      int B, int Cin, int H, int W, int Cout, void* stream);

/* Last convolution: x NHWC (B,H,W,Cin) -> y NCHW fp32 (B,Cout<=4,H,W), 3x3 pad 1, fp32 weights
 * laid out (Cout,3,3,Cin) and bias.  Replaces out_conv (model.py:549-551, 589). */
int mad_conv_out(const void* x, int dtype, const float* w, const float* bias, float* y, int B,
                 int Cin, int H, int W, int Cout, void* stream);

/* Last convolution with out_norm + SiLU folded into the loader:
 * y (B,Cout<=32,H,W) NCHW fp32 = conv3x3(act(x*coef[...,0] + coef[...,1])) + bias.  x NHWC (B,H,W,Cin),
 * w (32,3,3,Cin) in `dtype` with rows >= Cout zero, H and W multiples of 16.  bf16 with Cout <= 3 and
 * Cin <= 256 runs the tap-major kernel (the nine taps as the N dimension of one product per halo pixel,
 * every input pixel read and transformed once per tile); other cases run the halo-tile MFMA kernel.
 * Replaces out_conv(silu(out_norm(x))) (model.py:589). */
int mad_conv_out_fused(const void* x, int Cin, const float* coef, int act, const void* w,
                       const float* bias, float* y, int dtype, int B, int H, int W, int Cout,
                       void* stream);

#define MAD_RES_NONE 0
#define MAD_RES_SAME 1 /* residual (B,H,W,Cout)                                   */
#define MAD_RES_UP 2   /* residual (B,H/2,W/2,Cout), nearest-neighbour x2         */
#define MAD_RES_DOWN 3 /* residual (B,2H,2W,Cout), 2x2 mean                       */

/* Implicit-GEMM convolution on MFMA, stride 1, ksize 1 or 3 (pad ksize/2):
 *   y[b,oy,ox,n] = bias[n] + sum_{ky,kx,c} in(b, oy+ky-p, ox+kx-p, c) * w[n,ky,kx,c] (+ residual)
 * `in` is the channel concatenation of x1 (C1 channels) and x2 (C2 channels, may be NULL/0), both
 * NHWC at (H,W) -- or at (H/2,W/2) read through nearest-neighbour x2 when `upsample` is set.
 * w is (Cout,ksize,ksize,C1+C2) in `dtype`; bias fp32.  C1, C2 must be multiples of 64 and Cout a
 * multiple of 64.  Replaces Conv2d.forward (model.py:116-166) for the 3x3 convs, the 1x1
 * skip/qkv/proj convs, the `up` resample (model.py:147-153), the skip concat (model.py:586-587)
 * and the residual adds (model.py:353, 375). */
int mad_conv2d(const void* x1, int C1, const void* x2, int C2, const void* w, const float* bias,
               const void* residual, int res_mode, void* y, int dtype, int B, int H, int W,
               int Cout, int ksize, int upsample, void* stream);

/* mad_conv2d that also writes the GroupNorm partial sums of its output: stats fp32 (B*H*W/64, Cout, 2),
 * per-channel sum / sum of squares of every run of 64 consecutive output pixels (H*W a multiple of
 * 64), i.e. (B, nsplit = H*W/64, Cout, 2) as mad_gn_finalize consumes it.  bf16 1x1 layers that run on the
 * 256-pixel-tile GEMM kernel hold, as mad_conv3x3_fused does, the sums of channels 2i and 2i+1 together in
 * slot 2i and zeros in slot 2i+1: use the slab for sums over groups of whole channel pairs only (GroupNorm
 * groups with an even channel count).  Replaces the statistics pass of the GroupNorm that follows the
 * convolution (model.py:182-190). */
int mad_conv2d_stats(const void* x1, int C1, const void* x2, int C2, const void* w, const float* bias,
                     const void* residual, int res_mode, void* y, float* stats, int dtype, int B, int H,
                     int W, int Cout, int ksize, int upsample, void* stream);

/* mad_conv2d / mad_conv2d_stats (stats may be NULL) with a caller-provided device workspace.  Layers with few
 * output tiles and a long K (the 3x3 layers of the 8x8 stage at small batch) are then cut along K: every share
 * writes an fp32 partial slab into the workspace and a finishing kernel adds the shares in a fixed order before
 * bias, residual, rounding and statistics.  mad_conv2d_workspace_bytes returns the size such a layer needs and
 * 0 for a layer that runs in one pass (then workspace may be NULL).  Same reference lines as mad_conv2d. */
size_t mad_conv2d_workspace_bytes(int dtype, int B, int H, int W, int Cin, int Cout, int ksize);
int mad_conv2d_ws(const void* x1, int C1, const void* x2, int C2, const void* w, const float* bias,
                  const void* residual, int res_mode, void* y, float* stats, int dtype, int B, int H,
                  int W, int Cout, int ksize, int upsample, void* workspace, size_t workspace_bytes,
                  void* stream);

/* 3x3 convolution (pad 1, stride 1) for layers with H, W multiples of 16 and Cout a multiple of 64,
 * with the producer's GroupNorm folded into the operand loader and the consumer's GroupNorm
 * statistics produced by the epilogue:
 *   in(b,y,x,c) = act(x(b,y,x,c)*coef[b,c,0] + coef[b,c,1])   (coef NULL: in = x; zero outside the image)
 *   y = conv3x3(in) + bias (+ residual as in mad_conv2d)
 *   stats[b, tile, n, 0/1] = sum / sum of squares of y over the 16x16 tile (tile = ty*(W/16)+tx);
 *   stats may be NULL.  mad_gn_finalize consumes it with nsplit = (H/16)*(W/16).  In bf16 the sums are
 *   taken per channel PAIR: slot n = 2i holds the sums over channels 2i and 2i+1, slot 2i+1 holds zeros
 *   PRECONDITION on the caller: a GroupNorm that consumes them must have an EVEN number of channels per group
 *   (groups hold whole pairs); the reference's rule num_groups = min(32, C/4) (model.py:174-181) gives that for
 *   every channel count that is a multiple of 64, not for all (C = 160: five per group) -- nothing in
 *   mad_gn_finalize can detect a violation, so a caller with odd groups takes the statistics with mad_gn_partial.
 * x is the concatenation of x1 (C1) and x2 (C2), NHWC at (H,W), or at (H/2,W/2) when `upsample`.
 * C1, C2 multiples of 32 (bf16) / 16 (fp32).
 * act is a set of bits:
 *   MAD_ACT_SILU       act = SiLU (otherwise the identity);
 *   MAD_ACT_PRESCALED  (with coef and MAD_ACT_SILU) coef holds -log2(e) * (a, d), as mad_gn_finalize_scaled writes
 *                      it with scale = MAD_PRESCALE_COEF, and w holds MAD_PRESCALE_W * the layer's weights: the
 *                      loader then computes u = x*a' + d' and u / (1 + 2^u) = -log2(e) * silu(x*a + d), one
 *                      multiply-add per input element less, and the product with the scaled weights is the same
 *                      convolution (the 1x1 projection weights of mad_conv3x3_fused_skip stay unscaled);
 *   MAD_W_TILED        w is stored tile by tile instead of (Cout,3,3,Cin): with CK = 32 (bf16) / 16 (fp32)
 *                      channels per 64-byte chunk, the 1 KiB block [n/16][c/CK][tap] holds rows n%16 of 64 bytes,
 *                      and inside a row the 16-byte piece at position p holds channels
 *                      c0 + ((p ^ (-((n%16)/4) & 3)) * CK/4 ...) of the chunk starting at c0 -- the order the kernel
 *                      keeps a weight tile in LDS, so every copy of it reads 1 KiB in one piece.
 * Replaces GroupNorm -> SiLU -> Conv2d (model.py:341, 346-350) and the statistics pass of the
 * GroupNorm that follows (model.py:182-190). */
#define MAD_ACT_SILU 1
#define MAD_ACT_PRESCALED 2
#define MAD_W_TILED 4
#define MAD_PRESCALE_COEF (-1.44269504088896340736f) /* -log2(e) */
#define MAD_PRESCALE_W (-0.69314718055994530942f)    /* -ln(2) */
int mad_conv3x3_fused(const void* x1, int C1, const void* x2, int C2, const float* coef, int act,
                      const void* w, const float* bias, const void* residual, int res_mode, void* y,
                      float* stats, int dtype, int B, int H, int W, int Cout, int upsample,
                      void* stream);

/* mad_conv3x3_fused plus the residual branch's 1x1 channel projection in the same launch:
 *   y = conv3x3(in) + bias + sw . s        s = concatenation of sx1 (SC1) and sx2 (SC2), NHWC at (H,W),
 * sw is (Cout, SC1+SC2) in the activation dtype, `bias` is the sum of the two layers' biases.  The
 * projection is accumulated as extra K-tiles of the same matrix product (raw input, centre tap).
 * SC1, SC2 multiples of 32 (bf16) / 16 (fp32); no upsample, no other residual.
 * Replaces `x = conv1(...)` followed by `x = x + skip(orig)` of a UNetBlock whose channel count
 * changes (model.py:311-320, 346-353). */
int mad_conv3x3_fused_skip(const void* x1, int C1, const void* x2, int C2, const float* coef, int act,
                           const void* w, const float* bias, const void* sx1, int SC1, const void* sx2,
                           int SC2, const void* sw, void* y, float* stats, int dtype, int B, int H, int W,
                           int Cout, void* stream);

/* The same GroupNorm -> SiLU -> Conv2d(3x3) -> (+ residual) -> statistics fusion for the 8x8 layers
 * (H = W = 8 is implied): one workgroup holds a whole image's input in LDS.  x is the concatenation of
 * x1 (C1) and x2 (C2), NHWC (B,8,8,C); C1, C2 multiples of 64 (bf16) / 32 (fp32); Cout a multiple of 64.
 * coef NULL: the input is used as it is.  res_mode MAD_RES_NONE, MAD_RES_SAME or MAD_RES_DOWN
 * (residual (B,16,16,Cout), 2x2 mean).
 * stats (may be NULL) is fp32 (B, 1, Cout, 2): per-image sum / sum of squares of y (nsplit = 1 for
 * mad_gn_finalize).  Replaces model.py:341, 346-353 and the statistics of model.py:182-190 at the
 * innermost resolution. */
int mad_conv3x3_img8(const void* x1, int C1, const void* x2, int C2, const float* coef, int act,
                     const void* w, const float* bias, const void* residual, int res_mode, void* y,
                     float* stats, int dtype, int B, int Cout, void* stream);

/* GroupNorm statistics, pass 1: per-(sample, split, channel) sum and sum of squares over the
 * split's pixels of the concatenation of x1 (C1) and x2 (C2).  partial is fp32
 * (B, nsplit, C1+C2, 2).  HW must be a multiple of nsplit. */
int mad_gn_partial(const void* x1, int C1, const void* x2, int C2, int dtype, float* partial, int B,
                   int HW, int nsplit, void* stream);

/* GroupNorm statistics, pass 2: reduce the partials per group and fold mean, rstd, the affine
 * gain/offset and the optional AdaGN modulation into one (a, d) pair per (sample, channel):
 *   y = x*a + d  ==  shift + ((x-mean)*rstd*gamma + beta) * (scale + 1)
 * The channel axis is the concatenation of two slabs: partial1 fp32 (B, nsplit1, C1, 2) and
 * partial2 fp32 (B, nsplit2, C2, 2) (NULL/0 for a single source), as written by mad_gn_partial or by
 * the epilogue of mad_conv3x3_fused.  HW is the pixel count the sums run over.
 * ada is fp32 with row stride ada_stride: ada[b*ada_stride + c] = scale, ada[... + C + c] = shift
 * (scale first: model.py:343-346); NULL for a plain GroupNorm.  coef is fp32 (B, C1+C2, 2).
 * Replaces GroupNorm.forward (model.py:182-190) statistics. */
int mad_gn_finalize(const float* partial1, int C1, int nsplit1, const float* partial2, int C2,
                    int nsplit2, const float* gamma, const float* beta, const float* ada,
                    int ada_stride, float* coef, int B, int groups, int HW, float eps, void* stream);
/* mad_gn_finalize with every (a, d) pair multiplied by `scale` as it is written (scale = 1: the same values).
 * With MAD_PRESCALE_COEF the table feeds mad_conv3x3_fused in its MAD_ACT_PRESCALED form.  Same reference
 * lines as mad_gn_finalize. */
int mad_gn_finalize_scaled(const float* partial1, int C1, int nsplit1, const float* partial2, int C2,
                           int nsplit2, const float* gamma, const float* beta, const float* ada,
                           int ada_stride, float* coef, int B, int groups, int HW, float eps, float scale,
                           void* stream);

/* GroupNorm coefficients in ONE call that allocates nothing: mad_gn_partial followed by mad_gn_finalize with
 * the partial-sum slab between them held in a device array inside the library (8 MiB per GPU, kept until the
 * process ends) instead of a caller-owned buffer.  No counterpart in the reference: it exists for callers
 * that cannot allocate device memory between the two passes -- code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  x is the concatenation of x1 (C1) and x2 (C2, may be NULL/0), NHWC
 * with HW pixels per image; the split count is min(64, max(1, HW/64)) as the two-call form uses, and
 * B * splits * (C1+C2) * 2 floats must fit mad_gn_coef_static_max_floats() (rejected with a message otherwise).
 * Result coef fp32 (B, C1+C2, 2), bit-identical to the two-call form.  The first call on a device looks the
 * slab's address up (do it once outside graph capture).  Calls on one GPU share that GPU's slab: two calls
 * that overlap in time on different streams race; calls on one stream are ordered. */
size_t mad_gn_coef_static_max_floats(void);
int mad_gn_coef_static(const void* x1, int C1, const void* x2, int C2, int dtype, const float* gamma,
                       const float* beta, const float* ada, int ada_stride, float* coef, int B, int HW,
                       int groups, float eps, void* stream);

/* y = act(x*a + d) over the concatenation of x1/x2, written as one (B,H,W,C1+C2) tensor
 * (or (B,H/2,W/2,C) with the 2x2 mean of the activated values when `pool` is set: the `down`
 * resample of model.py:155-161 applied after norm+SiLU).  act: 0 none, 1 SiLU. */
int mad_gn_apply(const void* x1, int C1, const void* x2, int C2, const float* coef, void* y,
                 int dtype, int B, int H, int W, int act, int pool, void* stream);

/* mad_gn_apply (no pooling) followed by dropout of the result, in one pass: element e of y (its linear index)
 * is kept with probability 1-p and scaled by 1/(1-p), decided by a counter hash of (seed, e) -- the mask
 * mad_dropout and the mad_gn_bwd_* kernels rebuild from the same seed.
 * silu(norm1(...)) followed by F.dropout in UNetBlock.forward (model.py:346-352). */
int mad_gn_apply_dropout(const void* x1, int C1, const void* x2, int C2, const float* coef, void* y,
                         int dtype, int B, int H, int W, int act, unsigned long long seed, float p,
                         void* stream);

/* Self-attention over L = H*W positions, head_dim 64: qkv is NHWC (B,L,3*C) with channel
 * head*192 + {0..63 q, 64..127 k, 128..191 v}; out is (B,L,C) with channel head*64 + d.
 * softmax(q.k/8) in fp32.  Replaces model.py:357-375 / AttentionOp (model.py:199-212). */
int mad_attention(const void* qkv, void* out, int dtype, int B, int L, int heads, void* stream);

/* ------------------------------------------------------------------ training step --------- */
/* The reference trains through torch autograd (main.py:239-254); these are the gradient kernels of the
 * same operators.  Buffers named `workspace` are caller-owned device scratch of at least the size the
 * matching *_workspace() query returns. */

/* Weight gradient of a stride-1 convolution (ksize 1 or 3, pad ksize/2):
 *   dw[co][ci][ky][kx] (+)= sum_{b,y,x} dy[b,y,x,co] * x[b, y+ky-pad, x+kx-pad, ci]
 * x NHWC (B,H,W,Cin) -- or (B,H/2,W/2,Cin) read through a nearest x2 when `upsample` -- and dy NHWC
 * (B,H,W,Cout) in `dtype`; dw fp32 in the (Cout,Cin,k,k) layout of the master weights.  bf16 with Cin a
 * multiple of 32, Cout of 8 and (H % 8 == 0, W % 16 == 0, or H = W = 8) runs on MFMA; anything else on a
 * generic vector-ALU kernel.  Partial sums are added in a fixed order (deterministic).
 * Autograd of Conv2d.forward (model.py:135-166) with respect to `weight`. */
size_t mad_conv_wgrad_workspace(int dtype, int B, int H, int W, int Cin, int Cout, int ksize);
int mad_conv_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t workspace_bytes,
                   int dtype, int B, int H, int W, int Cin, int Cout, int ksize, int upsample,
                   int accumulate, void* stream);

/* mad_conv_wgrad that also produces the bias gradient dbias[co] (+)= sum_{b,y,x} dy[b,y,x,co] (fp32 (Cout,),
 * may be null) from the same pass over dy: on the MFMA path it is the weight gradient of an input channel of
 * ones, elsewhere a column sum.  Same workspace as mad_conv_wgrad.
 * Autograd of Conv2d.forward (model.py:135-166) with respect to `weight` and `bias`. */
int mad_conv_wgrad_bias(const void* x, const void* dy, float* dw, float* dbias, void* workspace,
                        size_t workspace_bytes, int dtype, int B, int H, int W, int Cin, int Cout, int ksize,
                        int upsample, int accumulate, int bias_accumulate, void* stream);

/* dbias[c] (+)= sum over rows of dy (rows, C): the bias gradient of a convolution or linear layer
 * (model.py:164-165, 62-63). */
size_t mad_bias_grad_workspace(int C);
int mad_bias_grad(const void* dy, int dtype, long rows, int C, float* dbias, void* workspace,
                  size_t workspace_bytes, int accumulate, void* stream);

/* The 3x3 data-gradient convolution (mad_conv3x3_fused without loader transform, bias or residual: x NHWC
 * (B,H,W,Cin), w (Cout,3,3,Cin) = the layer's weights transposed with flipped taps) whose output y (B,H,W,Cout)
 * is the gradient reaching h = act(GroupNorm([gx1, gx2])) -- and, from the same launch, the partial sums
 * mad_gn_bwd_partial would compute from y: gsum (B, (H/16)*(W/16), Cout, 2), one row per 16x16 output tile
 * (pass that tile count as nsplit_g to mad_gn_bwd_finalize).  gx1 (B,H,W,gC1), gx2 (B,H,W,gC2), gC1 + gC2 = Cout;
 * gcoef, act, drop_seed, drop_p as in mad_gn_bwd_partial.  Shapes as mad_conv3x3_fused.
 * Autograd of Conv2d.forward w.r.t. its input followed by GroupNorm/silu backward (model.py:135-166, 341-348). */
int mad_conv3x3_gn_bwd_sums(const void* x, int Cin, const void* w, void* y, float* gsum, const void* gx1, int gC1,
                            const void* gx2, int gC2, const float* gcoef, int act, unsigned long long drop_seed,
                            float drop_p, int dtype, int B, int H, int W, int Cout, void* stream);

/* Backward of h = act(GroupNorm(x)*(1+scale)+shift), in the three passes of the forward.  x is the
 * concatenation of x1 (C1) and x2 (C2), NHWC (B,H,W,.); dy is the gradient with respect to h over all
 * C1+C2 channels; coef is the forward's (a, d) table.  A gradient tensor's resolution mode: 0 = (H,W);
 * 1 = (H/2,W/2), each pixel takes a quarter of its cell (h was 2x2-averaged afterwards);
 * 2 = (2H,2W), each pixel takes the sum of its 2x2 block (h was upsampled afterwards).
 *   partial:  gsum[b, split, c, 0/1] = sum over the split's pixels of g, g*x, with g = dy*act'(x*a+d)
 *   finalize: from the forward's statistics partials (stat1/stat2 as in mad_gn_finalize) and gsum:
 *             bcoef (B, C, 4) = (p, q, r, 0) with dx = g*p + x*q + r; dgb (B, 2C) = per-sample
 *             [dgamma | dbeta] contributions (sum over b with mad_bias_grad); dada (may be NULL) laid out
 *             like ada: dada[b*ada_stride + c] = dscale, [.. + C + c] = dshift
 *   apply:    dx = g*p + x*q + r (+ add1 + add2, gradients reaching x by other branches, C1+C2 channels,
 *             each with its own resolution mode), written as dx1 (B,H,W,C1) and dx2 (B,H,W,C2).
 * drop_seed / drop_p: the dropout mad_gn_apply_dropout applied to h in the forward (p = 0: none); dy is then
 * masked and scaled the same way before anything else (dy_mode 0 only).
 * Autograd of GroupNorm.forward + silu + addcmul (model.py:182-190, 341-348). */
int mad_gn_bwd_partial(const void* x1, int C1, const void* x2, int C2, const void* dy, int dy_mode,
                       const float* coef, int act, unsigned long long drop_seed, float drop_p, int dtype,
                       float* gsum, int B, int H, int W, int nsplit, void* stream);
int mad_gn_bwd_finalize(const float* stat1, int C1, int nsplit1, const float* stat2, int C2, int nsplit2,
                        const float* gsum, int nsplit_g, const float* gamma, const float* beta,
                        const float* ada, int ada_stride, float* bcoef, float* dgb, float* dada, int B,
                        int groups, int HW, float eps, void* stream);
/* dgamma[c], dbeta[c] (+)= sum_b dgb[b][c], dgb[b][C + c]: the batch sum of finalize's per-image rows, written
 * straight to their (separate) destinations. */
int mad_gn_affine_grad(const float* dgb, int B, int C, float* dgamma, float* dbeta, int accumulate, void* stream);
int mad_gn_bwd_apply(const void* x1, int C1, const void* x2, int C2, const void* dy, int dy_mode,
                     const float* coef, int act, unsigned long long drop_seed, float drop_p,
                     const float* bcoef, const void* add1, int add1_mode,
                     const void* add2, int add2_mode, void* dx1, void* dx2, int dtype, int B, int H,
                     int W, void* stream);

/* mad_attention that also writes lse (B, heads, L) fp32: the log-sum-exp of each query's scaled scores,
 * which the backward needs. */
int mad_attention_lse(const void* qkv, void* out, float* lse, int dtype, int B, int L, int heads,
                      void* stream);

/* Backward of mad_attention (bf16 on MFMA; fp32 on a simple one-wave-per-row kernel): qkv, out and lse as the forward took / wrote them, dout (B,L,C)
 * the gradient of out; writes dqkv (B,L,3C) in the layout of qkv and uses dsum (B, heads, L) fp32 as
 * scratch (dout . out per query).  The probabilities are rebuilt tile by tile from lse.
 * Autograd of the attention of model.py:357-375 / AttentionOp.backward (model.py:214-229). */
int mad_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* dsum,
                      void* dqkv, int dtype, int B, int L, int heads, void* stream);

/* Working copies of the convolution weights of a training step, all layers in one launch: from the fp32
 * masters (Cout,Cin,k,k) inside `master` to the forward pack (Cout,k,k,Cin) inside `fwd` and the data-gradient
 * pack (Cin,k,k,Cout) with flipped taps inside `dgrad`, both in `dtype`.  `table` is a device array of 8 longs per
 * layer: master offset, forward offset, gradient offset (elements), Cout, Cin, k (1 or 3), first tile, 0; a layer
 * has ceil(Cout/32)*ceil(Cin/32) tiles and `total_tiles` is their sum.  The k <= 3 only.
 * Replaces the per-layer `weight.to(dtype)` of Conv2d.forward (model.py:142-143) and the transposed
 * weights autograd's conv backward builds. */
int mad_pack_conv_weights(const float* master, void* fwd, void* dgrad, const long* table, int nconv,
                          long total_tiles, int dtype, void* stream);

/* mad_conv_wgrad for the first and last layer, whose image-side operand is NCHW fp32: each operand
 * has its own dtype (MAD_BF16 / MAD_F32) and layout flag (1 = NCHW, 0 = NHWC).  Generic vector-ALU
 * kernel.  Autograd of enc.{R}x{R}_conv and out_conv (model.py:506-508, 549-551) w.r.t. `weight`. */
size_t mad_conv_wgrad_edge_workspace(int B, int H, int W, int Cin, int Cout, int ksize);
int mad_conv_wgrad_edge(const void* x, int x_dtype, int x_nchw, const void* dy, int dy_dtype, int dy_nchw,
                        float* dw, void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin,
                        int Cout, int ksize, int accumulate, void* stream);

/* Backward of mad_linear_f32 (y = act_in(xpre) @ W^T + bias), fp32:
 *   input : dx[r,k] (+)= (sum_n dy[r,n]*W[n,k]) * (act_in ? silu'(xpre[r,k]) : 1)   (xpre may be NULL without act_in)
 *   weight: dW[n,k]   = sum_r dy[r,n] * act_in(xpre[r,k])        (bias: mad_bias_grad over dy)
 * Autograd of Linear.forward (model.py:60-64). */
size_t mad_linear_bwd_input_workspace(int rows, int K, int N);
int mad_linear_bwd_input(const float* dy, const float* W, const float* xpre, float* dx, void* workspace,
                         size_t workspace_bytes, int rows, int K, int N, int act_in, int accumulate,
                         void* stream);
int mad_linear_bwd_weight(const float* dy, const float* xpre, float* dW, int rows, int K, int N,
                          int act_in, void* stream);

/* y[i] = keep(seed, i) ? x[i]/(1-p) : 0 with a counter-based keep decision, so the backward applies the
 * same mask to the gradient by calling it again with the same seed (in place allowed).
 * F.dropout in UNetBlock.forward (model.py:350-352); the random stream differs from torch's. */
int mad_dropout(const void* x, void* y, int dtype, size_t n, unsigned long long seed, float p,
                void* stream);

/* grad = 2*(pred-target)/n and partial[j] = the j-th workgroup's sum of (pred-target)^2 (loss =
 * sum(partial)/n), fp32.  F.mse_loss and its gradient (main.py:243). */
int mad_mse_loss_grad(const float* pred, const float* target, float* grad, float* partial, int npartial,
                      size_t n, void* stream);

/* One fused pass over flat fp32 buffers: g = nan_to_num(grad*grad_scale, 0, 1e5, -1e5) (main.py:249-251);
 * AdamW with decoupled weight decay as torch.optim.AdamW at 1-based `step` (main.py:173-178, 252);
 * ema -= (1-ema_decay)*(ema-param) when ema is not NULL (EMAModel.step, main.py:256-257). */
int mad_adamw_step(float* param, const float* grad, float* m, float* v, float* ema, size_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                   float ema_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAD_HIP_H */
