/* C ABI of libried_gan.so: the MI355X (gfx950) kernels under the RIED-GAN hot path.
 *
 * The reference (kevinmfreire/REID-GAN) is pure Python on torch.nn; it has no FFI of its own.
 * Each entry point below therefore replaces the stock torch op(s) a reference line invokes; the
 * citation names that line (paths relative to the reference repo).  A binding is a plain
 * ctypes / cffi declaration of these prototypes (reid_gan_amd/_lib.py generates one from this file).
 *
 * Conventions
 *  - Every function returns 0 on success, non-zero on failure; ried_last_error() then holds a
 *    message (thread local).  Arguments are validated before any launch.
 *  - All pointers are device pointers owned by the caller unless named *_host.  The library never
 *    allocates, frees or synchronises; every launch goes to the given stream, so calls may be
 *    captured into a HIP graph.
 *  - bf16 activations are NHWC.  A "view" (ptr, ld, C) addresses channels [0, C) of pixel p at
 *    ptr + p*ld; C and ld are multiples of 8 and ptr is 16-byte aligned.  Views let a producer write
 *    a channel slice of a concatenation buffer directly.
 *  - fp32 images (the CT slices, the generator output, SSIM inputs) are [nimg][H][W].
 */
#ifndef RIED_GAN_H
#define RIED_GAN_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* ried_stream_t; /* == hipStream_t */

const char* ried_last_error(void);
int ried_abi_version(void);
int ried_device_cus(void);
/* Diagnostics, no counterpart in the reference: on SIGABRT / SIGSEGV / SIGBUS / SIGFPE / SIGILL write the native
 * frames of the thread that received the signal (glibc backtrace) to stderr - or to extra_fd when it is >= 0 - and to
 * the file log_path_host (NULL: none; opened for append at installation), then pass the signal on to the handler
 * that was installed before (Python's faulthandler, or the default action).  Idempotent. */
int ried_install_crash_backtrace(const char* log_path_host, int extra_fd);

/* ---- implicit-GEMM convolutions (bf16 in, fp32 accumulate, MFMA) -------------------------------
 * Conv2d forward (transposed = 0): replaces nn.Conv2d at src/rinet.py:186 (BasicConv2d), :137 (SNConv2d),
 * :100 (identity_downsample), :156-158 (attention projections), :265,360 (sample), and the VGG16 convs built
 * at src/loss.py:74.
 * transposed = 1: ConvTranspose2d forward (src/rinet.py:198) and the data gradient of a Conv2d.
 *   x: view [nimg][H][W] with Cin channels; w: packed bf16 [Cout][kh][kw][Cin]; y: view [nimg][Ho][Wo], Cout.
 *   bias: fp32 [Cout] or NULL; act: 0 none, 1 ReLU, 2 LeakyReLU(0.2); stats: NULL, or fp32 [rows][2][Cout] that
 *   receives one partial (sum, sum of squares) row of the fp32 results per pixel tile, rows =
 *   ried_conv2d_stat_rows(...) (for train-mode BatchNorm; per-block rows instead of atomics, which serialise);
 *   accumulate: y += result; out_f32: y is fp32 instead of bf16; y2/ldy2: optional second, fp32 copy of the
 *   result (the perceptual-loss taps are compared in fp32) or NULL.
 *   mask/ldmask/mask_slope: optional view shaped like y; the final value (after accumulate) is multiplied by
 *   (mask > 0 ? 1 : mask_slope): the backward of the ReLU / LeakyReLU whose output `mask` is (autograd of
 *   src/rinet.py:139,190 and the VGG ReLUs), fused into the data gradient that produces the tensor.
 *   stats_post = 1: the stats rows hold column sums of the FINAL values (sum in [r][0][c]) - the bias gradient of
 *   the layer below (src/rinet.py:137) - instead of the raw results.
 *   stats_post = 2: the raw sums are ADDED (float atomics) into row (tile mod ried_bn_fold_rows()) of a buffer the
 *   caller zeroed, so that min(rows, ried_bn_fold_rows()) rows hold the totals: the form ried_bn_apply_train reads.
 *   stats_post = 3: the raw sums are ADDED into row (image index) of a zeroed [nimg][2][Cout] buffer: statistics per
 *   image, for the evaluation loop of src/test.py:107-131, whose batch-1 loader makes train-mode BatchNorm normalise
 *   every slice on its own (follow with ried_bn_finalize_images / ried_bn_apply_images).  Maps must consist of whole
 *   pixel tiles (any map of at least 16 x 16 with sides divisible by 16 does). */
int ried_conv2d(const void* x, long long ldx, const void* w, void* y, long long ldy, int nimg, int H, int W, int Cin,
                int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int transposed,
                const float* bias, int act, float* stats, int accumulate, int out_f32, float* y2, long long ldy2,
                const void* mask, long long ldmask, float mask_slope, int stats_post, ried_stream_t stream);
/* ried_conv2d (bf16 output) with the activation mask given as sign bytes, as ried_conv_c1_fwd_signs writes them:
 * signs[p][ldsigns bytes], bit q of byte c set where channel 8c + q of the forward output was positive.  The backward
 * of LeakyReLU (src/rinet.py:139) needs no more, and the 64-channel activation of the discriminator's first layer
 * (src/rinet.py:20) is the largest tensor of the step: 1 byte instead of 16 per 8-channel chunk. */
int ried_conv2d_signs(const void* x, long long ldx, const void* w, void* y, long long ldy, int nimg, int H, int W,
                      int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int transposed,
                      const float* bias, int act, float* stats, int accumulate, const unsigned char* signs,
                      long long ldsigns, float mask_slope, int stats_post, ried_stream_t stream);
/* Backward of the discriminator's second layer into its first (autograd of src/rinet.py:20-21) when only parameter
 * gradients are wanted (the discriminator's own updates, src/train.py:156-165): the data gradient of the 3x3 stride-2
 * convolution (dy: view [nimg][H][W] with Cin channels; wt: its packed transposed weights [64][3][3][Cin]) is masked
 * with the first layer's LeakyReLU (mask: that layer's output as a bf16 view with ldmask elements per pixel, or its
 * sign bytes with ldmask bytes per pixel when mask_bits = 1), its column sums go to stats (rows as ried_conv2d with
 * stats_post = 1: the first layer's bias gradient), and its product with the nine taps of the first layer's fp32 input
 * ximg [nimg][2H][2W] (used as ximg * in_scale + in_shift, zero padded) is ADDED to dw [64][9], the first layer's
 * weight gradient.  The masked gradient itself - the largest tensor of the step - is never written.  H, W multiples
 * of 16; scratch: fp32 [ried_conv2d_s2_dgrad_c1fold_scratch_floats(nimg, H, W)]. */
int ried_conv2d_s2_dgrad_c1fold(const void* dy, long long lddy, const void* wt, int nimg, int H, int W, int Cin,
                                const void* mask, long long ldmask, int mask_bits, float mask_slope, float* stats,
                                const float* ximg, float in_scale, float in_shift, float* dw, float* scratch,
                                ried_stream_t stream);
long long ried_conv2d_s2_dgrad_c1fold_scratch_floats(int nimg, int H, int W);
/* out[c] += sum over r < nrows of part[r][c], c < width: partial rows that blocks wrote without atomics */
int ried_rows_add(const float* part, int nrows, int width, float* out, ried_stream_t stream);
/* BasicConv2d / BasicDeconv2d (src/rinet.py:183-205) up to the BatchNorm statistics: y = raw bf16 conv result,
 * stats = partial rows as in ried_conv2d, and the last block to finish reduces them into save = (mean,
 * 1/sqrt(var+eps)) [2][Cout] and updates running_mean / running_var / num_batches_tracked (torch train-mode
 * semantics; any of the three may be NULL).  counter: one zero-initialised unsigned per layer (left at zero). */
int ried_conv2d_bn(const void* x, long long ldx, const void* w, void* y, long long ldy, int nimg, int H, int W,
                   int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int transposed,
                   float* stats, unsigned* counter, float eps, float momentum, float* save, float* running_mean,
                   float* running_var, long long* num_batches_tracked, ried_stream_t stream);
/* A tap layer of the perceptual loss on the prediction images (src/loss.py:179-189, VGG16 conv3x3 + bias + ReLU):
 * y as ried_conv2d would write it, and in the same pass d = y (fp32, before rounding) - target (the same tap of the
 * target images, fp32 [npix][ld_target]); stats rows (as ried_conv2d) receive the column sums of d and d*d,
 * gout [npix][ld_gout] (bf16) = gscale * d.  Replaces F.mse_loss on the two feature maps and its backward. */
int ried_conv2d_tap_mse(const void* x, long long ldx, const void* w, const float* bias, void* y, long long ldy,
                        int nimg, int H, int W, int Cin, int Cout, const float* target, long long ld_target, void* gout,
                        long long ld_gout, float gscale, float* stats, ried_stream_t stream);
/* The same with the target's tap given as its stored bf16 activations (a view of the target half of the feature map,
 * ld_target elements per pixel) instead of an fp32 copy: the target pass writes no fp32 taps and this pass reads half
 * the bytes; d is taken against the rounded target values.  All views 16-byte aligned with pitches divisible by 8. */
int ried_conv2d_tap_mse_bf16(const void* x, long long ldx, const void* w, const float* bias, void* y, long long ldy,
                             int nimg, int H, int W, int Cin, int Cout, const void* target, long long ld_target,
                             void* gout, long long ld_gout, float gscale, float* stats, ried_stream_t stream);
/* out[0] += scale * sum over rows and channels of stats[r][which][c] */
int ried_stat_rows_total(const float* stats, int nrows, int C, int which, float scale, float* out,
                         ried_stream_t stream);
/* out[c] += sum over rows of stats[r][0][c] (stats: [nrows][2][C], as written by ried_conv2d) */
int ried_stat_rows_sum(const float* stats, int nrows, int C, float* out, ried_stream_t stream);

int ried_conv2d_stat_rows(int nimg, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int sh, int sw,
                          int transposed);
/* tuning / test hook, no counterpart in the reference: force the (Cout tile, pixel tile, LDS stages) of later
 * ried_conv2d calls; (0, 0, 0) restores the built-in choice.  bm = -1: the general kernel with its own tile
 * choice (no halo-tile kernels); (0, 1, 0): the halo-tile kernels even where their block count is below a CU's
 * worth; nst = -1: no LDS-staged epilogue. */
int ried_conv2d_force_tile(int bm, int bn, int nst);

/* Weight-gradient partial sums (autograd of the convs above).  g is the tensor seen through the kernel window
 * (Conv2d: the layer input; ConvTranspose2d: the output gradient), p the tensor indexed by output pixel.
 * slab: fp32 [splitk][kh*kw][Cp][Cg].  Follow with ried_wgrad_reduce. */
int ried_conv2d_wgrad(const void* g, long long ldg, int Cg, int nimg, int H, int W, const void* p, long long ldp,
                      int Cp, int Ho, int Wo, int kh, int kw, int sh, int sw, int ph, int pw, float* slab, int splitk,
                      ried_stream_t stream);
/* Job-table form: the weight gradients of many layers in one launch per tile shape (the reference launches one
 * cuDNN weight-gradient per layer from autograd).  ried_conv2d_wgrad_make_job fills a HOST record of
 * ried_conv2d_wgrad_job_bytes() bytes for one layer (arguments as ried_conv2d_wgrad) and reports tile[2] (its
 * tile shape) and its block count; records of one tile shape, copied to the device back to back, are launched by
 * ried_conv2d_wgrad_jobs with map[b] = (job, block of that job) for b < nblocks. */
int ried_conv2d_wgrad_job_bytes(void);
int ried_conv2d_wgrad_make_job(void* job_host, const void* g, long long ldg, int Cg, int nimg, int H, int W,
                               const void* p, long long ldp, int Cp, int Ho, int Wo, int kh, int kw, int sh, int sw,
                               int ph, int pw, float* slab, int splitk, int* tile, int* nblocks);
int ried_conv2d_wgrad_jobs(const void* jobs, const int* map, int nblocks, int bm, int bn, ried_stream_t stream);
/* Job-table form of ried_conv2d: convolutions that do not depend on each other - the same-depth BasicConv2d layers
 * of an Inception block's branches (src/rinet.py:237-252, 325-347, 419-445), forward or data gradient - as ONE
 * launch of the general kernel with 64 x 64 tiles, instead of one launch per branch.  ried_conv2d_make_job fills a
 * HOST record of ried_conv2d_job_bytes() bytes (arguments as ried_conv2d; statistics only in the folded form,
 * stats_post = 2) and reports the job's block count; records are copied to the device back to back and launched
 * with ried_conv2d_jobs, map[2b], map[2b+1] = (job, block of that job) for b < nblocks. */
/* ried_conv2d for a consumer of a BasicConv2d layer (src/rinet.py:183-205, chains such as rinet.py:325-347): x is
 * that layer's RAW convolution output, and the kernel applies the layer's train-mode BatchNorm (xf_save = [2][Cin]:
 * mean, 1/sqrt(var + eps); xf_gamma, xf_beta) and activation (xf_act: 0 none, 1 ReLU) to the operand on its way
 * through LDS: the normalised tensor is never written or read.  General kernel only, Cin <= 1024. */
int ried_conv2d_xf(const void* x, long long ldx, const void* w, void* y, long long ldy, int nimg, int H, int W, int Cin,
                   int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int transposed,
                   const float* bias, int act, float* stats, int accumulate, int stats_post, const float* xf_save,
                   const float* xf_gamma, const float* xf_beta, int xf_act, ried_stream_t stream);
int ried_conv2d_job_bytes(void);
int ried_conv2d_make_job(void* job_host, const void* x, long long ldx, const void* w, void* y, long long ldy, int nimg,
                         int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                         int transposed, const float* bias, int act, float* stats, int accumulate, int stats_post,
                         int* nblocks);
int ried_conv2d_jobs(const void* jobs, const int* map, int nblocks, ried_stream_t stream);
/* Reporting aid (no counterpart in the reference): which kernel a ried_conv2d call with these arguments runs, without
 * launching it.  which[4] = {kind, bm, bn or tile side, LDS stages (+ 10: two K groups per block)}; kind
 * 0 = conv_glds_kernel<bm, bn, stages>, 1 = conv_halo_kernel<bm, side, stages>, 2 = conv_halo_s2t_kernel<side>,
 * 3 = conv_halo_d4_kernel<bm>. */
int ried_conv2d_which(const void* x, long long ldx, const void* w, void* y, long long ldy, int nimg, int H, int W,
                      int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int transposed,
                      const float* bias, int act, float* stats, int accumulate, int out_f32, float* y2, long long ldy2,
                      const void* mask, long long ldmask, float mask_slope, int stats_post, int* which);
int ried_conv2d_wgrad_tiles(int Cg, int Cp); /* blocks per (tap, split) of ried_conv2d_wgrad: for choosing splitk */
/* Taps of the kernel window one block of ried_conv2d_wgrad computes (layers with at most 64 gathered channels and
 * enough output pixels npix = nimg * Ho * Wo: 3 of a 3 x 3 window, 2 of a 4 x 4 window; else 1): the grid has
 * tiles * kh * kw / taps_per_block blocks per split.  For choosing splitk. */
int ried_conv2d_wgrad_taps_per_block(int Cg, int Cp, int kh, int kw, long long npix);
/* Tuning hook (no counterpart in the reference): on = 0 makes every later weight gradient one tap per block. */
int ried_conv2d_wgrad_row_blocks(int on);
/* out[tap*st + p*sp + g*sg] (+)= scale * sum over splits: scatters into the (Cout,Cin,kh,kw) or (Cin,Cout,kh,kw)
 * fp32 master gradient. */
int ried_wgrad_reduce(const float* slab, int splitk, int taps, int Cp, int Cg, float* out, long long st, long long sp,
                      long long sg, float scale, int accumulate, ried_stream_t stream);
/* Job-table forms (one launch for all layers of a network; the reference has one torch op per layer).  A job is a
 * row of 64-bit words holding pointers and integers; blockmap[b] = (job, chunk): for pack jobs block b handles
 * elements [2048*chunk, 2048*(chunk+1)) of its job, for reduce jobs the output row p = chunk (Cp blocks per job).
 *   pack job (13 words): src, dst, sigma|0, N, R, S, C, sn, sr, ss, sc, dst_pitch, dst_f32 - as ried_pack_weight,
 *     with dst[n*dst_pitch + (r*S+s)*C + c]; dst is fp32 instead of bf16 when dst_f32 == 1.
 *     dst_f32 == 2, a tile job: words src, dst1, sigma, A, B, T, dst2 - the master is [A][B][T] (T <= 16; A, B even)
 *     and is written as dst1 [A][T][B] and dst2 [B][T][A] (bf16, either may be 0): both packed layouts of a
 *     convolution weight from one read of it; blockmap chunk = tile (a / 32) * ceil(B / 32) + b / 32.
 *   reduce job (12 words): slab, out, splitk, split_stride, taps, Cp, Cg, st, sp, sg, accumulate, scale (the bits of
 *     an fp32 number; 0 means 1.0) - as ried_wgrad_reduce with slab[k*split_stride + (tap*Cp + p)*Cg + g]. */
int ried_pack_jobs(const long long* jobs, const int* blockmap, int nblocks, ried_stream_t stream);
int ried_wgrad_reduce_jobs(const long long* jobs, const int* blockmap, int nblocks, ried_stream_t stream);

/* Batched out[b][n][m] = alpha * sum_k A[b][m][k] * B[b][n][k]; a_tr / b_tr: operand stored [k][row].
 * Replaces torch.bmm at src/rinet.py:173,177 and their autograd.  gamma/resid: out = gamma[0]*value + resid. */
int ried_bgemm(const void* A, long long lda, long long bsa, int a_tr, const void* B, long long ldb, long long bsb,
               int b_tr, int M, int N, int K, int batch, void* out, long long ldc, long long bsc, int out_f32,
               float alpha, const float* gamma, const void* resid, long long ld_resid, long long bs_resid,
               int accumulate, ried_stream_t stream);

/* fp32 master weight with element strides (sn, sr, ss, sc) -> bf16 [N][R][S][C], divided by sigma[0] if given
 * (the W / sigma of spectral norm, torch/nn/utils/spectral_norm.py:113). */
int ried_pack_weight(const float* src, void* dst, int N, int R, int S, int C, long long sn, long long sr, long long ss,
                     long long sc, const float* sigma, ried_stream_t stream);

/* ---- BatchNorm2d, train mode (src/rinet.py:187,199; eps 1e-5, momentum 0.1) ------------------------------------ */
/* stats [nrows][2][C] (partial sum, sum sq rows; count values per channel in all) -> save [2][C] (mean, invstd);
 * updates running stats (unbiased variance) and num_batches_tracked when given.  fold: NULL, or fp32 scratch
 * [ried_bn_finalize_fold_floats(C)] that lets very many rows (512x512 inference) be reduced in two wide steps.
 * ldstat: channels per statistics row when the layer is a channel slice of a wider convolution (stats then points at
 * the slice's first channel; 0: C). */
int ried_bn_finalize(const float* stats, int nrows, int C, int ldstat, long long count, float eps, float momentum,
                     float* save, float* running_mean, float* running_var, long long* num_batches_tracked,
                     float* fold, ried_stream_t stream);
long long ried_bn_finalize_fold_floats(int C);
int ried_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* save,
                       ried_stream_t stream);
/* y = act(gamma*(x-mean)*invstd + beta) [+ y]   (the residual add of src/rinet.py:250,345,443 is the accumulate).
 * The `relu` argument of the BatchNorm functions is an activation code: 0 none, 1 ReLU, 2 LeakyReLU(0.2),
 * 3 Sigmoid (the last two: src/networks.py:533-542). */
int ried_bn_apply(const void* x, long long ldx, void* y, long long ldy, long long npix, int C, const float* save,
                  const float* gamma, const float* beta, int relu, int accumulate, ried_stream_t stream);
/* Small layers, one launch fewer (BatchNorm2d train mode, src/rinet.py:187-193,199-205): ried_bn_apply_train =
 * ried_bn_finalize + ried_bn_apply on stats [nrows][2][C] (nrows <= ried_bn_fold_rows(), written by ried_conv2d with
 * stats_post = 2; count = npix); the blocks of the apply pass sum the few rows themselves.  The float atomics of
 * the folding make the sums depend on the order the conv's blocks finish (last-bit differences between runs). */
int ried_bn_fold_rows(void);
int ried_bn_apply_train(const void* x, long long ldx, void* y, long long ldy, long long npix, int C,
                        const float* stats, int nrows, int ldstat, float eps, float momentum, float* save,
                        float* running_mean, float* running_var, long long* num_batches_tracked, const float* gamma,
                        const float* beta, int relu, int accumulate, ried_stream_t stream);
/* Job-table form of ried_bn_apply_train for njobs layers that share the pixel count (one launch): 16 64-bit words
 * per job: x, ldx, y, ldy, C, stats, nrows, ldstat, save, running_mean, running_var, num_batches_tracked, gamma, beta,
 * act, accumulate.  maxC / minC: the largest / smallest C among the jobs. */
int ried_bn_apply_train_jobs(const long long* jobs, int njobs, long long npix, int maxC, int minC, float eps,
                             float momentum, ried_stream_t stream);
/* ldstat: channels per statistics row, 0 = C; larger when the layer is a channel slice of a convolution that served
 * several layers at once (stats then points at the slice's first channel: rows [nrows][2][ldstat]). */
/* Per-image train-mode statistics: the evaluation loop of src/test.py:107-131 feeds one slice at a time, so its
 * BatchNorm layers normalise every slice on its own.  stats: [nimg * rows_per_img][2][C] partial rows, the rows of an
 * image consecutive (rows_per_img = 1 after ried_conv2d with stats_post = 3); count = pixels per image;
 * save [nimg][2][C] = (mean, 1/sqrt(var + eps)).  ried_bn_apply_images = ried_bn_apply with that per-image save on
 * views of nimg images of npix_per_img pixels.  Running statistics are not touched.  ldstat: channels per statistics
 * row when the layer is a channel slice of a wider convolution (0: C). */
int ried_bn_finalize_images(const float* stats, int rows_per_img, int nimg, int C, int ldstat, long long count,
                            float eps, float* save, ried_stream_t stream);
int ried_bn_apply_images(const void* x, long long ldx, void* y, long long ldy, long long npix_per_img, int nimg, int C,
                         const float* save, const float* gamma, const float* beta, int relu, int accumulate,
                         ried_stream_t stream);
/* dx = d(loss)/d(conv output) from dy = d(loss)/d(bn+relu output); scratch: fp32
 * [ried_bn_backward_scratch_floats(C)] (need not be zeroed); dgamma/dbeta may be NULL; counter: NULL, or one
 * zero-initialised unsigned per layer (left at zero) - the last block of the reduction then finishes the sums.
 * folded: NULL, or ZEROED fp32 [ried_bn_backward_folded_floats(C)] owned by the layer (left non-zero): the
 * reduction's blocks add into a few rows of it with float atomics and the apply pass adds those up itself - two
 * launches instead of three, sums that depend on the order the blocks finish in; scratch is unused.  folded AND
 * counter (then: a zeroed word that is LEFT NON-ZERO, e.g. a word of the same zeroed arena as folded): ONE launch -
 * the blocks add their sums into the rows, meet at a grid barrier and apply; for layers with
 * ried_bn_backward_fused_ok(npix, C, 1) only (a grid that is resident at once), an argument error otherwise. */
int ried_bn_backward(const void* dy, long long lddy, const void* x, long long ldx, void* dx, long long lddx,
                     long long npix, int C, const float* save, const float* gamma, const float* beta, int relu,
                     float* scratch, float* dgamma, float* dbeta, int accumulate, unsigned* counter, float* folded,
                     ried_stream_t stream);
/* The two launches of the folded form of ried_bn_backward as calls of their own, for synchronised BatchNorm under
 * data parallelism (the option torch.nn.SyncBatchNorm would add to the DataParallel the reference leaves commented
 * out at src/train.py:95-96): reduce adds the sums of g and g * x-hat into the ZEROED rows `folded`; the caller
 * averages `folded` over the ranks; apply writes dx from the rows (it divides by npix, so rank-averaged local sums
 * give the global-batch formula) and, when dgamma / dbeta are not NULL, the rows' totals as parameter gradients. */
int ried_bn_backward_reduce(const void* dy, long long lddy, const void* x, long long ldx, long long npix, int C,
                            const float* save, const float* gamma, const float* beta, int relu, float* folded,
                            ried_stream_t stream);
int ried_bn_backward_apply(const void* dy, long long lddy, const void* x, long long ldx, void* dx, long long lddx,
                           long long npix, int C, const float* save, const float* gamma, const float* beta, int relu,
                           const float* folded, float* dgamma, float* dbeta, int accumulate, ried_stream_t stream);
long long ried_bn_backward_scratch_floats(int C);
long long ried_bn_backward_folded_floats(int C);
/* 1 when njobs layers of npix pixels and at most maxC channels each can take the one-launch form (0: too large). */
int ried_bn_backward_fused_ok(long long npix, int maxC, int njobs);
/* *out = blocks that gave up waiting at a grid barrier of the one-launch forms since the library was loaded: 0
 * unless a launch could not get all its blocks resident (its results are then wrong; the wait is bounded so that
 * the grid always drains).  Synchronises with the device.  No counterpart in the reference (a health check). */
int ried_barrier_timeouts(unsigned* out);
/* Job-table form: ried_bn_backward for njobs layers that share the pixel count (the sibling 1x1 layers of an
 * Inception block, src/rinet.py:99-103,225-233) in three launches in all.  jobs: device table, 16 64-bit words per
 * job: dy, lddy, x, ldx, dx, lddx, C, save, gamma, beta, act, scratch, dgamma, dbeta, accumulate, folded (meanings
 * as the arguments of ried_bn_backward; every job its own scratch or folded rows).  maxC: the largest C among the
 * jobs; all_folded: 1 when every job carries folded rows (two launches in all); 2 when in addition the scratch
 * word of every job points at a zeroed barrier counter (left non-zero) and
 * ried_bn_backward_fused_ok(npix, maxC, njobs) holds: one launch in all. */
int ried_bn_backward_jobs(const long long* jobs, int njobs, long long npix, int maxC, int all_folded,
                          ried_stream_t stream);

/* ---- pooling / upsampling ------------------------------------------------------------------------------------- */
/* mode 0: MaxPool2d(3,2,1) (src/rinet.py:260,355); 1: MaxPool2d(2,2) (VGG16); 2: avg_pool2d(3,1,1) (:235,330,428) */
int ried_pool_fwd(const void* x, long long ldx, void* y, long long ldy, int nimg, int H, int W, int C, int mode,
                  ried_stream_t stream);
int ried_pool_bwd(const void* x, long long ldx, const void* dy, long long lddy, void* dx, long long lddx, int nimg,
                  int H, int W, int C, int mode, int accumulate, int relu_mask, ried_stream_t stream);
/* avg_pool2d(3,1,1) of the OUTPUT of a branch_pool 1x1 convolution (src/rinet.py:234,250-251; 320,344-345; 416,441-442:
 * pool and bias-free 1x1 convolution commute) with the BatchNorm partial statistics of the pooled tensor:
 * stats[ried_avgpool3_stats_rows(npix, C)][2][C], rows for ried_bn_finalize. */
int ried_avgpool3_stats_rows(long long npix, int C);
int ried_avgpool3_stats(const void* x, long long ldx, void* y, long long ldy, int nimg, int H, int W, int C,
                        float* stats, ried_stream_t stream);
/* ried_pool_bwd: dx (+)= pooled gradient (accumulate != 0 adds onto dx; max modes may overwrite instead);
 * relu_mask (max modes): the result is zeroed where x <= 0, the backward of the ReLU that produced x */ /* dx += */
/* nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (src/rinet.py:266,361) */
int ried_upsample2x_fwd(const void* x, long long ldx, void* y, long long ldy, int nimg, int H, int W, int C,
                        ried_stream_t stream);
int ried_upsample2x_bwd(const void* dy, long long lddy, void* dx, long long lddx, int nimg, int H, int W, int C,
                        ried_stream_t stream); /* dx += */

/* ---- elementwise / reductions --------------------------------------------------------------------------------- */
/* op 0 copy, 1 out+=a, 2 out=a*act'(y), 3 out+=a*act'(y), 4 out=s[0]*a+y (src/rinet.py:180), 5 out=s[0]*a;
 * act'(y) = y>0 ? 1 : slope, taken from the activation OUTPUT y (ReLU: slope 0; LeakyReLU: 0.2). */
int ried_ew(int op, const void* a, long long lda, const void* y, long long ldy, void* out, long long ldo,
            long long npix, int C, float slope, const float* s, ried_stream_t stream);
/* out[c] += sum_p a[p][c]; scratch: fp32 [ried_colsum_scratch_floats(C)] */
int ried_colsum(const void* a, long long lda, long long npix, int C, float* out, float* scratch, ried_stream_t stream);
long long ried_colsum_scratch_floats(int C);
/* ried_colsum without a workspace argument.  No counterpart in the reference: it is for callers that cannot
 * allocate device memory - code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  The
 * partial rows live in a 1 MiB device array built into the library (one copy per GPU, held until the process ends);
 * the call allocates and frees nothing.  C <= 1024; a larger C is refused with an error.  Same kernels and grid as
 * ried_colsum (results equal up to the order of that kernel's fp32 LDS atomics).  All calls on one GPU share that
 * array: calls that may overlap on one GPU (different streams, different threads) race; serialise them.  The first
 * call on a device resolves the array's address (hipGetSymbolAddress); make it outside stream capture. */
int ried_colsum_static(const void* a, long long lda, long long npix, int C, float* out, ried_stream_t stream);
/* out[0] += scale * sum a*b (b NULL: sum a) */
int ried_dot(const void* a, long long lda, const void* b, long long ldb, long long npix, int C, float scale, float* out,
             ried_stream_t stream);
/* softmax(q^T k, dim=-1) of Self_Attn (src/rinet.py:171-174) in one pass: qkv bf16 [nimg][n][ld], q in channels
 * [0, d), k in [d, 2d); P bf16 [nimg][n][n].  Only the (n, d) for which ried_attn_scores_softmax_supported is 1. */
int ried_attn_scores_softmax_supported(int n, int d);
int ried_attn_scores_softmax(const void* qkv, long long ld, int nimg, int n, int d, void* P, ried_stream_t stream);
/* Self_Attn (src/rinet.py:162-181) fused, for the (n, d, C) ried_attn_fused_supported accepts ((256, 32, 256) and
 * (64, 64, 512): the two layers at 64 x 64 patches): energy, softmax and the value product in one kernel, nothing
 * n x n in HBM.  qkv: bf16 [nimg][n][ld] with q in channels [0, d), k in [d, 2d), v in [2d, 2d + C) (the output of
 * the fused query / key / value 1x1 convolution); x: the layer input [nimg][n] (view, ldx).
 * ried_attn_fwd: O [nimg][n] (view, ldO) = attention . v;  out = gamma[0] * O + x (src/rinet.py:180);
 *   lse fp32 [nimg][n] = log sum_j exp(energy[i][j]) (what the backward pass rebuilds the attention from).
 * ried_attn_bwd (autograd of the above): dout = d loss / d out; dqkv (view, ldg, all 2d + C channels written) =
 *   gradients w.r.t. q, k, v; dgamma[0] += sum dout * O (NULL: not wanted).  The residual path dx += dout is the
 *   caller's. */
int ried_attn_fused_supported(int n, int d, int C);
int ried_attn_fwd(const void* qkv, long long ld, const void* x, long long ldx, void* out, long long ldo, void* O,
                  long long ldO, float* lse, const float* gamma, int nimg, int n, int d, int C, ried_stream_t stream);
int ried_attn_bwd(const void* qkv, long long ld, void* dqkv, long long ldg, const void* dout, long long lddo,
                  const void* O, long long ldO, const float* lse, const float* gamma, float* dgamma, int nimg, int n,
                  int d, int C, ried_stream_t stream);
/* out = s ? s[0] * value : value over a bf16 view */
int ried_fill(void* out, long long ldo, long long npix, int C, const float* s, float value, ried_stream_t stream);
int ried_softmax_fwd(const float* S, void* P, long long rows, int n, ried_stream_t stream); /* src/rinet.py:174 */
int ried_softmax_bwd(const float* dP, const void* P, void* dS, long long rows, int n, ried_stream_t stream);
int ried_nchw_to_nhwc(const float* x, void* y, long long ldy, int nimg, int C, long long HW, ried_stream_t stream);
int ried_nhwc_to_nchw(const void* x, long long ldx, float* y, int nimg, int C, long long HW, ried_stream_t stream);

/* torch.optim.Adam step over a flat fp32 buffer (src/train.py:104-105,165,177).  step: device int32 counter of
 * completed steps, incremented by this call before use (so a captured graph replays correctly). */
int ried_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
              int* step, ried_stream_t stream);
/* out[0] += scale * sum_i f(in[i*stride]), i < n, f = identity (square = 0) or x^2 (square = 1); fp32.
 * torch.mean of DLoss / GLoss (src/loss.py:142,155), np.mean / np.std of compute_CNR (src/measure.py:44-45). */
int ried_sum_f32(const float* in, long long n, long long stride, float scale, int square, float* out,
                 ried_stream_t stream);

/* ---- one-input-channel 3x3 pad-1 convolutions on the fp32 image ----------------------------------------------- */
/* generator conv1 (src/rinet.py:55), discriminator layer 0 (:20), VGG16 layer 0 on cat([x,x,x]) (src/loss.py:181)
 * with the three input planes folded.  w: fp32 [Cout][9]; sigma: NULL or the spectral norm to divide by;
 * the input is used as x*in_scale + in_shift (normalize_, src/loss.py:20-22), zero padded after the affine. */
int ried_conv_c1_fwd(const float* x, const float* w, const float* sigma, const float* bias, void* y, long long ldy,
                     int nimg, int H, int W, int Cout, int stride, float in_scale, float in_shift, int act,
                     float* stats, ried_stream_t stream); /* stats: NULL or [ried_conv_c1_stat_rows(..)][2][Cout] */
/* The same, and signs[p][Cout/8] (8-byte aligned; one byte per 8-channel chunk, bit q = channel 8*chunk + q) says
 * which of the stored values are positive (for ried_conv2d_signs). */
int ried_conv_c1_fwd_signs(const float* x, const float* w, const float* sigma, const float* bias, void* y,
                           long long ldy, int nimg, int H, int W, int Cout, int stride, float in_scale, float in_shift,
                           int act, float* stats, unsigned char* signs, ried_stream_t stream);
int ried_conv_c1_stat_rows(int nimg, int H, int W, int Cout, int stride);
/* dw[Cout][9] += ; scratch: fp32 [ried_conv_c1_wgrad_scratch_floats(Cout)] */
int ried_conv_c1_wgrad(const float* x, const void* dy, long long lddy, int nimg, int H, int W, int Cout, int stride,
                       float in_scale, float in_shift, float* dw, float* scratch, ried_stream_t stream);
long long ried_conv_c1_wgrad_scratch_floats(int Cout);
int ried_conv_c1_dgrad(const void* dy, long long lddy, const float* w, const float* sigma, float* dx, int nimg, int H,
                       int W, int Cout, float scale, int accumulate, ried_stream_t stream); /* stride 1 */

/* generator out_conv 33->1 on cat(identity, h) (src/rinet.py:70,90): out = w[0]*x + sum_c w[1+c]*h[:,c].
 * bn_save [2][32] (mean, invstd) with gamma, beta [32], or all three NULL: with them h is the RAW output of deconv2
 * and relu(gamma*(h-mean)*invstd+beta) (src/rinet.py:199-205) is applied on the fly - the forward-only path, where
 * nothing else reads the normalised tensor. */
int ried_out_conv_fwd(const float* x, const void* h, long long ldh, const float* w, float* out, long long npix,
                      const float* bn_save, const float* gamma, const float* beta, int bn_images,
                      ried_stream_t stream); /* bn_images > 1: bn_save is [bn_images][2][32], one set per image */
int ried_out_conv_bwd(const float* x, const void* h, long long ldh, const float* w, const float* dout, void* dh,
                      long long lddh, float* dw, long long npix, ried_stream_t stream); /* dw[33] += */

/* ---- spectral norm (torch/nn/utils/spectral_norm.py:97-113 as used at src/rinet.py:138) ------------------------ */
/* Wm: weight_orig as fp32 [R][K].  power_iter = 1 updates u [R] and v [K] in place, then sigma[0] = u.(W v).
 * scratch: fp32 [K + R]. */
int ried_sn_sigma(const float* Wm, int R, int K, float* u, float* v, float* sigma, float* scratch, int power_iter,
                  float eps, ried_stream_t stream);
/* dW (+)= (dWsn - (sum(dWsn*W)/sigma) u v^T)/sigma ; scratch1: one fp32 */
int ried_sn_backward(const float* dWsn, const float* Wm, const float* u, const float* v, const float* sigma, int R,
                     int K, float* dW, float* scratch1, int accumulate, ried_stream_t stream);
/* Job-table forms: all spectral-norm layers of a network per launch.  job = 10 64-bit words: W, u, v, sigma,
 * t (scratch fp32 [max(K, 64)]), s (scratch fp32 [R]), R, K, dWsn, dW.  Block maps are (job, chunk) pairs: map_cols
 * 64 columns of W per block, map_rows 4 rows per block, map_dot one of 64 slices of R*K per block, map_apply 2048
 * elements per block. */
int ried_sn_sigma_jobs(const long long* jobs, int njobs, const int* map_cols, int nblocks_cols, const int* map_rows,
                       int nblocks_rows, int power_iter, float eps, ried_stream_t stream);
int ried_sn_backward_jobs(const long long* jobs, const int* map_dot, int nblocks_dot, const int* map_apply,
                          int nblocks_apply, int accumulate, ried_stream_t stream);

/* ---- losses ---------------------------------------------------------------------------------------------------- */
/* nn.MSELoss on one VGG tap (src/loss.py:185-187): a, b fp32 [n]; loss[0] += loss_scale*sum (a-b)^2;
 * g (bf16 [n]) (+)= grad_scale*(a-b) */
int ried_mse_tap(const void* a, const void* b, long long n, float loss_scale, float* loss, void* g, float grad_scale,
                 int accumulate, ried_stream_t stream);
/* One tap of MPL (src/loss.py:101-109): a, b fp32 [npix][C] (C a power of two in [8, 512]);
 * loss[0] += loss_scale * sum_p ||a[p]-b[p]||_2; g (bf16 [npix][C], may be NULL) = grad_scale*(a-b)/||a[p]-b[p]|| */
int ried_l2norm_tap(const void* a, const void* b, long long npix, int C, float loss_scale, float* loss, void* g,
                    float grad_scale, ried_stream_t stream);
/* DLoss (src/loss.py:141-142): sc[0]=mean(pos), sc[1]=mean(neg) -> sc[2]=loss, sc[3], sc[4] = per-element gradients;
 * keep (may be NULL): keep[0] = loss as well */
int ried_dloss_scalars(float* sc, long long n_per_half, float* keep, ried_stream_t stream);
/* CompoundLoss weighting (src/loss.py:191-194): sc[1], sc[5]=vgg, sc[6]=ssim -> sc[7]=dssim, sc[8]=gen, sc[9]=loss */
int ried_gloss_scalars(float* sc, float wv, float ws, float wg, ried_stream_t stream);

/* ---- SSIM / metrics -------------------------------------------------------------------------------------------- */
/* compute_SSIM (src/loss.py:42-67, src/measure.py:49-74) fused with compute_MSE (src/measure.py:18-19).
 * sums[n][0] += sum ssim_map(a,ref), [1] += sum (a-ref)^2, and with b: [2], [3] the same for (b,ref).
 * do_clamp folds trunc() of src/test.py:77-80.  gmap (two-image form): fp32 [3][nimg][H][W] for ried_ssim_bwd. */
int ried_ssim_fwd(const float* a, const float* b, const float* ref, int nimg, int H, int W, float c1, float c2,
                  int do_clamp, float lo, float hi, float* sums, float* gmap, ried_stream_t stream);
/* gaussian(11, 1.5) of src/loss.py:24-26 / src/measure.py:77-79 as the caller's torch computes it: g_host = 11 fp32
 * weights in host memory, used by every later ried_ssim_fwd / ried_ssim_bwd call of the process.  Without this call
 * the library normalises the eleven taps with a left-to-right fp32 sum, which differs from torch's sum in the last
 * bit; on raw HU the SSIM value is sensitive to that bit (see csrc/ssim.hip). */
int ried_ssim_set_window(const float* g_host, int n);
/* da (+)= scale * scale_dev[0] * d(sum ssim_map)/d a */
int ried_ssim_bwd(const float* a, const float* ref, const float* gmap, int nimg, int H, int W, float scale,
                  const float* scale_dev, float* da, int accumulate, ried_stream_t stream);

/* ---- legacy networks (src/networks.py:460-556) ---------------------------------------------------------------------- */
/* GNet.conv2 = Conv2d(192, 1, 3, padding=1) with bias (src/networks.py:485,519) on a bf16 view x [nimg][H][W] with
 * C channels (C % 8 == 0, C <= 1024); w: the fp32 master weight (1, C, 3, 3); out / dout: fp32 [nimg][H][W].
 * _dgrad: dx (+)= data gradient (bf16 view); _wgrad: dw (1, C, 3, 3) += weight gradient.  The bias gradient is
 * ried_sum_f32 of dout. */
int ried_conv3_to1_fwd(const void* x, long long ldx, const float* w, const float* bias, float* out, int nimg, int H,
                       int W, int C, ried_stream_t stream);
int ried_conv3_to1_dgrad(const float* dout, const float* w, void* dx, long long lddx, int nimg, int H, int W, int C,
                         int accumulate, ried_stream_t stream);
int ried_conv3_to1_wgrad(const void* x, long long ldx, const float* dout, float* dw, int nimg, int H, int W, int C,
                         ried_stream_t stream);
/* Data gradient of the stride-2 form of ried_conv_c1_fwd (DNet.conv1, src/networks.py:529): dy is the view
 * [nimg][(H-1)/2+1][(W-1)/2+1] with Cout = 64 channels, dx fp32 [nimg][H][W]. */
int ried_conv_c1_dgrad_s2(const void* dy, long long lddy, const float* w, const float* sigma, float* dx, int nimg,
                          int H, int W, int Cout, float scale, int accumulate, ried_stream_t stream);
/* y += alpha * x, fp32 [n]: running_mean += momentum * bias for a Conv2d with bias in front of a train-mode
 * BatchNorm2d (src/networks.py:468-469), whose normalised output does not depend on that bias. */
int ried_axpy_f32(float* y, const float* x, float alpha, long long n, ried_stream_t stream);

/* ---- training patches ------------------------------------------------------------------------------------------ */
/* get_patch (src/loader.py:60-73) followed by the .float().view(-1,1,P,P) of src/train.py:136-142, on slices kept
 * resident in HBM: slices_in / slices_tg are int16 [nslices][H][W] (the .npy files of src/prep.py:24-25); for each
 * of the nsel slices from first_slice on (the loader does not shuffle, src/loader.py:81), patch_n crops of P x P
 * at the same position in both.  Patch j = (slice j / patch_n, draw j % patch_n) goes to x_out / y_out
 * [nsel*patch_n][P][P], fp32 (out_f32) or int16.  offsets: device int [nsel*patch_n][2] = (top, left) per patch,
 * e.g. drawn on the host with numpy.random.randint(0, H - P) in the reference's order; NULL: positions are
 * splitmix64(seed, first_draw + j), top from the high and left from the low 32 bits, scaled to [0, H - P) and
 * [0, W - P).  offsets_out (may be NULL) receives the positions used.  Requires P < H, P < W, P % 4 == 0. */
int ried_patch_crop(const short* slices_in, const short* slices_tg, int nslices, int H, int W, int first_slice,
                    int nsel, int patch_n, int P, const int* offsets, unsigned long long seed,
                    unsigned long long first_draw, void* x_out, void* y_out, int out_f32, int* offsets_out,
                    ried_stream_t stream);

/* get_pixels_hu (src/prep.py:46-59) for a stack of decoded slices resident in HBM: raw = the 16-bit stored pixel
 * values [nslices][npix] (either signedness: the reference casts to int16 first), slope fp64 [nslices] and
 * intercept int32 [nslices] = int16(RescaleIntercept) per slice; out int16 [nslices][npix] in Hounsfield units, the
 * layout ried_patch_crop reads.  npix % 8 == 0. */
int ried_hu_convert(const void* raw, const double* slope, const int* intercept, short* out, int nslices,
                    long long npix, ried_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
