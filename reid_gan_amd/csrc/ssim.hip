// Fused Gaussian-window SSIM for the DSSIM loss and the SSIM/PSNR/RMSE metrics.
//
// Reference: src/loss.py:42-67 (C1 = 1e-4, C2 = 9e-4 on raw HU) and src/measure.py:18-74
// (C1 = (0.01*range)^2, C2 = (0.03*range)^2 on values clamped to [-160, 240], src/test.py:77-80).
// The reference launches five 11x11 convolutions and ~15 elementwise ops per image pair; here one
// kernel stages a 42x42 halo tile per image in LDS, runs the 11-tap Gaussian separably
// (horizontal pass into LDS, vertical pass in registers), forms the SSIM quotient with the same
// sigma = E[x^2] - mu^2 algebra and reduces with wavefront shuffles.  fp32 throughout: the
// variance cancellation on raw HU makes bf16 unusable here.
// With three images (x, pred, y) one pass yields both metric triples of measure.compute_measure
// from a single read of each image: 12 bytes per pixel.
#include "common.h"

#define TS 32               // output tile side
#define HALO 5
#define TIN (TS + 2 * HALO) // 42
#define HQS (TS + 1)        // row pitch of the horizontally filtered planes of the forward kernel
#define TP (TIN + 1)        // row pitch of the input tiles of the forward kernel (43: odd)
#define TPL (TIN * TP)      // one input plane

struct SsimWindow { float g[11]; };

static SsimWindow make_window() {
  // built as the reference builds it: exp() in double per tap, stored fp32, normalised in fp32
  SsimWindow w;
  float s = 0.f;
  for (int i = 0; i < 11; ++i) { w.g[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += w.g[i]; }
  for (int i = 0; i < 11; ++i) w.g[i] = w.g[i] / s;
  return w;
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// NIMG = 2: images (a, r): quantities a, r, aa, rr, ar
// NIMG = 3: images (a, b, r): quantities a, b, r, aa, bb, rr, ar, br
// out[n][0] += sum ssim(a,r), out[n][1] += sum (a-r)^2, out[n][2] += sum ssim(b,r), out[n][3] += sum (b-r)^2
// gmap (NIMG = 2 only, optional): [3][nimg][H][W] partial derivatives of ssim wrt (mu_a, E[aa], E[ar]),
// i.e. the gradient is taken with respect to image a.
template <int NIMG>
__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float* __restrict__ ia, const float* __restrict__ ib,
                                                       const float* __restrict__ ir, int nimg, int H, int W, float c1,
                                                       float c2, int do_clamp, float lo, float hi, SsimWindow win,
                                                       float* __restrict__ out, float* __restrict__ gmap, int tpb) {
  constexpr int NQ = NIMG == 2 ? 5 : 8;
  extern __shared__ float lds[];
  float* tin = lds;                                // [NIMG][TIN][TP]: odd pitch, conflict-free column walks
  float* hq = lds + NIMG * TPL;              // [NQ][TIN][HQS]
  const int tiles_x = (W + TS - 1) / TS, ntiles = tiles_x * ((H + TS - 1) / TS);
  const int n = blockIdx.y;
  const long long ibase = (long long)n * H * W;
  const float* imgs[3] = {ia, NIMG == 3 ? ib : ir, ir};
  // A block walks tpb tiles of one image and adds its four sums once at the end: with one block per tile the
  // float atomics of the 256 tiles of a 512 x 512 image all land on one cache line and serialise - that, not the
  // filter, was what bounded the kernel on full slices.
  float s_ssim_a = 0.f, s_mse_a = 0.f, s_ssim_b = 0.f, s_mse_b = 0.f;
  for (int tile = blockIdx.x * tpb; tile < min(ntiles, (blockIdx.x + 1) * tpb); ++tile) {
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int x0 = tx * TS - HALO, y0 = ty * TS - HALO;
  if (tile != blockIdx.x * tpb) __syncthreads();        // the previous tile's LDS planes have been read

  {
    // every global load of the halo tile is issued before the first LDS store, so that the block pays the
    // memory latency once instead of once per pass of the loop
    constexpr int NL = (TIN * TIN + 255) / 256;
    float v[NL][NIMG];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      int i = threadIdx.x + 256 * j;
      int yy = i / TIN, xx = i - yy * TIN;
      int gy = y0 + yy, gx = x0 + xx;
      bool ok = i < TIN * TIN && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      long long off = ok ? ibase + (long long)gy * W + gx : ibase;
#pragma unroll
      for (int k = 0; k < NIMG; ++k) {
        float t = imgs[k][off];
        if (do_clamp) t = clampf(t, lo, hi);
        v[j][k] = ok ? t : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      int i = threadIdx.x + 256 * j;
      if (i < TIN * TIN) {
        int yy = i / TIN, xx = i - yy * TIN;
#pragma unroll
        for (int k = 0; k < NIMG; ++k) tin[k * TPL + yy * TP + xx] = v[j][k];
      }
    }
  }
  __syncthreads();
  // Horizontal pass, TIN rows x TS columns: a thread owns 8 consecutive outputs of one row and slides over their
  // 18 inputs, so that every input (and its products) is read and formed once instead of once per tap.  Lanes
  // differ in the row: input addresses are 42 floats apart (two-way bank conflicts at worst), output rows are
  // padded to HQS = 33 floats (conflict-free).
  for (int task = threadIdx.x; task < TIN * (TS / 8); task += 256) {
    const int seg = task / TIN, yy = task - seg * TIN;
    float o[NQ][8];
#pragma unroll
    for (int k = 0; k < NQ; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) o[k][j] = 0.f;
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const int ci = yy * TP + seg * 8 + i;
      float v[NQ];
      if constexpr (NIMG == 2) {
        const float a = tin[ci], r = tin[TPL + ci];
        v[0] = a; v[1] = r; v[2] = a * a; v[3] = r * r; v[4] = a * r;
      } else {
        const float a = tin[ci], b = tin[TPL + ci], r = tin[2 * TPL + ci];
        v[0] = a; v[1] = b; v[2] = r; v[3] = a * a; v[4] = b * b; v[5] = r * r; v[6] = a * r; v[7] = b * r;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (i - j >= 0 && i - j <= 10) {             // compile-time: tap i - j of output j, taps in ascending order
          const float g = win.g[i - j];
#pragma unroll
          for (int k = 0; k < NQ; ++k) o[k][j] += g * v[k];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NQ; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) hq[(k * TIN + yy) * HQS + seg * 8 + j] = o[k][j];
  }
  __syncthreads();
  // Vertical pass + quotient: a thread owns 4 consecutive outputs of one column (lanes: consecutive columns) and
  // slides over their 14 input rows.
  {
    const int xx = threadIdx.x & (TS - 1), ys = (threadIdx.x >> 5) * 4;     // 8 strips of 4 rows
    float q4[NQ][4];
#pragma unroll
    for (int k = 0; k < NQ; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) q4[k][j] = 0.f;
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      float v[NQ];
#pragma unroll
      for (int k = 0; k < NQ; ++k) v[k] = hq[(k * TIN + ys + i) * HQS + xx];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (i - j >= 0 && i - j <= 10) {
          const float g = win.g[i - j];
#pragma unroll
          for (int k = 0; k < NQ; ++k) q4[k][j] += g * v[k];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int yy = ys + j;
      const int gy = ty * TS + yy, gx = tx * TS + xx;
      if (gy >= H || gx >= W) continue;
      float q[NQ];
#pragma unroll
      for (int k = 0; k < NQ; ++k) q[k] = q4[k][j];
      if constexpr (NIMG == 2) {
        float mu1 = q[0], mu2 = q[1];
        float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
        float s1 = q[2] - mu1s, s2 = q[3] - mu2s, s12 = q[4] - mu12;
        float A1 = 2.f * mu12 + c1, A2 = 2.f * s12 + c2, B1 = mu1s + mu2s + c1, B2 = s1 + s2 + c2;
        float ssim = (A1 * A2) / (B1 * B2);
        s_ssim_a += ssim;
        float a = tin[(yy + HALO) * TP + xx + HALO], r = tin[TPL + (yy + HALO) * TP + xx + HALO];
        s_mse_a += (a - r) * (a - r);
        if (gmap) {
          float ib1b2 = 1.f / (B1 * B2);
          long long o = ibase + (long long)gy * W + gx, plane = (long long)nimg * H * W;
          // d ssim / d mu_a (E fixed), d ssim / d E[aa], d ssim / d E[ar]
          gmap[o] = (2.f * mu2 * (A2 - A1) - ssim * 2.f * mu1 * (B2 - B1)) * ib1b2;
          gmap[plane + o] = -ssim / B2;
          gmap[2 * plane + o] = 2.f * A1 * ib1b2;
        }
      } else {
        float mua = q[0], mub = q[1], mur = q[2];
        float murs = mur * mur, sr = q[5] - murs;
        {
          float m1s = mua * mua, m12 = mua * mur;
          float s1 = q[3] - m1s, s12 = q[6] - m12;
          s_ssim_a += ((2.f * m12 + c1) * (2.f * s12 + c2)) / ((m1s + murs + c1) * (s1 + sr + c2));
        }
        {
          float m1s = mub * mub, m12 = mub * mur;
          float s1 = q[4] - m1s, s12 = q[7] - m12;
          s_ssim_b += ((2.f * m12 + c1) * (2.f * s12 + c2)) / ((m1s + murs + c1) * (s1 + sr + c2));
        }
        int ci = (yy + HALO) * TP + xx + HALO;
        float a = tin[ci], b = tin[TPL + ci], r = tin[2 * TPL + ci];
        s_mse_a += (a - r) * (a - r);
        s_mse_b += (b - r) * (b - r);
      }
    }
  }
  }     // tiles of this block
  __shared__ float sh4[4];
  float t0 = block_sum_256(s_ssim_a, sh4);
  float t1 = block_sum_256(s_mse_a, sh4);
  if (threadIdx.x == 0) { atomicAdd(out + n * 4 + 0, t0); atomicAdd(out + n * 4 + 1, t1); }
  if constexpr (NIMG == 3) {
    float t2 = block_sum_256(s_ssim_b, sh4);
    float t3 = block_sum_256(s_mse_b, sh4);
    if (threadIdx.x == 0) { atomicAdd(out + n * 4 + 2, t2); atomicAdd(out + n * 4 + 3, t3); }
  }
}

// d a[q] (+)= scale * ( G*(g_mu)[q] + 2 a[q] G*(g_aa)[q] + r[q] G*(g_ar)[q] ),  G = the 11x11 Gaussian, zero padded
__global__ __launch_bounds__(256) void ssim_bwd_kernel(const float* __restrict__ ia, const float* __restrict__ ir,
                                                       const float* __restrict__ gmap, int nimg, int H, int W,
                                                       SsimWindow win, float scale, const float* __restrict__ scale_dev,
                                                       float* __restrict__ da, int accumulate) {
  extern __shared__ float lds[];
  float* tin = lds;                      // [3][TIN][TIN]
  float* hq = lds + 3 * TIN * TIN;       // [3][TIN][TS]
  const int tiles_x = (W + TS - 1) / TS;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, n = blockIdx.y;
  const int x0 = tx * TS - HALO, y0 = ty * TS - HALO;
  const long long ibase = (long long)n * H * W, plane = (long long)nimg * H * W;
  for (int i = threadIdx.x; i < TIN * TIN; i += 256) {
    int yy = i / TIN, xx = i - yy * TIN;
    int gy = y0 + yy, gx = x0 + xx;
    bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    long long o = ibase + (long long)gy * W + gx;
#pragma unroll
    for (int k = 0; k < 3; ++k) tin[k * TIN * TIN + i] = ok ? gmap[k * plane + o] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TIN * TS; i += 256) {
    int yy = i / TS, xx = i - yy * TS;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
    for (int t = 0; t < 11; ++t) {
      float g = win.g[t];
      q0 += g * tin[yy * TIN + xx + t];
      q1 += g * tin[TIN * TIN + yy * TIN + xx + t];
      q2 += g * tin[2 * TIN * TIN + yy * TIN + xx + t];
    }
    hq[i] = q0; hq[TIN * TS + i] = q1; hq[2 * TIN * TS + i] = q2;
  }
  __syncthreads();
  const float sc = scale * (scale_dev ? scale_dev[0] : 1.f);
  for (int i = threadIdx.x; i < TS * TS; i += 256) {
    int yy = i / TS, xx = i - yy * TS;
    int gy = ty * TS + yy, gx = tx * TS + xx;
    if (gy >= H || gx >= W) continue;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
    for (int t = 0; t < 11; ++t) {
      float g = win.g[t];
      q0 += g * hq[(yy + t) * TS + xx];
      q1 += g * hq[TIN * TS + (yy + t) * TS + xx];
      q2 += g * hq[2 * TIN * TS + (yy + t) * TS + xx];
    }
    long long o = ibase + (long long)gy * W + gx;
    float v = sc * (q0 + 2.f * ia[o] * q1 + ir[o] * q2);
    da[o] = accumulate ? da[o] + v : v;
  }
}

// sums: fp32 [nimg][4], accumulated onto (zero first).  b may be null (two-image form).
// gmap: null or fp32 [3][nimg][H][W] (two-image form only).
extern "C" int ried_ssim_fwd(const float* a, const float* b, const float* ref, int nimg, int H, int W, float c1,
                             float c2, int do_clamp, float lo, float hi, float* sums, float* gmap,
                             hipStream_t stream) {
  RIED_CHECK_ARG(a && ref && sums && nimg > 0 && nimg <= 65535 && H > 0 && W > 0, "ried_ssim_fwd: bad arguments");
  RIED_CHECK_ARG(!(b && gmap), "ried_ssim_fwd: gradient maps need the two-image form");
  static const SsimWindow win = make_window();
  int tiles = ((W + TS - 1) / TS) * ((H + TS - 1) / TS);
  // tiles per block: about four blocks per CU in all, at most 16 tiles each
  int tpb = (int)(((long long)tiles * nimg + 1023) / 1024);
  tpb = tpb < 1 ? 1 : tpb > 16 ? 16 : tpb;
  dim3 grid((tiles + tpb - 1) / tpb, nimg);
  if (b) {
    size_t sh = (3 * TPL + 8 * TIN * HQS) * sizeof(float);          // 66 KB: above the 64 KB default limit
    static int raised = 0;
    if (!raised) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ssim_fwd_kernel<3>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      RIED_CHECK_ARG(e == hipSuccess, "ried_ssim_fwd: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
      raised = 1;
    }
    hipLaunchKernelGGL(ssim_fwd_kernel<3>, grid, dim3(256), sh, stream, a, b, ref, nimg, H, W, c1, c2, do_clamp, lo,
                       hi, win, sums, (float*)nullptr, tpb);
  } else {
    size_t sh = (2 * TPL + 5 * TIN * HQS) * sizeof(float);
    hipLaunchKernelGGL(ssim_fwd_kernel<2>, grid, dim3(256), sh, stream, a, (const float*)nullptr, ref, nimg, H, W, c1,
                       c2, do_clamp, lo, hi, win, sums, gmap, tpb);
  }
  RIED_CHECK_LAUNCH("ried_ssim_fwd");
  return RIED_OK;
}

// da (+)= scale * scale_dev[0] * d(sum of ssim map)/d a
extern "C" int ried_ssim_bwd(const float* a, const float* ref, const float* gmap, int nimg, int H, int W, float scale,
                             const float* scale_dev, float* da, int accumulate, hipStream_t stream) {
  RIED_CHECK_ARG(a && ref && gmap && da && nimg > 0 && nimg <= 65535 && H > 0 && W > 0, "ried_ssim_bwd: bad arguments");
  static const SsimWindow win = make_window();
  int tiles = ((W + TS - 1) / TS) * ((H + TS - 1) / TS);
  size_t sh = (3 * TIN * TIN + 3 * TIN * TS) * sizeof(float);
  hipLaunchKernelGGL(ssim_bwd_kernel, dim3(tiles, nimg), dim3(256), sh, stream, a, ref, gmap, nimg, H, W, win, scale,
                     scale_dev, da, accumulate);
  RIED_CHECK_LAUNCH("ried_ssim_bwd");
  return RIED_OK;
}
