// Fused Gaussian-window SSIM for the DSSIM loss and the SSIM/PSNR/RMSE metrics.
//
// Reference: src/loss.py:42-67 (C1 = 1e-4, C2 = 9e-4 on raw HU) and src/measure.py:18-74
// (C1 = (0.01*range)^2, C2 = (0.03*range)^2 on values clamped to [-160, 240], src/test.py:77-80).
// The reference launches five 11x11 convolutions and ~15 elementwise ops per image pair; here one
// kernel stages a 42x42 halo tile per image in LDS, runs the 11-tap Gaussian separably
// (horizontal pass into LDS, vertical pass in registers), forms the SSIM quotient with the same
// sigma = E[x^2] - mu^2 algebra and reduces with wavefront shuffles.  fp32 throughout: the
// variance cancellation on raw HU makes bf16 unusable here.
//
// Conditioning.  On a nearly constant region at -1000 HU, E[x^2] ~ 1e6 and sigma^2 ~ 1: evaluated as written,
// fp32 keeps one digit (the reference's own fp32 result is 0.08 from its fp64 result at noise sd 0.5).  The
// kernels evaluate the SAME expression on values shifted by a per-tile constant c (the tile's first pixel):
//   mu = mu' + c S,  E[x^2] - mu^2 = (E'[(x-c)^2] - mu'^2) + (1 - S)(2 c mu' + c^2 S)
// where primes are the filtered shifted values and S = the sum of the 121 fp32 window weights (1 - S ~ 1e-8: the
// reference's window does not sum to one exactly, which gives its sigma^2 a bias of c^2 (1 - S) that is kept
// here).  This is an identity in exact arithmetic, so the result is the reference expression's exact value up to
// fp32 round-off of O(1) quantities, in every input regime.
// With three images (x, pred, y) one pass yields both metric triples of measure.compute_measure
// from a single read of each image: 12 bytes per pixel.
#include "common.h"

#define TS 32               // output tile side
#define HALO 5
#define TIN (TS + 2 * HALO) // 42
#define HQS (TS + 1)        // row pitch of the horizontally filtered planes of the forward kernel
#define TP (TIN + 1)        // row pitch of the input tiles of the forward kernel (43: odd)
#define TPL (TIN * TP)      // one input plane

struct SsimWindow { float g[11]; float S, dS; };    // S = sum of the 2-d window's fp32 weights, dS = 1 - S

static void finish_window(SsimWindow& w) {
  double S = 0.0;       // the reference's 2-d window is the fp32 outer product of the 1-d one (src/loss.py:30)
  for (int i = 0; i < 11; ++i)
    for (int j = 0; j < 11; ++j) S += (double)(float)(w.g[i] * w.g[j]);
  w.S = (float)S;
  w.dS = (float)(1.0 - S);
}

static SsimWindow make_window() {
  // built as the reference builds it: exp() in double per tap, stored fp32, normalised in fp32
  SsimWindow w;
  float s = 0.f;
  for (int i = 0; i < 11; ++i) { w.g[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += w.g[i]; }
  for (int i = 0; i < 11; ++i) w.g[i] = w.g[i] / s;
  finish_window(w);
  return w;
}

// The window in use.  The default normalises with a left-to-right fp32 sum; torch's sum of the same eleven floats
// differs from it in the last bit, which moves every weight by up to 3e-8 and 1 - S from +8e-8 to -7e-8 - and on raw
// HU that is visible: c^2 (1 - S) is +-0.08 at c = -1000.  A caller that wants the reference's numbers passes the
// weights its torch produces (ried_ssim_set_window); the Python binding does so when it loads the library.
static SsimWindow g_window = make_window();

extern "C" int ried_ssim_set_window(const float* g_host, int n) {
  RIED_CHECK_ARG(g_host && n == 11, "ried_ssim_set_window: the kernels take an 11-tap window");
  SsimWindow w;
  for (int i = 0; i < 11; ++i) w.g[i] = g_host[i];
  finish_window(w);
  g_window = w;
  return RIED_OK;
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
  // The halo tile of the NEXT tile is fetched into registers while the current one is filtered (the loads of a tile
  // are ~2 us of latency, as long as its whole filter): raw values and the tile's shift constants travel together.
  constexpr int NL = (TIN * TIN + 255) / 256;
  float pv[NL][NIMG], pcs[NIMG];
  unsigned pok = 0;
  const int tile_end = min(ntiles, (blockIdx.x + 1) * tpb);
  auto fetch = [&](int tile) {
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int x0 = tx * TS - HALO, y0 = ty * TS - HALO;
#pragma unroll
    for (int k = 0; k < NIMG; ++k) pcs[k] = imgs[k][ibase + (long long)(ty * TS) * W + tx * TS];
    pok = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      int i = threadIdx.x + 256 * j;
      int yy = i / TIN, xx = i - yy * TIN;
      int gy = y0 + yy, gx = x0 + xx;
      bool ok = i < TIN * TIN && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      long long off = ok ? ibase + (long long)gy * W + gx : ibase;
      pok |= (ok ? 1u : 0u) << j;
#pragma unroll
      for (int k = 0; k < NIMG; ++k) pv[j][k] = imgs[k][off];
    }
  };
  if (blockIdx.x * tpb < tile_end) fetch(blockIdx.x * tpb);
  for (int tile = blockIdx.x * tpb; tile < tile_end; ++tile) {
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  if (tile != blockIdx.x * tpb) __syncthreads();        // the previous tile's LDS planes have been read
  float cs[NIMG];                                       // shift constants: the tile's first pixel (always inside)
#pragma unroll
  for (int k = 0; k < NIMG; ++k) cs[k] = do_clamp ? clampf(pcs[k], lo, hi) : pcs[k];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    int i = threadIdx.x + 256 * j;
    if (i < TIN * TIN) {
      int yy = i / TIN, xx = i - yy * TIN;
      const bool ok = (pok >> j) & 1u;
#pragma unroll
      for (int k = 0; k < NIMG; ++k) {
        float t = pv[j][k];
        if (do_clamp) t = clampf(t, lo, hi);
        tin[k * TPL + yy * TP + xx] = (ok ? t : 0.f) - cs[k];       // zero padding, then the shift
      }
    }
  }
  if (tile + 1 < tile_end) fetch(tile + 1);             // in flight during the two filter passes below
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
      const float S = win.S, dS = win.dS;
      if constexpr (NIMG == 2) {
        const float ca = cs[0], cr = cs[1];
        float mu1 = q[0] + ca * S, mu2 = q[1] + cr * S;
        float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
        float s1 = (q[2] - q[0] * q[0]) + dS * (2.f * ca * q[0] + ca * ca * S);
        float s2 = (q[3] - q[1] * q[1]) + dS * (2.f * cr * q[1] + cr * cr * S);
        float s12 = (q[4] - q[0] * q[1]) + dS * (ca * q[1] + cr * q[0] + ca * cr * S);
        float A1 = 2.f * mu12 + c1, A2 = 2.f * s12 + c2, B1 = mu1s + mu2s + c1, B2 = s1 + s2 + c2;
        float ssim = (A1 * A2) / (B1 * B2);
        s_ssim_a += ssim;
        float a = tin[(yy + HALO) * TP + xx + HALO] + ca, r = tin[TPL + (yy + HALO) * TP + xx + HALO] + cr;
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
        const float ca = cs[0], cb = cs[1], cr = cs[2];
        float mua = q[0] + ca * S, mub = q[1] + cb * S, mur = q[2] + cr * S;
        float murs = mur * mur, sr = (q[5] - q[2] * q[2]) + dS * (2.f * cr * q[2] + cr * cr * S);
        {
          float m1s = mua * mua, m12 = mua * mur;
          float s1 = (q[3] - q[0] * q[0]) + dS * (2.f * ca * q[0] + ca * ca * S);
          float s12 = (q[6] - q[0] * q[2]) + dS * (ca * q[2] + cr * q[0] + ca * cr * S);
          s_ssim_a += ((2.f * m12 + c1) * (2.f * s12 + c2)) / ((m1s + murs + c1) * (s1 + sr + c2));
        }
        {
          float m1s = mub * mub, m12 = mub * mur;
          float s1 = (q[4] - q[1] * q[1]) + dS * (2.f * cb * q[1] + cb * cb * S);
          float s12 = (q[7] - q[1] * q[2]) + dS * (cb * q[2] + cr * q[1] + cb * cr * S);
          s_ssim_b += ((2.f * m12 + c1) * (2.f * s12 + c2)) / ((m1s + murs + c1) * (s1 + sr + c2));
        }
        int ci = (yy + HALO) * TP + xx + HALO;
        float a = tin[ci] + ca, b = tin[TPL + ci] + cb, r = tin[2 * TPL + ci] + cr;
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

// d a[q] (+)= scale * ( G*(g_mu)[q] + 2 a[q] G*(g_aa)[q] + r[q] G*(g_ar)[q] ),  G = the 11x11 Gaussian, zero padded.
// g_mu carries -2 mu_a g_aa - mu_r g_ar, which cancels against the other two terms on a nearly constant region;
// evaluated as G*(g_mu + 2 ca g_aa + cr g_ar) + 2 (a - ca) G*(g_aa) + (r - cr) G*(g_ar) with the tile's first
// pixels (ca, cr) - the same sum for any constants - every filtered quantity is O(gradient), not O(1000 x gradient).
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
  const long long o00 = ibase + (long long)(ty * TS) * W + tx * TS;
  const float ca = ia[o00], cr = ir[o00];
  for (int i = threadIdx.x; i < TIN * TIN; i += 256) {
    int yy = i / TIN, xx = i - yy * TIN;
    int gy = y0 + yy, gx = x0 + xx;
    bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    long long o = ibase + (long long)gy * W + gx;
    float g0 = ok ? gmap[o] : 0.f, g1 = ok ? gmap[plane + o] : 0.f, g2 = ok ? gmap[2 * plane + o] : 0.f;
    tin[i] = g0 + 2.f * ca * g1 + cr * g2;
    tin[TIN * TIN + i] = g1;
    tin[2 * TIN * TIN + i] = g2;
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
    float v = sc * (q0 + 2.f * (ia[o] - ca) * q1 + (ir[o] - cr) * q2);
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
  const SsimWindow win = g_window;
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
  const SsimWindow win = g_window;
  int tiles = ((W + TS - 1) / TS) * ((H + TS - 1) / TS);
  size_t sh = (3 * TIN * TIN + 3 * TIN * TS) * sizeof(float);
  hipLaunchKernelGGL(ssim_bwd_kernel, dim3(tiles, nimg), dim3(256), sh, stream, a, ref, gmap, nimg, H, W, win, scale,
                     scale_dev, da, accumulate);
  RIED_CHECK_LAUNCH("ried_ssim_bwd");
  return RIED_OK;
}
