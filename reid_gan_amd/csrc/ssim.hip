// Fused Gaussian-window SSIM for the DSSIM loss and the SSIM/PSNR/RMSE metrics.
//
// Reference: src/loss.py:42-67 (C1 = 1e-4, C2 = 9e-4 on raw HU) and src/measure.py:18-74
// (C1 = (0.01*range)^2, C2 = (0.03*range)^2 on values clamped to [-160, 240], src/test.py:77-80).
// The reference launches five 11x11 convolutions and ~15 elementwise ops per image pair; here one
// kernel stages a 42x42 halo tile per image in LDS, runs the 11-tap Gaussian separably
// (horizontal pass into LDS, vertical pass in registers, both on pairs of quantities through
// packed fp32 FMAs), forms the SSIM quotient with the same
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

// lo <= hi: the median of (v, lo, hi) is v clamped, in one instruction (NaN inputs do not occur: CT images)
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// LDS geometry of the forward kernel (floats).  Pitches are chosen so that the 16 lanes one ds_read_b128 /
// ds_write_b128 pass serves fall on 16 different 16-byte bank slots: lanes differ in the ROW in the horizontal pass
// (input pair rows 46 x 8 B = 7 slots mod 16, input scalar rows 44 x 4 B = 11 slots, filtered rows 33 x 16 B = 1
// slot) and in the COLUMN in the vertical pass (16-byte elements: consecutive slots).
#define TPP 46              // row pitch of the input pair plane, in pairs
#define TPR 44              // row pitch of the third image's plane, in floats
#define HQP 33              // row pitch of the horizontally filtered planes, in elements

// NIMG = 2: images (a, r): quantities a, r, aa, rr, ar
// NIMG = 3: images (a, b, r): quantities a, b, r, aa, bb, rr, ar, br
// out[n][0] += sum ssim(a,r), out[n][1] += sum (a-r)^2, out[n][2] += sum ssim(b,r), out[n][3] += sum (b-r)^2
// gmap (NIMG = 2 only, optional): [3][nimg][H][W] partial derivatives of ssim wrt (mu_a, E[aa], E[ar]),
// i.e. the gradient is taken with respect to image a.
//
// The quantities travel as PAIRS so that every filter tap is one v_pk_fma_f32 per pair (the kernel is bound by
// vector-instruction issue, not by memory or LDS):
//   NIMG = 3: (a, b), (aa, bb), (ar, br), (r, rr)        NIMG = 2: (a, r), (aa, rr), (ar, -)
// and as 16-byte LDS elements: the input tile holds (a, b) pairs [+ a plane of r], the filtered planes hold
// (pair 0, pair 1) and (pair 2, pair 3) per pixel.
template <int NIMG>
__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float* __restrict__ ia, const float* __restrict__ ib,
                                                       const float* __restrict__ ir, int nimg, int H, int W, float c1,
                                                       float c2, int do_clamp, float lo, float hi, SsimWindow win,
                                                       float* __restrict__ out, float* __restrict__ gmap, int tpb) {
  constexpr int NP = NIMG == 2 ? 3 : 4;            // pairs
  extern __shared__ __attribute__((aligned(16))) float lds[];
  f2* const tinp = reinterpret_cast<f2*>(lds);                        // [TIN][TPP] pairs: (a, b) or (a, r)
  float* const tinr = lds + 2 * TIN * TPP;                            // [TIN][TPR] r (NIMG = 3)
  f4* const hq0 = reinterpret_cast<f4*>(tinr + (NIMG == 3 ? TIN * TPR : 0));      // [TIN][HQP] (pair 0, pair 1)
  f4* const hq1 = hq0 + TIN * HQP;                                    // [TIN][HQP] (pair 2, pair 3); NIMG = 2: f2
  f2* const hq1s = reinterpret_cast<f2*>(hq1);
  const int tiles_x = (W + TS - 1) / TS, ntiles = tiles_x * ((H + TS - 1) / TS);
  const int n = blockIdx.y;
  const long long ibase = (long long)n * H * W;
  // (per-image bases are uniform: the loads take them as scalar operands and a 32-bit lane offset; H * W < 2^31)
  const float* const imgs[3] = {ia + ibase, (NIMG == 3 ? ib : ir) + ibase, ir + ibase};
  // A block walks tpb tiles of one image and adds its four sums once at the end: with one block per tile the
  // float atomics of the 256 tiles of a 512 x 512 image all land on one cache line and serialise.
  float s_ssim_a = 0.f, s_mse_a = 0.f, s_ssim_b = 0.f, s_mse_b = 0.f;
  // The halo tile of the NEXT tile is fetched into registers while the current one is filtered: raw values and the
  // tile's shift constants travel together.
  constexpr int NL = (TIN * TIN + 255) / 256;
  float pv[NL][NIMG], pcs[NIMG];
  unsigned pok = 0;
  const int tile_end = min(ntiles, (blockIdx.x + 1) * tpb);
  // this thread's elements of a halo tile: i = threadIdx.x + 256 j -> (row, column), the same for every tile;
  // eoff = their offset from the tile's first halo pixel, elive = bit j set: element j exists (i < TIN * TIN)
  int eyy[NL], exx[NL], eoff[NL];
  unsigned elive = 0;
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    int i = threadIdx.x + 256 * j;
    eyy[j] = i / TIN; exx[j] = i - eyy[j] * TIN;
    const bool live = eyy[j] < TIN;
    elive |= (live ? 1u : 0u) << j;
    eoff[j] = live ? eyy[j] * W + exx[j] : HALO * W + HALO;       // (a dead element reads the tile's first pixel)
  }
  auto fetch = [&](int tile) {
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int x0 = tx * TS - HALO, y0 = ty * TS - HALO;
    const int o00 = ty * TS * W + tx * TS;
#pragma unroll
    for (int k = 0; k < NIMG; ++k) pcs[k] = imgs[k][(unsigned)o00];
    if (x0 >= 0 && y0 >= 0 && x0 + TIN <= W && y0 + TIN <= H) {      // interior tile (uniform): every pixel exists
      const int t00 = y0 * W + x0;
      pok = elive;
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const unsigned off = (unsigned)(t00 + eoff[j]);
#pragma unroll
        for (int k = 0; k < NIMG; ++k) pv[j][k] = imgs[k][off];
      }
    } else {
      pok = 0;
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int gy = y0 + eyy[j], gx = x0 + exx[j];
        const bool ok = eyy[j] < TIN && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const unsigned off = ok ? (unsigned)(gy * W + gx) : 0u;
        pok |= (ok ? 1u : 0u) << j;
#pragma unroll
        for (int k = 0; k < NIMG; ++k) pv[j][k] = imgs[k][off];
      }
    }
  };
  if (blockIdx.x * tpb < tile_end) fetch(blockIdx.x * tpb);
  for (int tile = blockIdx.x * tpb; tile < tile_end; ++tile) {
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    if (tile != blockIdx.x * tpb) __syncthreads();        // the previous tile's LDS planes have been read
    float cs[NIMG];                                       // shift constants: the tile's first pixel (always inside)
#pragma unroll
    for (int k = 0; k < NIMG; ++k) cs[k] = do_clamp ? clampf(pcs[k], lo, hi) : pcs[k];
    const bool whole = pok == elive;                      // (uniform per tile except in the last, partial, wave pass)
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      if ((elive >> j) & 1u) {
        const bool ok = whole || ((pok >> j) & 1u);
        float t[NIMG];
#pragma unroll
        for (int k = 0; k < NIMG; ++k) {
          float v = pv[j][k];
          if (do_clamp) v = clampf(v, lo, hi);
          t[k] = (ok ? v : 0.f) - cs[k];                  // zero padding, then the shift
        }
        tinp[eyy[j] * TPP + exx[j]] = f2{t[0], t[1]};
        if constexpr (NIMG == 3) tinr[eyy[j] * TPR + exx[j]] = t[2];
      }
    }
    if (tile + 1 < tile_end) fetch(tile + 1);             // in flight during the two filter passes below
    __syncthreads();
    // Horizontal pass, TIN rows x TS columns: a thread owns 8 consecutive outputs of one row and slides over their
    // 18 inputs (two per 16-byte read), so that every input and its products are read and formed once.
    for (int task = threadIdx.x; task < TIN * (TS / 8); task += 256) {
      const int seg = task / TIN, yy = task - seg * TIN;
      f2 o[NP][8];
#pragma unroll
      for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) o[k][j] = f2{0.f, 0.f};
      const f4* rowp = reinterpret_cast<const f4*>(tinp + yy * TPP + seg * 8);      // 16-byte aligned: even pitch
      const float* rowr = tinr + yy * TPR + seg * 8;
#pragma unroll
      for (int i2 = 0; i2 < 9; ++i2) {
        const f4 x2 = rowp[i2];                           // the pairs of columns 2 i2, 2 i2 + 1
        f2 r2 = f2{0.f, 0.f};
        if constexpr (NIMG == 3) r2 = *reinterpret_cast<const f2*>(rowr + 2 * i2);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int i = 2 * i2 + h;
          const f2 x0 = h ? f2{x2[2], x2[3]} : f2{x2[0], x2[1]};
          f2 P[NP];
          P[0] = x0;
          P[1] = x0 * x0;
          if constexpr (NIMG == 3) {
            const float r = h ? r2[1] : r2[0];
            P[2] = x0 * r;
            P[3] = f2{r, r * r};
          } else {
            P[2] = f2{x0[0] * x0[1], 0.f};
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (i - j >= 0 && i - j <= 10) {              // compile-time: tap i - j of output j, taps in ascending order
              const float g = win.g[i - j];
#pragma unroll
              for (int k = 0; k < NP; ++k) o[k][j] += P[k] * g;
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = yy * HQP + seg * 8 + j;
        hq0[e] = f4{o[0][j][0], o[0][j][1], o[1][j][0], o[1][j][1]};
        if constexpr (NIMG == 3) hq1[e] = f4{o[2][j][0], o[2][j][1], o[3][j][0], o[3][j][1]};
        else hq1s[e] = o[2][j];
      }
    }
    __syncthreads();
    // Vertical pass + quotient: a thread owns 4 consecutive outputs of one column (lanes: consecutive columns) and
    // slides over their 14 input rows.
    {
      const int xx = threadIdx.x & (TS - 1), ys = (threadIdx.x >> 5) * 4;     // 8 strips of 4 rows
      const float S = win.S, dS = win.dS;
      [[maybe_unused]] const f2 cab = f2{cs[0], cs[1]};
      [[maybe_unused]] const float cr = cs[NIMG - 1];
      [[maybe_unused]] const f2 k_cabS = cab * S, k_2cab = 2.f * cab, k_cab2S = cab * cab * S, k_cabcrS = cab * (cr * S);
      [[maybe_unused]] const float k_crS = cr * S, k_2cr = 2.f * cr, k_cr2S = cr * cr * S;
      f2 q4[NP][4];
#pragma unroll
      for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) q4[k][j] = f2{0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 14; ++i) {
        const int e = (ys + i) * HQP + xx;
        const f4 u = hq0[e];
        f2 P[NP];
        P[0] = f2{u[0], u[1]};
        P[1] = f2{u[2], u[3]};
        if constexpr (NIMG == 3) {
          const f4 v = hq1[e];
          P[2] = f2{v[0], v[1]};
          P[3] = f2{v[2], v[3]};
        } else {
          P[2] = hq1s[e];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (i - j >= 0 && i - j <= 10) {
            const float g = win.g[i - j];
#pragma unroll
            for (int k = 0; k < NP; ++k) q4[k][j] += P[k] * g;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int yy = ys + j;
        const int gy = ty * TS + yy, gx = tx * TS + xx;
        if (gy >= H || gx >= W) continue;
        const f2 ctr = tinp[(yy + HALO) * TPP + xx + HALO];
        if constexpr (NIMG == 2) {
          const float q0 = q4[0][j][0], q1 = q4[0][j][1], q2 = q4[1][j][0], q3 = q4[1][j][1], q4_ = q4[2][j][0];
          const float ca = cs[0], cr = cs[1];
          float mu1 = q0 + ca * S, mu2 = q1 + cr * S;
          float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
          float s1 = (q2 - q0 * q0) + dS * (2.f * ca * q0 + ca * ca * S);
          float s2 = (q3 - q1 * q1) + dS * (2.f * cr * q1 + cr * cr * S);
          float s12 = (q4_ - q0 * q1) + dS * (ca * q1 + cr * q0 + ca * cr * S);
          float A1 = 2.f * mu12 + c1, A2 = 2.f * s12 + c2, B1 = mu1s + mu2s + c1, B2 = s1 + s2 + c2;
          const float ib1b2 = __builtin_amdgcn_rcpf(B1 * B2);      // (1 ulp; the loss is compared at 2e-4)
          float ssim = (A1 * A2) * ib1b2;
          s_ssim_a += ssim;
          float a = ctr[0] + ca, r = ctr[1] + cr;
          s_mse_a += (a - r) * (a - r);
          if (gmap) {
            long long o = ibase + (long long)gy * W + gx, plane = (long long)nimg * H * W;
            // d ssim / d mu_a (E fixed), d ssim / d E[aa], d ssim / d E[ar]
            gmap[o] = (2.f * mu2 * (A2 - A1) - ssim * 2.f * mu1 * (B2 - B1)) * ib1b2;
            gmap[plane + o] = -ssim * __builtin_amdgcn_rcpf(B2);
            gmap[2 * plane + o] = 2.f * A1 * ib1b2;
          }
        } else {
          // both SSIMs side by side: lane 0 of a pair is image a against r, lane 1 image b against r.  The terms
          // that depend only on the tile's shift constants are formed once per tile (k_*); the quotient is num *
          // rcp(den) (1 ulp: the metrics are compared at 2e-4).
          const f2 qm = q4[0][j], qss = q4[1][j], qxr = q4[2][j];
          const float qr = q4[3][j][0], qrr = q4[3][j][1];
          const f2 mu = qm + k_cabS;
          const float mur = qr + k_crS;
          const float sr = (qrr - qr * qr) + dS * (k_2cr * qr + k_cr2S);
          const f2 s1 = (qss - qm * qm) + (k_2cab * qm + k_cab2S) * dS;
          const f2 s12 = (qxr - qm * qr) + (cab * qr + (qm * cr + k_cabcrS)) * dS;
          const f2 num = (2.f * (mu * mur) + c1) * (2.f * s12 + c2);
          const f2 den = (mu * mu + (mur * mur + c1)) * (s1 + (sr + c2));
          s_ssim_a += num[0] * __builtin_amdgcn_rcpf(den[0]);
          s_ssim_b += num[1] * __builtin_amdgcn_rcpf(den[1]);
          const float r = tinr[(yy + HALO) * TPR + xx + HALO] + cr;
          const f2 d = (ctr + cab) - r;
          s_mse_a += d[0] * d[0];
          s_mse_b += d[1] * d[1];
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
    size_t sh = (2 * TIN * TPP + TIN * TPR + 8 * TIN * HQP) * sizeof(float);      // 66 KB: above the 64 KB default limit
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
    size_t sh = (2 * TIN * TPP + 6 * TIN * HQP) * sizeof(float);
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
