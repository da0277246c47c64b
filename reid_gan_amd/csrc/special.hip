// Layer-specific kernels that are not bf16 implicit GEMMs:
//  * single-input-channel 3x3 convolutions reading the fp32 CT image (generator conv1, discriminator layer 0,
//    VGG16 layer 0 with its three identical input channels folded into one) - K = 9 is bandwidth bound;
//  * the generator's 33->1 output convolution, kept in fp32 because its first input is the raw HU image;
//  * spectral normalisation (one power iteration) and its backward term;
//  * perceptual MSE taps, hinge / generator loss scalars, constant-gradient fills.
// Reference: src/rinet.py:20,55,70,131-148; src/loss.py:132-197; torch/nn/utils/spectral_norm.py:92-114.
#include <stdlib.h>
#include "common.h"

#define SP_THREADS 256

__device__ __forceinline__ void unpack8s(const u32x4& v, float (&f)[8]) {
  f[0] = bflo(v[0]); f[1] = bfhi(v[0]); f[2] = bflo(v[1]); f[3] = bfhi(v[1]);
  f[4] = bflo(v[2]); f[5] = bfhi(v[2]); f[6] = bflo(v[3]); f[7] = bfhi(v[3]);
}

// ---------------------------------------------------------------------------
// 3x3, pad 1, one input channel (fp32 image), stride 1 or 2
// ---------------------------------------------------------------------------
// y[p][co] = act( sum_t w[co][t] * (xin*in_scale + in_shift)[p@t] / sigma + bias[co] ), zero padding after the affine.
//
// One lane per output pixel: a block takes 256 consecutive pixels, each lane reads its 9 input values once
// (neighbouring lanes read neighbouring addresses) and produces all COUT channels from weights that are uniform
// across the wave (scalar loads).  The bf16 results go through a padded LDS tile so that the global stores
// are full 16-byte chunks in linear order instead of 64 scattered rows per instruction.
#define C1_TILE 256
typedef float c1_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void c1_taps(const float* __restrict__ x, long long p, long long npix, int H, int W, int Ho,
                                        int Wo, int stride, float in_scale, float in_shift, float (&xv)[9]) {
#pragma unroll
  for (int t = 0; t < 9; ++t) xv[t] = 0.f;
  if (p >= npix) return;
  const long long hw = (long long)Ho * Wo;
  const int n = (int)(p / hw);
  const int r0 = (int)(p - (long long)n * hw);
  const int oy = r0 / Wo, ox = r0 - oy * Wo;
  const float* xi = x + (long long)n * H * W;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    int hi = oy * stride - 1 + r;
#pragma unroll
    for (int z = 0; z < 3; ++z) {
      int wi = ox * stride - 1 + z;
      bool ok = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
      int off = ok ? hi * W + wi : 0;
      float v = xi[off] * in_scale + in_shift;
      xv[r * 3 + z] = ok ? v : 0.f;
    }
  }
}

// ACT (0 none, 1 ReLU, 2 LeakyReLU 0.2), BIAS, STATS and SIGNS are compile-time: as run-time flags inside the
// unrolled channel loop they became a scalar bias load with a full wait and three branches PER CHANNEL, a chain of
// 64 dependent scalar-cache round trips per wave that bound the kernel.
template <int COUT, int ACT, bool BIAS, bool STATS, bool SIGNS>
__global__ __launch_bounds__(C1_TILE) void conv_c1_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ sigma,
    const float* __restrict__ bias, bf16_t* __restrict__ y, long long ldy, int nimg, int H, int W, int Ho, int Wo,
    int stride, float in_scale, float in_shift, float* __restrict__ stats, unsigned char* __restrict__ signs) {
  constexpr int ROWB = COUT * 2 + 16;                 // bytes per pixel in the LDS tile (16 of padding)
  constexpr int CC = COUT / 8;
  __shared__ __attribute__((aligned(16))) char tile[C1_TILE * ROWB];
  __shared__ float red[STATS ? 4 : 1][2 * COUT];
  const float inv = sigma ? 1.f / sigma[0] : 1.f;
  const long long npix = (long long)nimg * Ho * Wo;
  const long long p0 = (long long)blockIdx.x * C1_TILE;
  float xv[9];
  c1_taps(x, p0 + threadIdx.x, npix, H, W, Ho, Wo, stride, in_scale, in_shift, xv);
#pragma unroll
  for (int t = 0; t < 9; ++t) xv[t] *= inv;           // 1/sigma on the nine inputs instead of the COUT sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned sg[SIGNS ? CC / 4 : 1];                    // one bit per channel: the stored value is positive
#pragma unroll
  for (int i = 0; i < (SIGNS ? CC / 4 : 1); ++i) sg[i] = 0u;
#pragma unroll
  for (int cc = 0; cc < CC; ++cc) {
    // The weights (and the bias) are indexed by compile-time constants off a kernel argument: they arrive through
    // the scalar cache as SGPR operands of the FMAs (an LDS copy, broadcast-read 144 times per pixel, was what bound
    // an earlier form).  The kernel is bound by vector-instruction issue, so eight of a channel's nine taps run as
    // four packed FMAs on (tap, tap + 1) pairs - the pairs start at tap 0 for even channels and at tap 1 for odd ones,
    // which keeps every weight pair on an even SGPR - and the two half sums are added at the end.
    float acc[8];
    c1_f2 a2[8];                                      // the eight channels' chains side by side, not one after another
    const float* wc = w + cc * 72;
#pragma unroll
    for (int q = 0; q < 8; ++q) {                     // o = q & 1: (cc * 8 + q) * 9 is odd exactly when q is
      const int o = q & 1;
      a2[q] = c1_f2{xv[o], xv[o + 1]} * c1_f2{wc[q * 9 + o], wc[q * 9 + o + 1]};
    }
#pragma unroll
    for (int k = 1; k < 4; ++k)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int o = q & 1;
        a2[q] = c1_f2{xv[2 * k + o], xv[2 * k + o + 1]} * c1_f2{wc[q * 9 + 2 * k + o], wc[q * 9 + 2 * k + o + 1]} + a2[q];
      }
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = fmaf(xv[(q & 1) ? 0 : 8], wc[q * 9 + ((q & 1) ? 0 : 8)], a2[q][0] + a2[q][1]);
    if constexpr (STATS) {      // per-channel sum and sum of squares of the raw results over the block's pixels
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float s1 = wave_sum(acc[q]), s2 = wave_sum(acc[q] * acc[q]);      // out-of-range pixels hold zeros
        if (lane == 0) { red[wave][cc * 8 + q] = s1; red[wave][COUT + cc * 8 + q] = s2; }
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = acc[q];
      if constexpr (BIAS) v += bias[cc * 8 + q];
      if constexpr (ACT == 1) v = fmaxf(v, 0.f);
      else if constexpr (ACT == 2) v = fmaxf(v, 0.2f * v);           // = v > 0 ? v : 0.2 v
      acc[q] = v;
      // 0 - v has its sign bit set exactly when v > 0 (both zeros give +0); one funnel shift moves it into the
      // word, most significant bit first, and the word is reversed once at the end
      if constexpr (SIGNS) sg[cc / 4] = __builtin_amdgcn_alignbit(sg[cc / 4], __float_as_uint(0.f - v), 31);
    }
    *reinterpret_cast<u32x4*>(tile + threadIdx.x * ROWB + cc * 16) =
        u32x4{pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]), pack2bf(acc[4], acc[5]), pack2bf(acc[6], acc[7])};
  }
  if constexpr (SIGNS) {
    if (p0 + threadIdx.x < npix) {                    // CC bytes per pixel, consecutive lanes consecutive pixels
      unsigned* sp = reinterpret_cast<unsigned*>(signs + (p0 + threadIdx.x) * CC);
      if constexpr (CC == 8) *reinterpret_cast<u32x2*>(sp) = u32x2{__builtin_bitreverse32(sg[0]), __builtin_bitreverse32(sg[1])};
      else sp[0] = __builtin_bitreverse32(sg[0]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < CC; ++i) {
    int j = i * C1_TILE + threadIdx.x;
    int px = j / CC, cc = j - px * CC;
    if (p0 + px < npix)
      *reinterpret_cast<u32x4*>(y + (p0 + px) * ldy + cc * 8) = *reinterpret_cast<const u32x4*>(tile + px * ROWB + cc * 16);
  }
  if constexpr (STATS) {
    float* row = stats + (long long)blockIdx.x * 2 * COUT;        // one partial row per block (no global atomics)
    for (int i = threadIdx.x; i < 2 * COUT; i += C1_TILE) row[i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
  }
}

struct C1Args {
  const float *x, *w, *sigma, *bias; bf16_t* y; long long ldy; int nimg, H, W, Ho, Wo, stride; float in_scale, in_shift;
  float* stats; unsigned char* signs; int grid; hipStream_t stream;
};
template <int COUT, int ACT, bool BIAS, bool STATS, bool SIGNS>
static void c1_launch(const C1Args& a) {
  hipLaunchKernelGGL((conv_c1_fwd_kernel<COUT, ACT, BIAS, STATS, SIGNS>), dim3(a.grid), dim3(C1_TILE), 0, a.stream, a.x,
                     a.w, a.sigma, a.bias, a.y, a.ldy, a.nimg, a.H, a.W, a.Ho, a.Wo, a.stride, a.in_scale, a.in_shift,
                     a.stats, a.signs);
}
template <int COUT, int ACT>
static void c1_dispatch(const C1Args& a) {
  const bool b = a.bias != nullptr, st = a.stats != nullptr;
  if (a.signs) {                     // (sign bytes only without statistics: the discriminator's first layer)
    if (b) c1_launch<COUT, ACT, true, false, true>(a); else c1_launch<COUT, ACT, false, false, true>(a);
  } else if (b && st) c1_launch<COUT, ACT, true, true, false>(a);
  else if (b) c1_launch<COUT, ACT, true, false, false>(a);
  else if (st) c1_launch<COUT, ACT, false, true, false>(a);
  else c1_launch<COUT, ACT, false, false, false>(a);
}

static int conv_c1_fwd_impl(const float* x, const float* w, const float* sigma, const float* bias, void* y,
                            long long ldy, int nimg, int H, int W, int Cout, int stride, float in_scale,
                            float in_shift, int act, float* stats, unsigned char* signs, hipStream_t stream) {
  RIED_CHECK_ARG(x && w && y && nimg > 0 && H > 0 && W > 0, "ried_conv_c1_fwd: bad arguments");
  RIED_CHECK_ARG(((uintptr_t)signs % 8) == 0 && !(signs && stats),
                 "ried_conv_c1_fwd_signs: the sign bytes must be 8-byte aligned (and exclude statistics)");
  RIED_CHECK_ARG(stride == 1 || stride == 2, "ried_conv_c1_fwd: stride must be 1 or 2");
  RIED_CHECK_ARG(ldy % 8 == 0 && ldy >= Cout && ((uintptr_t)y % 16) == 0, "ried_conv_c1_fwd: output view");
  RIED_CHECK_ARG(Cout == 32 || Cout == 64, "ried_conv_c1_fwd: Cout must be 32 or 64 (got %d)", Cout);
  RIED_CHECK_ARG(act >= 0 && act <= 2, "ried_conv_c1_fwd: act must be 0, 1 or 2");
  RIED_CHECK_ARG((long long)H * W < (1ll << 31), "ried_conv_c1_fwd: image too large");
  int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  long long npix = (long long)nimg * Ho * Wo;
  C1Args a{x, w, sigma, bias, (bf16_t*)y, ldy, nimg, H, W, Ho, Wo, stride, in_scale, in_shift, stats, signs,
           (int)ceil_div(npix, C1_TILE), stream};
  if (Cout == 32) {
    if (act == 0) c1_dispatch<32, 0>(a); else if (act == 1) c1_dispatch<32, 1>(a); else c1_dispatch<32, 2>(a);
  } else {
    if (act == 0) c1_dispatch<64, 0>(a); else if (act == 1) c1_dispatch<64, 1>(a); else c1_dispatch<64, 2>(a);
  }
  RIED_CHECK_LAUNCH("ried_conv_c1_fwd");
  return RIED_OK;
}
extern "C" int ried_conv_c1_fwd(const float* x, const float* w, const float* sigma, const float* bias, void* y,
                                long long ldy, int nimg, int H, int W, int Cout, int stride, float in_scale,
                                float in_shift, int act, float* stats, hipStream_t stream) {
  return conv_c1_fwd_impl(x, w, sigma, bias, y, ldy, nimg, H, W, Cout, stride, in_scale, in_shift, act, stats, nullptr,
                          stream);
}
// The same, and signs[p][Cout/8] (one byte per 8-channel chunk, bit q = channel 8*chunk + q) says which of the stored
// values are positive: all that the backward of the activation needs (ried_conv2d_signs), at 1/16 of the bytes.
extern "C" int ried_conv_c1_fwd_signs(const float* x, const float* w, const float* sigma, const float* bias, void* y,
                                      long long ldy, int nimg, int H, int W, int Cout, int stride, float in_scale,
                                      float in_shift, int act, float* stats, unsigned char* signs, hipStream_t stream) {
  RIED_CHECK_ARG(signs != nullptr, "ried_conv_c1_fwd_signs: null pointer");
  return conv_c1_fwd_impl(x, w, sigma, bias, y, ldy, nimg, H, W, Cout, stride, in_scale, in_shift, act, stats, signs,
                          stream);
}

// rows of the partial-statistics buffer ([rows][2][Cout]) that ried_conv_c1_fwd writes when stats is given
extern "C" int ried_conv_c1_stat_rows(int nimg, int H, int W, int Cout, int stride) {
  int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  return ceil_div((long long)nimg * Ho * Wo, C1_TILE);
}

// part[block][co][t] = sum over the block's pixels of dy[p][co] * xaff[p@t]; a second kernel adds the rows into dw.
// Tiles of 256 pixels, 8 waves: the dy tile is staged through LDS with coalesced 16-byte loads and the first four
// waves compute the 9 taps of one pixel each; then wave w accumulates channels [w*COUT/8, (w+1)*COUT/8) for all 256
// pixels, one pixel per lane and pass, in registers (72 accumulators at COUT = 64, so that four waves fit a SIMD
// and the loads of one block hide behind the arithmetic of another).  The lanes are summed once, after the block's
// last tile.
#define C1_WG_THREADS 512
template <int COUT>
__global__ __launch_bounds__(C1_WG_THREADS, 4) void conv_c1_wgrad_kernel(const float* __restrict__ x,
                                                                          const bf16_t* __restrict__ dy, long long lddy,
                                                                          int nimg, int H, int W, int Ho, int Wo,
                                                                          int stride, float in_scale, float in_shift,
                                                                          float* __restrict__ part) {
  constexpr int ROWB = COUT * 2 + 16;
  constexpr int CC = COUT / 8;
  constexpr int CPW = COUT / 8;                       // channels per wave: 8 or 4
  __shared__ __attribute__((aligned(16))) char tile[C1_TILE * ROWB];
  __shared__ float xs[9][C1_TILE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long npix = (long long)nimg * Ho * Wo;
  const int ntiles = (int)((npix + C1_TILE - 1) / C1_TILE);
  float acc[CPW][9];
#pragma unroll
  for (int c = 0; c < CPW; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
  for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const long long p0 = (long long)tl * C1_TILE;
    __syncthreads();                                  // the previous tile has been consumed
#pragma unroll
    for (int i = 0; i < C1_TILE * CC / C1_WG_THREADS; ++i) {
      int j = i * C1_WG_THREADS + threadIdx.x;
      int px = j / CC, cc = j - px * CC;
      u32x4 v = u32x4{0, 0, 0, 0};
      if (p0 + px < npix) v = *reinterpret_cast<const u32x4*>(dy + (p0 + px) * lddy + cc * 8);
      *reinterpret_cast<u32x4*>(tile + px * ROWB + cc * 16) = v;
    }
    if (threadIdx.x < C1_TILE) {
      float xv[9];
      c1_taps(x, p0 + threadIdx.x, npix, H, W, Ho, Wo, stride, in_scale, in_shift, xv);
#pragma unroll
      for (int t = 0; t < 9; ++t) xs[t][threadIdx.x] = xv[t];
    }
    __syncthreads();
#pragma unroll 1
    for (int j = 0; j < C1_TILE / 64; ++j) {
      const int px = lane + 64 * j;
      float g[CPW], xt[9];
      if constexpr (CPW == 8) {
        unpack8s(*reinterpret_cast<const u32x4*>(tile + px * ROWB + wave * 16), g);
      } else {
        const u32x2 r = *reinterpret_cast<const u32x2*>(tile + px * ROWB + wave * 8);
        g[0] = bflo(r[0]); g[1] = bfhi(r[0]); g[2] = bflo(r[1]); g[3] = bfhi(r[1]);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) xt[t] = xs[t][px];
#pragma unroll
      for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] += g[c] * xt[t];
    }
  }
  float* row = part + (long long)blockIdx.x * COUT * 9 + wave * CPW * 9;
#pragma unroll
  for (int c = 0; c < CPW; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v = wave_sum(acc[c][t]);
      if (lane == 0) row[c * 9 + t] = v;
    }
}

// out[c] += sum over rows of part[r][c]: 32 columns x 32 row lanes per block, four loads in flight per lane
__global__ __launch_bounds__(1024) void rows_add_kernel(const float* __restrict__ part, int nrows, int Wd,
                                                        float* __restrict__ out) {
  __shared__ float sh[32][33];
  const int cl = threadIdx.x & 31, ty = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
  float a = 0.f;
  if (c < Wd) {
#pragma unroll 4
    for (int r = ty; r < nrows; r += 32) a += part[(long long)r * Wd + c];
  }
  sh[ty][cl] = a;
  __syncthreads();
  if (ty == 0 && c < Wd) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += sh[k][cl];
    out[c] += t;
  }
}

// out[c] += sum over r < nrows of part[r][c], c < width (partial rows that blocks wrote without atomics)
extern "C" int ried_rows_add(const float* part, int nrows, int width, float* out, hipStream_t stream) {
  RIED_CHECK_ARG(part && out && nrows > 0 && width > 0, "ried_rows_add: bad arguments");
  hipLaunchKernelGGL(rows_add_kernel, dim3(ceil_div(width, 32)), dim3(1024), 0, stream, part, nrows, width, out);
  RIED_CHECK_LAUNCH("ried_rows_add");
  return RIED_OK;
}

#define C1_WGRAD_MAX_ROWS 512
// dw[Cout][9] += weight gradient.  scratch: fp32 [ried_conv_c1_wgrad_scratch_floats(Cout)].
extern "C" int ried_conv_c1_wgrad(const float* x, const void* dy, long long lddy, int nimg, int H, int W, int Cout,
                                  int stride, float in_scale, float in_shift, float* dw, float* scratch,
                                  hipStream_t stream) {
  RIED_CHECK_ARG(x && dy && dw && scratch && nimg > 0, "ried_conv_c1_wgrad: bad arguments");
  RIED_CHECK_ARG(stride == 1 || stride == 2, "ried_conv_c1_wgrad: stride must be 1 or 2");
  RIED_CHECK_ARG(lddy % 8 == 0 && lddy >= Cout && ((uintptr_t)dy % 16) == 0, "ried_conv_c1_wgrad: dy view");
  RIED_CHECK_ARG(Cout == 32 || Cout == 64, "ried_conv_c1_wgrad: Cout must be 32 or 64");
  RIED_CHECK_ARG((long long)H * W < (1ll << 31), "ried_conv_c1_wgrad: image too large");
  int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  long long npix = (long long)nimg * Ho * Wo;
  int grid = ew_grid(npix, C1_TILE * 2, C1_WGRAD_MAX_ROWS);
  if (Cout == 32)
    hipLaunchKernelGGL(conv_c1_wgrad_kernel<32>, dim3(grid), dim3(C1_WG_THREADS), 0, stream, x, (const bf16_t*)dy, lddy,
                       nimg, H, W, Ho, Wo, stride, in_scale, in_shift, scratch);
  else
    hipLaunchKernelGGL(conv_c1_wgrad_kernel<64>, dim3(grid), dim3(C1_WG_THREADS), 0, stream, x, (const bf16_t*)dy, lddy,
                       nimg, H, W, Ho, Wo, stride, in_scale, in_shift, scratch);
  hipLaunchKernelGGL(rows_add_kernel, dim3(ceil_div(Cout * 9, 32)), dim3(1024), 0, stream, scratch, grid, Cout * 9, dw);
  RIED_CHECK_LAUNCH("ried_conv_c1_wgrad");
  return RIED_OK;
}
extern "C" long long ried_conv_c1_wgrad_scratch_floats(int Cout) { return (long long)C1_WGRAD_MAX_ROWS * Cout * 9; }

// stride 1 data gradient: dx[h][w] (+)= scale * sum_{co,r,s} dy[h+1-r][w+1-s][co] * w[co][r][s] / sigma
// 8 lanes (one 8-channel chunk each) own a horizontal strip of PX output pixels: the 3 x (PX+2) dy pixels the
// strip needs are loaded once and feed up to three outputs each (18 loads per 4 pixels instead of 36: the kernel
// is bound by L2 -> L1 traffic), and the lanes' partial sums are reduced with shuffles.
template <int COUT, int PX>
__global__ __launch_bounds__(SP_THREADS) void conv_c1_dgrad_kernel(const bf16_t* __restrict__ dy, long long lddy,
                                                                   const float* __restrict__ w,
                                                                   const float* __restrict__ sigma,
                                                                   float* __restrict__ dx, int nimg, int H, int W,
                                                                   float scale, int accumulate) {
  constexpr int CC = COUT / 8;
  constexpr int SPB = SP_THREADS / CC;                      // strips per block pass
  __shared__ float wl[9][COUT];
  const float inv = sigma ? 1.f / sigma[0] : 1.f;
  for (int i = threadIdx.x; i < 9 * COUT; i += SP_THREADS) wl[i % 9][i / 9] = w[i] * inv;
  __syncthreads();
  const int cc = threadIdx.x % CC, c = cc * 8;
  const int sl = threadIdx.x / CC;
  const int WS = W / PX;                                    // strips per row (host-checked: W % PX == 0)
  const long long nstrips = (long long)nimg * H * WS;
  const long long send = (nstrips + SPB - 1) / SPB * SPB;   // keep all lanes in the shuffles
  for (long long s = (long long)blockIdx.x * SPB + sl; s < send; s += (long long)gridDim.x * SPB) {
    float acc[PX];
#pragma unroll
    for (int k = 0; k < PX; ++k) acc[k] = 0.f;
    const bool live = s < nstrips;
    const int wx = live ? (int)(s % WS) * PX : 0;
    const long long t = live ? s / WS : 0;
    const int hy = (int)(t % H), n = (int)(t / H);
    // No divergent control flow: pixels outside the image (and the strips past the end, which sit at strip 0) are
    // loaded from a clamped address and replaced by zeros with a select.
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int ho = hy + 1 - r;
      const bool rok = live && (unsigned)ho < (unsigned)H;
      const bf16_t* rowp = dy + ((long long)(n * H + (rok ? ho : hy)) * W) * lddy + c;
#pragma unroll
      for (int j = 0; j < PX + 2; ++j) {
        const int wo = wx - 1 + j;
        const bool ok = rok && (unsigned)wo < (unsigned)W;
        u32x4 raw = *reinterpret_cast<const u32x4*>(rowp + (long long)(ok ? wo : wx) * lddy);
        raw[0] = ok ? raw[0] : 0u; raw[1] = ok ? raw[1] : 0u; raw[2] = ok ? raw[2] : 0u; raw[3] = ok ? raw[3] : 0u;
        float g[8];
        unpack8s(raw, g);
#pragma unroll
        for (int k = 0; k < PX; ++k) {
          const int z = k + 2 - j;                          // dy column wo = (wx + k) + 1 - z
          if (z < 0 || z > 2) continue;                     // (compile-time)
#pragma unroll
          for (int q = 0; q < 8; ++q) acc[k] += g[q] * wl[r * 3 + z][c + q];
        }
      }
    }
    static_assert(CC == 8, "the lane-group sum below is for eight lanes");
#pragma unroll
    for (int k = 0; k < PX; ++k) {      // sum over the strip's eight lanes, DPP only (no LDS-crossbar shuffles)
      acc[k] += dpp_mov<0xB1>(acc[k]);  // quad_perm [1,0,3,2]
      acc[k] += dpp_mov<0x4E>(acc[k]);  // quad_perm [2,3,0,1]
      acc[k] += dpp_mov<0x141>(acc[k]); // row_half_mirror: the other quad of the eight (its lanes all hold its sum)
    }
    if (cc == 0 && live) {
      float* o = dx + (long long)(n * H + hy) * W + wx;
#pragma unroll
      for (int k = 0; k < PX; ++k) o[k] = accumulate ? o[k] + scale * acc[k] : scale * acc[k];
    }
  }
}

extern "C" int ried_conv_c1_dgrad(const void* dy, long long lddy, const float* w, const float* sigma, float* dx,
                                  int nimg, int H, int W, int Cout, float scale, int accumulate, hipStream_t stream) {
  RIED_CHECK_ARG(dy && w && dx && nimg > 0, "ried_conv_c1_dgrad: bad arguments");
  RIED_CHECK_ARG(lddy % 8 == 0 && lddy >= Cout && ((uintptr_t)dy % 16) == 0, "ried_conv_c1_dgrad: dy view");
  RIED_CHECK_ARG(Cout == 64, "ried_conv_c1_dgrad: Cout must be 64");
  long long npix = (long long)nimg * H * W;
  static const bool px1 = getenv("RIED_C1D_PX1") != nullptr;
  if (W % 4 == 0 && !px1)  // (strips of 8 measured slower: 83 vs 64 us on 256 x 64 x 64)
    hipLaunchKernelGGL((conv_c1_dgrad_kernel<64, 4>), dim3(ew_grid(npix / 4, 32, 8192)), dim3(SP_THREADS), 0, stream,
                       (const bf16_t*)dy, lddy, w, sigma, dx, nimg, H, W, scale, accumulate);
  else
    hipLaunchKernelGGL((conv_c1_dgrad_kernel<64, 1>), dim3(ew_grid(npix, 32, 8192)), dim3(SP_THREADS), 0, stream,
                       (const bf16_t*)dy, lddy, w, sigma, dx, nimg, H, W, scale, accumulate);
  RIED_CHECK_LAUNCH("ried_conv_c1_dgrad");
  return RIED_OK;
}

// ---------------------------------------------------------------------------
// generator output conv: out[p] = w[0]*x[p] + sum_c w[1+c]*h[p][c]   (src/rinet.py:70,90), C = 32
// ---------------------------------------------------------------------------
// bn_save / gamma / beta (all or none): h is the raw output of deconv2 and relu(batchnorm(h)) is formed on the fly
// (src/rinet.py:69,195-205), so that the normalised tensor is never written when nothing else reads it.
__global__ void out_conv_fwd_kernel(const float* __restrict__ x, const bf16_t* __restrict__ h, long long ldh,
                                    const float* __restrict__ w, float* __restrict__ out, long long npix,
                                    const float* __restrict__ bn_save, const float* __restrict__ gamma,
                                    const float* __restrict__ beta) {
  // 4 lanes per pixel, 8 channels each; gridDim.y > 1: npix pixels and one statistics set per image
  x += (long long)blockIdx.y * npix;
  out += (long long)blockIdx.y * npix;
  h += (long long)blockIdx.y * npix * ldh;
  if (bn_save) bn_save += blockIdx.y * 64;
  long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long nthreads = (long long)gridDim.x * blockDim.x;
  long long pend = (npix * 4 + nthreads - 1) / nthreads * nthreads;
  const int c = (int)(gid & 3) * 8;                    // nthreads is a multiple of 4: fixed per thread
  float wv[8], a[8], b[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    wv[q] = w[1 + c + q];
    a[q] = bn_save ? gamma[c + q] * bn_save[32 + c + q] : 1.f;
    b[q] = bn_save ? beta[c + q] - bn_save[c + q] * a[q] : 0.f;
  }
  for (long long i = gid; i < pend; i += nthreads) {
    long long p = i >> 2;
    float acc = 0.f;
    if (p < npix) {
      float f[8];
      unpack8s(*reinterpret_cast<const u32x4*>(h + p * ldh + c), f);
      if (bn_save) {
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = fmaxf(f[q] * a[q] + b[q], 0.f);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += f[q] * wv[q];
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if ((i & 3) == 0 && p < npix) out[p] = acc + w[0] * x[p];
  }
}

// dh[p][c] = w[1+c]*dout[p];  dw[0] += sum dout*x;  dw[1+c] += sum dout*h[p][c]
__global__ void out_conv_bwd_kernel(const float* __restrict__ x, const bf16_t* __restrict__ h, long long ldh,
                                    const float* __restrict__ w, const float* __restrict__ dout,
                                    bf16_t* __restrict__ dh, long long lddh, float* __restrict__ dw, long long npix) {
  __shared__ float ldw[33];
  if (threadIdx.x < 33) ldw[threadIdx.x] = 0.f;
  __syncthreads();
  const int c = (threadIdx.x & 3) * 8;
  float wv[8], s[8], s0 = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) { wv[q] = w[1 + c + q]; s[q] = 0.f; }
  long long nthreads = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npix * 4; i += nthreads) {
    long long p = i >> 2;
    float d = dout[p];
    float f[8], o[8];
    unpack8s(*reinterpret_cast<const u32x4*>(h + p * ldh + c), f);
#pragma unroll
    for (int q = 0; q < 8; ++q) { s[q] += d * f[q]; o[q] = d * wv[q]; }
    if (c == 0) s0 += d * x[p];
    *reinterpret_cast<u32x4*>(dh + p * lddh + c) =
        u32x4{pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
  }
  // lanes with equal lane % 4 hold the same channels: butterfly over the other lane bits, one LDS add per wave
#pragma unroll
  for (int q = 0; q < 8; ++q) s[q] = class_sum<4>(s[q]);
  s0 = class_sum<4>(s0);
  if ((threadIdx.x & 63) < 4) {
#pragma unroll
    for (int q = 0; q < 8; ++q) atomicAdd(&ldw[1 + c + q], s[q]);
    if (c == 0) atomicAdd(&ldw[0], s0);
  }
  __syncthreads();
  if (threadIdx.x < 33) atomicAdd(&dw[threadIdx.x], ldw[threadIdx.x]);
}

extern "C" int ried_out_conv_fwd(const float* x, const void* h, long long ldh, const float* w, float* out,
                                 long long npix, const float* bn_save, const float* gamma, const float* beta,
                                 int bn_images, hipStream_t stream) {
  RIED_CHECK_ARG(x && h && w && out && npix > 0 && ldh % 8 == 0 && ldh >= 32 && ((uintptr_t)h % 16) == 0,
                 "ried_out_conv_fwd: bad arguments");
  RIED_CHECK_ARG((!bn_save && !gamma && !beta) || (bn_save && gamma && beta),
                 "ried_out_conv_fwd: bn_save, gamma and beta go together");
  const int ni = bn_images > 1 ? bn_images : 1;       // per-image statistics: bn_save is [bn_images][2][32]
  RIED_CHECK_ARG(ni == 1 || (bn_save && npix % ni == 0 && ni <= 65535), "ried_out_conv_fwd: bad per-image statistics");
  hipLaunchKernelGGL(out_conv_fwd_kernel, dim3(ew_grid(npix / ni * 4, SP_THREADS, 2048), ni), dim3(SP_THREADS), 0,
                     stream, x, (const bf16_t*)h, ldh, w, out, npix / ni, bn_save, gamma, beta);
  RIED_CHECK_LAUNCH("ried_out_conv_fwd");
  return RIED_OK;
}

// dw (33 floats) is accumulated onto; zero it first for a plain gradient.
extern "C" int ried_out_conv_bwd(const float* x, const void* h, long long ldh, const float* w, const float* dout,
                                 void* dh, long long lddh, float* dw, long long npix, hipStream_t stream) {
  RIED_CHECK_ARG(x && h && w && dout && dh && dw && npix > 0 && ldh % 8 == 0 && lddh % 8 == 0 &&
                     ((uintptr_t)h % 16) == 0 && ((uintptr_t)dh % 16) == 0, "ried_out_conv_bwd: bad arguments");
  hipLaunchKernelGGL(out_conv_bwd_kernel, dim3(ew_grid(npix * 4, SP_THREADS * 4, 512)), dim3(SP_THREADS), 0, stream,
                     x, (const bf16_t*)h, ldh, w, dout, (bf16_t*)dh, lddh, dw, npix);
  RIED_CHECK_LAUNCH("ried_out_conv_bwd");
  return RIED_OK;
}

// ---------------------------------------------------------------------------
// spectral normalisation, one power iteration (torch/nn/utils/spectral_norm.py:97-113)
// W: fp32 [R][K] (weight_orig viewed as a matrix).  scratch: fp32 [K + R + 2], zeroed by the first kernel's caller.
// ---------------------------------------------------------------------------
// t[k] = sum_r W[r][k] u[r]: 64 columns per block, the rows split over 4 waves and combined through LDS
__global__ __launch_bounds__(256) void sn_wt_u_kernel(const float* __restrict__ Wm, const float* __restrict__ u, int R,
                                                      int K, float* __restrict__ t) {
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (k < K)
    for (int r = w; r < R; r += 4) acc += Wm[(long long)r * K + k] * u[r];
  sh[w][lane] = acc;
  __syncthreads();
  if (w == 0 && k < K) t[k] = sh[0][lane] + sh[1][lane] + sh[2][lane] + sh[3][lane];
}
// s[r] = sum_k W[r][k] t[k]     (one wave per row)
__global__ void sn_w_v_kernel(const float* __restrict__ Wm, const float* __restrict__ t, int R, int K,
                              float* __restrict__ s) {
  int lane = threadIdx.x & 63;
  int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= R) return;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += Wm[(long long)r * K + k] * t[k];
  acc = wave_sum(acc);
  if (lane == 0) s[r] = acc;
}
// power_iter = 1: v = t/|t|, Wv = s/|t|, u = Wv/|Wv|, sigma = u.Wv
// power_iter = 0: s = W v with the stored v: sigma = u.s with the stored u
__global__ void sn_finish_kernel(const float* __restrict__ t, const float* __restrict__ s, int R, int K, float eps,
                                 int power_iter, float* __restrict__ u, float* __restrict__ v,
                                 float* __restrict__ sigma) {
  __shared__ float sh[4];
  if (power_iter) {
    float a = 0.f, b = 0.f;
    for (int k = threadIdx.x; k < K; k += blockDim.x) a += t[k] * t[k];
    for (int r = threadIdx.x; r < R; r += blockDim.x) b += s[r] * s[r];
    float n2t = block_sum_256(a, sh);
    float n2s = block_sum_256(b, sh);
    float nt = fmaxf(sqrtf(n2t), eps);
    float nwv = sqrtf(n2s) / nt;                  // |W v|
    float nu = fmaxf(nwv, eps);
    for (int k = threadIdx.x; k < K; k += blockDim.x) v[k] = t[k] / nt;
    for (int r = threadIdx.x; r < R; r += blockDim.x) u[r] = (s[r] / nt) / nu;
    if (threadIdx.x == 0) sigma[0] = nwv * nwv / nu;
  } else {
    float acc = 0.f;
    for (int r = threadIdx.x; r < R; r += blockDim.x) acc += u[r] * s[r];
    float d = block_sum_256(acc, sh);
    if (threadIdx.x == 0) sigma[0] = d;
  }
}

extern "C" int ried_sn_sigma(const float* Wm, int R, int K, float* u, float* v, float* sigma, float* scratch,
                             int power_iter, float eps, hipStream_t stream) {
  RIED_CHECK_ARG(Wm && u && v && sigma && scratch && R > 0 && K > 0, "ried_sn_sigma: bad arguments");
  float* t = scratch;
  float* s = scratch + K;
  if (power_iter) {
    hipLaunchKernelGGL(sn_wt_u_kernel, dim3(ceil_div(K, 64)), dim3(256), 0, stream, Wm, u, R, K, t);
    hipLaunchKernelGGL(sn_w_v_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, stream, Wm, t, R, K, s);
  } else {
    hipLaunchKernelGGL(sn_w_v_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, stream, Wm, v, R, K, s);
  }
  hipLaunchKernelGGL(sn_finish_kernel, dim3(1), dim3(256), 0, stream, t, s, R, K, eps, power_iter, u, v, sigma);
  RIED_CHECK_LAUNCH("ried_sn_sigma");
  return RIED_OK;
}

// dW_orig = (dWsn - (sum(dWsn * W)/sigma) * u v^T) / sigma
__global__ void sn_bwd_dot_kernel(const float* __restrict__ dWsn, const float* __restrict__ Wm, long long n,
                                  float* __restrict__ dot) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    acc += dWsn[i] * Wm[i];
  float d = block_sum_256(acc, sh);
  if (threadIdx.x == 0) atomicAdd(dot, d);
}
__global__ void sn_bwd_apply_kernel(const float* __restrict__ dWsn, const float* __restrict__ u,
                                    const float* __restrict__ v, const float* __restrict__ sigma,
                                    const float* __restrict__ dot, int R, int K, float* __restrict__ dW,
                                    int accumulate) {
  long long n = (long long)R * K;
  float sg = sigma[0];
  float coef = dot[0] / sg;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    int r = (int)(i / K), k = (int)(i - (long long)r * K);
    float g = (dWsn[i] - coef * u[r] * v[k]) / sg;
    dW[i] = accumulate ? dW[i] + g : g;
  }
}

extern "C" int ried_sn_backward(const float* dWsn, const float* Wm, const float* u, const float* v,
                                const float* sigma, int R, int K, float* dW, float* scratch1, int accumulate,
                                hipStream_t stream) {
  RIED_CHECK_ARG(dWsn && Wm && u && v && sigma && dW && scratch1 && R > 0 && K > 0, "ried_sn_backward: bad arguments");
  hipError_t e = hipMemsetAsync(scratch1, 0, sizeof(float), stream);
  if (e != hipSuccess) { ried_set_error("ried_sn_backward: memset failed"); return RIED_ERR_HIP; }
  long long n = (long long)R * K;
  hipLaunchKernelGGL(sn_bwd_dot_kernel, dim3(ew_grid(n, 1024, 64)), dim3(256), 0, stream, dWsn, Wm, n, scratch1);
  hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(ew_grid(n, 256, 2048)), dim3(256), 0, stream, dWsn, u, v, sigma,
                     scratch1, R, K, dW, accumulate);
  RIED_CHECK_LAUNCH("ried_sn_backward");
  return RIED_OK;
}

// ---------------------------------------------------------------------------
// Job-table forms of the two functions above: every spectral-norm layer of a network in one launch per phase
// (3 launches forward, 2 backward, instead of 3 and 3 per layer).
//   job (10 x 64-bit words): W, u, v, sigma, t (scratch [K]), s (scratch [R]), R, K, dWsn, dW
//   maps: (job, chunk) pairs per block; cols: 64 columns of W per block; rows: 4 rows per block;
//   dot: one of SN_DOT_PARTS slices of R*K per block; apply: 2048 elements per block
// ---------------------------------------------------------------------------
#define SN_JOB_WORDS 10
#define SN_DOT_PARTS 64
__global__ __launch_bounds__(256) void sn_jobs_wt_u_kernel(const long long* __restrict__ jobs,
                                                           const int* __restrict__ map) {
  const long long* J = jobs + (long long)map[2 * blockIdx.x] * SN_JOB_WORDS;
  const float* Wm = (const float*)J[0];
  const float* u = (const float*)J[1];
  float* t = (float*)J[4];
  const int R = (int)J[6], K = (int)J[7];
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int k = map[2 * blockIdx.x + 1] * 64 + lane;
  float acc = 0.f;
  if (k < K) {
#pragma unroll 8
    for (int r = w; r < R; r += 4) acc += Wm[(long long)r * K + k] * u[r];
  }
  sh[w][lane] = acc;
  __syncthreads();
  if (w == 0 && k < K) t[k] = sh[0][lane] + sh[1][lane] + sh[2][lane] + sh[3][lane];
}
__global__ __launch_bounds__(256) void sn_jobs_w_v_kernel(const long long* __restrict__ jobs,
                                                          const int* __restrict__ map, int power_iter) {
  const long long* J = jobs + (long long)map[2 * blockIdx.x] * SN_JOB_WORDS;
  const float* Wm = (const float*)J[0];
  const float* t = power_iter ? (const float*)J[4] : (const float*)J[2];
  float* s = (float*)J[5];
  const int R = (int)J[6], K = (int)J[7];
  const int lane = threadIdx.x & 63;
  const int r = map[2 * blockIdx.x + 1] * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float acc = 0.f;
#pragma unroll 4
  for (int k = lane; k < K; k += 64) acc += Wm[(long long)r * K + k] * t[k];
  acc = wave_sum(acc);
  if (lane == 0) s[r] = acc;
}
__global__ __launch_bounds__(256) void sn_jobs_finish_kernel(const long long* __restrict__ jobs, float eps,
                                                             int power_iter) {
  const long long* J = jobs + (long long)blockIdx.x * SN_JOB_WORDS;
  float* u = (float*)J[1];
  float* v = (float*)J[2];
  float* sigma = (float*)J[3];
  const float* t = (const float*)J[4];
  const float* s = (const float*)J[5];
  const int R = (int)J[6], K = (int)J[7];
  __shared__ float sh[4];
  if (power_iter) {
    float a = 0.f, b = 0.f;
    for (int k = threadIdx.x; k < K; k += blockDim.x) a += t[k] * t[k];
    for (int r = threadIdx.x; r < R; r += blockDim.x) b += s[r] * s[r];
    float n2t = block_sum_256(a, sh);
    float n2s = block_sum_256(b, sh);
    float nt = fmaxf(sqrtf(n2t), eps);
    float nwv = sqrtf(n2s) / nt;
    float nu = fmaxf(nwv, eps);
    for (int k = threadIdx.x; k < K; k += blockDim.x) v[k] = t[k] / nt;
    for (int r = threadIdx.x; r < R; r += blockDim.x) u[r] = (s[r] / nt) / nu;
    if (threadIdx.x == 0) sigma[0] = nwv * nwv / nu;
  } else {
    float acc = 0.f;
    for (int r = threadIdx.x; r < R; r += blockDim.x) acc += u[r] * s[r];
    float d = block_sum_256(acc, sh);
    if (threadIdx.x == 0) sigma[0] = d;
  }
}

extern "C" int ried_sn_sigma_jobs(const long long* jobs, int njobs, const int* map_cols, int nblocks_cols,
                                  const int* map_rows, int nblocks_rows, int power_iter, float eps,
                                  hipStream_t stream) {
  RIED_CHECK_ARG(jobs && njobs > 0 && map_cols && map_rows && nblocks_cols > 0 && nblocks_rows > 0,
                 "ried_sn_sigma_jobs: bad arguments");
  if (power_iter)
    hipLaunchKernelGGL(sn_jobs_wt_u_kernel, dim3(nblocks_cols), dim3(256), 0, stream, jobs, map_cols);
  hipLaunchKernelGGL(sn_jobs_w_v_kernel, dim3(nblocks_rows), dim3(256), 0, stream, jobs, map_rows, power_iter);
  hipLaunchKernelGGL(sn_jobs_finish_kernel, dim3(njobs), dim3(256), 0, stream, jobs, eps, power_iter);
  RIED_CHECK_LAUNCH("ried_sn_sigma_jobs");
  return RIED_OK;
}

// partial[job][part] = sum over the part's slice of dWsn * W  (t, the [K] scratch, holds the SN_DOT_PARTS partials)
__global__ __launch_bounds__(256) void sn_jobs_bwd_dot_kernel(const long long* __restrict__ jobs,
                                                              const int* __restrict__ map) {
  const long long* J = jobs + (long long)map[2 * blockIdx.x] * SN_JOB_WORDS;
  const float* Wm = (const float*)J[0];
  const float* dWsn = (const float*)J[8];
  float* part = (float*)J[4];
  const long long n = J[6] * J[7];
  const int p = map[2 * blockIdx.x + 1];
  const long long per = (n + SN_DOT_PARTS - 1) / SN_DOT_PARTS;
  const long long lo = p * per, hi = lo + per < n ? lo + per : n;
  __shared__ float sh[4];
  float acc = 0.f;
#pragma unroll 4
  for (long long i = lo + threadIdx.x; i < hi; i += 256) acc += dWsn[i] * Wm[i];
  float d = block_sum_256(acc, sh);
  if (threadIdx.x == 0) part[p] = d;
}
__global__ __launch_bounds__(256) void sn_jobs_bwd_apply_kernel(const long long* __restrict__ jobs,
                                                                const int* __restrict__ map, int accumulate) {
  const long long* J = jobs + (long long)map[2 * blockIdx.x] * SN_JOB_WORDS;
  const float* u = (const float*)J[1];
  const float* v = (const float*)J[2];
  const float* part = (const float*)J[4];
  const float* dWsn = (const float*)J[8];
  float* dW = (float*)J[9];
  const int K = (int)J[7];
  const long long n = J[6] * J[7];
  __shared__ float sh[4];
  float d = block_sum_256(threadIdx.x < SN_DOT_PARTS ? part[threadIdx.x] : 0.f, sh);
  const float sg = ((const float*)J[3])[0];
  const float coef = d / sg;
  const long long i0 = (long long)map[2 * blockIdx.x + 1] * 2048 + threadIdx.x;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    long long i = i0 + q * 256;
    if (i >= n) break;
    int r = (int)(i / K), k = (int)(i - (long long)r * K);
    float g = (dWsn[i] - coef * u[r] * v[k]) / sg;
    dW[i] = accumulate ? dW[i] + g : g;
  }
}

extern "C" int ried_sn_backward_jobs(const long long* jobs, const int* map_dot, int nblocks_dot, const int* map_apply,
                                     int nblocks_apply, int accumulate, hipStream_t stream) {
  RIED_CHECK_ARG(jobs && map_dot && map_apply && nblocks_dot > 0 && nblocks_apply > 0,
                 "ried_sn_backward_jobs: bad arguments");
  hipLaunchKernelGGL(sn_jobs_bwd_dot_kernel, dim3(nblocks_dot), dim3(256), 0, stream, jobs, map_dot);
  hipLaunchKernelGGL(sn_jobs_bwd_apply_kernel, dim3(nblocks_apply), dim3(256), 0, stream, jobs, map_apply, accumulate);
  RIED_CHECK_LAUNCH("ried_sn_backward_jobs");
  return RIED_OK;
}

// ---------------------------------------------------------------------------
// perceptual-loss tap: loss[0] += loss_scale * sum (a-b)^2 ;  g (+)= grad_scale * (a-b)
// a = features of the prediction, b = features of the target (src/loss.py:184-189)
// ---------------------------------------------------------------------------
__global__ void mse_tap_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n8,
                               float loss_scale, float* __restrict__ loss, bf16_t* __restrict__ g, float grad_scale,
                               int accumulate) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    f32x4 a0 = *reinterpret_cast<const f32x4*>(a + i * 8), a1 = *reinterpret_cast<const f32x4*>(a + i * 8 + 4);
    f32x4 b0 = *reinterpret_cast<const f32x4*>(b + i * 8), b1 = *reinterpret_cast<const f32x4*>(b + i * 8 + 4);
    float fa[8] = {a0[0] - b0[0], a0[1] - b0[1], a0[2] - b0[2], a0[3] - b0[3],
                   a1[0] - b1[0], a1[1] - b1[1], a1[2] - b1[2], a1[3] - b1[3]};
    float fg[8];
    if (g && accumulate) unpack8s(*reinterpret_cast<const u32x4*>(g + i * 8), fg);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float d = fa[q];
      acc += d * d;
      fa[q] = grad_scale * d + ((g && accumulate) ? fg[q] : 0.f);
    }
    if (g)
      *reinterpret_cast<u32x4*>(g + i * 8) =
          u32x4{pack2bf(fa[0], fa[1]), pack2bf(fa[2], fa[3]), pack2bf(fa[4], fa[5]), pack2bf(fa[6], fa[7])};
  }
  float t = block_sum_256(acc, sh);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, t * loss_scale);
}

// a, b: contiguous fp32 arrays of n elements (n % 8 == 0); g: bf16 [n] or null
extern "C" int ried_mse_tap(const void* a, const void* b, long long n, float loss_scale, float* loss, void* g,
                            float grad_scale, int accumulate, hipStream_t stream) {
  RIED_CHECK_ARG(a && b && n > 0 && n % 8 == 0, "ried_mse_tap: bad arguments");
  RIED_CHECK_ARG(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 && ((uintptr_t)g % 16) == 0,
                 "ried_mse_tap: alignment");
  hipLaunchKernelGGL(mse_tap_kernel, dim3(ew_grid(n / 8, SP_THREADS * 2, 256)), dim3(SP_THREADS), 0, stream,
                     (const float*)a, (const float*)b, n / 8, loss_scale, loss, (bf16_t*)g, grad_scale, accumulate);
  RIED_CHECK_LAUNCH("ried_mse_tap");
  return RIED_OK;
}

// MPL tap (src/loss.py:101-109): loss[0] += loss_scale * sum over pixels of ||a[p][:] - b[p][:]||_2 ;
// g[p][c] = grad_scale * (a - b)[p][c] / ||a[p] - b[p]||  (0 where the norm is 0).  a, b: fp32 [npix][C].
// C/8 lanes (a power of two, at most 64) share a pixel; the groups of a wave work on different pixels.
__global__ __launch_bounds__(SP_THREADS) void l2norm_tap_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                long long npix, int C, float loss_scale,
                                                                float* __restrict__ loss, bf16_t* __restrict__ g,
                                                                float grad_scale) {
  __shared__ float sh[4];
  const int L = C >> 3, gl = threadIdx.x & (L - 1), pg = threadIdx.x / L, PPB = SP_THREADS / L;
  float acc = 0.f;
  for (long long base = (long long)blockIdx.x * PPB; base < npix; base += (long long)gridDim.x * PPB) {
    const long long p = base + pg;
    const bool valid = p < npix;
    float d[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) d[q] = 0.f;
    if (valid) {
      const long long o = p * C + gl * 8;
      f32x4 a0 = *reinterpret_cast<const f32x4*>(a + o), a1 = *reinterpret_cast<const f32x4*>(a + o + 4);
      f32x4 b0 = *reinterpret_cast<const f32x4*>(b + o), b1 = *reinterpret_cast<const f32x4*>(b + o + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) { d[q] = a0[q] - b0[q]; d[4 + q] = a1[q] - b1[q]; }
    }
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) ss += d[q] * d[q];
    for (int o = L >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float nrm = sqrtf(ss);
    if (valid && gl == 0) acc += nrm;
    if (valid && g) {
      const float s = nrm > 0.f ? grad_scale / nrm : 0.f;
      *reinterpret_cast<u32x4*>(g + p * C + gl * 8) = u32x4{pack2bf(d[0] * s, d[1] * s), pack2bf(d[2] * s, d[3] * s),
                                                            pack2bf(d[4] * s, d[5] * s), pack2bf(d[6] * s, d[7] * s)};
    }
  }
  float t = block_sum_256(acc, sh);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, t * loss_scale);
}

extern "C" int ried_l2norm_tap(const void* a, const void* b, long long npix, int C, float loss_scale, float* loss,
                               void* g, float grad_scale, hipStream_t stream) {
  RIED_CHECK_ARG(a && b && npix > 0, "ried_l2norm_tap: bad arguments");
  RIED_CHECK_ARG(C >= 8 && C <= 512 && (C & (C - 1)) == 0, "ried_l2norm_tap: C must be a power of two in [8, 512]");
  RIED_CHECK_ARG(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 && ((uintptr_t)g % 16) == 0,
                 "ried_l2norm_tap: alignment");
  const int ppb = SP_THREADS / (C / 8);
  hipLaunchKernelGGL(l2norm_tap_kernel, dim3(ew_grid(npix, ppb * 2, 1024)), dim3(SP_THREADS), 0, stream,
                     (const float*)a, (const float*)b, npix, C, loss_scale, loss, (bf16_t*)g, grad_scale);
  RIED_CHECK_LAUNCH("ried_l2norm_tap");
  return RIED_OK;
}

// ---------------------------------------------------------------------------
// loss scalars (src/loss.py:141-142, 154-155, 194).  sc is a small fp32 device array:
//  in : sc[0] = mean(pos)  sc[1] = mean(neg)
//  out: sc[2] = dloss = relu(1-mean(pos)) + relu(1+mean(neg))
//       sc[3] = d dloss / d pos_i = -[1-mean(pos) > 0]/n     sc[4] = d dloss / d neg_i = +[1+mean(neg) > 0]/n
//       keep[0] = dloss too when keep is given (the step keeps both iterations' values)
// ---------------------------------------------------------------------------
__global__ void dloss_scalars_kernel(float* sc, float inv_n, float* keep) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float a = 1.f - sc[0], b = 1.f + sc[1];
    sc[2] = fmaxf(a, 0.f) + fmaxf(b, 0.f);
    if (keep) keep[0] = sc[2];
    sc[3] = a > 0.f ? -inv_n : 0.f;
    sc[4] = b > 0.f ? inv_n : 0.f;
  }
}
extern "C" int ried_dloss_scalars(float* sc, long long n_per_half, float* keep, hipStream_t stream) {
  RIED_CHECK_ARG(sc && n_per_half > 0, "ried_dloss_scalars: bad arguments");
  hipLaunchKernelGGL(dloss_scalars_kernel, dim3(1), dim3(64), 0, stream, sc, 1.f / (float)n_per_half, keep);
  RIED_CHECK_LAUNCH("ried_dloss_scalars");
  return RIED_OK;
}

//  in : sc[1] = mean(neg), sc[5] = perceptual mean-of-5 MSE, sc[6] = mean SSIM
//  out: sc[7] = dssim = (1-ssim)/2, sc[8] = gen = 1-mean(neg), sc[9] = gloss = wv*sc[5] + ws*sc[7] + wg*sc[8]
__global__ void gloss_scalars_kernel(float* sc, float wv, float ws, float wg) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    sc[7] = (1.f - sc[6]) * 0.5f;
    sc[8] = 1.f - sc[1];
    sc[9] = wv * sc[5] + ws * sc[7] + wg * sc[8];
  }
}
extern "C" int ried_gloss_scalars(float* sc, float wv, float ws, float wg, hipStream_t stream) {
  RIED_CHECK_ARG(sc != nullptr, "ried_gloss_scalars: bad arguments");
  hipLaunchKernelGGL(gloss_scalars_kernel, dim3(1), dim3(64), 0, stream, sc, wv, ws, wg);
  RIED_CHECK_LAUNCH("ried_gloss_scalars");
  return RIED_OK;
}

// out[p][c] = (s ? s[0] : 1) * value
__global__ void fill_kernel(bf16_t* __restrict__ out, long long ldo, long long npix, int C, const float* s,
                            float value) {
  const int CC = C >> 3;
  float v = (s ? s[0] : 1.f) * value;
  unsigned pk = pack2bf(v, v);
  u32x4 vv{pk, pk, pk, pk};
  long long total = npix * CC;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long p = i / CC;
    int c = (int)(i - p * CC) << 3;
    *reinterpret_cast<u32x4*>(out + p * ldo + c) = vv;
  }
}
extern "C" int ried_fill(void* out, long long ldo, long long npix, int C, const float* s, float value,
                         hipStream_t stream) {
  RIED_CHECK_ARG(out && npix > 0 && C % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)out % 16) == 0,
                 "ried_fill: bad arguments");
  hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(npix * (C / 8), SP_THREADS, 2048)), dim3(SP_THREADS), 0, stream,
                     (bf16_t*)out, ldo, npix, C, s, value);
  RIED_CHECK_LAUNCH("ried_fill");
  return RIED_OK;
}

// out[0] += scale * sum_i f(in[i*stride]), f = identity or square
__global__ void sum_f32_kernel(const float* __restrict__ in, long long n, long long stride, float scale, int square,
                               float* __restrict__ out) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = in[i * stride];
    acc += square ? v * v : v;
  }
  float t = block_sum_256(acc, sh);
  if (threadIdx.x == 0) atomicAdd(out, t * scale);
}
extern "C" int ried_sum_f32(const float* in, long long n, long long stride, float scale, int square, float* out,
                            hipStream_t stream) {
  RIED_CHECK_ARG(in && out && n > 0 && stride > 0, "ried_sum_f32: bad arguments");
  hipLaunchKernelGGL(sum_f32_kernel, dim3(ew_grid(n, 256 * 16, 256)), dim3(256), 0, stream, in, n, stride, scale,
                     square, out);
  RIED_CHECK_LAUNCH("ried_sum_f32");
  return RIED_OK;
}
