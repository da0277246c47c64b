// Kernels that only the legacy networks of src/networks.py:460-556 (GNet, DNet) need beside the shared ones:
//  * GNet's last layer, Conv2d(192, 1, 3, padding=1) with bias (src/networks.py:485,519): one output channel is
//    a dot product per pixel, not a GEMM - bandwidth bound, result kept in fp32 because it is the HU image;
//  * the data gradient of a stride-2 one-input-channel 3x3 conv (DNet.conv1, src/networks.py:529);
//  * y += alpha * x on fp32 vectors (the conv bias a train-mode BatchNorm removes still shifts running_mean).
#include "common.h"

#define LG_THREADS 256

__device__ __forceinline__ void lg_unpack8(const u32x4& v, float (&f)[8]) {
  f[0] = bflo(v[0]); f[1] = bfhi(v[0]); f[2] = bflo(v[1]); f[3] = bfhi(v[1]);
  f[4] = bflo(v[2]); f[5] = bfhi(v[2]); f[6] = bflo(v[3]); f[7] = bfhi(v[3]);
}

// master weight (1, C, 3, 3) = w[c*9 + t]  ->  LDS [t][C]
__device__ __forceinline__ void lg_stage_weights(const float* __restrict__ w, int C, float* wl) {
  for (int i = threadIdx.x; i < 9 * C; i += LG_THREADS) wl[(i % 9) * C + i / 9] = w[i];
  __syncthreads();
}

// ---------------------------------------------------------------------------
// out[n][h][w] = bias + sum_{r,z,c} x[n][h-1+r][w-1+z][c] * w[c][r][z]
// 8 lanes per output pixel: lane j takes the 16-byte chunks j, j+8, ... of each of the 9 input pixels (8 lanes
// read 128 contiguous bytes), the partial sums meet through three shuffles.  The 9 reads per input pixel come
// from L1 / L2 (neighbouring output pixels sit in the same wave).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(LG_THREADS) void conv3_to1_fwd_kernel(const bf16_t* __restrict__ x, long long ldx,
                                                                   const float* __restrict__ w,
                                                                   const float* __restrict__ bias,
                                                                   float* __restrict__ out, int nimg, int H, int W,
                                                                   int C) {
  extern __shared__ float wl[];
  lg_stage_weights(w, C, wl);
  const int j = threadIdx.x & 7, pl = threadIdx.x >> 3;
  constexpr int PPB = LG_THREADS / 8;
  const long long npix = (long long)nimg * H * W;
  const long long pend = (npix + PPB - 1) / PPB * PPB;              // keep every lane in the shuffles
  const float b = bias ? bias[0] : 0.f;
  for (long long p = (long long)blockIdx.x * PPB + pl; p < pend; p += (long long)gridDim.x * PPB) {
    float acc = 0.f;
    if (p < npix) {
      const int wx = (int)(p % W);
      const long long t = p / W;
      const int hy = (int)(t % H);
      const long long n = t / H;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int hh = hy - 1 + r;
        if ((unsigned)hh >= (unsigned)H) continue;
#pragma unroll
        for (int z = 0; z < 3; ++z) {
          const int ww = wx - 1 + z;
          if ((unsigned)ww >= (unsigned)W) continue;
          const bf16_t* px = x + ((n * H + hh) * W + ww) * ldx;
          const float* wt = wl + (r * 3 + z) * C;
          for (int c = j * 8; c < C; c += 64) {
            float f[8];
            lg_unpack8(*reinterpret_cast<const u32x4*>(px + c), f);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt + c), w1 = *reinterpret_cast<const f32x4*>(wt + c + 4);
            acc += f[0] * w0[0] + f[1] * w0[1] + f[2] * w0[2] + f[3] * w0[3] + f[4] * w1[0] + f[5] * w1[1] +
                   f[6] * w1[2] + f[7] * w1[3];
          }
        }
      }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (j == 0 && p < npix) out[p] = acc + b;
  }
}

// dx[n][h][w][c] (+)= sum_{r,z} dout[n][h+1-r][w+1-z] * w[c][r][z]
__global__ __launch_bounds__(LG_THREADS) void conv3_to1_dgrad_kernel(const float* __restrict__ dout,
                                                                     const float* __restrict__ w,
                                                                     bf16_t* __restrict__ dx, long long lddx,
                                                                     int nimg, int H, int W, int C, int accumulate) {
  extern __shared__ float wl[];
  lg_stage_weights(w, C, wl);
  const int j = threadIdx.x & 7, pl = threadIdx.x >> 3;
  constexpr int PPB = LG_THREADS / 8;
  const long long npix = (long long)nimg * H * W;
  for (long long p = (long long)blockIdx.x * PPB + pl; p < npix; p += (long long)gridDim.x * PPB) {
    const int wx = (int)(p % W);
    const long long t = p / W;
    const int hy = (int)(t % H);
    const long long n = t / H;
    float d[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int ho = hy + 1 - r;
#pragma unroll
      for (int z = 0; z < 3; ++z) {
        const int wo = wx + 1 - z;
        const bool ok = (unsigned)ho < (unsigned)H && (unsigned)wo < (unsigned)W;
        d[r * 3 + z] = ok ? dout[(n * H + ho) * W + wo] : 0.f;
      }
    }
    bf16_t* row = dx + p * lddx;
    for (int c = j * 8; c < C; c += 64) {
      float s[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) s[q] = 0.f;
      if (accumulate) lg_unpack8(*reinterpret_cast<const u32x4*>(row + c), s);
#pragma unroll
      for (int tt = 0; tt < 9; ++tt) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wl + tt * C + c);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(wl + tt * C + c + 4);
        s[0] += d[tt] * w0[0]; s[1] += d[tt] * w0[1]; s[2] += d[tt] * w0[2]; s[3] += d[tt] * w0[3];
        s[4] += d[tt] * w1[0]; s[5] += d[tt] * w1[1]; s[6] += d[tt] * w1[2]; s[7] += d[tt] * w1[3];
      }
      *reinterpret_cast<u32x4*>(row + c) =
          u32x4{pack2bf(s[0], s[1]), pack2bf(s[2], s[3]), pack2bf(s[4], s[5]), pack2bf(s[6], s[7])};
    }
  }
}

// dw[c][r][z] += sum_{n,h,w} x[n][h][w][c] * dout[n][h+1-r][w+1-z]
// One 8-channel chunk per thread, C/8 threads per pixel, 256 / (C/8) pixels per block pass; the per-thread sums
// (72 registers) meet in an LDS copy of the gradient, which the block then adds to the global one.
__global__ __launch_bounds__(LG_THREADS) void conv3_to1_wgrad_kernel(const bf16_t* __restrict__ x, long long ldx,
                                                                     const float* __restrict__ dout,
                                                                     float* __restrict__ dw, int nimg, int H, int W,
                                                                     int C) {
  extern __shared__ float sums[];                                   // [9][C]
  for (int i = threadIdx.x; i < 9 * C; i += LG_THREADS) sums[i] = 0.f;
  __syncthreads();
  const int CC = C >> 3, PPB = LG_THREADS / CC;
  const int pl = threadIdx.x / CC, c = (threadIdx.x - pl * CC) * 8;
  const long long npix = (long long)nimg * H * W;
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[t][q] = 0.f;
  if (pl < PPB) {
    for (long long p = (long long)blockIdx.x * PPB + pl; p < npix; p += (long long)gridDim.x * PPB) {
      const int wx = (int)(p % W);
      const long long t = p / W;
      const int hy = (int)(t % H);
      const long long n = t / H;
      float f[8];
      lg_unpack8(*reinterpret_cast<const u32x4*>(x + p * ldx + c), f);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int ho = hy + 1 - r;
#pragma unroll
        for (int z = 0; z < 3; ++z) {
          const int wo = wx + 1 - z;
          const bool ok = (unsigned)ho < (unsigned)H && (unsigned)wo < (unsigned)W;
          const float d = ok ? dout[(n * H + ho) * W + wo] : 0.f;
#pragma unroll
          for (int q = 0; q < 8; ++q) acc[r * 3 + z][q] += d * f[q];
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int q = 0; q < 8; ++q) atomicAdd(&sums[t * C + c + q], acc[t][q]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 9 * C; i += LG_THREADS) atomicAdd(&dw[i], sums[(i % 9) * C + i / 9]);
}

#define LG_CHECK(name)                                                                                        \
  RIED_CHECK_ARG(nimg > 0 && H > 0 && W > 0 && C >= 8 && C % 8 == 0 && C <= 1024, name ": bad dimensions");
static inline size_t lg_smem(int C) { return (size_t)9 * C * sizeof(float); }

extern "C" int ried_conv3_to1_fwd(const void* x, long long ldx, const float* w, const float* bias, float* out,
                                  int nimg, int H, int W, int C, hipStream_t stream) {
  LG_CHECK("ried_conv3_to1_fwd");
  RIED_CHECK_ARG(x && w && out && ldx % 8 == 0 && ldx >= C && ((uintptr_t)x % 16) == 0, "ried_conv3_to1_fwd: x view");
  long long npix = (long long)nimg * H * W;
  hipLaunchKernelGGL(conv3_to1_fwd_kernel, dim3(ew_grid(npix, LG_THREADS / 8 * 4, 8192)), dim3(LG_THREADS), lg_smem(C),
                     stream, (const bf16_t*)x, ldx, w, bias, out, nimg, H, W, C);
  RIED_CHECK_LAUNCH("ried_conv3_to1_fwd");
  return RIED_OK;
}

extern "C" int ried_conv3_to1_dgrad(const float* dout, const float* w, void* dx, long long lddx, int nimg, int H,
                                    int W, int C, int accumulate, hipStream_t stream) {
  LG_CHECK("ried_conv3_to1_dgrad");
  RIED_CHECK_ARG(dout && w && dx && lddx % 8 == 0 && lddx >= C && ((uintptr_t)dx % 16) == 0,
                 "ried_conv3_to1_dgrad: dx view");
  long long npix = (long long)nimg * H * W;
  hipLaunchKernelGGL(conv3_to1_dgrad_kernel, dim3(ew_grid(npix, LG_THREADS / 8 * 4, 8192)), dim3(LG_THREADS),
                     lg_smem(C), stream, dout, w, (bf16_t*)dx, lddx, nimg, H, W, C, accumulate);
  RIED_CHECK_LAUNCH("ried_conv3_to1_dgrad");
  return RIED_OK;
}

extern "C" int ried_conv3_to1_wgrad(const void* x, long long ldx, const float* dout, float* dw, int nimg, int H, int W,
                                    int C, hipStream_t stream) {
  LG_CHECK("ried_conv3_to1_wgrad");
  RIED_CHECK_ARG(x && dout && dw && ldx % 8 == 0 && ldx >= C && ((uintptr_t)x % 16) == 0, "ried_conv3_to1_wgrad: x view");
  RIED_CHECK_ARG(C <= 8 * LG_THREADS, "ried_conv3_to1_wgrad: C too large");
  long long npix = (long long)nimg * H * W;
  int ppb = LG_THREADS / (C / 8);
  hipLaunchKernelGGL(conv3_to1_wgrad_kernel, dim3(ew_grid(npix, ppb * 64, 256)), dim3(LG_THREADS), lg_smem(C), stream,
                     (const bf16_t*)x, ldx, dout, dw, nimg, H, W, C);
  RIED_CHECK_LAUNCH("ried_conv3_to1_wgrad");
  return RIED_OK;
}

// ---------------------------------------------------------------------------
// stride-2 data gradient of the one-input-channel 3x3 pad-1 conv:
// dx[n][h][w] (+)= scale * sum over (r, z) with h+1-r and w+1-z even of  dy[n][(h+1-r)/2][(w+1-z)/2][:] . w[:][r][z]
// 8 lanes (one 8-channel chunk each) per image pixel, at most 4 contributing output pixels.
// ---------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(LG_THREADS) void conv_c1_dgrad_s2_kernel(const bf16_t* __restrict__ dy, long long lddy,
                                                                      const float* __restrict__ w,
                                                                      const float* __restrict__ sigma,
                                                                      float* __restrict__ dx, int nimg, int H, int W,
                                                                      int Ho, int Wo, float scale, int accumulate) {
  static_assert(COUT == 64, "8 lanes of 8 channels");
  __shared__ float wl[9][COUT];
  const float inv = sigma ? 1.f / sigma[0] : 1.f;
  for (int i = threadIdx.x; i < 9 * COUT; i += LG_THREADS) wl[i % 9][i / 9] = w[i] * inv;
  __syncthreads();
  const int j = threadIdx.x & 7, pl = threadIdx.x >> 3, c = j * 8;
  constexpr int PPB = LG_THREADS / 8;
  const long long npix = (long long)nimg * H * W;
  const long long pend = (npix + PPB - 1) / PPB * PPB;
  for (long long p = (long long)blockIdx.x * PPB + pl; p < pend; p += (long long)gridDim.x * PPB) {
    float acc = 0.f;
    if (p < npix) {
      const int wx = (int)(p % W);
      const long long t = p / W;
      const int hy = (int)(t % H);
      const long long n = t / H;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int a = hy + 1 - r;
        if ((a & 1) || (unsigned)(a >> 1) >= (unsigned)Ho) continue;
#pragma unroll
        for (int z = 0; z < 3; ++z) {
          const int b = wx + 1 - z;
          if ((b & 1) || (unsigned)(b >> 1) >= (unsigned)Wo) continue;
          float g[8];
          lg_unpack8(*reinterpret_cast<const u32x4*>(dy + ((n * Ho + (a >> 1)) * Wo + (b >> 1)) * lddy + c), g);
#pragma unroll
          for (int q = 0; q < 8; ++q) acc += g[q] * wl[r * 3 + z][c + q];
        }
      }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (j == 0 && p < npix) dx[p] = accumulate ? dx[p] + scale * acc : scale * acc;
  }
}

extern "C" int ried_conv_c1_dgrad_s2(const void* dy, long long lddy, const float* w, const float* sigma, float* dx,
                                     int nimg, int H, int W, int Cout, float scale, int accumulate,
                                     hipStream_t stream) {
  RIED_CHECK_ARG(dy && w && dx && nimg > 0 && H > 0 && W > 0, "ried_conv_c1_dgrad_s2: bad arguments");
  RIED_CHECK_ARG(lddy % 8 == 0 && lddy >= Cout && ((uintptr_t)dy % 16) == 0, "ried_conv_c1_dgrad_s2: dy view");
  RIED_CHECK_ARG(Cout == 64, "ried_conv_c1_dgrad_s2: Cout must be 64");
  int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  long long npix = (long long)nimg * H * W;
  hipLaunchKernelGGL(conv_c1_dgrad_s2_kernel<64>, dim3(ew_grid(npix, LG_THREADS / 8 * 4, 8192)), dim3(LG_THREADS), 0,
                     stream, (const bf16_t*)dy, lddy, w, sigma, dx, nimg, H, W, Ho, Wo, scale, accumulate);
  RIED_CHECK_LAUNCH("ried_conv_c1_dgrad_s2");
  return RIED_OK;
}

// ---------------------------------------------------------------------------
__global__ void axpy_f32_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] += alpha * x[i];
}
extern "C" int ried_axpy_f32(float* y, const float* x, float alpha, long long n, hipStream_t stream) {
  RIED_CHECK_ARG(y && x && n > 0, "ried_axpy_f32: bad arguments");
  hipLaunchKernelGGL(axpy_f32_kernel, dim3(ew_grid(n, 256, 1024)), dim3(256), 0, stream, y, x, alpha, n);
  RIED_CHECK_LAUNCH("ried_axpy_f32");
  return RIED_OK;
}
