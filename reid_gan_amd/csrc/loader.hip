// Training-patch pipeline on the device (reference: src/loader.py:60-73 get_patch, src/train.py:136-142).
// The int16 HU slices of the training set stay resident in HBM; one launch cuts patch_n random aligned crops out
// of each selected slice pair and converts them to fp32, so no host worker touches pixel data during training.
#include "common.h"

// splitmix64 of (seed, draw): the crop position of patch `draw` when the caller supplies no offsets
__device__ __forceinline__ unsigned long long patch_hash(unsigned long long seed, unsigned long long draw) {
  unsigned long long z = seed + (draw + 1ull) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// grid: (patch, chunk of 256 x 4 elements).  Rows of a patch are contiguous runs of P int16 at an arbitrary
// 2-byte alignment: four scalar loads per thread, one 16-byte (fp32) or 8-byte (int16) store.
template <bool F32>
__global__ __launch_bounds__(256) void patch_crop_kernel(const short* __restrict__ sin, const short* __restrict__ stg,
                                                         int H, int W, int first_slice, int patch_n, int P,
                                                         const int* __restrict__ offsets, unsigned long long seed,
                                                         unsigned long long first_draw, void* __restrict__ xo,
                                                         void* __restrict__ yo, int* __restrict__ offsets_out) {
  const int j = blockIdx.x;
  const int slice = first_slice + j / patch_n;
  int top, left;
  if (offsets) {
    top = min(max(offsets[2 * j], 0), H - P);          // never read outside the slice, whatever the caller passed
    left = min(max(offsets[2 * j + 1], 0), W - P);
  } else {
    // numpy.random.randint(0, h - p): uniform on [0, h - p), the last row / column is never the first of a patch
    const unsigned long long z = patch_hash(seed, first_draw + (unsigned long long)j);
    top = (int)(((z >> 32) * (unsigned long long)(H - P)) >> 32);
    left = (int)(((z & 0xffffffffull) * (unsigned long long)(W - P)) >> 32);
  }
  if (offsets_out && blockIdx.y == 0 && threadIdx.x == 0) { offsets_out[2 * j] = top; offsets_out[2 * j + 1] = left; }
  const int q4 = P >> 2;
  const int e = blockIdx.y * 256 + threadIdx.x;
  if (e >= P * q4) return;
  const int r = e / q4, c = (e - r * q4) << 2;
  const long long src = ((long long)slice * H + top + r) * W + left + c;
  const long long dst = ((long long)j * P + r) * P + c;
  const short a0 = sin[src], a1 = sin[src + 1], a2 = sin[src + 2], a3 = sin[src + 3];
  const short b0 = stg[src], b1 = stg[src + 1], b2 = stg[src + 2], b3 = stg[src + 3];
  if (F32) {
    *reinterpret_cast<f32x4*>((float*)xo + dst) = f32x4{(float)a0, (float)a1, (float)a2, (float)a3};
    *reinterpret_cast<f32x4*>((float*)yo + dst) = f32x4{(float)b0, (float)b1, (float)b2, (float)b3};
  } else {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    *reinterpret_cast<s16x4*>((short*)xo + dst) = s16x4{a0, a1, a2, a3};
    *reinterpret_cast<s16x4*>((short*)yo + dst) = s16x4{b0, b1, b2, b3};
  }
}

extern "C" int ried_patch_crop(const short* slices_in, const short* slices_tg, int nslices, int H, int W,
                               int first_slice, int nsel, int patch_n, int P, const int* offsets,
                               unsigned long long seed, unsigned long long first_draw, void* x_out, void* y_out,
                               int out_f32, int* offsets_out, hipStream_t stream) {
  RIED_CHECK_ARG(slices_in && slices_tg && x_out && y_out, "ried_patch_crop: null pointer");
  RIED_CHECK_ARG(nslices > 0 && H > 0 && W > 0 && nsel > 0 && patch_n > 0, "ried_patch_crop: bad sizes");
  RIED_CHECK_ARG(first_slice >= 0 && (long long)first_slice + nsel <= nslices,
                 "ried_patch_crop: slices [%d, %d) outside the %d resident ones", first_slice, first_slice + nsel,
                 nslices);
  RIED_CHECK_ARG(P > 0 && P % 4 == 0, "ried_patch_crop: patch size must be a multiple of 4 (got %d)", P);
  // numpy.random.randint(0, h - p) raises for h - p <= 0 (src/loader.py:67-68): a patch must be smaller than the slice
  RIED_CHECK_ARG(P < H && P < W, "ried_patch_crop: patch size %d must be smaller than the slice (%d x %d)", P, H, W);
  RIED_CHECK_ARG(((uintptr_t)x_out % 16) == 0 && ((uintptr_t)y_out % 16) == 0, "ried_patch_crop: outputs must be 16-byte aligned");
  RIED_CHECK_ARG((long long)nsel * patch_n < (1ll << 31), "ried_patch_crop: too many patches");
  dim3 grid(nsel * patch_n, ceil_div(P * (P / 4), 256));
  if (out_f32)
    hipLaunchKernelGGL(patch_crop_kernel<true>, grid, dim3(256), 0, stream, slices_in, slices_tg, H, W, first_slice,
                       patch_n, P, offsets, seed, first_draw, x_out, y_out, offsets_out);
  else
    hipLaunchKernelGGL(patch_crop_kernel<false>, grid, dim3(256), 0, stream, slices_in, slices_tg, H, W, first_slice,
                       patch_n, P, offsets, seed, first_draw, x_out, y_out, offsets_out);
  RIED_CHECK_LAUNCH("ried_patch_crop");
  return RIED_OK;
}

// ---------------------------------------------------------------------------
// get_pixels_hu (src/prep.py:46-59) on a stack of decoded DICOM slices kept in HBM:
//   v = int16(raw);  v == -2000 -> 0;  slope != 1: v = int16(slope * float64(v));  v += int16(intercept)
// with numpy's conversions: 16-bit wrap-around for the casts and the add, truncation toward zero from float64.
// One 16-byte chunk (8 pixels) per thread; blockIdx.y = slice, whose slope / intercept are uniform.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hu_convert_kernel(const unsigned short* __restrict__ raw,
                                                         const double* __restrict__ slope,
                                                         const int* __restrict__ intercept, short* __restrict__ out,
                                                         long long npix) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const int s = blockIdx.y;
  const double sl = slope[s];
  const short ic = (short)intercept[s];
  const unsigned short* src = raw + (long long)s * npix;
  short* dst = out + (long long)s * npix;
  const long long nchunk = npix >> 3;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nchunk; i += (long long)gridDim.x * 256) {
    s16x8 v = *reinterpret_cast<const s16x8*>(src + i * 8);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      short t = v[q];
      if (t == -2000) t = 0;
      if (sl != 1.0) t = (short)(int)(sl * (double)t);
      v[q] = (short)(t + ic);
    }
    *reinterpret_cast<s16x8*>(dst + i * 8) = v;
  }
}

extern "C" int ried_hu_convert(const void* raw, const double* slope, const int* intercept, short* out, int nslices,
                               long long npix, hipStream_t stream) {
  RIED_CHECK_ARG(raw && slope && intercept && out && nslices > 0 && npix > 0, "ried_hu_convert: bad arguments");
  RIED_CHECK_ARG(npix % 8 == 0 && ((uintptr_t)raw % 16) == 0 && ((uintptr_t)out % 16) == 0,
                 "ried_hu_convert: slices must hold a multiple of 8 pixels and be 16-byte aligned");
  RIED_CHECK_ARG(nslices <= 65535, "ried_hu_convert: too many slices for one launch");
  dim3 grid(ew_grid(npix / 8, 256, 64), nslices);
  hipLaunchKernelGGL(hu_convert_kernel, grid, dim3(256), 0, stream, (const unsigned short*)raw, slope, intercept, out,
                     npix);
  RIED_CHECK_LAUNCH("ried_hu_convert");
  return RIED_OK;
}
