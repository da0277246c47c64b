// Shared device/host helpers for the RIED-GAN gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef unsigned short bf16_t;                                   // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;     // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;  // one 16-byte chunk = 8 bf16
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define RIED_OK 0
#define RIED_ERR_ARG 1
#define RIED_ERR_HIP 2

extern "C" const char* ried_last_error(void);
void ried_set_error(const char* fmt, ...);

#define RIED_CHECK_ARG(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      ried_set_error(__VA_ARGS__);           \
      return RIED_ERR_ARG;                   \
    }                                        \
  } while (0)

#define RIED_CHECK_LAUNCH(name)                                            \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      ried_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return RIED_ERR_HIP;                                                 \
    }                                                                      \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;                                          // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(bf16_t, b);
}
// both values in ONE v_cvt_pk_bf16_f32 (two scalar conversions, a shift and an or take four instructions)
typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(pk_f32x2{lo, hi}, pk_bf16x2));
}
__device__ __forceinline__ float bflo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bfhi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// Cross-lane sums without the LDS crossbar (ds_bpermute, which __shfl_xor compiles to): rotations inside the
// 16-lane rows are DPP modifiers of the add itself, and the two row-combining steps are gfx950's
// v_permlane16_swap / v_permlane32_swap.  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the four 16-lane rows of a wave, lane by lane (lane l gets v[l%16] + v[l%16+16] + v[l%16+32] + v[l%16+48])
// (Inline asm: the compiler's builtins for the two swaps return the first result twice in this ROCm release.
// The s_nop covers the VALU-write -> permlane-read wait states the compiler would otherwise insert itself.)
__device__ __forceinline__ float rows_sum(float v) {
  float x = v, y = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));   // x rows 1,3 <-> y rows 0,2
  x += y;
  y = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));   // x lanes 32..63 <-> y lanes 0..31
  return x + y;
}
// sum over the lanes of a 16-lane row that share lane % CLASSES (CLASSES = 1, 2, 4, 8 or 16), in every lane
template <int CLASSES>
__device__ __forceinline__ float row_class_sum(float v) {
  if (CLASSES <= 8) v += dpp_mov<0x128>(v);       // row_ror:8
  if (CLASSES <= 4) v += dpp_mov<0x124>(v);       // row_ror:4
  if (CLASSES <= 2) v += dpp_mov<0x122>(v);       // row_ror:2
  if (CLASSES <= 1) v += dpp_mov<0x121>(v);       // row_ror:1
  return v;
}
// sum over the lanes of a wave that share lane % CLASSES, in every lane
template <int CLASSES>
__device__ __forceinline__ float class_sum(float v) { return rows_sum(row_class_sum<CLASSES>(v)); }

// 64-lane wave reductions
__device__ __forceinline__ float wave_sum(float v) { return class_sum<1>(v); }
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread
__device__ __forceinline__ float block_sum_256(float v, float* sh4) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh4[w] = v;
  __syncthreads();
  return sh4[0] + sh4[1] + sh4[2] + sh4[3];
}

// n / d for 0 <= n < 2^30, 0 < d with n / d < 2^21, from a float reciprocal (inv = 1.0f / d) and one
// correction step: about a fifth of the instructions of the compiler's exact 32-bit division.
__device__ __forceinline__ int fast_div(int n, int d, float inv, int& rem) {
  int q = (int)((float)n * inv);
  int r = n - q * d;
  if (r < 0) { --q; r += d; }
  else if (r >= d) { ++q; r -= d; }
  rem = r;
  return q;
}

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int ew_grid(long long n, int per_block, int cap = 4096) {
  long long g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}
