// Bandwidth-bound kernels of the RIED-GAN hot path on NHWC bf16 activations:
// batch-norm (train mode) apply/backward, pooling, bilinear upsampling,
// activation backward, weight packing, softmax, bias/column sums, fused Adam.
// Every kernel moves 16 bytes per lane (8 bf16 channels) and reduces with
// 64-lane wavefront shuffles.
//
// A tensor "view" is (ptr, ld, C): pixel p's channels are ptr[p*ld .. p*ld+C).
// Views let producers write straight into channel slices of a concatenation
// buffer, which removes every torch.cat of src/rinet.py:88-90,245,291,340,387,418-438.
#include <stdlib.h>
#include "common.h"
#include "bn_common.h"

#define EW_THREADS 256

__device__ __forceinline__ void unpack8(const u32x4& v, float (&f)[8]) {
  f[0] = bflo(v[0]); f[1] = bfhi(v[0]); f[2] = bflo(v[1]); f[3] = bfhi(v[1]);
  f[4] = bflo(v[2]); f[5] = bfhi(v[2]); f[6] = bflo(v[3]); f[7] = bfhi(v[3]);
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  return u32x4{pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7])};
}
__device__ __forceinline__ u32x4 ld8(const bf16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st8(bf16_t* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }

static int check_view(const char* fn, const void* p, long long ld, int C) {
  RIED_CHECK_ARG(p != nullptr, "%s: null pointer", fn);
  RIED_CHECK_ARG(C > 0 && C % 8 == 0 && ld % 8 == 0 && ld >= C, "%s: C=%d ld=%lld must be multiples of 8, ld >= C", fn,
                 C, ld);
  RIED_CHECK_ARG(((uintptr_t)p % 16) == 0, "%s: pointer not 16-byte aligned", fn);
  return RIED_OK;
}
#define CHECK_VIEW(fn, p, ld, C)                         \
  do {                                                   \
    int rc_ = check_view(fn, p, ld, C);                  \
    if (rc_) return rc_;                                 \
  } while (0)

// ---------------------------------------------------------------------------
// weight packing: fp32 master weight (any 4-d stride order) -> bf16 [N][R][S][C], optionally divided by sigma
// ---------------------------------------------------------------------------
__global__ void pack_weight_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int N, int R, int S, int C,
                                   long long sn, long long sr, long long ss, long long sc,
                                   const float* __restrict__ sigma) {
  long long total = (long long)N * R * S * C;
  float inv = sigma ? 1.f / sigma[0] : 1.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long long t = i / C;
    int s = (int)(t % S); t /= S;
    int r = (int)(t % R);
    int n = (int)(t / R);
    dst[i] = f2bf(src[n * sn + r * sr + s * ss + c * sc] * inv);
  }
}

extern "C" int ried_pack_weight(const float* src, void* dst, int N, int R, int S, int C, long long sn, long long sr,
                                long long ss, long long sc, const float* sigma, hipStream_t stream) {
  RIED_CHECK_ARG(src && dst && N > 0 && R > 0 && S > 0 && C > 0, "ried_pack_weight: bad arguments");
  long long total = (long long)N * R * S * C;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(ew_grid(total, EW_THREADS)), dim3(EW_THREADS), 0, stream, src,
                     (bf16_t*)dst, N, R, S, C, sn, sr, ss, sc, sigma);
  RIED_CHECK_LAUNCH("ried_pack_weight");
  return RIED_OK;
}

// weight-gradient slabs [splitk][taps][Cp][Cg] -> master gradient with arbitrary strides
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, int splitk, int taps, int Cp, int Cg,
                                    float* __restrict__ out, long long st, long long sp, long long sg, float scale,
                                    int accumulate) {
  long long total = (long long)taps * Cp * Cg;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    float a = 0.f;
    for (int k = 0; k < splitk; ++k) a += slab[k * total + i];
    int g = (int)(i % Cg);
    long long t = i / Cg;
    int p = (int)(t % Cp);
    int tap = (int)(t / Cp);
    float* o = out + tap * st + p * sp + g * sg;
    a *= scale;
    *o = accumulate ? *o + a : a;
  }
}

extern "C" int ried_wgrad_reduce(const float* slab, int splitk, int taps, int Cp, int Cg, float* out, long long st,
                                 long long sp, long long sg, float scale, int accumulate, hipStream_t stream) {
  RIED_CHECK_ARG(slab && out && splitk > 0 && taps > 0 && Cp > 0 && Cg > 0, "ried_wgrad_reduce: bad arguments");
  long long total = (long long)taps * Cp * Cg;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ew_grid(total, EW_THREADS)), dim3(EW_THREADS), 0, stream, slab, splitk,
                     taps, Cp, Cg, out, st, sp, sg, scale, accumulate);
  RIED_CHECK_LAUNCH("ried_wgrad_reduce");
  return RIED_OK;
}

// ---------------------------------------------------------------------------
// Job-table forms: one launch packs every weight of a network / reduces every weight-gradient slab of a
// backward pass.  A job is a row of 64-bit words (pointers and integers alike); blockmap[b] = (job, chunk):
// block b handles elements [chunk * JOB_CHUNK, (chunk + 1) * JOB_CHUNK) of its job.
// ---------------------------------------------------------------------------
#define JOB_CHUNK 2048
#define PACK_JOB_WORDS 13
#define PACK_TILE_MAX_T 16
#define PACK_TILE_PITCH (32 * PACK_TILE_MAX_T + 2)
// pack job: src, dst, sigma, N, R, S, C, sn, sr, ss, sc, dst_pitch, dst_f32
__global__ __launch_bounds__(256) void pack_jobs_kernel(const long long* __restrict__ jobs,
                                                        const int* __restrict__ blockmap) {
  const long long* J = jobs + (long long)blockmap[2 * blockIdx.x] * PACK_JOB_WORDS;
  const float* src = (const float*)J[0];
  const float* sigma = (const float*)J[2];
  if (J[12] == 2) {
    // Tile job: a master stored [A][B][T] (a conv weight (Cout, Cin, kh*kw) or a transposed conv's (Cin, Cout, kh*kw))
    // -> dst1 [A][T][B] and dst2 [B][T][A] (either may be null), bf16, divided by sigma.  One block moves a tile of
    // 32 a x 32 b x all T through LDS: the master is read in contiguous runs of 32 T floats and both layouts are
    // written in contiguous 64-byte runs, where the strided form reads one float per cache line for the
    // transposed layout.
    __shared__ bf16_t tile[32][PACK_TILE_PITCH];
    const int A = (int)J[3], B = (int)J[4], T = (int)J[5];
    bf16_t* d1 = (bf16_t*)J[1];
    bf16_t* d2 = (bf16_t*)J[6];
    const int tilesB = (B + 31) >> 5, chunk = blockmap[2 * blockIdx.x + 1];
    const int a0 = (chunk / tilesB) << 5, b0 = (chunk % tilesB) << 5;
    const int na = min(32, A - a0), nb = min(32, B - b0), run = nb * T;
    const float inv = sigma ? 1.f / sigma[0] : 1.f;
    // (no integer divisions in the loops: a wave walks whole rows; the (a, t) of a row comes from a float reciprocal)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int a = wave; a < na; a += 4) {
      const float* row = src + ((long long)(a0 + a) * B + b0) * T;
      for (int j = lane; j < run; j += 64) tile[a][j] = f2bf(row[j] * inv);
    }
    __syncthreads();
    const float invT = 1.f / (float)T;
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;      // 16 lanes per output row of 32 elements
    if (d1 && sub < (nb >> 1)) {       // [a][t][b]: pairs of b
      for (int r = grp; r < na * T; r += 16) {
        int t;
        const int a = fast_div(r, T, invT, t);
        const unsigned v = (unsigned)tile[a][(2 * sub) * T + t] | ((unsigned)tile[a][(2 * sub + 1) * T + t] << 16);
        *reinterpret_cast<unsigned*>(d1 + ((long long)(a0 + a) * T + t) * B + b0 + 2 * sub) = v;
      }
    }
    if (d2 && sub < (na >> 1)) {       // [b][t][a]: pairs of a
      for (int r = grp; r < nb * T; r += 16) {
        int t;
        const int b = fast_div(r, T, invT, t);
        const unsigned v = (unsigned)tile[2 * sub][b * T + t] | ((unsigned)tile[2 * sub + 1][b * T + t] << 16);
        *reinterpret_cast<unsigned*>(d2 + ((long long)(b0 + b) * T + t) * A + a0 + 2 * sub) = v;
      }
    }
    return;
  }
  const int R = (int)J[4], S = (int)J[5], C = (int)J[6];
  const long long sn = J[7], sr = J[8], ss = J[9], sc = J[10], pitch = J[11];
  const long long inner = (long long)R * S * C, total = J[3] * inner;
  const float inv = sigma ? 1.f / sigma[0] : 1.f;
  long long i0 = (long long)blockmap[2 * blockIdx.x + 1] * JOB_CHUNK;
  if (total < (1ll << 30) && !J[12] && (C & 1) == 0 && (pitch & 1) == 0) {
    // the common case: 32-bit index arithmetic (divisions by float reciprocal + one correction step), two
    // channels per thread, one 4-byte store (4 passes of 256 pairs)
    const int tot = (int)total, inn = (int)inner, j0 = (int)i0;
    const float invC = 1.f / (float)C, invS = 1.f / (float)S, invR = 1.f / (float)R;
    bf16_t* dst = (bf16_t*)J[1];
#pragma unroll
    for (int u = 0; u < JOB_CHUNK / 512; ++u) {
      const int i = j0 + (u * 256 + threadIdx.x) * 2;
      if (i >= tot) break;
      int c, s_, r;
      const int t = fast_div(i, C, invC, c);
      const int t2 = fast_div(t, S, invS, s_);
      const int n = fast_div(t2, R, invR, r);
      const float* q = src + (long long)n * sn + r * sr + s_ * ss + c * sc;
      const float v0 = q[0] * inv, v1 = q[sc] * inv;
      *reinterpret_cast<unsigned*>(dst + (long long)n * pitch + (i - n * inn)) = pack2bf(v0, v1);
    }
    return;
  }
#pragma unroll
  for (int u = 0; u < JOB_CHUNK / 256; ++u) {
    long long i = i0 + u * 256 + threadIdx.x;
    if (i >= total) break;
    int c = (int)(i % C);
    long long t = i / C;
    int s_ = (int)(t % S); t /= S;
    int r = (int)(t % R);
    long long n = t / R;
    float v = src[n * sn + r * sr + s_ * ss + c * sc] * inv;
    long long d = n * pitch + (i - n * inner);
    if (J[12]) ((float*)J[1])[d] = v;
    else ((bf16_t*)J[1])[d] = f2bf(v);
  }
}

extern "C" int ried_pack_jobs(const long long* jobs, const int* blockmap, int nblocks, hipStream_t stream) {
  RIED_CHECK_ARG(jobs && blockmap && nblocks > 0, "ried_pack_jobs: bad arguments");
  hipLaunchKernelGGL(pack_jobs_kernel, dim3(nblocks), dim3(256), 0, stream, jobs, blockmap);
  RIED_CHECK_LAUNCH("ried_pack_jobs");
  return RIED_OK;
}

#define WRED_JOB_WORDS 12
// reduce job: slab, out, splitk, split_stride, taps, Cp, Cg, st, sp, sg, accumulate, scale (fp32 bits; 0 = 1.0)
//   out[tap*st + p*sp + g*sg] (+)= scale * sum_k slab[k*split_stride + (tap*Cp + p)*Cg + g]
// blockmap[b] = (job, p): one block per output row p.  The slab is read in its own order (g fastest) and the row
// is transposed through LDS so that masters whose row is laid out [g][tap] (st == 1, sg == taps: every conv,
// deconv and attention weight) are written as one contiguous run; other layouts fall back to scattered stores.
#define WRED_MAX_ROW 8192
__global__ __launch_bounds__(256) void wgrad_reduce_jobs_kernel(const long long* __restrict__ jobs,
                                                                const int* __restrict__ blockmap) {
  __shared__ float rowbuf[WRED_MAX_ROW];
  const long long* J = jobs + (long long)blockmap[2 * blockIdx.x] * WRED_JOB_WORDS;
  const int p = blockmap[2 * blockIdx.x + 1];
  const float* slab = (const float*)J[0];
  float* out = (float*)J[1];
  const int splitk = (int)J[2], taps = (int)J[4], Cp = (int)J[5], Cg = (int)J[6];
  const long long stride = J[3], st = J[7], sp = J[8], sg = J[9];
  const float scale = J[11] ? __int_as_float((int)J[11]) : 1.f;
  const int E = taps * Cg;
  const bool contiguous = st == 1 && sg == taps && E <= WRED_MAX_ROW;
  // 256 threads = Ew four-element groups x KT lanes over the splits (short rows would otherwise leave most threads
  // idle and each busy thread with a long chain of dependent-latency loads); 16-byte loads: Cg is a multiple of 8
  __shared__ f32x4 ksum[256];
  const int E4 = E >> 2;
  const int Ew = E4 < 256 ? E4 : 256, KT = 256 / Ew;
  const int el = threadIdx.x % Ew, kl = threadIdx.x / Ew;
  for (int e0 = 0; e0 < E4; e0 += Ew) {
    const int e = (e0 + el) << 2;
    const bool live = kl < KT && e < E;
    const int tap = live ? e / Cg : 0, g = live ? e - tap * Cg : 0;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (live) {
      const float* s = slab + ((long long)tap * Cp + p) * Cg + g;
#pragma unroll 4
      for (int k = kl; k < splitk; k += KT) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(s + k * stride);
        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
      }
    }
    if (KT > 1) {
      ksum[threadIdx.x] = acc;
      __syncthreads();
      if (live && kl == 0)
        for (int q = 1; q < KT; ++q) {
          const f32x4 v = ksum[q * Ew + el];
          acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
        }
      __syncthreads();
    }
    if (live && kl == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[q] *= scale;
        if (contiguous) rowbuf[(g + q) * taps + tap] = acc[q];
        else {
          float* o = out + tap * st + p * sp + (g + q) * sg;
          *o = J[10] ? *o + acc[q] : acc[q];
        }
      }
    }
  }
  if (contiguous) {
    __syncthreads();
    float* o = out + p * sp;
    for (int j = threadIdx.x; j < E; j += 256) o[j] = J[10] ? o[j] + rowbuf[j] : rowbuf[j];
  }
}

extern "C" int ried_wgrad_reduce_jobs(const long long* jobs, const int* blockmap, int nblocks, hipStream_t stream) {
  RIED_CHECK_ARG(jobs && blockmap && nblocks > 0, "ried_wgrad_reduce_jobs: bad arguments");
  hipLaunchKernelGGL(wgrad_reduce_jobs_kernel, dim3(nblocks), dim3(256), 0, stream, jobs, blockmap);
  RIED_CHECK_LAUNCH("ried_wgrad_reduce_jobs");
  return RIED_OK;
}

// ---------------------------------------------------------------------------
// BatchNorm2d, train mode (reference: src/rinet.py:187-193,199-205; torch semantics, eps 1e-5, momentum 0.1)
// ---------------------------------------------------------------------------
// Sum over rows of a [nrows][2][C] fp32 buffer for 32 channels per block: 1024 threads = 32 channels x 32 row
// lanes, four rows in flight per lane.  Returns (sum, sum2) for channel c in the threads with ty == 0.
#define ROWRED_THREADS 1024
__device__ __forceinline__ void reduce_rows_2c(const float* __restrict__ part, int nrows, int C, int c, int ty,
                                               float& s0, float& s1, float (*sh)[32][32], int ldp = 0) {
  float a = 0.f, b = 0.f;
  if (ldp <= 0) ldp = C;          // channels per row of the buffer (C: the layer owns the whole rows)
  if (c < C) {
#pragma unroll 4
    for (int r = ty; r < nrows; r += 32) {
      a += part[(long long)r * 2 * ldp + c];
      b += part[(long long)r * 2 * ldp + ldp + c];
    }
  }
  sh[0][ty][threadIdx.x & 31] = a;
  sh[1][ty][threadIdx.x & 31] = b;
  __syncthreads();
  s0 = 0.f; s1 = 0.f;
  if (ty == 0) {
#pragma unroll
    for (int k = 0; k < 32; ++k) { s0 += sh[0][k][threadIdx.x & 31]; s1 += sh[1][k][threadIdx.x & 31]; }
  }
}

// stats: [nrows][2][C] per-block partial sum / sum of squares from the conv epilogue.  save: [2][C] mean, invstd.
__global__ __launch_bounds__(ROWRED_THREADS) void bn_finalize_kernel(
    const float* __restrict__ stats, int nrows, int C, int ldstat, float count, float eps, float momentum,
    float* __restrict__ save, float* __restrict__ running_mean, float* __restrict__ running_var,
    long long* __restrict__ nbt) {
  __shared__ float sh[2][32][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), ty = threadIdx.x >> 5;
  float s0, s1;
  reduce_rows_2c(stats, nrows, C, c, ty, s0, s1, sh, ldstat);
  if (ty == 0 && c < C) {
    float mean = s0 / count;
    float var = fmaxf(s1 / count - mean * mean, 0.f);
    save[c] = mean;
    save[C + c] = rsqrtf(var + eps);
    if (running_mean) {
      float unb = count > 1.f ? var * (count / (count - 1.f)) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += 1;
}

// fold[r'][j] = sum over rows r = r' (mod FOLD_ROWS) of stats[r][j], j < 2C: with tens of thousands of partial
// rows (512x512 inference) one block per 32 channels would crawl through them alone.
#define FOLD_ROWS 256
__global__ __launch_bounds__(256) void rows_fold_kernel(const float* __restrict__ stats, int nrows, int C, int ldstat,
                                                        float* __restrict__ fold) {
  const int j = blockIdx.x * 256 + threadIdx.x, rp = blockIdx.y;
  if (j >= 2 * C) return;
  const int k = j >= C;
  const float* src = stats + k * ldstat + (j - k * C);       // column j of the layer's [2][C] slice of a row
  float a = 0.f;
#pragma unroll 4
  for (int r = rp; r < nrows; r += FOLD_ROWS) a += src[(long long)r * 2 * ldstat];
  fold[(long long)rp * 2 * C + j] = a;
}

// fold: NULL, or fp32 scratch [ried_bn_finalize_fold_floats(C)]; with it, more than 4 * FOLD_ROWS partial rows are
// first folded down to FOLD_ROWS by a wide kernel.
extern "C" long long ried_bn_finalize_fold_floats(int C) { return (long long)FOLD_ROWS * 2 * C; }

// ldstat: channels per statistics row when the layer is a channel slice of a wider convolution (0: C).
extern "C" int ried_bn_finalize(const float* stats, int nrows, int C, int ldstat, long long count, float eps,
                                float momentum, float* save, float* running_mean, float* running_var,
                                long long* num_batches_tracked, float* fold, hipStream_t stream) {
  RIED_CHECK_ARG(stats && save && C > 0 && count > 0 && nrows > 0, "ried_bn_finalize: bad arguments");
  if (ldstat <= 0) ldstat = C;
  RIED_CHECK_ARG(ldstat >= C, "ried_bn_finalize: ldstat smaller than C");
  if (fold && nrows > 4 * FOLD_ROWS) {
    hipLaunchKernelGGL(rows_fold_kernel, dim3(ceil_div(2 * C, 256), FOLD_ROWS), dim3(256), 0, stream, stats, nrows,
                       C, ldstat, fold);
    stats = fold;
    nrows = FOLD_ROWS;
    ldstat = C;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 32)), dim3(ROWRED_THREADS), 0, stream, stats, nrows, C,
                     ldstat, (float)count, eps, momentum, save, running_mean, running_var, num_batches_tracked);
  RIED_CHECK_LAUNCH("ried_bn_finalize");
  return RIED_OK;
}

// eval mode: save <- (running_mean, rsqrt(running_var + eps))
__global__ void bn_eval_stats_kernel(const float* rm, const float* rv, int C, float eps, float* save) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) { save[c] = rm[c]; save[C + c] = rsqrtf(rv[c] + eps); }
}
extern "C" int ried_bn_eval_stats(const float* rm, const float* rv, int C, float eps, float* save, hipStream_t stream) {
  RIED_CHECK_ARG(rm && rv && save && C > 0, "ried_bn_eval_stats: bad arguments");
  hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, stream, rm, rv, C, eps, save);
  RIED_CHECK_LAUNCH("ried_bn_eval_stats");
  return RIED_OK;
}

// Pixel-linear mapping of the per-channel kernels below: a thread keeps one 8-channel chunk c for its whole loop
// (its per-channel constants live in registers) and walks pixels pl, pl + step, ...; threads beyond the last
// whole pixel of a block pass stay idle.  No integer division inside the loops.
struct ChanWalk {
  int c;                // first channel of the chunk
  long long p, step;    // first pixel, pixel stride
  bool active;
  __device__ __forceinline__ ChanWalk(int C) {
    const int CC = C >> 3, PPB = blockDim.x / CC;           // PPB >= 1: C <= 8 * blockDim.x (host-checked)
    const int pl = threadIdx.x / CC;
    c = (threadIdx.x - pl * CC) << 3;
    active = pl < PPB;
    p = (long long)blockIdx.x * PPB + pl;
    step = (long long)gridDim.x * PPB;
  }
};
static inline int chan_grid(long long npix, int C, int iters, int cap) {
  int ppb = EW_THREADS / (C >> 3);
  return ew_grid(npix, ppb * iters, cap);
}

// Activation after BatchNorm: 0 none, 1 ReLU (src/rinet.py:188,200), 2 LeakyReLU(0.2) and 3 Sigmoid (the legacy
// discriminator, src/networks.py:533-542).  Codes 0-2 are max(v, slope * v) with slope 1, 0, 0.2 - no branches (as
// per-element tests of the run-time code these were three scalar branches per value in kernels whose whole run is a
// few dependent round trips); the sigmoid is one uniform branch per eight values.
__device__ __forceinline__ float act_slope(int act) { return act == 1 ? 0.f : (act == 2 ? 0.2f : 1.f); }
// v -> act(v); sl = act_slope(act)
__device__ __forceinline__ void bn_act8(float (&v)[8], int act, float sl) {
  if (act == 3) {
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = 1.f / (1.f + __expf(-v[q]));
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], sl * v[q]);
  }
}
// g -> g * act'(v), v the pre-activation values
__device__ __forceinline__ void bn_act_grad8(const float (&v)[8], float (&g)[8], int act, float sl) {
  if (act == 3) {
#pragma unroll
    for (int q = 0; q < 8; ++q) { float t = 1.f / (1.f + __expf(-v[q])); g[q] *= t * (1.f - t); }
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) g[q] *= v[q] > 0.f ? 1.f : sl;
  }
}

// y = act(gamma*(x-mean)*invstd + beta) [+ y]
// gridDim.y > 1: one statistics set per image (save [gridDim.y][2][C], npix pixels each)
__global__ __launch_bounds__(EW_THREADS) void bn_apply_kernel(
    const bf16_t* __restrict__ x, long long ldx, bf16_t* __restrict__ y, long long ldy, long long npix, int C,
    const float* __restrict__ save, const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
    int accumulate) {
  ChanWalk w(C);
  if (!w.active) return;
  x += (long long)blockIdx.y * npix * ldx;
  y += (long long)blockIdx.y * npix * ldy;
  save += (long long)blockIdx.y * 2 * C;
  const int c = w.c;
  const float sl = act_slope(relu);
  float a[8], b[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    a[q] = gamma[c + q] * save[C + c + q];
    b[q] = beta[c + q] - save[c + q] * a[q];
  }
#pragma unroll 2
  for (long long p = w.p; p < npix; p += w.step) {
    float f[8];
    unpack8(ld8(x + p * ldx + c), f);
#pragma unroll
    for (int q = 0; q < 8; ++q) f[q] = f[q] * a[q] + b[q];
    bn_act8(f, relu, sl);
    if (accumulate) {
      float o[8];
      unpack8(ld8(y + p * ldy + c), o);
#pragma unroll
      for (int q = 0; q < 8; ++q) f[q] += o[q];
    }
    st8(y + p * ldy + c, pack8(f));
  }
}

extern "C" int ried_bn_apply(const void* x, long long ldx, void* y, long long ldy, long long npix, int C,
                             const float* save, const float* gamma, const float* beta, int relu, int accumulate,
                             hipStream_t stream) {
  CHECK_VIEW("ried_bn_apply", x, ldx, C);
  CHECK_VIEW("ried_bn_apply", y, ldy, C);
  RIED_CHECK_ARG(save && gamma && beta && npix > 0 && C <= 8 * EW_THREADS, "ried_bn_apply: bad arguments");
  hipLaunchKernelGGL(bn_apply_kernel, dim3(chan_grid(npix, C, 2, 8192)), dim3(EW_THREADS), 0, stream,
                     (const bf16_t*)x, ldx, (bf16_t*)y, ldy, npix, C, save, gamma, beta, relu, accumulate);
  RIED_CHECK_LAUNCH("ried_bn_apply");
  return RIED_OK;
}

// Per-image train-mode statistics (the batch-1 evaluation loop of src/test.py:107-131 on a batch).
// stats: [nimg * rows_per_img][2][C] partial (sum, sum of squares) rows, the rows of an image consecutive
// (rows_per_img = 1: what ried_conv2d leaves with stats_post = 3)  ->  save [nimg][2][C] = (mean, 1/sqrt(var+eps)).
__global__ void bn_finalize_images_kernel(const float* __restrict__ stats, int rows_per_img, int nimg, int C,
                                          int ldstat, float count, float eps, float* __restrict__ save) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nimg * C) return;
  const int n = i / C, c = i - n * C;
  const float* p = stats + (long long)n * rows_per_img * 2 * ldstat + c;
  float s0 = 0.f, s1 = 0.f;
  for (int r = 0; r < rows_per_img; ++r) { s0 += p[(long long)r * 2 * ldstat]; s1 += p[(long long)r * 2 * ldstat + ldstat]; }
  const float mean = s0 / count;
  const float var = fmaxf(s1 / count - mean * mean, 0.f);
  save[(long long)n * 2 * C + c] = mean;
  save[(long long)n * 2 * C + C + c] = rsqrtf(var + eps);
}
// ldstat: channels per statistics row when the layer is a channel slice of a wider convolution (0: C).
extern "C" int ried_bn_finalize_images(const float* stats, int rows_per_img, int nimg, int C, int ldstat,
                                       long long count, float eps, float* save, hipStream_t stream) {
  RIED_CHECK_ARG(stats && save && rows_per_img > 0 && nimg > 0 && C > 0 && count > 0, "ried_bn_finalize_images: bad arguments");
  if (ldstat <= 0) ldstat = C;
  RIED_CHECK_ARG(ldstat >= C, "ried_bn_finalize_images: ldstat smaller than C");
  hipLaunchKernelGGL(bn_finalize_images_kernel, dim3(ceil_div((long long)nimg * C, 256)), dim3(256), 0, stream, stats,
                     rows_per_img, nimg, C, ldstat, (float)count, eps, save);
  RIED_CHECK_LAUNCH("ried_bn_finalize_images");
  return RIED_OK;
}
// ried_bn_apply with one statistics set per image: x, y are views of nimg images of npix_per_img pixels each
extern "C" int ried_bn_apply_images(const void* x, long long ldx, void* y, long long ldy, long long npix_per_img,
                                    int nimg, int C, const float* save, const float* gamma, const float* beta,
                                    int relu, int accumulate, hipStream_t stream) {
  CHECK_VIEW("ried_bn_apply_images", x, ldx, C);
  CHECK_VIEW("ried_bn_apply_images", y, ldy, C);
  RIED_CHECK_ARG(save && gamma && beta && npix_per_img > 0 && nimg > 0 && nimg <= 65535 && C <= 8 * EW_THREADS,
                 "ried_bn_apply_images: bad arguments");
  hipLaunchKernelGGL(bn_apply_kernel, dim3(chan_grid(npix_per_img, C, 2, 8192), nimg), dim3(EW_THREADS), 0, stream,
                     (const bf16_t*)x, ldx, (bf16_t*)y, ldy, npix_per_img, C, save, gamma, beta, relu, accumulate);
  RIED_CHECK_LAUNCH("ried_bn_apply_images");
  return RIED_OK;
}

// Rows of the folded form of the backward sums: the reduction's blocks add their partial sums into row
// (block % BN_BWD_FOLD_ROWS) of a zeroed [BN_BWD_FOLD_ROWS][2][C] buffer with float atomics and the apply pass adds
// the rows up itself, so no launch stands between the two (the sums then depend on the order the blocks finish in).
#define BN_BWD_FOLD_ROWS 8
// backward pass 1: per-block partial sums[0][c] = sum g, sums[1][c] = sum g*xhat, g = dy * [bn(x) > 0]
__device__ __forceinline__ void bn_bwd_reduce_body(
    const bf16_t* __restrict__ dy, long long lddy, const bf16_t* __restrict__ x, long long ldx, long long npix, int C,
    const float* __restrict__ save, const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
    float* __restrict__ sums, unsigned* counter, float* __restrict__ fin, float* dgamma, float* dbeta,
    int accumulate, float* __restrict__ folded) {
  __shared__ float red[EW_THREADS][17];           // per-thread partials, padded against bank conflicts
  __shared__ int last_flag;
  ChanWalk w(C);
  const int c = w.c;
  const float sl = act_slope(relu);
  float s1[8], s2[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
  if (w.active) {
    float a[8], b[8], mean[8], istd[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      mean[q] = save[c + q]; istd[q] = save[C + c + q];
      a[q] = gamma[c + q] * istd[q]; b[q] = beta[c + q];
    }
#pragma unroll 2
    for (long long p = w.p; p < npix; p += w.step) {
      float g[8], xv[8];
      unpack8(ld8(dy + p * lddy + c), g);
      unpack8(ld8(x + p * ldx + c), xv);
      float pre[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { xv[q] -= mean[q]; pre[q] = xv[q] * a[q] + b[q]; }
      bn_act_grad8(pre, g, relu, sl);
#pragma unroll
      for (int q = 0; q < 8; ++q) { s1[q] += g[q]; s2[q] += g[q] * (xv[q] * istd[q]); }
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) { red[threadIdx.x][q] = s1[q]; red[threadIdx.x][8 + q] = s2[q]; }
  __syncthreads();
  // channel j of statistic k: sum over the block's pixel lanes of the thread partials
  const int CC = C >> 3, PPB = EW_THREADS / CC;
  float* row = folded ? folded + (long long)(blockIdx.x % BN_BWD_FOLD_ROWS) * 2 * C
                      : sums + (long long)blockIdx.x * 2 * C;
  for (int j = threadIdx.x; j < 2 * C; j += EW_THREADS) {
    int k = j >= C, ch = j - k * C;
    int t0 = ch >> 3, q = (ch & 7) + 8 * k;
    float acc = 0.f;
    for (int pl = 0; pl < PPB; ++pl) acc += red[pl * CC + t0][q];
    if (folded) __hip_atomic_fetch_add(&row[j], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (counter) ried_publish(&row[j], acc);      // read by the last block of this launch: write-through
    else row[j] = acc;
  }
  // the last block to arrive reduces the rows (instead of a separate, dependent launch)
  if (counter && ried_last_block(counter, gridDim.x, &last_flag))
    bn_bwd_finish_rows(sums, gridDim.x, C, fin, dgamma, dbeta, accumulate, &red[0][0]);
}

__global__ __launch_bounds__(EW_THREADS) void bn_bwd_reduce_kernel(
    const bf16_t* __restrict__ dy, long long lddy, const bf16_t* __restrict__ x, long long ldx, long long npix, int C,
    const float* __restrict__ save, const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
    float* __restrict__ sums, unsigned* counter, float* __restrict__ fin, float* dgamma, float* dbeta,
    int accumulate, float* __restrict__ folded) {
  bn_bwd_reduce_body(dy, lddy, x, ldx, npix, C, save, gamma, beta, relu, sums, counter, fin, dgamma, dbeta, accumulate,
                     folded);
}

// backward pass 2: dx = gamma*invstd*(g - s1/n - xhat*s2/n)
__device__ __forceinline__ void bn_bwd_apply_body(
    const bf16_t* __restrict__ dy, long long lddy, const bf16_t* __restrict__ x, long long ldx,
    bf16_t* __restrict__ dx, long long lddx, long long npix, int C, const float* __restrict__ save,
    const float* __restrict__ gamma, const float* __restrict__ beta, int relu, const float* __restrict__ sums,
    int fold_rows, float* dgamma, float* dbeta, int accumulate) {
  ChanWalk w(C);
  if (!w.active) return;
  const int c = w.c;
  const float sl = act_slope(relu);
  const float invn = 1.f / (float)npix;
  // the first two pixels of the walk are fetched together with the sums, not after them (see bn_apply_train_body)
  u32x4 pg[2], px[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const long long p = w.p + k * w.step;
    if (p < npix) { pg[k] = ld8(dy + p * lddy + c); px[k] = ld8(x + p * ldx + c); }
  }
  float a[8], b[8], mean[8], istd[8], m1[8], m2[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    mean[q] = save[c + q]; istd[q] = save[C + c + q];
    a[q] = gamma[c + q] * istd[q]; b[q] = beta[c + q];
    m1[q] = sums[c + q]; m2[q] = sums[C + c + q];
  }
  // the folded form: every thread adds the rows up for its channels - four rows' loads in flight at a time (one row
  // per trip made the sums a chain of dependent round trips: most of this kernel's time on the small layers)
  for (int r0 = 1; r0 < fold_rows; r0 += 4) {
    f32x4 t[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (r0 + k < fold_rows) {
        const float* row = sums + (long long)(r0 + k) * 2 * C + c;
        t[k][0] = *reinterpret_cast<const f32x4*>(row);     t[k][1] = *reinterpret_cast<const f32x4*>(row + 4);
        t[k][2] = *reinterpret_cast<const f32x4*>(row + C); t[k][3] = *reinterpret_cast<const f32x4*>(row + C + 4);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (r0 + k < fold_rows) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          m1[q] += t[k][0][q]; m1[4 + q] += t[k][1][q]; m2[q] += t[k][2][q]; m2[4 + q] += t[k][3][q];
        }
      }
    }
  }
  if (fold_rows && dgamma && w.p == 0) {          // ... and the first pixel's threads write the parameter gradients
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      dbeta[c + q] = accumulate ? dbeta[c + q] + m1[q] : m1[q];
      dgamma[c + q] = accumulate ? dgamma[c + q] + m2[q] : m2[q];
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) { m1[q] *= invn; m2[q] *= invn; }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const long long p = w.p + k * w.step;
    if (p < npix) {
      float g[8], xv[8];
      unpack8(pg[k], g);
      unpack8(px[k], xv);
      float pre[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { xv[q] -= mean[q]; pre[q] = xv[q] * a[q] + b[q]; }
      bn_act_grad8(pre, g, relu, sl);
#pragma unroll
      for (int q = 0; q < 8; ++q) g[q] = a[q] * (g[q] - m1[q] - xv[q] * istd[q] * m2[q]);
      st8(dx + p * lddx + c, pack8(g));
    }
  }
#pragma unroll 2
  for (long long p = w.p + 2 * w.step; p < npix; p += w.step) {
    float g[8], xv[8];
    unpack8(ld8(dy + p * lddy + c), g);
    unpack8(ld8(x + p * ldx + c), xv);
    float pre[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { xv[q] -= mean[q]; pre[q] = xv[q] * a[q] + b[q]; }
    bn_act_grad8(pre, g, relu, sl);
#pragma unroll
    for (int q = 0; q < 8; ++q) g[q] = a[q] * (g[q] - m1[q] - xv[q] * istd[q] * m2[q]);
    st8(dx + p * lddx + c, pack8(g));
  }
}

__global__ __launch_bounds__(EW_THREADS) void bn_bwd_apply_kernel(
    const bf16_t* __restrict__ dy, long long lddy, const bf16_t* __restrict__ x, long long ldx,
    bf16_t* __restrict__ dx, long long lddx, long long npix, int C, const float* __restrict__ save,
    const float* __restrict__ gamma, const float* __restrict__ beta, int relu, const float* __restrict__ sums,
    int fold_rows, float* dgamma, float* dbeta, int accumulate) {
  bn_bwd_apply_body(dy, lddy, x, ldx, dx, lddx, npix, C, save, gamma, beta, relu, sums, fold_rows, dgamma, dbeta,
                    accumulate);
}

// partial rows [nrows][2][C] -> final sums [2][C] (read by the apply pass) and the parameter gradients
__device__ __forceinline__ void bn_bwd_finish_body(const float* __restrict__ part, int nrows, int C,
                                                   float* __restrict__ fin, float* dgamma, float* dbeta,
                                                   int accumulate) {
  __shared__ float sh[2][32][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), ty = threadIdx.x >> 5;
  float s0, s1;
  reduce_rows_2c(part, nrows, C, c, ty, s0, s1, sh);
  if (ty == 0 && c < C) {
    fin[c] = s0; fin[C + c] = s1;
    if (dgamma) {
      dbeta[c] = accumulate ? dbeta[c] + s0 : s0;
      dgamma[c] = accumulate ? dgamma[c] + s1 : s1;
    }
  }
}

__global__ __launch_bounds__(ROWRED_THREADS) void bn_bwd_finish_kernel(const float* __restrict__ part, int nrows,
                                                                       int C, float* __restrict__ fin, float* dgamma,
                                                                       float* dbeta, int accumulate) {
  bn_bwd_finish_body(part, nrows, C, fin, dgamma, dbeta, accumulate);
}

#define BN_BWD_MAX_ROWS 1024
// Job-table form: the BatchNorm backward of several layers that share the pixel count (the sibling 1x1 layers of an
// Inception block) in three launches instead of three per layer.  A job is 16 64-bit words:
// dy, lddy, x, ldx, dx, lddx, C, save, gamma, beta, act, scratch, dgamma, dbeta, accumulate, folded
// (as ried_bn_backward; folded = 0, or the job's zeroed rows of the folded form, and then scratch may be 0).
#define BN_JOB_WORDS 16
__global__ __launch_bounds__(EW_THREADS) void bn_bwd_reduce_jobs_kernel(const long long* __restrict__ jobs,
                                                                        long long npix) {
  const long long* J = jobs + (long long)blockIdx.y * BN_JOB_WORDS;
  bn_bwd_reduce_body((const bf16_t*)J[0], J[1], (const bf16_t*)J[2], J[3], npix, (int)J[6], (const float*)J[7],
                     (const float*)J[8], (const float*)J[9], (int)J[10], (float*)J[11], nullptr, nullptr, nullptr,
                     nullptr, 0, (float*)J[15]);
}
__global__ __launch_bounds__(ROWRED_THREADS) void bn_bwd_finish_jobs_kernel(const long long* __restrict__ jobs,
                                                                            int nrows) {
  const long long* J = jobs + (long long)blockIdx.y * BN_JOB_WORDS;
  const int C = (int)J[6];
  if ((int)blockIdx.x * 32 >= C || J[15]) return;
  float* scratch = (float*)J[11];
  float* dg = (J[12] && J[13]) ? (float*)J[12] : nullptr;
  bn_bwd_finish_body(scratch, nrows, C, scratch + (long long)BN_BWD_MAX_ROWS * 2 * C, dg, (float*)J[13], (int)J[14]);
}
__global__ __launch_bounds__(EW_THREADS) void bn_bwd_apply_jobs_kernel(const long long* __restrict__ jobs,
                                                                       long long npix) {
  const long long* J = jobs + (long long)blockIdx.y * BN_JOB_WORDS;
  const int C = (int)J[6];
  float* dg = (J[12] && J[13]) ? (float*)J[12] : nullptr;
  if (J[15])
    bn_bwd_apply_body((const bf16_t*)J[0], J[1], (const bf16_t*)J[2], J[3], (bf16_t*)J[4], J[5], npix, C,
                      (const float*)J[7], (const float*)J[8], (const float*)J[9], (int)J[10], (const float*)J[15],
                      BN_BWD_FOLD_ROWS, dg, (float*)J[13], (int)J[14]);
  else
    bn_bwd_apply_body((const bf16_t*)J[0], J[1], (const bf16_t*)J[2], J[3], (bf16_t*)J[4], J[5], npix, C,
                      (const float*)J[7], (const float*)J[8], (const float*)J[9], (int)J[10],
                      (const float*)J[11] + (long long)BN_BWD_MAX_ROWS * 2 * C, 0, nullptr, nullptr, 0);
}
// ---------------------------------------------------------------------------
// BatchNorm backward in ONE launch for layers small enough that a grid of at most BN_FUSED_MAX_BLOCKS blocks walks
// them in a few passes: the blocks add their partial sums into the folded rows with float atomics, meet at a grid
// barrier, add the rows up and apply.  With at most KEEP passes per thread the masked gradient and x-hat stay in
// registers across the barrier; otherwise the second walk reads the (L2-resident) operands again.
// ---------------------------------------------------------------------------
#define BN_FUSED_MAX_BLOCKS 256
#define BN_FUSED_MAX_PASSES 16
#define BN_FUSED_KEEP 4
template <int KEEP>
__device__ __forceinline__ void bn_bwd_fused_body(
    const bf16_t* __restrict__ dy, long long lddy, const bf16_t* __restrict__ x, long long ldx,
    bf16_t* __restrict__ dx, long long lddx, long long npix, int C, const float* __restrict__ save,
    const float* __restrict__ gamma, const float* __restrict__ beta, int relu, float* __restrict__ folded,
    unsigned* counter, float* dgamma, float* dbeta, int accumulate) {
  __shared__ float red[EW_THREADS][17];
  ChanWalk w(C);
  const int c = w.c;
  const float sl = act_slope(relu);
  float s1[8], s2[8], a[8], b[8], mean[8], istd[8];
  float kg[KEEP ? KEEP : 1][8], kx[KEEP ? KEEP : 1][8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
  if (w.active) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      mean[q] = save[c + q]; istd[q] = save[C + c + q];
      a[q] = gamma[c + q] * istd[q]; b[q] = beta[c + q];
    }
    if (KEEP) {
      u32x4 vg[KEEP ? KEEP : 1], vx[KEEP ? KEEP : 1];
#pragma unroll
      for (int k = 0; k < KEEP; ++k) {
        const long long p = w.p + k * w.step;
        if (p < npix) { vg[k] = ld8(dy + p * lddy + c); vx[k] = ld8(x + p * ldx + c); }
      }
#pragma unroll
      for (int k = 0; k < KEEP; ++k) {
        const long long p = w.p + k * w.step;
        float g[8], xv[8], pre[8];
        if (p < npix) { unpack8(vg[k], g); unpack8(vx[k], xv); }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          xv[q] = p < npix ? xv[q] - mean[q] : 0.f;
          g[q] = p < npix ? g[q] : 0.f;
          pre[q] = xv[q] * a[q] + b[q];
        }
        bn_act_grad8(pre, g, relu, sl);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          kg[k][q] = g[q]; kx[k][q] = xv[q] * istd[q];
          s1[q] += g[q]; s2[q] += g[q] * kx[k][q];
        }
      }
    } else {
#pragma unroll 2
      for (long long p = w.p; p < npix; p += w.step) {
        float g[8], xv[8];
        unpack8(ld8(dy + p * lddy + c), g);
        unpack8(ld8(x + p * ldx + c), xv);
        float pre[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { xv[q] -= mean[q]; pre[q] = xv[q] * a[q] + b[q]; }
        bn_act_grad8(pre, g, relu, sl);
#pragma unroll
        for (int q = 0; q < 8; ++q) { s1[q] += g[q]; s2[q] += g[q] * (xv[q] * istd[q]); }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) { red[threadIdx.x][q] = s1[q]; red[threadIdx.x][8 + q] = s2[q]; }
  __syncthreads();
  const int CC = C >> 3, PPB = EW_THREADS / CC;
  float* row = folded + (long long)(blockIdx.x % BN_BWD_FOLD_ROWS) * 2 * C;
  for (int j = threadIdx.x; j < 2 * C; j += EW_THREADS) {
    int k = j >= C, ch = j - k * C;
    int t0 = ch >> 3, q = (ch & 7) + 8 * k;
    float acc = 0.f;
    for (int pl = 0; pl < PPB; ++pl) acc += red[pl * CC + t0][q];
    __hip_atomic_fetch_add(&row[j], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  ried_grid_barrier(counter, gridDim.x);
  // every block adds the rows up for all channels (through LDS: [2][C] floats fit the reduction's array)
  float* tot = &red[0][0];
  for (int j = threadIdx.x; j < 2 * C; j += EW_THREADS) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < BN_BWD_FOLD_ROWS; ++r) t += ried_peek(folded + (long long)r * 2 * C + j);
    tot[j] = t;
    if (blockIdx.x == 0 && dgamma) {
      float* dst = j >= C ? dgamma + (j - C) : dbeta + j;
      *dst = accumulate ? *dst + t : t;
    }
  }
  __syncthreads();
  if (!w.active) return;
  const float invn = 1.f / (float)npix;
  float m1[8], m2[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { m1[q] = tot[c + q] * invn; m2[q] = tot[C + c + q] * invn; }
  if (KEEP) {
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
      const long long p = w.p + k * w.step;
      float o[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = a[q] * (kg[k][q] - m1[q] - kx[k][q] * m2[q]);
      if (p < npix) st8(dx + p * lddx + c, pack8(o));
    }
  } else {
#pragma unroll 2
    for (long long p = w.p; p < npix; p += w.step) {
      float g[8], xv[8];
      unpack8(ld8(dy + p * lddy + c), g);
      unpack8(ld8(x + p * ldx + c), xv);
      float pre[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { xv[q] -= mean[q]; pre[q] = xv[q] * a[q] + b[q]; }
      bn_act_grad8(pre, g, relu, sl);
#pragma unroll
      for (int q = 0; q < 8; ++q) g[q] = a[q] * (g[q] - m1[q] - xv[q] * istd[q] * m2[q]);
      st8(dx + p * lddx + c, pack8(g));
    }
  }
}
template <int KEEP>
__global__ __launch_bounds__(EW_THREADS) void bn_bwd_fused_kernel(
    const bf16_t* __restrict__ dy, long long lddy, const bf16_t* __restrict__ x, long long ldx,
    bf16_t* __restrict__ dx, long long lddx, long long npix, int C, const float* __restrict__ save,
    const float* __restrict__ gamma, const float* __restrict__ beta, int relu, float* __restrict__ folded,
    unsigned* counter, float* dgamma, float* dbeta, int accumulate) {
  bn_bwd_fused_body<KEEP>(dy, lddy, x, ldx, dx, lddx, npix, C, save, gamma, beta, relu, folded, counter, dgamma, dbeta,
                          accumulate);
}
// jobs as ried_bn_backward_jobs; word 11 is the job's barrier counter, word 15 its folded rows
template <int KEEP>
__global__ __launch_bounds__(EW_THREADS) void bn_bwd_fused_jobs_kernel(const long long* __restrict__ jobs,
                                                                       long long npix) {
  const long long* J = jobs + (long long)blockIdx.y * BN_JOB_WORDS;
  float* dg = (J[12] && J[13]) ? (float*)J[12] : nullptr;
  bn_bwd_fused_body<KEEP>((const bf16_t*)J[0], J[1], (const bf16_t*)J[2], J[3], (bf16_t*)J[4], J[5], npix, (int)J[6],
                          (const float*)J[7], (const float*)J[8], (const float*)J[9], (int)J[10], (float*)J[15],
                          (unsigned*)J[11], dg, (float*)J[13], (int)J[14]);
}
// blocks per job of the one-launch form, 0 when the layers are too large for it
static inline int bn_fused_grid(long long npix, int maxC, int njobs, int* passes) {
  if (njobs <= 0 || njobs > BN_FUSED_MAX_BLOCKS || maxC < 8 || maxC > 8 * EW_THREADS || npix <= 0) return 0;
  const int ppb = EW_THREADS / (maxC >> 3);               // pixels per block pass of the widest layer
  const int grid = ew_grid(npix, ppb, BN_FUSED_MAX_BLOCKS / njobs);
  const long long per_pass = (long long)grid * ppb;       // (narrower layers cover more pixels per pass)
  const long long np = (npix + per_pass - 1) / per_pass;
  if (np > BN_FUSED_MAX_PASSES) return 0;
  *passes = (int)np;
  return grid;
}
extern "C" int ried_bn_backward_fused_ok(long long npix, int maxC, int njobs) {
  int passes = 0;
  return bn_fused_grid(npix, maxC, njobs, &passes) > 0;
}
// Blocks that gave up waiting at a grid barrier since the library was loaded (0 unless a launch could not get all
// its blocks resident: its results are then wrong).  Synchronises with the device.
extern "C" int ried_barrier_timeouts(unsigned* out) {
  RIED_CHECK_ARG(out, "ried_barrier_timeouts: null argument");
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ried_barrier_timeouts_dev), sizeof(unsigned)) != hipSuccess) {
    ried_set_error("ried_barrier_timeouts: copy failed");
    return RIED_ERR_HIP;
  }
  return RIED_OK;
}

// jobs: device table of njobs records; maxC: the largest channel count among them (sizes the grids);
// all_folded: every job carries folded rows, so the launch that finishes the per-block rows is left out
extern "C" int ried_bn_backward_jobs(const long long* jobs, int njobs, long long npix, int maxC, int all_folded,
                                     hipStream_t stream) {
  RIED_CHECK_ARG(jobs && njobs > 0 && njobs <= 65535 && npix > 0 && maxC >= 8 && maxC % 8 == 0 && maxC <= 8 * EW_THREADS,
                 "ried_bn_backward_jobs: bad arguments");
  if (all_folded == 2) {        // one launch: word 11 of every job is its zeroed barrier counter
    int passes = 0;
    const int fg = bn_fused_grid(npix, maxC, njobs, &passes);
    RIED_CHECK_ARG(fg > 0, "ried_bn_backward_jobs: layers too large for the one-launch form");
    if (passes <= BN_FUSED_KEEP)
      hipLaunchKernelGGL(bn_bwd_fused_jobs_kernel<BN_FUSED_KEEP>, dim3(fg, njobs), dim3(EW_THREADS), 0, stream, jobs,
                         npix);
    else
      hipLaunchKernelGGL(bn_bwd_fused_jobs_kernel<0>, dim3(fg, njobs), dim3(EW_THREADS), 0, stream, jobs, npix);
    RIED_CHECK_LAUNCH("ried_bn_backward_jobs");
    return RIED_OK;
  }
  const int grid = chan_grid(npix, maxC, 4, BN_BWD_MAX_ROWS);
  hipLaunchKernelGGL(bn_bwd_reduce_jobs_kernel, dim3(grid, njobs), dim3(EW_THREADS), 0, stream, jobs, npix);
  if (!all_folded)
    hipLaunchKernelGGL(bn_bwd_finish_jobs_kernel, dim3(ceil_div(maxC, 32), njobs), dim3(ROWRED_THREADS), 0, stream,
                       jobs, grid);
  hipLaunchKernelGGL(bn_bwd_apply_jobs_kernel, dim3(chan_grid(npix, maxC, 2, 8192), njobs), dim3(EW_THREADS), 0, stream,
                     jobs, npix);
  RIED_CHECK_LAUNCH("ried_bn_backward_jobs");
  return RIED_OK;
}

// scratch: fp32 [(BN_BWD_MAX_ROWS + 1) * 2 * C] (need not be zeroed).  dgamma/dbeta may be null.
// counter: null, or one zero-initialised unsigned (left at zero): the reduction's last block then finishes the
// sums itself instead of a separate launch.
// folded: null, or ZEROED fp32 [ried_bn_backward_folded_floats(C)] (left non-zero): the folded form, two launches,
// scratch and counter unused (scratch may be null).
extern "C" int ried_bn_backward(const void* dy, long long lddy, const void* x, long long ldx, void* dx, long long lddx,
                                long long npix, int C, const float* save, const float* gamma, const float* beta,
                                int relu, float* scratch, float* dgamma, float* dbeta, int accumulate,
                                unsigned* counter, float* folded, hipStream_t stream) {
  CHECK_VIEW("ried_bn_backward", dy, lddy, C);
  CHECK_VIEW("ried_bn_backward", x, ldx, C);
  CHECK_VIEW("ried_bn_backward", dx, lddx, C);
  RIED_CHECK_ARG(save && gamma && beta && (scratch || folded) && npix > 0, "ried_bn_backward: bad arguments");
  RIED_CHECK_ARG(C <= 8 * EW_THREADS, "ried_bn_backward: C too large");
  int grid = chan_grid(npix, C, 4, BN_BWD_MAX_ROWS);
  float* dg = (dgamma && dbeta) ? dgamma : nullptr;
  if (folded && counter) {
    int passes = 0;
    const int fg = bn_fused_grid(npix, C, 1, &passes);
    RIED_CHECK_ARG(fg > 0, "ried_bn_backward: layer too large for the one-launch form (ried_bn_backward_fused_ok)");
    if (passes <= BN_FUSED_KEEP)
      hipLaunchKernelGGL(bn_bwd_fused_kernel<BN_FUSED_KEEP>, dim3(fg), dim3(EW_THREADS), 0, stream, (const bf16_t*)dy,
                         lddy, (const bf16_t*)x, ldx, (bf16_t*)dx, lddx, npix, C, save, gamma, beta, relu, folded,
                         counter, dg, dbeta, accumulate);
    else
      hipLaunchKernelGGL(bn_bwd_fused_kernel<0>, dim3(fg), dim3(EW_THREADS), 0, stream, (const bf16_t*)dy, lddy,
                         (const bf16_t*)x, ldx, (bf16_t*)dx, lddx, npix, C, save, gamma, beta, relu, folded, counter,
                         dg, dbeta, accumulate);
    RIED_CHECK_LAUNCH("ried_bn_backward");
    return RIED_OK;
  }
  if (folded) {
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(grid), dim3(EW_THREADS), 0, stream, (const bf16_t*)dy, lddy,
                       (const bf16_t*)x, ldx, npix, C, save, gamma, beta, relu, (float*)nullptr, (unsigned*)nullptr,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, 0, folded);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(chan_grid(npix, C, 2, 8192)), dim3(EW_THREADS), 0, stream,
                       (const bf16_t*)dy, lddy, (const bf16_t*)x, ldx, (bf16_t*)dx, lddx, npix, C, save, gamma, beta,
                       relu, (const float*)folded, BN_BWD_FOLD_ROWS, dg, dbeta, accumulate);
    RIED_CHECK_LAUNCH("ried_bn_backward");
    return RIED_OK;
  }
  float* fin = scratch + (long long)BN_BWD_MAX_ROWS * 2 * C;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(grid), dim3(EW_THREADS), 0, stream, (const bf16_t*)dy, lddy,
                     (const bf16_t*)x, ldx, npix, C, save, gamma, beta, relu, scratch, counter, fin, dg, dbeta,
                     accumulate, (float*)nullptr);
  if (!counter)
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(ceil_div(C, 32)), dim3(ROWRED_THREADS), 0, stream, scratch, grid,
                       C, fin, dg, dbeta, accumulate);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(chan_grid(npix, C, 2, 8192)), dim3(EW_THREADS), 0, stream,
                     (const bf16_t*)dy, lddy, (const bf16_t*)x, ldx, (bf16_t*)dx, lddx, npix, C, save, gamma, beta,
                     relu, (const float*)fin, 0, (float*)nullptr, (float*)nullptr, 0);
  RIED_CHECK_LAUNCH("ried_bn_backward");
  return RIED_OK;
}
// The folded form of ried_bn_backward as its two launches, for callers that must touch the sums between them
// (synchronised BatchNorm: the per-channel sums are averaged over the data-parallel ranks before they are applied).
// reduce: folded (ZEROED fp32 [ried_bn_backward_folded_floats(C)]) += the partial sums of g and g * x-hat.
extern "C" int ried_bn_backward_reduce(const void* dy, long long lddy, const void* x, long long ldx, long long npix,
                                       int C, const float* save, const float* gamma, const float* beta, int relu,
                                       float* folded, hipStream_t stream) {
  CHECK_VIEW("ried_bn_backward_reduce", dy, lddy, C);
  CHECK_VIEW("ried_bn_backward_reduce", x, ldx, C);
  RIED_CHECK_ARG(save && gamma && beta && folded && npix > 0 && C <= 8 * EW_THREADS,
                 "ried_bn_backward_reduce: bad arguments");
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(chan_grid(npix, C, 4, BN_BWD_MAX_ROWS)), dim3(EW_THREADS), 0, stream,
                     (const bf16_t*)dy, lddy, (const bf16_t*)x, ldx, npix, C, save, gamma, beta, relu, (float*)nullptr,
                     (unsigned*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0, folded);
  RIED_CHECK_LAUNCH("ried_bn_backward_reduce");
  return RIED_OK;
}
// apply: dx from the rows of `folded` (their sum = the sums over the npix pixels, or the mean over ranks of the
// ranks' sums: the formula divides by npix either way); dgamma / dbeta (may be null) receive the rows' totals.
extern "C" int ried_bn_backward_apply(const void* dy, long long lddy, const void* x, long long ldx, void* dx,
                                      long long lddx, long long npix, int C, const float* save, const float* gamma,
                                      const float* beta, int relu, const float* folded, float* dgamma, float* dbeta,
                                      int accumulate, hipStream_t stream) {
  CHECK_VIEW("ried_bn_backward_apply", dy, lddy, C);
  CHECK_VIEW("ried_bn_backward_apply", x, ldx, C);
  CHECK_VIEW("ried_bn_backward_apply", dx, lddx, C);
  RIED_CHECK_ARG(save && gamma && beta && folded && npix > 0 && C <= 8 * EW_THREADS,
                 "ried_bn_backward_apply: bad arguments");
  float* dg = (dgamma && dbeta) ? dgamma : nullptr;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(chan_grid(npix, C, 2, 8192)), dim3(EW_THREADS), 0, stream,
                     (const bf16_t*)dy, lddy, (const bf16_t*)x, ldx, (bf16_t*)dx, lddx, npix, C, save, gamma, beta,
                     relu, folded, BN_BWD_FOLD_ROWS, dg, dbeta, accumulate);
  RIED_CHECK_LAUNCH("ried_bn_backward_apply");
  return RIED_OK;
}
extern "C" long long ried_bn_backward_scratch_floats(int C) { return (long long)(BN_BWD_MAX_ROWS + 1) * 2 * C; }
extern "C" long long ried_bn_backward_folded_floats(int C) { return (long long)BN_BWD_FOLD_ROWS * 2 * C; }

// ---------------------------------------------------------------------------
// BatchNorm forward for small layers in one launch after the conv instead of two.  Blocks own a slice of up to 64
// channels (blockIdx.y) and a group of pixels (blockIdx.x), so that the few folded rows of the statistics can be
// re-reduced by every block for its own slice (a few KB from L2) instead of by a separate dependent launch.
// ---------------------------------------------------------------------------
#define BN_SLICE 64
#define BN_FOLD_ROWS 32            // rows of the folded forward statistics (atomics of the conv epilogue)
struct SliceWalk {
  int c0, CCs, SW;      // first channel of the slice, 8-channel chunks per slice, channels per slice
  int cl, c;            // this thread's chunk within the slice, its first channel
  long long p, step;
  bool active;
  __device__ __forceinline__ SliceWalk(int C) {
    CCs = min(BN_SLICE / 8, C >> 3);
    SW = CCs * 8;
    c0 = blockIdx.y * BN_SLICE;
    const int PPB = blockDim.x / CCs, pl = threadIdx.x / CCs;
    cl = threadIdx.x - pl * CCs;
    c = c0 + cl * 8;
    active = pl < PPB && c < C;
    p = (long long)blockIdx.x * PPB + pl;
    step = (long long)gridDim.x * PPB;
  }
};

// sums[k][ch] (k < 2, ch < SW) = sum over rows of part[r][k][c0 + ch]; part: [nrows][2][C].  256 threads.
// ldp: channels per row of `part` (>= C: the layer's channels may be a slice of a wider statistics buffer)
__device__ __forceinline__ void slice_row_sums(const float* __restrict__ part, int nrows, int C, int ldp, int c0,
                                               int SW, float* lds /*256*/, float* sums /*[2][BN_SLICE]*/) {
  const int ncol = 2 * SW, RL = 256 / ncol;
  const int rl = threadIdx.x / ncol, col = threadIdx.x - rl * ncol;
  const int k = col >= SW, ch = col - k * SW;
  if (rl < RL) {
    float a = 0.f;
    if (c0 + ch < C) {       // sixteen rows' loads in flight at a time (the folded form has 32 rows in all)
      const float* src = part + (long long)k * ldp + c0 + ch;
      for (int r = rl; r < nrows; r += 16 * RL) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = r + i * RL < nrows ? src[(long long)(r + i * RL) * 2 * ldp] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) a += v[i];
      }
    }
    lds[rl * ncol + col] = a;
  }
  __syncthreads();
  if (threadIdx.x < ncol) {
    float t = 0.f;
    for (int j = 0; j < RL; ++j) t += lds[j * ncol + threadIdx.x];
    sums[k * BN_SLICE + ch] = t;
  }
  __syncthreads();
}

// y = relu(gamma*(x-mean)*invstd + beta) [+ y], train mode, with the statistics taken from the folded partial rows
// of the producing conv; the first pixel group of each slice also writes save and the running statistics.
__device__ __forceinline__ void bn_apply_train_body(
    const bf16_t* __restrict__ x, long long ldx, bf16_t* __restrict__ y, long long ldy, long long npix, int C,
    const float* __restrict__ stats, int nrows, int ldstat, float eps, float momentum, float* __restrict__ save,
    float* __restrict__ running_mean, float* __restrict__ running_var, long long* __restrict__ nbt,
    const float* __restrict__ gamma, const float* __restrict__ beta, int relu, int accumulate) {
  __shared__ float lds[256];
  __shared__ float sums[2][BN_SLICE];
  __shared__ float ab[2][BN_SLICE];
  SliceWalk w(C);
  // The first two pixels of the thread's walk are fetched before the statistics are (a dependent round trip to
  // memory otherwise: most of this kernel's time on the small layers, which are a single pass).
  u32x4 px[2], po[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const long long p = w.p + k * w.step;
    if (w.active && p < npix) {
      px[k] = ld8(x + p * ldx + w.c);
      if (accumulate) po[k] = ld8(y + p * ldy + w.c);
    }
  }
  slice_row_sums(stats, nrows, C, ldstat, w.c0, w.SW, lds, &sums[0][0]);
  if (threadIdx.x < w.SW && w.c0 + threadIdx.x < C) {
    const int c = w.c0 + threadIdx.x;
    const float count = (float)npix;
    const float mean = sums[0][threadIdx.x] / count;
    const float var = fmaxf(sums[1][threadIdx.x] / count - mean * mean, 0.f);
    const float istd = rsqrtf(var + eps);
    const float a = gamma[c] * istd;
    ab[0][threadIdx.x] = a;
    ab[1][threadIdx.x] = beta[c] - mean * a;
    if (blockIdx.x == 0) {
      save[c] = mean;
      save[C + c] = istd;
      if (running_mean) {
        float unb = count > 1.f ? var * (count / (count - 1.f)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
      }
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && nbt) nbt[0] += 1;
  __syncthreads();
  if (!w.active) return;
  const int c = w.c;
  const float sl = act_slope(relu);
  float a[8], b[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { a[q] = ab[0][w.cl * 8 + q]; b[q] = ab[1][w.cl * 8 + q]; }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const long long p = w.p + k * w.step;
    if (p < npix) {
      float f[8];
      unpack8(px[k], f);
#pragma unroll
      for (int q = 0; q < 8; ++q) f[q] = f[q] * a[q] + b[q];
      bn_act8(f, relu, sl);
      if (accumulate) {
        float o[8];
        unpack8(po[k], o);
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] += o[q];
      }
      st8(y + p * ldy + c, pack8(f));
    }
  }
#pragma unroll 2
  for (long long p = w.p + 2 * w.step; p < npix; p += w.step) {
    float f[8];
    unpack8(ld8(x + p * ldx + c), f);
#pragma unroll
    for (int q = 0; q < 8; ++q) f[q] = f[q] * a[q] + b[q];
    bn_act8(f, relu, sl);
    if (accumulate) {
      float o[8];
      unpack8(ld8(y + p * ldy + c), o);
#pragma unroll
      for (int q = 0; q < 8; ++q) f[q] += o[q];
    }
    st8(y + p * ldy + c, pack8(f));
  }
}

__global__ __launch_bounds__(EW_THREADS) void bn_apply_train_kernel(
    const bf16_t* __restrict__ x, long long ldx, bf16_t* __restrict__ y, long long ldy, long long npix, int C,
    const float* __restrict__ stats, int nrows, int ldstat, float eps, float momentum, float* __restrict__ save,
    float* __restrict__ running_mean, float* __restrict__ running_var, long long* __restrict__ nbt,
    const float* __restrict__ gamma, const float* __restrict__ beta, int relu, int accumulate) {
  bn_apply_train_body(x, ldx, y, ldy, npix, C, stats, nrows, ldstat, eps, momentum, save, running_mean, running_var,
                      nbt, gamma, beta, relu, accumulate);
}
// Job-table form (blockIdx.z = job): 16 64-bit words per job:
// x, ldx, y, ldy, C, stats, nrows, ldstat, save, running_mean, running_var, nbt, gamma, beta, act, accumulate
__global__ __launch_bounds__(EW_THREADS) void bn_apply_train_jobs_kernel(const long long* __restrict__ jobs,
                                                                         long long npix, float eps, float momentum) {
  const long long* J = jobs + (long long)blockIdx.z * 16;
  const int C = (int)J[4];
  if ((int)blockIdx.y * BN_SLICE >= C) return;
  bn_apply_train_body((const bf16_t*)J[0], J[1], (bf16_t*)J[2], J[3], npix, C, (const float*)J[5], (int)J[6], (int)J[7],
                      eps, momentum, (float*)J[8], (float*)J[9], (float*)J[10], (long long*)J[11], (const float*)J[12],
                      (const float*)J[13], (int)J[14], (int)J[15]);
}

// Pixel groups per block pass x `iters` passes per block.  Every block of these kernels first adds up the folded
// statistics rows of its channel slice (a fixed cost: 32 rows x 2 x 64 floats and two barriers), so on large layers
// the blocks take more passes: about `njobs`-th of SLICE_TARGET_BLOCKS blocks in all.
static int slice_target_blocks() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("RIED_BN_SLICE_BLOCKS");
    v = e ? atoi(e) : 2048;
  }
  return v;
}
static inline void slice_grid(long long npix, int C, int iters, int cap, dim3& grid, int njobs = 1) {
  int ccs = C / 8 < BN_SLICE / 8 ? C / 8 : BN_SLICE / 8;
  int ppb = EW_THREADS / ccs;
  const int gy = ceil_div(C, BN_SLICE);
  const int target = slice_target_blocks();
  if (target > 0) {
    const long long gx = target / (gy * njobs) > 0 ? target / (gy * njobs) : 1;
    const long long want = (npix + ppb * gx - 1) / (ppb * gx);
    if (want > iters) iters = (int)(want < 64 ? want : 64);
  }
  grid = dim3(ew_grid(npix, ppb * iters, cap), gy);
}

// stats: [nrows][2][C] folded partial sums (ried_conv2d with stats_post = 2 writes min(BN_FOLD_ROWS, tiles) rows).
extern "C" int ried_bn_fold_rows() { return BN_FOLD_ROWS; }
extern "C" int ried_bn_apply_train(const void* x, long long ldx, void* y, long long ldy, long long npix, int C,
                                   const float* stats, int nrows, int ldstat, float eps, float momentum,
                                   float* save, float* running_mean, float* running_var,
                                   long long* num_batches_tracked, const float* gamma, const float* beta, int relu,
                                   int accumulate, hipStream_t stream) {
  CHECK_VIEW("ried_bn_apply_train", x, ldx, C);
  CHECK_VIEW("ried_bn_apply_train", y, ldy, C);
  RIED_CHECK_ARG(stats && save && gamma && beta && npix > 0 && nrows > 0, "ried_bn_apply_train: bad arguments");
  if (ldstat <= 0) ldstat = C;
  RIED_CHECK_ARG(ldstat >= C, "ried_bn_apply_train: ldstat smaller than C");
  dim3 grid;
  slice_grid(npix, C, 2, 8192, grid);
  hipLaunchKernelGGL(bn_apply_train_kernel, grid, dim3(EW_THREADS), 0, stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy,
                     npix, C, stats, nrows, ldstat, eps, momentum, save, running_mean, running_var, num_batches_tracked,
                     gamma, beta, relu, accumulate);
  RIED_CHECK_LAUNCH("ried_bn_apply_train");
  return RIED_OK;
}

// jobs: device table of njobs records (layers sharing npix); maxC / minC: the largest / smallest C among them
extern "C" int ried_bn_apply_train_jobs(const long long* jobs, int njobs, long long npix, int maxC, int minC, float eps,
                                        float momentum, hipStream_t stream) {
  RIED_CHECK_ARG(jobs && njobs > 0 && njobs <= 65535 && npix > 0 && minC >= 8 && maxC >= minC && maxC % 8 == 0,
                 "ried_bn_apply_train_jobs: bad arguments");
  dim3 grid, g2;
  slice_grid(npix, maxC, 2, 8192, grid, njobs);
  slice_grid(npix, minC, 2, 8192, g2, njobs);          // fewer channels per pixel: fewer pixels per block pass
  grid.x = grid.x > g2.x ? grid.x : g2.x;
  grid.z = njobs;
  hipLaunchKernelGGL(bn_apply_train_jobs_kernel, grid, dim3(EW_THREADS), 0, stream, jobs, npix, eps, momentum);
  RIED_CHECK_LAUNCH("ried_bn_apply_train_jobs");
  return RIED_OK;
}

// ---------------------------------------------------------------------------
// pooling / upsampling (reference: src/rinet.py:235,260,266,330,355,361,428; VGG16 MaxPool2d(2,2))
// ---------------------------------------------------------------------------
// mode 0: max 3x3 s2 p1   mode 1: max 2x2 s2 p0   mode 2: avg 3x3 s1 p1 (count_include_pad)
// One thread per output chunk (8 channels).  The window is walked with compile-time bounds and clamped addresses, so
// that a thread's (up to nine) loads are independent and in flight together; index arithmetic is 32-bit.
template <int MODE>
__global__ __launch_bounds__(EW_THREADS) void pool_fwd_kernel(const bf16_t* __restrict__ x, long long ldx,
                                                              bf16_t* __restrict__ y, long long ldy, int total, int H,
                                                              int W, int Ho, int Wo, int C) {
  constexpr int K = MODE == 1 ? 2 : 3, S = MODE == 2 ? 1 : 2, PAD = MODE == 1 ? 0 : 1;
  const int CC = C >> 3;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int p = i / CC, c = (i - p * CC) << 3;
    const int ox = p % Wo, t = p / Wo, oy = t % Ho, n = t / Ho;
    const bf16_t* xi = x + (long long)n * H * W * ldx + c;
    u32x4 v[K * K];
    bool ok[K * K];
#pragma unroll
    for (int r = 0; r < K; ++r) {
      const int hi = oy * S - PAD + r;
#pragma unroll
      for (int z = 0; z < K; ++z) {
        const int wi = ox * S - PAD + z;
        ok[r * K + z] = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
        const int hc = min(max(hi, 0), H - 1), wc = min(max(wi, 0), W - 1);
        v[r * K + z] = ld8(xi + (long long)(hc * W + wc) * ldx);
      }
    }
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = MODE == 2 ? 0.f : -INFINITY;
#pragma unroll
    for (int e = 0; e < K * K; ++e) {
      float f[8];
      unpack8(v[e], f);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (MODE == 2) acc[q] += ok[e] ? f[q] : 0.f;
        else acc[q] = ok[e] ? fmaxf(acc[q], f[q]) : acc[q];
      }
    }
    if (MODE == 2) {
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] *= (1.f / 9.f);
    }
    st8(y + (long long)p * ldy + c, pack8(acc));
  }
}

// max-pool backward: the gradient of an output goes to the first maximal element of its window in
// row-major scan order.  One thread owns one INPUT pixel chunk, so no atomics.  The (at most four) windows that
// contain the pixel and their (at most nine) elements are walked with compile-time bounds and clamped addresses,
// so that all the loads of a thread are independent and in flight together.
template <int K, int PAD>
__global__ __launch_bounds__(EW_THREADS) void maxpool_bwd_kernel(const bf16_t* __restrict__ x, long long ldx,
                                                                 const bf16_t* __restrict__ dy, long long lddy,
                                                                 bf16_t* __restrict__ dx, long long lddx, int total,
                                                                 int H, int W, int Ho, int Wo, int C, int accumulate,
                                                                 int relu_mask) {
  constexpr int S = 2, NWIN = (K + S - 1) / S;             // windows per axis that can contain a pixel
  const int CC = C >> 3;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int p = i / CC, c = (i - p * CC) << 3;
    const int wx = p % W, t = p / W, hy = t % H, n = t / H;
    float g[8], me[8];
    if (accumulate) unpack8(ld8(dx + (long long)p * lddx + c), g);
    else {
#pragma unroll
      for (int q = 0; q < 8; ++q) g[q] = 0.f;
    }
    unpack8(ld8(x + (long long)p * ldx + c), me);
    const int oy0 = (hy + PAD) / S, ox0 = (wx + PAD) / S;  // the last window (per axis) that contains the pixel
#pragma unroll
    for (int a = 0; a < NWIN; ++a) {
#pragma unroll
      for (int b = 0; b < NWIN; ++b) {
        const int oy = oy0 - a, ox = ox0 - b;
        // window (oy, ox) exists and contains (hy, wx)?
        const bool wok = oy >= 0 && ox >= 0 && oy < Ho && ox < Wo && oy * S - PAD + K > hy && ox * S - PAD + K > wx;
        // first maximum of the window: larger than everything before it (scan order), not smaller than anything after
        float mb[8], ma[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { mb[q] = -INFINITY; ma[q] = -INFINITY; }
#pragma unroll
        for (int r = 0; r < K; ++r) {
#pragma unroll
          for (int z = 0; z < K; ++z) {
            const int hi = oy * S - PAD + r, wi = ox * S - PAD + z;
            const bool ok = wok && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W && !(hi == hy && wi == wx);
            const int hc = min(max(hi, 0), H - 1), wc = min(max(wi, 0), W - 1);
            float f[8];
            unpack8(ld8(x + ((long long)(n * H + hc) * W + wc) * ldx + c), f);
            const bool before = hi < hy || (hi == hy && wi < wx);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const float v = ok ? f[q] : -INFINITY;
              if (before) mb[q] = fmaxf(mb[q], v);
              else ma[q] = fmaxf(ma[q], v);
            }
          }
        }
        bool win[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) win[q] = wok && me[q] > mb[q] && me[q] >= ma[q];
        float d[8];
        const int oyc = min(max(oy, 0), Ho - 1), oxc = min(max(ox, 0), Wo - 1);
        unpack8(ld8(dy + ((long long)(n * Ho + oyc) * Wo + oxc) * lddy + c), d);
#pragma unroll
        for (int q = 0; q < 8; ++q) g[q] += win[q] ? d[q] : 0.f;
      }
    }
    if (relu_mask) {
#pragma unroll
      for (int q = 0; q < 8; ++q) g[q] = me[q] > 0.f ? g[q] : 0.f;
    }
    st8(dx + (long long)p * lddx + c, pack8(g));
  }
}

// avg 3x3 s1 p1 backward == the same stencil applied to dy, accumulated into dx
__global__ void avgpool_bwd_kernel(const bf16_t* __restrict__ dy, long long lddy, bf16_t* __restrict__ dx,
                                   long long lddx, int nimg, int H, int W, int C) {
  const int CC = C >> 3;
  long long total = (long long)nimg * H * W * CC;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long p = i / CC;
    int c = (int)(i - p * CC) << 3;
    int wx = (int)(p % W);
    long long t = p / W;
    int hy = (int)(t % H);
    int n = (int)(t / H);
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    for (int r = -1; r <= 1; ++r) {
      int hi = hy + r;
      if ((unsigned)hi >= (unsigned)H) continue;
      for (int z = -1; z <= 1; ++z) {
        int wi = wx + z;
        if ((unsigned)wi >= (unsigned)W) continue;
        float f[8];
        unpack8(ld8(dy + ((long long)(n * H + hi) * W + wi) * lddy + c), f);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += f[q];
      }
    }
    float g[8];
    unpack8(ld8(dx + p * lddx + c), g);
#pragma unroll
    for (int q = 0; q < 8; ++q) g[q] += acc[q] * (1.f / 9.f);
    st8(dx + p * lddx + c, pack8(g));
  }
}

// avg_pool2d(3, 1, 1) in vertical strips: a thread owns 4 vertically adjacent outputs of one column (8 channels),
// loads the 6 x 3 inputs around them once (all loads in flight together), sums each input row horizontally and
// combines three row sums per output: 4.5 loads per output instead of 9 (the kernel is bound by L1 / L2 reads).
__global__ __launch_bounds__(EW_THREADS) void avgpool3_strip_kernel(const bf16_t* __restrict__ x, long long ldx,
                                                                    bf16_t* __restrict__ y, long long ldy, int total,
                                                                    int H, int W, int C) {
  constexpr int R = 4;
  const int CC = C >> 3, HS = H / R;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int p = i / CC, c = (i - p * CC) << 3;
    const int wx = p % W, t = p / W, hs = t % HS, n = t / HS;
    const int h0 = hs * R;
    const bf16_t* xi = x + (long long)n * H * W * ldx + c;
    u32x4 v[R + 2][3];
#pragma unroll
    for (int r = 0; r < R + 2; ++r) {
      const int hi = min(max(h0 - 1 + r, 0), H - 1);
#pragma unroll
      for (int z = 0; z < 3; ++z) v[r][z] = ld8(xi + (long long)(hi * W + min(max(wx - 1 + z, 0), W - 1)) * ldx);
    }
    float rs[R + 2][8];
#pragma unroll
    for (int r = 0; r < R + 2; ++r) {
      const bool rok = (unsigned)(h0 - 1 + r) < (unsigned)H;
#pragma unroll
      for (int q = 0; q < 8; ++q) rs[r][q] = 0.f;
#pragma unroll
      for (int z = 0; z < 3; ++z) {
        const bool ok = rok && (unsigned)(wx - 1 + z) < (unsigned)W;
        float f[8];
        unpack8(v[r][z], f);
#pragma unroll
        for (int q = 0; q < 8; ++q) rs[r][q] += ok ? f[q] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      float o[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = (rs[j][q] + rs[j + 1][q] + rs[j + 2][q]) * (1.f / 9.f);
      st8(y + ((long long)(n * H + h0 + j) * W + wx) * ldy + c, pack8(o));
    }
  }
}

extern "C" int ried_pool_fwd(const void* x, long long ldx, void* y, long long ldy, int nimg, int H, int W, int C,
                             int mode, hipStream_t stream) {
  CHECK_VIEW("ried_pool_fwd", x, ldx, C);
  CHECK_VIEW("ried_pool_fwd", y, ldy, C);
  RIED_CHECK_ARG(mode >= 0 && mode <= 2 && nimg > 0 && H > 0 && W > 0, "ried_pool_fwd: bad arguments");
  int Ho = mode == 0 ? (H + 2 - 3) / 2 + 1 : mode == 1 ? H / 2 : H;
  int Wo = mode == 0 ? (W + 2 - 3) / 2 + 1 : mode == 1 ? W / 2 : W;
  const long long total = (long long)nimg * Ho * Wo * (C / 8);
  RIED_CHECK_ARG(total < (1ll << 31) && (long long)H * W < (1ll << 31), "ried_pool_fwd: tensor too large");
  const dim3 grid(ew_grid(total, EW_THREADS, 65536)), block(EW_THREADS);
#define RIED_POOL_FWD(M)                                                                                      \
  hipLaunchKernelGGL(pool_fwd_kernel<M>, grid, block, 0, stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, (int)total, \
                     H, W, Ho, Wo, C)
  if (mode == 0) RIED_POOL_FWD(0);
  else if (mode == 1) RIED_POOL_FWD(1);
  else if (H % 4 == 0 && total >= (1 << 20)) {       // large maps: vertical strips (small ones are launch-bound)
    const int strips = (int)(total / 4);
    hipLaunchKernelGGL(avgpool3_strip_kernel, dim3(ew_grid(strips, EW_THREADS, 65536)), block, 0, stream,
                       (const bf16_t*)x, ldx, (bf16_t*)y, ldy, strips, H, W, C);
  } else RIED_POOL_FWD(2);
#undef RIED_POOL_FWD
  RIED_CHECK_LAUNCH("ried_pool_fwd");
  return RIED_OK;
}

// avg_pool2d(3, 1, 1) (count_include_pad) of a NARROW tensor together with the train-mode BatchNorm statistics of the
// result.  avg_pool2d followed by a bias-free 1x1 convolution (the branch_pool of Inception A / C / E,
// src/rinet.py:234,250-251; 320,344-345; 416,441-442) equals the convolution followed by the pool - both are linear
// and zero padding commutes with a convolution that has no bias - and the convolution's output has 16-96 channels
// where its input has 64-512: pooling AFTER the convolution moves 4-16x less data and lets the 1x1 layer ride in the
// block's fused sibling convolution.  This kernel is that pool: y = avgpool3(x), and per block one row
// stats[block][2][C] = column sums of y and y*y (fp32, before rounding to bf16), the partial rows ried_bn_finalize
// reduces.  A thread owns one 8-channel chunk column and walks pixels; 256 / (C / 8) pixels per pass of the block.
__global__ __launch_bounds__(EW_THREADS) void avgpool3_stats_kernel(const bf16_t* __restrict__ x, long long ldx,
                                                                   bf16_t* __restrict__ y, long long ldy, int npix, int H,
                                                                   int W, int C, float* __restrict__ stats) {
  __shared__ float red[EW_THREADS][17];
  const int CC = C >> 3, ppp = EW_THREADS / CC;            // pixels per pass
  const int cc = threadIdx.x % CC, pp = threadIdx.x / CC, c = cc << 3;
  float s1[8], s2[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
  if (pp < ppp) {
    for (int p = blockIdx.x * ppp + pp; p < npix; p += gridDim.x * ppp) {
      const int wx = p % W, t = p / W, hy = t % H, n = t / H;
      const bf16_t* xi = x + (long long)n * H * W * ldx + c;
      float acc[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = 0.f;
#pragma unroll
      for (int r = -1; r <= 1; ++r) {
#pragma unroll
        for (int z = -1; z <= 1; ++z) {
          const int hi = hy + r, wi = wx + z;
          const bool ok = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
          const int hc = min(max(hi, 0), H - 1), wc = min(max(wi, 0), W - 1);
          float f[8];
          unpack8(ld8(xi + (long long)(hc * W + wc) * ldx), f);
#pragma unroll
          for (int q = 0; q < 8; ++q) acc[q] += ok ? f[q] : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) { acc[q] *= (1.f / 9.f); s1[q] += acc[q]; s2[q] += acc[q] * acc[q]; }
      st8(y + (long long)p * ldy + c, pack8(acc));
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) { red[threadIdx.x][q] = s1[q]; red[threadIdx.x][8 + q] = s2[q]; }
  __syncthreads();
  // thread t < 2 * C: statistic t / C (sum, sum of squares) of channel t % C over the block's pixel slots
  for (int t = threadIdx.x; t < 2 * C; t += EW_THREADS) {
    const int which = t / C, ch = t - which * C, k = ch >> 3, q = ch & 7;
    float v = 0.f;
    for (int j = 0; j < ppp; ++j) v += red[j * CC + k][which * 8 + q];
    stats[(long long)blockIdx.x * 2 * C + t] = v;
  }
}

// rows of the statistics buffer ([rows][2][C]) that ried_avgpool3_stats writes
extern "C" int ried_avgpool3_stats_rows(long long npix, int C) {
  const int ppp = EW_THREADS / (C / 8);
  long long rows = (npix + (long long)ppp * 2 - 1) / ((long long)ppp * 2);          // two passes per block, at most 1024 rows
  return (int)(rows < 1 ? 1 : rows > 1024 ? 1024 : rows);
}

extern "C" int ried_avgpool3_stats(const void* x, long long ldx, void* y, long long ldy, int nimg, int H, int W, int C,
                                   float* stats, hipStream_t stream) {
  CHECK_VIEW("ried_avgpool3_stats", x, ldx, C);
  CHECK_VIEW("ried_avgpool3_stats", y, ldy, C);
  RIED_CHECK_ARG(stats && nimg > 0 && H > 0 && W > 0 && C >= 8 && C <= 8 * EW_THREADS,
                 "ried_avgpool3_stats: bad arguments");
  const long long npix = (long long)nimg * H * W;
  RIED_CHECK_ARG(npix < (1ll << 31), "ried_avgpool3_stats: tensor too large");
  const int rows = ried_avgpool3_stats_rows(npix, C);
  hipLaunchKernelGGL(avgpool3_stats_kernel, dim3(rows), dim3(EW_THREADS), 0, stream, (const bf16_t*)x, ldx, (bf16_t*)y,
                     ldy, (int)npix, H, W, C, stats);
  RIED_CHECK_LAUNCH("ried_avgpool3_stats");
  return RIED_OK;
}

// MaxPool2d(2, 2) backward (even H, W: the windows tile the input).  One thread owns one WINDOW chunk: it reads the
// four inputs and the output gradient once and writes the four input gradients - every tensor exactly once.
__global__ __launch_bounds__(EW_THREADS) void maxpool2_bwd_kernel(const bf16_t* __restrict__ x, long long ldx,
                                                                  const bf16_t* __restrict__ dy, long long lddy,
                                                                  bf16_t* __restrict__ dx, long long lddx, int total,
                                                                  int Ho, int Wo, int C, int accumulate,
                                                                  int relu_mask) {
  const int CC = C >> 3, W = 2 * Wo;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int po = i / CC, c = (i - po * CC) << 3;
    const int ox = po % Wo, t = po / Wo;                   // t = n * Ho + oy
    const long long p00 = ((long long)t * 2 * W) + 2 * ox;  // input pixel (2*oy, 2*ox) of image n
    const long long pp[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
    float v[4][8], d[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) unpack8(ld8(x + pp[e] * ldx + c), v[e]);
    unpack8(ld8(dy + (long long)po * lddy + c), d);
    float g[4][8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      int best = 0;                                        // the first maximum in row-major scan order
      float m = v[0][q];
#pragma unroll
      for (int e = 1; e < 4; ++e)
        if (v[e][q] > m) { m = v[e][q]; best = e; }
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e][q] = (e == best && !(relu_mask && v[e][q] <= 0.f)) ? d[q] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (accumulate) {
        float o[8];
        unpack8(ld8(dx + pp[e] * lddx + c), o);
#pragma unroll
        for (int q = 0; q < 8; ++q) g[e][q] = relu_mask && v[e][q] <= 0.f ? 0.f : g[e][q] + o[q];
      }
      st8(dx + pp[e] * lddx + c, pack8(g[e]));
    }
  }
}

// dx += pool backward of dy.  x is the forward input (max modes only).
extern "C" int ried_pool_bwd(const void* x, long long ldx, const void* dy, long long lddy, void* dx, long long lddx,
                             int nimg, int H, int W, int C, int mode, int accumulate, int relu_mask,
                             hipStream_t stream) {
  CHECK_VIEW("ried_pool_bwd", dy, lddy, C);
  RIED_CHECK_ARG(mode != 2 || (accumulate && !relu_mask), "ried_pool_bwd: the average mode only accumulates");
  CHECK_VIEW("ried_pool_bwd", dx, lddx, C);
  RIED_CHECK_ARG(mode >= 0 && mode <= 2 && nimg > 0 && H > 0 && W > 0, "ried_pool_bwd: bad arguments");
  int grid = ew_grid((long long)nimg * H * W * (C / 8), EW_THREADS);
  if (mode == 2) {
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid), dim3(EW_THREADS), 0, stream, (const bf16_t*)dy, lddy,
                       (bf16_t*)dx, lddx, nimg, H, W, C);
  } else {
    CHECK_VIEW("ried_pool_bwd", x, ldx, C);
    int Ho = mode == 0 ? (H + 2 - 3) / 2 + 1 : H / 2;
    int Wo = mode == 0 ? (W + 2 - 3) / 2 + 1 : W / 2;
    const long long nwin = (long long)nimg * Ho * Wo * (C / 8);
    if (mode == 1 && H % 2 == 0 && W % 2 == 0 && nwin < (1ll << 31)) {
      hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_grid(nwin, EW_THREADS)), dim3(EW_THREADS), 0, stream,
                         (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, (int)nwin, Ho, Wo, C,
                         accumulate, relu_mask);
      RIED_CHECK_LAUNCH("ried_pool_bwd");
      return RIED_OK;
    }
    RIED_CHECK_ARG((long long)nimg * H * W * (C / 8) < (1ll << 31), "ried_pool_bwd: tensor too large");
    const int total = nimg * H * W * (C / 8);
    if (mode == 0)
      hipLaunchKernelGGL((maxpool_bwd_kernel<3, 1>), dim3(grid), dim3(EW_THREADS), 0, stream, (const bf16_t*)x, ldx,
                         (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, total, H, W, Ho, Wo, C, accumulate, relu_mask);
    else
      hipLaunchKernelGGL((maxpool_bwd_kernel<2, 0>), dim3(grid), dim3(EW_THREADS), 0, stream, (const bf16_t*)x, ldx,
                         (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, total, H, W, Ho, Wo, C, accumulate, relu_mask);
  }
  RIED_CHECK_LAUNCH("ried_pool_bwd");
  return RIED_OK;
}

// bilinear x2, align_corners=True: src = dst * (H-1)/(2H-1)
__global__ void upsample2x_fwd_kernel(const bf16_t* __restrict__ x, long long ldx, bf16_t* __restrict__ y,
                                      long long ldy, int nimg, int H, int W, int C) {
  const int CC = C >> 3, Ho = 2 * H, Wo = 2 * W;
  const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  long long total = (long long)nimg * Ho * Wo * CC;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long p = i / CC;
    int c = (int)(i - p * CC) << 3;
    int ox = (int)(p % Wo);
    long long t = p / Wo;
    int oy = (int)(t % Ho);
    int n = (int)(t / Ho);
    float fy = sy * oy, fx = sx * ox;
    int y0 = (int)fy, x0 = (int)fx;
    int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    float ly = fy - y0, lx = fx - x0;
    const bf16_t* b = x + (long long)n * H * W * ldx + c;
    float a[8], bq[8], cq[8], d[8], o[8];
    unpack8(ld8(b + ((long long)y0 * W + x0) * ldx), a);
    unpack8(ld8(b + ((long long)y0 * W + x1) * ldx), bq);
    unpack8(ld8(b + ((long long)y1 * W + x0) * ldx), cq);
    unpack8(ld8(b + ((long long)y1 * W + x1) * ldx), d);
#pragma unroll
    for (int q = 0; q < 8; ++q)
      o[q] = (1.f - ly) * ((1.f - lx) * a[q] + lx * bq[q]) + ly * ((1.f - lx) * cq[q] + lx * d[q]);
    st8(y + p * ldy + c, pack8(o));
  }
}

// backward as a gather over input pixels: dx[h][w] = sum of the (at most 4x4) outputs that read it
__global__ void upsample2x_bwd_kernel(const bf16_t* __restrict__ dy, long long lddy, bf16_t* __restrict__ dx,
                                      long long lddx, int nimg, int H, int W, int C) {
  const int CC = C >> 3, Ho = 2 * H, Wo = 2 * W;
  const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  long long total = (long long)nimg * H * W * CC;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long p = i / CC;
    int c = (int)(i - p * CC) << 3;
    int wx = (int)(p % W);
    long long t = p / W;
    int hy = (int)(t % H);
    int n = (int)(t / H);
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    // outputs oy whose source interval [y0, y0+1] touches hy: scan a safe range
    int oy_lo = max(0, 2 * hy - 3), oy_hi = min(Ho - 1, 2 * hy + 3);
    int ox_lo = max(0, 2 * wx - 3), ox_hi = min(Wo - 1, 2 * wx + 3);
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      float fy = sy * oy;
      int y0 = (int)fy, y1 = min(y0 + 1, H - 1);
      float ly = fy - y0;
      float wy = (y0 == hy ? 1.f - ly : 0.f) + (y1 == hy ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        float fx = sx * ox;
        int x0 = (int)fx, x1 = min(x0 + 1, W - 1);
        float lx = fx - x0;
        float wxx = (x0 == wx ? 1.f - lx : 0.f) + (x1 == wx ? lx : 0.f);
        if (wxx == 0.f) continue;
        float f[8];
        unpack8(ld8(dy + ((long long)(n * Ho + oy) * Wo + ox) * lddy + c), f);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += wy * wxx * f[q];
      }
    }
    float g[8];
    unpack8(ld8(dx + p * lddx + c), g);
#pragma unroll
    for (int q = 0; q < 8; ++q) g[q] += acc[q];
    st8(dx + p * lddx + c, pack8(g));
  }
}

extern "C" int ried_upsample2x_fwd(const void* x, long long ldx, void* y, long long ldy, int nimg, int H, int W, int C,
                                   hipStream_t stream) {
  CHECK_VIEW("ried_upsample2x_fwd", x, ldx, C);
  CHECK_VIEW("ried_upsample2x_fwd", y, ldy, C);
  RIED_CHECK_ARG(nimg > 0 && H > 0 && W > 0, "ried_upsample2x_fwd: bad arguments");
  hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(ew_grid((long long)nimg * 4 * H * W * (C / 8), EW_THREADS)),
                     dim3(EW_THREADS), 0, stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, nimg, H, W, C);
  RIED_CHECK_LAUNCH("ried_upsample2x_fwd");
  return RIED_OK;
}

extern "C" int ried_upsample2x_bwd(const void* dy, long long lddy, void* dx, long long lddx, int nimg, int H, int W,
                                   int C, hipStream_t stream) {
  CHECK_VIEW("ried_upsample2x_bwd", dy, lddy, C);
  CHECK_VIEW("ried_upsample2x_bwd", dx, lddx, C);
  RIED_CHECK_ARG(nimg > 0 && H > 0 && W > 0, "ried_upsample2x_bwd: bad arguments");
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(ew_grid((long long)nimg * H * W * (C / 8), EW_THREADS)),
                     dim3(EW_THREADS), 0, stream, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, nimg, H, W, C);
  RIED_CHECK_LAUNCH("ried_upsample2x_bwd");
  return RIED_OK;
}

// ---------------------------------------------------------------------------
// generic elementwise helpers
// ---------------------------------------------------------------------------
// op 0: out = a                         (copy a view)
// op 1: out = out + a                   (accumulate)
// op 2: out = a * (y > 0 ? 1 : slope)   (activation backward from the activation OUTPUT y)
// op 3: out = out + a * (y > 0 ? 1 : slope)
// op 4: out = s[0]*a + y                (self-attention combine, s = gamma)
// op 5: out = s[0]*a                    (scale by a device scalar)
__global__ void ew_kernel(int op, const bf16_t* __restrict__ a, long long lda, const bf16_t* __restrict__ y,
                          long long ldy, bf16_t* __restrict__ out, long long ldo, long long npix, int C, float slope,
                          const float* __restrict__ s) {
  const int CC = C >> 3;
  long long total = npix * CC;
  const float sv = s ? s[0] : 1.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long p = i / CC;
    int c = (int)(i - p * CC) << 3;
    float fa[8], fy[8], fo[8];
    unpack8(ld8(a + p * lda + c), fa);
    if (op >= 2 && op <= 4) unpack8(ld8(y + p * ldy + c), fy);
    if (op == 1 || op == 3) unpack8(ld8(out + p * ldo + c), fo);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = fa[q];
      if (op == 2 || op == 3) v *= (fy[q] > 0.f ? 1.f : slope);
      if (op == 1 || op == 3) v += fo[q];
      if (op == 4) v = sv * v + fy[q];
      if (op == 5) v = sv * v;
      fa[q] = v;
    }
    st8(out + p * ldo + c, pack8(fa));
  }
}

extern "C" int ried_ew(int op, const void* a, long long lda, const void* y, long long ldy, void* out, long long ldo,
                       long long npix, int C, float slope, const float* s, hipStream_t stream) {
  RIED_CHECK_ARG(op >= 0 && op <= 5, "ried_ew: bad op %d", op);
  CHECK_VIEW("ried_ew", a, lda, C);
  CHECK_VIEW("ried_ew", out, ldo, C);
  if (op >= 2 && op <= 4) CHECK_VIEW("ried_ew", y, ldy, C);
  RIED_CHECK_ARG(npix > 0, "ried_ew: npix");
  hipLaunchKernelGGL(ew_kernel, dim3(ew_grid(npix * (C / 8), EW_THREADS)), dim3(EW_THREADS), 0, stream, op,
                     (const bf16_t*)a, lda, (const bf16_t*)y, ldy, (bf16_t*)out, ldo, npix, C, slope, s);
  RIED_CHECK_LAUNCH("ried_ew");
  return RIED_OK;
}

// column sums (bias gradients): per-block partial rows, then a row reduction; no global atomics.
// A thread owns one 8-channel chunk column and walks pixel rows (no index arithmetic in the loop, four loads in
// flight); the block's pixel lanes are added through LDS.  C <= 8 * EW_THREADS.
__global__ __launch_bounds__(EW_THREADS) void colsum_kernel(const bf16_t* __restrict__ a, long long lda, long long npix,
                                                            int C, float* __restrict__ part) {
  __shared__ float red[EW_THREADS][9];
  ChanWalk w(C);
  float s[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) s[q] = 0.f;
  if (w.active) {
    const bf16_t* src = a + w.c;
    long long p = w.p;
    for (; p + 3 * w.step < npix; p += 4 * w.step) {
      u32x4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = ld8(src + (p + k * w.step) * lda);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float f[8];
        unpack8(v[k], f);
#pragma unroll
        for (int q = 0; q < 8; ++q) s[q] += f[q];
      }
    }
    for (; p < npix; p += w.step) {
      float f[8];
      unpack8(ld8(src + p * lda), f);
#pragma unroll
      for (int q = 0; q < 8; ++q) s[q] += f[q];
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) red[threadIdx.x][q] = s[q];
  __syncthreads();
  const int CC = C >> 3, PPB = EW_THREADS / CC;
  for (int j = threadIdx.x; j < C; j += EW_THREADS) {
    float t = 0.f;
    for (int pl = 0; pl < PPB; ++pl) t += red[pl * CC + (j >> 3)][j & 7];
    part[(long long)blockIdx.x * C + j] = t;
  }
}
// the form for C > 8 * EW_THREADS: a thread's chunk column changes as it strides through the tensor
__global__ void colsum_wide_kernel(const bf16_t* __restrict__ a, long long lda, long long npix, int C,
                                   float* __restrict__ part) {
  extern __shared__ float lsum[];
  for (int i = threadIdx.x; i < C; i += blockDim.x) lsum[i] = 0.f;
  __syncthreads();
  const int CC = C >> 3;
  // (threads beyond the largest multiple of CC stay idle so no element is skipped)
  long long stride = (long long)gridDim.x * blockDim.x / CC * CC;
  long long total = npix * CC;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < total && i < stride) {
    int c = (int)(i % CC) << 3;
    float s[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) s[q] = 0.f;
    for (; i < total; i += stride) {
      float f[8];
      unpack8(ld8(a + (i / CC) * lda + c), f);
#pragma unroll
      for (int q = 0; q < 8; ++q) s[q] += f[q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) atomicAdd(&lsum[c + q], s[q]);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < C; j += blockDim.x) part[(long long)blockIdx.x * C + j] = lsum[j];
}
// out[c] += sum over the rows of part[r][c]: 32 columns x 32 row lanes per block, the loads of a lane independent
__global__ __launch_bounds__(1024) void colsum_finish_kernel(const float* __restrict__ part, int nrows, int C,
                                                             float* __restrict__ out) {
  __shared__ float sh[32][33];
  const int cl = threadIdx.x & 31, ty = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
  float a = 0.f;
  if (c < C) {
#pragma unroll 8
    for (int r = ty; r < nrows; r += 32) a += part[(long long)r * C + c];
  }
  sh[ty][cl] = a;
  __syncthreads();
  if (ty == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += sh[k][cl];
    out[c] += t;
  }
}

#define ROWSUM_ROWS_PER_BLOCK 1024
__global__ __launch_bounds__(ROWRED_THREADS) void stat_rows_sum_kernel(const float* __restrict__ part, int nrows,
                                                                       int C, float* __restrict__ out) {
  __shared__ float sh[32][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), ty = threadIdx.x >> 5;
  const int r0 = blockIdx.y * ROWSUM_ROWS_PER_BLOCK, r1 = min(nrows, r0 + ROWSUM_ROWS_PER_BLOCK);
  float a = 0.f;
  if (c < C) {
#pragma unroll 4
    for (int r = r0 + ty; r < r1; r += 32) a += part[(long long)r * 2 * C + c];
  }
  sh[ty][threadIdx.x & 31] = a;
  __syncthreads();
  if (ty == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += sh[k][threadIdx.x & 31];
    if (gridDim.y == 1) out[c] += t;
    else atomicAdd(&out[c], t);           // a handful of adders per address
  }
}
// out[0] += scale * sum over rows r and channels c of stats[r][which][c]
__global__ __launch_bounds__(1024) void stat_rows_total_kernel(const float* __restrict__ stats, int nrows, int C,
                                                               int which, float scale, float* __restrict__ out) {
  __shared__ float sh[16];
  const long long total = (long long)nrows * C;
  float a = 0.f;
#pragma unroll 4
  for (long long i = blockIdx.x * 1024ll + threadIdx.x; i < total; i += gridDim.x * 1024ll) {
    long long r = i / C;
    a += stats[(r * 2 + which) * C + (i - r * C)];
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k];
    atomicAdd(out, t * scale);
  }
}
extern "C" int ried_stat_rows_total(const float* stats, int nrows, int C, int which, float scale, float* out,
                                    hipStream_t stream) {
  RIED_CHECK_ARG(stats && out && nrows > 0 && C > 0 && (which == 0 || which == 1), "ried_stat_rows_total: bad arguments");
  hipLaunchKernelGGL(stat_rows_total_kernel, dim3(ew_grid((long long)nrows * C, 4096, 32)), dim3(1024), 0, stream,
                     stats, nrows, C, which, scale, out);
  RIED_CHECK_LAUNCH("ried_stat_rows_total");
  return RIED_OK;
}

// out[c] += sum_r stats[r][0][c]; stats: [nrows][2][C] partial rows written by a conv epilogue
extern "C" int ried_stat_rows_sum(const float* stats, int nrows, int C, float* out, hipStream_t stream) {
  RIED_CHECK_ARG(stats && out && nrows > 0 && C > 0, "ried_stat_rows_sum: bad arguments");
  hipLaunchKernelGGL(stat_rows_sum_kernel, dim3(ceil_div(C, 32), ceil_div(nrows, ROWSUM_ROWS_PER_BLOCK)),
                     dim3(ROWRED_THREADS), 0, stream, stats, nrows, C, out);
  RIED_CHECK_LAUNCH("ried_stat_rows_sum");
  return RIED_OK;
}

#define COLSUM_MAX_ROWS 256
// partial rows [grid][C] into scratch; returns grid
static int colsum_launch(const bf16_t* a, long long lda, long long npix, int C, float* scratch, hipStream_t stream) {
  int grid = ew_grid(npix * (C / 8), EW_THREADS * 8, COLSUM_MAX_ROWS);
  if (C <= 8 * EW_THREADS) {
    hipLaunchKernelGGL(colsum_kernel, dim3(grid), dim3(EW_THREADS), 0, stream, a, lda, npix, C, scratch);
  } else {
    if (grid * EW_THREADS < C / 8) grid = ceil_div(C / 8, EW_THREADS);
    hipLaunchKernelGGL(colsum_wide_kernel, dim3(grid), dim3(EW_THREADS), C * sizeof(float), stream, a, lda, npix, C,
                       scratch);
  }
  return grid;
}
// out[c] += sum_p a[p][c].  scratch: fp32 [COLSUM_MAX_ROWS * C].
extern "C" int ried_colsum(const void* a, long long lda, long long npix, int C, float* out, float* scratch,
                           hipStream_t stream) {
  CHECK_VIEW("ried_colsum", a, lda, C);
  RIED_CHECK_ARG(out && scratch && npix > 0 && C <= 8192, "ried_colsum: bad arguments");
  int grid = colsum_launch((const bf16_t*)a, lda, npix, C, scratch, stream);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3(ceil_div(C, 32)), dim3(1024), 0, stream, scratch, grid, C, out);
  RIED_CHECK_LAUNCH("ried_colsum");
  return RIED_OK;
}
extern "C" long long ried_colsum_scratch_floats(int C) { return (long long)COLSUM_MAX_ROWS * C; }

// ried_colsum for callers that cannot allocate or pass a workspace (HIP graph capture, serving loops): the partial
// rows live in a file-scope device array instead of a caller buffer.  Each GPU holds its own copy of the array at
// its own address; the host-side name below is only a handle for hipGetSymbolAddress, never a kernel argument.
#define COLSUM_STATIC_MAX_C 1024
#define COLSUM_STATIC_MAX_DEVICES 64
static __device__ __align__(256) float colsum_static_scratch[COLSUM_MAX_ROWS * COLSUM_STATIC_MAX_C];   // 1 MiB per GPU
static float* colsum_static_ptr[COLSUM_STATIC_MAX_DEVICES];      // device address of the array, per device ordinal

extern "C" int ried_colsum_static(const void* a, long long lda, long long npix, int C, float* out,
                                  hipStream_t stream) {
  CHECK_VIEW("ried_colsum_static", a, lda, C);
  RIED_CHECK_ARG(out && npix > 0, "ried_colsum_static: bad arguments");
  RIED_CHECK_ARG(C <= COLSUM_STATIC_MAX_C, "ried_colsum_static: C = %d exceeds the built-in scratch (C <= %d); use "
                 "ried_colsum with a caller workspace", C, COLSUM_STATIC_MAX_C);
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= COLSUM_STATIC_MAX_DEVICES) {
    ried_set_error("ried_colsum_static: hipGetDevice: %s (device %d)", hipGetErrorString(e), dev);
    return RIED_ERR_HIP;
  }
  float* scratch = __atomic_load_n(&colsum_static_ptr[dev], __ATOMIC_ACQUIRE);
  if (!scratch) {      // first call on this device (racing threads resolve the same address)
    e = hipGetSymbolAddress((void**)&scratch, HIP_SYMBOL(colsum_static_scratch));
    if (e != hipSuccess || !scratch) {
      ried_set_error("ried_colsum_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
      return RIED_ERR_HIP;
    }
    __atomic_store_n(&colsum_static_ptr[dev], scratch, __ATOMIC_RELEASE);
  }
  // same grid and kernels as ried_colsum: rows [0, grid) of the scratch are fully overwritten for channels [0, C)
  // before the second kernel reads them, so what an earlier call left behind is never read
  int grid = colsum_launch((const bf16_t*)a, lda, npix, C, scratch, stream);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3(ceil_div(C, 32)), dim3(1024), 0, stream, scratch, grid, C, out);
  RIED_CHECK_LAUNCH("ried_colsum_static");
  return RIED_OK;
}

// ---------------------------------------------------------------------------
// softmax over the last dimension (reference: src/rinet.py:174) and its backward
// ---------------------------------------------------------------------------
// one wave per row; S fp32 [rows][n] -> P bf16 [rows][n]
__global__ void softmax_fwd_kernel(const float* __restrict__ S, bf16_t* __restrict__ Pm, long long rows, int n) {
  int lane = threadIdx.x & 63;
  long long row = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* s = S + row * n;
  float mx = -INFINITY;
  for (int j = lane; j < n; j += 64) mx = fmaxf(mx, s[j]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) sum += __expf(s[j] - mx);
  sum = wave_sum(sum);
  float inv = 1.f / sum;
  bf16_t* o = Pm + row * n;
  for (int j = lane; j < n; j += 64) o[j] = f2bf(__expf(s[j] - mx) * inv);
}

// dS = P * (dP - sum_j dP*P); dP fp32, P bf16, dS bf16
__global__ void softmax_bwd_kernel(const float* __restrict__ dP, const bf16_t* __restrict__ Pm,
                                   bf16_t* __restrict__ dS, long long rows, int n) {
  int lane = threadIdx.x & 63;
  long long row = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* d = dP + row * n;
  const bf16_t* p = Pm + row * n;
  float dot = 0.f;
  for (int j = lane; j < n; j += 64) dot += d[j] * bf2f(p[j]);
  dot = wave_sum(dot);
  bf16_t* o = dS + row * n;
  for (int j = lane; j < n; j += 64) o[j] = f2bf(bf2f(p[j]) * (d[j] - dot));
}

// Attention probabilities in one pass (src/rinet.py:171-174): P[b][i][:] = softmax_j(q_i . k_j) without the
// fp32 energies ever reaching memory.  A wave owns 16 query rows of one image and all NK keys: the MFMA
// accumulators hold S[j][i] with i = lane % 16 and four consecutive j per lane, so the row maximum and sum are a
// scan over the lane's registers plus two shuffles, and the bf16 probabilities leave as 8-byte pieces.  The
// fragments (16 bytes of q or k per lane) are read straight from global memory: q and k of an image are 32 KB.
template <int D, int NK>
__global__ __launch_bounds__(256) void attn_scores_softmax_kernel(const bf16_t* __restrict__ qkv, long long ld,
                                                                  bf16_t* __restrict__ Pm) {
  constexpr int NT = NK / 16, KH = D / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long img = blockIdx.y;
  const int i0 = blockIdx.x * 64 + wave * 16;
  const bf16_t* base = qkv + img * NK * ld;
  const int kc = 8 * (lane >> 4);
  bf16x8_t fb[KH];
#pragma unroll
  for (int h = 0; h < KH; ++h)
    fb[h] = *reinterpret_cast<const bf16x8_t*>(base + (long long)(i0 + (lane & 15)) * ld + h * 32 + kc);
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < KH; ++h) {
      const bf16x8_t fa =
          *reinterpret_cast<const bf16x8_t*>(base + (long long)(t * 16 + (lane & 15)) * ld + D + h * 32 + kc);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[h], acc[t], 0, 0, 0);
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) mx = fmaxf(mx, acc[t][q]);
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc[t][q] = __expf(acc[t][q] - mx); sum += acc[t][q]; }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
  bf16_t* o = Pm + (img * NK + i0 + (lane & 15)) * NK + (lane >> 4) * 4;
#pragma unroll
  for (int t = 0; t < NT; ++t)
    *reinterpret_cast<u32x2*>(o + t * 16) =
        u32x2{pack2bf(acc[t][0] * inv, acc[t][1] * inv), pack2bf(acc[t][2] * inv, acc[t][3] * inv)};
}

// qkv: bf16 [nimg][n][ld] with q in channels [0, d) and k in [d, 2d); P: bf16 [nimg][n][n].
// Supported (n, d): (256, 32) and (64, 64) - the two attention layers at 64x64 patches; returns 0 otherwise so
// that callers can ask before planning (ried_attn_scores_softmax_supported).
extern "C" int ried_attn_scores_softmax_supported(int n, int d) { return (n == 256 && d == 32) || (n == 64 && d == 64); }
extern "C" int ried_attn_scores_softmax(const void* qkv, long long ld, int nimg, int n, int d, void* P,
                                        hipStream_t stream) {
  RIED_CHECK_ARG(qkv && P && nimg > 0, "ried_attn_scores_softmax: bad arguments");
  RIED_CHECK_ARG(ried_attn_scores_softmax_supported(n, d), "ried_attn_scores_softmax: unsupported n=%d d=%d", n, d);
  RIED_CHECK_ARG(ld % 8 == 0 && ld >= 2 * d && ((uintptr_t)qkv % 16) == 0 && ((uintptr_t)P % 8) == 0,
                 "ried_attn_scores_softmax: alignment");
  if (n == 256)
    hipLaunchKernelGGL((attn_scores_softmax_kernel<32, 256>), dim3(n / 64, nimg), dim3(256), 0, stream,
                       (const bf16_t*)qkv, ld, (bf16_t*)P);
  else
    hipLaunchKernelGGL((attn_scores_softmax_kernel<64, 64>), dim3(n / 64, nimg), dim3(256), 0, stream,
                       (const bf16_t*)qkv, ld, (bf16_t*)P);
  RIED_CHECK_LAUNCH("ried_attn_scores_softmax");
  return RIED_OK;
}

extern "C" int ried_softmax_fwd(const float* S, void* P, long long rows, int n, hipStream_t stream) {
  RIED_CHECK_ARG(S && P && rows > 0 && n > 0, "ried_softmax_fwd: bad arguments");
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, S, (bf16_t*)P, rows, n);
  RIED_CHECK_LAUNCH("ried_softmax_fwd");
  return RIED_OK;
}
extern "C" int ried_softmax_bwd(const float* dP, const void* P, void* dS, long long rows, int n, hipStream_t stream) {
  RIED_CHECK_ARG(dP && P && dS && rows > 0 && n > 0, "ried_softmax_bwd: bad arguments");
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, dP, (const bf16_t*)P,
                     (bf16_t*)dS, rows, n);
  RIED_CHECK_LAUNCH("ried_softmax_bwd");
  return RIED_OK;
}

// ---------------------------------------------------------------------------
// dot / sums to device scalars
// ---------------------------------------------------------------------------
// out[0] += scale * sum_{p,c} a*b   (b may be null: plain sum of a)
// Few, large blocks: each block ends in one atomic add on the same address, and those serialise.
#define DOT_THREADS 1024
__global__ __launch_bounds__(DOT_THREADS) void dot_kernel(const bf16_t* __restrict__ a, long long lda,
                                                          const bf16_t* __restrict__ b, long long ldb, long long npix,
                                                          int C, float scale, float* __restrict__ out) {
  __shared__ float sh[DOT_THREADS / 64];
  ChanWalk w(C);
  float acc = 0.f;
  if (w.active) {
    const int c = w.c;
#pragma unroll 4
    for (long long p = w.p; p < npix; p += w.step) {
      float fa[8], fb[8];
      unpack8(ld8(a + p * lda + c), fa);
      if (b) {
        unpack8(ld8(b + p * ldb + c), fb);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += fa[q] * fb[q];
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += fa[q];
      }
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < DOT_THREADS / 64; ++k) t += sh[k];
    atomicAdd(out, t * scale);
  }
}

extern "C" int ried_dot(const void* a, long long lda, const void* b, long long ldb, long long npix, int C, float scale,
                        float* out, hipStream_t stream) {
  CHECK_VIEW("ried_dot", a, lda, C);
  if (b) CHECK_VIEW("ried_dot", b, ldb, C);
  RIED_CHECK_ARG(out && npix > 0, "ried_dot: bad arguments");
  RIED_CHECK_ARG(C <= 8 * DOT_THREADS, "ried_dot: C too large");
  hipLaunchKernelGGL(dot_kernel, dim3(ew_grid(npix, (DOT_THREADS / (C / 8)) * 8, 128)), dim3(DOT_THREADS), 0, stream,
                     (const bf16_t*)a, lda, (const bf16_t*)b, ldb, npix, C, scale, out);
  RIED_CHECK_LAUNCH("ried_dot");
  return RIED_OK;
}

// ---------------------------------------------------------------------------
// fused Adam over a flat fp32 parameter buffer (torch.optim.Adam defaults; reference: src/train.py:104-105)
// ---------------------------------------------------------------------------
__global__ void adam_tick_kernel(int* step) {
  if (threadIdx.x == 0 && blockIdx.x == 0) step[0] += 1;
}
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                            const int* __restrict__ step) {
  const float st = (float)step[0];
  const float bc1 = 1.f - powf(b1, st);
  const float bc2_sqrt = sqrtf(1.f - powf(b2, st));
  const float a = lr / bc1;
  for (long long i = (blockIdx.x * (long long)blockDim.x + threadIdx.x) * 4; i < n;
       i += (long long)gridDim.x * blockDim.x * 4) {
    if (i + 4 <= n) {
      f32x4 pp = *reinterpret_cast<f32x4*>(p + i), gg = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mm[q] = b1 * mm[q] + (1.f - b1) * gg[q];
        vv[q] = b2 * vv[q] + (1.f - b2) * gg[q] * gg[q];
        float denom = sqrtf(vv[q]) / bc2_sqrt + eps;
        pp[q] -= a * (mm[q] / denom);
      }
      *reinterpret_cast<f32x4*>(p + i) = pp;
      *reinterpret_cast<f32x4*>(m + i) = mm;
      *reinterpret_cast<f32x4*>(v + i) = vv;
    } else {
      for (long long j = i; j < n; ++j) {
        float mj = b1 * m[j] + (1.f - b1) * g[j];
        float vj = b2 * v[j] + (1.f - b2) * g[j] * g[j];
        m[j] = mj; v[j] = vj;
        p[j] -= a * (mj / (sqrtf(vj) / bc2_sqrt + eps));
      }
    }
  }
}

// step: device int32 counter of completed steps; it is incremented first, then used for the bias corrections,
// so a captured HIP graph replays with the right step number.
extern "C" int ried_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2,
                         float eps, int* step, hipStream_t stream) {
  RIED_CHECK_ARG(p && g && m && v && step && n > 0, "ried_adam: bad arguments");
  RIED_CHECK_ARG(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 &&
                     ((uintptr_t)v % 16) == 0, "ried_adam: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, stream, step);
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n, EW_THREADS * 4, 2048)), dim3(EW_THREADS), 0, stream, p, g, m, v, n,
                     lr, b1, b2, eps, step);
  RIED_CHECK_LAUNCH("ried_adam");
  return RIED_OK;
}

// ---------------------------------------------------------------------------
// layout / dtype conversion at the module boundary (the reference API is NCHW fp32)
// ---------------------------------------------------------------------------
// NCHW fp32 -> NHWC bf16 view (C % 8 == 0)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long long ldy, int nimg,
                                    int C, long long HW) {
  const int CC = C >> 3;
  long long total = (long long)nimg * HW * CC;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    // consecutive threads walk hw first so the fp32 reads coalesce
    long long hw = i % HW;
    long long t = i / HW;
    int c = (int)(t % CC) << 3;
    int n = (int)(t / CC);
    float f[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) f[q] = x[((long long)n * C + c + q) * HW + hw];
    st8(y + ((long long)n * HW + hw) * ldy + c, pack8(f));
  }
}
// NHWC bf16 view -> NCHW fp32
__global__ void nhwc_to_nchw_kernel(const bf16_t* __restrict__ x, long long ldx, float* __restrict__ y, int nimg,
                                    int C, long long HW) {
  const int CC = C >> 3;
  long long total = (long long)nimg * HW * CC;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long hw = i % HW;
    long long t = i / HW;
    int c = (int)(t % CC) << 3;
    int n = (int)(t / CC);
    float f[8];
    unpack8(ld8(x + ((long long)n * HW + hw) * ldx + c), f);
#pragma unroll
    for (int q = 0; q < 8; ++q) y[((long long)n * C + c + q) * HW + hw] = f[q];
  }
}

extern "C" int ried_nchw_to_nhwc(const float* x, void* y, long long ldy, int nimg, int C, long long HW,
                                 hipStream_t stream) {
  CHECK_VIEW("ried_nchw_to_nhwc", y, ldy, C);
  RIED_CHECK_ARG(x && nimg > 0 && HW > 0, "ried_nchw_to_nhwc: bad arguments");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_grid((long long)nimg * HW * (C / 8), EW_THREADS)), dim3(EW_THREADS),
                     0, stream, x, (bf16_t*)y, ldy, nimg, C, HW);
  RIED_CHECK_LAUNCH("ried_nchw_to_nhwc");
  return RIED_OK;
}
extern "C" int ried_nhwc_to_nchw(const void* x, long long ldx, float* y, int nimg, int C, long long HW,
                                 hipStream_t stream) {
  CHECK_VIEW("ried_nhwc_to_nchw", x, ldx, C);
  RIED_CHECK_ARG(y && nimg > 0 && HW > 0, "ried_nhwc_to_nchw: bad arguments");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_grid((long long)nimg * HW * (C / 8), EW_THREADS)), dim3(EW_THREADS),
                     0, stream, (const bf16_t*)x, ldx, y, nimg, C, HW);
  RIED_CHECK_LAUNCH("ried_nhwc_to_nchw");
  return RIED_OK;
}
