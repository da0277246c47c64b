// Implicit-GEMM family for gfx950 (CDNA4): bf16 operands, fp32 accumulate on
// v_mfma_f32_16x16x32_bf16, LDS-staged tiles, 64-wide wavefronts.
//
// One kernel template computes   out[n][m] = sum_k A[m][k] * B[n][k]
// with m the contiguous output dimension (NHWC channels), for four operand
// shapes:
//   PLAIN_K   matrix stored [row][k]           (packed weights, q/k, P, dS)
//   PLAIN_TR  matrix stored [k][row]           (v, dO, dy in weight-gradient)
//   GATHER_K  NHWC activations, row = output pixel, k = (tap, channel):
//             the im2col gather of a convolution (or of a transposed
//             convolution, decomposed by output phase so no MFMA work is
//             spent on structurally-zero taps)
//   GATHER_TR NHWC activations, k = output pixel at a fixed tap, row =
//             channel (the x operand of a weight gradient)
// [k][row] tiles are consumed with ds_read_b64_tr_b16 (hardware transpose),
// [row][k] tiles with ds_read_b128.
//
// Uses: Conv2d / ConvTranspose2d forward, data gradient and weight gradient
// of every bf16 layer of the generator, discriminator and VGG16 extractor
// (reference: src/rinet.py:183-205, 131-148; src/loss.py:69-87), and the
// batched products of the self-attention layers (src/rinet.py:162-181).
#include <stdlib.h>
#include "common.h"
#include "bn_common.h"

enum { OP_PLAIN_K = 0, OP_PLAIN_TR = 1, OP_GATHER_K = 2, OP_GATHER_TR = 3 };

struct GemmOperand {
  const bf16_t* p;
  long long ld;            // plain: stride between rows (K) or between k (TR); gather: pixel stride
  long long batch_stride;  // plain, batched
  int rows;                // valid extent along m (operand A) or n (operand B)
  int nimg, H, W;          // gathered tensor
  int Ho, Wo;              // logical output grid per image (transposed: the per-phase grid)
  int kh, kw, sh, sw, ph, pw;
  int transposed;
  int C;                   // channels of the gathered tensor
};

struct GemmEpilogue {
  void* out;
  long long ldc, batch_stride, tap_stride;
  int out_f32;
  const float* bias;
  int act;                 // 0 none, 1 relu, 2 leaky relu 0.2
  float* stats;            // [2][M] sum / sum of squares (fp32 atomics) or null
  int accumulate;          // out += value
  const float* gamma;      // out = gamma[0]*value + resid (self-attention) or null
  const bf16_t* resid;
  long long ld_resid, resid_batch_stride;
  float alpha;
  float* out2;             // optional second copy of the result in fp32 (same row mapping), or null
  long long ldc2;
  const bf16_t* mask;      // optional: final value *= (mask[row][m] > 0 ? 1 : mask_slope)  (activation backward
  long long ld_mask;       // fused into the data gradient that produces the tensor); applied after accumulate
  float mask_slope;
  int mask_bits;           // 1: mask is sign BYTES instead ([row][ld_mask bytes], bit q of byte c = channel 8c + q positive)
  int stats_post;          // stats = column sums of the final values (bias gradient) instead of the raw results
  int stats_fold;          // > 0: rows are added (float atomics) into row (tile mod stats_fold) of a zeroed buffer
  int stats_img_tiles;     // > 0: rows are added into row (pixel tile / stats_img_tiles) = the tile's image
  int stats_zero;          // f > 1: the caller sized the buffer for f times as many rows as the stats_rows this
  int stats_rows;          //        launch writes; the writer of row t also zeroes rows stats_rows + t*(f-1) .. + f-2
  BnFinalize fin;          // fin.counter != null: the last block to finish turns the stats rows into BN statistics
  const float* sub;        // perceptual-loss tap: d = value - sub[row][m] (the target's features, fp32);
  int sub_bf16;            //   1: `sub` points at bf16 values (the target's stored activations), ld_sub in elements
  long long ld_sub;        //   stats rows = column sums of d and d*d, gout[row][m] = gscale * d (the seed of the
  bf16_t* gout;            //   tap's gradient); the value itself is still stored as the layer's output
  long long ld_gout;
  float gscale;
  int phase_mode, Wo_full, HoWo_full, Hq, Wq, sh, sw;   // logical row -> physical pixel
  // conv_halo_s2t_kernel<16, true> only: the result (64 channels: the gradient of the first discriminator layer's
  // pre-activation) is not stored; its product with the nine taps of the fp32 image that layer reads - that layer's
  // weight gradient - leaves the block as one row [64][9] of fold_rows
  const float* fold_x;     // fp32 [nimg][2H][2W], used as fold_x * fold_scale + fold_shift, zero padded
  float fold_scale, fold_shift;
  float* fold_rows;        // [pixel tiles][64 * 9]
  int staged;              // bf16 output with every view 16-byte aligned: the convolution kernels may take the
                           // LDS-staged epilogue (full-line loads and stores)
};

struct GemmParams {
  GemmOperand A, B;
  GemmEpilogue E;
  int M, N;
  int K;          // per tap (conv: Cin), plain: K, weight gradient: total output pixels
  int taps;       // conv: kh*kw; otherwise 1 (transposed: computed per phase)
  int splitk;
  int tiles_m;
  unsigned A_bytes, B_bytes;   // extent of the operands in bytes (buffer descriptors of the LDS-DMA kernel)
  int grid[3];                 // the problem's own grid (job-table launches)
  // Input transform of the convolution kernel (xf_save != null): the B operand is the RAW output of a BasicConv2d
  // layer's convolution and the kernel applies that layer's train-mode BatchNorm and activation on the fly,
  // x -> act(gamma * (x - mean) * invstd + beta), with xf_save = [2][K] (mean, invstd), so that the normalised
  // tensor is never written or read.  Padding stays zero (it pads the activated tensor).
  const float* xf_save;
  const float* xf_gamma;
  const float* xf_beta;
  int xf_act;                  // 0 none, 1 ReLU
};

constexpr int BK = 64;                            // k per stage: two MFMA k-slices of 32
constexpr int LDK_PAD = 8;

template <int R>
struct TileGeom {
  static constexpr int NCH = R * (BK / 8) / 256;  // 16-byte chunks per thread per tile
  static constexpr int LDK = BK + LDK_PAD;        // [row][k] image: elements per row (144-byte rows: the 16 rows
                                                  // of a ds_read_b128 group fall on 16 different 16-byte bank slots)
  static constexpr int LDR = R;                   // [k][row] image: elements per k, XOR-swizzled (tr_swz)
  static constexpr int ELEMS = (R * LDK > BK * LDR) ? R * LDK : BK * LDR;
  static constexpr int CPR = R / 8;               // chunks per k row (TR)
  static constexpr int KR = 256 / CPR;            // k rows per pass (TR)
  // [k][row] image: the 32-byte slot (16 rows of the operand) s of k-row k is stored in slot s ^ h(k), so that the
  // 8 k-rows one 32-lane half of a ds_read_b64_tr_b16 touches (k = 8g + q; g = 0..1, q = 0..3, same columns) fall
  // on 8 different 32-byte slots of the 256-byte bank window.  Returns the element offset to XOR with.
  __device__ static __forceinline__ int tr_swz(int k) {
    if constexpr (R >= 128) return ((((k >> 3) & 1) << 2) | (k & 3)) << 4;
    else return ((((k >> 3) & 1) << 1) | ((k >> 1) & 1)) << 4;      // 128-byte rows: two k-rows per bank window
  }
};

// ---------------------------------------------------------------------------
// operand loaders: global -> registers -> LDS
// ---------------------------------------------------------------------------
template <int KIND, int R>
struct Loader {
  using G = TileGeom<R>;
  const bf16_t* ptr[G::NCH];
  bool valid[G::NCH];
  int c8;                       // element offset of this thread's chunk inside a row
  // gather state
  int gn[G::NCH], gy[G::NCH], gx[G::NCH];
  int dn, dy, dx;               // mixed-radix increment of BK pixels (GATHER_TR)
  int kidx[G::NCH];             // PLAIN_TR / GATHER_TR: k row of chunk i inside the tile

  __device__ __forceinline__ void init(const GemmOperand& op, int row0, long long batch, int kbeg,
                                       int pa, int pb, int r0, int s0) {
    const int t = threadIdx.x;
    if constexpr (KIND == OP_PLAIN_K) {
      c8 = (t & 7) * 8;
#pragma unroll
      for (int i = 0; i < G::NCH; ++i) {
        int rowg = row0 + (t >> 3) + 32 * i;
        valid[i] = rowg < op.rows;
        ptr[i] = op.p + batch * op.batch_stride + (long long)rowg * op.ld + c8;
      }
    } else if constexpr (KIND == OP_GATHER_K) {
      c8 = (t & 7) * 8;
#pragma unroll
      for (int i = 0; i < G::NCH; ++i) {
        int rowg = row0 + (t >> 3) + 32 * i;
        valid[i] = rowg < op.rows;
        int hw = op.Ho * op.Wo;
        int n = rowg / hw, rem = rowg - n * hw;
        int oy = rem / op.Wo, ox = rem - oy * op.Wo;
        gn[i] = n;
        if (op.transposed) {      // hi = oy + (a + ph - r0)/sh - i_tap
          gy[i] = oy + (pa + op.ph - r0) / op.sh;
          gx[i] = ox + (pb + op.pw - s0) / op.sw;
        } else {                  // hi = oy*sh - ph + r
          gy[i] = oy * op.sh - op.ph;
          gx[i] = ox * op.sw - op.pw;
        }
        ptr[i] = op.p;
      }
    } else if constexpr (KIND == OP_PLAIN_TR) {
      c8 = (t % G::CPR) * 8;
      bool cv = row0 + c8 < op.rows;
#pragma unroll
      for (int i = 0; i < G::NCH; ++i) {
        kidx[i] = t / G::CPR + G::KR * i;
        valid[i] = cv;
        ptr[i] = op.p + batch * op.batch_stride + row0 + c8;
      }
    } else {  // OP_GATHER_TR: k = output pixel, fixed tap (r0, s0)
      c8 = (t % G::CPR) * 8;
      bool cv = row0 + c8 < op.C;
      int hw = op.Ho * op.Wo;
      dn = BK / hw;
      int rem = BK - dn * hw;
      dy = rem / op.Wo;
      dx = rem - dy * op.Wo;
#pragma unroll
      for (int i = 0; i < G::NCH; ++i) {
        kidx[i] = t / G::CPR + G::KR * i;
        int k = kbeg + kidx[i];
        int n = k / hw, r2 = k - n * hw;
        int oy = r2 / op.Wo, ox = r2 - oy * op.Wo;
        gn[i] = n; gy[i] = oy; gx[i] = ox;
        valid[i] = cv;
        ptr[i] = op.p + row0 + c8;
      }
    }
  }

  // GATHER_K: a new tap (ti, tj) -> row pointers and validity
  __device__ __forceinline__ void set_tap(const GemmOperand& op, int ti, int tj) {
    if constexpr (KIND == OP_GATHER_K) {
#pragma unroll
      for (int i = 0; i < G::NCH; ++i) {
        int hi = op.transposed ? gy[i] - ti : gy[i] + ti;
        int wi = op.transposed ? gx[i] - tj : gx[i] + tj;
        bool ok = (unsigned)hi < (unsigned)op.H && (unsigned)wi < (unsigned)op.W;
        // keep the row-valid bit in the sign of gn so it survives tap changes
        valid[i] = ok && gn[i] >= 0;
        ptr[i] = op.p + ((long long)(gn[i] * op.H + hi) * op.W + wi) * op.ld + c8;
      }
    }
  }

  __device__ __forceinline__ void mark_invalid_rows() {
    if constexpr (KIND == OP_GATHER_K) {
#pragma unroll
      for (int i = 0; i < G::NCH; ++i)
        if (!valid[i]) gn[i] = -1;
    }
  }

  // koff: element offset along k for PLAIN_K (tap offset + kc); kc: offset inside the tap;
  // kt: k extent of the tap (or end of this block's k range for TR kinds, with kglob the tile's first k)
  __device__ __forceinline__ void load(const GemmOperand& op, u32x4 (&v)[G::NCH], long long koff, int kc, int kt,
                                       int kglob, int r0, int s0) {
    if constexpr (KIND == OP_PLAIN_K) {
#pragma unroll
      for (int i = 0; i < G::NCH; ++i) {
        bool ok = valid[i] && (kc + c8 < kt);
        v[i] = ok ? *reinterpret_cast<const u32x4*>(ptr[i] + koff + kc) : u32x4{0, 0, 0, 0};
      }
    } else if constexpr (KIND == OP_GATHER_K) {
#pragma unroll
      for (int i = 0; i < G::NCH; ++i) {
        bool ok = valid[i] && (kc + c8 < kt);
        v[i] = ok ? *reinterpret_cast<const u32x4*>(ptr[i] + kc) : u32x4{0, 0, 0, 0};
      }
    } else if constexpr (KIND == OP_PLAIN_TR) {
#pragma unroll
      for (int i = 0; i < G::NCH; ++i) {
        int k = kglob + kidx[i];
        bool ok = valid[i] && k < kt;
        v[i] = ok ? *reinterpret_cast<const u32x4*>(ptr[i] + (long long)k * op.ld) : u32x4{0, 0, 0, 0};
      }
    } else {
#pragma unroll
      for (int i = 0; i < G::NCH; ++i) {
        int k = kglob + kidx[i];
        int hi = gy[i] * op.sh - op.ph + r0;
        int wi = gx[i] * op.sw - op.pw + s0;
        bool ok = valid[i] && k < kt && (unsigned)hi < (unsigned)op.H && (unsigned)wi < (unsigned)op.W;
        const bf16_t* q = ptr[i] + ((long long)(gn[i] * op.H + hi) * op.W + wi) * op.ld;
        v[i] = ok ? *reinterpret_cast<const u32x4*>(q) : u32x4{0, 0, 0, 0};
        // advance this chunk's pixel by BK for the next tile
        int x = gx[i] + dx, y = gy[i] + dy, n = gn[i] + dn;
        if (x >= op.Wo) { x -= op.Wo; ++y; }
        if (y >= op.Ho) { y -= op.Ho; ++n; }
        gx[i] = x; gy[i] = y; gn[i] = n;
      }
    }
  }

  __device__ __forceinline__ void store(bf16_t* lds, const u32x4 (&v)[G::NCH]) const {
    const int t = threadIdx.x;
    if constexpr (KIND == OP_PLAIN_K || KIND == OP_GATHER_K) {
#pragma unroll
      for (int i = 0; i < G::NCH; ++i) {
        int r = (t >> 3) + 32 * i;
        *reinterpret_cast<u32x4*>(lds + r * G::LDK + c8) = v[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < G::NCH; ++i)
        *reinterpret_cast<u32x4*>(lds + kidx[i] * G::LDR + (c8 ^ G::tr_swz(kidx[i]))) = v[i];
    }
  }
};

// fragment of 16 rows x 32 k (k-slice ks of the stage) for the MFMA: lane l holds row (l&15), k = 8*(l>>4) .. +7
template <int KIND, int R>
__device__ __forceinline__ bf16x8_t read_frag(const bf16_t* lds, int rowbase, int lane, int ks) {
  using G = TileGeom<R>;
  if constexpr (KIND == OP_PLAIN_K || KIND == OP_GATHER_K) {
    const bf16_t* q = lds + (rowbase + (lane & 15)) * G::LDK + ks * 32 + (lane >> 4) * 8;
    return *reinterpret_cast<const bf16x8_t*>(q);
  } else {
    // [k][row] image; ds_read_b64_tr_b16 per 16-lane group: lane 4q+p supplies the address of
    // k row q, rows-of-operand 4p..4p+3; lane i receives operand row i for the 4 k rows.
    int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int k = ks * 32 + 8 * g + q;                      // tr_swz(k) == tr_swz(k + 4)
    const bf16_t* a0 = lds + k * G::LDR + ((rowbase + 4 * p) ^ G::tr_swz(k));
    const bf16_t* a1 = a0 + 4 * G::LDR;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf16x8_t, r);
  }
}

// ---------------------------------------------------------------------------
// epilogue shared by the GEMM kernels: acc[i][j] is the 16x16 tile (i along m, j along n) of wave (wm, wn)
// ---------------------------------------------------------------------------
// Row n of the GEMM (counted from the block's first) -> row of the output tensor; false: outside the problem.
struct LinearRowMap {
  const GemmParams& P;
  int n0, pa, pb;
  __device__ __forceinline__ bool operator()(int nl, long long& orow) const {
    const GemmEpilogue& E = P.E;
    const int n = n0 + nl;
    orow = n;
    if (E.phase_mode) {          // logical (img, hq, wq) of this phase -> physical pixel
      int hw = E.Hq * E.Wq;
      int rem, wq;
      int img = fast_div(n, hw, 1.0f / (float)hw, rem);
      int hq = fast_div(rem, E.Wq, 1.0f / (float)E.Wq, wq);
      orow = (long long)img * E.HoWo_full + (long long)(hq * E.sh + pa) * E.Wo_full + (wq * E.sw + pb);
    }
    return n < P.N;
  }
};

// Waves are arranged 2 (along m) x WN (along n); wave (wm, wn) owns rows [wm*BM/2, +BM/2) x [wn*BN/WN, +BN/WN).
template <int BM, int BN, bool WGRAD, int WN = 2, class RowMap = LinearRowMap>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& P, f32x4 (&acc)[BM / 32][BN / (16 * WN)], int m0,
                                              const RowMap& rowmap, int tn, long long batch, int ntiles_n, int by,
                                              int bz, unsigned nblocks, void* smem) {
  constexpr int TM = BM / 32, TN = BN / (16 * WN);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const GemmEpilogue& E = P.E;
  const float gam = E.gamma ? E.gamma[0] : 0.f;
  char* outb = (char*)E.out;
  long long obase = batch * E.batch_stride + (long long)by * E.tap_stride;
  if constexpr (WGRAD) obase += (long long)bz * E.batch_stride;
  float ssum[TM][4], ssq[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) { ssum[i][q] = 0.f; ssq[i][q] = 0.f; }

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    long long orow;
    const bool nok = rowmap(wn * (BN / WN) + j * 16 + (lane & 15), orow);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int m = m0 + wm * (BM / 2) + i * 16 + (lane >> 4) * 4;
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = acc[i][j][q] * E.alpha;
      if (E.stats && !E.stats_post && !E.sub) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { ssum[i][q] += v[q]; ssq[i][q] += v[q] * v[q]; }
      }
      if (!(nok && m < P.M)) continue;
      if (E.bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += E.bias[m + q];
      }
      if (E.act == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
      } else if (E.act == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.2f * v[q];
      }
      if (E.gamma) {
        const bf16_t* rp = E.resid + batch * E.resid_batch_stride + orow * E.ld_resid + m;
        u32x2 rr = *reinterpret_cast<const u32x2*>(rp);
        v[0] = gam * v[0] + bflo(rr[0]); v[1] = gam * v[1] + bfhi(rr[0]);
        v[2] = gam * v[2] + bflo(rr[1]); v[3] = gam * v[3] + bfhi(rr[1]);
      }
      if (E.out2) *reinterpret_cast<f32x4*>(E.out2 + orow * E.ldc2 + m) = f32x4{v[0], v[1], v[2], v[3]};
      if (E.sub) {
        f32x4 t = *reinterpret_cast<const f32x4*>(E.sub + orow * E.ld_sub + m);
        float d[4] = {v[0] - t[0], v[1] - t[1], v[2] - t[2], v[3] - t[3]};
#pragma unroll
        for (int q = 0; q < 4; ++q) { ssum[i][q] += d[q]; ssq[i][q] += d[q] * d[q]; }
        *reinterpret_cast<u32x2*>(E.gout + orow * E.ld_gout + m) =
            u32x2{pack2bf(E.gscale * d[0], E.gscale * d[1]), pack2bf(E.gscale * d[2], E.gscale * d[3])};
      }
      long long off = obase + orow * E.ldc + m;
      if (E.accumulate) {
        if (E.out_f32) {
          f32x4 o = *reinterpret_cast<const f32x4*>((const float*)outb + off);
          v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
        } else {
          u32x2 o = *reinterpret_cast<const u32x2*>((const bf16_t*)outb + off);
          v[0] += bflo(o[0]); v[1] += bfhi(o[0]); v[2] += bflo(o[1]); v[3] += bfhi(o[1]);
        }
      }
      if (E.mask && E.mask_bits) {
        const unsigned b = reinterpret_cast<const unsigned char*>(E.mask)[orow * E.ld_mask + (m >> 3)] >> (m & 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] *= ((b >> q) & 1u) ? 1.f : E.mask_slope;
      } else if (E.mask) {
        u32x2 mk = *reinterpret_cast<const u32x2*>(E.mask + orow * E.ld_mask + m);
        v[0] *= bflo(mk[0]) > 0.f ? 1.f : E.mask_slope; v[1] *= bfhi(mk[0]) > 0.f ? 1.f : E.mask_slope;
        v[2] *= bflo(mk[1]) > 0.f ? 1.f : E.mask_slope; v[3] *= bfhi(mk[1]) > 0.f ? 1.f : E.mask_slope;
      }
      if (E.stats && E.stats_post) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { ssum[i][q] += v[q]; ssq[i][q] += v[q] * v[q]; }
      }
      if (E.out_f32) *reinterpret_cast<f32x4*>((float*)outb + off) = f32x4{v[0], v[1], v[2], v[3]};
      else *reinterpret_cast<u32x2*>((bf16_t*)outb + off) = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    }
  }
  if (E.stats) {
    // Per-block partial sums, one row of [2][M] per (phase, pixel tile): same-address float atomics serialise at
    // about 0.1 us each on this part, which at thousands of blocks would dwarf the convolution itself.
    // Rows n that are out of range were loaded as zeros, so they add nothing.
    float* red = reinterpret_cast<float*>(smem);          // [WN-1][2][BM]; the k loop ended with a barrier
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ssum[i][q] = row_class_sum<1>(ssum[i][q]);      // over the 16 pixel columns a row of lanes holds
        ssq[i][q] = row_class_sum<1>(ssq[i][q]);
      }
    }
    if (wn > 0 && (lane & 15) == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int ml = wm * (BM / 2) + i * 16 + (lane >> 4) * 4 + q;
          red[(wn - 1) * 2 * BM + ml] = ssum[i][q]; red[(wn - 1) * 2 * BM + BM + ml] = ssq[i][q];
        }
    }
    __syncthreads();
    if (wn == 0 && (lane & 15) == 0) {
      long long ridx = (long long)by * ntiles_n + tn;
      if (E.stats_img_tiles) ridx = tn / E.stats_img_tiles;
      else if (E.stats_fold) ridx %= E.stats_fold;
      float* row = E.stats + ridx * 2 * P.M;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int ml = wm * (BM / 2) + i * 16 + (lane >> 4) * 4 + q;
          int m = m0 + ml;
          if (m < P.M) {
            float s0 = ssum[i][q], s1 = ssq[i][q];
#pragma unroll
            for (int w = 0; w < WN - 1; ++w) { s0 += red[w * 2 * BM + ml]; s1 += red[w * 2 * BM + BM + ml]; }
            if (E.stats_fold || E.stats_img_tiles) {     // few rows: a short chain of atomics each
              __hip_atomic_fetch_add(&row[m], s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_fetch_add(&row[P.M + m], s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (E.fin.counter) {           // read by the last block of this launch: write-through
              ried_publish(&row[m], s0);
              ried_publish(&row[P.M + m], s1);
            } else {
              row[m] = s0;
              row[P.M + m] = s1;
              for (int e = 0; e < E.stats_zero - 1; ++e) {
                float* z = E.stats + ((long long)E.stats_rows + ((long long)by * ntiles_n + tn) * (E.stats_zero - 1) + e) *
                                         2 * P.M;
                z[m] = 0.f; z[P.M + m] = 0.f;
              }
            }
          }
        }
    }
    if (E.fin.counter) {       // fused BatchNorm finalize: saves a dependent launch per layer
      float* scratch = reinterpret_cast<float*>(smem) + 1024;
      if (ried_last_block(E.fin.counter, nblocks, reinterpret_cast<int*>(smem) + 768))
        bn_finalize_rows(E.stats, P.M, E.fin, scratch);
    }
  }
}

// ---------------------------------------------------------------------------
// The same epilogue for the convolution kernels, staged through LDS.  In the MFMA layout a lane owns 4 channels of
// one pixel, so the direct form above reads and writes 8-byte pieces: a wave instruction touches 16 pixel rows
// with 32 bytes each.  For layers whose output (plus mask, accumulate or tap traffic) is large - measured: more
// than half the time of the stride-2 data gradients - that is what bounds the kernel.  Here the accumulators go
// to LDS as an fp32 [pixel][channel] tile, half of the block's channels at a time, and every thread then owns a
// 16-byte chunk (8 channels) of a pixel row: 8 or 4 consecutive lanes read and write one contiguous 128 / 64
// bytes of each tensor.  Statistics are summed per thread (its chunk column is fixed), across the lanes of a wave
// by shuffles and across waves through LDS.
// Handles bf16 outputs with bias, activation, fp32 copy, perceptual tap, accumulate, mask and statistics rows
// (plain, folded, zero-padded); not the BatchNorm hand-off, the attention residual or fp32 outputs.
// ---------------------------------------------------------------------------
// SMEM: bytes of LDS the kernel owns; if the whole tile fits it is staged at once, else in two halves.
// CARRY (kernels that run several epilogues per block - the output phases of a tile; whole-tile staging only):
// 1 = the thread's statistics sums are added into carry[16] and nothing is reduced or written, 2 = carry is
// added in before the block's one reduction.  A thread's chunk column is the same in every call.
// FOLD: the final bf16 chunks go to the LDS tile dt ([pixel row of the tile][FOLD_DT_PITCH bytes]) instead of the
// output tensor (rows outside the problem as zeros).
constexpr int FOLD_DT_PITCH = 144, FOLD_XT_PITCH = 80;
template <int BM, int BN, int WN, int SMEM, class RowMap, int CARRY = 0, bool FOLD = false>
__device__ __forceinline__ void gemm_epilogue_staged(const GemmParams& P, f32x4 (&acc)[BM / 32][BN / (16 * WN)],
                                                     int m0, const RowMap& rowmap, int tn, int ntiles_n, int by,
                                                     void* smem, float* carry = nullptr, char* dt = nullptr) {
  constexpr int NT = 128 * WN, NW = 2 * WN, TM = BM / 32, TN = BN / (16 * WN);
  constexpr int HALVES = (BN * (BM + 4) * 4 + NW * 2 * BM * 4 <= SMEM) ? 1 : 2;
  static_assert(BN * (BM / HALVES + 4) * 4 + NW * 2 * (BM / HALVES) * 4 <= SMEM, "staging area");
  static_assert(CARRY == 0 || HALVES == 1, "carried statistics need whole-tile staging");
  constexpr int HM = BM / HALVES;             // channels per pass: all, or the rows of waves wm = 0, then wm = 1
  constexpr int LDT = HM + 4;                 // floats per staged pixel row (16-byte pad: conflict-free both ways)
  constexpr int CPR = HM / 8;                 // 16-byte output chunks per pixel row
  constexpr int RPP = NT / CPR;               // pixel rows per pass of the block
  static_assert(BN % RPP == 0, "tile shape");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const GemmEpilogue& E = P.E;
  float* T = reinterpret_cast<float*>(smem);
  float* red = T + BN * LDT;                  // [NW][2][HM] wave totals of the statistics
  const int c = tid % CPR, r0 = tid / CPR;
  bf16_t* outb = (bf16_t*)E.out + (long long)by * E.tap_stride;
  const bool pre = E.stats && !E.stats_post && !E.sub;
#pragma unroll
  for (int h = 0; h < HALVES; ++h) {
    if (h) __syncthreads();                   // the first half has been read
    if (HALVES == 1 || wm == h) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = wn * (BN / WN) + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          f32x4 v = acc[i][j];
          v[0] *= E.alpha; v[1] *= E.alpha; v[2] *= E.alpha; v[3] *= E.alpha;
          *reinterpret_cast<f32x4*>(T + n * LDT + (HALVES == 1 ? wm * (BM / 2) : 0) + i * 16 + (lane >> 4) * 4) = v;
        }
      }
    }
    __syncthreads();
    const int m = m0 + h * HM + c * 8;
    const bool mok = m < P.M;
    float bias8[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) bias8[q] = (E.bias && mok) ? E.bias[m + q] : 0.f;
    float s0[8], s1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { s0[q] = 0.f; s1[q] = 0.f; }
#pragma unroll 2
    for (int n = r0; n < BN; n += RPP) {
      float v[8];
      {
        const f32x4 a = *reinterpret_cast<const f32x4*>(T + n * LDT + c * 8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(T + n * LDT + c * 8 + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
      }
      if (pre) {                              // rows outside the problem were computed from zeros
#pragma unroll
        for (int q = 0; q < 8; ++q) { s0[q] += v[q]; s1[q] += v[q] * v[q]; }
      }
      long long orow;
      if (!(rowmap(n, orow) && mok)) {
        if constexpr (FOLD) *reinterpret_cast<u32x4*>(dt + n * FOLD_DT_PITCH + c * 16) = u32x4{0, 0, 0, 0};
        continue;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] += bias8[q];
      if (E.act == 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
      } else if (E.act == 2) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = v[q] > 0.f ? v[q] : 0.2f * v[q];
      }
      if (E.out2) {
        float* o2 = E.out2 + orow * E.ldc2 + m;
        *reinterpret_cast<f32x4*>(o2) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(o2 + 4) = f32x4{v[4], v[5], v[6], v[7]};
      }
      if (E.sub) {
        f32x4 t0, t1;
        if (E.sub_bf16) {
          const u32x4 tb = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(E.sub) + orow * E.ld_sub + m);
          t0 = f32x4{bflo(tb[0]), bfhi(tb[0]), bflo(tb[1]), bfhi(tb[1])};
          t1 = f32x4{bflo(tb[2]), bfhi(tb[2]), bflo(tb[3]), bfhi(tb[3])};
        } else {
          const float* tp = E.sub + orow * E.ld_sub + m;
          t0 = *reinterpret_cast<const f32x4*>(tp); t1 = *reinterpret_cast<const f32x4*>(tp + 4);
        }
        float d[8] = {v[0] - t0[0], v[1] - t0[1], v[2] - t0[2], v[3] - t0[3],
                      v[4] - t1[0], v[5] - t1[1], v[6] - t1[2], v[7] - t1[3]};
#pragma unroll
        for (int q = 0; q < 8; ++q) { s0[q] += d[q]; s1[q] += d[q] * d[q]; d[q] *= E.gscale; }
        *reinterpret_cast<u32x4*>(E.gout + orow * E.ld_gout + m) =
            u32x4{pack2bf(d[0], d[1]), pack2bf(d[2], d[3]), pack2bf(d[4], d[5]), pack2bf(d[6], d[7])};
      }
      bf16_t* op = outb + orow * E.ldc + m;
      if (E.accumulate) {
        const u32x4 o = *reinterpret_cast<const u32x4*>(op);
        v[0] += bflo(o[0]); v[1] += bfhi(o[0]); v[2] += bflo(o[1]); v[3] += bfhi(o[1]);
        v[4] += bflo(o[2]); v[5] += bfhi(o[2]); v[6] += bflo(o[3]); v[7] += bfhi(o[3]);
      }
      if (E.mask && E.mask_bits) {
        const unsigned b = reinterpret_cast<const unsigned char*>(E.mask)[orow * E.ld_mask + (m >> 3)];
        const float sl = E.mask_slope;
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] *= ((b >> q) & 1u) ? 1.f : sl;
      } else if (E.mask) {
        const u32x4 mk = *reinterpret_cast<const u32x4*>(E.mask + orow * E.ld_mask + m);
        const float sl = E.mask_slope;
        v[0] *= bflo(mk[0]) > 0.f ? 1.f : sl; v[1] *= bfhi(mk[0]) > 0.f ? 1.f : sl;
        v[2] *= bflo(mk[1]) > 0.f ? 1.f : sl; v[3] *= bfhi(mk[1]) > 0.f ? 1.f : sl;
        v[4] *= bflo(mk[2]) > 0.f ? 1.f : sl; v[5] *= bfhi(mk[2]) > 0.f ? 1.f : sl;
        v[6] *= bflo(mk[3]) > 0.f ? 1.f : sl; v[7] *= bfhi(mk[3]) > 0.f ? 1.f : sl;
      }
      if (E.stats && E.stats_post) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { s0[q] += v[q]; s1[q] += v[q] * v[q]; }
      }
      const u32x4 packed = u32x4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
      if constexpr (FOLD) *reinterpret_cast<u32x4*>(dt + n * FOLD_DT_PITCH + c * 16) = packed;
      else *reinterpret_cast<u32x4*>(op) = packed;
    }
    if (CARRY != 0 && E.stats) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (CARRY == 1) { carry[q] += s0[q]; carry[8 + q] += s1[q]; }
        else { s0[q] += carry[q]; s1[q] += carry[8 + q]; }
      }
    }
    if (CARRY != 1 && E.stats) {
      // lanes with equal lane % CPR hold the same chunk column: butterfly over the others, then one row per wave
      static_assert(CPR == 4 || CPR == 8 || CPR == 16, "chunks per pixel row");
#pragma unroll
      for (int q = 0; q < 8; ++q) { s0[q] = class_sum<CPR>(s0[q]); s1[q] = class_sum<CPR>(s1[q]); }
      if (lane < CPR) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          red[(wave * 2 + 0) * HM + lane * 8 + q] = s0[q];
          red[(wave * 2 + 1) * HM + lane * 8 + q] = s1[q];
        }
      }
      __syncthreads();
      if (tid < 2 * HM) {
        const int k = tid / HM, ch = tid - k * HM, mm = m0 + h * HM + ch;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[(w * 2 + k) * HM + ch];
        if (mm < P.M) {
          long long ridx = (long long)by * ntiles_n + tn;
          if (E.stats_fold || E.stats_img_tiles) {
            const long long row = E.stats_img_tiles ? tn / E.stats_img_tiles : ridx % E.stats_fold;
            __hip_atomic_fetch_add(E.stats + row * 2 * P.M + (long long)k * P.M + mm, t,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            E.stats[ridx * 2 * P.M + (long long)k * P.M + mm] = t;
            for (int e = 0; e < E.stats_zero - 1; ++e)
              E.stats[((long long)E.stats_rows + ridx * (E.stats_zero - 1) + e) * 2 * P.M + (long long)k * P.M + mm] = 0.f;
          }
        }
      }
    }
  }
}

// (bx, by, bz): the block's position in the problem's own grid: tile, tap or phase, k split or batch
template <int BM, int BN, int AKIND, int BKIND>
__device__ __forceinline__ void igemm_body(const GemmParams& P, int bx, int by, int bz, int ntiles_n,
                                           unsigned nblocks) {
  using GA = TileGeom<BM>;
  using GB = TileGeom<BN>;
  constexpr int TM = BM / 32, TN = BN / 32;       // 16x16 tiles per wave (2x2 waves)
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * (GA::ELEMS + GB::ELEMS)];
  constexpr int BUF = GA::ELEMS + GB::ELEMS;        // one buffer: A image then B image

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tm = bx % P.tiles_m, tn = bx / P.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;

  // block-uniform schedule
  constexpr bool CONV = (BKIND == OP_GATHER_K);
  constexpr bool WGRAD = (AKIND == OP_GATHER_TR);
  int pa = 0, pb = 0, r0 = 0, s0 = 0, tj_n = 1, taps = P.taps;
  long long batch = 0;
  int kbeg = 0, kend = P.K;
  if constexpr (CONV) {
    if (P.B.transposed) {                 // by = output phase
      pa = by / P.B.sw; pb = by % P.B.sw;
      r0 = (pa + P.B.ph) % P.B.sh; s0 = (pb + P.B.pw) % P.B.sw;
      int nr = (P.B.kh - r0 + P.B.sh - 1) / P.B.sh, ns = (P.B.kw - s0 + P.B.sw - 1) / P.B.sw;
      tj_n = ns; taps = nr * ns;
    } else {
      tj_n = P.B.kw;
    }
  } else if constexpr (WGRAD) {           // by = tap, bz = k split
    r0 = by / P.A.kw; s0 = by % P.A.kw;
    int per = ((P.K + P.splitk - 1) / P.splitk + BK - 1) / BK * BK;
    kbeg = bz * per;
    kend = min(P.K, kbeg + per);
  } else {
    batch = bz;
  }

  Loader<AKIND, BM> la;
  Loader<BKIND, BN> lb;
  la.init(P.A, m0, batch, kbeg, pa, pb, r0, s0);
  lb.init(P.B, n0, batch, kbeg, pa, pb, r0, s0);
  lb.mark_invalid_rows();

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kt = CONV ? P.K : kend;                      // k extent the loaders mask against
  const int nchunks = CONV ? (P.K + BK - 1) / BK : (max(kend - kbeg, 0) + BK - 1) / BK;
  const int nsteps = (CONV ? taps : 1) * nchunks;

  // step state
  int tap = 0, ti = 0, tj = 0, kc = 0;
  long long koffA = 0;
  auto tap_setup = [&]() {
    if constexpr (CONV) {
      lb.set_tap(P.B, ti, tj);
      int r = P.B.transposed ? r0 + ti * P.B.sh : ti;
      int s = P.B.transposed ? s0 + tj * P.B.sw : tj;
      koffA = (long long)(r * P.B.kw + s) * P.K;
    }
  };
  auto advance = [&]() {
    kc += BK;
    if constexpr (CONV) {
      if (kc >= P.K) {
        kc = 0; ++tap; ++tj;
        if (tj == tj_n) { tj = 0; ++ti; }
        if (tap < taps) tap_setup();
      }
    }
  };

  u32x4 ra[GA::NCH], rb[GB::NCH];
  if (nsteps > 0) {
    tap_setup();
    la.load(P.A, ra, koffA, kc, kt, kbeg + kc, r0, s0);
    lb.load(P.B, rb, 0, kc, kt, kbeg + kc, r0, s0);
    la.store(smem, ra);
    lb.store(smem + GA::ELEMS, rb);
    advance();
  }
  __syncthreads();

  for (int step = 0; step < nsteps; ++step) {
    const bf16_t* a_cur = smem + (step & 1) * BUF;
    const bf16_t* b_cur = a_cur + GA::ELEMS;
    bf16_t* a_nxt = smem + ((step & 1) ^ 1) * BUF;
    const bool more = step + 1 < nsteps;
    if (more) {
      la.load(P.A, ra, koffA, kc, kt, kbeg + kc, r0, s0);
      lb.load(P.B, rb, 0, kc, kt, kbeg + kc, r0, s0);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8_t fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = read_frag<AKIND, BM>(a_cur, wm * (BM / 2) + i * 16, lane, ks);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = read_frag<BKIND, BN>(b_cur, wn * (BN / 2) + j * 16, lane, ks);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      la.store(a_nxt, ra);
      lb.store(a_nxt + GA::ELEMS, rb);
      advance();
    }
    __syncthreads();
  }

  gemm_epilogue<BM, BN, WGRAD>(P, acc, m0, LinearRowMap{P, n0, pa, pb}, tn, batch, ntiles_n, by, bz, nblocks, smem);
}

template <int BM, int BN, int AKIND, int BKIND>
__global__ __launch_bounds__(256) void igemm_kernel(const GemmParams P) {
  igemm_body<BM, BN, AKIND, BKIND>(P, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x / P.tiles_m,
                                   gridDim.x * gridDim.y * gridDim.z);
}

// Job-table form: many independent problems of one tile shape in one launch.  map[b] = (job, block of that job);
// a job's blocks are numbered x fastest, then y, then z, with its own grid in GemmParams::grid.
template <int BM, int BN, int AKIND, int BKIND>
__global__ __launch_bounds__(256) void igemm_jobs_kernel(const GemmParams* __restrict__ jobs,
                                                         const int* __restrict__ map) {
  const GemmParams& P = jobs[map[2 * blockIdx.x]];
  const int lb = map[2 * blockIdx.x + 1];
  const int nx = P.grid[0], ny = P.grid[1];
  const int bx = lb % nx, r = lb / nx;
  igemm_body<BM, BN, AKIND, BKIND>(P, bx, r % ny, r / ny, nx / P.tiles_m, nx * ny * P.grid[2]);
}

// ---------------------------------------------------------------------------
// Convolution forward / data gradient, second form: both operands go HBM -> LDS by LDS-DMA
// (global_load_lds, 16 bytes per lane) into a ring of NST stage buffers, BK = 64 per stage, with the
// loads of NST-1 stages in flight across the per-step barrier (counted vmcnt, raw s_barrier).
//
// K is the flattened (tap, channel) axis in 16-byte chunks, so a stage may straddle taps and layers
// with few input channels do not pad every tap up to the stage depth.  A lane owns one chunk column
// (kq) and NLA + NLB rows; padding pixels, rows past the edge of the problem and chunks past the end
// of K use a buffer offset beyond the descriptor's range, for which the DMA writes zeros: no branches.
//
// LDS image of an operand stage: [rows][8 chunks] with 128-byte rows, chunk q of row r stored in slot
// q ^ (r & 7).  The DMA destination is lane-linear (wave base + 16 * lane), so the swizzle is applied
// to the SOURCE chunk a lane fetches and again when the fragments are read (ds_read_b128,
// conflict-free per 16-lane group).
// ---------------------------------------------------------------------------
#define GLDS_OOB 0xFFFFFFF0u     // a byte offset no operand reaches (host-checked): the DMA then writes zeros

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16,
                                           (int)byte_off, 0, 0, 0);
}

// NT threads: 256 (2 x 2 waves) or 512 (2 x 4 waves: twice the pixels per block for the same weight traffic).
// (bx, by) of (nbx, nby): the block's place in the problem's own grid (the launch grid, or a job's share of one)
// KS = 2 (layers whose grid leaves one block per CU and whose K loop is long): the block has two groups of NT threads,
// each with its own stage ring, that take alternate stages of K - two waves per SIMD hide each other's LDS and MFMA
// latency, which is what bounds a block alone on its CU - and the second group hands its accumulators to the first
// through LDS before the epilogue.
template <int BM, int BN, int NST, int NT, int KS = 1>
__device__ __forceinline__ void conv_glds_body(const GemmParams& P, int bx, int by, int nbx, int nby, char* smem) {
  constexpr int WN = NT / 128;                        // waves along the pixel axis
  constexpr int RPP = NT / 8;                         // rows per DMA pass (8 lanes = one 128-byte row)
  constexpr int TM = BM / 32, TN = BN / (16 * WN);
  constexpr int NLA = BM / RPP, NLB = BN / RPP;       // 16-byte DMA loads per lane per stage
  constexpr int STAGE = (BM + BN) * 128;              // bytes per stage: A image then B image

  const int kg = KS == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x / NT);      // K group of this wave
  const int tid = (int)threadIdx.x - kg * NT, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  char* const smem0 = smem;
  smem += kg * (NST * STAGE);

  // XCD-aware order: consecutive hardware block ids go round-robin over the 8 XCDs; give each XCD a
  // contiguous run of logical tiles so that the blocks sharing a pixel tile hit one L2.
  int bid = bx;
  {
    const int nwg = nbx, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid % P.tiles_m, tn = bid / P.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;

  const GemmOperand& X = P.B;
  int pa = 0, pb = 0, r0 = 0, s0 = 0, ti_n = X.kh, tj_n = X.kw;
  const int dir = X.transposed ? -1 : 1;
  if (X.transposed) {                     // by = output phase; only its non-zero taps
    pa = by / X.sw; pb = by % X.sw;
    r0 = (pa + X.ph) % X.sh; s0 = (pb + X.pw) % X.sw;
    ti_n = (X.kh - r0 + X.sh - 1) / X.sh; tj_n = (X.kw - s0 + X.sw - 1) / X.sw;
  }
  const int rstep = X.transposed ? X.sh : 1, sstep = X.transposed ? X.sw : 1;
  const int CPT = P.K >> 3;                             // chunks per tap
  const int nsteps = (((ti_n * tj_n * CPT + 7) >> 3) + KS - 1) / KS;      // stages per K group (a zero stage may end the last)

  // this lane's rows and chunk column
  const int rowp = tid >> 3;
  const int kq = (tid & 7) ^ (rowp & 7);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A.p, 0, P.A_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)P.B.p, 0, P.B_bytes, 0x00020000);
  unsigned wrow[NLA];                                   // byte offset of the row in the packed weights
#pragma unroll
  for (int i = 0; i < NLA; ++i) {
    int m = m0 + i * RPP + rowp;
    wrow[i] = m < P.M ? (unsigned)m * (unsigned)P.A.ld * 2u : GLDS_OOB;      // stays out of range after + wk: wk < 2^28
  }
  int gy[NLB], gx[NLB], pix[NLB];
  const int hw = X.Ho * X.Wo;
  const float inv_hw = 1.0f / (float)hw, inv_wo = 1.0f / (float)X.Wo;
#pragma unroll
  for (int i = 0; i < NLB; ++i) {
    int n = n0 + i * RPP + rowp;
    int rem, ox;
    int img = fast_div(n, hw, inv_hw, rem);
    int oy = fast_div(rem, X.Wo, inv_wo, ox);
    if (X.transposed) { gy[i] = oy + (pa + X.ph - r0) / X.sh; gx[i] = ox + (pb + X.pw - s0) / X.sw; }
    else              { gy[i] = oy * X.sh - X.ph;             gx[i] = ox * X.sw - X.pw; }
    pix[i] = (img * X.H + gy[i]) * X.W + gx[i];
    if (n >= P.N) gy[i] = -(1 << 24);                   // never inside the image
  }
  int c, ti, tj;
  { int k0 = kq + 8 * kg; int tap = k0 / CPT; c = k0 - tap * CPT; ti = tap / tj_n; tj = tap - ti * tj_n; }

  char* const lds_wave = smem + wave * 1024;            // + i * NT * 16 per row pass, + BM * 128 for B
  const unsigned ldx2 = (unsigned)X.ld * 2u;
  // Offsets are 16-byte aligned, so OR-ing GLDS_OOB into one leaves it out of range whatever its value: validity
  // is carried as a mask (0 or GLDS_OOB) and never as a branch around a load, which keeps the number of DMA
  // instructions per stage exact for the counted waits below.
  auto issue = [&](int buf) {
    char* dst = lds_wave + buf * STAGE;
    unsigned inval = ti < ti_n ? 0u : GLDS_OOB;
    asm volatile("" : "+v"(inval));
    const unsigned wk = (unsigned)(((r0 + ti * rstep) * X.kw + (s0 + tj * sstep)) * P.K + c * 8) * 2u;
#pragma unroll
    for (int i = 0; i < NLA; ++i) glds16(rsA, (wrow[i] + wk) | inval, dst + i * (NT * 16));
    const int dpx = dir * (ti * X.W + tj);
    const unsigned cb = (unsigned)c * 16u;
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      int hi = gy[i] + dir * ti, wi = gx[i] + dir * tj;
      unsigned bad = ((unsigned)hi < (unsigned)X.H && (unsigned)wi < (unsigned)X.W) ? 0u : GLDS_OOB;
      glds16(rsB, ((unsigned)(pix[i] + dpx) * ldx2 + cb) | bad | inval, dst + BM * 128 + i * (NT * 16));
    }
    c += 8 * KS;
    while (c >= CPT) {
      c -= CPT;
      if (++tj == tj_n) { tj = 0; ++ti; }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = (lane & 15) * 128;
  const int sw0 = (((lane >> 4)) ^ (lane & 7)) << 4, sw1 = ((4 + (lane >> 4)) ^ (lane & 7)) << 4;
  bf16x8_t fa[2][TM], fb[2][TN];
  auto load_frags = [&](int buf, int ks) {
    const char* sa = smem + buf * STAGE + wm * (BM / 2) * 128 + frow;
    const char* sb = smem + buf * STAGE + BM * 128 + wn * (BN / WN) * 128 + frow;
    const int sw = ks ? sw1 : sw0;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8_t*>(sa + i * 2048 + sw);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8_t*>(sb + j * 2048 + sw);
  };
  auto mfmas = [&](int ks) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
  };

  // prologue: NST-1 stages in flight (stages past the end of K fetch zeros: uniform counts)
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) issue(s);
  int rd = 0, wr = NST - 1;
  for (int step = 0; step < nsteps; ++step) {
    // the DMA of stage `step` has landed for this wave's loads once at most NST-2 later stages are outstanding;
    // the barrier extends that to every wave's loads and retires all reads of the buffer refilled next
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (NLA + NLB)) : "memory");
    __builtin_amdgcn_s_barrier();
    load_frags(rd, 0);          // the LDS reads of the first k-slice fly while the next stage's addresses are built
    issue(wr);
    load_frags(rd, 1);
    mfmas(0);
    mfmas(1);
    rd = rd + 1 == NST ? 0 : rd + 1;
    wr = wr + 1 == NST ? 0 : wr + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no DMA may outlive the block or race the epilogue
  __builtin_amdgcn_s_barrier();
  if constexpr (KS == 2) {      // the second group's sums go to the first through the second group's own ring, then it leaves
    static_assert(TM * TN * NT * 16 <= NST * STAGE, "accumulator hand-over area");
    f32x4* dump = reinterpret_cast<f32x4*>(smem0 + NST * STAGE) + tid;
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) dump[(i * TN + j) * NT] = acc[i][j];
    }
    __syncthreads();
    if (kg == 1) return;        // (a barrier counts the waves still running: the epilogue's see the first group only)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const f32x4 o = dump[(i * TN + j) * NT];
        acc[i][j][0] += o[0]; acc[i][j][1] += o[1]; acc[i][j][2] += o[2]; acc[i][j][3] += o[3];
      }
  }
  if (P.E.staged)
    gemm_epilogue_staged<BM, BN, WN, NST * STAGE>(P, acc, m0, LinearRowMap{P, n0, pa, pb}, tn, nbx / P.tiles_m, by,
                                                  smem);
  else
    gemm_epilogue<BM, BN, false, WN>(P, acc, m0, LinearRowMap{P, n0, pa, pb}, tn, 0, nbx / P.tiles_m, by, 0,
                                     nbx * nby, smem);
}

template <int BM, int BN, int NST, int NT = 256, int KS = 1>
__global__ __launch_bounds__(NT * KS) void conv_glds_kernel(const GemmParams P) {
  __shared__ __attribute__((aligned(1024))) char smem[KS * NST * (BM + BN) * 128];
  conv_glds_body<BM, BN, NST, NT, KS>(P, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, smem);
}

// ---------------------------------------------------------------------------
// The same convolution with an input transform (GemmParams::xf_*): the pixel operand is the RAW output of a
// BasicConv2d layer's convolution and is normalised (train-mode BatchNorm) and activated on its way IN:
// global -> registers -> scale, shift, ReLU -> LDS.  The weights keep the LDS-DMA path.  Compared with transforming
// the operand in place in LDS (measured: +70 % on the consumer, more than the apply pass it saves) this adds no LDS
// traffic: the register path's one ds_write_b128 per chunk replaces the DMA's write.
//
// Rings: three weight stages (small: BM x 128 bytes each), two pixel stages in LDS and two register sets of pixel
// chunks.  In step s the loads of stage s + 2 are issued (weights by DMA, pixels into the register set that stage s
// vacated), the products of stage s run, and then the pixels of stage s + 1 - loaded one step earlier - are
// transformed and written to the LDS stage that step s - 1 read; the barrier at the top of step s + 1 publishes
// them.  Loads complete in order, so both waits are counted: everything but the newest issue group.
// Padding pixels, rows past the problem and the K tail load zeros (out-of-range buffer offsets) and must STAY zero
// (zero padding pads the activated tensor): one validity bit per chunk.  K <= XF_MAX_K input channels.
// ---------------------------------------------------------------------------
#define XF_MAX_K 1024
template <int BM, int BN, int NT>
__global__ __launch_bounds__(NT) void conv_glds_xf_kernel(const GemmParams P) {
  constexpr int WN = NT / 128;
  constexpr int RPP = NT / 8;
  constexpr int TM = BM / 32, TN = BN / (16 * WN);
  constexpr int NLA = BM / RPP, NLB = BN / RPP;
  constexpr int ASTAGE = BM * 128, BSTAGE = BN * 128;
  constexpr int RINGS = 3 * ASTAGE + 2 * BSTAGE;
  __shared__ __attribute__((aligned(1024))) char smem[RINGS + 2 * XF_MAX_K * 4];
  char* const smemB = smem + 3 * ASTAGE;
  float* const xf_ab = reinterpret_cast<float*>(smem + RINGS);      // [2][K]: scale, shift per input channel

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int nbx = gridDim.x, by = blockIdx.y;
  int bid = blockIdx.x;
  {      // XCD-aware order, as in conv_glds_body
    const int nwg = nbx, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid % P.tiles_m, tn = bid / P.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;

  const GemmOperand& X = P.B;
  int pa = 0, pb = 0, r0 = 0, s0 = 0, ti_n = X.kh, tj_n = X.kw;
  const int dir = X.transposed ? -1 : 1;
  if (X.transposed) {
    pa = by / X.sw; pb = by % X.sw;
    r0 = (pa + X.ph) % X.sh; s0 = (pb + X.pw) % X.sw;
    ti_n = (X.kh - r0 + X.sh - 1) / X.sh; tj_n = (X.kw - s0 + X.sw - 1) / X.sw;
  }
  const int rstep = X.transposed ? X.sh : 1, sstep = X.transposed ? X.sw : 1;
  const int CPT = P.K >> 3;
  const int nsteps = (ti_n * tj_n * CPT + 7) >> 3;

  for (int ch = tid; ch < P.K; ch += NT) {
    const float a = P.xf_gamma[ch] * P.xf_save[P.K + ch];
    xf_ab[ch] = a;
    xf_ab[P.K + ch] = P.xf_beta[ch] - P.xf_save[ch] * a;
  }

  const int rowp = tid >> 3;
  const int kq = (tid & 7) ^ (rowp & 7);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A.p, 0, P.A_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)P.B.p, 0, P.B_bytes, 0x00020000);
  unsigned wrow[NLA];
#pragma unroll
  for (int i = 0; i < NLA; ++i) {
    int m = m0 + i * RPP + rowp;
    wrow[i] = m < P.M ? (unsigned)m * (unsigned)P.A.ld * 2u : GLDS_OOB;
  }
  int gy[NLB], gx[NLB], pix[NLB];
  const int hw = X.Ho * X.Wo;
  const float inv_hw = 1.0f / (float)hw, inv_wo = 1.0f / (float)X.Wo;
#pragma unroll
  for (int i = 0; i < NLB; ++i) {
    int n = n0 + i * RPP + rowp;
    int rem, ox;
    int img = fast_div(n, hw, inv_hw, rem);
    int oy = fast_div(rem, X.Wo, inv_wo, ox);
    if (X.transposed) { gy[i] = oy + (pa + X.ph - r0) / X.sh; gx[i] = ox + (pb + X.pw - s0) / X.sw; }
    else              { gy[i] = oy * X.sh - X.ph;             gx[i] = ox * X.sw - X.pw; }
    pix[i] = (img * X.H + gy[i]) * X.W + gx[i];
    if (n >= P.N) gy[i] = -(1 << 24);
  }
  int c, ti, tj;
  { int tap = kq / CPT; c = kq - tap * CPT; ti = tap / tj_n; tj = tap - ti * tj_n; }

  char* const ldsA_wave = smem + wave * 1024;
  char* const ldsB_own = smemB + wave * 1024 + lane * 16;      // where the DMA would have put this lane's chunk
  const unsigned ldx2 = (unsigned)X.ld * 2u;

  u32x4 rb[2][NLB];                 // pixel chunks in flight: set (stage & 1)
  int rc[2] = {0, 0};               // first input channel of the set's chunks
  unsigned rok[2] = {0u, 0u};       // bit i: chunk i is a real pixel (not padding, edge or K tail)

  // issue the loads of the next K stage: weights by DMA into ring slot abuf, pixels into register set SET
  auto issue = [&](int abuf, auto SETC) {
    constexpr int SET = decltype(SETC)::value;
    char* dst = ldsA_wave + abuf * ASTAGE;
    unsigned inval = ti < ti_n ? 0u : GLDS_OOB;
    asm volatile("" : "+v"(inval));
    const unsigned wk = (unsigned)(((r0 + ti * rstep) * X.kw + (s0 + tj * sstep)) * P.K + c * 8) * 2u;
#pragma unroll
    for (int i = 0; i < NLA; ++i) glds16(rsA, (wrow[i] + wk) | inval, dst + i * (NT * 16));
    const int dpx = dir * (ti * X.W + tj);
    const unsigned cb = (unsigned)c * 16u;
    rc[SET] = c * 8;
    unsigned ok = 0u;
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      int hi = gy[i] + dir * ti, wi = gx[i] + dir * tj;
      unsigned bad = ((unsigned)hi < (unsigned)X.H && (unsigned)wi < (unsigned)X.W) ? 0u : GLDS_OOB;
      rb[SET][i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(((unsigned)(pix[i] + dpx) * ldx2 + cb) | bad | inval),
                                                         0, 0);
      ok |= ((bad | inval) ? 0u : 1u) << i;
    }
    rok[SET] = ok;
    c += 8;
    while (c >= CPT) {
      c -= CPT;
      if (++tj == tj_n) { tj = 0; ++ti; }
    }
  };

  // normalise + activate register set SET and store it as pixel stage SET of the LDS ring
  auto xf_store = [&](auto SETC) {
    constexpr int SET = decltype(SETC)::value;
    float sc[8], sh[8];
    const int c0 = rc[SET] < P.K ? rc[SET] : 0;       // (K tail of a stage past the end: chunks are invalid anyway)
#pragma unroll
    for (int q = 0; q < 8; q += 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(xf_ab + c0 + q);
      const f32x4 b = *reinterpret_cast<const f32x4*>(xf_ab + P.K + c0 + q);
      sc[q] = a[0]; sc[q + 1] = a[1]; sc[q + 2] = a[2]; sc[q + 3] = a[3];
      sh[q] = b[0]; sh[q + 1] = b[1]; sh[q + 2] = b[2]; sh[q + 3] = b[3];
    }
    char* const own = ldsB_own + SET * BSTAGE;
    const bool relu = P.xf_act == 1;
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      u32x4 v = rb[SET][i];
      float f[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) { f[2 * q] = bflo(v[q]) * sc[2 * q] + sh[2 * q]; f[2 * q + 1] = bfhi(v[q]) * sc[2 * q + 1] + sh[2 * q + 1]; }
      if (relu) {
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = fmaxf(f[q], 0.f);
      }
      u32x4 o = u32x4{pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7])};
      if (!((rok[SET] >> i) & 1u)) o = u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(own + i * (NT * 16)) = o;
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = (lane & 15) * 128;
  const int sw0 = (((lane >> 4)) ^ (lane & 7)) << 4, sw1 = ((4 + (lane >> 4)) ^ (lane & 7)) << 4;
  bf16x8_t fa[2][TM], fb[2][TN];
  auto load_frags = [&](int abuf, int bbuf, int ks) {
    const char* sa = smem + abuf * ASTAGE + wm * (BM / 2) * 128 + frow;
    const char* sb = smemB + bbuf * BSTAGE + wn * (BN / WN) * 128 + frow;
    const int sw = ks ? sw1 : sw0;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8_t*>(sa + i * 2048 + sw);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8_t*>(sb + j * 2048 + sw);
  };
  auto mfmas = [&](int ks) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  // prologue: stages 0 and 1 in flight; stage 0's pixels transformed into LDS (the table first: it was written above)
  issue(0, S0{});
  issue(1, S1{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // the scale / shift table is complete
  __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLA + NLB) : "memory");
  xf_store(S0{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  int ard = 0, awr = 2;             // weight ring: slot read this step, slot filled this step
  auto step = [&](auto CURC) {
    constexpr int CUR = decltype(CURC)::value;          // pixel stage / register set of this step: step & 1
    using NXT = std::integral_constant<int, CUR ^ 1>;
    // stage s: weights landed once only the newest issue group is outstanding; the barrier extends that to every
    // wave's DMA and publishes the pixel stage written at the end of the previous step
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLA + NLB) : "memory");
    __builtin_amdgcn_s_barrier();
    load_frags(ard, CUR, 0);
    issue(awr, CURC);                                   // stage s + 2: register set CUR is free (stored last step)
    load_frags(ard, CUR, 1);
    mfmas(0);
    mfmas(1);
    ard = ard == 2 ? 0 : ard + 1;
    awr = awr == 2 ? 0 : awr + 1;
    // stage s + 1's pixels (issued one step ago) -> LDS stage CUR ^ 1, which step s - 1 read (all waves are past
    // this step's barrier, so those reads have retired)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLA + NLB) : "memory");
    xf_store(NXT{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  for (int s = 0; s < nsteps; s += 2) {
    step(S0{});
    if (s + 1 < nsteps) step(S1{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (P.E.staged)
    gemm_epilogue_staged<BM, BN, WN, RINGS>(P, acc, m0, LinearRowMap{P, n0, pa, pb}, tn, nbx / P.tiles_m, by, smem);
  else
    gemm_epilogue<BM, BN, false, WN>(P, acc, m0, LinearRowMap{P, n0, pa, pb}, tn, 0, nbx / P.tiles_m, by, 0,
                                     nbx * gridDim.y, smem);
}

// Job-table form: several convolutions that do not depend on each other (the same-depth layers of an Inception
// block's branches) as ONE grid.  map[2b], map[2b+1] = (job, block of that job's own grid[0] x grid[1]).
template <int BM, int BN, int NST, int NT = 256>
__global__ __launch_bounds__(NT) void conv_glds_jobs_kernel(const GemmParams* __restrict__ jobs,
                                                            const int* __restrict__ map) {
  __shared__ __attribute__((aligned(1024))) char smem[NST * (BM + BN) * 128];
  const GemmParams& P = jobs[map[2 * blockIdx.x]];
  const int lb = map[2 * blockIdx.x + 1];
  const int nx = P.grid[0];
  conv_glds_body<BM, BN, NST, NT>(P, lb % nx, lb / nx, nx, P.grid[1], smem);
}

// ---------------------------------------------------------------------------
// 3x3, stride 1, pad 1 convolution (forward, or the data gradient of one: flip = 1), third form.  The kernel
// above fetches the input pixels of a tile once per tap: nine times the bytes through L2, which is what bounds
// layers with 64-256 input channels on large maps.  Here a block owns 256 output pixels as TS x TS tiles at one
// position of 256 / TS^2 consecutive images (one 16x16 tile, or 8x8 tiles of four images) and loads the input
// tiles - the outputs' halo included, (TS+2)^2 pixels each - into LDS once per 64-channel slab; the nine taps
// read their operand fragments from that image at shifted rows.  Only the weights (BM x 64 per tap) stream
// through a three-stage ring.  Per 64-channel slab a 128 x 16x16 block moves 9 * 128 * 128 + 324 * 128 bytes for
// 2 * 128 * 256 * 576 flops: 3.1x the arithmetic intensity of the 128x128 form.
//
// 512 threads = 2 x 4 waves, each wave BM/2 rows x 64 pixels (4 tile rows, or one 8x8 tile).  All loads are
// LDS-DMA with out-of-range offsets for padding (zeros).  The taps of a slab are unrolled, so every wait is a
// counted s_waitcnt with a compile-time count: the weights of step k+2 and one part of the next slab's input
// tile are issued in step k and stay in flight across two barriers.
// NBB = 1: layers with exactly 64 input channels have one slab and need no second input-tile buffer; with 64
// output channels per block two blocks then fit a CU (73 KB of LDS, 128 registers), so that one block's prologue
// and epilogue - most of its life when K is nine steps - overlap the other's main loop.
// Requires Cin % 64 == 0, H % TS == 0, W % TS == 0 (host-checked).
// ---------------------------------------------------------------------------
template <int BM, int TS, int NBB = 2>
__global__ __launch_bounds__(512, NBB == 1 ? 4 : 2) void conv_halo_kernel(const GemmParams P) {
  constexpr int BN = 256, WN = 4;
  constexpr int IMG = BN / (TS * TS);                      // images per block: 1 or 4
  constexpr int TWP = TS + 2, HRI = TWP * TWP, HR = IMG * HRI;   // input pixels per tile, per block: 324 / 400
  constexpr int NLB = (HR * 8 + 511) / 512;                // 16-byte DMA loads per lane per slab: 6 / 7
  constexpr int BBUF = NLB * 512 * 16;
  static_assert(TS == 16 || TS == 8, "tile side");
  constexpr int NLA = BM * 8 / 512;                        // per weight stage: 2 (BM 128) or 1 (BM 64)
  constexpr int ABUF = BM * 128;
  constexpr int TM = BM / 32, TN = 4;
  constexpr unsigned A_OOB = 0xFFFF0000u, B_OOB = 0xFFFFF000u;   // + the largest k offset: still out of range, no wrap
  constexpr int PARTS = NBB == 2 ? NLB : 0;                // input-tile loads issued inside the loop per slab
  __shared__ __attribute__((aligned(1024))) char smem[3 * ABUF + NBB * BBUF];
  char* const sA = smem;
  char* const sB = smem + 3 * ABUF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const GemmOperand& X = P.B;
  const int tm = bid % P.tiles_m, tn = bid / P.tiles_m;
  const int m0 = tm * BM;
  const int tX = X.W / TS, tY = X.H / TS;
  const int img0 = (tn / (tX * tY)) * IMG, trem = tn % (tX * tY);
  const int ty0 = (trem / tX) * TS, tx0 = (trem % tX) * TS;
  const int flip = X.transposed;
  const int nsl = P.K >> 6;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A.p, 0, P.A_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)P.B.p, 0, P.B_bytes, 0x00020000);
  unsigned aoff[NLA], boff[NLB];
#pragma unroll
  for (int i = 0; i < NLA; ++i) {
    const int ci = i * 512 + tid, m = ci >> 3, q = (ci & 7) ^ (m & 7);
    aoff[i] = m0 + m < P.M ? (unsigned)(m0 + m) * (unsigned)P.A.ld * 2u + (unsigned)q * 16u : A_OOB;
  }
#pragma unroll
  for (int i = 0; i < NLB; ++i) {
    const int ci = i * 512 + tid, hr = ci >> 3;
    const int il = hr / HRI, hp = hr - il * HRI;
    const int hy = hp / TWP, hx = hp - hy * TWP;
    // chunk swizzle by the row number for 18-pixel tile rows, by the column for 10-pixel ones (a fragment read
    // takes 8 + 8 pixels of two tile rows there: rows 10 apart have the same parity and collide two-way)
    const int q = (ci & 7) ^ ((TS == 8 ? hx : hr) & 7);
    const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
    const bool ok = hr < HR && img0 + il < X.nimg && (unsigned)iy < (unsigned)X.H && (unsigned)ix < (unsigned)X.W;
    boff[i] = ok ? (unsigned)(((img0 + il) * X.H + iy) * X.W + ix) * (unsigned)X.ld * 2u + (unsigned)q * 16u : B_OOB;
  }
  char* const dA = sA + wave * 1024;
  char* const dB = sB + wave * 1024;
  // weights of step (slab, tap) -> ring stage; steps past the end fetch zeros into a stage nobody reads any more
  auto issueA = [&](int slab, int tap, int stage) {
    unsigned dead = slab < nsl ? 0u : A_OOB;
    asm volatile("" : "+v"(dead));
    const unsigned wk = (unsigned)(tap * P.K + slab * 64) * 2u;
#pragma unroll
    for (int i = 0; i < NLA; ++i) glds16(rsA, (aoff[i] + wk) | dead, dA + stage * ABUF + i * 8192);
  };
  auto issueB = [&](int slab, int part) {
    unsigned dead = slab < nsl ? 0u : B_OOB;
    asm volatile("" : "+v"(dead));
    glds16(rsB, (boff[part] + (unsigned)slab * 128u) | dead, dB + (slab & (NBB - 1)) * BBUF + part * 8192);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = (lane & 15) * 128;
  const int kq0 = lane >> 4, kq1 = 4 + (lane >> 4);
  const int sw0 = (kq0 ^ (lane & 7)) << 4, sw1 = (kq1 ^ (lane & 7)) << 4;
  int hr0[TN];                       // LDS row of this lane's pixel (n = wn*64 + j*16 + lane%16) before the tap shift
  const int hx0 = lane & 7;          // TS == 8: its column in the tile (16 consecutive pixels = two tile rows)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nl = wn * 64 + j * 16 + (lane & 15);
    const int il = nl / (TS * TS), p = nl % (TS * TS);
    hr0[j] = il * HRI + (p / TS) * TWP + (p % TS);
  }
  bf16x8_t fa[2][TM], fb[2][TN];
  auto load_frags = [&](int stage, int bbuf, int toff, int tcol, int ks) {
    const char* sa = sA + stage * ABUF + wm * (BM / 2) * 128 + frow;
    const char* sb = sB + bbuf * BBUF;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8_t*>(sa + i * 2048 + (ks ? sw1 : sw0));
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int hr = hr0[j] + toff;
      const int swz = (TS == 8 ? hx0 + tcol : hr) & 7;
      fb[ks][j] = *reinterpret_cast<const bf16x8_t*>(sb + hr * 128 + (((ks ? kq1 : kq0) ^ swz) << 4));
    }
  };
  auto mfmas = [&](int ks) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
  };

#pragma unroll
  for (int part = 0; part < NLB; ++part) issueB(0, part);
  issueA(0, 0, 0);
  issueA(0, 1, 1);
  for (int slab = 0; slab < nsl; ++slab) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // loads issued after the weights of this step: the input-tile part of step-2 (if any), the weights of
      // step+1, the input-tile part of step-1 (if any); everything older has landed once only those remain
      const int n_after = ((tap + 7) % 9 < PARTS ? 1 : 0) + NLA + ((tap + 8) % 9 < PARTS ? 1 : 0);
      if (n_after == NLA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLA) : "memory");
      else if (n_after == NLA + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLA + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLA + 2) : "memory");
      __builtin_amdgcn_s_barrier();
      const int r = tap / 3, c = tap - r * 3;
      const int toff = flip ? (2 - r) * TWP + (2 - c) : r * TWP + c;
      const int tcol = flip ? 2 - c : c;
      load_frags(tap % 3, slab & (NBB - 1), toff, tcol, 0);
      if (tap < 7) issueA(slab, tap + 2, (tap + 2) % 3);
      else issueA(slab + 1, tap - 7, (tap + 2) % 3);
      if (tap < PARTS) issueB(slab + 1, tap);
      load_frags(tap % 3, slab & (NBB - 1), toff, tcol, 1);
      mfmas(0);
      mfmas(1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  struct TileRowMap {
    long long base; int W, HW, nimg_left;
    __device__ __forceinline__ bool operator()(int nl, long long& orow) const {
      const int il = nl / (TS * TS), p = nl % (TS * TS);
      orow = base + (long long)il * HW + (long long)(p / TS) * W + (p % TS);
      return il < nimg_left;
    }
  };
  const TileRowMap rm{((long long)img0 * X.H + ty0) * X.W + tx0, X.W, X.H * X.W, X.nimg - img0};
  gemm_epilogue_staged<BM, BN, WN, 3 * ABUF + NBB * BBUF>(P, acc, m0, rm, tn, gridDim.x / P.tiles_m, 0, smem);
}

// ---------------------------------------------------------------------------
// 3x3, stride 2, pad 1 convolution FORWARD (src/rinet.py:21-24: the discriminator's strided layers; :56 conv2),
// halo-tile form.  The general kernel fetches the input pixels once per tap and is bound by that L2 -> LDS traffic
// (51 flop per byte at its 64 x 256 tile).  A stride-2 window needs (2 TS + 1)^2 input pixels for TS^2 outputs - 1089
// for a 16 x 16 tile, 139 KB per 64-channel slab - so the tile is held in 32-channel HALF slabs (64 bytes per pixel,
// 70 KB): a block owns 16 x 16 output pixels of one image and BM output channels, loads the 33 x 33 input tile of a
// half slab once and runs its nine taps from it, one k-slice of 32 per tap; only the weights stream (four-stage ring
// of BM x 32).  Per half slab 1089 * 64 + 9 * BM * 64 bytes for 2 * BM * 256 * 288 flops: 132 flop per byte at BM =
// 128.  There is room for ONE tile buffer: the next half slab is fetched when every wave has read the current one
// (one memory round trip per half slab; a layer with 64 input channels pays it once per block).
//
// LDS images have 64-byte rows.  Input tile: pixel (hy, hx) is row hy * 33 + (hx & 1) * 17 + (hx >> 1) - even and odd
// columns apart, so that the sixteen pixels a fragment read takes (one output row: columns 2 ox + c) are sixteen
// CONSECUTIVE rows; chunk q of row r sits in slot q ^ ((r >> 2) & 3), which puts the 16 lanes of a ds_read_b128
// pass on 16 different bank slots (rows 64 bytes apart: four rows per 256-byte window, four slots each).  The same
// swizzle for the weight rows.  All loads are LDS-DMA; padding and rows past Cout use out-of-range offsets (zeros).
// Requires Cin % 32 == 0, Ho % 16 == 0, Wo % 16 == 0, H == 2 Ho, W == 2 Wo (host-checked).
// ---------------------------------------------------------------------------
// TH output rows per tile: 16 (512 threads, 104 KB of LDS at BM = 128: one block per CU) or 8 (256 threads, 68 KB: two
// blocks per CU, so that one block's prologue, refill and epilogue run under the other's main loop).
template <int BM, int TH>
__global__ __launch_bounds__(TH * 32, 2) void conv_halo_s2f_kernel(const GemmParams P) {
  constexpr int TS = 16, BN = TH * TS, NT = 2 * BN, WN = NT / 128;
  constexpr int TWX = 2 * TS + 1, TWY = 2 * TH + 1, HR = TWX * TWY;      // 33 x 33 or 33 x 17 input pixels
  constexpr int NLB = (HR * 4 + NT - 1) / NT;              // DMA loads per lane per half slab
  constexpr int BBUF = NLB * NT * 16;
  constexpr int ABUF = BM * 64, NSA = 4;                   // weight ring: four stages of BM rows x 32 channels
  constexpr int NLA = (BM * 4 + NT - 1) / NT;              // DMA loads per lane per weight stage (lanes past BM * 4: none)
  constexpr int AW = BM * 4 >= NT ? 2 * WN : BM / 16;      // waves that carry weights
  constexpr int TM = BM / 32, TN = 4;
  static_assert((BM == 64 || BM == 128) && (TH == 16 || TH == 8), "tile");
  constexpr unsigned A_OOB = 0xFFFF0000u, B_OOB = 0xFFFFF000u;
  __shared__ __attribute__((aligned(1024))) char smem[NSA * ABUF + BBUF];
  char* const sA = smem;
  char* const sB = smem + NSA * ABUF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const GemmOperand& X = P.B;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A.p, 0, P.A_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)P.B.p, 0, P.B_bytes, 0x00020000);
  const int ntotal = gridDim.x;
  int bid = blockIdx.x;
  {
    const int nwg = ntotal, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid % P.tiles_m, tn = bid / P.tiles_m;
  const int m0 = tm * BM;
  const int tX = X.Wo / TS, tY = X.Ho / TH;
  const int img0 = tn / (tX * tY), trem = tn % (tX * tY);
  const int ty0 = (trem / tX) * TH, tx0 = (trem % tX) * TS;        // output tile origin
  const int nhalf = P.K >> 5;

  unsigned aoff[NLA];
#pragma unroll
  for (int i = 0; i < NLA; ++i) {
    const int ci = i * NT + tid, m = ci >> 2, q = (ci & 3) ^ ((m >> 2) & 3);
    aoff[i] = (ci < BM * 4 && m0 + m < P.M) ? (unsigned)(m0 + m) * (unsigned)P.A.ld * 2u + (unsigned)q * 16u : A_OOB;
  }
  unsigned boff[NLB];
#pragma unroll
  for (int i = 0; i < NLB; ++i) {
    const int ci = i * NT + tid, row = ci >> 2, q = (ci & 3) ^ ((row >> 2) & 3);
    const int hy = row / TWX, rem = row - hy * TWX;
    const int hx = rem < TS + 1 ? 2 * rem : 2 * (rem - (TS + 1)) + 1;
    const int iy = 2 * ty0 - 1 + hy, ix = 2 * tx0 - 1 + hx;
    const bool ok = row < HR && (unsigned)iy < (unsigned)X.H && (unsigned)ix < (unsigned)X.W;
    boff[i] = ok ? (unsigned)((img0 * X.H + iy) * X.W + ix) * (unsigned)X.ld * 2u + (unsigned)q * 16u : B_OOB;
  }
  char* const dA = sA + wave * 1024;
  char* const dB = sB + wave * 1024;
  // weights of global step sg = half * 9 + tap -> ring stage sg % 4; steps past the end fetch zeros nobody reads
  auto issueA = [&](int sg) {
    const int h = sg / 9, tap = sg - h * 9;
    unsigned dead = h < nhalf ? 0u : A_OOB;
    asm volatile("" : "+v"(dead));
    const unsigned wk = (unsigned)(tap * P.K + h * 32) * 2u;
    if (wave < AW) {
#pragma unroll
      for (int i = 0; i < NLA; ++i) glds16(rsA, (aoff[i] + wk) | dead, dA + (sg & (NSA - 1)) * ABUF + i * (NT * 16));
    }
  };
  auto issueB = [&](int h) {
#pragma unroll
    for (int part = 0; part < NLB; ++part) glds16(rsB, boff[part] + (unsigned)h * 64u, dB + part * (NT * 16));
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kq = lane >> 4;                                  // this lane's 16-byte k chunk of the 32-channel slice
  const int fa_off = (lane & 15) * 64 + ((kq ^ ((lane >> 2) & 3)) << 4);      // rows wm * (BM/2) + 16 i + lane % 16
  int hr0[TN];                        // LDS row of this lane's output pixel (oy = 4 wn + j, ox = lane % 16) at tap (0, 0)
#pragma unroll
  for (int j = 0; j < TN; ++j) hr0[j] = 2 * (wn * 4 + j) * TWX + (lane & 15);
  bf16x8_t fa[TM], fb[TN];
  auto load_frags = [&](int stage, int toff) {
    const char* sa = sA + stage * ABUF + wm * (BM / 2) * 64 + fa_off;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sa + i * 1024);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int hr = hr0[j] + toff;
      fb[j] = *reinterpret_cast<const bf16x8_t*>(sB + hr * 64 + ((kq ^ ((hr >> 2) & 3)) << 4));
    }
  };

  issueB(0);
  issueA(0);
  issueA(1);
  issueA(2);
  for (int h = 0; h < nhalf; ++h) {
    if (h > 0) {                         // every wave has read the tile: fetch the next half slab over it
      __builtin_amdgcn_s_barrier();
      issueB(h);
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int sg = h * 9 + tap;
      // this step's weights were issued three steps ago; the two steps issued since may stay in flight.  The first
      // step of a later half slab also waits for the tile, which was issued last.  (Waves that issue no weights
      // have nothing but the tile in flight.)
      if ((tap == 0 && h > 0) || wave >= AW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLA) : "memory");
      __builtin_amdgcn_s_barrier();
      const int r = tap / 3, c = tap - r * 3;
      load_frags(sg & (NSA - 1), r * TWX + (c & 1) * (TS + 1) + (c >> 1));
      issueA(sg + 3);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  struct TileRowMap {
    long long base; int W;
    __device__ __forceinline__ bool operator()(int nl, long long& orow) const {
      orow = base + (long long)(nl / TS) * W + (nl % TS);
      return true;
    }
  };
  const TileRowMap rm{((long long)img0 * X.Ho + ty0) * X.Wo + tx0, X.Wo};
  gemm_epilogue_staged<BM, BN, WN, NSA * ABUF + BBUF>(P, acc, m0, rm, tn, ntotal / P.tiles_m, 0, smem);
}

// ---------------------------------------------------------------------------
// Data gradient of a 3x3, stride 2, pad 1 convolution (= transposed convolution onto a 2H x 2W map), halo-tile
// form.  The general kernel runs each of the four output phases (row parity, column parity) as its own short
// GEMM - 1, 2, 2 and 4 taps - and fetches the input once per tap.  Here a block owns TS x TS input positions
// (of one image, or 8x8 of four) and ALL FOUR phases of the 2TS x 2TS output pixels they produce: the
// (TS+1)^2 input tile is loaded into LDS once per 64-channel slab and the nine (phase, tap) products of a slab
// run as the nine steps of the same pipeline as conv_halo_kernel, each step adding into its phase's accumulators
// (4 x 64 x 256 fp32 per block: 64 output channels per block).  One block thus has a K loop of 9 * Cin/64 steps
// and four epilogues instead of four blocks with loops of 1-4 steps each.
// Requires Cin % 64 == 0, H % TS == 0, W % TS == 0 (host-checked).
// ---------------------------------------------------------------------------
struct S2tStep { int phase, wtap, shift_i, shift_j; };
// phase = 2*(row parity) + column parity; weight tap r*3+s with (parity + 1 - r) even; input shift (parity+1-r)/2
// Ordered by input shift: steps with the same shift reuse the pixel fragments in registers (four fragment loads
// per slab instead of nine).
__device__ constexpr S2tStep S2T_STEPS[9] = {{0, 4, 0, 0}, {1, 5, 0, 0}, {2, 7, 0, 0}, {3, 8, 0, 0}, {1, 3, 0, 1},
                                             {3, 6, 0, 1}, {2, 1, 1, 0}, {3, 2, 1, 0}, {3, 0, 1, 1}};

// FOLD (TS = 16, 64 output channels = the discriminator's first activation, src/rinet.py:20): the masked gradient is
// not stored.  Each phase's 256 x 64 tile goes to LDS in bf16 (the rounding the stored tensor would have had) and is
// multiplied on the matrix cores by the nine taps of the fp32 image under the same pixels - a [256][9] table built
// from a 34 x 34 patch of the image that the block fetched before its main loop, split into bf16 high and low parts
// so that the image keeps 16 bits of mantissa.  The block's [64][9] sums are one row of E.fold_rows: the first layer's
// weight gradient without the 134 MB round trip of its output gradient and without a launch of its own.
template <int TS, bool FOLD = false>
__global__ __launch_bounds__(512, 2) void conv_halo_s2t_kernel(const GemmParams P) {
  constexpr int BM = 64, BN = 256, WN = 4;
  constexpr int IMG = BN / (TS * TS);
  constexpr int TWP = TS + 1, HRI = TWP * TWP, HR = IMG * HRI;   // 289 / 324 input pixels per block
  constexpr int NLB = (HR * 8 + 511) / 512;                // 5 / 6
  constexpr int BBUF = NLB * 512 * 16;
  static_assert(TS == 16 || TS == 8, "tile side");
  constexpr int NLA = BM * 8 / 512;                        // 1
  constexpr int ABUF = BM * 128;
  constexpr int TM = BM / 32, TN = 4;
  constexpr unsigned A_OOB = 0xFFFF0000u, B_OOB = 0xFFFFF000u;
  static_assert(!FOLD || TS == 16, "the fold needs one image per block");
  constexpr int PIPE = 3 * ABUF + 2 * BBUF;
  // FOLD: [staging tile + wave totals | gradient tile | tap table | image patch]; the patch lies behind the pipeline
  constexpr int STG = BN * (BM + 4) * 4 + 8 * 2 * BM * 4;
  constexpr int DT_OFF = STG, XT_OFF = DT_OFF + BN * FOLD_DT_PITCH, PATCH_OFF = XT_OFF + BN * FOLD_XT_PITCH;
  constexpr int PW = 2 * TS + 2;                                      // patch side
  static_assert(!FOLD || PATCH_OFF >= PIPE, "the patch must survive the main loop");
  constexpr int SMEM = FOLD ? PATCH_OFF + PW * PW * 4 : PIPE;
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];
  char* const sA = smem;
  char* const sB = smem + 3 * ABUF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const GemmOperand& X = P.B;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A.p, 0, P.A_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)P.B.p, 0, P.B_bytes, 0x00020000);
  // (A persistent loop over tiles, so that one tile's stores drain under the next tile's loads, was measured
  // 20 % slower than letting the hardware start fresh blocks.)
  const int ntotal = gridDim.x;
  int bid = blockIdx.x;
  {
    const int nwg = ntotal, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid % P.tiles_m, tn = bid / P.tiles_m;
  const int m0 = tm * BM;
  const int tX = X.W / TS, tY = X.H / TS;
  const int img0 = (tn / (tX * tY)) * IMG, trem = tn % (tX * tY);
  const int ty0 = (trem / tX) * TS, tx0 = (trem % tX) * TS;
  const int nsl = P.K >> 6;

  unsigned aoff[NLA], boff[NLB];
#pragma unroll
  for (int i = 0; i < NLA; ++i) {
    const int ci = i * 512 + tid, m = ci >> 3, q = (ci & 7) ^ (m & 7);
    aoff[i] = m0 + m < P.M ? (unsigned)(m0 + m) * (unsigned)P.A.ld * 2u + (unsigned)q * 16u : A_OOB;
  }
#pragma unroll
  for (int i = 0; i < NLB; ++i) {
    const int ci = i * 512 + tid, hr = ci >> 3, q = (ci & 7) ^ (hr & 7);
    const int il = hr / HRI, hp = hr - il * HRI;
    const int hy = hp / TWP, hx = hp - hy * TWP;
    const int iy = ty0 + hy, ix = tx0 + hx;                // the tile's own positions and one more row / column
    const bool ok = hr < HR && img0 + il < X.nimg && iy < X.H && ix < X.W;
    boff[i] = ok ? (unsigned)(((img0 + il) * X.H + iy) * X.W + ix) * (unsigned)X.ld * 2u + (unsigned)q * 16u : B_OOB;
  }
  char* const dA = sA + wave * 1024;
  char* const dB = sB + wave * 1024;
  auto issueA = [&](int slab, int step, int stage) {
    unsigned dead = slab < nsl ? 0u : A_OOB;
    asm volatile("" : "+v"(dead));
    const unsigned wk = (unsigned)(S2T_STEPS[step].wtap * P.K + slab * 64) * 2u;
#pragma unroll
    for (int i = 0; i < NLA; ++i) glds16(rsA, (aoff[i] + wk) | dead, dA + stage * ABUF + i * 8192);
  };
  auto issueB = [&](int slab, int part) {
    unsigned dead = slab < nsl ? 0u : B_OOB;
    asm volatile("" : "+v"(dead));
    glds16(rsB, (boff[part] + (unsigned)slab * 128u) | dead, dB + (slab & 1) * BBUF + part * 8192);
  };

  // FOLD: the image patch under the tile's 32 x 32 output pixels and one pixel around them, held in registers
  // across the main loop (the loads return under it) and written to LDS before the epilogues
  constexpr int NPV = FOLD ? (PW * PW + 511) / 512 : 1;
  float pv[NPV];
  if constexpr (FOLD) {
    const int IH = 2 * X.H, IW = 2 * X.W;
    const float* img = P.E.fold_x + (long long)img0 * IH * IW;
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int e = i * 512 + tid, py = e / PW, px = e - py * PW;
      const int iy = 2 * ty0 - 1 + py, ix = 2 * tx0 - 1 + px;
      const bool ok = e < PW * PW && (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
      const float v = img[ok ? iy * IW + ix : 0];
      pv[i] = ok ? v * P.E.fold_scale + P.E.fold_shift : 0.f;
    }
  }

  f32x4 acc[4][TM][TN];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[p][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = (lane & 15) * 128;
  const int kq0 = lane >> 4, kq1 = 4 + (lane >> 4);
  const int sw0 = (kq0 ^ (lane & 7)) << 4, sw1 = (kq1 ^ (lane & 7)) << 4;
  int hr0[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nl = wn * 64 + j * 16 + (lane & 15);
    const int il = nl / (TS * TS), p = nl % (TS * TS);
    hr0[j] = il * HRI + (p / TS) * TWP + (p % TS);
  }
  bf16x8_t fa[2][TM], fb[2][TN];
  auto load_frags = [&](int stage, int bbuf, int toff, int ks, bool new_pixels) {
    const char* sa = sA + stage * ABUF + wm * (BM / 2) * 128 + frow;
    const char* sb = sB + bbuf * BBUF;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8_t*>(sa + i * 2048 + (ks ? sw1 : sw0));
    if (!new_pixels) return;              // same shift as the step before: the pixel fragments are still in fb
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      int hr = hr0[j] + toff;
      asm volatile("" : "+v"(hr));      // recomputed per step: hoisted, the 9 x 8 addresses cost spills
      fb[ks][j] = *reinterpret_cast<const bf16x8_t*>(sb + hr * 128 + (((ks ? kq1 : kq0) ^ (hr & 7)) << 4));
    }
  };

#pragma unroll
  for (int part = 0; part < NLB; ++part) issueB(0, part);
  issueA(0, 0, 0);
  issueA(0, 1, 1);
  for (int slab = 0; slab < nsl; ++slab) {
#pragma unroll
    for (int step = 0; step < 9; ++step) {
      const int n_after = ((step + 7) % 9 < NLB ? 1 : 0) + NLA + ((step + 8) % 9 < NLB ? 1 : 0);
      if (n_after == NLA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLA) : "memory");
      else if (n_after == NLA + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLA + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLA + 2) : "memory");
      __builtin_amdgcn_s_barrier();
      const int toff = S2T_STEPS[step].shift_i * TWP + S2T_STEPS[step].shift_j;
      const int ph = S2T_STEPS[step].phase;
      const bool newp = step == 0 || S2T_STEPS[step].shift_i != S2T_STEPS[step - (step ? 1 : 0)].shift_i ||
                        S2T_STEPS[step].shift_j != S2T_STEPS[step - (step ? 1 : 0)].shift_j;
      load_frags(step % 3, slab & 1, toff, 0, newp);
      if (step < 7) issueA(slab, step + 2, (step + 2) % 3);
      else issueA(slab + 1, step - 7, (step + 2) % 3);
      if (step < NLB) issueB(slab + 1, step);
      load_frags(step % 3, slab & 1, toff, 1, newp);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[ph][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][i], fb[ks][j], acc[ph][i][j], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  struct PhaseRowMap {
    long long base; int Wo, HWo, nimg_left;
    __device__ __forceinline__ bool operator()(int nl, long long& orow) const {
      const int il = nl / (TS * TS), p = nl % (TS * TS);
      orow = base + (long long)il * HWo + (long long)(2 * (p / TS)) * Wo + 2 * (p % TS);
      return il < nimg_left;
    }
  };
  const int Wo = 2 * X.W, HWo = 4 * X.H * X.W;
  // the four phases' statistics are summed per thread and reduced once: one row per tile
  float carry[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) carry[q] = 0.f;
  f32x4 facc[4];              // FOLD: [tap 16][channel 16 j + ..] partial products of this wave's 32 pixels per phase
  float* const patch = reinterpret_cast<float*>(smem + (FOLD ? PATCH_OFF : 0));
  if constexpr (FOLD) {
#pragma unroll
    for (int j = 0; j < 4; ++j) facc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NPV; ++i)
      if (i * 512 + tid < PW * PW) patch[i * 512 + tid] = pv[i];
    __syncthreads();
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const PhaseRowMap rm{(long long)img0 * HWo + (long long)(2 * ty0 + (p >> 1)) * Wo + 2 * tx0 + (p & 1), Wo, HWo,
                         X.nimg - img0};
    if constexpr (FOLD) {
      // tap table of the phase: row n = pixel n of the tile, 16 bf16 high parts then 16 low parts (taps 0..8 used);
      // waves 0-3 write the high halves, waves 4-7 the low halves
      const int n = tid & 255, py = n >> 4, px = n & 15, half = tid >> 8;
      const float* pp = patch + (2 * py + (p >> 1)) * PW + 2 * px + (p & 1);
      unsigned short t16[16];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int z = 0; z < 3; ++z) {
          const float xv = pp[r * PW + z];
          const unsigned hi = pack2bf(xv, 0.f) & 0xFFFFu;
          const float rest = xv - __uint_as_float(hi << 16);
          t16[r * 3 + z] = half ? (unsigned short)(pack2bf(rest, 0.f) & 0xFFFFu) : (unsigned short)hi;
        }
#pragma unroll
      for (int t = 9; t < 16; ++t) t16[t] = 0;
      u32x4 w0, w1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        w0[e] = (unsigned)t16[2 * e] | ((unsigned)t16[2 * e + 1] << 16);
        w1[e] = (unsigned)t16[8 + 2 * e] | ((unsigned)t16[9 + 2 * e] << 16);
      }
      char* xr = smem + XT_OFF + n * FOLD_XT_PITCH + half * 32;
      *reinterpret_cast<u32x4*>(xr) = w0;
      *reinterpret_cast<u32x4*>(xr + 16) = w1;
    }
    if (p < 3)
      gemm_epilogue_staged<BM, BN, WN, PIPE, PhaseRowMap, 1, FOLD>(P, acc[p], m0, rm, tn, ntotal / P.tiles_m, 0, smem,
                                                                  carry, smem + DT_OFF);
    else
      gemm_epilogue_staged<BM, BN, WN, PIPE, PhaseRowMap, 2, FOLD>(P, acc[p], m0, rm, tn, ntotal / P.tiles_m, 0, smem,
                                                                  carry, smem + DT_OFF);
    __syncthreads();          // the LDS staging area is reused by the next phase
    if constexpr (FOLD) {
      // wave w: pixels [32 w, 32 w + 32) of the phase; both tiles are [pixel][row] images read with the transposing load
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
      const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
      const int k = wave * 32 + 8 * g + q;
      auto tr_frag = [&](const char* base, int pitch, int row0) {
        const char* a0 = base + k * pitch + (row0 + 4 * pq) * 2;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * pitch));
        s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return __builtin_bit_cast(bf16x8_t, r);
      };
      const bf16x8_t xh = tr_frag(smem + XT_OFF, FOLD_XT_PITCH, 0), xl = tr_frag(smem + XT_OFF, FOLD_XT_PITCH, 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8_t d = tr_frag(smem + DT_OFF, FOLD_DT_PITCH, 16 * j);
        facc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, d, facc[j], 0, 0, 0);
        facc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, d, facc[j], 0, 0, 0);
      }
      __syncthreads();        // the tables are rewritten by the next phase
    }
  }
  if constexpr (FOLD) {
    // the eight waves' sums through LDS (the staging tile is free), then one row [64][9] per block
    float* red = reinterpret_cast<float*>(smem);           // [wave][tap 16][channel 64]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        red[wave * 1024 + ((lane >> 4) * 4 + e) * 64 + j * 16 + (lane & 15)] = facc[j][e];
    __syncthreads();
    float* row = P.E.fold_rows + (long long)tn * (BM * 9);
    for (int o = tid; o < BM * 9; o += 512) {
      const int co = o / 9, tap = o - co * 9;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += red[w * 1024 + tap * 64 + co];
      row[o] = t;
    }
  }
}

// ---------------------------------------------------------------------------
// ConvTranspose2d(kernel 4, stride 2, padding 1) forward (src/rinet.py:216,219,272,276 in decoder mode, :68-69;
// src/networks.py:479-484), halo-tile form.  Output (2i+a, 2j+b) takes input (i+di, j+dj) through tap (r, s) with
// a + 1 - r = 2 di and b + 1 - s = 2 dj: every phase (a, b) is a 2x2-tap convolution, and the four of them read
// the same 3x3 neighbourhood.  The general kernel runs the phases as four GEMMs of four taps each and fetches the
// input once per tap: with few output channels (the generator's last layers have 32 and 64) that is all
// L2 -> LDS traffic and no arithmetic.  Here, as in conv_halo_s2t_kernel, a block owns 16 x 16 input positions of
// one image and all four phases of their 32 x 32 output pixels: the 18 x 18 input tile is loaded once per
// 64-channel slab and the sixteen (phase, tap) products of the slab are the steps of the pipeline.
// Cin % 32 == 0: the second half of the last slab is skipped when Cin % 64 == 32 (its chunks are fetched - the
// next pixel's channels, the next tap's weights, finite values or zeros - but never multiplied).
// BM output channels per block (32 or 64); the weight stage always holds 64 rows (rows past Cout load as zeros).
// ---------------------------------------------------------------------------
struct D4Step { int phase, wtap, shift_i, shift_j; };
// Phase (a, b) = 2a + b uses rows r in {1, 3} (a = 0) or {0, 2} (a = 1) and columns s likewise.  The sixteen
// products are ordered by input shift: steps that read the same shifted pixels follow each other and reuse the
// pixel fragments in registers (nine fragment loads per slab instead of sixteen); neighbours in the list add into
// different phases' accumulators.
__device__ constexpr D4Step D4_STEPS[16] = {
    {0, 5, 0, 0},   {1, 6, 0, 0},   {2, 9, 0, 0},  {3, 10, 0, 0},        // shift ( 0,  0)
    {0, 7, 0, -1},  {2, 11, 0, -1},                                      // shift ( 0, -1)
    {0, 13, -1, 0}, {1, 14, -1, 0},                                      // shift (-1,  0)
    {0, 15, -1, -1},                                                     // shift (-1, -1)
    {1, 4, 0, 1},   {3, 8, 0, 1},                                        // shift ( 0,  1)
    {1, 12, -1, 1},                                                      // shift (-1,  1)
    {2, 1, 1, 0},   {3, 2, 1, 0},                                        // shift ( 1,  0)
    {2, 3, 1, -1},                                                       // shift ( 1, -1)
    {3, 0, 1, 1}};                                                       // shift ( 1,  1)

// TSY = 8 (BM = 32): tiles of 8 x 16 input positions.  Half the accumulators and half the input tile: 80 KB of LDS and
// under 128 registers, so two blocks share a CU and one's four epilogues run under the other's main loop (the layer
// with 32 output channels moves 0.9 GB at 512 x 512 output and was bound by neither pipe nor memory: one block per
// CU, half of its time in the epilogues).
// XF (input transform, GemmParams::xf_*): the input is the RAW output of the producing layer(s) and each 64-channel
// slab of the tile is normalised in place in LDS - x -> relu(x * scale[c] + shift[c]), pixels outside the map stay zero
// - once, before its sixteen products read it (the tile serves all taps, so the transform is paid once per element, not
// once per tap as in the general kernel).  The per-channel scale and shift sit in a thread's registers from the start
// of the kernel and reach the slab's threads through the slack behind the OTHER tile buffer (the lane-linear DMA
// destination is longer than the tile), which no load touches between the write and the read.
template <int BM, int TSY = 16, bool XF = false>
__global__ __launch_bounds__(512, TSY == 8 ? 4 : 2) void conv_halo_d4_kernel(const GemmParams P) {
  // BM = 32: a weight stage (64 rows) holds the taps of TWO consecutive products, rows 0-31 and 32-63, and a step runs
  // both between one pair of barriers: 16 MFMAs per wave and barrier instead of 8 (the layer with 32 output channels
  // was bound by its barriers), eight steps per slab.
  constexpr int PAIR = BM == 32 ? 2 : 1;
  constexpr int TS = 16, BN = TSY * TS, WN = 4, NSTEP = 16 / PAIR;
  constexpr int TWP = TS + 2, HR = (TSY + 2) * TWP;        // 324 (180) input pixels per block
  constexpr int NLB = (HR * 8 + 511) / 512;                // 6 (3)
  constexpr int BBUF = NLB * 512 * 16;
  constexpr int NLA = 1, ABUF = 64 * 128, NSA = 4;         // weight ring: 4 stages of 64 rows x 64 channels
  constexpr int TM = BM / 32, TN = BN / 64;
  static_assert(BM == 32 || BM == 64, "output-channel tile");
  static_assert(TSY == 16 || TSY == 8, "tile height");
  constexpr unsigned A_OOB = 0xFFFF0000u, B_OOB = 0xFFFFF000u;
  __shared__ __attribute__((aligned(1024))) char smem[NSA * ABUF + 2 * BBUF];
  char* const sA = smem;
  char* const sB = smem + NSA * ABUF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const GemmOperand& X = P.B;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A.p, 0, P.A_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)P.B.p, 0, P.B_bytes, 0x00020000);
  const int ntotal = gridDim.x;
  int bid = blockIdx.x;
  {
    const int nwg = ntotal, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid % P.tiles_m, tn = bid / P.tiles_m;
  const int m0 = tm * BM;
  const int tX = X.W / TS, tY = X.H / TSY;
  const int img0 = tn / (tX * tY), trem = tn % (tX * tY);
  const int ty0 = (trem / tX) * TSY, tx0 = (trem % tX) * TS;
  const int nsl = (P.K + 63) >> 6;
  const bool half_tail = (P.K & 63) != 0;                  // Cin % 64 == 32

  unsigned aoff[NLA], boff[NLB];
#pragma unroll
  for (int i = 0; i < NLA; ++i) {
    const int ci = i * 512 + tid, m = ci >> 3, q = (ci & 7) ^ (m & 7);
    const int mr = PAIR == 2 ? (m & 31) : m;               // (paired stages: rows 32-63 are the second tap's rows 0-31)
    aoff[i] = m0 + mr < P.M ? (unsigned)(m0 + mr) * (unsigned)P.A.ld * 2u + (unsigned)q * 16u : A_OOB;
  }
  int second_tap_i = (PAIR == 2 && ((tid >> 3) & 32)) ? 1 : 0;   // this lane's weight row belongs to the step's second tap
  asm volatile("" : "+v"(second_tap_i));
#pragma unroll
  for (int i = 0; i < NLB; ++i) {
    const int ci = i * 512 + tid, hr = ci >> 3, q = (ci & 7) ^ (hr & 7);
    const int hy = hr / TWP, hx = hr - hy * TWP;
    const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;        // the tile's own positions and a ring of one around them
    const bool ok = hr < HR && (unsigned)iy < (unsigned)X.H && (unsigned)ix < (unsigned)X.W;
    boff[i] = ok ? (unsigned)((img0 * X.H + iy) * X.W + ix) * (unsigned)X.ld * 2u + (unsigned)q * 16u : B_OOB;
  }
  char* const dA = sA + wave * 1024;
  char* const dB = sB + wave * 1024;
  // XF: this thread's channel (Cin <= 512, host-checked) and the validity of the pixels its chunks belong to
  static_assert(!XF || BBUF - HR * 128 >= 512, "room for a slab's scale / shift table behind the tile");
  float xsc = 0.f, xsh = 0.f;
  unsigned xok = 0u;
  if constexpr (XF) {
    if (tid < P.K) {
      const float a = P.xf_gamma[tid] * P.xf_save[P.K + tid];
      xsc = a; xsh = P.xf_beta[tid] - P.xf_save[tid] * a;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int hr = (i * 512 + tid) >> 3, hy = hr / TWP, hx = hr - hy * TWP;
      const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
      if (hr < HR && (unsigned)iy < (unsigned)X.H && (unsigned)ix < (unsigned)X.W) xok |= 1u << i;
    }
  }
  // table of slab s: [scale 64 | shift 64] floats behind tile buffer (s + 1) & 1, written by threads 64 s .. 64 s + 63
  auto xf_table = [&](int slab) { return reinterpret_cast<float*>(sB + ((slab + 1) & 1) * BBUF + HR * 128); };
  auto xf_publish = [&](int slab) {
    if ((tid >> 6) == slab) { float* t = xf_table(slab); t[tid & 63] = xsc; t[64 + (tid & 63)] = xsh; }
  };
  auto xf_apply = [&](int slab) {
    const float* tab = xf_table(slab);
    char* tile = sB + (slab & 1) * BBUF;
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int ci = i * 512 + tid, hr = ci >> 3, q = (ci & 7) ^ (hr & 7);
      if (!((xok >> i) & 1u) || slab * 64 + q * 8 >= P.K) continue;
      u32x4* p = reinterpret_cast<u32x4*>(tile + ci * 16);
      const u32x4 v = *p;
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(tab + q * 8), s1 = *reinterpret_cast<const f32x4*>(tab + q * 8 + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(tab + 64 + q * 8), h1 = *reinterpret_cast<const f32x4*>(tab + 64 + q * 8 + 4);
      float f[8] = {bflo(v[0]), bfhi(v[0]), bflo(v[1]), bfhi(v[1]), bflo(v[2]), bfhi(v[2]), bflo(v[3]), bfhi(v[3])};
#pragma unroll
      for (int e = 0; e < 4; ++e) { f[e] = f[e] * s0[e] + h0[e]; f[4 + e] = f[4 + e] * s1[e] + h1[e]; }
      if (P.xf_act == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
      }
      *p = u32x4{pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7])};
    }
  };
  auto issueA = [&](int slab, int step, int stage) {
    unsigned dead = slab < nsl ? 0u : A_OOB;
    asm volatile("" : "+v"(dead));
    // (the second tap's rows: tap A + (tap B - tap A) * 1, in arithmetic - written as a choice between two table
    // entries the compiler loads the entry from memory through a per-lane index and waits for vmcnt(0), which drains
    // every weight DMA in flight, at every step)
    const int wta = D4_STEPS[step * PAIR].wtap, wtd = D4_STEPS[step * PAIR + PAIR - 1].wtap - wta;
    const unsigned wk = (unsigned)((wta + wtd * second_tap_i) * P.K + slab * 64) * 2u;
#pragma unroll
    for (int i = 0; i < NLA; ++i) glds16(rsA, (aoff[i] + wk) | dead, dA + stage * ABUF + i * 8192);
  };
  auto issueB = [&](int slab, int part) {
    unsigned dead = slab < nsl ? 0u : B_OOB;
    asm volatile("" : "+v"(dead));
    glds16(rsB, (boff[part] + (unsigned)slab * 128u) | dead, dB + (slab & 1) * BBUF + part * 8192);
  };

  f32x4 acc[4][TM][TN];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[p][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = (lane & 15) * 128;
  const int kq0 = lane >> 4, kq1 = 4 + (lane >> 4);
  const int sw0 = (kq0 ^ (lane & 7)) << 4, sw1 = (kq1 ^ (lane & 7)) << 4;
  int hr0[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int p = wn * (BN / 4) + j * 16 + (lane & 15);
    hr0[j] = (p / TS + 1) * TWP + (p % TS) + 1;
  }
  bf16x8_t fa[2][TM], fb[2][TN];
  auto load_frags = [&](int stage, int sub, int bbuf, int toff, int ks, bool new_pixels) {
    const char* sa = sA + stage * ABUF + (sub * 32 + wm * (BM / 2)) * 128 + frow;
    const char* sb = sB + bbuf * BBUF;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8_t*>(sa + i * 2048 + (ks ? sw1 : sw0));
    if (!new_pixels) return;              // same shift as the step before: the pixel fragments are still in fb
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      int hr = hr0[j] + toff;
      // recomputed per step: hoisted out of the slab loop, the 16 x 8 distinct addresses would not fit the
      // register file beside the four accumulator sets
      asm volatile("" : "+v"(hr));
      fb[ks][j] = *reinterpret_cast<const bf16x8_t*>(sb + hr * 128 + (((ks ? kq1 : kq0) ^ (hr & 7)) << 4));
    }
  };

#pragma unroll
  for (int part = 0; part < NLB; ++part) issueB(0, part);
  issueA(0, 0, 0);
  issueA(0, 1, 1);
  issueA(0, 2, 2);
  if constexpr (XF) xf_publish(0);
  for (int slab = 0; slab < nsl; ++slab) {
    const bool both = !(half_tail && slab == nsl - 1);
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      // still in flight after this wait: what was issued after this step's weights (three steps ago): the
      // weights of the next two steps and up to three parts of the next slab's input tile.  (A block is alone on
      // its CU and a step is only 8-16 MFMAs per wave: three steps of weights in flight, not two, cover the L2
      // latency.)
      const int n_after = 2 * NLA + ((step + NSTEP - 3) % NSTEP < NLB ? 1 : 0) +
                          ((step + NSTEP - 2) % NSTEP < NLB ? 1 : 0) + ((step + NSTEP - 1) % NSTEP < NLB ? 1 : 0);
      if (n_after == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (n_after == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (n_after == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if constexpr (XF) {
        if (step == 0) {       // the slab's tile has landed and nobody has read it yet
          xf_apply(slab);
          xf_publish(slab + 1);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (no vmcnt wait: the weight DMAs stay in flight)
          __builtin_amdgcn_s_barrier();
        }
      }
#pragma unroll
      for (int sub = 0; sub < PAIR; ++sub) {
        const int t = step * PAIR + sub;                       // index into the sixteen (phase, tap) products
        const int toff = D4_STEPS[t].shift_i * TWP + D4_STEPS[t].shift_j;
        const int ph = D4_STEPS[t].phase;
        const bool newp = t == 0 || D4_STEPS[t].shift_i != D4_STEPS[t - (t ? 1 : 0)].shift_i ||
                          D4_STEPS[t].shift_j != D4_STEPS[t - (t ? 1 : 0)].shift_j;
        load_frags(step % NSA, sub, slab & 1, toff, 0, newp);
        if (sub == 0) {
          if (step < NSTEP - 3) issueA(slab, step + 3, (step + 3) % NSA);
          else issueA(slab + 1, step + 3 - NSTEP, (step + 3) % NSA);
          if (step < NLB) issueB(slab + 1, step);
        }
        if (both) load_frags(step % NSA, sub, slab & 1, toff, 1, newp);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[ph][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[0][j], acc[ph][i][j], 0, 0, 0);
        if (both) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[ph][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][i], fb[1][j], acc[ph][i][j], 0, 0, 0);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  struct PhaseRowMap {
    long long base; int Wo;
    __device__ __forceinline__ bool operator()(int nl, long long& orow) const {
      orow = base + (long long)(2 * (nl / TS)) * Wo + 2 * (nl % TS);
      return true;
    }
  };
  const int Wo = 2 * X.W, HWo = 4 * X.H * X.W;
  // the four phases' statistics are summed per thread and reduced once: one row per tile
  float carry[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) carry[q] = 0.f;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const PhaseRowMap rm{(long long)img0 * HWo + (long long)(2 * ty0 + (p >> 1)) * Wo + 2 * tx0 + (p & 1), Wo};
    if (p < 3)
      gemm_epilogue_staged<BM, BN, WN, NSA * ABUF + 2 * BBUF, PhaseRowMap, 1>(P, acc[p], m0, rm, tn,
                                                                              ntotal / P.tiles_m, 0, smem, carry);
    else
      gemm_epilogue_staged<BM, BN, WN, NSA * ABUF + 2 * BBUF, PhaseRowMap, 2>(P, acc[p], m0, rm, tn,
                                                                              ntotal / P.tiles_m, 0, smem, carry);
    __syncthreads();          // the LDS staging area is reused by the next phase
  }
}

// ---------------------------------------------------------------------------
// Weight gradient, LDS-DMA form.  Both operands are pixel-major (K = pixels is the slow axis), so a 16-byte DMA
// chunk is 8 channels of one pixel and the lane-linear destination gives the [k][row] image that
// ds_read_b64_tr_b16 wants; its XOR swizzle (TileGeom::tr_swz) is applied to the SOURCE chunk a lane fetches.
// A = the gathered tensor (shifted by the block's tap, zero padding by out-of-range offsets), B = the tensor
// indexed by output pixel.  Grid as the register-staged form: (tile, tap, k split); stages past the block's K
// range fetch zeros so that the DMA count per stage is exact for the counted waits.  Measured against the
// register-staged form on the discriminator's layers: 2-10 % faster with two stages; three stages (one block per
// CU) are 40 % slower.
// ---------------------------------------------------------------------------
// NT > 1: a block owns the NT taps s0 .. s0 + NT - 1 of kernel row r0 (kw % NT == 0) and keeps one accumulator set
// per tap.  A stage then holds NT gathered tiles and ONE tile of the pixel-indexed tensor, which serves all of them:
// with few channels these kernels are bound by L2 -> LDS traffic (a 64 x 128 tile moves 24 KB per 64 pixels and tap),
// and per tap the block now fetches (NT * A + B) / NT instead of A + B.  Grid: (tile, kh * kw / NT, k split).
template <int BM, int BN, int NST, int NT = 1>
__device__ __forceinline__ void wgrad_glds_body(const GemmParams& P, int bx, int by, int bz, int ntiles_n,
                                                unsigned nblocks, char* smem) {
  using GA = TileGeom<BM>;
  using GB = TileGeom<BN>;
  constexpr int TM = BM / 32, TN = BN / 32;
  constexpr int NCA = BM / 32, NCB = BN / 32;          // DMA loads per lane per stage and operand tile
  constexpr int ASZ = BK * BM * 2, BSZ = BK * BN * 2, STAGE = NT * ASZ + BSZ;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tm = bx % P.tiles_m, tn = bx / P.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const GemmOperand& G = P.A;
  const int gpr = G.kw / NT;                           // tap groups per kernel row
  const int r0 = by / gpr, s0 = (by % gpr) * NT;
  const int per = ((P.K + P.splitk - 1) / P.splitk + BK - 1) / BK * BK;
  const int kbeg = bz * per, kend = min(P.K, kbeg + per);
  const int nsteps = (max(kend - kbeg, 0) + BK - 1) / BK;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A.p, 0, P.A_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)P.B.p, 0, P.B_bytes, 0x00020000);
  // A: LDS chunk (k = kA + KR*i, slot) <- channels (slot ^ swizzle(k)) * 8 of that pixel; the swizzle is the same
  // for every i (KR is a multiple of 16)
  const int kA = tid / GA::CPR, slotA = tid % GA::CPR;
  const int chA = m0 + ((slotA * 8) ^ GA::tr_swz(kA));
  unsigned deadA = chA < P.M ? 0u : GLDS_OOB;
  const unsigned colA = (unsigned)chA * 2u;
  const int kB = tid / GB::CPR, slotB = tid % GB::CPR;
  const int chB = n0 + ((slotB * 8) ^ GB::tr_swz(kB));
  unsigned deadB = chB < P.N ? 0u : GLDS_OOB;
  asm volatile("" : "+v"(deadA), "+v"(deadB));
  const int hw = G.Ho * G.Wo;
  int gn[NCA], gy[NCA], gx[NCA];
#pragma unroll
  for (int i = 0; i < NCA; ++i) {
    const int k = kbeg + kA + GA::KR * i;
    const int n = k / hw, r2 = k - n * hw;
    gn[i] = n; gy[i] = r2 / G.Wo; gx[i] = r2 - (r2 / G.Wo) * G.Wo;
  }
  const int dn = BK / hw, drem = BK - dn * hw, dy = drem / G.Wo, dx = drem - dy * G.Wo;
  const unsigned ldg2 = (unsigned)G.ld * 2u, ldp2 = (unsigned)P.B.ld * 2u;
  unsigned boff[NCB];
#pragma unroll
  for (int i = 0; i < NCB; ++i) boff[i] = (unsigned)(kbeg + kB + GB::KR * i) * ldp2 + (unsigned)chB * 2u;
  int kcur = kbeg;
  char* const lds_wave = smem + wave * 1024;
  auto issue = [&](int buf) {
    char* dst = lds_wave + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
      const int hi = gy[i] * G.sh - G.ph + r0, wi0 = gx[i] * G.sw - G.pw + s0;
      const bool rok = kcur + kA + GA::KR * i < kend && (unsigned)hi < (unsigned)G.H;
      const unsigned rowoff = (unsigned)((gn[i] * G.H + hi) * G.W + wi0) * ldg2 + colA;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bool ok = rok && (unsigned)(wi0 + t) < (unsigned)G.W;
        unsigned bad = ok ? 0u : GLDS_OOB;
        glds16(rsA, (rowoff + (unsigned)t * ldg2) | bad | deadA, dst + t * ASZ + i * 4096);
      }
      int x = gx[i] + dx, y = gy[i] + dy, n = gn[i] + dn;
      if (x >= G.Wo) { x -= G.Wo; ++y; }
      if (y >= G.Ho) { y -= G.Ho; ++n; }
      gx[i] = x; gy[i] = y; gn[i] = n;
    }
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      unsigned bad = kcur + kB + GB::KR * i < kend ? 0u : GLDS_OOB;
      glds16(rsB, boff[i] | bad | deadB, dst + NT * ASZ + i * 4096);
      boff[i] += (unsigned)BK * ldp2;
    }
    kcur += BK;
  };

  f32x4 acc[NT][TM][TN];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8_t fa[2][TM], fb[2][TN];
  auto load_a = [&](int buf, int t, int ks) {
    const bf16_t* sa = reinterpret_cast<const bf16_t*>(smem + buf * STAGE + t * ASZ);
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[ks][i] = read_frag<OP_PLAIN_TR, BM>(sa, wm * (BM / 2) + i * 16, lane, ks);
  };
  auto load_b = [&](int buf, int ks) {
    const bf16_t* sb = reinterpret_cast<const bf16_t*>(smem + buf * STAGE + NT * ASZ);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[ks][j] = read_frag<OP_PLAIN_TR, BN>(sb, wn * (BN / 2) + j * 16, lane, ks);
  };
#pragma unroll
  for (int st = 0; st < NST - 1; ++st) issue(st);
  int rd = 0, wr = NST - 1;
  for (int step = 0; step < nsteps; ++step) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (NT * NCA + NCB)) : "memory");
    __builtin_amdgcn_s_barrier();
    load_a(rd, 0, 0);
    load_b(rd, 0);
    issue(wr);
    load_a(rd, 0, 1);
    load_b(rd, 1);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t) { load_a(rd, t, 0); load_a(rd, t, 1); }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][i], fb[ks][j], acc[t][i][j], 0, 0, 0);
    }
    rd = rd + 1 == NST ? 0 : rd + 1;
    wr = wr + 1 == NST ? 0 : wr + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int t = 0; t < NT; ++t)
    gemm_epilogue<BM, BN, true>(P, acc[t], m0, LinearRowMap{P, n0, 0, 0}, tn, 0, ntiles_n, r0 * G.kw + s0 + t, bz,
                                nblocks, smem);
}

template <int BM, int BN, int NST, int NT = 1>
__global__ __launch_bounds__(256) void wgrad_glds_kernel(const GemmParams P) {
  __shared__ __attribute__((aligned(1024))) char smem[NST * BK * (NT * BM + BN) * 2];
  // Workgroups go to the XCDs round-robin by their linear index (tile fastest, then tap, then k split), while the
  // blocks that read the same pixels are those of one k split: renumbered so that a split's blocks run on ONE XCD
  // (its pixels are fetched into one L2, not eight).  The splits beyond the last whole round of 8 keep their numbers.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  const int per = gridDim.x * gridDim.y, whole = (int)gridDim.z & ~7;
  const int id = bx + (int)gridDim.x * (by + (int)gridDim.y * bz);
  if (id < per * whole) {
    const int xcd = id & 7, slot = id >> 3, w = slot % per;
    bz = (slot / per) * 8 + xcd;
    bx = w % (int)gridDim.x;
    by = w / (int)gridDim.x;
  }
  wgrad_glds_body<BM, BN, NST, NT>(P, bx, by, bz, gridDim.x / P.tiles_m, gridDim.x * gridDim.y * gridDim.z, smem);
}

template <int BM, int BN, int NST, int NT = 1>
__global__ __launch_bounds__(256) void wgrad_glds_jobs_kernel(const GemmParams* __restrict__ jobs,
                                                              const int* __restrict__ map) {
  __shared__ __attribute__((aligned(1024))) char smem[NST * BK * (NT * BM + BN) * 2];
  const GemmParams& P = jobs[map[2 * blockIdx.x]];
  const int lb = map[2 * blockIdx.x + 1];
  const int nx = P.grid[0], ny = P.grid[1];
  const int bx = lb % nx, r = lb / nx;
  wgrad_glds_body<BM, BN, NST, NT>(P, bx, r % ny, r / ny, nx / P.tiles_m, nx * ny * P.grid[2], smem);
}

// Tile of the convolution kernel for a problem: Cout x pixels per phase.  Small problems take 64-pixel tiles
// (more blocks than CUs matter more than MFMA efficiency there).
struct ConvTile { int bm, bn, nst, ks; };      // ks: K groups per block (conv_glds_body)
static int g_forced_tile[3] = {0, 0, 0};
// Tuning hook: force the tile (bm, bn, nst) of every later ried_conv2d call in this process; bm = 0 restores the
// built-in choice; bm = -1 keeps the built-in tiles of the general kernel.  Any non-zero bm turns the halo-tile
// kernel off.  Layers with Cout <= 64 keep bm = 64.
extern "C" int ried_conv2d_force_tile(int bm, int bn, int nst) {
  g_forced_tile[0] = bm; g_forced_tile[1] = bn; g_forced_tile[2] = nst;
  return RIED_OK;
}
extern "C" int ried_bn_fold_rows();
#define KSPLIT_MIN_STEPS 12
static ConvTile choose_conv_tile(int Cout, long long npix, int phases, int ksteps, int taps) {
  const int* forced = g_forced_tile;
  ConvTile t;
  // Measured on MI355X over the layers of the training step: two LDS stages always beat three or four (blocks per
  // CU are worth more than DMA depth), and the 64-pixel tile wins whenever the 128-pixel grid would leave CUs
  // idle or end in a mostly empty last round.  Model: time ~ rounds over the resident-block slots x tile cost.
  t.bm = Cout <= 64 ? 64 : 128;
  t.nst = 2; t.ks = 1;
  const long long tiles_m = ceil_div(Cout, t.bm);
  const long long b128 = tiles_m * ceil_div(npix, 128) * phases, b64 = tiles_m * ceil_div(npix, 64) * phases;
  const int cus = 256;
  const int slots128 = cus * (t.bm == 128 ? 2 : 3), slots64 = cus * (t.bm == 128 ? 3 : 5);
  const double cost128 = (double)ceil_div(b128, slots128) * 1.0, cost64 = (double)ceil_div(b64, slots64) * 0.62;
  t.bn = (cost128 <= cost64 || b128 >= 4 * slots128) ? 128 : 64;     // many rounds: the larger tile's reuse wins
  // Small grids (measured: every layer on 4x4 maps, and 8x8 maps with few output channels): 64x64 tiles put
  // twice as many blocks on the chip and win by 15-25 %; with a long K loop and at most two blocks per CU
  // anyway, four LDS stages hide the per-step latency.
  if (t.bm == 128 && b64 <= 256) {
    t.bm = 64; t.bn = 64;
    if (ksteps >= 24) t.nst = 4;
  }
  // Large maps of heavy layers (measured on the discriminator's strided and 1x1 layers and the 4x4 deconvs):
  // 256-pixel tiles on 8 waves halve the weight traffic per pixel; 64 output channels per block unless the
  // layer has at least 256.
  if (npix >= 32768 && (double)npix * ksteps * 64.0 * Cout * phases >= 2.5e9) {
    t.bn = 256; t.nst = 2;
    t.bm = Cout >= 256 ? 128 : 64;
    // measured exception (512x512 inference, deconv1: 4 phases of 128x128 maps, 64 output channels): 383 vs 506 us
    if (phases > 1 && Cout == 64 && npix >= (1 << 19)) t.bn = 128;
    // the heaviest 1x1 layers (512x512 inference: 256 -> 224 at 128x128 maps, 512 -> 512 at 64x64) stream their
    // operands once and want more bytes in flight: 128 channels per block and three stages, 223 -> 193 and 136 -> 122 us
    // 1x1 layers stream both operands once: 64 channels per block (more blocks in flight) measured faster than 128
    // for 256 -> 320 (35.2 -> 27.5 us), 320 -> 256 (30.8 -> 28.4) and 256 -> 256 at 64x64 maps (55.0 -> 45.1) ...
    static const bool small1x1 = !getenv("RIED_NO_SMALL1X1");
    if (small1x1 && taps == 1 && phases == 1) t.bm = 64;
    // ... except the heaviest ones:
    static const bool big1x1 = !getenv("RIED_NO_BIG1X1");
    if (big1x1 && taps == 1 && phases == 1 && Cout >= 192 && (double)npix * ksteps * 64.0 * Cout >= 2.5e10) {
      t.bm = 128; t.nst = 3;
    }
  }
  // Layers with at most 32 output channels (the narrow Inception branches, deconv2): 32-row tiles waste no MFMA
  // rows and measured 10-30 % faster everywhere; 128 pixels per block only for multi-tap layers on large maps.
  if (Cout <= 32) {
    t.bm = 32; t.nst = 2;
    t.bn = (taps > 1 && npix >= 131072) ? 128 : 64;
  }
  // Short layers on grids that leave the chip under-filled (the generator's 8x8 and 4x4 maps): 32 output channels
  // per block double the blocks again; measured 6-10 % faster alone (6.7 -> 6.1, 14.0 -> 12.6 us) where K is short.
  static const bool tile32 = !getenv("RIED_NO_TILE32");
  const long long tiles64 = ceil_div(Cout, 64) * ceil_div(npix, 64) * phases;
  if (tile32 && Cout > 32 && t.bm == 64 && t.bn == 64 && t.nst == 2 && tiles64 <= 384 && ksteps < 24)
    t.bm = 32;
  // One block per CU at most and a long K loop: two K groups per block (measured alone on the VGG16 conv5 layers of
  // the perceptual loss at 4 x 4 maps, 72 stages: see conv_glds_body).
  static const bool ksplit = !getenv("RIED_NO_KSPLIT");
  if (ksplit && t.bm == 64 && t.bn == 64) {
    // measured alone (us, one group -> two): conv5 of VGG16 at 4 x 4 maps 28.2 -> 20.9 (three stages; 26.1 with two),
    // 3 x 3 224 -> 192 at 4 x 4 14.0 -> 10.8, 1 x 7 128 -> 128 at 8 x 8 9.0 -> 7.9, and with two blocks per CU the
    // discriminator's 3 x 3 stride-2 512 -> 512 at 8 x 8 35.0 -> 31.7
    if (tiles64 <= 256 && ksteps >= KSPLIT_MIN_STEPS) { t.ks = 2; t.nst = ksteps >= 48 ? 3 : 2; }
    else if (tiles64 <= 512 && ksteps >= 48 && phases == 1) { t.ks = 2; t.nst = 2; }
  }
  if (forced[0] > 0) {       // (nst + 10: two K groups)
    t.bm = forced[0]; t.bn = forced[1]; t.nst = forced[2] % 10; t.ks = forced[2] >= 10 ? 2 : 1;
  }
  return t;
}

// K steps (of 64) of the longest phase
static int conv_ksteps(int Cin, int kh, int kw, int sh, int sw, int transposed) {
  int taps = transposed ? ceil_div(kh, sh) * ceil_div(kw, sw) : kh * kw;
  return ceil_div((long long)taps * Cin, 64);
}

// rows of the partial-statistics buffer a ried_conv2d call with stats writes: [rows][2][Cout]
extern "C" int ried_conv2d_stat_rows(int nimg, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int sh, int sw,
                                     int transposed) {
  long long npix = transposed ? (long long)nimg * (Ho / sh) * (Wo / sw) : (long long)nimg * Ho * Wo;
  int phases = transposed ? sh * sw : 1;
  ConvTile t = choose_conv_tile(Cout, npix, phases, conv_ksteps(Cin, kh, kw, sh, sw, transposed), kh * kw);
  return phases * ceil_div(npix, t.bn);
}

// ---------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------
template <int BM, int BN, int AK, int BKIND>
static void launch(const GemmParams& P, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL((igemm_kernel<BM, BN, AK, BKIND>), grid, dim3(256), 0, s, P);
}

static void default_epilogue(GemmEpilogue& E) {
  memset(&E, 0, sizeof(E));
  E.alpha = 1.f;
}

// job: instead of launching, write the problem as a record for ried_conv2d_jobs (general kernel, 64 x 64 tiles)
// which: instead of launching, report the kernel the call would take: {kind, bm, bn or tile side, stages} with kind
// 0 = conv_glds_kernel, 1 = conv_halo_kernel, 2 = conv_halo_s2t_kernel, 3 = conv_halo_d4_kernel, 4 = conv_halo_s2f_kernel
extern "C" int ried_rows_add(const float* part, int nrows, int width, float* out, hipStream_t stream);
struct ConvJobOut { void* rec; int* nblocks; int* which; };
// input transform of the general kernel (GemmParams::xf_*)
struct ConvXf { const float* save; const float* gamma; const float* beta; int act; };
// first-layer weight gradient folded into the stride-2 data gradient (GemmEpilogue::fold_*)
struct ConvFold { const float* x; float scale, shift; float* rows; };

// Conv2d forward (transposed = 0) or ConvTranspose2d forward / strided-conv data gradient
// (transposed = 1).  x: NHWC bf16 [nimg][H][W][ldx], channels 0..Cin-1 used.
// w: packed bf16 [Cout][kh][kw][Cin].  y: NHWC [nimg][Ho][Wo][ldy], channels 0..Cout-1 written.
static int conv2d_impl(const void* x, long long ldx, const void* w, void* y, long long ldy, int nimg, int H, int W,
                       int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                       int transposed, const float* bias, int act, float* stats, int accumulate, int out_f32,
                       float* y2, long long ldy2, const void* mask, long long ldmask, float mask_slope,
                       int stats_post, const BnFinalize* fin, hipStream_t stream, const float* sub = nullptr,
                       long long ld_sub = 0, void* gout = nullptr, long long ld_gout = 0, float gscale = 0.f,
                       const ConvJobOut* job = nullptr, const ConvXf* xf = nullptr, int mask_bits = 0,
                       const ConvFold* fold = nullptr, int sub_bf16 = 0) {
  RIED_CHECK_ARG(x && w && (y || fold), "ried_conv2d: null pointer");
  RIED_CHECK_ARG(nimg > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "ried_conv2d: bad spatial dims");
  RIED_CHECK_ARG(Cin % 8 == 0 && Cout % 8 == 0 && ldx % 8 == 0 && ldy % 4 == 0 && Cin <= ldx && Cout <= ldy,
                 "ried_conv2d: channels/strides must be multiples of 8 (Cin=%d Cout=%d ldx=%lld ldy=%lld)", Cin, Cout,
                 ldx, ldy);
  RIED_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)y % 8) == 0,
                 "ried_conv2d: pointers must be 16-byte aligned");
  RIED_CHECK_ARG(kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0, "ried_conv2d: bad kernel geometry");
  GemmParams P;
  memset(&P, 0, sizeof(P));
  default_epilogue(P.E);
  int phases = 1;
  int Hl = Ho, Wl = Wo;
  if (transposed) {
    RIED_CHECK_ARG(Ho % sh == 0 && Wo % sw == 0, "ried_conv2d: transposed mode needs Ho %% sh == 0");
    // the strided convolution that this transposes must map (Ho, Wo) -> (H, W); covers output_padding
    RIED_CHECK_ARG((Ho + 2 * ph - kh) / sh + 1 == H && (Wo + 2 * pw - kw) / sw + 1 == W && Ho + 2 * ph >= kh &&
                       Wo + 2 * pw >= kw,
                   "ried_conv2d: transposed output size mismatch (H=%d Ho=%d)", H, Ho);
    RIED_CHECK_ARG(kh >= sh && kw >= sw, "ried_conv2d: transposed mode needs kernel >= stride");
    phases = sh * sw;
    Hl = Ho / sh; Wl = Wo / sw;
    P.E.phase_mode = 1;
    P.E.Wo_full = Wo; P.E.HoWo_full = Ho * Wo; P.E.Hq = Hl; P.E.Wq = Wl; P.E.sh = sh; P.E.sw = sw;
  } else {
    RIED_CHECK_ARG(Ho == (H + 2 * ph - kh) / sh + 1 && Wo == (W + 2 * pw - kw) / sw + 1,
                   "ried_conv2d: output size mismatch (H=%d Ho=%d)", H, Ho);
  }
  long long npix = (long long)nimg * Hl * Wl;
  RIED_CHECK_ARG(npix < (1ll << 30) && (long long)nimg * H * W < (1ll << 30), "ried_conv2d: too many pixels");
  P.A.p = (const bf16_t*)w; P.A.ld = (long long)kh * kw * Cin; P.A.rows = Cout;
  P.B.p = (const bf16_t*)x; P.B.ld = ldx; P.B.rows = (int)npix;
  P.B.nimg = nimg; P.B.H = H; P.B.W = W; P.B.Ho = Hl; P.B.Wo = Wl;
  P.B.kh = kh; P.B.kw = kw; P.B.sh = sh; P.B.sw = sw; P.B.ph = ph; P.B.pw = pw;
  P.B.transposed = transposed; P.B.C = Cin;
  P.M = Cout; P.N = (int)npix; P.K = Cin; P.taps = kh * kw; P.splitk = 1;
  P.E.out = y; P.E.ldc = ldy; P.E.out_f32 = out_f32; P.E.bias = bias; P.E.act = act; P.E.stats = stats;
  P.E.accumulate = accumulate;
  RIED_CHECK_ARG(!y2 || (ldy2 % 4 == 0 && ldy2 >= Cout && ((uintptr_t)y2 % 16) == 0), "ried_conv2d: bad y2 view");
  P.E.out2 = y2; P.E.ldc2 = ldy2;
  if (mask_bits) RIED_CHECK_ARG(mask && ldmask >= Cout / 8 && !job && !fin, "ried_conv2d_signs: bad sign-byte view");
  else RIED_CHECK_ARG(!mask || (ldmask % 4 == 0 && ldmask >= Cout && ((uintptr_t)mask % 8) == 0), "ried_conv2d: bad mask view");
  P.E.mask_bits = mask_bits;
  P.E.mask = (const bf16_t*)mask; P.E.ld_mask = ldmask; P.E.mask_slope = mask_slope; P.E.stats_post = stats_post == 1;
  P.E.stats_fold = stats_post == 2 ? ried_bn_fold_rows() : 0;
  const bool img_stats = stats_post == 3;      // one row per image; every kernel below sets its tiles per image
  RIED_CHECK_ARG(!img_stats || (stats && !fin), "ried_conv2d: per-image statistics need a stats buffer");
  RIED_CHECK_ARG(!sub || (stats && gout && ld_sub % 4 == 0 && ld_gout % 4 == 0 && ((uintptr_t)sub % 16) == 0 &&
                          ((uintptr_t)gout % 8) == 0),
                 "ried_conv2d: bad perceptual-tap arguments");
  P.E.sub = sub; P.E.ld_sub = ld_sub; P.E.gout = (bf16_t*)gout; P.E.ld_gout = ld_gout; P.E.gscale = gscale;
  P.E.sub_bf16 = sub_bf16;
  P.E.staged = !fin && !out_f32 && ldy % 8 == 0 && ((uintptr_t)y % 16) == 0 &&
               (!mask || mask_bits || (ldmask % 8 == 0 && ((uintptr_t)mask % 16) == 0)) && (!y2 || ((uintptr_t)y2 % 16) == 0) &&
               (!gout || (ld_gout % 8 == 0 && ((uintptr_t)gout % 16) == 0)) && g_forced_tile[2] != -1;
  RIED_CHECK_ARG(!sub_bf16 || P.E.staged, "ried_conv2d_tap_mse_bf16: needs the staged epilogue (aligned bf16 views)");
  long long a_bytes = (long long)Cout * kh * kw * Cin * 2;
  long long b_bytes = (((long long)nimg * H * W - 1) * ldx + Cin) * 2;
  RIED_CHECK_ARG(a_bytes < (long long)GLDS_OOB - (1 << 28) && b_bytes < (long long)GLDS_OOB,
                 "ried_conv2d: operand larger than 4 GiB");
  P.A_bytes = (unsigned)a_bytes; P.B_bytes = (unsigned)b_bytes;
  ConvTile t = choose_conv_tile(Cout, npix, phases, conv_ksteps(Cin, kh, kw, sh, sw, transposed), kh * kw);
  if (xf) {           // only the general kernel transforms its input (register-staged pixel operand)
    RIED_CHECK_ARG(xf->save && xf->gamma && xf->beta && Cin <= XF_MAX_K && !job && !fin,
                   "ried_conv2d_xf: bad transform arguments (Cin=%d, at most %d)", Cin, XF_MAX_K);
    P.xf_save = xf->save; P.xf_gamma = xf->gamma; P.xf_beta = xf->beta; P.xf_act = xf->act;
  }
  int* const which = job ? job->which : nullptr;
  if (job && !which) {
    RIED_CHECK_ARG(!fin && !img_stats && (!stats || P.E.stats_fold),
                   "ried_conv2d_make_job: statistics must be the folded form (stats_post = 2)");
    t.bm = 64; t.bn = 64; t.nst = 2; t.ks = 1;
    P.tiles_m = ceil_div(Cout, t.bm);
    P.grid[0] = P.tiles_m * ceil_div(npix, t.bn); P.grid[1] = phases; P.grid[2] = 1;
    memcpy(job->rec, &P, sizeof(P));
    *job->nblocks = P.grid[0] * P.grid[1];
    return RIED_OK;
  }
  // 3x3 stride-1 layers on maps of whole 16x16 tiles: the halo-tile kernel (the input tile serves all nine taps)
  if (!xf && g_forced_tile[0] == 0 && P.E.staged && kh == 3 && kw == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 &&
      Cin % 64 == 0 && H % 8 == 0 && W % 8 == 0 && a_bytes < 0xFFFF0000ll - 16384 && b_bytes < 0xFFFFF000ll - 1024 &&
      (nimg % 4 == 0 || (H % 16 == 0 && W % 16 == 0)) && (!img_stats || (H % 16 == 0 && W % 16 == 0))) {
    // 16x16 tiles where the map allows; 8x8 tiles of four images otherwise (64 output channels per block there:
    // the small maps have few pixel tiles, and LDS holds two 400-pixel input tiles beside the weight ring)
    const int ts = (H % 16 == 0 && W % 16 == 0) ? 16 : 8;
    const int bm = (Cout <= 64 || ts == 8 || Cin == 64) ? 64 : 128;
    P.tiles_m = ceil_div(Cout, bm);
    const int ntiles = ceil_div(nimg, 256 / (ts * ts)) * (H / ts) * (W / ts);
    // the caller sized the statistics buffer for the other kernel's rows (npix / t.bn): the spare ones are zeroed
    if (img_stats) P.E.stats_img_tiles = (H / ts) * (W / ts);
    else if (stats && !P.E.stats_fold) { P.E.stats_zero = 256 / t.bn; P.E.stats_rows = ntiles; }
    const dim3 hgrid(P.tiles_m * ntiles);
    if (which) { which[0] = 1; which[1] = bm; which[2] = ts; which[3] = (bm == 64 && Cin == 64 && ts == 16) ? 1 : 2; return RIED_OK; }
    if (ts == 8) hipLaunchKernelGGL((conv_halo_kernel<64, 8>), hgrid, dim3(512), 0, stream, P);
    else if (bm == 64 && Cin == 64) hipLaunchKernelGGL((conv_halo_kernel<64, 16, 1>), hgrid, dim3(512), 0, stream, P);
    else if (bm == 64) hipLaunchKernelGGL((conv_halo_kernel<64, 16>), hgrid, dim3(512), 0, stream, P);
    else hipLaunchKernelGGL((conv_halo_kernel<128, 16>), hgrid, dim3(512), 0, stream, P);
    RIED_CHECK_LAUNCH("ried_conv2d");
    return RIED_OK;
  }
  // 3x3 stride-2 forward on maps of whole 16x16 output tiles: the half-slab halo kernel.  OFF unless asked for
  // (ried_conv2d_force_tile(0, 1, 0): 16 x 16 output tiles, one block per CU; (0, 2, 0): 8 x 16 tiles, two per CU).
  // Measured on the discriminator's layers at 256 images against the general kernel: 64 -> 128 channels at 64 x 64:
  // 89.7 / 77.0 against 78.7 us; 128 -> 256 at 32 x 32: 61.1 / 53.9 against 51.6; 256 -> 512 at 16 x 16: equal.  It
  // moves 1.7-2.6x fewer bytes from L2 to LDS and gains nothing: the first layer reads 134 MB and writes 67 MB - 50 us
  // of HBM time beside 40 us of matrix work that the one- or two-block-per-CU schedules do not overlap - so L2 -> LDS
  // traffic was not what bounds these layers.
  if (!xf && g_forced_tile[0] == 0 && P.E.staged && !transposed && kh == 3 && kw == 3 && sh == 2 && sw == 2 && ph == 1 &&
      pw == 1 && H == 2 * Ho && W == 2 * Wo && Cin % 32 == 0 && Ho % 16 == 0 && Wo % 16 == 0 && Cout >= 64 && !img_stats &&
      a_bytes < 0xFFFF0000ll - 16384 && b_bytes < 0xFFFFF000ll - 1024 && g_forced_tile[2] != 3 &&
      (g_forced_tile[1] == 1 || g_forced_tile[1] == 2)) {      // 1: 16 x 16 output tiles, 2: 8 x 16
    const int bm = Cout >= 128 ? 128 : 64, th = g_forced_tile[1] == 1 ? 16 : 8;
    P.tiles_m = ceil_div(Cout, bm);
    const int ntiles = nimg * (Ho / th) * (Wo / 16);
    // the caller sized the statistics buffer for the general kernel's rows (npix / t.bn): the spare ones are zeroed
    if (stats && !P.E.stats_fold) {
      RIED_CHECK_ARG(th * 16 >= t.bn, "ried_conv2d: statistics rows of the 8 x 16 tiles do not fit the caller's buffer");
      P.E.stats_zero = th * 16 / t.bn; P.E.stats_rows = ntiles;
    }
    const dim3 hgrid(P.tiles_m * ntiles);
    if (which) { which[0] = 4; which[1] = bm; which[2] = th; which[3] = 0; return RIED_OK; }
    if (bm == 128 && th == 16) hipLaunchKernelGGL((conv_halo_s2f_kernel<128, 16>), hgrid, dim3(512), 0, stream, P);
    else if (bm == 128) hipLaunchKernelGGL((conv_halo_s2f_kernel<128, 8>), hgrid, dim3(256), 0, stream, P);
    else if (th == 16) hipLaunchKernelGGL((conv_halo_s2f_kernel<64, 16>), hgrid, dim3(512), 0, stream, P);
    else hipLaunchKernelGGL((conv_halo_s2f_kernel<64, 8>), hgrid, dim3(256), 0, stream, P);
    RIED_CHECK_LAUNCH("ried_conv2d");
    return RIED_OK;
  }
  // data gradients of 3x3 stride-2 layers: all four phases of a tile in one block
  if (!xf && g_forced_tile[0] == 0 && P.E.staged && transposed && kh == 3 && kw == 3 && sh == 2 && sw == 2 && ph == 1 && pw == 1 &&
      Ho == 2 * H && Wo == 2 * W && Cin % 64 == 0 && a_bytes < 0xFFFF0000ll - 16384 && b_bytes < 0xFFFFF000ll - 1024 &&
      H % 16 == 0 && W % 16 == 0) {       // (8x8 maps, four images per block: measured no faster than the general kernel)
    const int ts = 16;
    P.tiles_m = ceil_div(Cout, 64);
    const int ntiles = ceil_div(nimg, 256 / (ts * ts)) * (H / ts) * (W / ts);
    if (img_stats) P.E.stats_img_tiles = (H / 16) * (W / 16);
    else if (stats && !P.E.stats_fold) { P.E.stats_zero = 4 * 256 / t.bn; P.E.stats_rows = ntiles; }   // one row per tile
    const dim3 hgrid(P.tiles_m * ntiles);
    if (which) { which[0] = 2; which[1] = 64; which[2] = 16; which[3] = 0; return RIED_OK; }
    if (fold) {
      RIED_CHECK_ARG(Cout == 64 && fold->x && fold->rows && !accumulate && !bias && !act && !y2 && !sub && !img_stats,
                     "ried_conv2d_s2_dgrad_c1fold: 64 output channels, no accumulate / bias / activation / taps");
      P.E.fold_x = fold->x; P.E.fold_scale = fold->scale; P.E.fold_shift = fold->shift; P.E.fold_rows = fold->rows;
      hipLaunchKernelGGL((conv_halo_s2t_kernel<16, true>), hgrid, dim3(512), 0, stream, P);
    } else {
      hipLaunchKernelGGL(conv_halo_s2t_kernel<16>, hgrid, dim3(512), 0, stream, P);
    }
    RIED_CHECK_LAUNCH("ried_conv2d");
    return RIED_OK;
  }
  RIED_CHECK_ARG(!fold, "ried_conv2d_s2_dgrad_c1fold: needs a 3x3 stride-2 layer on maps of whole 16 x 16 tiles");
  // ConvTranspose2d(4, 2, 1) forward on maps of whole 16x16 tiles: all four phases of a tile in one block
  // (an input transform only in the 8 x 16-tile form for at most 32 output channels)
  if ((!xf || (Cout <= 32 && Cin <= 512 && g_forced_tile[2] != 4)) &&
      g_forced_tile[0] == 0 && P.E.staged && transposed && kh == 4 && kw == 4 && sh == 2 && sw == 2 && ph == 1 && pw == 1 &&
      Ho == 2 * H && Wo == 2 * W && Cin % 32 == 0 && H % 16 == 0 && W % 16 == 0 && a_bytes < 0xFFFF0000ll - 16384 &&
      b_bytes < 0xFFFFF000ll - 1024 &&
      // one block per CU: below a block per CU the general kernel's smaller tiles fill the chip better
      // (ried_conv2d_force_tile(0, 1, 0) lifts this rule: tests reach the kernel with small inputs)
      ((long long)nimg * (H / 16) * (W / 16) * ceil_div(Cout, Cout <= 32 ? 32 : 64) >= 256 || g_forced_tile[1] == 1)) {
    const int bm = Cout <= 32 ? 32 : 64;
    // (tuning hooks: nst = 4: 16 x 16 tiles always; nst = 5: 8 x 16 tiles for 64 output channels too)
    const int tsy = ((bm == 32 || g_forced_tile[2] == 5) && g_forced_tile[2] != 4) ? 8 : 16;
    P.tiles_m = ceil_div(Cout, bm);
    const int ntiles = nimg * (H / tsy) * (W / 16);
    if (img_stats) P.E.stats_img_tiles = (H / tsy) * (W / 16);
    else if (stats && !P.E.stats_fold) { P.E.stats_zero = 4 * tsy * 16 / t.bn; P.E.stats_rows = ntiles; }   // one row per tile
    const dim3 hgrid(P.tiles_m * ntiles);
    if (which) { which[0] = 3; which[1] = bm; which[2] = tsy; which[3] = 0; return RIED_OK; }
    if (bm == 32 && tsy == 8 && xf) hipLaunchKernelGGL((conv_halo_d4_kernel<32, 8, true>), hgrid, dim3(512), 0, stream, P);
    else if (bm == 32 && tsy == 8) hipLaunchKernelGGL((conv_halo_d4_kernel<32, 8>), hgrid, dim3(512), 0, stream, P);
    else if (bm == 32) hipLaunchKernelGGL((conv_halo_d4_kernel<32, 16>), hgrid, dim3(512), 0, stream, P);
    else if (tsy == 8) hipLaunchKernelGGL((conv_halo_d4_kernel<64, 8>), hgrid, dim3(512), 0, stream, P);
    else hipLaunchKernelGGL((conv_halo_d4_kernel<64, 16>), hgrid, dim3(512), 0, stream, P);
    RIED_CHECK_LAUNCH("ried_conv2d");
    return RIED_OK;
  }
  if (img_stats) {
    RIED_CHECK_ARG((Hl * Wl) % t.bn == 0, "ried_conv2d: per-image statistics need maps of whole %d-pixel tiles (%d x %d)",
                   t.bn, Hl, Wl);
    P.E.stats_img_tiles = (Hl * Wl) / t.bn;
  }
  P.tiles_m = ceil_div(Cout, t.bm);
  dim3 grid(P.tiles_m * ceil_div(npix, t.bn), phases, 1);
  if (which) { which[0] = 0; which[1] = t.bm; which[2] = t.bn; which[3] = t.nst + 10 * (t.ks - 1); return RIED_OK; }
  if (fin) {
    P.E.fin = *fin;
    P.E.fin.nrows = phases * ceil_div(npix, t.bn);
  }
#define RIED_CONV_CASE(BM_, BN_, NST_)                                                                  \
  if (t.bm == BM_ && t.bn == BN_ && t.nst == NST_) {                                                    \
    hipLaunchKernelGGL((conv_glds_kernel<BM_, BN_, NST_>), grid, dim3(256), 0, stream, P);              \
    launched = true;                                                                                    \
  }
#define RIED_CONV_CASE8(BM_, BN_, NST_)                                                                 \
  if (t.bm == BM_ && t.bn == BN_ && t.nst == NST_) {                                                    \
    hipLaunchKernelGGL((conv_glds_kernel<BM_, BN_, NST_, 512>), grid, dim3(512), 0, stream, P);         \
    launched = true;                                                                                    \
  }
  bool launched = false;
  if (xf) {
#define RIED_XF_CASE(BM_, BN_, NT_)                                                                      \
    if (t.bm == BM_ && t.bn == BN_) {                                                                    \
      hipLaunchKernelGGL((conv_glds_xf_kernel<BM_, BN_, NT_>), grid, dim3(NT_), 0, stream, P);            \
      launched = true;                                                                                   \
    }
    RIED_XF_CASE(128, 256, 512) RIED_XF_CASE(64, 256, 512) RIED_XF_CASE(128, 128, 256) RIED_XF_CASE(128, 64, 256)
    RIED_XF_CASE(64, 128, 256) RIED_XF_CASE(64, 64, 256) RIED_XF_CASE(32, 128, 256) RIED_XF_CASE(32, 64, 256)
#undef RIED_XF_CASE
    RIED_CHECK_ARG(launched, "ried_conv2d_xf: no kernel for tile %dx%d", t.bm, t.bn);
    RIED_CHECK_LAUNCH("ried_conv2d_xf");
    return RIED_OK;
  }
  if (t.ks == 2) {
    RIED_CHECK_ARG(t.bm == 64 && t.bn == 64 && (t.nst == 2 || t.nst == 3 || t.nst == 4),
                   "ried_conv2d: two K groups only with 64 x 64 tiles");
    if (t.nst == 2) hipLaunchKernelGGL((conv_glds_kernel<64, 64, 2, 256, 2>), grid, dim3(512), 0, stream, P);
    else if (t.nst == 3) hipLaunchKernelGGL((conv_glds_kernel<64, 64, 3, 256, 2>), grid, dim3(512), 0, stream, P);
    else hipLaunchKernelGGL((conv_glds_kernel<64, 64, 4, 256, 2>), grid, dim3(512), 0, stream, P);
    RIED_CHECK_LAUNCH("ried_conv2d");
    return RIED_OK;
  }
  RIED_CONV_CASE8(128, 256, 2) RIED_CONV_CASE8(128, 256, 3) RIED_CONV_CASE8(64, 256, 2) RIED_CONV_CASE8(64, 256, 3)
  RIED_CONV_CASE(128, 128, 2) RIED_CONV_CASE(128, 128, 3) RIED_CONV_CASE(128, 64, 2) RIED_CONV_CASE(128, 64, 3)
  RIED_CONV_CASE(128, 64, 4) RIED_CONV_CASE(64, 128, 2) RIED_CONV_CASE(64, 128, 3) RIED_CONV_CASE(64, 128, 4)
  RIED_CONV_CASE(64, 64, 2) RIED_CONV_CASE(64, 64, 3) RIED_CONV_CASE(64, 64, 4) RIED_CONV_CASE(64, 64, 6)
  RIED_CONV_CASE(64, 64, 8)
  RIED_CONV_CASE(32, 128, 2) RIED_CONV_CASE(32, 64, 2)
#undef RIED_CONV_CASE
#undef RIED_CONV_CASE8
  RIED_CHECK_ARG(launched, "ried_conv2d: no kernel for tile %dx%dx%d", t.bm, t.bn, t.nst);
  RIED_CHECK_LAUNCH("ried_conv2d");
  return RIED_OK;
}

extern "C" int ried_conv2d(const void* x, long long ldx, const void* w, void* y, long long ldy, int nimg, int H, int W,
                           int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                           int transposed, const float* bias, int act, float* stats, int accumulate, int out_f32,
                           float* y2, long long ldy2, const void* mask, long long ldmask, float mask_slope,
                           int stats_post, hipStream_t stream) {
  return conv2d_impl(x, ldx, w, y, ldy, nimg, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, ph, pw, transposed, bias, act,
                     stats, accumulate, out_f32, y2, ldy2, mask, ldmask, mask_slope, stats_post, nullptr, stream);
}

// ried_conv2d with the activation mask given as sign bytes (ried_conv_c1_fwd_signs): mask[p][ldmask bytes], bit q of
// byte c set where channel 8c + q of the forward output was positive.
extern "C" int ried_conv2d_signs(const void* x, long long ldx, const void* w, void* y, long long ldy, int nimg, int H,
                                 int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                                 int transposed, const float* bias, int act, float* stats, int accumulate,
                                 const unsigned char* signs, long long ldsigns, float mask_slope, int stats_post,
                                 hipStream_t stream) {
  return conv2d_impl(x, ldx, w, y, ldy, nimg, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, ph, pw, transposed, bias, act,
                     stats, accumulate, 0, nullptr, 0, signs, ldsigns, mask_slope, stats_post, nullptr, stream, nullptr, 0,
                     nullptr, 0, 0.f, nullptr, nullptr, 1);
}

// Backward of the discriminator's second layer into its first (src/rinet.py:20-21) when only parameter gradients are
// wanted: the data gradient of the 3x3 stride-2 convolution (dy: [nimg][H][W] with Cin channels, wt: its packed
// transposed weights [64][3][3][Cin]) is masked with the first layer's LeakyReLU (mask: its output as bf16 view, or
// its sign bytes when mask_bits = 1), its column sums go to stats (the first layer's bias gradient, stats_post = 1
// rows) and its product with the nine taps of the first layer's fp32 input ximg [nimg][2H][2W] (used as
// ximg * in_scale + in_shift, zero padded) is ADDED to dw [64][9] - the first layer's weight gradient - without the
// masked gradient ever being written.  H, W multiples of 16; scratch: fp32 [ried_conv2d_s2_dgrad_c1fold_scratch_floats].
extern "C" long long ried_conv2d_s2_dgrad_c1fold_scratch_floats(int nimg, int H, int W) {
  return (long long)nimg * (H / 16) * (W / 16) * 64 * 9;
}
extern "C" int ried_conv2d_s2_dgrad_c1fold(const void* dy, long long lddy, const void* wt, int nimg, int H, int W,
                                           int Cin, const void* mask, long long ldmask, int mask_bits,
                                           float mask_slope, float* stats, const float* ximg, float in_scale,
                                           float in_shift, float* dw, float* scratch, hipStream_t stream) {
  RIED_CHECK_ARG(ximg && dw && scratch && mask && H % 16 == 0 && W % 16 == 0, "ried_conv2d_s2_dgrad_c1fold: bad arguments");
  ConvFold f{ximg, in_scale, in_shift, scratch};
  int rc = conv2d_impl(dy, lddy, wt, nullptr, 64, nimg, H, W, Cin, 2 * H, 2 * W, 64, 3, 3, 2, 2, 1, 1, 1, nullptr, 0,
                       stats, 0, 0, nullptr, 0, mask, ldmask, mask_slope, 1, nullptr, stream, nullptr, 0, nullptr, 0, 0.f,
                       nullptr, nullptr, mask_bits, &f);
  if (rc != RIED_OK) return rc;
  return ried_rows_add(scratch, nimg * (H / 16) * (W / 16), 64 * 9, dw, stream);
}

// ried_conv2d whose input x is the RAW convolution output of a BasicConv2d layer (src/rinet.py:183-205): the kernel
// applies that layer's train-mode BatchNorm (xf_save = [2][Cin] mean, 1/sqrt(var + eps); xf_gamma, xf_beta) and its
// activation (xf_act: 0 none, 1 ReLU) to the operand on its way through LDS, so that the normalised tensor is never
// written or read - one pass over every activation less.  Always the general kernel (Cin <= 1024).
extern "C" int ried_conv2d_xf(const void* x, long long ldx, const void* w, void* y, long long ldy, int nimg, int H, int W,
                              int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                              int transposed, const float* bias, int act, float* stats, int accumulate, int stats_post,
                              const float* xf_save, const float* xf_gamma, const float* xf_beta, int xf_act,
                              hipStream_t stream) {
  ConvXf xf{xf_save, xf_gamma, xf_beta, xf_act};
  return conv2d_impl(x, ldx, w, y, ldy, nimg, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, ph, pw, transposed, bias, act,
                     stats, accumulate, 0, nullptr, 0, nullptr, 0, 0.f, stats_post, nullptr, stream, nullptr, 0,
                     nullptr, 0, 0.f, nullptr, &xf);
}

// Job-table form of ried_conv2d: convolutions that do not depend on each other in one launch of the general kernel
// (64 x 64 tiles).  ried_conv2d_make_job() fills a HOST record (ried_conv2d_job_bytes() bytes; arguments as
// ried_conv2d, statistics only in the folded form stats_post = 2) and reports the job's block count; the records are
// copied to the device back to back and launched with ried_conv2d_jobs, map[2b], map[2b+1] = (job, block of the job).
extern "C" int ried_conv2d_job_bytes(void) { return (int)sizeof(GemmParams); }
extern "C" int ried_conv2d_make_job(void* job_host, const void* x, long long ldx, const void* w, void* y, long long ldy,
                                    int nimg, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh,
                                    int sw, int ph, int pw, int transposed, const float* bias, int act, float* stats,
                                    int accumulate, int stats_post, int* nblocks) {
  RIED_CHECK_ARG(job_host && nblocks, "ried_conv2d_make_job: null pointer");
  ConvJobOut job{job_host, nblocks, nullptr};
  return conv2d_impl(x, ldx, w, y, ldy, nimg, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, ph, pw, transposed, bias, act,
                     stats, accumulate, 0, nullptr, 0, nullptr, 0, 0.f, stats_post, nullptr, nullptr, nullptr, 0,
                     nullptr, 0, 0.f, &job);
}
// Which kernel a ried_conv2d call with these arguments runs (for reports): which[4] = {kind, bm, bn or tile side,
// LDS stages (+ 10: two K groups per block)}; kind 0 = conv_glds_kernel<bm, bn, stages>, 1 = conv_halo_kernel<bm, side, stages>,
// 2 = conv_halo_s2t_kernel<side>, 3 = conv_halo_d4_kernel<bm>.  Nothing is launched.
extern "C" int ried_conv2d_which(const void* x, long long ldx, const void* w, void* y, long long ldy, int nimg, int H,
                                 int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw, int ph,
                                 int pw, int transposed, const float* bias, int act, float* stats, int accumulate,
                                 int out_f32, float* y2, long long ldy2, const void* mask, long long ldmask,
                                 float mask_slope, int stats_post, int* which) {
  RIED_CHECK_ARG(which != nullptr, "ried_conv2d_which: null pointer");
  ConvJobOut q{nullptr, nullptr, which};
  return conv2d_impl(x, ldx, w, y, ldy, nimg, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, ph, pw, transposed, bias, act,
                     stats, accumulate, out_f32, y2, ldy2, mask, ldmask, mask_slope, stats_post, nullptr, nullptr,
                     nullptr, 0, nullptr, 0, 0.f, &q);
}
extern "C" int ried_conv2d_jobs(const void* jobs, const int* map, int nblocks, hipStream_t stream) {
  RIED_CHECK_ARG(jobs && map && nblocks > 0, "ried_conv2d_jobs: bad arguments");
  hipLaunchKernelGGL((conv_glds_jobs_kernel<64, 64, 2>), dim3(nblocks), dim3(256), 0, stream,
                     (const GemmParams*)jobs, map);
  RIED_CHECK_LAUNCH("ried_conv2d_jobs");
  return RIED_OK;
}

// Convolution (or transposed convolution) of a BasicConv2d / BasicDeconv2d block (src/rinet.py:183-205) with the
// train-mode BatchNorm statistics fused: y = raw bf16 result; stats = partial rows as in ried_conv2d; the last
// block to finish reduces them into save = (mean, 1/sqrt(var+eps)) and updates the running statistics.
// counter: one zero-initialised unsigned per layer (left at zero).
extern "C" int ried_conv2d_bn(const void* x, long long ldx, const void* w, void* y, long long ldy, int nimg, int H,
                              int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                              int transposed, float* stats, unsigned* counter, float eps, float momentum, float* save,
                              float* running_mean, float* running_var, long long* num_batches_tracked,
                              hipStream_t stream) {
  RIED_CHECK_ARG(stats && counter && save, "ried_conv2d_bn: null pointer");
  BnFinalize f;
  f.counter = counter; f.save = save; f.running_mean = running_mean; f.running_var = running_var;
  f.nbt = num_batches_tracked; f.count = (float)((long long)nimg * Ho * Wo); f.eps = eps; f.momentum = momentum;
  f.nrows = 0;
  return conv2d_impl(x, ldx, w, y, ldy, nimg, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, ph, pw, transposed, nullptr, 0,
                     stats, 0, 0, nullptr, 0, nullptr, 0, 0.f, 0, &f, stream);
}

// Taps per block of the weight-gradient kernel (wgrad_glds_body's NT), for layers with at most 64 gathered channels.  ried_conv2d_wgrad_row_blocks(0) turns that off for later calls (A/B runs, tests); plans
// built before keep the split counts and job tables they were built with.
static int g_wgrad_rows = 1;
extern "C" int ried_conv2d_wgrad_row_blocks(int on) { g_wgrad_rows = on != 0; return RIED_OK; }
// (Measured, 3 x 3 layers at 256 / 128 images: 64 -> 128 channels stride 2 on 64 x 64 maps 86 -> 63 us, 64 -> 64 stride
// 1 on 64 x 64 maps 102 -> 64 us; 48 -> 48 on 16 x 16 and 32 -> 64 stride 2 on 32 x 32 maps 11 -> 16 us: with few
// pixels the kernel row's three-fold work per block leaves too few blocks.  Hence the rule below.)
// The rule: a 3 x 3 layer's kernel row (3 taps) or half a 4 x 4 layer's (2 taps) per block when, at the 384 blocks a
// launch wants, every block still walks at least 1024 pixels.
static int wgrad_taps_per_block(int Cg, int Cp, int kh, int kw, long long npix) {
  if (Cg > 64 || !g_wgrad_rows || g_forced_tile[0] == -3) return 1;
  const int nt = (kh == 3 && kw == 3) ? 3 : (kh == 4 && kw == 4) ? 2 : 1;
  if (nt == 1) return 1;
  const long long per_split = (long long)ceil_div(Cp, Cp <= 64 ? 64 : 128) * (kh * kw / nt);
  return npix * per_split >= 384ll * 1024 ? nt : 1;
}
extern "C" int ried_conv2d_wgrad_taps_per_block(int Cg, int Cp, int kh, int kw, long long npix) {
  return wgrad_taps_per_block(Cg, Cp, kh, kw, npix);
}

static int wgrad_params(GemmParams& P, const void* g, long long ldg, int Cg, int nimg, int H, int W, const void* p,
                        long long ldp, int Cp, int Ho, int Wo, int kh, int kw, int sh, int sw, int ph, int pw,
                        float* slab, int splitk, int& bm, int& bn, int& nt) {
  RIED_CHECK_ARG(g && p && slab, "ried_conv2d_wgrad: null pointer");
  RIED_CHECK_ARG(Cg % 8 == 0 && Cp % 8 == 0 && ldg % 8 == 0 && ldp % 8 == 0, "ried_conv2d_wgrad: channels %% 8");
  RIED_CHECK_ARG(((uintptr_t)g % 16) == 0 && ((uintptr_t)p % 16) == 0 && ((uintptr_t)slab % 16) == 0,
                 "ried_conv2d_wgrad: alignment");
  RIED_CHECK_ARG(Ho == (H + 2 * ph - kh) / sh + 1 && Wo == (W + 2 * pw - kw) / sw + 1,
                 "ried_conv2d_wgrad: output size mismatch");
  RIED_CHECK_ARG(splitk >= 1 && splitk <= 65535, "ried_conv2d_wgrad: bad splitk");
  long long npix = (long long)nimg * Ho * Wo;
  RIED_CHECK_ARG(npix < (1ll << 30) && (long long)nimg * H * W < (1ll << 30), "ried_conv2d_wgrad: too many pixels");
  memset(&P, 0, sizeof(P));
  default_epilogue(P.E);
  P.A.p = (const bf16_t*)g; P.A.ld = ldg; P.A.rows = Cg; P.A.C = Cg;
  P.A.nimg = nimg; P.A.H = H; P.A.W = W; P.A.Ho = Ho; P.A.Wo = Wo;
  P.A.kh = kh; P.A.kw = kw; P.A.sh = sh; P.A.sw = sw; P.A.ph = ph; P.A.pw = pw;
  P.B.p = (const bf16_t*)p; P.B.ld = ldp; P.B.rows = Cp;
  P.M = Cg; P.N = Cp; P.K = (int)npix; P.taps = 1; P.splitk = splitk;
  P.E.out = slab; P.E.ldc = Cg; P.E.out_f32 = 1;
  P.E.tap_stride = (long long)Cp * Cg;
  P.E.batch_stride = (long long)kh * kw * Cp * Cg;      // per split
  bm = Cg <= 64 ? 64 : 128; bn = Cp <= 64 ? 64 : 128;    // ried_conv2d_wgrad_tiles mirrors this
  nt = wgrad_taps_per_block(Cg, Cp, kh, kw, npix);
  P.tiles_m = ceil_div(Cg, bm);
  P.grid[0] = P.tiles_m * ceil_div(Cp, bn); P.grid[1] = kh * kw / nt; P.grid[2] = splitk;
  const long long a_bytes = (((long long)nimg * H * W - 1) * ldg + Cg) * 2, b_bytes = ((npix - 1) * ldp + Cp) * 2;
  RIED_CHECK_ARG(a_bytes < (long long)GLDS_OOB && b_bytes < (long long)GLDS_OOB, "ried_conv2d_wgrad: operand larger than 4 GiB");
  P.A_bytes = (unsigned)a_bytes; P.B_bytes = (unsigned)b_bytes;
  return RIED_OK;
}

// A VGG16 tap layer of the perceptual loss on the PREDICTION images (src/loss.py:179-189): y = relu(conv3x3(x) + bias)
// as usual, and in the same pass d = y(fp32, before rounding) - target (the same tap of the target images, fp32
// [npix][ld_target]); stats rows [rows][2][Cout] receive the column sums of d and d*d (rows as for ried_conv2d:
// sum of the second halves / count = the tap's MSE), and gout [npix][ld_gout] = gscale * d, the seed of the tap's
// gradient.  No fp32 copy of the prediction's features is written or re-read.
extern "C" int ried_conv2d_tap_mse(const void* x, long long ldx, const void* w, const float* bias, void* y,
                                   long long ldy, int nimg, int H, int W, int Cin, int Cout, const float* target,
                                   long long ld_target, void* gout, long long ld_gout, float gscale, float* stats,
                                   hipStream_t stream) {
  RIED_CHECK_ARG(target && gout && stats, "ried_conv2d_tap_mse: null pointer");
  return conv2d_impl(x, ldx, w, y, ldy, nimg, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1, 0, bias, 1, stats, 0, 0,
                     nullptr, 0, nullptr, 0, 0.f, 0, nullptr, stream, target, ld_target, gout, ld_gout, gscale);
}

// The same with the target's tap given as its stored bf16 activations (a view: ld_target elements per pixel) instead of
// an fp32 copy: the target pass then writes no fp32 taps and this pass reads half the bytes.  The difference is taken
// against the rounded target values (relative error 2^-9 per value, the error every other activation already carries).
extern "C" int ried_conv2d_tap_mse_bf16(const void* x, long long ldx, const void* w, const float* bias, void* y,
                                        long long ldy, int nimg, int H, int W, int Cin, int Cout, const void* target,
                                        long long ld_target, void* gout, long long ld_gout, float gscale, float* stats,
                                        hipStream_t stream) {
  RIED_CHECK_ARG(target && gout && stats && ld_target % 8 == 0 && ld_target >= Cout && ((uintptr_t)target % 16) == 0 &&
                     ldy % 8 == 0 && ((uintptr_t)y % 16) == 0 && ld_gout % 8 == 0 && ((uintptr_t)gout % 16) == 0,
                 "ried_conv2d_tap_mse_bf16: null pointer or unaligned view");
  return conv2d_impl(x, ldx, w, y, ldy, nimg, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1, 0, bias, 1, stats, 0, 0,
                     nullptr, 0, nullptr, 0, 0.f, 0, nullptr, stream, (const float*)target, ld_target, gout, ld_gout, gscale,
                     nullptr, nullptr, 0, nullptr, 1);
}

// Weight gradient partial sums.  g: the tensor that is gathered through the kernel window
// (the conv input; for a ConvTranspose2d layer, the output gradient), NHWC bf16 [nimg][H][W][ldg], Cg channels.
// p: the tensor indexed by output pixel (conv: output gradient; ConvTranspose2d: the layer input),
// NHWC bf16 [nimg][Ho][Wo][ldp], Cp channels.   slab: fp32 [splitk][kh*kw][Cp][Cg].
extern "C" int ried_conv2d_wgrad(const void* g, long long ldg, int Cg, int nimg, int H, int W, const void* p,
                                 long long ldp, int Cp, int Ho, int Wo, int kh, int kw, int sh, int sw, int ph, int pw,
                                 float* slab, int splitk, hipStream_t stream) {
  GemmParams P;
  int bm, bn, nt;
  int rc = wgrad_params(P, g, ldg, Cg, nimg, H, W, p, ldp, Cp, Ho, Wo, kh, kw, sh, sw, ph, pw, slab, splitk, bm, bn, nt);
  if (rc) return rc;
  dim3 grid(P.grid[0], P.grid[1], P.grid[2]);
  if (nt == 3) {                             // a kernel row per block (bm = 64)
    if (bn == 64) hipLaunchKernelGGL((wgrad_glds_kernel<64, 64, 2, 3>), grid, dim3(256), 0, stream, P);
    else hipLaunchKernelGGL((wgrad_glds_kernel<64, 128, 2, 3>), grid, dim3(256), 0, stream, P);
  } else if (nt == 2) {
    if (bn == 64) hipLaunchKernelGGL((wgrad_glds_kernel<64, 64, 2, 2>), grid, dim3(256), 0, stream, P);
    else hipLaunchKernelGGL((wgrad_glds_kernel<64, 128, 2, 2>), grid, dim3(256), 0, stream, P);
  } else if (g_forced_tile[0] == -3) {       // tuning hook: the register-staged form
    if (bm == 64 && bn == 64) launch<64, 64, OP_GATHER_TR, OP_PLAIN_TR>(P, grid, stream);
    else if (bm == 64) launch<64, 128, OP_GATHER_TR, OP_PLAIN_TR>(P, grid, stream);
    else if (bn == 64) launch<128, 64, OP_GATHER_TR, OP_PLAIN_TR>(P, grid, stream);
    else launch<128, 128, OP_GATHER_TR, OP_PLAIN_TR>(P, grid, stream);
  } else {
    if (bm == 64 && bn == 64) hipLaunchKernelGGL((wgrad_glds_kernel<64, 64, 2>), grid, dim3(256), 0, stream, P);
    else if (bm == 64) hipLaunchKernelGGL((wgrad_glds_kernel<64, 128, 2>), grid, dim3(256), 0, stream, P);
    else if (bn == 64) hipLaunchKernelGGL((wgrad_glds_kernel<128, 64, 2>), grid, dim3(256), 0, stream, P);
    else hipLaunchKernelGGL((wgrad_glds_kernel<128, 128, 2>), grid, dim3(256), 0, stream, P);
  }
  RIED_CHECK_LAUNCH("ried_conv2d_wgrad");
  return RIED_OK;
}

// Job-table form of ried_conv2d_wgrad: the weight gradients of many layers in one launch per tile shape.
//   ried_conv2d_wgrad_job_bytes(): size of one job record.
//   ried_conv2d_wgrad_make_job(): fills a HOST record for one layer (same arguments as ried_conv2d_wgrad) and
//     reports its tile shape (tile[0] x tile[1]) and block count; records of one tile shape are copied to the
//     device back to back and launched with ried_conv2d_wgrad_jobs, map[b] = (job, block of that job).
extern "C" int ried_conv2d_wgrad_job_bytes(void) { return (int)sizeof(GemmParams); }

extern "C" int ried_conv2d_wgrad_make_job(void* job_host, const void* g, long long ldg, int Cg, int nimg, int H, int W,
                                          const void* p, long long ldp, int Cp, int Ho, int Wo, int kh, int kw, int sh,
                                          int sw, int ph, int pw, float* slab, int splitk, int* tile, int* nblocks) {
  RIED_CHECK_ARG(job_host && tile && nblocks, "ried_conv2d_wgrad_make_job: null pointer");
  GemmParams P;
  int nt;
  int rc = wgrad_params(P, g, ldg, Cg, nimg, H, W, p, ldp, Cp, Ho, Wo, kh, kw, sh, sw, ph, pw, slab, splitk, tile[0],
                        tile[1], nt);
  if (rc) return rc;
  tile[1] |= (nt - 1) << 16;        // jobs of one launch share the taps per block too: part of the tile key
  memcpy(job_host, &P, sizeof(P));
  *nblocks = P.grid[0] * P.grid[1] * P.grid[2];
  return RIED_OK;
}

extern "C" int ried_conv2d_wgrad_jobs(const void* jobs, const int* map, int nblocks, int bm, int bn,
                                      hipStream_t stream) {
  RIED_CHECK_ARG(jobs && map && nblocks > 0, "ried_conv2d_wgrad_jobs: bad arguments");
  const GemmParams* J = (const GemmParams*)jobs;
  const int nt = (bn >> 16) + 1;          // taps per block of these jobs (ried_conv2d_wgrad_make_job's tile key)
  bn &= 0xFFFF;
  if (nt > 1 && nt <= 3 && bm == 64 && (bn == 64 || bn == 128)) {
    if (nt == 3 && bn == 64) hipLaunchKernelGGL((wgrad_glds_jobs_kernel<64, 64, 2, 3>), dim3(nblocks), dim3(256), 0, stream, J, map);
    else if (nt == 3) hipLaunchKernelGGL((wgrad_glds_jobs_kernel<64, 128, 2, 3>), dim3(nblocks), dim3(256), 0, stream, J, map);
    else if (bn == 64) hipLaunchKernelGGL((wgrad_glds_jobs_kernel<64, 64, 2, 2>), dim3(nblocks), dim3(256), 0, stream, J, map);
    else hipLaunchKernelGGL((wgrad_glds_jobs_kernel<64, 128, 2, 2>), dim3(nblocks), dim3(256), 0, stream, J, map);
    RIED_CHECK_LAUNCH("ried_conv2d_wgrad_jobs");
    return RIED_OK;
  }
  RIED_CHECK_ARG(nt == 1, "ried_conv2d_wgrad_jobs: no kernel for %d taps per block at tile %dx%d", nt, bm, bn);
#define RIED_WJ(BM_, BN_)                                                                                  \
  if (bm == BM_ && bn == BN_) {                                                                            \
    if (g_forced_tile[0] == -3)                                                                            \
      hipLaunchKernelGGL((igemm_jobs_kernel<BM_, BN_, OP_GATHER_TR, OP_PLAIN_TR>), dim3(nblocks), dim3(256), 0, \
                         stream, J, map);                                                                  \
    else                                                                                                   \
      hipLaunchKernelGGL((wgrad_glds_jobs_kernel<BM_, BN_, 2>), dim3(nblocks), dim3(256), 0, stream, J, map); \
    RIED_CHECK_LAUNCH("ried_conv2d_wgrad_jobs");                                                           \
    return RIED_OK;                                                                                        \
  }
  RIED_WJ(64, 64) RIED_WJ(64, 128) RIED_WJ(128, 64) RIED_WJ(128, 128)
#undef RIED_WJ
  ried_set_error("ried_conv2d_wgrad_jobs: no kernel for tile %dx%d", bm, bn);
  return RIED_ERR_ARG;
}

// output tiles (blocks per tap and split) of ried_conv2d_wgrad for Cg x Cp: for choosing splitk
extern "C" int ried_conv2d_wgrad_tiles(int Cg, int Cp) {
  return ceil_div(Cg, Cg <= 64 ? 64 : 128) * ceil_div(Cp, Cp <= 64 ? 64 : 128);
}

// Batched out[b][n][m] = alpha * sum_k A[b][m][k] B[b][n][k]  (+ gamma*value + resid, or accumulate).
// a_tr / b_tr: the operand is stored [k][row] instead of [row][k].
extern "C" int ried_bgemm(const void* A, long long lda, long long bsa, int a_tr, const void* B, long long ldb,
                          long long bsb, int b_tr, int M, int N, int K, int batch, void* out, long long ldc,
                          long long bsc, int out_f32, float alpha, const float* gamma, const void* resid,
                          long long ld_resid, long long bs_resid, int accumulate, hipStream_t stream) {
  RIED_CHECK_ARG(A && B && out, "ried_bgemm: null pointer");
  RIED_CHECK_ARG(M % 8 == 0 && N % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0,
                 "ried_bgemm: M, N, K and strides must be multiples of 8 (M=%d N=%d K=%d)", M, N, K);
  RIED_CHECK_ARG(bsa % 8 == 0 && bsb % 8 == 0 && bsc % 4 == 0, "ried_bgemm: batch strides must be multiples of 8");
  RIED_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)out % 16) == 0,
                 "ried_bgemm: alignment");
  RIED_CHECK_ARG(batch >= 1 && batch <= 65535, "ried_bgemm: bad batch");
  RIED_CHECK_ARG(!gamma || resid, "ried_bgemm: gamma needs resid");
  GemmParams P;
  memset(&P, 0, sizeof(P));
  default_epilogue(P.E);
  P.A.p = (const bf16_t*)A; P.A.ld = lda; P.A.batch_stride = bsa; P.A.rows = M;
  P.B.p = (const bf16_t*)B; P.B.ld = ldb; P.B.batch_stride = bsb; P.B.rows = N;
  P.M = M; P.N = N; P.K = K; P.taps = 1; P.splitk = 1;
  P.E.out = out; P.E.ldc = ldc; P.E.batch_stride = bsc; P.E.out_f32 = out_f32; P.E.alpha = alpha;
  P.E.gamma = gamma; P.E.resid = (const bf16_t*)resid; P.E.ld_resid = ld_resid; P.E.resid_batch_stride = bs_resid;
  P.E.accumulate = accumulate;
  bool small = (M <= 64);
  int bm = small ? 64 : 128;
  P.tiles_m = ceil_div(M, bm);
  dim3 grid(P.tiles_m * ceil_div(N, 128), 1, batch);
  if (small) {
    if (!a_tr && !b_tr) launch<64, 128, OP_PLAIN_K, OP_PLAIN_K>(P, grid, stream);
    else if (a_tr && !b_tr) launch<64, 128, OP_PLAIN_TR, OP_PLAIN_K>(P, grid, stream);
    else if (a_tr && b_tr) launch<64, 128, OP_PLAIN_TR, OP_PLAIN_TR>(P, grid, stream);
    else launch<64, 128, OP_PLAIN_K, OP_PLAIN_TR>(P, grid, stream);
  } else {
    if (!a_tr && !b_tr) launch<128, 128, OP_PLAIN_K, OP_PLAIN_K>(P, grid, stream);
    else if (a_tr && !b_tr) launch<128, 128, OP_PLAIN_TR, OP_PLAIN_K>(P, grid, stream);
    else if (a_tr && b_tr) launch<128, 128, OP_PLAIN_TR, OP_PLAIN_TR>(P, grid, stream);
    else launch<128, 128, OP_PLAIN_K, OP_PLAIN_TR>(P, grid, stream);
  }
  RIED_CHECK_LAUNCH("ried_bgemm");
  return RIED_OK;
}
