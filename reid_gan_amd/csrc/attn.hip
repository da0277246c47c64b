// Fused self-attention for the discriminator's Self_Attn layers (reference: src/rinet.py:162-181):
//   energy = q^T k, attention = softmax(energy, dim=-1), out = gamma * (v attention^T) + x
// forward and backward, one image per workgroup (or per 64 queries of it), nothing n x n ever written to HBM.
// The reference materialises energy and attention with two torch.bmm calls and a softmax (and autograd keeps both).
//
// MFMA v_mfma_f32_16x16x32_bf16, D[i][j] += sum_k A[i][k] B[k][j], 64 lanes, g = lane >> 4, l = lane & 15:
//   A operand  lane supplies A[i = l][k = 8g .. 8g+7]
//   B operand  lane supplies B[k = 8g .. 8g+7][j = l]
//   D          lane holds    D[i = 4g + r][j = l], r = 0..3
// The order of k inside a 32-slice is free as long as both operands agree.  "Paired" order: lane group g supplies
// k = {4g .. 4g+3, 16+4g .. 16+4g+3} - exactly what a lane holds, as D, of two adjacent 16-row tiles.  A result can
// therefore be fed back as a B operand (contraction over its i index) without leaving registers; the matching A
// operand comes out of LDS through ds_read_b64_tr_b16 with its k rows addressed in the same order.
#include "common.h"

typedef __attribute__((address_space(3))) s16x4 attn_lds_s16x4;

__device__ __forceinline__ f32x4 mfma16(bf16x8_t a, bf16x8_t b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8_t ld_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8_t*>(p); }
__device__ __forceinline__ bf16x8_t pack_pair(const f32x4& lo, const f32x4& hi) {
  u32x4 u = {pack2bf(lo[0], lo[1]), pack2bf(lo[2], lo[3]), pack2bf(hi[0], hi[1]), pack2bf(hi[2], hi[3])};
  return __builtin_bit_cast(bf16x8_t, u);
}
// XOR swizzle (in elements, a multiple of 16) of k row k of an LDS image [k][row] with at least 128 elements per row
__device__ __forceinline__ int attn_swz(int k) { return ((((k >> 3) & 1) << 2) | (k & 3)) << 4; }

// A fragment (operand rows rowbase .. +15, 32 k rows from kbase) out of an LDS image stored [k][row]: per 16-lane group
// lane 4q+p supplies the address of k row q, operand rows 4p .. 4p+3; lane i receives operand row i for the 4 k rows.
// PAIRED: k rows kbase + {4g+q, 16+4g+q}; else the standard order kbase + {8g+q, 8g+4+q}.
template <bool SWZ, bool PAIRED>
__device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* lds, int pitch, int kbase, int rowbase, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int k0 = kbase + (PAIRED ? 4 * g : 8 * g) + q, k1 = k0 + (PAIRED ? 16 : 4);
  const bf16_t* a0 = lds + k0 * pitch + ((rowbase + 4 * p) ^ (SWZ ? attn_swz(k0) : 0));
  const bf16_t* a1 = lds + k1 * pitch + ((rowbase + 4 * p) ^ (SWZ ? attn_swz(k1) : 0));
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((attn_lds_s16x4*)a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((attn_lds_s16x4*)a1);
  s16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, r);
}

// ---------------------------------------------------------------------------------------------------------------
// forward: grid (N / 64, nimg, C / CB); a wave owns 16 queries.
//   qkv [nimg][N][ld]: q in channels [0, D), k in [D, 2D), v in [2D, 2D + C)
//   O   [nimg][N][ldO] = attention . v (bf16);  out = gamma * O + x;  lse [nimg][N] = log sum_j exp(energy[i][j])
// ---------------------------------------------------------------------------------------------------------------
template <int N, int D, int CB, bool STAGED>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, long long ld,
                                                       const bf16_t* __restrict__ x, long long ldx,
                                                       bf16_t* __restrict__ out, long long ldo, bf16_t* __restrict__ O,
                                                       long long ldO, float* __restrict__ lse,
                                                       const float* __restrict__ gamma) {
  constexpr int NT = N / 16, KH = D / 32, CT = CB / 16, NKC = N / 64;
  extern __shared__ __align__(16) unsigned char attn_smem[];
  bf16_t* vl = reinterpret_cast<bf16_t*>(attn_smem);          // [64 keys][CB], swizzled
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l = lane & 15;
  // Workgroups go to the XCDs round-robin by their linear index, and every query block of an image reads all of that
  // image's v: the blocks are renumbered so that an image's query blocks run on ONE XCD (its v is fetched into one
  // L2 instead of N / 64 of them).  Needs whole rounds of 8 images; otherwise the plain numbering.
  int qb = blockIdx.x;
  long long img = blockIdx.y;
  if ((gridDim.y & 7) == 0) {
    const int id = blockIdx.x + gridDim.x * blockIdx.y, xcd = id & 7, slot = id >> 3;
    img = (long long)(slot / (int)gridDim.x) * 8 + xcd;
    qb = slot % (int)gridDim.x;
  }
  const int q0 = qb * 64 + wave * 16, c0 = blockIdx.z * CB;
  const bf16_t* base = qkv + img * N * ld;
  // energy^T tiles: D[i = key][j = query] with A = k rows, B = q rows
  bf16x8_t fq[KH];
#pragma unroll
  for (int h = 0; h < KH; ++h) fq[h] = ld_frag(base + (long long)(q0 + l) * ld + h * 32 + 8 * g);
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < KH; ++h)
      acc[t] = mfma16(ld_frag(base + (long long)(t * 16 + l) * ld + D + h * 32 + 8 * g), fq[h], acc[t]);
  }
  // softmax over the keys of query l: in the lane over (t, r), across the four lane groups
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[t][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc[t][r] = __expf(acc[t][r] - mx); sum += acc[t][r]; }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] *= inv;
  if (blockIdx.z == 0 && g == 0) lse[img * N + q0 + l] = mx + __logf(sum);
  // O^T tiles: D[i = channel][j = query] = sum_key v[key][channel] * attention[query][key];
  // B = attention straight from the registers (paired order), A = v out of LDS (transposing read)
  f32x4 oacc[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) oacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the next chunk of v travels global -> registers while the current one is multiplied
  u32x4 pv[CB / 32];
  auto fetch = [&](int kc) {
#pragma unroll
    for (int u = 0; u < CB / 32; ++u) {
      const int idx = tid + 256 * u, key = idx / (CB / 8), c8 = (idx % (CB / 8)) * 8;
      pv[u] = *reinterpret_cast<const u32x4*>(base + (long long)(kc * 64 + key) * ld + 2 * D + c0 + c8);
    }
  };
  fetch(0);
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc) {
    if (kc) __syncthreads();
#pragma unroll
    for (int u = 0; u < CB / 32; ++u) {
      const int idx = tid + 256 * u, key = idx / (CB / 8), c8 = (idx % (CB / 8)) * 8;
      *reinterpret_cast<u32x4*>(vl + key * CB + (c8 ^ attn_swz(key))) = pv[u];
    }
    __syncthreads();
    if (kc + 1 < NKC) fetch(kc + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8_t b = pack_pair(acc[kc * 4 + 2 * ks], acc[kc * 4 + 2 * ks + 1]);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) oacc[ct] = mfma16(tr_frag<true, true>(vl, CB, 32 * ks, 16 * ct, lane), b, oacc[ct]);
    }
  }
  if constexpr (!STAGED) {
    const float gam = gamma[0];
    const long long row = img * N + q0 + l;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int ch = c0 + 16 * ct + 4 * g;
      const u32x2 ob = {pack2bf(oacc[ct][0], oacc[ct][1]), pack2bf(oacc[ct][2], oacc[ct][3])};
      *reinterpret_cast<u32x2*>(O + row * ldO + ch) = ob;
      const u32x2 xr = *reinterpret_cast<const u32x2*>(x + row * ldx + ch);
      *reinterpret_cast<u32x2*>(out + row * ldo + ch) =
          u32x2{pack2bf(gam * bflo(ob[0]) + bflo(xr[0]), gam * bfhi(ob[0]) + bfhi(xr[0])),
                pack2bf(gam * bflo(ob[1]) + bflo(xr[1]), gam * bfhi(ob[1]) + bfhi(xr[1]))};
    }
    return;
  }
  // Epilogue through LDS: in the MFMA layout a lane owns 4 channels of one query (8-byte pieces of 16 rows per
  // instruction); staged, 32 consecutive lanes read and write one contiguous 512-byte row of O, x and out.
  constexpr int PO = CB + 8;
  bf16_t* ol = vl + wave * 16 * PO;                           // this wave's [16 queries][PO]
  __syncthreads();                                            // every wave has read the last chunk of v
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
    *reinterpret_cast<u32x2*>(ol + l * PO + 16 * ct + 4 * g) =
        u32x2{pack2bf(oacc[ct][0], oacc[ct][1]), pack2bf(oacc[ct][2], oacc[ct][3])};
  const float gam = gamma[0];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16 * (CB / 8) / 64; ++u) {
    const int idx = lane + 64 * u, r = idx / (CB / 8), c8 = (idx % (CB / 8)) * 8;
    const long long row = img * N + q0 + r;
    const u32x4 ob = *reinterpret_cast<const u32x4*>(ol + r * PO + c8);
    const u32x4 xr = *reinterpret_cast<const u32x4*>(x + row * ldx + c0 + c8);
    *reinterpret_cast<u32x4*>(O + row * ldO + c0 + c8) = ob;
    u32x4 o;
    // (the stored, rounded O enters the residual sum, as it does when the two steps are separate kernels)
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = pack2bf(gam * bflo(ob[k]) + bflo(xr[k]), gam * bfhi(ob[k]) + bfhi(xr[k]));
    *reinterpret_cast<u32x4*>(out + row * ldo + c0 + c8) = o;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward: one workgroup per image; wave w owns keys 16w .. 16w+15 of the current block of 64 keys (dv, dk) and
// queries 16w .. 16w+15 of every block of 64 queries (dq).
//   dout = d loss / d out (the residual path is the caller's), dO = gamma * dout
//   Dq[i] = sum_c dO[i][c] O[i][c];  P = exp(energy - lse);  dP = dO v^T;  dS = P (dP - Dq)
//   dv = P^T dO,  dk = dS^T q,  dq = dS k;  dgamma[0] += sum dout * O
// ---------------------------------------------------------------------------------------------------------------
// VMODE: where a wave keeps the v fragments of its 16 keys while it walks the query blocks - 0 nowhere (fetched
// again for every query block), 1 registers, 2 its own slice of LDS (lane-linear: written and read by the same
// lanes, no barrier, no bank conflict).
template <int N, int D, int C, int VMODE, int NG>
__global__ __launch_bounds__(256 * NG) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, long long ld,
                                                            bf16_t* __restrict__ dqkv, long long ldg,
                                                            const bf16_t* __restrict__ dout, long long lddo,
                                                            const bf16_t* __restrict__ O, long long ldO,
                                                            const float* __restrict__ lse,
                                                            const float* __restrict__ gamma,
                                                            float* __restrict__ dgamma) {
  // NG groups of four waves: group r takes key blocks r, r + NG, ... while all groups share the staged query block
  // (two waves per SIMD instead of one, half as many passes over the query blocks); the groups' partial dq are
  // added up through LDS at the end.
  constexpr int NT = 256 * NG;
  constexpr int NB = N / 64, KH = D / 32, CK = C / 32, CT = C / 16, ET = D / 16, PQ = D + 8, PS = 72;
  static_assert(NB % NG == 0 && (64 * C / 8) % NT == 0, "key blocks per group, staging loads per thread");
  static_assert(NG == 1 || NB * ET * 4 * 64 * 16 <= 64 * C * 2, "the dq exchange reuses the dout tile");
  extern __shared__ __align__(16) unsigned char attn_smem[];
  bf16_t* dOl = reinterpret_cast<bf16_t*>(attn_smem);        // [64 queries][C], swizzled
  bf16_t* Ql = dOl + 64 * C;                                  // [64 queries][PQ]
  bf16_t* Kl0 = Ql + 64 * PQ;                                 // [NG][64 keys][PQ]
  bf16_t* dSl0 = Kl0 + NG * 64 * PQ;                          // [NG][64 keys][PS] (queries along the row)
  float* Dq = reinterpret_cast<float*>(dSl0 + NG * 64 * PS);  // [N]
  float* lsel = Dq + N;                                       // [N]
  float* red = lsel + N;                                      // [4 * NG]
  u32x4* vkeep = reinterpret_cast<u32x4*>(red + 4 * NG);      // VMODE 2: [4 * NG waves][C / 32][64 lanes] fragments
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l = lane & 15;
  const int grp = wave >> 2, w4 = wave & 3, gtid = tid & 255;
  bf16_t* Kl = Kl0 + grp * 64 * PQ;
  bf16_t* dSl = dSl0 + grp * 64 * PS;
  const long long img = blockIdx.x;
  const bf16_t* base = qkv + img * N * ld;
  bf16_t* gbase = dqkv + img * N * ldg;
  const bf16_t* dob = dout + img * N * lddo;
  const bf16_t* Ob = O + img * N * ldO;
  const float gam = gamma[0];
  // Dq (without gamma) and the block's share of dgamma
  {
    constexpr int LPR = C / 8, RPP = NT / LPR;
    const int c8 = (tid % LPR) * 8;
    float tot = 0.f;
    for (int q = tid / LPR; q < N; q += RPP) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(dob + (long long)q * lddo + c8);
      const u32x4 b = *reinterpret_cast<const u32x4*>(Ob + (long long)q * ldO + c8);
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += bflo(a[k]) * bflo(b[k]) + bfhi(a[k]) * bfhi(b[k]);
      tot += s;
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if ((tid % LPR) == 0) Dq[q] = s;
    }
    for (int q = tid; q < N; q += NT) lsel[q] = lse[img * N + q];
    if (dgamma) {
      tot = wave_sum(tot);
      if (lane == 0) red[wave] = tot;
      __syncthreads();
      if (tid == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 4 * NG; ++k) t += red[k];
        atomicAdd(dgamma, t);
      }
    }
  }
  f32x4 dQacc[NB][ET];                                        // D[i = e][j = query 16w + l] of query block ib
#pragma unroll
  for (int ib = 0; ib < NB; ++ib)
#pragma unroll
    for (int et = 0; et < ET; ++et) dQacc[ib][et] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int jb2 = 0; jb2 < NB / NG; ++jb2) {
    const int jb = jb2 * NG + grp;
    const bf16_t* krow = base + (long long)(jb * 64 + w4 * 16 + l) * ld;          // this lane's key (B operands)
    bf16x8_t fk[KH];
#pragma unroll
    for (int h = 0; h < KH; ++h) fk[h] = ld_frag(krow + D + h * 32 + 8 * g);
    bf16x8_t fv[VMODE == 1 ? CK : 1];
    if constexpr (VMODE == 1) {
#pragma unroll
      for (int s = 0; s < CK; ++s) fv[s] = ld_frag(krow + 2 * D + s * 32 + 8 * g);
    }
    if constexpr (VMODE == 2) {
#pragma unroll
      for (int s = 0; s < CK; ++s)
        vkeep[(wave * CK + s) * 64 + lane] = *reinterpret_cast<const u32x4*>(krow + 2 * D + s * 32 + 8 * g);
    }
    f32x4 dVacc[CT], dKacc[ET];                               // D[i = channel | e][j = key 16w + l]
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) dVacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int et = 0; et < ET; ++et) dKacc[et] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                          // the previous key block's last dq pass has read Kl
    for (int idx = gtid; idx < 64 * D / 8; idx += 256) {      // every group its own key block
      const int r = idx / (D / 8), c8 = (idx % (D / 8)) * 8;
      *reinterpret_cast<u32x4*>(Kl + r * PQ + c8) =
          *reinterpret_cast<const u32x4*>(base + (long long)(jb * 64 + r) * ld + D + c8);
    }
    // The next query block's dout and q rows travel global -> registers while the current block is computed.
    constexpr int NDO = 64 * C / 8 / NT, NQ = (64 * D / 8 + NT - 1) / NT;
    u32x4 pdo[NDO], pq[NQ];
    auto fetch = [&](int ib) {
#pragma unroll
      for (int u = 0; u < NDO; ++u) {
        const int idx = tid + NT * u, r = idx / (C / 8), c8 = (idx % (C / 8)) * 8;
        pdo[u] = *reinterpret_cast<const u32x4*>(dob + (long long)(ib * 64 + r) * lddo + c8);
      }
#pragma unroll
      for (int u = 0; u < NQ; ++u) {
        const int idx = tid + NT * u, r = idx / (D / 8), c8 = (idx % (D / 8)) * 8;
        if (idx < 64 * D / 8) pq[u] = *reinterpret_cast<const u32x4*>(base + (long long)(ib * 64 + r) * ld + c8);
      }
    };
    fetch(0);
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) {
      // stage this query block: dout rows (swizzled) and q rows
#pragma unroll
      for (int u = 0; u < NDO; ++u) {
        const int idx = tid + NT * u, r = idx / (C / 8), c8 = (idx % (C / 8)) * 8;
        *reinterpret_cast<u32x4*>(dOl + r * C + (c8 ^ attn_swz(r))) = pdo[u];
      }
#pragma unroll
      for (int u = 0; u < NQ; ++u) {
        const int idx = tid + NT * u, r = idx / (D / 8), c8 = (idx % (D / 8)) * 8;
        if (idx < 64 * D / 8) *reinterpret_cast<u32x4*>(Ql + r * PQ + c8) = pq[u];
      }
      __syncthreads();
      if (ib + 1 < NB) fetch(ib + 1);
      // (LDS addresses recomputed per query block: hoisted out of the unrolled loops they cost ~100 registers)
      int lane_v = lane;
      asm volatile("" : "+v"(lane_v));
      const int l = lane_v & 15, g = lane_v >> 4;
      // energy and dP tiles: D[i = query 16t + 4g + r][j = key]
      f32x4 s[4], dp[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < KH; ++h) s[t] = mfma16(ld_frag(Ql + (16 * t + l) * PQ + h * 32 + 8 * g), fk[h], s[t]);
      }
#pragma unroll
      for (int k = 0; k < CK; ++k) {
        bf16x8_t bv;
        if constexpr (VMODE == 1) bv = fv[k];
        else if constexpr (VMODE == 2) bv = __builtin_bit_cast(bf16x8_t, vkeep[(wave * CK + k) * 64 + lane]);
        else bv = ld_frag(krow + 2 * D + k * 32 + 8 * g);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = 16 * t + l;
          dp[t] = mfma16(ld_frag(dOl + r * C + ((k * 32 + 8 * g) ^ attn_swz(r))), bv, dp[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int q = 16 * t + 4 * g;
        const f32x4 ls = *reinterpret_cast<const f32x4*>(lsel + ib * 64 + q);
        const f32x4 dq4 = *reinterpret_cast<const f32x4*>(Dq + ib * 64 + q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __expf(s[t][r] - ls[r]);
          dp[t][r] = gam * p * (dp[t][r] - dq4[r]);
          s[t][r] = p;
        }
        // dS^T image: row = this lane's key, four consecutive queries
        *reinterpret_cast<u32x2*>(dSl + (w4 * 16 + l) * PS + q) =
            u32x2{pack2bf(dp[t][0], dp[t][1]), pack2bf(dp[t][2], dp[t][3])};
      }
      // contractions over the 64 queries: dv^T and dk^T for this wave's keys
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bf16x8_t bp = pack_pair(s[2 * h], s[2 * h + 1]);
        const bf16x8_t bs = pack_pair(dp[2 * h], dp[2 * h + 1]);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          dVacc[ct] = mfma16(tr_frag<true, true>(dOl, C, 32 * h, 16 * ct, lane_v), bp, dVacc[ct]);
#pragma unroll
        for (int et = 0; et < ET; ++et)
          dKacc[et] = mfma16(tr_frag<false, true>(Ql, PQ, 32 * h, 16 * et, lane_v), bs, dKacc[et]);
      }
      __syncthreads();
      // dq^T for this wave's queries: D[i = e][j = query] = sum_key k[key][e] dS[query][key]; both operands are
      // [key][.] images read through the transposing load
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bf16x8_t b = tr_frag<false, false>(dSl, PS, 32 * h, 16 * w4, lane_v);
#pragma unroll
        for (int et = 0; et < ET; ++et)
          dQacc[ib][et] = mfma16(tr_frag<false, false>(Kl, PQ, 32 * h, 16 * et, lane_v), b, dQacc[ib][et]);
      }
    }
    bf16_t* grow = gbase + (long long)(jb * 64 + w4 * 16 + l) * ldg;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      *reinterpret_cast<u32x2*>(grow + 2 * D + 16 * ct + 4 * g) =
          u32x2{pack2bf(gam * dVacc[ct][0], gam * dVacc[ct][1]), pack2bf(gam * dVacc[ct][2], gam * dVacc[ct][3])};
#pragma unroll
    for (int et = 0; et < ET; ++et)
      *reinterpret_cast<u32x2*>(grow + D + 16 * et + 4 * g) =
          u32x2{pack2bf(dKacc[et][0], dKacc[et][1]), pack2bf(dKacc[et][2], dKacc[et][3])};
  }
  if constexpr (NG > 1) {            // the other groups' dq partial sums, through the (now dead) dout tile
    f32x4* xch = reinterpret_cast<f32x4*>(attn_smem);
    static_assert(NG == 2, "one exchange step");
    if (grp == 1) {
#pragma unroll
      for (int ib = 0; ib < NB; ++ib)
#pragma unroll
        for (int et = 0; et < ET; ++et) xch[((ib * ET + et) * 4 + w4) * 64 + lane] = dQacc[ib][et];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int ib = 0; ib < NB; ++ib)
#pragma unroll
      for (int et = 0; et < ET; ++et) {
        const f32x4 o = xch[((ib * ET + et) * 4 + w4) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) dQacc[ib][et][r] += o[r];
      }
  }
#pragma unroll
  for (int ib = 0; ib < NB; ++ib) {
    bf16_t* grow = gbase + (long long)(ib * 64 + w4 * 16 + l) * ldg;
#pragma unroll
    for (int et = 0; et < ET; ++et)
      *reinterpret_cast<u32x2*>(grow + 16 * et + 4 * g) =
          u32x2{pack2bf(dQacc[ib][et][0], dQacc[ib][et][1]), pack2bf(dQacc[ib][et][2], dQacc[ib][et][3])};
  }
}

// ---------------------------------------------------------------------------------------------------------------
extern "C" int ried_attn_fused_supported(int n, int d, int C) {
  return (n == 256 && d == 32 && C == 256) || (n == 64 && d == 64 && C == 512);
}

static int attn_check(const char* fn, const void* qkv, long long ld, int nimg, int n, int d, int C) {
  if (!qkv || nimg <= 0 || nimg > 65535) { ried_set_error("%s: bad arguments", fn); return RIED_ERR_ARG; }
  if (!ried_attn_fused_supported(n, d, C)) {
    ried_set_error("%s: unsupported n=%d d=%d C=%d", fn, n, d, C);
    return RIED_ERR_ARG;
  }
  if (ld % 8 || ld < 2 * d + C || ((uintptr_t)qkv % 16)) { ried_set_error("%s: qkv alignment", fn); return RIED_ERR_ARG; }
  return RIED_OK;
}
static bool aligned_view(const void* p, long long ld, int C) { return p && ld % 8 == 0 && ld >= C && ((uintptr_t)p % 16) == 0; }

template <class K>
static int raise_lds(K kernel, size_t bytes, const char* fn) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)bytes);
  if (e != hipSuccess) {
    ried_set_error("%s: cannot raise the dynamic LDS limit: %s", fn, hipGetErrorString(e));
    return RIED_ERR_HIP;
  }
  return RIED_OK;
}

extern "C" int ried_attn_fwd(const void* qkv, long long ld, const void* x, long long ldx, void* out, long long ldo,
                             void* O, long long ldO, float* lse, const float* gamma, int nimg, int n, int d, int C,
                             hipStream_t stream) {
  int rc = attn_check("ried_attn_fwd", qkv, ld, nimg, n, d, C);
  if (rc) return rc;
  RIED_CHECK_ARG(aligned_view(x, ldx, C) && aligned_view(out, ldo, C) && aligned_view(O, ldO, C) && lse && gamma,
                 "ried_attn_fwd: bad view");
  constexpr int CB = 256;
  const size_t sh = 64 * (CB + 8) * sizeof(bf16_t);       // the v chunk, then the staged output rows
  dim3 grid(n / 64, nimg, C / CB);
  if (n == 256)
    hipLaunchKernelGGL((attn_fwd_kernel<256, 32, CB, false>), grid, dim3(256), sh, stream, (const bf16_t*)qkv, ld,
                       (const bf16_t*)x, ldx, (bf16_t*)out, ldo, (bf16_t*)O, ldO, lse, gamma);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<64, 64, CB, true>), grid, dim3(256), sh, stream, (const bf16_t*)qkv, ld,
                       (const bf16_t*)x, ldx, (bf16_t*)out, ldo, (bf16_t*)O, ldO, lse, gamma);
  RIED_CHECK_LAUNCH("ried_attn_fwd");
  return RIED_OK;
}

template <int N, int D, int C, int NG, int VMODE>
static constexpr size_t attn_bwd_lds() {
  return (size_t)(64 * C + (1 + NG) * 64 * (D + 8) + NG * 64 * 72) * sizeof(bf16_t) + (2 * N + 4 * NG) * sizeof(float) +
         (VMODE == 2 ? (size_t)4 * NG * (C / 32) * 64 * 16 : 0);
}

extern "C" int ried_attn_bwd(const void* qkv, long long ld, void* dqkv, long long ldg, const void* dout, long long lddo,
                             const void* O, long long ldO, const float* lse, const float* gamma, float* dgamma, int nimg,
                             int n, int d, int C, hipStream_t stream) {
  int rc = attn_check("ried_attn_bwd", qkv, ld, nimg, n, d, C);
  if (rc) return rc;
  RIED_CHECK_ARG(aligned_view(dqkv, ldg, 2 * d + C) && aligned_view(dout, lddo, C) && aligned_view(O, ldO, C) && lse && gamma,
                 "ried_attn_bwd: bad view");
  static int raised = 0, raised2 = 0;
  if (n == 256) {            // two groups of four waves: two key blocks at a time
    auto k = attn_bwd_kernel<256, 32, 256, 2, 2>;
    const size_t sh = attn_bwd_lds<256, 32, 256, 2, 2>();
    if (!raised2) {
      rc = raise_lds(k, sh, "ried_attn_bwd");
      if (rc) return rc;
      raised2 = 1;
    }
    hipLaunchKernelGGL(k, dim3(nimg), dim3(512), sh, stream, (const bf16_t*)qkv, ld, (bf16_t*)dqkv, ldg,
                       (const bf16_t*)dout, lddo, (const bf16_t*)O, ldO, lse, gamma, dgamma);
  } else {
    auto k = attn_bwd_kernel<64, 64, 512, 0, 1>;
    const size_t sh = attn_bwd_lds<64, 64, 512, 1, 0>();
    if (!raised) {
      rc = raise_lds(k, sh, "ried_attn_bwd");
      if (rc) return rc;
      raised = 1;
    }
    hipLaunchKernelGGL(k, dim3(nimg), dim3(256), sh, stream, (const bf16_t*)qkv, ld, (bf16_t*)dqkv, ldg,
                       (const bf16_t*)dout, lddo, (const bf16_t*)O, ldO, lse, gamma, dgamma);
  }
  RIED_CHECK_LAUNCH("ried_attn_bwd");
  return RIED_OK;
}
