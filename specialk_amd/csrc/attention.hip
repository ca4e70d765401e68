// Scaled dot-product attention, forward and backward, for any (Lq, Lk, dk, dv <= 128) and
// both storage types, computing in fp32 on the vector ALUs.
//
// Replaces ScaledDotProductAttention.forward (lib/transformer/Modules.py:17-28: bmm, /T,
// masked_fill(-inf), softmax, dropout, bmm) plus the head split/merge copies and the
// h-fold mask repeat of MultiHeadAttention (lib/transformer/SubLayers.py:50-60).  Nothing of
// size (h*B, Lq, Lk) is ever written to HBM: scores live in registers, the softmax is the
// online (running max / running sum) form, and backward recomputes probabilities from the
// saved row log-sum-exp.
//
// Work decomposition (all three kernels share it): a block owns 16 "own" rows of one
// (batch, head) — query rows for forward / dQ, key rows for dK/dV — 4 per wave, and sweeps
// the "other" sequence in tiles of 64 rows staged in LDS; inside a tile lane l owns other-row
// l, so a score row is one value per lane and the row reductions are wave reductions.
// The second product (P·V, dS·K, P^T·dO, dS^T·Q) broadcasts each lane's coefficient with
// v_readlane (compile-time lane index) and accumulates one output column per lane.
//
// This is the exact-fp32 path used for the <=1e-5 parity runs and the fallback for head
// sizes the MFMA kernel does not cover.
#include "attention.h"
#include <stdlib.h>

using namespace skattn;

namespace {

constexpr int OWN = 16;    // own rows per block
constexpr int TILE = 64;   // other rows per tile (= wave width)
constexpr int MAXD = 128;  // max head dim

// Stage `rows` x `dim` elements starting at row r0 of a (row stride ld, column offset col0)
// matrix into LDS as fp32 with row stride dim+1 (conflict-free "one row per lane" reads).
template <typename T>
__device__ __forceinline__ void stage(float* dst, const void* src, int64_t ld, int64_t base_row, int r0, int nrows_total,
                                      int rows, int col0, int dim, float mul) {
  for (int e = threadIdx.x; e < rows * dim; e += blockDim.x) {
    int r = e / dim, c = e - r * dim;
    float v = 0.f;
    if (r0 + r < nrows_total) v = Cvt<T>::to_f32(((const T*)src)[(base_row + r0 + r) * ld + col0 + c]) * mul;
    dst[r * (dim + 1) + c] = v;
  }
}

// sum_c a[c] * tile[lane][c]   (a broadcast from LDS, tile row per lane)
__device__ __forceinline__ float dot_row(const float* a, const float* tile_row, int dim) {
  float s = 0.f;
  for (int c = 0; c < dim; ++c) s += a[c] * tile_row[c];
  return s;
}

// acc[0..1] += sum_l coef(lane l) * tile[l][col], col = lane (+64)
__device__ __forceinline__ void accum_cols(float (&acc)[2], float coef, const float* tile, int dim, int lane) {
  const int stride = dim + 1;
  if (dim <= 64) {
    const int c = lane < dim ? lane : 0;
#pragma unroll 8
    for (int l = 0; l < TILE; ++l) {
      float cf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(coef), l));
      acc[0] += cf * tile[l * stride + c];
    }
  } else {
    const int c1 = lane + 64 < dim ? lane + 64 : 0;
#pragma unroll 8
    for (int l = 0; l < TILE; ++l) {
      float cf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(coef), l));
      acc[0] += cf * tile[l * stride + lane];
      acc[1] += cf * tile[l * stride + c1];
    }
  }
}

// ------------------------------------------------------------------------------ forward
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Ks = (float*)smem_raw;                 // [TILE][dk+1]
  float* Vs = Ks + TILE * (p.dk + 1);           // [TILE][dv+1]
  float* qs = Vs + TILE * (p.dv + 1);           // [OWN][dk+1]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * OWN;
  const Seq sq = seq_of(p, b);
  const int Lq = sq.Lq, Lk = sq.Lk;
  if (i0 >= Lq) return;  // block-uniform (shorter sentence of a packed batch)
  const int64_t qrow0 = sq.qbase, krow0 = sq.kbase;

  stage<T>(qs, p.Q, p.ldq, qrow0, i0, Lq, OWN, h * p.dk, p.dk, p.scale);
  float m[4], l[4], acc[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; acc[r][0] = acc[r][1] = 0.f; }

  for (int j0 = 0; j0 < Lk; j0 += TILE) {
    __syncthreads();
    stage<T>(Ks, p.K, p.ldk, krow0, j0, Lk, TILE, h * p.dk, p.dk, 1.f);
    stage<T>(Vs, p.V, p.ldv, krow0, j0, Lk, TILE, h * p.dv, p.dv, 1.f);
    __syncthreads();
    const int j = j0 + lane;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + wave * 4 + r;
      float s = masked(p, sq, b, i, j) ? -INFINITY : dot_row(qs + (wave * 4 + r) * (p.dk + 1), Ks + lane * (p.dk + 1), p.dk);
      float mn = fmaxf(m[r], wave_max(s));
      float pr = 0.f, alpha = 1.f;
      if (mn != -INFINITY) {
        alpha = expf(m[r] - mn);
        pr = expf(s - mn);
      }
      l[r] = l[r] * alpha + wave_sum(pr);
      m[r] = mn;
      float pd = keep(p, b, h, i, j) ? pr * p.keep_scale : 0.f;
      acc[r][0] *= alpha; acc[r][1] *= alpha;
      accum_cols(acc[r], pd, Vs, p.dv, lane);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + wave * 4 + r;
    if (i < Lq) {
      float inv = 1.0f / l[r];  // l == 0 (fully masked row) -> 0 * inf = NaN, as torch's softmax gives
      int64_t off = (qrow0 + i) * p.ldo + h * p.dv;
      if (lane < p.dv) st<T>(p.Oout, off + lane, acc[r][0] * inv);
      if (lane + 64 < p.dv) st<T>(p.Oout, off + lane + 64, acc[r][1] * inv);
      if (lane == 0) p.LSE[((int64_t)b * p.H + h) * p.Lq + i] = m[r] + logf(l[r]);
    }
  }
  if (p.P_out) {  // optional: materialise the (dropped-out) probabilities, head-major like the reference
    for (int j0 = 0; j0 < Lk; j0 += TILE) {
      __syncthreads();
      stage<T>(Ks, p.K, p.ldk, krow0, j0, Lk, TILE, h * p.dk, p.dk, 1.f);
      __syncthreads();
      const int j = j0 + lane;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wave * 4 + r;
        if (i < Lq && j < Lk) {
          float pr;
          if (l[r] == 0.f) pr = NAN;
          else if (masked(p, sq, b, i, j)) pr = 0.f;
          else pr = expf(dot_row(qs + (wave * 4 + r) * (p.dk + 1), Ks + lane * (p.dk + 1), p.dk) - m[r]) / l[r];
          pr = keep(p, b, h, i, j) ? pr * p.keep_scale : 0.f;
          st<T>(p.P_out, (((int64_t)h * p.B + b) * p.Lq + i) * p.Lk + j, pr);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------ delta = rowsum(dO * O)
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t total = (int64_t)p.B * p.H * p.Lq;
  for (int64_t idx = (int64_t)blockIdx.x * 4 + wave; idx < total; idx += (int64_t)gridDim.x * 4) {
    int i = (int)(idx % p.Lq);
    int h = (int)((idx / p.Lq) % p.H);
    int b = (int)(idx / ((int64_t)p.Lq * p.H));
    const Seq sq = seq_of(p, b);
    if (i >= sq.Lq) continue;  // wave-uniform
    int64_t off = (sq.qbase + i) * p.ldo + h * p.dv;
    float s = 0.f;
    for (int c = lane; c < p.dv; c += 64) s += ld<T>(p.dO, off + c) * ld<T>(p.O, off + c);
    s = wave_sum(s);
    if (lane == 0) p.delta[idx] = s;
  }
}

// ------------------------------------------------------------------------------ dQ
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Ks = (float*)smem_raw;                 // [TILE][dk+1]
  float* Vs = Ks + TILE * (p.dk + 1);           // [TILE][dv+1]
  float* qs = Vs + TILE * (p.dv + 1);           // [OWN][dk+1]  (pre-scaled)
  float* dos = qs + OWN * (p.dk + 1);           // [OWN][dv+1]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * OWN;
  const Seq sq = seq_of(p, b);
  const int Lq = sq.Lq, Lk = sq.Lk;
  if (i0 >= Lq) return;  // block-uniform
  const int64_t qrow0 = sq.qbase, krow0 = sq.kbase;
  stage<T>(qs, p.Q, p.ldq, qrow0, i0, Lq, OWN, h * p.dk, p.dk, p.scale);
  stage<T>(dos, p.dO, p.ldo, qrow0, i0, Lq, OWN, h * p.dv, p.dv, 1.f);
  float lse[4], dl[4], acc[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + wave * 4 + r;
    int64_t si = ((int64_t)b * p.H + h) * p.Lq + (i < Lq ? i : 0);
    lse[r] = p.LSE[si]; dl[r] = p.delta[si];
    acc[r][0] = acc[r][1] = 0.f;
  }
  for (int j0 = 0; j0 < Lk; j0 += TILE) {
    __syncthreads();
    stage<T>(Ks, p.K, p.ldk, krow0, j0, Lk, TILE, h * p.dk, p.dk, 1.f);
    stage<T>(Vs, p.V, p.ldv, krow0, j0, Lk, TILE, h * p.dv, p.dv, 1.f);
    __syncthreads();
    const int j = j0 + lane;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + wave * 4 + r;
      float ds = 0.f;
      if (!masked(p, sq, b, i, j)) {
        float s = dot_row(qs + (wave * 4 + r) * (p.dk + 1), Ks + lane * (p.dk + 1), p.dk);
        float pr = expf(s - lse[r]);
        float dp = dot_row(dos + (wave * 4 + r) * (p.dv + 1), Vs + lane * (p.dv + 1), p.dv);
        dp = keep(p, b, h, i, j) ? dp * p.keep_scale : 0.f;
        ds = pr * (dp - dl[r]) * p.scale;
      }
      accum_cols(acc[r], ds, Ks, p.dk, lane);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + wave * 4 + r;
    if (i < Lq) {
      int64_t off = (qrow0 + i) * p.lddq + h * p.dk;
      if (lane < p.dk) st<T>(p.dQ, off + lane, acc[r][0]);
      if (lane + 64 < p.dk) st<T>(p.dQ, off + lane + 64, acc[r][1]);
    }
  }
}

// ------------------------------------------------------------------------------ dK, dV
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Qs = (float*)smem_raw;                 // [TILE][dk+1]  (pre-scaled)
  float* dOs = Qs + TILE * (p.dk + 1);          // [TILE][dv+1]
  float* ks = dOs + TILE * (p.dv + 1);          // [OWN][dk+1]
  float* vs = ks + OWN * (p.dk + 1);            // [OWN][dv+1]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.z, h = blockIdx.y, j0 = blockIdx.x * OWN;
  const Seq sq = seq_of(p, b);
  const int Lq = sq.Lq, Lk = sq.Lk;
  if (j0 >= Lk) return;  // block-uniform
  const int64_t qrow0 = sq.qbase, krow0 = sq.kbase;
  stage<T>(ks, p.K, p.ldk, krow0, j0, Lk, OWN, h * p.dk, p.dk, 1.f);
  stage<T>(vs, p.V, p.ldv, krow0, j0, Lk, OWN, h * p.dv, p.dv, 1.f);
  float accK[4][2], accV[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r) accK[r][0] = accK[r][1] = accV[r][0] = accV[r][1] = 0.f;
  for (int i0 = 0; i0 < Lq; i0 += TILE) {
    __syncthreads();
    stage<T>(Qs, p.Q, p.ldq, qrow0, i0, Lq, TILE, h * p.dk, p.dk, p.scale);
    stage<T>(dOs, p.dO, p.ldo, qrow0, i0, Lq, TILE, h * p.dv, p.dv, 1.f);
    __syncthreads();
    const int i = i0 + lane;
    int64_t si = ((int64_t)b * p.H + h) * p.Lq + (i < Lq ? i : 0);
    const float lse = p.LSE[si], dl = p.delta[si];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = j0 + wave * 4 + r;
      float pd = 0.f, ds = 0.f;
      if (!masked(p, sq, b, i, j)) {
        float s = dot_row(ks + (wave * 4 + r) * (p.dk + 1), Qs + lane * (p.dk + 1), p.dk);
        float pr = expf(s - lse);
        float dp = dot_row(vs + (wave * 4 + r) * (p.dv + 1), dOs + lane * (p.dv + 1), p.dv);
        bool kp = keep(p, b, h, i, j);
        pd = kp ? pr * p.keep_scale : 0.f;
        dp = kp ? dp * p.keep_scale : 0.f;
        ds = pr * (dp - dl);  // Qs is pre-scaled, so dK = ds^T (scale Q) needs no extra factor
      }
      accum_cols(accV[r], pd, dOs, p.dv, lane);
      accum_cols(accK[r], ds, Qs, p.dk, lane);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = j0 + wave * 4 + r;
    if (j < Lk) {
      int64_t offk = (krow0 + j) * p.lddk + h * p.dk, offv = (krow0 + j) * p.lddv + h * p.dv;
      if (lane < p.dk) st<T>(p.dK, offk + lane, accK[r][0]);
      if (lane + 64 < p.dk) st<T>(p.dK, offk + lane + 64, accK[r][1]);
      if (lane < p.dv) st<T>(p.dV, offv + lane, accV[r][0]);
      if (lane + 64 < p.dv) st<T>(p.dV, offv + lane + 64, accV[r][1]);
    }
  }
}

// SKHIP_ATTN_VALU=1 forces the vector-ALU kernels for bf16 too (A/B testing of the MFMA path).
bool force_valu() {
  const char* e = getenv("SKHIP_ATTN_VALU");
  return e && e[0] == '1';
}

int check_common(const char* who, int B, int H, int Lq, int Lk, int dk, int dv, float p_drop, int dtype) {
  SK_CHECK_ARG(dtype == SK_F32 || dtype == SK_BF16, "%s: bad dtype %d", who, dtype);
  SK_CHECK_ARG(B >= 0 && H > 0 && Lq >= 0 && Lk >= 0, "%s: bad shape", who);
  SK_CHECK_ARG(dk > 0 && dv > 0 && dk <= MAXD && dv <= MAXD, "%s: head dims (%d,%d) must be in 1..%d", who, dk, dv, MAXD);
  SK_CHECK_ARG(H <= 65535 && B <= 65535, "%s: B and n_head must be <= 65535", who);
  SK_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "%s: p_drop=%f", who, p_drop);
  return SK_OK;
}

int attn_fwd_impl(const void* Q, const void* K, const void* V, void* O, float* LSE, void* P_out,
                  const uint8_t* key_pad, const uint8_t* dense_mask, const int32_t* q_off, const int32_t* k_off, int B,
                  int n_head, int Lq, int Lk, int dk, int dv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                  float scale, int causal, float p_drop, uint64_t seed, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common("skhip_attn_fwd", B, n_head, Lq, Lk, dk, dv, p_drop, dtype);
  if (rc) return rc;
  if (B == 0 || Lq == 0) return SK_OK;
  SK_CHECK_ARG(Q && K && V && O && LSE, "skhip_attn_fwd: null pointer");
  AttnParams p = {};
  p.Q = Q; p.K = K; p.V = V; p.Oout = O; p.LSE = LSE; p.P_out = P_out;
  p.key_pad = key_pad; p.dense = dense_mask; p.q_off = q_off; p.k_off = k_off;
  p.B = B; p.H = n_head; p.Lq = Lq; p.Lk = Lk; p.dk = dk; p.dv = dv;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.scale = scale; p.causal = causal;
  p.drop_thr = sk_drop_threshold(p_drop); p.keep_scale = 1.0f / (1.0f - p_drop); p.seed = seed;
  { const char* e = getenv("SKHIP_ATTN_WIDE"); p.wide_store = !(e && e[0] == '0'); }
  if (Lk > 0 && !force_valu() && sk_attn_mfma_supported(p, dtype, P_out != nullptr)) return sk_attn_fwd_mfma(p, stream);
  dim3 grid((Lq + OWN - 1) / OWN, n_head, B);
  size_t smem = sizeof(float) * ((size_t)TILE * (dk + 1) + (size_t)TILE * (dv + 1) + (size_t)OWN * (dk + 1));
  if (dtype == SK_BF16) {
    if ((rc = sk_allow_big_lds(attn_fwd_kernel<bf16_t>, smem))) return rc;
    hipLaunchKernelGGL((attn_fwd_kernel<bf16_t>), grid, dim3(256), smem, stream, p);
  } else {
    if ((rc = sk_allow_big_lds(attn_fwd_kernel<float>, smem))) return rc;
    hipLaunchKernelGGL((attn_fwd_kernel<float>), grid, dim3(256), smem, stream, p);
  }
  SK_CHECK_LAUNCH("skhip_attn_fwd");
  return SK_OK;
}

int attn_bwd_impl(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                  void* dQ, void* dK, void* dV, float* delta_ws, const uint8_t* key_pad, const uint8_t* dense_mask,
                  const int32_t* q_off, const int32_t* k_off, int B, int n_head, int Lq, int Lk, int dk, int dv,
                  int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv,
                  float scale, int causal, float p_drop, uint64_t seed, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common("skhip_attn_bwd", B, n_head, Lq, Lk, dk, dv, p_drop, dtype);
  if (rc) return rc;
  if (B == 0 || Lq == 0 || Lk == 0) return SK_OK;
  SK_CHECK_ARG(Q && K && V && O && dO && LSE && dQ && dK && dV && delta_ws, "skhip_attn_bwd: null pointer");
  AttnParams p = {};
  p.Q = Q; p.K = K; p.V = V; p.O = O; p.dO = dO; p.LSE = (float*)LSE; p.delta = delta_ws;
  p.dQ = dQ; p.dK = dK; p.dV = dV;
  p.key_pad = key_pad; p.dense = dense_mask; p.q_off = q_off; p.k_off = k_off;
  p.B = B; p.H = n_head; p.Lq = Lq; p.Lk = Lk; p.dk = dk; p.dv = dv;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
  p.scale = scale; p.causal = causal;
  p.drop_thr = sk_drop_threshold(p_drop); p.keep_scale = 1.0f / (1.0f - p_drop); p.seed = seed;
  { const char* e = getenv("SKHIP_ATTN_WIDE"); p.wide_store = !(e && e[0] == '0'); }
  if (!force_valu() && sk_attn_mfma_supported(p, dtype, false)) return sk_attn_bwd_mfma(p, stream);
  int64_t rows = (int64_t)B * n_head * Lq;
  unsigned dblocks = (unsigned)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
  size_t smem_q = sizeof(float) * ((size_t)TILE * (dk + 1) + (size_t)TILE * (dv + 1) + (size_t)OWN * (dk + 1) + (size_t)OWN * (dv + 1));
  dim3 gq((Lq + OWN - 1) / OWN, n_head, B), gk((Lk + OWN - 1) / OWN, n_head, B);
  if (dtype == SK_BF16) {
    if ((rc = sk_allow_big_lds(attn_bwd_dq_kernel<bf16_t>, smem_q))) return rc;
    if ((rc = sk_allow_big_lds(attn_bwd_dkv_kernel<bf16_t>, smem_q))) return rc;
    hipLaunchKernelGGL((attn_delta_kernel<bf16_t>), dim3(dblocks), dim3(256), 0, stream, p);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<bf16_t>), gq, dim3(256), smem_q, stream, p);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<bf16_t>), gk, dim3(256), smem_q, stream, p);
  } else {
    if ((rc = sk_allow_big_lds(attn_bwd_dq_kernel<float>, smem_q))) return rc;
    if ((rc = sk_allow_big_lds(attn_bwd_dkv_kernel<float>, smem_q))) return rc;
    hipLaunchKernelGGL((attn_delta_kernel<float>), dim3(dblocks), dim3(256), 0, stream, p);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<float>), gq, dim3(256), smem_q, stream, p);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<float>), gk, dim3(256), smem_q, stream, p);
  }
  SK_CHECK_LAUNCH("skhip_attn_bwd");
  return SK_OK;
}

}  // namespace

extern "C" int skhip_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, void* P_out,
                              const uint8_t* key_pad, const uint8_t* dense_mask, int B, int n_head, int Lq, int Lk, int dk,
                              int dv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, int causal,
                              float p_drop, uint64_t seed, int dtype, void* stream) {
  return attn_fwd_impl(Q, K, V, O, LSE, P_out, key_pad, dense_mask, nullptr, nullptr, B, n_head, Lq, Lk, dk, dv, ldq, ldk,
                       ldv, ldo, scale, causal, p_drop, seed, dtype, stream);
}

extern "C" int skhip_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                              void* dQ, void* dK, void* dV, float* delta_ws, const uint8_t* key_pad,
                              const uint8_t* dense_mask, int B, int n_head, int Lq, int Lk, int dk, int dv, int64_t ldq,
                              int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, float scale,
                              int causal, float p_drop, uint64_t seed, int dtype, void* stream) {
  return attn_bwd_impl(Q, K, V, O, dO, LSE, dQ, dK, dV, delta_ws, key_pad, dense_mask, nullptr, nullptr, B, n_head, Lq, Lk,
                       dk, dv, ldq, ldk, ldv, ldo, lddq, lddk, lddv, scale, causal, p_drop, seed, dtype, stream);
}

extern "C" int skhip_attn_varlen_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE,
                                     const int32_t* q_offsets, const int32_t* k_offsets, int B, int n_head, int max_Lq,
                                     int max_Lk, int dk, int dv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                                     float scale, int causal, float p_drop, uint64_t seed, int dtype, void* stream) {
  SK_CHECK_ARG(q_offsets && k_offsets, "skhip_attn_varlen_fwd: null offsets");
  return attn_fwd_impl(Q, K, V, O, LSE, nullptr, nullptr, nullptr, q_offsets, k_offsets, B, n_head, max_Lq, max_Lk, dk, dv,
                       ldq, ldk, ldv, ldo, scale, causal, p_drop, seed, dtype, stream);
}

extern "C" int skhip_attn_varlen_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                                     const float* LSE, void* dQ, void* dK, void* dV, float* delta_ws,
                                     const int32_t* q_offsets, const int32_t* k_offsets, int B, int n_head, int max_Lq,
                                     int max_Lk, int dk, int dv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                                     int64_t lddq, int64_t lddk, int64_t lddv, float scale, int causal, float p_drop,
                                     uint64_t seed, int dtype, void* stream) {
  SK_CHECK_ARG(q_offsets && k_offsets, "skhip_attn_varlen_bwd: null offsets");
  return attn_bwd_impl(Q, K, V, O, dO, LSE, dQ, dK, dV, delta_ws, nullptr, nullptr, q_offsets, k_offsets, B, n_head, max_Lq,
                       max_Lk, dk, dv, ldq, ldk, ldv, ldo, lddq, lddk, lddv, scale, causal, p_drop, seed, dtype, stream);
}
