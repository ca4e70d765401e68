// bf16 attention on the matrix cores (v_mfma_f32_16x16x32_bf16), forward and backward, for
// head sizes dk == dv in {32, 64, 128}.
//
// Replaces ScaledDotProductAttention.forward (lib/transformer/Modules.py:17-28) and its
// autograd backward, plus the head split / merge copies of MultiHeadAttention
// (lib/transformer/SubLayers.py:50-60): heads are addressed in place inside the (T, h*d)
// projection buffers.
//
// One scheme serves all three kernels.  A wave OWNS 16 rows (query rows for forward and dQ,
// key rows for dK/dV) whose operand fragments it keeps in registers, and sweeps the OTHER
// sequence in 64-row tiles staged in LDS.  Every product is issued as
//     D[rho][c] = sum_k X[rho][k] * Y[c][k],   X from LDS, Y from registers,
// so the accumulator has the wave's own row on the LANE (c = lane & 15) and the other row in
// the registers (rho = 16*tile + 4*(lane>>4) + r).  Consequences:
//   * a softmax row is 16 values per lane plus two cross-lane steps (xor 16, 32);
//   * the accumulator of the first product (scores / probabilities / dS), converted to bf16,
//     IS the Y operand of the second product with no lane movement: the contraction order
//     within a 32-deep MFMA step is (lane>>4, j) -> other row 32*s + 16*(j>>2) + 4*(lane>>4)
//     + (j&3), and the X operand of that product (V^T, K^T, dO^T, Q^T) is fetched with the
//     transposing ds_read_b64_tr_b16 in the same order.
// Scores never touch HBM; backward recomputes probabilities from the forward's row
// log-sum-exp.  dQ and dK/dV are separate launches (own rows = queries resp. keys), so no
// atomics are needed and results are bitwise reproducible.
#include "attention.h"
#include <stdlib.h>

using namespace skattn;

namespace {

constexpr int OT = 64;  // other-sequence rows per LDS tile

template <int D> struct Tile {
  static constexpr int LD = D + 8;  // padded row (elements): 16-byte aligned rows, spreads banks
  bf16_t a[OT][LD];
};

__device__ __forceinline__ bf16x8_t ld_frag_global(const bf16_t* p) { return *(const bf16x8_t*)p; }

// X fragment by rows: lane (rho = l&15, g = l>>4) <- tile[row0 + rho][32*s + 8*g .. +7]
template <int D>
__device__ __forceinline__ bf16x8_t x_rows(const Tile<D>& t, int row0, int s, int lane) {
  return *(const bf16x8_t*)&t.a[row0 + (lane & 15)][32 * s + 8 * (lane >> 4)];
}

// X fragment by columns (transposed): lane (rho, g) <- tile[32*s2 + 16*(j>>2) + 4*g + (j&3)][col0 + rho]
template <int D>
__device__ __forceinline__ bf16x8_t x_cols(const Tile<D>& t, int col0, int s2, int lane) {
  typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
  const int i = lane & 15, g = lane >> 4;
  const int row = 32 * s2 + 4 * g + (i >> 2), col = col0 + 4 * (i & 3);
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)&t.a[row][col]);
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)&t.a[row + 16][col]);
  s16x8_t f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, f);
}

// accumulators of other-row tiles 2*s2 and 2*s2+1 -> Y fragment of MFMA step s2
__device__ __forceinline__ bf16x8_t pack_y(const f32x4_t& a, const f32x4_t& b) {
  bf16x8_t f;
  f[0] = (__bf16)a[0]; f[1] = (__bf16)a[1]; f[2] = (__bf16)a[2]; f[3] = (__bf16)a[3];
  f[4] = (__bf16)b[0]; f[5] = (__bf16)b[1]; f[6] = (__bf16)b[2]; f[7] = (__bf16)b[3];
  return f;
}

// Stage 64 rows x D columns (bf16) of a (row stride ld) matrix into an LDS tile; rows beyond
// `nrows` are zero-filled.
template <int D>
__device__ __forceinline__ void stage_tile(Tile<D>& t, const bf16_t* base, int64_t ld, int r0, int nrows) {
  constexpr int CH = D / 8;  // 16-byte chunks per row
  for (int e = threadIdx.x; e < OT * CH; e += 256) {
    const int r = e / CH, c = e - r * CH;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r0 + r < nrows) v = *(const uint4*)(base + (int64_t)(r0 + r) * ld + c * 8);
    *(uint4*)&t.a[r][c * 8] = v;
  }
}

__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// ------------------------------------------------------------------------------ forward
// A row of D output columns leaves the multiplications as w[t] = the four bf16 of columns 16 t + 4 g .. of lane
// (row, g): stored as they are, an instruction writes four 8-byte pieces per row and a 128-byte line comes together
// from sixteen writes in the L2.  Two exchange rounds among the row's four lanes per 32 columns — piece index against
// lane bit 0, then the two lane bits against each other — leave lane g with columns 32 u + 8 g .. + 7, and an
// instruction writes 64 consecutive bytes per row.  All four lanes of a row call this together.
// piece(t): the row's w[t], made on demand (the fp32 values it comes from die as the loop goes).
template <int D, class F>
__device__ __forceinline__ void store_row_wide(bf16_t* out, F&& piece, int lane) {
  const int g = lane >> 4;
  const bool b = g & 1;
  const int src = (lane & 15) + 16 * (((g & 1) << 1) | (g >> 1));
#pragma unroll
  for (int u = 0; u < D / 32; ++u) {
    const uint2 p0 = piece(2 * u), p1 = piece(2 * u + 1);
    const uint2 send = b ? p0 : p1;
    uint2 recv;
    recv.x = __shfl_xor(send.x, 16, 64);
    recv.y = __shfl_xor(send.y, 16, 64);
    const uint2 r0 = b ? recv : p0, r1 = b ? p1 : recv;
    uint4 q;
    q.x = __shfl(r0.x, src, 64);
    q.y = __shfl(r0.y, src, 64);
    q.z = __shfl(r1.x, src, 64);
    q.w = __shfl(r1.y, src, 64);
    if (out) *(uint4*)(out + 32 * u + 8 * g) = q;
  }
}

template <int D, class F>
__device__ __forceinline__ void store_row(bf16_t* out, bool row_ok, bool wide, F&& piece, int lane) {
  if (wide) {   // kernel-uniform
    store_row_wide<D>(row_ok ? out : nullptr, piece, lane);
  } else if (row_ok) {
    const int g = lane >> 4;
#pragma unroll
    for (int t = 0; t < D / 16; ++t) *(uint2*)(out + 16 * t + 4 * g) = piece(t);
  }
}

__device__ __forceinline__ uint2 pack4_bf16(const f32x4_t v, float scale) {
  uint2 w;
  bf16_t* e = (bf16_t*)&w;
#pragma unroll
  for (int r = 0; r < 4; ++r) e[r] = f32_to_bf16(v[r] * scale);
  return w;
}

template <int D>
__global__ __launch_bounds__(256, (D <= 64 ? 6 : 4)) void attn_fwd_mfma_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) Tile<D> Ks;
  __shared__ __attribute__((aligned(16))) Tile<D> Vs;
  __shared__ uint8_t kmask[OT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const int c = lane & 15, g = lane >> 4;
  const int q = blockIdx.x * 64 + wave * 16 + c;
  const Seq sq = seq_of(p, b);
  const int Lq = sq.Lq, Lk = sq.Lk;
  if ((int)blockIdx.x * 64 >= Lq) return;  // block-uniform (shorter sentence of a packed batch)
  const int qrow = min(q, Lq - 1);
  const bf16_t* Q = (const bf16_t*)p.Q + (sq.qbase + qrow) * p.ldq + h * D;
  const bf16_t* K = (const bf16_t*)p.K + sq.kbase * p.ldk + h * D;
  const bf16_t* V = (const bf16_t*)p.V + sq.kbase * p.ldv + h * D;
  bf16x8_t fq[D / 32];
#pragma unroll
  for (int s = 0; s < D / 32; ++s) fq[s] = ld_frag_global(Q + 32 * s + 8 * g);

  float m = -INFINITY, l = 0.f;
  f32x4_t o[D / 16];
#pragma unroll
  for (int t = 0; t < D / 16; ++t) o[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // 16-row groups beyond the sentence are skipped (wave-uniform branches; every wave meets every barrier)
  const bool wave_has_q = (int)blockIdx.x * 64 + wave * 16 < Lq;
  for (int j0 = 0; j0 < Lk; j0 += OT) {
    __syncthreads();
    stage_tile<D>(Ks, K, p.ldk, j0, Lk);
    stage_tile<D>(Vs, V, p.ldv, j0, Lk);
    if (threadIdx.x < OT) {
      const int j = j0 + threadIdx.x;
      kmask[threadIdx.x] = (j >= Lk) || (p.key_pad && p.key_pad[(int64_t)b * p.Lk + j]);
    }
    __syncthreads();
    if (!wave_has_q) continue;
    const int nkt = min(4, (Lk - j0 + 15) >> 4);  // key tiles of this pass that hold keys
    f32x4_t sc[4];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      sc[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (kt < nkt) {
#pragma unroll
        for (int s = 0; s < D / 32; ++s)
          sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_rows<D>(Ks, 16 * kt, s, lane), fq[s], sc[kt], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int jl = 16 * kt + 4 * g + r;
          const bool msk = kmask[jl] || (p.causal && j0 + jl > q);
          sc[kt][r] = msk ? -INFINITY : sc[kt][r] * p.scale;
          mx = fmaxf(mx, sc[kt][r]);
        }
      }
    }
    const float mn = fmaxf(m, quad_max(mx));
    const bool dead = mn == -INFINITY;
    const float alpha = dead ? 1.f : __expf(m - mn);
    float rs = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      if (kt < nkt) {
        const uint32_t km = keep4(p, b, h, q, j0 + 16 * kt + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pr = dead ? 0.f : __expf(sc[kt][r] - mn);
          rs += pr;
          if (p.drop_thr) pr = (km >> r) & 1 ? pr * p.keep_scale : 0.f;
          sc[kt][r] = pr;
        }
      }
    }
    l = l * alpha + quad_sum(rs);
    m = mn;
    const bf16x8_t y0 = pack_y(sc[0], sc[1]), y1 = pack_y(sc[2], sc[3]);
#pragma unroll
    for (int t = 0; t < D / 16; ++t) {
      o[t] *= alpha;
      o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols<D>(Vs, 16 * t, 0, lane), y0, o[t], 0, 0, 0);
      if (nkt > 2)
        o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols<D>(Vs, 16 * t, 1, lane), y1, o[t], 0, 0, 0);
    }
  }
  if (q < Lq) {
    // (8-byte pieces here: this kernel runs six waves to a SIMD on 80 registers, and the exchange of the wide form
    // put four of them into scratch memory — 33 -> 36 us per launch)
    const float inv = 1.0f / l;  // fully masked row: 0 * inf = NaN, like the reference's softmax
    bf16_t* O = (bf16_t*)p.Oout + (sq.qbase + q) * p.ldo + h * D;
#pragma unroll
    for (int t = 0; t < D / 16; ++t) {
      uint2 w;
      bf16_t* e = (bf16_t*)&w;
#pragma unroll
      for (int r = 0; r < 4; ++r) e[r] = f32_to_bf16(o[t][r] * inv);
      *(uint2*)(O + 16 * t + 4 * g) = w;
    }
    if (g == 0) p.LSE[((int64_t)b * p.H + h) * p.Lq + q] = m + __logf(l);
  }
}

// ------------------------------------------------------------------------------ dQ (+ delta)
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dq_mfma_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) Tile<D> Ks;
  __shared__ __attribute__((aligned(16))) Tile<D> Vs;
  __shared__ uint8_t kmask[OT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const int c = lane & 15, g = lane >> 4;
  const int q = blockIdx.x * 64 + wave * 16 + c;
  const Seq sq = seq_of(p, b);
  const int Lq = sq.Lq, Lk = sq.Lk;
  if ((int)blockIdx.x * 64 >= Lq) return;  // block-uniform
  const int qrow = min(q, Lq - 1);
  const int64_t qoff = sq.qbase + qrow;
  const bf16_t* Q = (const bf16_t*)p.Q + qoff * p.ldq + h * D;
  const bf16_t* dO = (const bf16_t*)p.dO + qoff * p.ldo + h * D;
  const bf16_t* O = (const bf16_t*)p.O + qoff * p.ldo + h * D;
  const bf16_t* K = (const bf16_t*)p.K + sq.kbase * p.ldk + h * D;
  const bf16_t* V = (const bf16_t*)p.V + sq.kbase * p.ldv + h * D;
  bf16x8_t fq[D / 32], fdo[D / 32];
  float dl = 0.f;
#pragma unroll
  for (int s = 0; s < D / 32; ++s) {
    fq[s] = ld_frag_global(Q + 32 * s + 8 * g);
    fdo[s] = ld_frag_global(dO + 32 * s + 8 * g);
    bf16x8_t fo = ld_frag_global(O + 32 * s + 8 * g);
#pragma unroll
    for (int j = 0; j < 8; ++j) dl += (float)fdo[s][j] * (float)fo[j];
  }
  dl = quad_sum(dl);  // delta = rowsum(dO * O)
  const int64_t si = ((int64_t)b * p.H + h) * p.Lq + qrow;
  const float lse = p.LSE[si];
  if (q < Lq && g == 0) p.delta[si] = dl;

  f32x4_t dq[D / 16];
#pragma unroll
  for (int t = 0; t < D / 16; ++t) dq[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  for (int j0 = 0; j0 < Lk; j0 += OT) {
    __syncthreads();
    stage_tile<D>(Ks, K, p.ldk, j0, Lk);
    stage_tile<D>(Vs, V, p.ldv, j0, Lk);
    if (threadIdx.x < OT) {
      const int j = j0 + threadIdx.x;
      kmask[threadIdx.x] = (j >= Lk) || (p.key_pad && p.key_pad[(int64_t)b * p.Lk + j]);
    }
    __syncthreads();
    f32x4_t ds[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      f32x4_t sc = f32x4_t{0.f, 0.f, 0.f, 0.f}, dp = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < D / 32; ++s) {
        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_rows<D>(Ks, 16 * kt, s, lane), fq[s], sc, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_rows<D>(Vs, 16 * kt, s, lane), fdo[s], dp, 0, 0, 0);
      }
      const uint32_t km = keep4(p, b, h, q, j0 + 16 * kt + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jl = 16 * kt + 4 * g + r;
        const bool msk = kmask[jl] || (p.causal && j0 + jl > q) || q >= Lq;
        float pr = msk ? 0.f : __expf(sc[r] * p.scale - lse);
        float dpv = dp[r];
        if (p.drop_thr) dpv = (km >> r) & 1 ? dpv * p.keep_scale : 0.f;
        ds[kt][r] = pr * (dpv - dl) * p.scale;
      }
    }
    const bf16x8_t y0 = pack_y(ds[0], ds[1]), y1 = pack_y(ds[2], ds[3]);
#pragma unroll
    for (int t = 0; t < D / 16; ++t) {
      dq[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols<D>(Ks, 16 * t, 0, lane), y0, dq[t], 0, 0, 0);
      dq[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols<D>(Ks, 16 * t, 1, lane), y1, dq[t], 0, 0, 0);
    }
  }
  {
    store_row<D>((bf16_t*)p.dQ + (sq.qbase + min(q, Lq - 1)) * p.lddq + h * D, q < Lq, p.wide_store && !(p.lddq & 7),
                 [&](int t) { return pack4_bf16(dq[t], 1.0f); }, lane);
  }
}

// ------------------------------------------------------------------------------ dK, dV
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) Tile<D> Qs;
  __shared__ __attribute__((aligned(16))) Tile<D> dOs;
  __shared__ float s_lse[OT], s_dl[OT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const int c = lane & 15, g = lane >> 4;
  const int j = blockIdx.x * 64 + wave * 16 + c;  // own key row
  const Seq sq = seq_of(p, b);
  const int Lq = sq.Lq, Lk = sq.Lk;
  if ((int)blockIdx.x * 64 >= Lk) return;  // block-uniform
  const int jrow = min(j, Lk - 1);
  const bf16_t* K = (const bf16_t*)p.K + (sq.kbase + jrow) * p.ldk + h * D;
  const bf16_t* V = (const bf16_t*)p.V + (sq.kbase + jrow) * p.ldv + h * D;
  const bf16_t* Q = (const bf16_t*)p.Q + sq.qbase * p.ldq + h * D;
  const bf16_t* dO = (const bf16_t*)p.dO + sq.qbase * p.ldo + h * D;
  bf16x8_t fk[D / 32], fv[D / 32];
#pragma unroll
  for (int s = 0; s < D / 32; ++s) {
    fk[s] = ld_frag_global(K + 32 * s + 8 * g);
    fv[s] = ld_frag_global(V + 32 * s + 8 * g);
  }
  const bool jmask = (j >= Lk) || (p.key_pad && p.key_pad[(int64_t)b * p.Lk + jrow]);

  f32x4_t dk[D / 16], dv[D / 16];
#pragma unroll
  for (int t = 0; t < D / 16; ++t) { dk[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

  for (int i0 = 0; i0 < Lq; i0 += OT) {
    __syncthreads();
    stage_tile<D>(Qs, Q, p.ldq, i0, Lq);
    stage_tile<D>(dOs, dO, p.ldo, i0, Lq);
    if (threadIdx.x < OT) {
      const int i = min(i0 + (int)threadIdx.x, Lq - 1);
      const int64_t si = ((int64_t)b * p.H + h) * p.Lq + i;
      s_lse[threadIdx.x] = p.LSE[si];
      s_dl[threadIdx.x] = p.delta[si];
    }
    __syncthreads();
    f32x4_t pd[4], ds[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      f32x4_t sc = f32x4_t{0.f, 0.f, 0.f, 0.f}, dp = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < D / 32; ++s) {
        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_rows<D>(Qs, 16 * qt, s, lane), fk[s], sc, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_rows<D>(dOs, 16 * qt, s, lane), fv[s], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int il = 16 * qt + 4 * g + r, i = i0 + il;
        const bool msk = jmask || i >= Lq || (p.causal && j > i);
        float pr = msk ? 0.f : __expf(sc[r] * p.scale - s_lse[il]);
        float dpv = dp[r], pdv = pr;
        if (p.drop_thr) {
          const bool kp = keep(p, b, h, i, j);
          dpv = kp ? dpv * p.keep_scale : 0.f;
          pdv = kp ? pr * p.keep_scale : 0.f;
        }
        pd[qt][r] = pdv;
        ds[qt][r] = pr * (dpv - s_dl[il]) * p.scale;
      }
    }
    const bf16x8_t yp0 = pack_y(pd[0], pd[1]), yp1 = pack_y(pd[2], pd[3]);
    const bf16x8_t ys0 = pack_y(ds[0], ds[1]), ys1 = pack_y(ds[2], ds[3]);
#pragma unroll
    for (int t = 0; t < D / 16; ++t) {
      dv[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols<D>(dOs, 16 * t, 0, lane), yp0, dv[t], 0, 0, 0);
      dv[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols<D>(dOs, 16 * t, 1, lane), yp1, dv[t], 0, 0, 0);
      dk[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols<D>(Qs, 16 * t, 0, lane), ys0, dk[t], 0, 0, 0);
      dk[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols<D>(Qs, 16 * t, 1, lane), ys1, dk[t], 0, 0, 0);
    }
  }
  {
    const bool wide = p.wide_store && !((p.lddk | p.lddv) & 7);
    store_row<D>((bf16_t*)p.dK + (sq.kbase + min(j, Lk - 1)) * p.lddk + h * D, j < Lk, wide,
                 [&](int t) { return pack4_bf16(dk[t], 1.0f); }, lane);
    store_row<D>((bf16_t*)p.dV + (sq.kbase + min(j, Lk - 1)) * p.lddv + h * D, j < Lk, wide,
                 [&](int t) { return pack4_bf16(dv[t], 1.0f); }, lane);
  }
}

// ------------------------------------------------------------------------------ dQ, dK, dV in one pass
// Sequences of at most 64 positions (one tile on either side): one block per (batch, head) stages
// Q, K, V and dO once and produces all three gradients — phase A with the waves' own QUERY rows
// (delta, dS, dQ), phase B with their own KEY rows (dV, dK) — instead of two launches that each
// re-read every operand from HBM.
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_fused_mfma_kernel(AttnParams p) {
  // K and V tiles; after phase A their storage holds P and dS (64 x 64, query-major)
  __shared__ __attribute__((aligned(16))) char ks_raw[sizeof(Tile<(D > 64 ? D : 64)>)];
  __shared__ __attribute__((aligned(16))) char vs_raw[sizeof(Tile<(D > 64 ? D : 64)>)];
  Tile<D>& Ks = *(Tile<D>*)ks_raw;
  Tile<D>& Vs = *(Tile<D>*)vs_raw;
  Tile<64>& Ps = *(Tile<64>*)ks_raw;
  Tile<64>& dSs = *(Tile<64>*)vs_raw;
  __shared__ __attribute__((aligned(16))) Tile<D> Qs;
  __shared__ __attribute__((aligned(16))) Tile<D> dOs;
  __shared__ uint8_t kmask[OT];
  __shared__ float s_lse[OT], s_dl[OT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const int c = lane & 15, g = lane >> 4;
  const int own = wave * 16 + c;  // own query row (phase A) / own key row (phase B)
  const Seq sq = seq_of(p, b);
  const int Lq = sq.Lq, Lk = sq.Lk;
  if (Lq <= 0 || Lk <= 0) return;  // block-uniform
  stage_tile<D>(Ks, (const bf16_t*)p.K + sq.kbase * p.ldk + h * D, p.ldk, 0, Lk);
  stage_tile<D>(Vs, (const bf16_t*)p.V + sq.kbase * p.ldv + h * D, p.ldv, 0, Lk);
  stage_tile<D>(Qs, (const bf16_t*)p.Q + sq.qbase * p.ldq + h * D, p.ldq, 0, Lq);
  stage_tile<D>(dOs, (const bf16_t*)p.dO + sq.qbase * p.ldo + h * D, p.ldo, 0, Lq);
  if (threadIdx.x < OT) {
    const int j = threadIdx.x;
    kmask[j] = (j >= Lk) || (p.key_pad && p.key_pad[(int64_t)b * p.Lk + j]);
  }
  // delta = rowsum(dO * O) and LSE of the own query row
  {
    const int qrow = min(own, Lq - 1);
    const bf16_t* O = (const bf16_t*)p.O + (sq.qbase + qrow) * p.ldo + h * D;
    const bf16_t* dO = (const bf16_t*)p.dO + (sq.qbase + qrow) * p.ldo + h * D;
    float dl = 0.f;
#pragma unroll
    for (int s = 0; s < D / 32; ++s) {
      const bf16x8_t fo = ld_frag_global(O + 32 * s + 8 * g), fd = ld_frag_global(dO + 32 * s + 8 * g);
#pragma unroll
      for (int j = 0; j < 8; ++j) dl += (float)fd[j] * (float)fo[j];
    }
    dl = quad_sum(dl);
    if (g == 0) {
      s_dl[own] = dl;
      s_lse[own] = p.LSE[((int64_t)b * p.H + h) * p.Lq + qrow];
    }
  }
  __syncthreads();

  // Sentences are usually shorter than the 64-row tile: 16-row groups beyond the sentence are
  // skipped — key tiles kt >= nkt in phase A, query halves beyond Lq in phase B, and whole waves
  // whose own rows lie beyond it.  All branches are wave-uniform (MFMA and the transposing reads
  // need every lane of an executing wave), and every wave still meets every barrier.
  const int nkt = (Lk + 15) >> 4;                 // key tiles that hold keys
  const bool two_q_halves = Lq > 32, two_k_halves = Lk > 32;
  const bool wave_has_q = 16 * wave < Lq, wave_has_k = 16 * wave < Lk;

  // ---- phase A: own query rows -> dQ; P and dS of the row for phase B
  bf16x8_t y0, y1, z0, z1;  // packed dS (y) and P with dropout scaling (z): key tiles {0,1} and {2,3}
  {
    const int q = own;
    f32x4_t ds[4], pd[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) ds[kt] = pd[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (wave_has_q) {
      bf16x8_t fq[D / 32], fdo[D / 32];
#pragma unroll
      for (int s = 0; s < D / 32; ++s) {
        fq[s] = *(const bf16x8_t*)&Qs.a[q][32 * s + 8 * g];
        fdo[s] = *(const bf16x8_t*)&dOs.a[q][32 * s + 8 * g];
      }
      const float lse = s_lse[q], dl = s_dl[q];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        if (kt < nkt) {
          f32x4_t sc = f32x4_t{0.f, 0.f, 0.f, 0.f}, dp = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < D / 32; ++s) {
            sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_rows<D>(Ks, 16 * kt, s, lane), fq[s], sc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_rows<D>(Vs, 16 * kt, s, lane), fdo[s], dp, 0, 0, 0);
          }
          const uint32_t km = keep4(p, b, h, q, 16 * kt + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int jl = 16 * kt + 4 * g + r;
            const bool msk = kmask[jl] || (p.causal && jl > q) || q >= Lq;
            const float pr = msk ? 0.f : __expf(sc[r] * p.scale - lse);
            float dpv = dp[r], pdv = pr;
            if (p.drop_thr) {
              const bool kp = (km >> r) & 1;
              dpv = kp ? dpv * p.keep_scale : 0.f;
              pdv = kp ? pr * p.keep_scale : 0.f;
            }
            pd[kt][r] = pdv;
            ds[kt][r] = pr * (dpv - dl) * p.scale;
          }
        }
      }
    }
    y0 = pack_y(ds[0], ds[1]); y1 = pack_y(ds[2], ds[3]);
    z0 = pack_y(pd[0], pd[1]); z1 = pack_y(pd[2], pd[3]);
    if (wave_has_q) {
      f32x4_t dq[D / 16];
#pragma unroll
      for (int t = 0; t < D / 16; ++t) {
        dq[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        dq[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols<D>(Ks, 16 * t, 0, lane), y0, dq[t], 0, 0, 0);
        if (two_k_halves)
          dq[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols<D>(Ks, 16 * t, 1, lane), y1, dq[t], 0, 0, 0);
      }
      {  // the multiplications above ran with every lane; only the stores are guarded
        store_row<D>((bf16_t*)p.dQ + (sq.qbase + min(q, Lq - 1)) * p.lddq + h * D, q < Lq, p.wide_store && !(p.lddq & 7),
                     [&](int t) { return pack4_bf16(dq[t], 1.0f); }, lane);
      }
    }
    // P (with the dropout scaling) and dS of the own query row replace the K and V tiles, which
    // no wave reads any more after the barrier; phase B then takes them as transposed operands
    // instead of recomputing scores, exponentials and dropout decisions for its key rows
    __syncthreads();
    {
      const uint4 zp0 = __builtin_bit_cast(uint4, z0), zp1 = __builtin_bit_cast(uint4, z1);
      const uint4 yd0 = __builtin_bit_cast(uint4, y0), yd1 = __builtin_bit_cast(uint4, y1);
      *(uint2*)&Ps.a[q][4 * g] = make_uint2(zp0.x, zp0.y);
      *(uint2*)&Ps.a[q][16 + 4 * g] = make_uint2(zp0.z, zp0.w);
      *(uint2*)&Ps.a[q][32 + 4 * g] = make_uint2(zp1.x, zp1.y);
      *(uint2*)&Ps.a[q][48 + 4 * g] = make_uint2(zp1.z, zp1.w);
      *(uint2*)&dSs.a[q][4 * g] = make_uint2(yd0.x, yd0.y);
      *(uint2*)&dSs.a[q][16 + 4 * g] = make_uint2(yd0.z, yd0.w);
      *(uint2*)&dSs.a[q][32 + 4 * g] = make_uint2(yd1.x, yd1.y);
      *(uint2*)&dSs.a[q][48 + 4 * g] = make_uint2(yd1.z, yd1.w);
    }
    __syncthreads();
  }

  // ---- phase B: own key rows -> dK = dS^T Q, dV = P^T dO
  if (wave_has_k) {
    const int j = own;
    const bf16x8_t yp0 = x_cols<64>(Ps, 16 * wave, 0, lane), ys0 = x_cols<64>(dSs, 16 * wave, 0, lane);
    bf16x8_t yp1 = yp0, ys1 = ys0;
    if (two_q_halves) { yp1 = x_cols<64>(Ps, 16 * wave, 1, lane); ys1 = x_cols<64>(dSs, 16 * wave, 1, lane); }
    f32x4_t dk[D / 16], dv[D / 16];
#pragma unroll
    for (int t = 0; t < D / 16; ++t) {
      dv[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      dk[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      dv[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols<D>(dOs, 16 * t, 0, lane), yp0, dv[t], 0, 0, 0);
      dk[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols<D>(Qs, 16 * t, 0, lane), ys0, dk[t], 0, 0, 0);
      if (two_q_halves) {
        dv[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols<D>(dOs, 16 * t, 1, lane), yp1, dv[t], 0, 0, 0);
        dk[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols<D>(Qs, 16 * t, 1, lane), ys1, dk[t], 0, 0, 0);
      }
    }
    {
      const bool wide = p.wide_store && !((p.lddk | p.lddv) & 7);
      store_row<D>((bf16_t*)p.dK + (sq.kbase + min(j, Lk - 1)) * p.lddk + h * D, j < Lk, wide,
                   [&](int t) { return pack4_bf16(dk[t], 1.0f); }, lane);
      store_row<D>((bf16_t*)p.dV + (sq.kbase + min(j, Lk - 1)) * p.lddv + h * D, j < Lk, wide,
                   [&](int t) { return pack4_bf16(dv[t], 1.0f); }, lane);
    }
  }
}

// ------------------------------------------------------------------------------ the same for heads of 64, leaner
// The one-pass kernel above moves 25 KB per (batch, head) and does a few hundred matrix instructions on them: its time
// is the sum of a memory phase, an arithmetic phase and a store phase that only other blocks on the CU can overlap.
// This form makes each phase cheaper and lets a fifth block share the CU:
//   * tiles are 64 x 64 bf16 without padding (8 KB; four of them are exactly a fifth of the LDS), their 16-byte
//     chunks XOR-swizzled with the row number so that fragment reads by rows and by columns spread over the banks;
//   * operands go from memory straight into the tiles (direct-to-LDS loads, eight rows per instruction, the swizzle
//     applied to the SOURCE address): no staging registers, no address arithmetic per chunk, no LDS writes;
//   * rows beyond the sentence are not zero-filled but read the sentence's last row again — finite values that only
//     ever meet probabilities and score gradients that the masks have set to exactly zero;
//   * the key mask is a 64-bit ballot in scalar registers, delta and the log-sum-exp stay in the lane that owns the row.
namespace f64 {
constexpr int ROWB = 128, TILE = 64 * ROWB;   // bytes per row / per tile

__device__ __forceinline__ int sw(int row, int chunk) { return row * ROWB + ((chunk ^ (row & 7)) << 4); }

__device__ __forceinline__ bf16x8_t x_rows(const char* t, int row0, int s, int lane) {
  return *(const bf16x8_t*)(t + sw(row0 + (lane & 15), 4 * s + (lane >> 4)));
}
__device__ __forceinline__ bf16x8_t x_cols(const char* t, int col0, int s2, int lane) {
  typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
  const int i = lane & 15, g = lane >> 4;
  const int row = 32 * s2 + 4 * g + (i >> 2), col = col0 + 4 * (i & 3);
  const char* a = t + sw(row, col >> 3) + (col & 7) * 2;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)a);
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(a + 16 * ROWB));   // row + 16: same swizzle key
  s16x8_t f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, f);
}
// rows 16 wave + 8 j .. + 7 (j = 0, 1) of a 64-row tile from a matrix with `L` valid rows of stride ld
__device__ __forceinline__ void dma_tile(char* t, const bf16_t* base, int64_t ld, int L, int wave, int lane) {
  typedef __attribute__((address_space(1))) void* gbl_ptr_t;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = 2 * wave + j, row = 8 * idx + (lane >> 3), pos = lane & 7;
    const bf16_t* src = base + (int64_t)min(row, L - 1) * ld + 8 * (pos ^ (row & 7));
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(t + idx * 1024), 16, 0, 0);
  }
}

__global__ __launch_bounds__(256, 5) void attn_bwd_fused_kernel(AttnParams p) {
  constexpr int D = 64;
  __shared__ __attribute__((aligned(1024))) char lds[4 * TILE];
  char* Ks = lds;              // after phase A: P (with the dropout scaling), query-major
  char* Vs = lds + TILE;       // after phase A: dS
  char* Qs = lds + 2 * TILE;
  char* dOs = lds + 3 * TILE;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.z, h = blockIdx.y;
  const int c = lane & 15, g = lane >> 4;
  const int own = wave * 16 + c;  // own query row (phase A) / own key row (phase B)
  const Seq sq = seq_of(p, b);
  const int Lq = sq.Lq, Lk = sq.Lk;
  if (Lq <= 0 || Lk <= 0) return;  // block-uniform
  dma_tile(Ks, (const bf16_t*)p.K + sq.kbase * p.ldk + h * D, p.ldk, Lk, wave, lane);
  dma_tile(Vs, (const bf16_t*)p.V + sq.kbase * p.ldv + h * D, p.ldv, Lk, wave, lane);
  dma_tile(Qs, (const bf16_t*)p.Q + sq.qbase * p.ldq + h * D, p.ldq, Lq, wave, lane);
  dma_tile(dOs, (const bf16_t*)p.dO + sq.qbase * p.ldo + h * D, p.ldo, Lq, wave, lane);
  // the own query row's log-sum-exp, the key mask
  const int qrow = min(own, Lq - 1);
  const float lse = p.LSE[((int64_t)b * p.H + h) * p.Lq + qrow];
  bool kmsk = lane >= Lk;
  if (p.key_pad) kmsk = kmsk || p.key_pad[(int64_t)b * p.Lk + min(lane, Lk - 1)];
  const uint64_t kmask = __ballot(kmsk);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int nkt = (Lk + 15) >> 4;                 // key tiles that hold keys
  const bool two_q_halves = Lq > 32, two_k_halves = Lk > 32;
  const bool wave_has_q = 16 * wave < Lq, wave_has_k = 16 * wave < Lk;

  // ---- phase A: own query rows -> dQ; P and dS of the row for phase B
  bf16x8_t y0, y1, z0, z1;  // packed dS (y) and P with dropout scaling (z): key tiles {0,1} and {2,3}
  {
    const int q = own;
    // delta = rowsum(dO * O) without reading O: O = sum_j Pd_j V_j (Pd: probabilities after dropout), so
    // dO . O = sum_j Pd_j (dO . V_j) — both factors are in the lane's registers; the row's 64 keys sit in the four
    // lanes of its quad.  dS = scale (Pd (dO . V) - P delta).
    f32x4_t ds[4], pr[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) ds[kt] = pr[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    z0 = z1 = bf16x8_t{};
    if (wave_has_q) {
      bf16x8_t fq[2], fdo[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        fq[s] = *(const bf16x8_t*)(Qs + sw(q, 4 * s + g));
        fdo[s] = *(const bf16x8_t*)(dOs + sw(q, 4 * s + g));
      }
      float dl = 0.f;
      f32x4_t pd[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        pd[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (kt < nkt) {
          f32x4_t sc = f32x4_t{0.f, 0.f, 0.f, 0.f}, dp = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_rows(Ks, 16 * kt, s, lane), fq[s], sc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_rows(Vs, 16 * kt, s, lane), fdo[s], dp, 0, 0, 0);
          }
          const uint32_t km = keep4(p, b, h, q, 16 * kt + 4 * g);
          const uint32_t mk4 = (uint32_t)(kmask >> (16 * kt + 4 * g)) & 15u;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int jl = 16 * kt + 4 * g + r;
            const bool msk = ((mk4 >> r) & 1) || (p.causal && jl > q) || q >= Lq;
            const float prob = msk ? 0.f : __expf(sc[r] * p.scale - lse);
            float pdv = prob;
            if (p.drop_thr) pdv = (km >> r) & 1 ? prob * p.keep_scale : 0.f;
            const float t = pdv * dp[r];
            pd[kt][r] = pdv;
            pr[kt][r] = prob;
            ds[kt][r] = t;
            dl += t;
          }
        }
      }
      z0 = pack_y(pd[0], pd[1]); z1 = pack_y(pd[2], pd[3]);
      dl = quad_sum(dl);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[kt][r] = (ds[kt][r] - pr[kt][r] * dl) * p.scale;
    }
    y0 = pack_y(ds[0], ds[1]); y1 = pack_y(ds[2], ds[3]);
    if (wave_has_q) {
      f32x4_t dq[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        dq[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        dq[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols(Ks, 16 * t, 0, lane), y0, dq[t], 0, 0, 0);
        if (two_k_halves)
          dq[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols(Ks, 16 * t, 1, lane), y1, dq[t], 0, 0, 0);
      }
      store_row<D>((bf16_t*)p.dQ + (sq.qbase + min(q, Lq - 1)) * p.lddq + h * D, q < Lq, p.wide_store && !(p.lddq & 7),
                   [&](int t) { return pack4_bf16(dq[t], 1.0f); }, lane);
    }
    // P and dS of the own query row replace the K and V tiles, which no wave reads any more after the barrier
    __syncthreads();
    {
      const uint4 zp0 = __builtin_bit_cast(uint4, z0), zp1 = __builtin_bit_cast(uint4, z1);
      const uint4 yd0 = __builtin_bit_cast(uint4, y0), yd1 = __builtin_bit_cast(uint4, y1);
      // columns 4 g, 16 + 4 g, 32 + 4 g, 48 + 4 g of row q: chunk (col >> 3), half (g & 1)
      const int half = (g & 1) * 8;
      *(uint2*)(Ks + sw(q, (g >> 1)) + half) = make_uint2(zp0.x, zp0.y);
      *(uint2*)(Ks + sw(q, 2 + (g >> 1)) + half) = make_uint2(zp0.z, zp0.w);
      *(uint2*)(Ks + sw(q, 4 + (g >> 1)) + half) = make_uint2(zp1.x, zp1.y);
      *(uint2*)(Ks + sw(q, 6 + (g >> 1)) + half) = make_uint2(zp1.z, zp1.w);
      *(uint2*)(Vs + sw(q, (g >> 1)) + half) = make_uint2(yd0.x, yd0.y);
      *(uint2*)(Vs + sw(q, 2 + (g >> 1)) + half) = make_uint2(yd0.z, yd0.w);
      *(uint2*)(Vs + sw(q, 4 + (g >> 1)) + half) = make_uint2(yd1.x, yd1.y);
      *(uint2*)(Vs + sw(q, 6 + (g >> 1)) + half) = make_uint2(yd1.z, yd1.w);
    }
    __syncthreads();
  }

  // ---- phase B: own key rows -> dK = dS^T Q, dV = P^T dO
  if (wave_has_k) {
    const int j = own;
    const bf16x8_t yp0 = x_cols(Ks, 16 * wave, 0, lane), ys0 = x_cols(Vs, 16 * wave, 0, lane);
    bf16x8_t yp1 = yp0, ys1 = ys0;
    if (two_q_halves) { yp1 = x_cols(Ks, 16 * wave, 1, lane); ys1 = x_cols(Vs, 16 * wave, 1, lane); }
    f32x4_t dk[4], dv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      dv[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      dk[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      dv[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols(dOs, 16 * t, 0, lane), yp0, dv[t], 0, 0, 0);
      dk[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols(Qs, 16 * t, 0, lane), ys0, dk[t], 0, 0, 0);
      if (two_q_halves) {
        dv[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols(dOs, 16 * t, 1, lane), yp1, dv[t], 0, 0, 0);
        dk[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols(Qs, 16 * t, 1, lane), ys1, dk[t], 0, 0, 0);
      }
    }
    const bool wide = p.wide_store && !((p.lddk | p.lddv) & 7);
    store_row<D>((bf16_t*)p.dK + (sq.kbase + min(j, Lk - 1)) * p.lddk + h * D, j < Lk, wide,
                 [&](int t) { return pack4_bf16(dk[t], 1.0f); }, lane);
    store_row<D>((bf16_t*)p.dV + (sq.kbase + min(j, Lk - 1)) * p.lddv + h * D, j < Lk, wide,
                 [&](int t) { return pack4_bf16(dv[t], 1.0f); }, lane);
  }
}

// Forward with the same staging: K and V tiles of 64 keys straight into swizzled 8 KB tiles, the key mask as a ballot.
// Any number of keys (64 per sweep).  Keys beyond the sequence read its last row again and are masked.
__global__ __launch_bounds__(256, 5) void attn_fwd_kernel(AttnParams p) {
  constexpr int D = 64;
  __shared__ __attribute__((aligned(1024))) char lds[2 * TILE];
  char* Ks = lds;
  char* Vs = lds + TILE;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.z, h = blockIdx.y;
  const int c = lane & 15, g = lane >> 4;
  const int q = blockIdx.x * 64 + wave * 16 + c;
  const Seq sq = seq_of(p, b);
  const int Lq = sq.Lq, Lk = sq.Lk;
  if ((int)blockIdx.x * 64 >= Lq) return;  // block-uniform (shorter sentence of a packed batch)
  const int qrow = min(q, Lq - 1);
  const bf16_t* Q = (const bf16_t*)p.Q + (sq.qbase + qrow) * p.ldq + h * D;
  const bf16_t* K = (const bf16_t*)p.K + sq.kbase * p.ldk + h * D;
  const bf16_t* V = (const bf16_t*)p.V + sq.kbase * p.ldv + h * D;
  const bf16x8_t fq0 = ld_frag_global(Q + 8 * g), fq1 = ld_frag_global(Q + 32 + 8 * g);

  float m = -INFINITY, l = 0.f;
  f32x4_t o[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const bool wave_has_q = (int)blockIdx.x * 64 + wave * 16 < Lq;
  for (int j0 = 0; j0 < Lk; j0 += 64) {
    if (j0) __syncthreads();   // the sweep before has read its tiles
    dma_tile(Ks, K + (int64_t)j0 * p.ldk, p.ldk, Lk - j0, wave, lane);
    dma_tile(Vs, V + (int64_t)j0 * p.ldv, p.ldv, Lk - j0, wave, lane);
    bool kmsk = j0 + lane >= Lk;
    if (p.key_pad) kmsk = kmsk || p.key_pad[(int64_t)b * p.Lk + min(j0 + lane, Lk - 1)];
    const uint64_t kmask = __ballot(kmsk);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!wave_has_q) continue;
    const int nkt = min(4, (Lk - j0 + 15) >> 4);  // key tiles of this pass that hold keys
    f32x4_t sc[4];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      sc[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (kt < nkt) {
        sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_rows(Ks, 16 * kt, 0, lane), fq0, sc[kt], 0, 0, 0);
        sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_rows(Ks, 16 * kt, 1, lane), fq1, sc[kt], 0, 0, 0);
        const uint32_t mk4 = (uint32_t)(kmask >> (16 * kt + 4 * g)) & 15u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int jl = 16 * kt + 4 * g + r;
          const bool msk = ((mk4 >> r) & 1) || (p.causal && j0 + jl > q);
          sc[kt][r] = msk ? -INFINITY : sc[kt][r] * p.scale;
          mx = fmaxf(mx, sc[kt][r]);
        }
      }
    }
    const float mn = fmaxf(m, quad_max(mx));
    const bool dead = mn == -INFINITY;
    const float alpha = dead ? 1.f : __expf(m - mn);
    float rs = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      if (kt < nkt) {
        const uint32_t km = keep4(p, b, h, q, j0 + 16 * kt + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pr = dead ? 0.f : __expf(sc[kt][r] - mn);
          rs += pr;
          if (p.drop_thr) pr = (km >> r) & 1 ? pr * p.keep_scale : 0.f;
          sc[kt][r] = pr;
        }
      }
    }
    l = l * alpha + quad_sum(rs);
    m = mn;
    const bf16x8_t y0 = pack_y(sc[0], sc[1]), y1 = pack_y(sc[2], sc[3]);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      o[t] *= alpha;
      o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols(Vs, 16 * t, 0, lane), y0, o[t], 0, 0, 0);
      if (nkt > 2)
        o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_cols(Vs, 16 * t, 1, lane), y1, o[t], 0, 0, 0);
    }
  }
  if (q < Lq) {
    const float inv = 1.0f / l;  // fully masked row: 0 * inf = NaN, like the reference's softmax
    bf16_t* O = (bf16_t*)p.Oout + (sq.qbase + q) * p.ldo + h * D;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint2 w;
      bf16_t* e = (bf16_t*)&w;
#pragma unroll
      for (int r = 0; r < 4; ++r) e[r] = f32_to_bf16(o[t][r] * inv);
      *(uint2*)(O + 16 * t + 4 * g) = w;
    }
    if (g == 0) p.LSE[((int64_t)b * p.H + h) * p.Lq + q] = m + __logf(l);
  }
}
}  // namespace f64

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

bool sk_attn_mfma_supported(const AttnParams& p, int dtype, bool want_p) {
  if (dtype != SK_BF16 || want_p || p.dense) return false;
  if (p.dk != p.dv || !(p.dk == 32 || p.dk == 64 || p.dk == 128)) return false;
  if (p.Lq <= 0 || p.Lk <= 0) return false;
  if ((p.ldq | p.ldk | p.ldv | p.ldo) & 7) return false;
  if (!aligned16(p.Q) || !aligned16(p.K) || !aligned16(p.V)) return false;
  if (p.dQ && (((p.lddq | p.lddk | p.lddv) & 3) || !aligned16(p.dQ) || !aligned16(p.dK) || !aligned16(p.dV) ||
               !aligned16(p.O) || !aligned16(p.dO)))
    return false;
  if (p.Oout && !aligned16(p.Oout)) return false;
  return true;
}

static bool fwd_lean() {
  static const bool on = !getenv("SK_ATTN_FWD_LEAN") || atoi(getenv("SK_ATTN_FWD_LEAN")) != 0;
  return on;
}

int sk_attn_fwd_mfma(const AttnParams& p, hipStream_t stream) {
  dim3 grid((p.Lq + 63) / 64, p.H, p.B);
  if (p.dk == 32) hipLaunchKernelGGL(attn_fwd_mfma_kernel<32>, grid, dim3(256), 0, stream, p);
  else if (p.dk == 64 && fwd_lean()) hipLaunchKernelGGL(f64::attn_fwd_kernel, grid, dim3(256), 0, stream, p);
  else if (p.dk == 64) hipLaunchKernelGGL(attn_fwd_mfma_kernel<64>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(attn_fwd_mfma_kernel<128>, grid, dim3(256), 0, stream, p);
  SK_CHECK_LAUNCH("skhip_attn_fwd(mfma)");
  return SK_OK;
}

int sk_attn_bwd_mfma(const AttnParams& p, hipStream_t stream) {
  if (p.Lq <= OT && p.Lk <= OT && p.dk <= 64) {  // single tile on both sides: one fused launch (4 tiles of LDS)
    dim3 grid(1, p.H, p.B);
    static const bool lean = !getenv("SK_ATTN_BWD_LEAN") || atoi(getenv("SK_ATTN_BWD_LEAN")) != 0;
    if (p.dk == 32) hipLaunchKernelGGL(attn_bwd_fused_mfma_kernel<32>, grid, dim3(256), 0, stream, p);
    else if (lean) hipLaunchKernelGGL(f64::attn_bwd_fused_kernel, grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(attn_bwd_fused_mfma_kernel<64>, grid, dim3(256), 0, stream, p);
    SK_CHECK_LAUNCH("skhip_attn_bwd(mfma, fused)");
    return SK_OK;
  }
  dim3 gq((p.Lq + 63) / 64, p.H, p.B), gk((p.Lk + 63) / 64, p.H, p.B);
  if (p.dk == 32) {
    hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<32>, gq, dim3(256), 0, stream, p);
    hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<32>, gk, dim3(256), 0, stream, p);
  } else if (p.dk == 64) {
    hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<64>, gq, dim3(256), 0, stream, p);
    hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<64>, gk, dim3(256), 0, stream, p);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<128>, gq, dim3(256), 0, stream, p);
    hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<128>, gk, dim3(256), 0, stream, p);
  }
  SK_CHECK_LAUNCH("skhip_attn_bwd(mfma)");
  return SK_OK;
}
