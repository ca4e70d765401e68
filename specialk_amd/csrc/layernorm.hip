// Fused  y = LayerNorm(res + dropout(x)) * gamma + beta,  pad rows zeroed  — forward and
// backward.  Replaces `self.layer_norm(self.dropout(out) + residual)`
// (lib/transformer/SubLayers.py:62-63,83-84) and the `*= non_pad_mask` after every sublayer
// (lib/transformer/Layers.py:22,25,50,55,58).
//
// HBM-bound: one wave per row, 16-byte loads per lane, the row lives in registers between
// the statistics and the normalisation, so each element is read once and written once.
// Statistics are fp32 and two-pass (mean, then centred variance) like torch's.
#include "common.h"
#include <algorithm>
#include <map>
#include <mutex>

namespace {

constexpr int LN_MAX_D = 2048;    // a row is held in registers: up to 32 values per lane
constexpr int LN_MAX_BLOCKS = 1024;
constexpr int LN_NPART = 3;  // dgamma, dbeta, colsum(dx)

// MAXV = 16-byte vectors per lane (compile time, so that a d = 512 bf16 row costs 8 registers,
// not 32).  Every wave walks its rows two at a time and issues the loads of both rows before it
// reduces either: with one row in flight per wave the kernels were latency bound at a third of
// the HBM rate.
// Keep bits of VN consecutive elements starting at the even element index `base` (bit j = keep
// element base + j): one pair hash per two elements; all ones without dropout (keep_scale is 1).
template <int VN>
__device__ __forceinline__ uint32_t keep_mask(SkSeed sd, uint64_t base, uint32_t drop_thr) {
  if (!drop_thr) return 0xFFFFFFFFu;
  uint32_t m = 0;
#pragma unroll
  for (int jp = 0; jp < VN / 2; ++jp) {
    const uint32_t r = sk_hash_pair(sd, (base >> 1) + jp);
    m |= (uint32_t)sk_keep_half(r, 0, drop_thr) << (2 * jp);
    m |= (uint32_t)sk_keep_half(r, 1, drop_thr) << (2 * jp + 1);
  }
  return m;
}

template <typename T, int MAXV>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const uint8_t* __restrict__ row_pad, T* __restrict__ y,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                         int64_t T_rows, int d, float eps, uint32_t drop_thr,
                                                         float keep_scale, uint64_t seed) {
  constexpr int VN = Vec16<T>::N;
  typedef decltype(Vec16<T>().v) raw_t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = d / VN;  // vectors per row
  const SkSeed sd = sk_seed(seed);
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t row0 = (int64_t)blockIdx.x * 4 + wave; row0 < T_rows; row0 += 2 * stride) {
    Vec16<T> a[2][MAXV], b[2][MAXV];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int64_t row = row0 + r * stride;
      if (row < T_rows)
#pragma unroll
        for (int v = 0; v < MAXV; ++v) {
          const int vi = lane + v * 64;
          if (vi < nvec) {
            a[r][v].v = *(const raw_t*)(x + row * d + vi * VN);
            b[r][v].v = *(const raw_t*)(res + row * d + vi * VN);
          }
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int64_t row = row0 + r * stride;
      if (row >= T_rows) break;
      float z[MAXV][VN];
      float sum = 0.f;
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const int vi = lane + v * 64;
        if (vi < nvec) {
          const uint32_t km = keep_mask<VN>(sd, (uint64_t)(row * d + vi * VN), drop_thr);
#pragma unroll
          for (int j = 0; j < VN; ++j) {
            const float xv = (km >> j) & 1 ? a[r][v].get(j) * keep_scale : 0.f;
            z[v][j] = b[r][v].get(j) + xv;
            sum += z[v][j];
          }
        }
      }
      const float mean = wave_sum(sum) / (float)d;
      float sq = 0.f;
#pragma unroll
      for (int v = 0; v < MAXV; ++v)
        if (lane + v * 64 < nvec)
#pragma unroll
          for (int j = 0; j < VN; ++j) { float c = z[v][j] - mean; sq += c * c; }
      const float var = wave_sum(sq) / (float)d;
      const float rstd = 1.0f / sqrtf(var + eps);
      const bool pad = row_pad && row_pad[row];
      if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
      T* yr = y + row * d;
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const int vi = lane + v * 64;
        if (vi < nvec) {
          Vec16<T> o;
          const float4* g4 = (const float4*)(gamma + vi * VN);
          const float4* b4 = (const float4*)(beta + vi * VN);
#pragma unroll
          for (int j4 = 0; j4 < VN / 4; ++j4) {
            float4 g = g4[j4], bt = b4[j4];
            const float gg[4] = {g.x, g.y, g.z, g.w}, bb[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const int j = j4 * 4 + jj;
              const float val = (z[v][j] - mean) * rstd * gg[jj] + bb[jj];
              o.set(j, pad ? 0.f : val);
            }
          }
          *(raw_t*)(yr + vi * VN) = o.v;
        }
      }
    }
  }
}

// Backward.  Each block walks rows with a grid stride and keeps per-column partial sums of
// dgamma, dbeta and dx (the latter is the bias gradient of the linear layer that produced x)
// in registers; they are combined across the block's 4 waves through LDS and written to
// partial[blockIdx][3][d]; a second kernel reduces over blocks in a fixed order.
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                         const T* __restrict__ res, const float* __restrict__ gamma,
                                                         const uint8_t* __restrict__ row_pad,
                                                         const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                         T* __restrict__ dx, T* __restrict__ dres, float* __restrict__ partial,
                                                         int64_t T_rows, int d, uint32_t drop_thr, float keep_scale,
                                                         uint64_t seed) {
  constexpr int VN = Vec16<T>::N;
  typedef decltype(Vec16<T>().v) raw_t;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = (float*)smem_raw;  // [4 waves][3][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = d / VN;
  const SkSeed sd = sk_seed(seed);
  float pg[MAXV][VN], pb[MAXV][VN], px[MAXV][VN], gm[MAXV][VN];
#pragma unroll
  for (int v = 0; v < MAXV; ++v)
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      pg[v][j] = 0.f; pb[v][j] = 0.f; px[v][j] = 0.f;
      const int c = (lane + v * 64) * VN + j;
      gm[v][j] = c < d ? gamma[c] : 0.f;
    }

  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t row0 = (int64_t)blockIdx.x * 4 + wave; row0 < T_rows; row0 += 2 * stride) {
    Vec16<T> a[2][MAXV], b[2][MAXV], g[2][MAXV];
    float mean_r[2], rstd_r[2];
    bool pad_r[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int64_t row = row0 + r * stride;
      if (row < T_rows) {
        mean_r[r] = mean_in[row];
        rstd_r[r] = rstd_in[row];
        pad_r[r] = row_pad && row_pad[row];
#pragma unroll
        for (int v = 0; v < MAXV; ++v) {
          const int vi = lane + v * 64;
          if (vi < nvec) {
            a[r][v].v = *(const raw_t*)(x + row * d + vi * VN);
            b[r][v].v = *(const raw_t*)(res + row * d + vi * VN);
            g[r][v].v = *(const raw_t*)(dy + row * d + vi * VN);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int64_t row = row0 + r * stride;
      if (row >= T_rows) break;
      const bool pad = pad_r[r];
      const float mean = mean_r[r], rstd = rstd_r[r];
      float xhat[MAXV][VN], dxh[MAXV][VN];
      uint32_t km[MAXV];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const int vi = lane + v * 64;
        if (vi < nvec) {
          km[v] = keep_mask<VN>(sd, (uint64_t)(row * d + vi * VN), drop_thr);
#pragma unroll
          for (int j = 0; j < VN; ++j) {
            const float xv = (km[v] >> j) & 1 ? a[r][v].get(j) * keep_scale : 0.f;
            const float xh = (b[r][v].get(j) + xv - mean) * rstd;
            const float dyv = pad ? 0.f : g[r][v].get(j);
            const float dh = dyv * gm[v][j];
            xhat[v][j] = xh;
            dxh[v][j] = dh;
            pg[v][j] += dyv * xh;
            pb[v][j] += dyv;
            s1 += dh;
            s2 += dh * xh;
          }
        }
      }
      const float c1 = wave_sum(s1) / (float)d, c2 = wave_sum(s2) / (float)d;
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const int vi = lane + v * 64;
        if (vi < nvec) {
          Vec16<T> o1, o2;
#pragma unroll
          for (int j = 0; j < VN; ++j) {
            const float dz = rstd * (dxh[v][j] - c1 - xhat[v][j] * c2);
            o1.set(j, dz);
            const float dxv = (km[v] >> j) & 1 ? dz * keep_scale : 0.f;
            o2.set(j, dxv);
            px[v][j] += dxv;
          }
          *(raw_t*)(dres + row * d + vi * VN) = o1.v;
          if (dx != dres) *(raw_t*)(dx + row * d + vi * VN) = o2.v;
        }
      }
    }
  }
  // block reduction of the parameter-gradient partials
#pragma unroll
  for (int v = 0; v < MAXV; ++v) {
    const int vi = lane + v * 64;
    if (vi < nvec)
#pragma unroll
      for (int j = 0; j < VN; ++j) {
        red[(wave * LN_NPART + 0) * d + vi * VN + j] = pg[v][j];
        red[(wave * LN_NPART + 1) * d + vi * VN + j] = pb[v][j];
        red[(wave * LN_NPART + 2) * d + vi * VN + j] = px[v][j];
      }
  }
  __syncthreads();
  const int nd = LN_NPART * d;
  for (int i = threadIdx.x; i < nd; i += 256) {
    const float s = (red[i] + red[nd + i]) + (red[2 * nd + i] + red[3 * nd + i]);
    partial[(int64_t)blockIdx.x * nd + i] = s;
  }
}

// 64 columns x 16 row-lanes per block; row-lane q adds partials q, q+16, ... (four independent
// chains of loads in flight), then the 16 lane sums are combined in a fixed tree.
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float* __restrict__ dxsum, int d,
                                                               int nblocks, int accumulate) {
  __shared__ float part[16][64];
  const int nd = LN_NPART * d;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < nd) {
    int b = rl;
    for (; b + 48 < nblocks; b += 64) {
      s0 += partial[(int64_t)b * nd + i];
      s1 += partial[(int64_t)(b + 16) * nd + i];
      s2 += partial[(int64_t)(b + 32) * nd + i];
      s3 += partial[(int64_t)(b + 48) * nd + i];
    }
    for (; b < nblocks; b += 16) s0 += partial[(int64_t)b * nd + i];
  }
  part[rl][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rl == 0 && i < nd) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = part[2 * k][cl] + part[2 * k + 1][cl];
    const float s = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    float* out = i < d ? dgamma + i : (i < 2 * d ? dbeta + (i - d) : (dxsum ? dxsum + (i - 2 * d) : nullptr));
    if (out) *out = (accumulate && i < 2 * d) ? *out + s : s;
  }
}

int check_d(const char* who, int d, int dtype) {
  int vn = dtype == SK_BF16 ? 8 : 4;
  SK_CHECK_ARG(dtype == SK_F32 || dtype == SK_BF16, "%s: bad dtype %d", who, dtype);
  SK_CHECK_ARG(d > 0 && d % vn == 0 && d <= LN_MAX_D, "%s: d=%d must be a multiple of %d and <= %d", who, d, vn,
               LN_MAX_D);
  return SK_OK;
}

// Blocks of 256 threads of `kernel` that are resident at once on the CUs no other work holds (blocks per CU
// cached per kernel): a grid of exactly one resident round whose blocks share the rows evenly takes twice as long
// if even a few of its blocks have to wait for a CU.
int64_t resident_blocks(const void* kernel, size_t smem) {
  static std::mutex mu;
  static std::map<const void*, int64_t> cache;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(kernel);
  if (it != cache.end()) return it->second * sk_sched_cus();
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, smem) != hipSuccess || per_cu < 1) per_cu = 1;
  cache[kernel] = per_cu;
  return (int64_t)per_cu * sk_sched_cus();
}

}  // namespace

extern "C" int skhip_add_ln_fwd(const void* x, const void* res, const float* gamma, const float* beta,
                                const uint8_t* row_pad, void* y, float* mean, float* rstd, int64_t T, int d, float eps,
                                float p_drop, uint64_t seed, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_d("skhip_add_ln_fwd", d, dtype);
  if (rc) return rc;
  SK_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "skhip_add_ln_fwd: p_drop=%f", p_drop);
  if (T == 0) return SK_OK;
  SK_CHECK_ARG(x && res && gamma && beta && y && mean && rstd, "skhip_add_ln_fwd: null pointer");
  int64_t blocks = (T + 7) / 8;  // two rows per wave per pass
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  uint32_t thr = sk_drop_threshold(p_drop);
  float ks = 1.0f / (1.0f - p_drop);
  // the grid is one resident round of the chip (the waves stride over the rows): a few blocks more
  // than fit at once would run as a second, mostly empty round
#define LN_FWD(TT, MV)                                                                                          \
  blocks = std::min<int64_t>(blocks, resident_blocks((const void*)add_ln_fwd_kernel<TT, MV>, 0));              \
  hipLaunchKernelGGL((add_ln_fwd_kernel<TT, MV>), dim3((unsigned)blocks), dim3(256), 0, stream, (const TT*)x,   \
                     (const TT*)res, gamma, beta, row_pad, (TT*)y, mean, rstd, T, d, eps, thr, ks, seed)
  const int nvl = (d / (dtype == SK_BF16 ? 8 : 4) + 63) / 64;  // 16-byte vectors per lane
  if (dtype == SK_BF16) {
    if (nvl <= 1) { LN_FWD(bf16_t, 1); } else if (nvl <= 2) { LN_FWD(bf16_t, 2); } else { LN_FWD(bf16_t, 4); }
  } else {
    if (nvl <= 2) { LN_FWD(float, 2); } else if (nvl <= 4) { LN_FWD(float, 4); } else { LN_FWD(float, 8); }
  }
#undef LN_FWD
  SK_CHECK_LAUNCH("skhip_add_ln_fwd");
  return SK_OK;
}

extern "C" int64_t skhip_add_ln_bwd_ws_floats(int d) { return (int64_t)LN_MAX_BLOCKS * LN_NPART * d; }

extern "C" int skhip_add_ln_bwd(const void* dy, const void* x, const void* res, const float* gamma,
                                const uint8_t* row_pad, const float* mean, const float* rstd, void* dx, void* dres,
                                float* dgamma, float* dbeta, float* dx_colsum, float* partial_ws, int64_t T, int d,
                                float p_drop, uint64_t seed, int accumulate, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_d("skhip_add_ln_bwd", d, dtype);
  if (rc) return rc;
  SK_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "skhip_add_ln_bwd: p_drop=%f", p_drop);
  SK_CHECK_ARG(dy && x && res && gamma && mean && rstd && dx && dres && dgamma && dbeta && partial_ws,
               "skhip_add_ln_bwd: null pointer");
  SK_CHECK_ARG(p_drop == 0.f || dx != dres, "skhip_add_ln_bwd: dx may alias dres only without dropout");
  int64_t blocks = (T + 7) / 8;  // two rows per wave per pass
  if (blocks > LN_MAX_BLOCKS) blocks = LN_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  uint32_t thr = sk_drop_threshold(p_drop);
  float ks = 1.0f / (1.0f - p_drop);
  size_t smem = (size_t)4 * LN_NPART * d * sizeof(float);
#define LN_BWD(TT, MV)                                                                                            \
  blocks = std::min<int64_t>(blocks, resident_blocks((const void*)add_ln_bwd_kernel<TT, MV>, smem));             \
  hipLaunchKernelGGL((add_ln_bwd_kernel<TT, MV>), dim3((unsigned)blocks), dim3(256), smem, stream, (const TT*)dy, \
                     (const TT*)x, (const TT*)res, gamma, row_pad, mean, rstd, (TT*)dx, (TT*)dres, partial_ws, T, \
                     d, thr, ks, seed)
  const int nvl = (d / (dtype == SK_BF16 ? 8 : 4) + 63) / 64;  // 16-byte vectors per lane
  if (dtype == SK_BF16) {
    if (nvl <= 1) { LN_BWD(bf16_t, 1); } else if (nvl <= 2) { LN_BWD(bf16_t, 2); } else { LN_BWD(bf16_t, 4); }
  } else {
    if (nvl <= 2) { LN_BWD(float, 2); } else if (nvl <= 4) { LN_BWD(float, 4); } else { LN_BWD(float, 8); }
  }
#undef LN_BWD
  SK_CHECK_LAUNCH("skhip_add_ln_bwd");
  hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((LN_NPART * d + 63) / 64), dim3(1024), 0, stream, partial_ws, dgamma,
                     dbeta, dx_colsum, d, (int)blocks, accumulate);
  SK_CHECK_LAUNCH("skhip_add_ln_bwd(param reduce)");
  return SK_OK;
}
