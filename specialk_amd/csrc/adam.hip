// Multi-tensor Adam and dtype conversion helpers.
// Replaces optim.Adam.step as configured by setup_optimiser (lib/TransformerModel.py:159-165:
// betas (0.9, 0.98), eps 1e-9, no weight decay) under ScheduledOptim
// (lib/transformer/Optim.py:15-39).  The reference issues several small launches per
// parameter tensor; here one launch walks every tensor: blockIdx.y selects the tensor,
// blockIdx.x strides over its elements.  HBM-bound: 16 B read (p, g, m, v) + 12 B written
// per element (+2 B for the bf16 working copy).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adam_multi_kernel(const uint64_t* __restrict__ pp, const uint64_t* __restrict__ gp,
                                                         const uint64_t* __restrict__ mp, const uint64_t* __restrict__ vp,
                                                         const uint64_t* __restrict__ bp, const int64_t* __restrict__ counts,
                                                         float step_size, float beta1, float beta2, float eps,
                                                         float bc2_sqrt, float grad_scale) {
  const int t = blockIdx.y;
  const int64_t n = counts[t];
  float* p = (float*)pp[t];
  const float* g = (const float*)gp[t];
  float* m = (float*)mp[t];
  float* v = (float*)vp[t];
  bf16_t* b = bp ? (bf16_t*)bp[t] : nullptr;
  const float w1 = 1.0f - beta1, w2 = 1.0f - beta2;
  const int64_t n4 = n >> 2;
  const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  auto upd = [&](float& pv, float gv, float& mv, float& vv) {
    gv *= grad_scale;
    mv = mv + w1 * (gv - mv);            // exp_avg.lerp_(grad, 1 - beta1)
    vv = vv * beta2 + w2 * gv * gv;      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    float denom = sqrtf(vv) / bc2_sqrt + eps;
    pv = pv - step_size * (mv / denom);  // param.addcdiv_(exp_avg, denom, value=-step_size)
  };
  if (vec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      float4 P = ((float4*)p)[i], G = ((const float4*)g)[i], M = ((float4*)m)[i], V = ((float4*)v)[i];
      upd(P.x, G.x, M.x, V.x); upd(P.y, G.y, M.y, V.y); upd(P.z, G.z, M.z, V.z); upd(P.w, G.w, M.w, V.w);
      ((float4*)p)[i] = P; ((float4*)m)[i] = M; ((float4*)v)[i] = V;
      if (b) { b[4 * i] = f32_to_bf16(P.x); b[4 * i + 1] = f32_to_bf16(P.y); b[4 * i + 2] = f32_to_bf16(P.z); b[4 * i + 3] = f32_to_bf16(P.w); }
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
      float P = p[i], M = m[i], V = v[i];
      upd(P, g[i], M, V);
      p[i] = P; m[i] = M; v[i] = V;
      if (b) b[i] = f32_to_bf16(P);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
      float P = p[i], M = m[i], V = v[i];
      upd(P, g[i], M, V);
      p[i] = P; m[i] = M; v[i] = V;
      if (b) b[i] = f32_to_bf16(P);
    }
  }
}

__global__ __launch_bounds__(256) void cast_multi_kernel(const uint64_t* __restrict__ sp, const uint64_t* __restrict__ dp,
                                                         const int64_t* __restrict__ counts) {
  const int t = blockIdx.y;
  const int64_t n = counts[t];
  const float* s = (const float*)sp[t];
  bf16_t* d = (bf16_t*)dp[t];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = f32_to_bf16(s[i]);
}

// Entry t: fp32 matrix src [R][C] (contiguous) -> dst, row stride ld, as bf16 or fp32, either as it
// is (dst[r][c]) or transposed (dst[c][r]).  64 x 64 tiles through LDS: 16-byte loads along the source rows, and for
// bf16 destinations 16-byte stores along the destination rows (eight lanes a 128-byte line) or, fragment-major,
// whole 1 KB fragments per wave and instruction.  fp32 or unaligned destinations are written element by element.
__global__ __launch_bounds__(256) void cast2d_multi_kernel(const uint64_t* __restrict__ sp, const uint64_t* __restrict__ dp,
                                                           const int64_t* __restrict__ meta) {
  __shared__ float tile[64][65];
  const int t = blockIdx.y;
  const int64_t R = meta[4 * t], C = meta[4 * t + 1], ld = meta[4 * t + 2], flags = meta[4 * t + 3];
  const bool transpose = flags & 1, dst_f32 = flags & 2;
  const int TT = (int)((flags >> 2) & 3) * 2;  // 0: row-major destination, else fragment-major with 2 / 4 tiles per group
  const float* s = (const float*)sp[t];
  const int64_t nrow = transpose ? C : R, ncol = transpose ? R : C;   // the destination's shape
  const int64_t tr = (R + 63) / 64, tc = (C + 63) / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool src_vec = (C & 3) == 0 && ((uintptr_t)s & 15) == 0;
  const bool dst_vec = !dst_f32 && (TT || ((ld & 7) == 0 && (dp[t] & 15) == 0));
  for (int64_t id = blockIdx.x; id < tr * tc; id += gridDim.x) {  // block-uniform trip count
    const int64_t r0 = (id / tc) * 64, c0 = (id % tc) * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = (tid >> 4) + 16 * i, cc = (tid & 15) * 4;
      const int64_t r = r0 + rr, c = c0 + cc;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < R) {
        if (src_vec && c + 4 <= C) v = *(const float4*)(s + r * C + c);
        else {
          if (c < C) v.x = s[r * C + c];
          if (c + 1 < C) v.y = s[r * C + c + 1];
          if (c + 2 < C) v.z = s[r * C + c + 2];
          if (c + 3 < C) v.w = s[r * C + c + 3];
        }
      }
      tile[rr][cc] = v.x; tile[rr][cc + 1] = v.y; tile[rr][cc + 2] = v.z; tile[rr][cc + 3] = v.w;
    }
    __syncthreads();
    const int64_t O0 = transpose ? c0 : r0, OC0 = transpose ? r0 : c0;   // the destination tile's origin
    auto at = [&](int o, int oc) { return transpose ? tile[oc][o] : tile[o][oc]; };
    if (dst_vec && TT) {
      // fragment-major [n / (16 T)][k / 32][T][64 lanes][8]: tile ti of a group holds as its row i the group's row
      // 4 T (i >> 2) + 4 ti + (i & 3), lane (i, gk) the elements k = 8 gk .. 8 gk + 7 of it.  The 64 x 64 tile
      // holds 4 / T groups x T tiles x 2 K-steps = 8 fragments; a wave writes two of them.
      const int64_t ksteps = (ld > 0 ? ld : ncol) >> 5;  // K-steps of the whole destination (this entry may be a K slice of it)
      const int i_row = lane & 15, gk = lane >> 4;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int u = wave * 2 + i, ks = u & 1, tg = u >> 1, ti = tg % TT, gi = tg / TT;
        const int o = gi * 16 * TT + 4 * TT * (i_row >> 2) + 4 * ti + (i_row & 3), oc = 32 * ks + 8 * gk;
        if (O0 + o < nrow && OC0 + oc < ncol) {
          Vec16<bf16_t> w;
#pragma unroll
          for (int j = 0; j < 8; ++j) w.set(j, at(o, oc + j));
          const int64_t G = (O0 + o) / (16 * TT);
          const int64_t idx = ((((G * ksteps + ((OC0 + oc) >> 5)) * TT + ti) * 64) + lane) * 8;
          *(uint4*)((bf16_t*)dp[t] + idx) = w.v;
        }
      }
    } else if (dst_vec) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = tid + 256 * i, o = q >> 3, oc = (q & 7) * 8;
        if (O0 + o < nrow && OC0 + oc < ncol) {
          bf16_t* d = (bf16_t*)dp[t] + (O0 + o) * ld + OC0 + oc;
          if (OC0 + oc + 8 <= ncol) {
            Vec16<bf16_t> w;
#pragma unroll
            for (int j = 0; j < 8; ++j) w.set(j, at(o, oc + j));
            *(uint4*)d = w.v;
          } else {
            for (int j = 0; OC0 + oc + j < ncol; ++j) d[j] = f32_to_bf16(at(o, oc + j));
          }
        }
      }
    } else {
      for (int e = tid; e < 64 * 64; e += 256) {
        const int o = e >> 6, oc = e & 63;
        if (O0 + o < nrow && OC0 + oc < ncol) {
          const float v = at(o, oc);
          if (dst_f32) ((float*)dp[t])[(O0 + o) * ld + OC0 + oc] = v;
          else ((bf16_t*)dp[t])[(O0 + o) * ld + OC0 + oc] = f32_to_bf16(v);
        }
      }
    }
    __syncthreads();
  }
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void convert_kernel(const TS* __restrict__ s, TD* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    d[i] = Cvt<TD>::from_f32(Cvt<TS>::to_f32(s[i]));
}

unsigned blocks_for(int64_t max_count, int per_thread) {
  int64_t b = (max_count + 256 * per_thread - 1) / (256 * per_thread);
  if (b < 1) b = 1;
  if (b > 1024) b = 1024;
  return (unsigned)b;
}

}  // namespace

extern "C" int skhip_adam_multi(const uint64_t* param_ptrs, const uint64_t* grad_ptrs, const uint64_t* m_ptrs,
                                const uint64_t* v_ptrs, const uint64_t* bf16_copy_ptrs, const int64_t* counts, int n,
                                int64_t max_count, float lr, float beta1, float beta2, float eps, int step,
                                float grad_scale, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(n >= 0 && n <= 65535, "skhip_adam_multi: n=%d", n);
  SK_CHECK_ARG(step >= 1, "skhip_adam_multi: step=%d must start at 1", step);
  if (n == 0) return SK_OK;
  SK_CHECK_ARG(param_ptrs && grad_ptrs && m_ptrs && v_ptrs && counts, "skhip_adam_multi: null table");
  // Scalars as torch computes them: Python doubles, rounded to fp32 where they meet tensors.
  double bc1 = 1.0 - pow((double)beta1, (double)step);
  double bc2 = 1.0 - pow((double)beta2, (double)step);
  float step_size = (float)((double)lr / bc1);
  float bc2_sqrt = (float)sqrt(bc2);
  dim3 grid(blocks_for(max_count, 4), (unsigned)n);
  hipLaunchKernelGGL(adam_multi_kernel, grid, dim3(256), 0, stream, param_ptrs, grad_ptrs, m_ptrs, v_ptrs,
                     bf16_copy_ptrs, counts, step_size, beta1, beta2, eps, bc2_sqrt, grad_scale);
  SK_CHECK_LAUNCH("skhip_adam_multi");
  return SK_OK;
}

extern "C" int skhip_cast_multi(const uint64_t* src_f32_ptrs, const uint64_t* dst_bf16_ptrs, const int64_t* counts,
                                int n, int64_t max_count, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(n >= 0 && n <= 65535, "skhip_cast_multi: n=%d", n);
  if (n == 0) return SK_OK;
  SK_CHECK_ARG(src_f32_ptrs && dst_bf16_ptrs && counts, "skhip_cast_multi: null table");
  dim3 grid(blocks_for(max_count, 4), (unsigned)n);
  hipLaunchKernelGGL(cast_multi_kernel, grid, dim3(256), 0, stream, src_f32_ptrs, dst_bf16_ptrs, counts);
  SK_CHECK_LAUNCH("skhip_cast_multi");
  return SK_OK;
}

extern "C" int skhip_cast2d_multi(const uint64_t* src_f32_ptrs, const uint64_t* dst_ptrs, const int64_t* meta, int n,
                                  int64_t max_tiles, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(n >= 0 && n <= 65535, "skhip_cast2d_multi: n=%d", n);
  if (n == 0) return SK_OK;
  SK_CHECK_ARG(src_f32_ptrs && dst_ptrs && meta, "skhip_cast2d_multi: null table");
  int64_t bx = max_tiles < 1 ? 1 : (max_tiles > 512 ? 512 : max_tiles);
  hipLaunchKernelGGL(cast2d_multi_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, stream, src_f32_ptrs, dst_ptrs, meta);
  SK_CHECK_LAUNCH("skhip_cast2d_multi");
  return SK_OK;
}

extern "C" int skhip_convert(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG((src_dtype == SK_F32 || src_dtype == SK_BF16) && (dst_dtype == SK_F32 || dst_dtype == SK_BF16),
               "skhip_convert: bad dtype");
  if (n == 0) return SK_OK;
  SK_CHECK_ARG(src && dst, "skhip_convert: null pointer");
  dim3 grid(blocks_for(n, 4));
  if (src_dtype == SK_F32 && dst_dtype == SK_BF16)
    hipLaunchKernelGGL((convert_kernel<float, bf16_t>), grid, dim3(256), 0, stream, (const float*)src, (bf16_t*)dst, n);
  else if (src_dtype == SK_BF16 && dst_dtype == SK_F32)
    hipLaunchKernelGGL((convert_kernel<bf16_t, float>), grid, dim3(256), 0, stream, (const bf16_t*)src, (float*)dst, n);
  else if (src_dtype == SK_F32)
    hipLaunchKernelGGL((convert_kernel<float, float>), grid, dim3(256), 0, stream, (const float*)src, (float*)dst, n);
  else
    hipLaunchKernelGGL((convert_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)src, (bf16_t*)dst, n);
  SK_CHECK_LAUNCH("skhip_convert");
  return SK_OK;
}
