// Fused log-softmax + label smoothing + pad masking + gradient + argmax accuracy.
// Replaces calculate_loss / performance (lib/TransformerModel.py:340-375), which builds five
// (T,V) temporaries; here the gradient is written once, optionally in place, by one of two kernels:
//  * xent_rowreg_kernel (bf16 rows of 4096..49152 elements): a 256-thread block holds the whole row
//    in registers, so memory sees one read and one write per element;
//  * xent_kernel (everything else): two sweeps over the row (statistics, then gradient).  With
//    eight blocks per CU the rows in flight exceed the L2s at V = 35000, so its second sweep is
//    served from beyond L2.
// Both run at about 3 TB/s of algorithmic traffic at V = 35000 (1.2 ms for 25.6k rows): two
// exponentials and ~25 vector operations per element keep the vector ALUs about half busy.
//
// Per row r with gold g (scaled logits x = logit_scale * logits):
//   lse = log sum_v exp(x_v);  lp_v = x_v - lse
//   smoothing eps:  loss_r = -[(1-eps) lp_g + eps/(V-1) (sum_v lp_v - lp_g)]
//   eps = 0:        loss_r = -lp_g                      (F.cross_entropy, ignore_index = PAD)
//   d loss_r / d x_v = softmax_v - q_v,   q_g = 1-eps, q_{v != g} = eps/(V-1)
// Rows with g == pad_id contribute nothing and get a zero gradient.  The loss is a SUM.
#include "common.h"

namespace {

// fp32 storage is the parity path: use the accurate expf there, the hardware exp2-based
// __expf for bf16 storage (its error is far below bf16 resolution).
template <typename T> __device__ __forceinline__ float sk_exp(float v);
template <> __device__ __forceinline__ float sk_exp<float>(float v) { return expf(v); }
template <> __device__ __forceinline__ float sk_exp<bf16_t>(float v) { return __expf(v); }

__device__ __forceinline__ void block_reduce4(float& mx, float& se, float& sx, int& am, float& amv) {
  // (max, sum-exp relative to max) pairs combine; sx = plain sum; (amv, am) = arg max, lowest index on ties
  __shared__ float s_mx[4], s_se[4], s_sx[4], s_amv[4];
  __shared__ int s_am[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float omx = __shfl_xor(mx, o, 64), ose = __shfl_xor(se, o, 64), osx = __shfl_xor(sx, o, 64);
    float oamv = __shfl_xor(amv, o, 64);
    int oam = __shfl_xor(am, o, 64);
    float nm = fmaxf(mx, omx);
    se = (mx == -INFINITY ? 0.f : se * expf(mx - nm)) + (omx == -INFINITY ? 0.f : ose * expf(omx - nm));
    mx = nm;
    sx += osx;
    if (oamv > amv || (oamv == amv && oam < am)) { amv = oamv; am = oam; }
  }
  if (lane == 0) { s_mx[wave] = mx; s_se[wave] = se; s_sx[wave] = sx; s_amv[wave] = amv; s_am[wave] = am; }
  __syncthreads();
  mx = s_mx[0]; se = s_se[0]; sx = s_sx[0]; amv = s_amv[0]; am = s_am[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    float nm = fmaxf(mx, s_mx[w]);
    se = (mx == -INFINITY ? 0.f : se * expf(mx - nm)) + (s_mx[w] == -INFINITY ? 0.f : s_se[w] * expf(s_mx[w] - nm));
    mx = nm;
    sx += s_sx[w];
    if (s_amv[w] > amv || (s_amv[w] == amv && s_am[w] < am)) { amv = s_amv[w]; am = s_am[w]; }
  }
  __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void xent_kernel(const T* __restrict__ logits, const int64_t* __restrict__ gold,
                                                   T* dlogits, double* __restrict__ stats, int64_t T_rows, int64_t V,
                                                   int64_t ldl, float logit_scale, float eps, int64_t pad_id,
                                                   float grad_scale, const float* __restrict__ grad_scale_dev, int vec_ok) {
  constexpr int VN = Vec16<T>::N;
  for (int64_t row = blockIdx.x; row < T_rows; row += gridDim.x) {
    const int64_t g = gold[row];
    const T* x = logits + row * ldl;
    T* dx = dlogits ? dlogits + row * ldl : nullptr;
    if (g == pad_id) {  // block-uniform
      if (dx) {
        if (vec_ok) {
          Vec16<T> z;
#pragma unroll
          for (int j = 0; j < VN; ++j) z.set(j, 0.f);
          for (int64_t c = (int64_t)threadIdx.x * VN; c < V; c += 256 * VN) *(decltype(z.v)*)(dx + c) = z.v;
        } else {
          for (int64_t c = threadIdx.x; c < V; c += 256) dx[c] = Cvt<T>::from_f32(0.f);
        }
      }
      continue;
    }
    // ---- sweep 1: online max / sum-exp, plain sum, arg max
    float mx = -INFINITY, se = 0.f, sx = 0.f, amv = -INFINITY;
    int am = 0x7fffffff;
    auto visit = [&](float v, int idx) {
      v *= logit_scale;
      sx += v;
      if (v > amv) { amv = v; am = idx; }
      if (v > mx) { se = se * sk_exp<T>(mx - v) + 1.f; mx = v; }
      else se += sk_exp<T>(v - mx);
    };
    if (vec_ok) {
      // one rescale of the running sum per 16-byte vector instead of a branch per element
      for (int64_t c = (int64_t)threadIdx.x * VN; c < V; c += 256 * VN) {
        Vec16<T> a;
        a.v = *(const decltype(a.v)*)(x + c);
        float v[VN];
        float vm = -INFINITY;
#pragma unroll
        for (int j = 0; j < VN; ++j) {
          v[j] = a.get(j) * logit_scale;
          sx += v[j];
          if (v[j] > amv) { amv = v[j]; am = (int)(c + j); }
          vm = fmaxf(vm, v[j]);
        }
        const float nm = fmaxf(mx, vm);
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < VN; ++j) part += sk_exp<T>(v[j] - nm);
        se = se * sk_exp<T>(mx - nm) + part;  // mx = -inf on the first vector: exp(-inf) = 0 and se = 0
        mx = nm;
      }
    } else {
      for (int64_t c = threadIdx.x; c < V; c += 256) visit(Cvt<T>::to_f32(x[c]), (int)c);
    }
    block_reduce4(mx, se, sx, am, amv);
    const float lse = mx + logf(se);
    const float xg = Cvt<T>::to_f32(x[g]) * logit_scale;
    const float lpg = xg - lse;
    if (threadIdx.x == 0) {
      float sum_lp = sx - (float)V * lse;
      float loss = eps > 0.f ? -((1.f - eps) * lpg + eps / (float)(V - 1) * (sum_lp - lpg)) : -lpg;
      atomicAdd(stats + 0, (double)loss);
      atomicAdd(stats + 1, am == (int)g ? 1.0 : 0.0);
      atomicAdd(stats + 2, 1.0);
    }
    // ---- sweep 2: gradient
    if (dx) {
      __syncthreads();  // in-place use: every thread has read x[g] before any thread overwrites the row
      const float q_other = eps > 0.f ? eps / (float)(V - 1) : 0.f;
      const float q_gold = 1.f - eps;
      const float gs = grad_scale * logit_scale * (grad_scale_dev ? grad_scale_dev[0] : 1.f);
      if (vec_ok) {
        for (int64_t c = (int64_t)threadIdx.x * VN; c < V; c += 256 * VN) {
          Vec16<T> a, o;
          a.v = *(const decltype(a.v)*)(x + c);
#pragma unroll
          for (int j = 0; j < VN; ++j) {
            float p = sk_exp<T>(a.get(j) * logit_scale - lse);
            o.set(j, gs * (p - ((c + j) == g ? q_gold : q_other)));
          }
          *(decltype(o.v)*)(dx + c) = o.v;
        }
      } else {
        for (int64_t c = threadIdx.x; c < V; c += 256) {
          float p = sk_exp<T>(Cvt<T>::to_f32(x[c]) * logit_scale - lse);
          dx[c] = Cvt<T>::from_f32(gs * (p - (c == g ? q_gold : q_other)));
        }
      }
    }
  }
}

// Register-resident variant for bf16 rows of up to NV * 2048 elements: each thread loads its NV
// 16-byte vectors of the row once (all loads in flight together), the statistics and the
// gradient are computed from registers, so the row is read from memory exactly once and written
// once — the two-sweep kernel above reads it a second time, and with V = 35000 the rows in flight
// do not fit the L2s, so that second read also comes from beyond L2.
// element j (0..7) of a 16-byte vector of bf16 as fp32, by register arithmetic only (taking the
// address of the vector would move the whole row array to scratch memory)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float vec_elem(const u32x4_t v, int j) {
  const uint32_t w = v[j >> 1];
  return __uint_as_float((j & 1) ? (w & 0xffff0000u) : (w << 16));
}

template <int NV>
__global__ __launch_bounds__(256, (NV <= 8 ? 4 : 3)) void xent_rowreg_kernel(const bf16_t* __restrict__ logits,
                                                          const int64_t* __restrict__ gold, bf16_t* dlogits,
                                                          double* __restrict__ stats, int64_t T_rows, int V, int64_t ldl,
                                                          float logit_scale, float eps, int64_t pad_id, float grad_scale,
                                                          const float* __restrict__ grad_scale_dev) {
  const int nvec = V >> 3;
  for (int64_t row = blockIdx.x; row < T_rows; row += gridDim.x) {
    const int64_t g = gold[row];
    const bf16_t* x = logits + row * ldl;
    bf16_t* dx = dlogits ? dlogits + row * ldl : nullptr;
    if (g == pad_id) {  // block-uniform
      if (dx)
        for (int i = threadIdx.x; i < nvec; i += 256) *(uint4*)(dx + 8 * (int64_t)i) = make_uint4(0, 0, 0, 0);
      continue;
    }
    u32x4_t r[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = threadIdx.x + 256 * k;
      r[k] = i < nvec ? *(const u32x4_t*)(x + 8 * (int64_t)i) : u32x4_t{0, 0, 0, 0};
    }
    // ---- statistics from registers: local max / plain sum / arg max, then sum-exp against the local max
    float mx = -INFINITY, sx = 0.f, amv = -INFINITY;
    int am = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = threadIdx.x + 256 * k;
      if (i < nvec) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = vec_elem(r[k], j) * logit_scale;
          sx += v;
          if (v > amv) { amv = v; am = 8 * k + j; }  // position inside this thread's registers (a constant)
        }
      }
    }
    if (am != 0x7fffffff) am = 8 * ((int)threadIdx.x + 256 * (am >> 3)) + (am & 7);  // -> column
    mx = amv;
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = threadIdx.x + 256 * k;
      if (i < nvec) {
#pragma unroll
        for (int j = 0; j < 8; ++j) se += __expf(vec_elem(r[k], j) * logit_scale - mx);
      }
    }
    block_reduce4(mx, se, sx, am, amv);
    const float lse = mx + logf(se);
    const float xg = bf16_to_f32(x[g]) * logit_scale;
    const float lpg = xg - lse;
    if (threadIdx.x == 0) {
      float sum_lp = sx - (float)V * lse;
      float loss = eps > 0.f ? -((1.f - eps) * lpg + eps / (float)(V - 1) * (sum_lp - lpg)) : -lpg;
      atomicAdd(stats + 0, (double)loss);
      atomicAdd(stats + 1, am == (int)g ? 1.0 : 0.0);
      atomicAdd(stats + 2, 1.0);
    }
    if (dx) {
      __syncthreads();  // in-place use: every thread has read x[g] before any thread overwrites the row
      const float q_other = eps > 0.f ? eps / (float)(V - 1) : 0.f;
      const float q_gold = 1.f - eps;
      const float gs = grad_scale * logit_scale * (grad_scale_dev ? grad_scale_dev[0] : 1.f);
      // register position (8 * k + j) of the gold column if this thread holds it, else -1: comparing
      // against constants keeps 8 * NV column numbers out of the registers
      const int gv = (int)(g >> 3);
      const int gpos = (gv & 255) == (int)threadIdx.x ? 8 * (gv >> 8) + (int)(g & 7) : -1;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < nvec) {
          uint32_t w[4];
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const float p0 = __expf(vec_elem(r[k], j) * logit_scale - lse);
            const float p1 = __expf(vec_elem(r[k], j + 1) * logit_scale - lse);
            const uint32_t lo = f32_to_bf16(gs * (p0 - ((8 * k + j) == gpos ? q_gold : q_other)));
            const uint32_t hi = f32_to_bf16(gs * (p1 - ((8 * k + j + 1) == gpos ? q_gold : q_other)));
            w[j >> 1] = lo | (hi << 16);
          }
          *(u32x4_t*)(dx + 8 * (int64_t)i) = u32x4_t{w[0], w[1], w[2], w[3]};
        }
      }
    }
  }
}

// Blocks of `kernel` (256 threads, no dynamic LDS) resident on the chip at once.
int64_t xent_resident_blocks(const void* kernel) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  return (int64_t)per_cu * sk_num_cus();
}

}  // namespace

extern "C" int skhip_xent_fwd_bwd(const void* logits, const int64_t* gold, void* dlogits, double* out_stats, int64_t T,
                                  int64_t V, int64_t ldl, float logit_scale, float smoothing_eps, int64_t pad_id,
                                  float grad_scale, const float* grad_scale_dev, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(dtype == SK_F32 || dtype == SK_BF16, "skhip_xent_fwd_bwd: bad dtype");
  SK_CHECK_ARG(V > 1 && V < (1ll << 31) && ldl >= V, "skhip_xent_fwd_bwd: bad V=%lld ldl=%lld", (long long)V, (long long)ldl);
  SK_CHECK_ARG(smoothing_eps >= 0.f && smoothing_eps < 1.f, "skhip_xent_fwd_bwd: eps=%f", smoothing_eps);
  if (T == 0) return SK_OK;
  SK_CHECK_ARG(logits && gold && out_stats, "skhip_xent_fwd_bwd: null pointer");
  const int vn = dtype == SK_BF16 ? 8 : 4;
  int vec_ok = (V % vn == 0) && (ldl % vn == 0) && ((uintptr_t)logits % 16 == 0) && (!dlogits || (uintptr_t)dlogits % 16 == 0);
  // bf16 rows that fit a block's registers: read once (SKHIP_XENT_ROWREG=0 keeps the two-sweep kernel)
  const char* env_rr = getenv("SKHIP_XENT_ROWREG");
  if (dtype == SK_BF16 && vec_ok && V >= 4096 && V <= 24 * 2048 && !(env_rr && env_rr[0] == '0')) {
#define XENT_RR(NV)                                                                                              \
  do {                                                                                                           \
    static int64_t cap = 0;                                                                                      \
    if (!cap) cap = xent_resident_blocks((const void*)xent_rowreg_kernel<NV>);                                   \
    const int64_t nb = T < cap ? T : cap;                                                                        \
    hipLaunchKernelGGL((xent_rowreg_kernel<NV>), dim3((unsigned)nb), dim3(256), 0, stream, (const bf16_t*)logits, \
                       gold, (bf16_t*)dlogits, out_stats, T, (int)V, ldl, logit_scale, smoothing_eps, pad_id,    \
                       grad_scale, grad_scale_dev);                                                              \
  } while (0)
    if (V <= 8 * 2048) XENT_RR(8);
    else if (V <= 20 * 2048) XENT_RR(20);
    else XENT_RR(24);
#undef XENT_RR
    SK_CHECK_LAUNCH("skhip_xent_fwd_bwd");
    return SK_OK;
  }
  int64_t blocks = T < 8192 ? T : 8192;
  if (dtype == SK_BF16)
    hipLaunchKernelGGL((xent_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16_t*)logits, gold,
                       (bf16_t*)dlogits, out_stats, T, V, ldl, logit_scale, smoothing_eps, pad_id, grad_scale, grad_scale_dev,
                       vec_ok);
  else
    hipLaunchKernelGGL((xent_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)logits, gold,
                       (float*)dlogits, out_stats, T, V, ldl, logit_scale, smoothing_eps, pad_id, grad_scale, grad_scale_dev,
                       vec_ok);
  SK_CHECK_LAUNCH("skhip_xent_fwd_bwd");
  return SK_OK;
}
