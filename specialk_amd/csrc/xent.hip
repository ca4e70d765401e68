// Fused log-softmax + label smoothing + pad masking + gradient + argmax accuracy.
// Replaces calculate_loss / performance (lib/TransformerModel.py:340-375), which builds five
// (T,V) temporaries; here the gradient is written once, optionally in place, by one of two kernels:
//  * xent_rowreg_kernel (bf16 rows of 4096..49152 elements): a 256-thread block holds the whole row
//    in registers, so memory sees one read and one write per element, and loads its next row while
//    it computes this one's gradient: 0.73 ms for 25.6k rows of V = 35000 (4.9 TB/s read + write;
//    an in-place elementwise pass over the same bytes takes 0.60 ms);
//  * xent_kernel (everything else): two sweeps over the row (statistics, then gradient).  With
//    eight blocks per CU the rows in flight exceed the L2s at V = 35000, so its second sweep is
//    served from beyond L2.
// The loss / correct / word counts are summed per block and added to the three global doubles once
// per block: one atomic per row on the same addresses serialises in the L2 and used to cost more
// (0.4 ms for 25.6k rows) than all of the kernel's memory traffic.
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
  // thread 0 sums its block's rows and adds them to the three global statistics once at the end:
  // one atomic per row on the same three addresses serialises in the L2 (tens of thousands of
  // them take longer than the kernel's memory traffic)
  double acc_loss = 0.0;
  int acc_ok = 0, acc_n = 0;
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
      acc_loss += (double)loss;
      acc_ok += am == (int)g;
      ++acc_n;
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
  if (threadIdx.x == 0 && acc_n) {
    atomicAdd(stats + 0, acc_loss);
    atomicAdd(stats + 1, (double)acc_ok);
    atomicAdd(stats + 2, (double)acc_n);
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

// acc + both bf16 halves of w, in one v_dot2 (exact products with 1.0, fp32 accumulation)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ float pair_sum(uint32_t w, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), __builtin_bit_cast(bf16x2_t, 0x3f803f80u), acc,
                                         false);
}

// max(a, b, c) as the bare instruction: fmaxf() first quiets each operand with a v_max of its
// own because values assembled from bits might be signalling NaNs, which quadruples the count.
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// Block-wide (256 threads) maxima of a and b and sum of s; ends with a barrier.
__device__ __forceinline__ void block_reduce_max2_sum(float& a, float& b, float& s) {
  __shared__ float s_a[4], s_b[4], s_s[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a = fmaxf(a, __shfl_xor(a, o, 64));
    b = fmaxf(b, __shfl_xor(b, o, 64));
    s += __shfl_xor(s, o, 64);
  }
  if (lane == 0) { s_a[wave] = a; s_b[wave] = b; s_s[wave] = s; }
  __syncthreads();
  a = fmaxf(fmaxf(s_a[0], s_a[1]), fmaxf(s_a[2], s_a[3]));
  b = fmaxf(fmaxf(s_b[0], s_b[1]), fmaxf(s_b[2], s_b[3]));
  s = (s_s[0] + s_s[1]) + (s_s[2] + s_s[3]);
  __syncthreads();
}
__device__ __forceinline__ float block_reduce_sum(float s) {
  __shared__ float s_s[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  s = wave_sum(s);
  if (lane == 0) s_s[wave] = s;
  __syncthreads();
  s = (s_s[0] + s_s[1]) + (s_s[2] + s_s[3]);
  __syncthreads();
  return s;
}

template <int NV>
__global__ __launch_bounds__(256, (NV <= 8 ? 4 : NV <= 20 ? 3 : 2)) void xent_rowreg_kernel(const bf16_t* __restrict__ logits,
                                                          const int64_t* __restrict__ gold, bf16_t* dlogits,
                                                          double* __restrict__ stats, int64_t T_rows, int V, int64_t ldl,
                                                          float logit_scale, float eps, int64_t pad_id, float grad_scale,
                                                          const float* __restrict__ grad_scale_dev) {
  const int nvec = V >> 3;
  // First row with a real gold word at or after `from` in this block's sequence; the pad rows
  // passed on the way get their zero gradient here.  Block-uniform.
  auto next_row = [&](int64_t from) {
    while (from < T_rows && gold[from] == pad_id) {
      if (dlogits)
        for (int i = threadIdx.x; i < nvec; i += 256)
          *(uint4*)(dlogits + from * ldl + 8 * (int64_t)i) = make_uint4(0, 0, 0, 0);
      from += gridDim.x;
    }
    return from;
  };
  int64_t row = next_row(blockIdx.x);
  if (row >= T_rows) return;
  double acc_loss = 0.0;  // thread 0: this block's share of the three statistics, added once at the end
  int acc_ok = 0, acc_n = 0;
  // The block's next row is loaded into each register vector as soon as the gradient pass is done
  // with it, so those loads (and the gradient's stores) are in flight while the vector ALUs work;
  // without this every block alternates between waiting for memory and computing, and the blocks
  // of a CU fall into step instead of covering for each other.  Every load is unconditional (the
  // vector index is clamped, the last iteration re-reads its own row): registers that are only
  // defined under a branch end up in scratch memory.
  u32x4_t r[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = min((int)threadIdx.x + 256 * k, nvec - 1);
    r[k] = *(const u32x4_t*)(logits + row * ldl + 8 * (int64_t)i);
  }
  for (;;) {
    // per-thread vector numbers are recomputed every row: hoisted out of the loop they would hold
    // NV registers for values that cost one add each
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int64_t g = gold[row];
    const bf16_t* x = logits + row * ldl;
    bf16_t* dx = dlogits ? dlogits + row * ldl : nullptr;
    const int64_t nrow = next_row(row + gridDim.x);
    const bf16_t* xn = logits + (nrow < T_rows ? nrow : row) * ldl;
    // ---- statistics from registers.  The vector ALUs bound this kernel, so the per-element work is
    // cut to the minimum: the plain sum takes two elements per v_dot2, the arg max is replaced by
    // the maxima of the columns before and after the gold one (the prediction is right exactly
    // when nothing before the gold column reaches its logit and nothing after exceeds it — the
    // lowest index wins ties, as torch.max does), and the scale and the base-2 conversion of the
    // exponentials are folded into one multiply-add against the row's final maximum.
    const int gv = (int)(g >> 3), gj = (int)(g & 7);
    float m_lt = -INFINITY, m_gt = -INFINITY, sx = 0.f;  // raw (unscaled) values; logit_scale > 0
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = tid + 256 * k;
      if (i < nvec) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sx = pair_sum(r[k][j], sx);
        if (i != gv) {
          float vm = max3_raw(vec_elem(r[k], 0), vec_elem(r[k], 1), vec_elem(r[k], 2));
          vm = max3_raw(vm, vec_elem(r[k], 3), vec_elem(r[k], 4));
          vm = max3_raw(vm, vec_elem(r[k], 5), vec_elem(r[k], 6));
          vm = max3_raw(vm, vec_elem(r[k], 7), vm);
          m_lt = i < gv ? max3_raw(m_lt, vm, vm) : m_lt;
          m_gt = i > gv ? max3_raw(m_gt, vm, vm) : m_gt;
        }
      }
    }
    if (threadIdx.x == 0) {  // the one vector that holds the gold column, element by element
      const u32x4_t gvec = *(const u32x4_t*)(x + 8 * (int64_t)gv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = vec_elem(gvec, j);
        if (j < gj) m_lt = fmaxf(m_lt, v);
        if (j > gj) m_gt = fmaxf(m_gt, v);
      }
    }
    block_reduce_max2_sum(m_lt, m_gt, sx);
    const float xg_raw = bf16_to_f32(x[g]);
    const float mx = fmaxf(fmaxf(m_lt, m_gt), xg_raw) * logit_scale;
    constexpr float LOG2E = 1.4426950408889634f;
    const float c1 = logit_scale * LOG2E;
    float se = 0.f;
    {
      const float c2 = -mx * LOG2E;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int i = tid + 256 * k;
        if (i < nvec) {
#pragma unroll
          for (int j = 0; j < 8; ++j) se += __builtin_amdgcn_exp2f(fmaf(vec_elem(r[k], j), c1, c2));
        }
      }
    }
    se = block_reduce_sum(se);
    sx *= logit_scale;
    const float lse = mx + logf(se);
    const float xg = xg_raw * logit_scale;
    const float lpg = xg - lse;
    if (threadIdx.x == 0) {
      float sum_lp = sx - (float)V * lse;
      float loss = eps > 0.f ? -((1.f - eps) * lpg + eps / (float)(V - 1) * (sum_lp - lpg)) : -lpg;
      acc_loss += (double)loss;
      acc_ok += m_lt < xg_raw && m_gt <= xg_raw;
      ++acc_n;
    }
    if (dx) {
      // in-place use: every thread has read x[g] before any thread overwrites the row — the two
      // block reductions above are barriers every thread passes after that read
      const float q_other = eps > 0.f ? eps / (float)(V - 1) : 0.f;
      const float q_gold = 1.f - eps;
      const float gs = grad_scale * logit_scale * (grad_scale_dev ? grad_scale_dev[0] : 1.f);
      const float c3 = -lse * LOG2E, c4 = -gs * q_other;  // gs * (p - q_other) = fma(p, gs, c4)
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int i = tid + 256 * k;
        if (i < nvec) {
          uint32_t w[4];
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const float p0 = __builtin_amdgcn_exp2f(fmaf(vec_elem(r[k], j), c1, c3));
            const float p1 = __builtin_amdgcn_exp2f(fmaf(vec_elem(r[k], j + 1), c1, c3));
            const uint32_t lo = f32_to_bf16(fmaf(p0, gs, c4));
            const uint32_t hi = f32_to_bf16(fmaf(p1, gs, c4));
            w[j >> 1] = lo | (hi << 16);
          }
          *(u32x4_t*)(dx + 8 * (int64_t)i) = u32x4_t{w[0], w[1], w[2], w[3]};
          // the gold column takes q_gold instead: its owner rewrites that one element after the
          // vector (same thread, same address: program order holds)
          if (i == gv) dx[g] = f32_to_bf16(gs * (__builtin_amdgcn_exp2f(fmaf(xg_raw, c1, c3)) - q_gold));
        }
        r[k] = *(const u32x4_t*)(xn + 8 * (int64_t)min(i, nvec - 1));
      }
    } else {
#pragma unroll
      for (int k = 0; k < NV; ++k)
        r[k] = *(const u32x4_t*)(xn + 8 * (int64_t)min(tid + 256 * k, nvec - 1));
    }
    if (nrow >= T_rows) break;
    row = nrow;
  }
  if (threadIdx.x == 0) {
    atomicAdd(stats + 0, acc_loss);
    atomicAdd(stats + 1, (double)acc_ok);
    atomicAdd(stats + 2, (double)acc_n);
  }
}

// Blocks of `kernel` (256 threads, no dynamic LDS) resident on one CU at once.
int64_t xent_blocks_per_cu(const void* kernel) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  return per_cu;
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
  if (dtype == SK_BF16 && vec_ok && V >= 4096 && V <= 24 * 2048 && logit_scale > 0.f && !(env_rr && env_rr[0] == '0')) {
#define XENT_RR(NV)                                                                                              \
  do {                                                                                                           \
    static int64_t per_cu = 0;                                                                                   \
    if (!per_cu) per_cu = xent_blocks_per_cu((const void*)xent_rowreg_kernel<NV>);                               \
    const int64_t cap = per_cu * sk_sched_cus(); /* one resident round on the CUs no other work holds */         \
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
