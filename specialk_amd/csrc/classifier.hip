// Kernels of the style-transfer classifier path (base/train_decoder.py:385-424 feeding
// base/classifier/onmt/CNNModels.py:48-65): row softmax over the classifier vocabulary, the
// time-padded / trimmed convolution windows of the embedded probabilities, and the ReLU-max-pool +
// linear + sigmoid + binary-cross-entropy head.  The three matrix products between them (class-input
// projection, probabilities x embedding table, convolution as a product over windows) are skhip_gemm calls.
#include "common.h"

namespace {

// y = softmax(x) per row.  HBM-bound: one read of x from memory (the second sweep of a 70 KB row
// comes from L2) and one write of y; in place is allowed (each thread rewrites only what it read).
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* x, T* y, int64_t rows, int64_t V,
                                                           int64_t ldx, int64_t ldy, int vec_ok) {
  constexpr int VN = Vec16<T>::N;
  __shared__ float red[2][4];
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const T* xr = x + row * ldx;
    T* yr = y + row * ldy;
    float mx = -INFINITY, se = 0.f;
    if (vec_ok) {
      for (int64_t c = (int64_t)threadIdx.x * VN; c < V; c += 256 * VN) {
        Vec16<T> a;
        a.v = *(const decltype(a.v)*)(xr + c);
        float vm = -INFINITY;
#pragma unroll
        for (int j = 0; j < VN; ++j) vm = fmaxf(vm, a.get(j));
        const float nm = fmaxf(mx, vm);
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < VN; ++j) part += __expf(a.get(j) - nm);
        se = se * __expf(mx - nm) + part;
        mx = nm;
      }
    } else {
      for (int64_t c = threadIdx.x; c < V; c += 256) {
        const float v = Cvt<T>::to_f32(xr[c]);
        const float nm = fmaxf(mx, v);
        se = se * __expf(mx - nm) + __expf(v - nm);
        mx = nm;
      }
    }
    // block reduction of (max, sum-exp); a thread that saw no element holds (-inf, 0)
    const float wm = wave_max(mx);
    const float ws = wave_sum(mx == -INFINITY ? 0.f : se * __expf(mx - wm));
    const int wave = threadIdx.x >> 6;
    __syncthreads();  // the previous row's readers are done with `red`
    if ((threadIdx.x & 63) == 0) { red[0][wave] = wm; red[1][wave] = ws; }
    __syncthreads();
    float bm = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    float bs = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) bs += red[0][w] == -INFINITY ? 0.f : red[1][w] * __expf(red[0][w] - bm);
    const float inv = 1.f / bs;
    if (vec_ok) {
      for (int64_t c = (int64_t)threadIdx.x * VN; c < V; c += 256 * VN) {
        Vec16<T> a, o;
        a.v = *(const decltype(a.v)*)(xr + c);
#pragma unroll
        for (int j = 0; j < VN; ++j) o.set(j, __expf(a.get(j) - bm) * inv);
        *(decltype(o.v)*)(yr + c) = o.v;
      }
    } else {
      for (int64_t c = threadIdx.x; c < V; c += 256) yr[c] = Cvt<T>::from_f32(__expf(Cvt<T>::to_f32(xr[c]) - bm) * inv);
    }
  }
}

// cols[(b * P + t), j * E + e] = step (t + j) of sentence b, channel e, where a sentence is its own rows,
// then `pad_row` up to min(L, S) steps (the constant row a zeroed decoder output produces), then zeros
// up to the classifier's window S.  Sentence b owns rows off[b] .. off[b+1]-1 of emb (packed batch), or,
// when off is null, step i of sentence b is row b*stride_b + i*stride_t (padded batch, batch- or
// time-major; pad rows already hold their constant row).
template <typename T>
__global__ __launch_bounds__(256) void conv_windows_kernel(const T* __restrict__ emb, const int* __restrict__ off,
                                                           T* __restrict__ cols, int B, int L, int S, int fs, int E,
                                                           int64_t lde, int64_t ldcols, int64_t pad_row,
                                                           int64_t stride_b, int64_t stride_t) {
  const int P = S - fs + 1;
  const int steps = L < S ? L : S;
  const int64_t per_row = (int64_t)fs * E;
  const int64_t total = (int64_t)B * P * per_row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / per_row;
    const int k = (int)(i - r * per_row);
    const int j = k / E, e = k - j * E;
    const int b = (int)(r / P), t = (int)(r - (int64_t)b * P);
    const int pos = t + j;
    T v = Cvt<T>::from_f32(0.f);
    if (pos < steps) {
      int64_t src;
      if (off) {
        const int lo = off[b], len = off[b + 1] - lo;
        src = pos < len ? (int64_t)lo + pos : pad_row;
      } else {
        src = (int64_t)b * stride_b + (int64_t)pos * stride_t;
      }
      v = emb[src * lde + e];
    }
    cols[r * ldcols + k] = v;
  }
}

// One block per sentence: max over the P window positions of h (already ReLU'd by the convolution
// product's epilogue), dot with the linear layer, sigmoid, and this sentence's cross-entropy term.
template <typename T>
__global__ __launch_bounds__(64) void cnn_head_kernel(const T* __restrict__ h, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      float* __restrict__ terms, int P, int F, int64_t ldh,
                                                      int target_label) {
  const int b = blockIdx.x;
  float acc = 0.f;
  for (int f = threadIdx.x; f < F; f += 64) {
    float m = -INFINITY;
    for (int t = 0; t < P; ++t) m = fmaxf(m, Cvt<T>::to_f32(h[((int64_t)b * P + t) * ldh + f]));
    acc += m * w[f];
  }
  acc = wave_sum(acc);
  if (threadIdx.x == 0) {
    const float out = 1.f / (1.f + expf(-(acc + bias[0])));
    y[b] = out;
    // nn.BCELoss clamps each logarithm at -100
    terms[b] = -(target_label ? fmaxf(logf(out), -100.f) : fmaxf(logf(1.f - out), -100.f));
  }
}

// mean of the B terms, in index order (reproducible)
__global__ __launch_bounds__(64) void mean_terms_kernel(const float* __restrict__ terms, float* __restrict__ loss, int B) {
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += 64) s += terms[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) loss[0] = s / (float)B;
}

}  // namespace

extern "C" int skhip_softmax_rows(const void* x, void* y, int64_t rows, int64_t V, int64_t ldx, int64_t ldy, int dtype,
                                  void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(dtype == SK_F32 || dtype == SK_BF16, "skhip_softmax_rows: bad dtype");
  SK_CHECK_ARG(rows >= 0 && V >= 1 && ldx >= V && ldy >= V, "skhip_softmax_rows: bad shape rows=%lld V=%lld ldx=%lld ldy=%lld",
               (long long)rows, (long long)V, (long long)ldx, (long long)ldy);
  if (rows == 0) return SK_OK;
  SK_CHECK_ARG(x && y, "skhip_softmax_rows: null pointer");
  const int vn = dtype == SK_BF16 ? 8 : 4;
  const int vec_ok = (V % vn == 0) && (ldx % vn == 0) && (ldy % vn == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
  const int64_t cap = (int64_t)sk_num_cus() * 8;
  const unsigned blocks = (unsigned)(rows < cap ? rows : cap);
  if (dtype == SK_BF16)
    hipLaunchKernelGGL((softmax_rows_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, rows,
                       V, ldx, ldy, vec_ok);
  else
    hipLaunchKernelGGL((softmax_rows_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)x, (float*)y, rows, V,
                       ldx, ldy, vec_ok);
  SK_CHECK_LAUNCH("skhip_softmax_rows");
  return SK_OK;
}

extern "C" int skhip_conv_windows(const void* emb, const int* off, void* cols, int B, int L, int S, int filter_size, int E,
                                  int64_t lde, int64_t ldcols, int64_t emb_rows, int64_t pad_row, int64_t stride_b,
                                  int64_t stride_t, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(dtype == SK_F32 || dtype == SK_BF16, "skhip_conv_windows: bad dtype");
  SK_CHECK_ARG(B >= 0 && L >= 1 && E >= 1 && filter_size >= 1 && S >= filter_size,
               "skhip_conv_windows: bad shape B=%d L=%d S=%d filter=%d E=%d", B, L, S, filter_size, E);
  SK_CHECK_ARG(lde >= E && ldcols >= (int64_t)filter_size * E, "skhip_conv_windows: leading dimension too small");
  const int64_t last_row = (B > 0 ? (int64_t)(B - 1) * stride_b : 0) + (int64_t)((L < S ? L : S) - 1) * stride_t;
  SK_CHECK_ARG(off ? (pad_row >= 0 && pad_row < emb_rows) : (stride_b >= 0 && stride_t >= 0 && last_row < emb_rows),
               "skhip_conv_windows: emb_rows=%lld does not cover the rows addressed (pad_row=%lld)", (long long)emb_rows,
               (long long)pad_row);
  if (B == 0) return SK_OK;
  SK_CHECK_ARG(emb && cols, "skhip_conv_windows: null pointer");
  const int64_t total = (int64_t)B * (S - filter_size + 1) * filter_size * E;
  const int64_t want = (total + 255) / 256, cap = (int64_t)sk_num_cus() * 16;
  const unsigned blocks = (unsigned)(want < cap ? want : cap);
  if (dtype == SK_BF16)
    hipLaunchKernelGGL((conv_windows_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)emb, off,
                       (bf16_t*)cols, B, L, S, filter_size, E, lde, ldcols, pad_row, stride_b, stride_t);
  else
    hipLaunchKernelGGL((conv_windows_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)emb, off, (float*)cols,
                       B, L, S, filter_size, E, lde, ldcols, pad_row, stride_b, stride_t);
  SK_CHECK_LAUNCH("skhip_conv_windows");
  return SK_OK;
}

extern "C" int skhip_cnn_head(const void* h, const float* w, const float* bias, float* y, float* terms, float* loss, int B,
                              int P, int F, int64_t ldh, int target_label, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(dtype == SK_F32 || dtype == SK_BF16, "skhip_cnn_head: bad dtype");
  SK_CHECK_ARG(B >= 1 && P >= 1 && F >= 1 && ldh >= F, "skhip_cnn_head: bad shape B=%d P=%d F=%d ldh=%lld", B, P, F,
               (long long)ldh);
  SK_CHECK_ARG(target_label == 0 || target_label == 1, "skhip_cnn_head: target_label=%d", target_label);
  SK_CHECK_ARG(h && w && bias && y && terms && loss, "skhip_cnn_head: null pointer");
  if (dtype == SK_BF16)
    hipLaunchKernelGGL((cnn_head_kernel<bf16_t>), dim3(B), dim3(64), 0, stream, (const bf16_t*)h, w, bias, y, terms, P, F, ldh,
                       target_label);
  else
    hipLaunchKernelGGL((cnn_head_kernel<float>), dim3(B), dim3(64), 0, stream, (const float*)h, w, bias, y, terms, P, F, ldh,
                       target_label);
  hipLaunchKernelGGL(mean_terms_kernel, dim3(1), dim3(64), 0, stream, (const float*)terms, loss, B);
  SK_CHECK_LAUNCH("skhip_cnn_head");
  return SK_OK;
}
