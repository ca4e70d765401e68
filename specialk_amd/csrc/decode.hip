// Beam-search decode step kernels: single-query attention over a KV cache, fused
// log-softmax + beam expansion + top-k, and cache row gather.
//
// The reference re-runs the whole decoder on the whole prefix at every step
// (lib/transformer/Translator.py:130-140) and keeps beams as Python objects whose advance()
// calls topk twice and syncs with .item() (lib/transformer/Beam.py:46-79).  Here a step only
// computes the newest position against cached keys/values, and the beam bookkeeping of all
// sentences is two launches with no host round trip.
#include "common.h"

namespace {

// ------------------------------------------------------------------ single-query attention
// One wave per (row, head).  Keys sweep in tiles of 64 (lane = key), online softmax, then
// each lane accumulates output column(s) lane (+64) with v_readlane broadcasts.  K and V are
// read straight from HBM/L2 once: the step is bandwidth-bound on the cache.
template <typename T>
__global__ __launch_bounds__(256) void decode_attn_kernel(const T* __restrict__ q, const T* __restrict__ Kc,
                                                          const T* __restrict__ Vc, T* __restrict__ out,
                                                          const int32_t* __restrict__ kv_len,
                                                          const uint8_t* __restrict__ key_pad,
                                                          const int32_t* __restrict__ hist_row,
                                                          const T* __restrict__ new_k, const T* __restrict__ new_v,
                                                          int64_t ld_new, T* Kw, T* Vw, int R, int H, int len,
                                                          int Lmax, int dk, int dv, int64_t ldq, int64_t ldo, float scale,
                                                          int kv_row_div, int vec_k, int vec_v) {
  __shared__ float qs[4][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pair = blockIdx.x * 4 + wave;
  if (pair >= R * H) return;  // wave-uniform
  const int r = pair / H, h = pair - r * H;
  const int kr_row = r / kv_row_div;
  const int n = kv_len ? kv_len[kr_row] : len;
  for (int c = lane; c < dk; c += 64) qs[wave][c] = Cvt<T>::to_f32(q[(int64_t)r * ldq + h * dk + c]) * scale;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  // hist_row (optional, [R][Lmax]): the cache row that holds position j of this query row's history
  // (beam search re-parents hypotheses every step; following the table replaces re-ordering the caches)
  const int32_t* hr = hist_row ? hist_row + (int64_t)kr_row * Lmax : nullptr;
  const T* Kh = Kc + h * dk;
  const T* Vh = Vc + h * dv;
  float m = -INFINITY, l = 0.f, acc0 = 0.f, acc1 = 0.f;
  const int c0 = lane < dv ? lane : 0, c1 = lane + 64 < dv ? lane + 64 : 0;
  // P.V with 16-byte loads: lane = (row group rg, column chunk ch); a lane accumulates its VN columns
  // over rows rg, rg + nrg, ... of every tile (independent loads, all in flight together) and the
  // row groups are summed across lanes once at the end.
  constexpr int VN = Vec16<T>::N;
  const int nch = vec_v ? dv / VN : 1, nrg = 64 / nch;
  const int ch = lane & (nch - 1), rg = lane / nch;
  float accv[VN];
#pragma unroll
  for (int e = 0; e < VN; ++e) accv[e] = 0.f;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    float s = -INFINITY;
    const int64_t jrow = (int64_t)((hr && j < n) ? hr[j] : kr_row) * Lmax + j;  // cache (row, position) of key j
    // new_k / new_v: key and value of the newest position (n - 1) come from the projection output
    // and are appended to the cache here, instead of by a projection launch of their own
    const bool fresh = new_k && j == n - 1;
    if (j < n && (fresh || !(key_pad && key_pad[jrow]))) {
      const T* kr = fresh ? new_k + (int64_t)r * ld_new + h * dk : Kh + jrow * H * dk;
      T* kw = Kw + h * dk + jrow * H * dk;
      float a = 0.f;
      if (vec_k) {  // 16-byte loads of the lane's key row
        constexpr int VN = Vec16<T>::N;
        for (int c = 0; c < dk; c += VN) {
          Vec16<T> kv;
          kv.v = *(const decltype(kv.v)*)(kr + c);
          if (fresh) *(decltype(kv.v)*)(kw + c) = kv.v;
#pragma unroll
          for (int e = 0; e < VN; ++e) a += qs[wave][c + e] * kv.get(e);
        }
      } else {
        for (int c = 0; c < dk; ++c) {
          if (fresh) kw[c] = kr[c];
          a += qs[wave][c] * Cvt<T>::to_f32(kr[c]);
        }
      }
      s = (key_pad && key_pad[jrow]) ? -INFINITY : a;
    }
    float mn = fmaxf(m, wave_max(s));
    float pr = 0.f, alpha = 1.f;
    if (mn != -INFINITY) { alpha = expf(m - mn); pr = expf(s - mn); }
    l = l * alpha + wave_sum(pr);
    m = mn;
    if (vec_v) {
#pragma unroll
      for (int e = 0; e < VN; ++e) accv[e] *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) {  // nch <= 16 for dv <= 128
        if (i >= nch) break;
        const int t = rg + nrg * i;
        const float cf = __shfl(pr, t, 64);
        const int64_t trow = __shfl(jrow, t, 64);
        if (j0 + t < n) {
          Vec16<T> vv;
          const bool fresh_v = new_v && j0 + t == n - 1;
          vv.v = *(const decltype(vv.v)*)((fresh_v ? new_v + (int64_t)r * ld_new + h * dv : Vh + trow * H * dv) + ch * VN);
          if (fresh_v) *(decltype(vv.v)*)(Vw + h * dv + trow * H * dv + ch * VN) = vv.v;
#pragma unroll
          for (int e = 0; e < VN; ++e) accv[e] += cf * vv.get(e);
        }
      }
      continue;
    }
    acc0 *= alpha; acc1 *= alpha;
    const int cnt = min(64, n - j0);
    for (int t = 0; t < cnt; ++t) {
      float cf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pr), t));
      const int64_t trow = __shfl(jrow, t, 64);
      const bool fresh_v = new_v && j0 + t == n - 1;
      const T* vr = fresh_v ? new_v + (int64_t)r * ld_new + h * dv : Vh + trow * H * dv;
      if (fresh_v) {
        T* vw = Vw + h * dv + trow * H * dv;
        if (lane < dv) vw[lane] = vr[lane];
        if (lane + 64 < dv) vw[lane + 64] = vr[lane + 64];
      }
      acc0 += cf * Cvt<T>::to_f32(vr[c0]);
      if (dv > 64) acc1 += cf * Cvt<T>::to_f32(vr[c1]);
    }
  }
  const float inv = 1.0f / l;
  if (vec_v) {
    for (int o = nch; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < VN; ++e) accv[e] += __shfl_xor(accv[e], o, 64);
    if (rg == 0) {
      Vec16<T> ov;
#pragma unroll
      for (int e = 0; e < VN; ++e) ov.set(e, accv[e] * inv);
      *(decltype(ov.v)*)(out + (int64_t)r * ldo + h * dv + ch * VN) = ov.v;
    }
    return;
  }
  if (lane < dv) out[(int64_t)r * ldo + h * dv + lane] = Cvt<T>::from_f32(acc0 * inv);
  if (lane + 64 < dv) out[(int64_t)r * ldo + h * dv + lane + 64] = Cvt<T>::from_f32(acc1 * inv);
}

// ------------------------------------------------------------------ beam top-k
// (value, index) ordering used everywhere: larger value first, lower index on ties.
__device__ __forceinline__ bool better(float v, int i, float ov, int oi) { return v > ov || (v == ov && i < oi); }

__device__ __forceinline__ void wave_argbest(float& v, int& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(v, o, 64);
    int oi = __shfl_xor(i, o, 64);
    if (better(ov, oi, v, i)) { v = ov; i = oi; }
  }
}

// Pass A: one block per (sentence, beam) row: log-sum-exp over V, then the row's best KC
// tokens.  Candidate value = (x - max) - log(sum) + score_in, the same operation order as
// F.log_softmax followed by `word_prob + scores` (Translator.py:137, Beam.py:52).
// normalize == 2: as 1 with the hardware exponential (v_exp_f32, ~1 ulp) instead of expf in the normaliser; bf16
// models decode with it, fp32 models keep expf and with it the reference's bits.
// Measured and not kept: the row held in registers (one pass over memory, every load in flight at once): 102 us
// against 67 us (fewer waves to issue the instructions that bound this kernel); one candidate list per WAVE with a
// vote per element instead of a list per thread: 75 us against 61 us (a cross-lane arg-max per accepted element).
template <int KC>
__global__ __launch_bounds__(256) void beam_row_topk_kernel(const float* __restrict__ logits,
                                                            const float* __restrict__ scores_in,
                                                            float* __restrict__ cand_val, int32_t* __restrict__ cand_idx,
                                                            int bm, int64_t V, int first_step, int normalize) {
  __shared__ float s_red[4];
  __shared__ float s_val[4];
  __shared__ int s_idx[4];
  __shared__ int s_best_i;
  const int row = blockIdx.x, beam = row % bm;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* cv = cand_val + (int64_t)row * bm;
  int32_t* ci = cand_idx + (int64_t)row * bm;
  if (first_step && beam != 0) {  // Beam.py:51-54: only word_prob[0] takes part in the first step
    if (threadIdx.x < bm) { cv[threadIdx.x] = -INFINITY; ci[threadIdx.x] = 0x7fffffff; }
    return;
  }
  const float* x = logits + (int64_t)row * V;
  // thread-local top-KC, kept sorted (best first)
  float tv[KC];
  int ti[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) { tv[k] = -INFINITY; ti[k] = 0x7fffffff; }
  float mx = -INFINITY;
  auto visit = [&](float v, int c) {
    mx = fmaxf(mx, v);
    if (better(v, c, tv[KC - 1], ti[KC - 1])) {
      tv[KC - 1] = v; ti[KC - 1] = c;
#pragma unroll
      for (int k = KC - 1; k > 0; --k)
        if (better(tv[k], ti[k], tv[k - 1], ti[k - 1])) {
          float fv = tv[k]; tv[k] = tv[k - 1]; tv[k - 1] = fv;
          int fi = ti[k]; ti[k] = ti[k - 1]; ti[k - 1] = fi;
        }
    }
  };
  const bool vec = (V & 3) == 0 && ((uintptr_t)x & 15) == 0;
  if (vec) {  // eight 16-byte loads in flight per thread (the row streams from HBM: a thread's share is ~35 loads)
    const float4* x4 = (const float4*)x;
    const int64_t n4 = V >> 2;
    int64_t c = threadIdx.x;
    for (; c + 7 * 256 < n4; c += 8 * 256) {
      float4 a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = x4[c + 256 * u];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t e = c + 256 * u;
        visit(a[u].x, (int)(4 * e)); visit(a[u].y, (int)(4 * e + 1)); visit(a[u].z, (int)(4 * e + 2)); visit(a[u].w, (int)(4 * e + 3));
      }
    }
    for (; c + 256 < n4; c += 512) {
      const float4 a = x4[c], b = x4[c + 256];
      visit(a.x, (int)(4 * c)); visit(a.y, (int)(4 * c + 1)); visit(a.z, (int)(4 * c + 2)); visit(a.w, (int)(4 * c + 3));
      const int64_t e = c + 256;
      visit(b.x, (int)(4 * e)); visit(b.y, (int)(4 * e + 1)); visit(b.z, (int)(4 * e + 2)); visit(b.w, (int)(4 * e + 3));
    }
    for (; c < n4; c += 256) {
      const float4 a = x4[c];
      visit(a.x, (int)(4 * c)); visit(a.y, (int)(4 * c + 1)); visit(a.z, (int)(4 * c + 2)); visit(a.w, (int)(4 * c + 3));
    }
  } else {
    for (int64_t c = threadIdx.x; c < V; c += 256) visit(x[c], (int)c);
  }
  mx = wave_max(mx);
  if (lane == 0) s_red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  __syncthreads();
  float logse = 0.f;
  if (normalize) {
    float se = 0.f;
    if (vec) {   // four loads in flight, summed in the order of the plain loop
      const float4* x4 = (const float4*)x;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      const bool fast = normalize == 2;
      auto ex = [&](float t) { return fast ? __expf(t) : expf(t); };
      int64_t c = threadIdx.x;
      for (; c + 3 * 256 < (V >> 2); c += 4 * 256) {
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = x4[c + 256 * u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          s0 += ex(q[u].x - mx); s1 += ex(q[u].y - mx); s2 += ex(q[u].z - mx); s3 += ex(q[u].w - mx);
        }
      }
      for (; c < (V >> 2); c += 256) {
        const float4 a = x4[c];
        s0 += ex(a.x - mx); s1 += ex(a.y - mx); s2 += ex(a.z - mx); s3 += ex(a.w - mx);
      }
      se = (s0 + s1) + (s2 + s3);
    } else {
      for (int64_t c = threadIdx.x; c < V; c += 256) se += expf(x[c] - mx);
    }
    se = wave_sum(se);
    if (lane == 0) s_red[wave] = se;
    __syncthreads();
    se = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    logse = logf(se);
  } else {
    mx = 0.f;
  }
  const float sc = first_step ? 0.f : scores_in[row];
  // bm rounds: block-wide best of the per-thread heads; the winner pops its head.
  for (int round = 0; round < bm; ++round) {
    float v = tv[0];
    int i = ti[0];
    wave_argbest(v, i);
    if (lane == 0) { s_val[wave] = v; s_idx[wave] = i; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float bv = s_val[0]; int bi = s_idx[0];
#pragma unroll
      for (int w = 1; w < 4; ++w) if (better(s_val[w], s_idx[w], bv, bi)) { bv = s_val[w]; bi = s_idx[w]; }
      s_best_i = bi;
      float lp = (bv - mx) - logse;
      cv[round] = first_step ? lp : lp + sc;
      ci[round] = bi;
    }
    __syncthreads();
    if (ti[0] == s_best_i && s_best_i != 0x7fffffff) {
#pragma unroll
      for (int k = 0; k < KC - 1; ++k) { tv[k] = tv[k + 1]; ti[k] = ti[k + 1]; }
      tv[KC - 1] = -INFINITY; ti[KC - 1] = 0x7fffffff;
    }
    __syncthreads();
  }
}

// Pass A with chunk statistics: the generator product has left, per row and 64-column chunk, the chunk's maximum
// and its sum of exponentials against that maximum (SK_EPI_ROWSTATS).  One wave per row: the normaliser comes from
// the statistics alone, and only the chunks whose maximum reaches the bm-th largest chunk maximum can hold one of the
// row's best bm tokens (the bm largest chunk maxima are bm different elements, so the bm-th best element is at
// least the bm-th largest chunk maximum, and every better element lifts its chunk's maximum above it): those few
// chunks are read, 64 logits at a time, instead of the whole row twice.  Same (value, index) order as above.
constexpr int STAT_SLOTS = 16;   // chunk statistics per lane: rows of up to 64 * 64 * 16 = 65 536 tokens
template <int KC>
__device__ __forceinline__ void row_topk_stats(const float* logits, const float2* stats, const float* scores_in,
                                               float* cand_val, int32_t* cand_idx, int bm, int64_t V, int first_step,
                                               int row, int lane, int cw) {
  const int beam = row % bm;
  float* cv = cand_val + (int64_t)row * bm;
  int32_t* ci = cand_idx + (int64_t)row * bm;
  if (first_step && beam != 0) {
    if (lane < bm) { cv[lane] = -INFINITY; ci[lane] = 0x7fffffff; }
    return;
  }
  const int nchunk = (int)((V + cw - 1) / cw);   // cw: columns per statistics chunk (at most 64: a lane per column)
  const float2* st = stats + (int64_t)row * nchunk;
  float cm[STAT_SLOTS], cs[STAT_SLOTS];
  float mx = -INFINITY;
#pragma unroll
  for (int u = 0; u < STAT_SLOTS; ++u) {
    const int c = lane + 64 * u;
    const float2 v = c < nchunk ? st[c] : make_float2(-INFINITY, 0.f);
    cm[u] = v.x; cs[u] = v.y;
    mx = fmaxf(mx, v.x);
  }
  mx = wave_max(mx);
  float se = 0.f;
#pragma unroll
  for (int u = 0; u < STAT_SLOTS; ++u) se += cm[u] == -INFINITY ? 0.f : cs[u] * __expf(cm[u] - mx);
  se = wave_sum(se);
  const float logse = logf(se);
  // the bm-th largest chunk maximum: bm rounds of (best remaining, lowest chunk on ties)
  float rem[STAT_SLOTS];
#pragma unroll
  for (int u = 0; u < STAT_SLOTS; ++u) rem[u] = cm[u];
  float thr = -INFINITY;
  for (int round = 0; round < bm; ++round) {
    float bv = -INFINITY; int bc = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < STAT_SLOTS; ++u)
      if (better(rem[u], lane + 64 * u, bv, bc)) { bv = rem[u]; bc = lane + 64 * u; }
    wave_argbest(bv, bc);
    thr = bv;
#pragma unroll
    for (int u = 0; u < STAT_SLOTS; ++u) if (lane + 64 * u == bc) rem[u] = -INFINITY;
  }
  // thread-local top-KC over the candidate chunks' elements (a lane sees one element per chunk)
  float tv[KC];
  int ti[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) { tv[k] = -INFINITY; ti[k] = 0x7fffffff; }
  const float* x = logits + (int64_t)row * V;
  auto visit = [&](float v, int idx) {
    if (better(v, idx, tv[KC - 1], ti[KC - 1])) {
      tv[KC - 1] = v; ti[KC - 1] = idx;
#pragma unroll
      for (int k = KC - 1; k > 0; --k)
        if (better(tv[k], ti[k], tv[k - 1], ti[k - 1])) {
          float fv = tv[k]; tv[k] = tv[k - 1]; tv[k - 1] = fv;
          int fi = ti[k]; ti[k] = ti[k - 1]; ti[k - 1] = fi;
        }
    }
  };
  // the candidate chunks, eight at a time with their loads in flight together (there are bm of them unless chunk
  // maxima tie)
  constexpr int BATCH = 8;
  int batch[BATCH], nb = 0;
  auto flush = [&]() {
    float v[BATCH];
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      const int64_t c = cw * (int64_t)batch[k < nb ? k : 0] + lane;
      const bool in = k < nb && lane < cw && c < V;
      v[k] = in ? x[in ? c : 0] : -INFINITY;
    }
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      const int64_t c = cw * (int64_t)batch[k < nb ? k : 0] + lane;
      if (k < nb) visit(v[k], lane < cw && c < V ? (int)c : 0x7fffffff);
    }
    nb = 0;
  };
#pragma unroll 1
  for (int u = 0; u < STAT_SLOTS; ++u) {
    uint64_t todo = __ballot(cm[u] >= thr && cm[u] != -INFINITY);
    while (todo) {
      batch[nb++] = __builtin_ctzll(todo) + 64 * u;
      todo &= todo - 1;
      if (nb == BATCH) flush();
    }
  }
  if (nb) flush();
  const float sc = first_step ? 0.f : scores_in[row];
  for (int round = 0; round < bm; ++round) {
    float bv = tv[0]; int bi = ti[0];
    wave_argbest(bv, bi);
    if (lane == 0) {
      const float lp = (bv - mx) - logse;
      cv[round] = first_step ? lp : lp + sc;
      ci[round] = bi;
    }
    if (ti[0] == bi && bi != 0x7fffffff) {
#pragma unroll
      for (int k = 0; k < KC - 1; ++k) { tv[k] = tv[k + 1]; ti[k] = ti[k + 1]; }
      tv[KC - 1] = -INFINITY; ti[KC - 1] = 0x7fffffff;
    }
  }
}

template <int KC>
__global__ __launch_bounds__(64) void beam_row_topk_stats_kernel(const float* __restrict__ logits,
                                                                 const float2* __restrict__ stats,
                                                                 const float* __restrict__ scores_in,
                                                                 float* __restrict__ cand_val, int32_t* __restrict__ cand_idx,
                                                                 int bm, int64_t V, int first_step, int cw) {
  row_topk_stats<KC>(logits, stats, scores_in, cand_val, cand_idx, bm, V, first_step, blockIdx.x, threadIdx.x, cw);
}

// Pass B: one wave per sentence merges its bm*bm row candidates into the new beam,
// ordered like flat_beam_lk.topk(size, 0, True, True) with flat id = beam*V + token.
__device__ __forceinline__ void beam_merge(const float* cand_val, const int32_t* cand_idx, float* scores_out,
                                           int32_t* prev_out, int32_t* tok_out, int bm, int64_t V, int sent, int lane) {
  const int n = bm * bm;  // <= 256
  float v[4];
  int id[4];  // index into the sentence's candidate list; flat order = (beam, rank) and within a
              // beam candidates are already sorted, so comparing (value, beam*V+tok) needs the token
  long long flat[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int e = lane + 64 * k;
    if (e < n) {
      v[k] = cand_val[(int64_t)sent * n + e];
      int tok = cand_idx[(int64_t)sent * n + e];
      id[k] = e;
      flat[k] = tok == 0x7fffffff ? 0x7fffffffffffffffll : (long long)(e / bm) * V + tok;
    } else { v[k] = -INFINITY; id[k] = -1; flat[k] = 0x7fffffffffffffffll; }
  }
  for (int round = 0; round < bm; ++round) {
    float bv = -INFINITY; long long bf = 0x7fffffffffffffffll; int bk = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (v[k] > bv || (v[k] == bv && flat[k] < bf)) { bv = v[k]; bf = flat[k]; bk = k; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ov = __shfl_xor(bv, o, 64);
      long long of = __shfl_xor(bf, o, 64);
      if (ov > bv || (ov == bv && of < bf)) { bv = ov; bf = of; }
    }
    // the unique owner of (bv, bf) retires it
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (flat[k] == bf && v[k] == bv && bf != 0x7fffffffffffffffll) { v[k] = -INFINITY; flat[k] = 0x7fffffffffffffffll; }
    (void)bk;
    if (lane == 0) {
      int64_t o = (int64_t)sent * bm + round;
      scores_out[o] = bv;
      prev_out[o] = bf == 0x7fffffffffffffffll ? 0 : (int32_t)(bf / V);
      tok_out[o] = bf == 0x7fffffffffffffffll ? 0 : (int32_t)(bf % V);
    }
  }
}

__global__ __launch_bounds__(64) void beam_merge_kernel(const float* __restrict__ cand_val,
                                                        const int32_t* __restrict__ cand_idx,
                                                        float* __restrict__ scores_out, int32_t* __restrict__ prev_out,
                                                        int32_t* __restrict__ tok_out, int bm, int64_t V) {
  beam_merge(cand_val, cand_idx, scores_out, prev_out, tok_out, bm, V, blockIdx.x, threadIdx.x);
}

// ------------------------------------------------------------------ row gather
template <int VB>  // bytes per access
__global__ __launch_bounds__(256) void gather_rows_kernel(const char* __restrict__ src, char* __restrict__ dst,
                                                          const int32_t* __restrict__ src_row, int64_t row_bytes,
                                                          int64_t src_stride_b, int64_t dst_stride_b) {
  const int64_t r = blockIdx.y;
  const char* s = src + (int64_t)src_row[r] * src_stride_b;
  char* d = dst + r * dst_stride_b;
  for (int64_t o = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VB; o < row_bytes; o += (int64_t)gridDim.x * 256 * VB) {
    if (VB == 16) *(uint4*)(d + o) = *(const uint4*)(s + o);
    else if (VB == 4) *(uint32_t*)(d + o) = *(const uint32_t*)(s + o);
    else *(uint16_t*)(d + o) = *(const uint16_t*)(s + o);
  }
}

// ------------------------------------------------------------------ beam state update
// Everything Beam.advance / Translator.collect_active_inst_idx_list / prepare_beam_dec_seq do between two
// decoder steps (lib/transformer/Beam.py:46-79, Translator.py:113-168), for all sentences, in one launch with
// every step-dependent quantity read from and written to device memory (so a step can be replayed from a
// captured graph).  One wave per beam row r = (sentence s, beam b); t = cur_len[r] is the step just taken:
//   scores[s] <- new_scores[s] unless sentence s had finished; back-pointers and tokens recorded in ks[t-1], ys[t];
//   sentence finishes (length[s] = t, done[s] = 1, *n_done += 1) when its best beam's token is EOS;
//   history table re-parented: hist_out[r][0..t) = hist_in[s*bm + prev[s][b]][0..t), hist_out[r][t] = r;
//   inputs of the next step: tok[r], pos[r] = t + 1, row_pad[r] and pad_hist[r][t] = (token is PAD), cur_len[r] = t + 1.
struct BeamState {
  float* scores;
  int32_t *ys, *ks, *done, *length, *n_done, *cur_len;
  int64_t *tok, *pos;
  uint8_t *row_pad, *pad_hist;
  const int32_t* hist_in;
  int32_t* hist_out;
  int n_sent, bm, Tc, max_len, pad_id, eos_id;
};

__device__ __forceinline__ void beam_update_row(const float* new_scores, const int32_t* prev, const int32_t* new_tok,
                                                const BeamState& st, int r, int lane) {
  float* scores = st.scores;
  int32_t *ys = st.ys, *ks = st.ks, *done = st.done, *length = st.length, *n_done = st.n_done, *cur_len = st.cur_len;
  int64_t *tok = st.tok, *pos = st.pos;
  uint8_t *row_pad = st.row_pad, *pad_hist = st.pad_hist;
  const int32_t* hist_in = st.hist_in;
  int32_t* hist_out = st.hist_out;
  const int n_sent = st.n_sent, bm = st.bm, Tc = st.Tc, max_len = st.max_len, pad_id = st.pad_id, eos_id = st.eos_id;
  const int s = r / bm, b = r - s * bm;
  const int t = cur_len[r];
  if (t < 1 || t > max_len) return;  // past the last step (a replayed graph may overshoot): nothing to do
  const int R = n_sent * bm;
  const int p = prev[r], w = new_tok[r];
  const int32_t* hin = hist_in + (int64_t)(s * bm + p) * Tc;
  int32_t* hout = hist_out + (int64_t)r * Tc;
  for (int j = lane; j < t; j += 64) hout[j] = hin[j];
  if (lane == 0) {
    const bool is_pad = w == pad_id;
    if (t < Tc) { hout[t] = r; pad_hist[(int64_t)r * Tc + t] = is_pad; }
    tok[r] = w;
    pos[r] = t + 1;
    row_pad[r] = is_pad;
    cur_len[r] = t + 1;
    ks[(int64_t)(t - 1) * R + r] = p;
    ys[(int64_t)t * R + r] = w;
    if (b == 0 && !done[s]) {
      for (int k = 0; k < bm; ++k) scores[s * bm + k] = new_scores[s * bm + k];
      if (w == eos_id) { length[s] = t; done[s] = 1; atomicAdd(n_done, 1); }
    }
  }
}

__global__ __launch_bounds__(64) void beam_update_kernel(const float* __restrict__ new_scores, const int32_t* __restrict__ prev,
                                                         const int32_t* __restrict__ new_tok, BeamState st) {
  beam_update_row(new_scores, prev, new_tok, st, blockIdx.x, threadIdx.x);
}

// Hypotheses of all sentences at the end of a search: Beam.get_hypothesis (lib/transformer/Beam.py:103-110) walks the
// back-pointers from the last step to the first for one beam; here a workgroup per sentence brings the sentence's
// columns of ys / ks (the tokens and back-pointers of every step, [step][R]) into LDS and one thread per wanted rank
// walks them.  out[s][r][j] = token j of the hypothesis that ended as rank r, j < T = min(length[s], steps).
__global__ __launch_bounds__(64) void beam_backtrack_kernel(const int32_t* __restrict__ ys, const int32_t* __restrict__ ks,
                                                            const int32_t* __restrict__ length, int32_t* __restrict__ out,
                                                            int R, int bm, int n_best, int steps, int64_t out_ld) {
  extern __shared__ int32_t bt[];
  const int s = blockIdx.x, T = min(length[s], steps);
  int32_t* y = bt;              // [T][bm]: ys[j + 1]
  int32_t* k = bt + steps * bm; // [T][bm]: ks[j]
  for (int i = threadIdx.x; i < T * bm; i += 64) {
    const int j = i / bm, b = i - j * bm;
    y[i] = ys[(int64_t)(j + 1) * R + s * bm + b];
    k[i] = ks[(int64_t)j * R + s * bm + b];
  }
  __syncthreads();
  if ((int)threadIdx.x < n_best) {
    int32_t* o = out + ((int64_t)s * n_best + threadIdx.x) * out_ld;
    int b = threadIdx.x;
    for (int j = T - 1; j >= 0; --j) {
      o[j] = y[j * bm + b];
      b = k[j * bm + b];
    }
  }
}

// The three passes of a step in one launch: a workgroup per sentence, a wave per beam row.  The row candidates and
// the merged beam travel through global memory as in the separate launches (same bytes, same results); what goes
// is two of the three kernel boundaries of a step, some 10 us each inside a replayed graph.
template <int KC>
__global__ __launch_bounds__(1024) void beam_step_kernel(const float* logits, const float2* stats, float* cand_val,
                                                         int32_t* cand_idx, float* new_scores, int32_t* prev,
                                                         int32_t* new_tok, int64_t V, int first_step, int cw, BeamState st) {
  const int sent = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = sent * st.bm + wave;
  row_topk_stats<KC>(logits, stats, st.scores, cand_val, cand_idx, st.bm, V, first_step, row, lane, cw);
  __syncthreads();
  if (wave == 0) beam_merge(cand_val, cand_idx, new_scores, prev, new_tok, st.bm, V, sent, lane);
  __syncthreads();
  beam_update_row(new_scores, prev, new_tok, st, row, lane);
}

}  // namespace

extern "C" int skhip_decode_attn(const void* q, void* Kc, void* Vc, void* out, const int32_t* kv_len,
                                 const uint8_t* key_pad, const int32_t* hist_row, const void* new_k, const void* new_v,
                                 int64_t ld_new, int R, int n_head, int len, int Lmax, int dk, int dv, int64_t ldq,
                                 int64_t ldo, float scale, int kv_row_div, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(dtype == SK_F32 || dtype == SK_BF16, "skhip_decode_attn: bad dtype");
  SK_CHECK_ARG(kv_row_div >= 1, "skhip_decode_attn: kv_row_div=%d", kv_row_div);
  SK_CHECK_ARG(R >= 0 && n_head > 0 && dk > 0 && dv > 0 && dk <= 128 && dv <= 128, "skhip_decode_attn: bad shape");
  SK_CHECK_ARG(len >= 0 && len <= Lmax, "skhip_decode_attn: len=%d exceeds Lmax=%d", len, Lmax);
  if (R == 0) return SK_OK;
  SK_CHECK_ARG(q && Kc && Vc && out, "skhip_decode_attn: null pointer");
  SK_CHECK_ARG(!new_k == !new_v, "skhip_decode_attn: new_k and new_v come together");
  SK_CHECK_ARG(!new_k || (kv_row_div == 1 && (kv_len || len >= 1) && ld_new >= (int64_t)n_head * (dk > dv ? dk : dv)),
               "skhip_decode_attn: appending needs one cache row per query row, a length >= 1 and ld_new >= row width");
  unsigned blocks = (unsigned)(((int64_t)R * n_head + 3) / 4);
  const int vn = dtype == SK_BF16 ? 8 : 4;
  const int vec_k = (dk % vn == 0) && ((uintptr_t)Kc % 16 == 0);  // key rows start on 16-byte boundaries
  const int nch = dv / vn;                                          // 16-byte column chunks of a value row
  const int vec_v = (dv % vn == 0) && nch <= 16 && (nch & (nch - 1)) == 0 && ((uintptr_t)Vc % 16 == 0) &&
                    ((uintptr_t)out % 16 == 0) && (ldo % vn == 0);
  if (dtype == SK_BF16)
    hipLaunchKernelGGL((decode_attn_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)q,
                       (const bf16_t*)Kc, (const bf16_t*)Vc, (bf16_t*)out, kv_len, key_pad, hist_row, (const bf16_t*)new_k,
                       (const bf16_t*)new_v, ld_new, (bf16_t*)Kc, (bf16_t*)Vc, R, n_head, len, Lmax, dk, dv,
                       ldq, ldo, scale, kv_row_div, vec_k, vec_v);
  else
    hipLaunchKernelGGL((decode_attn_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)q, (const float*)Kc,
                       (const float*)Vc, (float*)out, kv_len, key_pad, hist_row, (const float*)new_k, (const float*)new_v, ld_new, (float*)Kc, (float*)Vc,
                       R, n_head, len, Lmax, dk, dv, ldq, ldo, scale, kv_row_div, vec_k, vec_v);
  SK_CHECK_LAUNCH("skhip_decode_attn");
  return SK_OK;
}

extern "C" int64_t skhip_beam_topk_ws_bytes(int n_sent, int bm, int64_t V) {
  (void)V;
  return (int64_t)n_sent * bm * bm * 8;
}

extern "C" int skhip_beam_topk(const float* logits, const float* scores_in, float* scores_out, int32_t* prev_out,
                               int32_t* tok_out, void* ws, int n_sent, int bm, int64_t V, int first_step, int normalize,
                               void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(bm >= 1 && bm <= 16, "skhip_beam_topk: beam size %d not in 1..16", bm);
  SK_CHECK_ARG(V >= bm && V < (1ll << 31), "skhip_beam_topk: V=%lld", (long long)V);
  if (n_sent == 0) return SK_OK;
  SK_CHECK_ARG(logits && scores_in && scores_out && prev_out && tok_out && ws, "skhip_beam_topk: null pointer");
  float* cv = (float*)ws;
  int32_t* ci = (int32_t*)(cv + (int64_t)n_sent * bm * bm);
  dim3 grid((unsigned)(n_sent * bm));
#define BEAM_A(KC) hipLaunchKernelGGL((beam_row_topk_kernel<KC>), grid, dim3(256), 0, stream, logits, scores_in, cv, ci, bm, V, first_step, normalize)
  if (bm <= 1) BEAM_A(1);
  else if (bm <= 2) BEAM_A(2);
  else if (bm <= 4) BEAM_A(4);
  else if (bm <= 8) BEAM_A(8);
  else BEAM_A(16);
  SK_CHECK_LAUNCH("skhip_beam_topk(rows)");
  hipLaunchKernelGGL(beam_merge_kernel, dim3((unsigned)n_sent), dim3(64), 0, stream, cv, ci, scores_out, prev_out, tok_out, bm, V);
  SK_CHECK_LAUNCH("skhip_beam_topk(merge)");
  return SK_OK;
}

extern "C" int skhip_beam_topk_stats(const float* logits, const float* stats, const float* scores_in, float* scores_out,
                                     int32_t* prev_out, int32_t* tok_out, void* ws, int n_sent, int bm, int64_t V,
                                     int first_step, int chunk_cols, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(bm >= 1 && bm <= 16, "skhip_beam_topk_stats: beam size %d not in 1..16", bm);
  SK_CHECK_ARG(chunk_cols >= 1 && chunk_cols <= 64, "skhip_beam_topk_stats: chunk_cols=%d not in 1..64", chunk_cols);
  SK_CHECK_ARG(V >= bm && V <= (int64_t)chunk_cols * 64 * STAT_SLOTS, "skhip_beam_topk_stats: V=%lld not in bm..%d", (long long)V,
               chunk_cols * 64 * STAT_SLOTS);
  if (n_sent == 0) return SK_OK;
  SK_CHECK_ARG(logits && stats && scores_in && scores_out && prev_out && tok_out && ws && (uintptr_t)stats % 8 == 0,
               "skhip_beam_topk_stats: null or misaligned pointer");
  float* cv = (float*)ws;
  int32_t* ci = (int32_t*)(cv + (int64_t)n_sent * bm * bm);
  dim3 grid((unsigned)(n_sent * bm));
#define BEAM_S(KC) hipLaunchKernelGGL((beam_row_topk_stats_kernel<KC>), grid, dim3(64), 0, stream, logits, (const float2*)stats, scores_in, cv, ci, bm, V, first_step, chunk_cols)
  if (bm <= 1) BEAM_S(1);
  else if (bm <= 2) BEAM_S(2);
  else if (bm <= 4) BEAM_S(4);
  else if (bm <= 8) BEAM_S(8);
  else BEAM_S(16);
#undef BEAM_S
  SK_CHECK_LAUNCH("skhip_beam_topk_stats(rows)");
  hipLaunchKernelGGL(beam_merge_kernel, dim3((unsigned)n_sent), dim3(64), 0, stream, cv, ci, scores_out, prev_out, tok_out, bm, V);
  SK_CHECK_LAUNCH("skhip_beam_topk_stats(merge)");
  return SK_OK;
}

extern "C" int skhip_gather_rows(const void* src, void* dst, const int32_t* src_row, int64_t n_rows, int64_t row_elems,
                                 int64_t src_stride, int64_t dst_stride, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(dtype == SK_F32 || dtype == SK_BF16, "skhip_gather_rows: bad dtype");
  SK_CHECK_ARG(n_rows >= 0 && n_rows <= 65535 && row_elems >= 0, "skhip_gather_rows: bad shape");
  if (n_rows == 0 || row_elems == 0) return SK_OK;
  SK_CHECK_ARG(src && dst && src_row && src != dst, "skhip_gather_rows: null or aliased pointer");
  const int esz = dtype == SK_BF16 ? 2 : 4;
  const int64_t rb = row_elems * esz, ss = src_stride * esz, ds = dst_stride * esz;
  const bool v16 = rb % 16 == 0 && ss % 16 == 0 && ds % 16 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0;
  const int vb = v16 ? 16 : esz;
  int64_t bx = (rb / vb + 255) / 256;
  if (bx > 64) bx = 64;
  if (bx < 1) bx = 1;
  dim3 grid((unsigned)bx, (unsigned)n_rows);
  if (vb == 16)
    hipLaunchKernelGGL((gather_rows_kernel<16>), grid, dim3(256), 0, stream, (const char*)src, (char*)dst, src_row, rb, ss, ds);
  else if (vb == 4)
    hipLaunchKernelGGL((gather_rows_kernel<4>), grid, dim3(256), 0, stream, (const char*)src, (char*)dst, src_row, rb, ss, ds);
  else
    hipLaunchKernelGGL((gather_rows_kernel<2>), grid, dim3(256), 0, stream, (const char*)src, (char*)dst, src_row, rb, ss, ds);
  SK_CHECK_LAUNCH("skhip_gather_rows");
  return SK_OK;
}

extern "C" int skhip_beam_update(const float* new_scores, const int32_t* prev, const int32_t* new_tok, float* scores,
                                 int32_t* ys, int32_t* ks, int32_t* done, int32_t* length, int32_t* n_done,
                                 int32_t* cur_len, int64_t* tok, int64_t* pos, uint8_t* row_pad, uint8_t* pad_hist,
                                 const int32_t* hist_in, int32_t* hist_out, int n_sent, int bm, int Tc, int max_len,
                                 int pad_id, int eos_id, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(n_sent >= 0 && bm >= 1 && Tc >= 1 && max_len >= 1, "skhip_beam_update: bad shape");
  if (n_sent == 0) return SK_OK;
  SK_CHECK_ARG(new_scores && prev && new_tok && scores && ys && ks && done && length && n_done && cur_len && tok && pos &&
               row_pad && pad_hist && hist_in && hist_out && hist_in != hist_out, "skhip_beam_update: null or aliased pointer");
  const BeamState st{scores, ys, ks, done, length, n_done, cur_len, tok, pos, row_pad, pad_hist, hist_in, hist_out,
                     n_sent, bm, Tc, max_len, pad_id, eos_id};
  hipLaunchKernelGGL(beam_update_kernel, dim3((unsigned)(n_sent * bm)), dim3(64), 0, stream, new_scores, prev, new_tok, st);
  SK_CHECK_LAUNCH("skhip_beam_update");
  return SK_OK;
}

extern "C" int skhip_beam_step(const float* logits, const float* stats, float* new_scores, int32_t* prev, int32_t* new_tok,
                               void* ws, float* scores, int32_t* ys, int32_t* ks, int32_t* done, int32_t* length,
                               int32_t* n_done, int32_t* cur_len, int64_t* tok, int64_t* pos, uint8_t* row_pad,
                               uint8_t* pad_hist, const int32_t* hist_in, int32_t* hist_out, int n_sent, int bm, int64_t V,
                               int first_step, int Tc, int max_len, int pad_id, int eos_id, int chunk_cols, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(bm >= 1 && bm <= 16, "skhip_beam_step: beam size %d not in 1..16", bm);
  SK_CHECK_ARG(chunk_cols >= 1 && chunk_cols <= 64, "skhip_beam_step: chunk_cols=%d not in 1..64", chunk_cols);
  SK_CHECK_ARG(V >= bm && V <= (int64_t)chunk_cols * 64 * STAT_SLOTS, "skhip_beam_step: V=%lld not in bm..%d", (long long)V,
               chunk_cols * 64 * STAT_SLOTS);
  SK_CHECK_ARG(n_sent >= 0 && Tc >= 1 && max_len >= 1, "skhip_beam_step: bad shape");
  if (n_sent == 0) return SK_OK;
  SK_CHECK_ARG(logits && stats && new_scores && prev && new_tok && ws && (uintptr_t)stats % 8 == 0 && scores && ys && ks &&
               done && length && n_done && cur_len && tok && pos && row_pad && pad_hist && hist_in && hist_out &&
               hist_in != hist_out && new_scores != scores, "skhip_beam_step: null, misaligned or aliased pointer");
  float* cv = (float*)ws;
  int32_t* ci = (int32_t*)(cv + (int64_t)n_sent * bm * bm);
  const BeamState st{scores, ys, ks, done, length, n_done, cur_len, tok, pos, row_pad, pad_hist, hist_in, hist_out,
                     n_sent, bm, Tc, max_len, pad_id, eos_id};
  const dim3 grid((unsigned)n_sent), block((unsigned)(64 * bm));
#define BEAM_F(KC) hipLaunchKernelGGL((beam_step_kernel<KC>), grid, block, 0, stream, logits, (const float2*)stats, cv, ci, new_scores, prev, new_tok, V, first_step, chunk_cols, st)
  if (bm <= 1) BEAM_F(1);
  else if (bm <= 2) BEAM_F(2);
  else if (bm <= 4) BEAM_F(4);
  else if (bm <= 8) BEAM_F(8);
  else BEAM_F(16);
#undef BEAM_F
  SK_CHECK_LAUNCH("skhip_beam_step");
  return SK_OK;
}

extern "C" int skhip_beam_backtrack(const int32_t* ys, const int32_t* ks, const int32_t* length, int32_t* out, int n_sent,
                                    int bm, int n_best, int steps, int64_t out_ld, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(n_sent >= 0 && bm >= 1 && n_best >= 1 && n_best <= bm && n_best <= 64 && steps >= 0 && out_ld >= steps,
               "skhip_beam_backtrack: bad shape (n_best %d of beam %d, %d steps, row stride %lld)", n_best, bm, steps,
               (long long)out_ld);
  const int64_t lds = 2ll * steps * bm * 4;
  SK_CHECK_ARG(lds <= 64 * 1024, "skhip_beam_backtrack: %d steps x beam %d do not fit 64 KB of LDS", steps, bm);
  if (n_sent == 0 || steps == 0) return SK_OK;
  SK_CHECK_ARG(ys && ks && length && out, "skhip_beam_backtrack: null pointer");
  hipLaunchKernelGGL(beam_backtrack_kernel, dim3((unsigned)n_sent), dim3(64), (size_t)lds, stream, ys, ks, length, out,
                     n_sent * bm, bm, n_best, steps, out_ld);
  SK_CHECK_LAUNCH("skhip_beam_backtrack");
  return SK_OK;
}
