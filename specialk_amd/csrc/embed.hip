// Embedding gather-add and its scatter-add backward.
// Replaces `enc_output = self.src_word_emb(src_seq); enc_output += self.position_enc(src_pos)`
// (lib/transformer/Models.py:117-118, 181-182).  No sqrt(d) scaling, as in the reference.
// HBM-bound row gathers: one wave per token row, 16 bytes per lane per access.
#include "common.h"
#include <algorithm>

namespace {

template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ seq, const int64_t* __restrict__ pos,
                                                        const float* __restrict__ word, const float* __restrict__ ptab,
                                                        T* __restrict__ out, int64_t T_rows, int d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < T_rows; row += (int64_t)gridDim.x * 4) {
    const float* w = word + seq[row] * d;
    const float* p = ptab + pos[row] * d;
    T* o = out + row * d;
    for (int c = lane * 4; c < d; c += 256) {
      float4 a = *(const float4*)(w + c), b = *(const float4*)(p + c);
      o[c + 0] = Cvt<T>::from_f32(a.x + b.x);
      o[c + 1] = Cvt<T>::from_f32(a.y + b.y);
      o[c + 2] = Cvt<T>::from_f32(a.z + b.z);
      o[c + 3] = Cvt<T>::from_f32(a.w + b.w);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ seq, const T* __restrict__ dout,
                                                        float* __restrict__ dword, int64_t T_rows, int d, int64_t pad_id) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < T_rows; row += (int64_t)gridDim.x * 4) {
    int64_t id = seq[row];
    if (id == pad_id) continue;
    float* g = dword + id * d;
    const T* s = dout + row * d;
    for (int c = lane; c < d; c += 64) atomicAdd(g + c, Cvt<T>::to_f32(s[c]));
  }
}

// ---- row bookkeeping of a right-padded id batch held without its padding (one block)
// lens[b] = max(count of non-PAD ids of row b - trim, 0); off = exclusive prefix sums (off[B] = n);
// facts = {every row is right-padded, min over rows of (count - trim), max len, n}.
__global__ __launch_bounds__(1024) void pack_lens_kernel(const int64_t* __restrict__ seq, int B, int L, int trim,
                                                         int64_t pad_id, int32_t* __restrict__ lens,
                                                         int32_t* __restrict__ off, int64_t* __restrict__ facts) {
  __shared__ int s_part[1024];
  __shared__ int s_red[3][16];
  const int tid = threadIdx.x;
  const int per = (B + 1023) / 1024;          // consecutive rows per thread
  const int b0 = tid * per, b1 = min(B, b0 + per);
  int ok = 1, lo = 0x7fffffff, hi = 0, sum = 0;
  for (int b = b0; b < b1; ++b) {
    const int64_t* row = seq + (int64_t)b * L;
    int cnt = 0, last = -1;
    for (int i = 0; i < L; ++i)
      if (row[i] != pad_id) { ++cnt; last = i; }
    ok &= (last + 1 == cnt);
    const int raw = cnt - trim, ln = max(raw, 0);
    lo = min(lo, raw); hi = max(hi, ln);
    lens[b] = ln;
    sum += ln;
  }
  // block scan of the per-thread sums (Hillis-Steele over 1024 entries)
  s_part[tid] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = tid >= d ? s_part[tid - d] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  int run = s_part[tid] - sum;  // exclusive prefix of this thread's first row
  for (int b = b0; b < b1; ++b) { off[b] = run; run += lens[b]; }
  if (tid == 1023) off[B] = s_part[1023];
  // reductions: ok (and), lo (min), hi (max)
  const int lane = tid & 63, wave = tid >> 6;
  for (int o = 32; o > 0; o >>= 1) {
    ok &= __shfl_xor(ok, o, 64);
    lo = min(lo, __shfl_xor(lo, o, 64));
    hi = max(hi, __shfl_xor(hi, o, 64));
  }
  if (lane == 0) { s_red[0][wave] = ok; s_red[1][wave] = lo; s_red[2][wave] = hi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w) { ok &= s_red[0][w]; lo = min(lo, s_red[1][w]); hi = max(hi, s_red[2][w]); }
    facts[0] = ok; facts[1] = lo; facts[2] = hi; facts[3] = s_part[1023];
  }
}

// idx[r] = b * L + (r - off[b]) for packed row r of sentence b (binary search in off)
__global__ __launch_bounds__(256) void pack_index_kernel(const int32_t* __restrict__ off, int B, int L,
                                                         int64_t* __restrict__ idx) {
  const int n = off[B];
  for (int r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) {
    int lo = 0, hi = B;  // largest b with off[b] <= r
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (off[mid] <= r) lo = mid; else hi = mid;
    }
    idx[r] = (int64_t)lo * L + (r - off[lo]);
  }
}

// ---- the same bookkeeping for several batches in one launch (block = batch), with the range check of ids and
// positions that nn.Embedding makes on the host.  A wave reads a row's ids 64 at a time (one coalesced request; eight
// rows' requests in flight together), counts them with a ballot; the prefix sums run over the lengths as above.
constexpr int PACK_MAX = 4, PACK_FIELDS = 12, PACK_FACTS = 6;
struct PackDesc {
  const int64_t* seq; const int64_t* pos;   // pos: null = no position check
  int64_t ld_seq, ld_pos;                    // row strides (elements)
  int64_t n_vocab, n_pos;                    // n_vocab <= 0: no id check
  int32_t* lens; int32_t* off; int64_t* idx;
  int B, L, trim;
};
struct PackArgs {
  PackDesc d[PACK_MAX];
  int64_t pad_id;
  int64_t* facts;   // [n][PACK_FACTS]: right-padded, min (count - trim), max len, n, ids in range, positions in range
};

__global__ __launch_bounds__(1024) void pack_lens_multi_kernel(PackArgs a) {
  __shared__ int s_part[1024];
  __shared__ int s_red[5][16];
  const PackDesc& d = a.d[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = d.B, L = d.L;
  int ok = 1, lo = 0x7fffffff, hi = 0, ids_ok = 1, pos_ok = 1;
  constexpr int U = 8;
  for (int b0 = wave; b0 < B; b0 += 16 * U) {
    int cnt[U], last[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { cnt[u] = 0; last[u] = -1; }
    for (int c0 = 0; c0 < L; c0 += 64) {
      int64_t v[U], pv[U];
      const int i = c0 + lane;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int b = min(b0 + 16 * u, B - 1);
        v[u] = i < L ? d.seq[(int64_t)b * d.ld_seq + i] : a.pad_id;
        pv[u] = (i < L && d.pos) ? d.pos[(int64_t)b * d.ld_pos + i] : 0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t m = __ballot(v[u] != a.pad_id);
        cnt[u] += __popcll(m);
        if (m) last[u] = c0 + 63 - __clzll(m);
        if (d.n_vocab > 0 && (v[u] < 0 || v[u] >= d.n_vocab)) ids_ok = 0;
        if (d.pos && (pv[u] < 0 || pv[u] >= d.n_pos)) pos_ok = 0;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = b0 + 16 * u;
      if (b < B) {
        ok &= (last[u] + 1 == cnt[u]);
        const int raw = cnt[u] - d.trim, ln = max(raw, 0);
        lo = min(lo, raw); hi = max(hi, ln);
        if (lane == 0) d.lens[b] = ln;
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  // exclusive prefix sums of the lengths: a thread sums consecutive rows, the block scans the 1024 sums
  const int per = (B + 1023) / 1024;
  const int r0 = tid * per, r1 = min(B, r0 + per);
  int sum = 0;
  for (int b = r0; b < r1; ++b) sum += __hip_atomic_load(&d.lens[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  s_part[tid] = sum;
  __syncthreads();
  for (int dd = 1; dd < 1024; dd <<= 1) {
    const int v = tid >= dd ? s_part[tid - dd] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  int run = s_part[tid] - sum;
  for (int b = r0; b < r1; ++b) {
    d.off[b] = run;
    run += __hip_atomic_load(&d.lens[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid == 1023) d.off[B] = s_part[1023];
  for (int o = 32; o > 0; o >>= 1) {
    ids_ok &= __shfl_xor(ids_ok, o, 64);
    pos_ok &= __shfl_xor(pos_ok, o, 64);
  }
  if (lane == 0) { s_red[0][wave] = ok; s_red[1][wave] = lo; s_red[2][wave] = hi; s_red[3][wave] = ids_ok; s_red[4][wave] = pos_ok; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w) {
      ok &= s_red[0][w]; lo = min(lo, s_red[1][w]); hi = max(hi, s_red[2][w]);
      ids_ok &= s_red[3][w]; pos_ok &= s_red[4][w];
    }
    int64_t* f = a.facts + PACK_FACTS * blockIdx.x;
    f[0] = ok; f[1] = lo; f[2] = hi; f[3] = s_part[1023]; f[4] = ids_ok; f[5] = pos_ok;
  }
}

__global__ __launch_bounds__(256) void pack_index_multi_kernel(PackArgs a) {
  const PackDesc& d = a.d[blockIdx.y];
  const int B = d.B, n = d.off[B];
  for (int r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) {
    int lo = 0, hi = B;  // largest b with off[b] <= r
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (d.off[mid] <= r) lo = mid; else hi = mid;
    }
    d.idx[r] = (int64_t)lo * d.L + (r - d.off[lo]);
  }
}

// seq[b][i] = i-th id of sentence b (its ids are flat[off[b] .. off[b+1]-1]) or pad_id past its end;
// pos[b][i] = i + 1 where seq holds a real id, else 0
__global__ __launch_bounds__(256) void collate_kernel(const int64_t* __restrict__ flat, const int64_t* __restrict__ off,
                                                      int B, int L, int64_t pad_id, int64_t* __restrict__ seq,
                                                      int64_t* __restrict__ pos) {
  const int64_t total = (int64_t)B * L;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int b = (int)(e / L), i = (int)(e - (int64_t)b * L);
    const int64_t lo = off[b], len = off[b + 1] - lo;
    const int64_t id = i < len ? flat[lo + i] : pad_id;
    seq[e] = id;
    pos[e] = id != pad_id ? i + 1 : 0;
  }
}

}  // namespace

extern "C" int skhip_embed_fwd(const int64_t* seq, const int64_t* pos, const float* word_emb, const float* pos_table,
                               void* out, int64_t T, int d, int64_t n_vocab, int64_t n_pos, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(dtype == SK_F32 || dtype == SK_BF16, "skhip_embed_fwd: bad dtype");
  SK_CHECK_ARG(d > 0 && d % 4 == 0, "skhip_embed_fwd: d=%d must be a multiple of 4", d);
  (void)n_vocab; (void)n_pos;  // ids are validated by the host wrapper (as nn.Embedding does on CPU)
  if (T == 0) return SK_OK;
  SK_CHECK_ARG(seq && pos && word_emb && pos_table && out, "skhip_embed_fwd: null pointer");
  int64_t blocks = (T + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  if (dtype == SK_BF16)
    hipLaunchKernelGGL((embed_fwd_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, stream, seq, pos, word_emb,
                       pos_table, (bf16_t*)out, T, d);
  else
    hipLaunchKernelGGL((embed_fwd_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, stream, seq, pos, word_emb,
                       pos_table, (float*)out, T, d);
  SK_CHECK_LAUNCH("skhip_embed_fwd");
  return SK_OK;
}

extern "C" int skhip_embed_bwd(const int64_t* seq, const void* dout, float* dword_emb, int64_t T, int d, int64_t n_vocab,
                               int64_t pad_id, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(dtype == SK_F32 || dtype == SK_BF16, "skhip_embed_bwd: bad dtype");
  (void)n_vocab;
  if (T == 0) return SK_OK;
  SK_CHECK_ARG(seq && dout && dword_emb && d > 0, "skhip_embed_bwd: bad argument");
  int64_t blocks = (T + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  if (dtype == SK_BF16)
    hipLaunchKernelGGL((embed_bwd_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, stream, seq,
                       (const bf16_t*)dout, dword_emb, T, d, pad_id);
  else
    hipLaunchKernelGGL((embed_bwd_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, stream, seq, (const float*)dout,
                       dword_emb, T, d, pad_id);
  SK_CHECK_LAUNCH("skhip_embed_bwd");
  return SK_OK;
}

extern "C" int skhip_pack_plan(const int64_t* seq, int B, int L, int trim, int64_t pad_id, int32_t* lens, int32_t* off,
                               int64_t* idx, int64_t* facts, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(B >= 0 && L >= 0 && trim >= 0 && (int64_t)B * L < (1ll << 31), "skhip_pack_plan: bad shape %d x %d", B, L);
  SK_CHECK_ARG(off && facts && (B == 0 || (seq && lens && idx)), "skhip_pack_plan: null pointer");
  hipLaunchKernelGGL(pack_lens_kernel, dim3(1), dim3(1024), 0, stream, seq, B, L, trim, pad_id, lens, off, facts);
  SK_CHECK_LAUNCH("skhip_pack_plan(lens)");
  if (B > 0 && L > 0) {
    const int64_t rows = (int64_t)B * L;
    const unsigned blocks = (unsigned)std::min<int64_t>((rows + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_index_kernel, dim3(blocks), dim3(256), 0, stream, off, B, L, idx);
    SK_CHECK_LAUNCH("skhip_pack_plan(index)");
  }
  return SK_OK;
}

extern "C" int skhip_pack_plans(const int64_t* desc, int n, int64_t pad_id, int64_t* facts, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(n >= 1 && n <= PACK_MAX, "skhip_pack_plans: %d batches (1..%d)", n, PACK_MAX);
  SK_CHECK_ARG(desc && facts, "skhip_pack_plans: null pointer");
  PackArgs a;
  int64_t max_rows = 0;
  for (int k = 0; k < n; ++k) {
    const int64_t* q = desc + (int64_t)PACK_FIELDS * k;
    PackDesc& d = a.d[k];
    d.seq = (const int64_t*)q[0]; d.pos = (const int64_t*)q[1];
    d.ld_seq = q[2]; d.ld_pos = q[3];
    d.B = (int)q[4]; d.L = (int)q[5]; d.trim = (int)q[6];
    d.n_vocab = q[7]; d.n_pos = q[8];
    d.lens = (int32_t*)q[9]; d.off = (int32_t*)q[10]; d.idx = (int64_t*)q[11];
    SK_CHECK_ARG(q[4] >= 1 && q[5] >= 1 && q[6] >= 0 && q[4] * q[5] < (1ll << 31), "skhip_pack_plans: batch %d: bad shape %lld x %lld",
                 k, (long long)q[4], (long long)q[5]);
    SK_CHECK_ARG(d.seq && d.lens && d.off && d.idx && d.ld_seq >= d.L && (!d.pos || (d.ld_pos >= d.L && d.n_pos >= 1)),
                 "skhip_pack_plans: batch %d: null pointer or row stride below the row length", k);
    max_rows = std::max<int64_t>(max_rows, q[4] * q[5]);
  }
  a.pad_id = pad_id;
  a.facts = facts;
  hipLaunchKernelGGL(pack_lens_multi_kernel, dim3(n), dim3(1024), 0, stream, a);
  SK_CHECK_LAUNCH("skhip_pack_plans(lens)");
  const unsigned blocks = (unsigned)std::min<int64_t>((max_rows + 255) / 256, 2048);
  hipLaunchKernelGGL(pack_index_multi_kernel, dim3(blocks, n), dim3(256), 0, stream, a);
  SK_CHECK_LAUNCH("skhip_pack_plans(index)");
  return SK_OK;
}

extern "C" int skhip_collate(const int64_t* flat, const int64_t* off, int B, int L, int64_t pad_id, int64_t* seq,
                             int64_t* pos, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(B >= 0 && L >= 0 && (int64_t)B * L < (1ll << 31), "skhip_collate: bad shape %d x %d", B, L);
  if (B == 0 || L == 0) return SK_OK;
  SK_CHECK_ARG(flat && off && seq && pos, "skhip_collate: null pointer");
  const int64_t total = (int64_t)B * L;
  const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(collate_kernel, dim3(blocks), dim3(256), 0, stream, flat, off, B, L, pad_id, seq, pos);
  SK_CHECK_LAUNCH("skhip_collate");
  return SK_OK;
}
