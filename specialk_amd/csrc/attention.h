// Shared parameter block and mask / dropout predicates of the attention kernels
// (attention.hip: exact fp32 vector-ALU path; attention_mfma.hip: bf16 matrix-core path).
#pragma once
#include "common.h"

namespace skattn {

struct AttnParams {
  const void *Q, *K, *V, *O, *dO;
  void *Oout, *dQ, *dK, *dV, *P_out;
  float* LSE;
  float* delta;
  const uint8_t* key_pad;
  const uint8_t* dense;
  const int32_t* q_off;  // packed ("varlen") batches: rows of batch b are q_off[b] .. q_off[b+1]-1
  const int32_t* k_off;  //   of Q/O/dQ resp. K/V/dK/dV; NULL = padded (B, L) rows.  Lq, Lk = maxima.
  int B, H, Lq, Lk, dk, dv;
  int64_t ldq, ldk, ldv, ldo, lddq, lddk, lddv;
  float scale;
  int causal;
  uint32_t drop_thr;
  float keep_scale;
  uint64_t seed;
  int wide_store;   // output rows stored as 16-byte pieces, 64 consecutive bytes per row and instruction (MFMA kernels)
};

template <typename T>
__device__ __forceinline__ float ld(const void* base, int64_t off) { return Cvt<T>::to_f32(((const T*)base)[off]); }
template <typename T>
__device__ __forceinline__ void st(void* base, int64_t off, float v) { ((T*)base)[off] = Cvt<T>::from_f32(v); }

// First row and length of batch b's query and key sequences.
struct Seq {
  int64_t qbase, kbase;
  int Lq, Lk;
};
__device__ __forceinline__ Seq seq_of(const AttnParams& p, int b) {
  Seq s;
  if (p.q_off) {
    s.qbase = p.q_off[b]; s.Lq = p.q_off[b + 1] - p.q_off[b];
    s.kbase = p.k_off[b]; s.Lk = p.k_off[b + 1] - p.k_off[b];
  } else {
    s.qbase = (int64_t)b * p.Lq; s.Lq = p.Lq;
    s.kbase = (int64_t)b * p.Lk; s.Lk = p.Lk;
  }
  return s;
}

__device__ __forceinline__ bool masked(const AttnParams& p, const Seq& sq, int b, int i, int j) {
  if (j >= sq.Lk || i >= sq.Lq) return true;
  if (p.key_pad && p.key_pad[(int64_t)b * p.Lk + j]) return true;
  if (p.causal && j > i) return true;
  if (p.dense && p.dense[((int64_t)b * p.Lq + i) * p.Lk + j]) return true;
  return false;
}
// Dropout on the probabilities: element (b, h, i, j) is half j & 1 of pair
// ((b*H + h)*Lq + i) * ceil(Lk/2) + j/2, so that two neighbouring keys of a row share one hash.
__device__ __forceinline__ uint64_t keep_row(const AttnParams& p, int b, int h, int i) {
  return (((uint64_t)b * p.H + h) * p.Lq + i) * (uint64_t)((p.Lk + 1) >> 1);
}
__device__ __forceinline__ bool keep(const AttnParams& p, int b, int h, int i, int j) {
  if (!p.drop_thr) return true;
  return sk_keep_half(sk_hash_pair(sk_seed(p.seed), keep_row(p, b, h, i) + (j >> 1)), j & 1, p.drop_thr);
}
// keys j4 .. j4+3 (j4 a multiple of 4) of one row: bit r = keep key j4 + r
__device__ __forceinline__ uint32_t keep4(const AttnParams& p, int b, int h, int i, int j4) {
  if (!p.drop_thr) return 15u;
  const SkSeed sd = sk_seed(p.seed);
  const uint64_t base = keep_row(p, b, h, i) + (j4 >> 1);
  const uint32_t r0 = sk_hash_pair(sd, base), r1 = sk_hash_pair(sd, base + 1);
  return (uint32_t)sk_keep_half(r0, 0, p.drop_thr) | ((uint32_t)sk_keep_half(r0, 1, p.drop_thr) << 1) |
         ((uint32_t)sk_keep_half(r1, 0, p.drop_thr) << 2) | ((uint32_t)sk_keep_half(r1, 1, p.drop_thr) << 3);
}

}  // namespace skattn

// bf16 MFMA path (attention_mfma.hip); return SK_OK after enqueueing, or a negative error.
bool sk_attn_mfma_supported(const skattn::AttnParams& p, int dtype, bool want_p);
int sk_attn_fwd_mfma(const skattn::AttnParams& p, hipStream_t stream);
int sk_attn_bwd_mfma(const skattn::AttnParams& p, hipStream_t stream);
