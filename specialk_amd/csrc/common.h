// Shared device/host helpers for the gfx950 kernels of libspecialk_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/specialk_hip.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ------------------------------------------------------------------ error handling
void sk_set_error(const char* fmt, ...);
int sk_num_cus();
int sk_sched_cus();  // CUs the GEMM schedulers plan for (sk_num_cus() minus the reserved ones)
int sk_colsum_scratch(float** ptr);  // gemm.hip: device address of the built-in column-sum scratch

#define SK_CHECK_ARG(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      sk_set_error(__VA_ARGS__);       \
      return SK_ERR_ARG;               \
    }                                  \
  } while (0)

#define SK_CHECK_LAUNCH(what)                                                      \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      sk_set_error("%s: launch failed: %s", what, hipGetErrorString(e__));         \
      return SK_ERR_LAUNCH;                                                        \
    }                                                                              \
  } while (0)

// ------------------------------------------------------------------ bf16 <-> f32
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// Round-to-nearest-even; a plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950).
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

template <typename T> struct Cvt;
template <> struct Cvt<float> {
  static __device__ __forceinline__ float to_f32(float v) { return v; }
  static __device__ __forceinline__ float from_f32(float v) { return v; }
};
template <> struct Cvt<bf16_t> {
  static __device__ __forceinline__ float to_f32(bf16_t v) { return bf16_to_f32(v); }
  static __device__ __forceinline__ bf16_t from_f32(float v) { return f32_to_bf16(v); }
};

// 16-byte vector of T: 4 floats or 8 bf16.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  float4 v;
  __device__ __forceinline__ float get(int i) const { return ((const float*)&v)[i]; }
  __device__ __forceinline__ void set(int i, float f) { ((float*)&v)[i] = f; }
};
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  uint4 v;
  __device__ __forceinline__ float get(int i) const { return bf16_to_f32(((const bf16_t*)&v)[i]); }
  __device__ __forceinline__ void set(int i, float f) { ((bf16_t*)&v)[i] = f32_to_bf16(f); }
};

// ------------------------------------------------------------------ wave reductions (wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ------------------------------------------------------------------ counter-based RNG for dropout
// Stateless: forward and backward regenerate identical masks from (seed, element index) with no
// stored mask.  One 32-bit mix (two multiply-xorshift rounds, keyed by both seed words) of a PAIR
// index yields 16 random bits for each of two neighbouring elements; 64-bit multiplies and one
// hash per element made the mask generation, not HBM, the cost of the LayerNorm kernels.
struct SkSeed { uint32_t s0, s1; };
__device__ __forceinline__ SkSeed sk_seed(uint64_t seed) {
  SkSeed k;
  k.s0 = (uint32_t)seed * 0x9E3779B1u + 0x7F4A7C15u;
  k.s1 = (uint32_t)(seed >> 32) * 0x85EBCA77u + (uint32_t)seed;
  return k;
}
__device__ __forceinline__ uint32_t sk_hash_pair(SkSeed k, uint64_t pair) {
  uint32_t x = (uint32_t)pair ^ k.s0;
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= k.s1 + (uint32_t)(pair >> 32) * 0xC2B2AE3Du;
  x ^= x >> 15; x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
// element `half` (0 / 1) of the pair is kept with probability 1 - thr16 / 65536
__device__ __forceinline__ bool sk_keep_half(uint32_t r, int half, uint32_t thr16) {
  return (half ? (r >> 16) : (r & 0xFFFFu)) >= thr16;
}
__device__ __forceinline__ bool sk_keep(SkSeed k, uint64_t idx, uint32_t thr16) {
  return sk_keep_half(sk_hash_pair(k, idx >> 1), (int)(idx & 1), thr16);
}
// drop probability p as a 16-bit threshold (0 = no dropout)
static inline uint32_t sk_drop_threshold(float p) {
  double t = (double)p * 65536.0 + 0.5;
  if (t < 1.0) return 0u;
  if (t >= 65535.0) return 65535u;
  return (uint32_t)t;
}

// Kernels that ask for more than 64 KiB of dynamic LDS must opt in once per process.
template <typename K>
static inline int sk_allow_big_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return SK_OK;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    sk_set_error("cannot raise dynamic LDS limit to %zu bytes: %s", bytes, hipGetErrorString(e));
    return SK_ERR_LAUNCH;
  }
  return SK_OK;
}
