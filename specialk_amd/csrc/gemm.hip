// MFMA GEMMs for gfx950: bf16 (v_mfma_f32_16x16x32_bf16) and exact fp32
// (v_mfma_f32_16x16x4_f32), three operand layouts, fused epilogues, deterministic split-K.
//
// Replaces nn.Linear / Conv1d(k=1) / generator projections of the reference
// (lib/transformer/SubLayers.py:46-48,62,73-74,81; lib/transformer/Models.py:252,278) and
// their autograd backward (input gradient = NN layout, weight gradient = TN layout).
//
// Kernel families (namespaces below), chosen on the host per shape:
//   b16   128x128 tile, register-staged (global -> registers -> LDS, double buffered), BK = 64:
//         the general kernel — any alignment, all layouts;
//   b16g  128x128 tile, direct-to-LDS loads (global_load_lds) into a three-stage ring, three
//         blocks per CU: K-contiguous operands (NT) with short K;
//   b16x  256x256 tile, 8 waves, four-stage direct-to-LDS ring, the two wave halves half a K-step
//         apart: long K or tile counts that fill whole rounds of the CUs; also the grouped
//         weight-gradient kernel (many small products in one launch);
//   f32   128x128 tile on the exact fp32 MFMA.
// The 128-tile kernels: 256 threads = 4 waves in a 2x2 arrangement, each wave owns a 64x64
// sub-tile = 4x4 MFMA tiles of 16x16 (64 fp32 accumulators per lane).
//
// The MFMA is issued with the operands swapped (B fragment first), so the accumulator tile
// is C^T: a lane owns ONE output row and several consecutive output columns, and the
// epilogue (bias / ReLU / sign pattern / ReLU-mask / accumulate) works on 16-byte vectors and stores whole
// 16-byte pieces of a row instead of 2-byte scalars.
//
// LDS images of the register-staged kernel (bf16, BK = 64, no padding, 16 KiB per operand tile):
//   K-contiguous operand ([rows][K] in memory): [128 rows][64 k], the eight 16-byte chunks of
//     a row XOR-swizzled with a per-row key chosen so that the ds_read_b128 fragment reads are
//     bank-conflict free (A: key = row & 7; B, whose fragment rows are permuted so that a
//     lane ends up with 16 consecutive output columns: key from bits 4-5 and 1 of the row).
//   row-contiguous operand ([K][rows] in memory, the "T" layouts of the backward GEMMs):
//     [64 k][128 rows], the eight 32-byte segments of a k-row XOR-swizzled by
//     (k & 3) | ((k >> 3) & 1) << 2, consumed with the transposing ds_read_b64_tr_b16
//     (4 k x 16 rows per 16-lane group) — no transposed copy of an activation or weight is
//     ever materialised in HBM.
//
// Loads: the FAST instantiation (16-byte aligned operands, K resp. row counts multiples of 8)
// has no per-element guards: out-of-range rows are clamped (their results are never stored)
// and out-of-range 8-element chunks are zero-filled by one predicate per chunk.  Anything
// else runs the GENERIC instantiation with element-wise guards.
#include "common.h"
#include <algorithm>
#include <vector>
#include <stdlib.h>
#include <mutex>

namespace {

struct GemmParams {
  const void* A;
  const void* B;
  void* C;
  const float* bias;
  const void* aux;
  float* ws;
  int64_t lda, ldb, ldc;
  int M, N, K;
  int k_per_split;
  int tiles_m, tiles_n;
  int epi;
  int vec_c;  // C (and aux) allow 16-byte vector access
  float alpha;
  int group;  // tile rows per rasterisation group (see tile_of)
  int tile_h; // 256-tile kernel, K-contiguous A: rows of a tile actually computed (a multiple of 16, <= 256)
  // TN products of the grouped launch only: also out[m] (+)= sum_k A[k][m], the column sums of the first operand
  // (the bias gradient that belongs to a weight gradient dy^T x); null otherwise
  float* colsum;
  float* colsum_ws;  // [splits][M] partial sums when the product is K-split
  int colsum_acc;
};

constexpr int BM = 128, BN = 128;

// Bijective XCD-aware remap: blocks that share an XCD (bid % 8) get a contiguous run of
// tiles, so the n-tiles that re-read one A row panel hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + local;
}

// Block -> output tile.  Blocks that share an XCD get a contiguous run of tile numbers
// (xcd_remap); inside a run the tiles walk down `group` tile rows before moving to the next tile
// column, so the blocks resident on one XCD at a time cover a group x (resident/group) patch of
// the output and re-use each other's A and B panels in that XCD's L2 instead of streaming a
// whole operand through it.
__device__ __forceinline__ void tile_of(const GemmParams& p, int bid, int& tile_m, int& tile_n) {
  const int pid = xcd_remap(bid, p.tiles_m * p.tiles_n);
  const int g = p.group;
  if (g <= 1) { tile_n = pid % p.tiles_n; tile_m = pid / p.tiles_n; return; }
  const int per_group = g * p.tiles_n;
  const int first_m = (pid / per_group) * g;
  const int rows = min(g, p.tiles_m - first_m);
  const int in_group = pid % per_group;
  tile_m = first_m + in_group % rows;
  tile_n = in_group / rows;
}

// ---------------------------------------------------------------------------- epilogue
template <typename TOut, typename TIn>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, float v, int row, int col) {
  v *= p.alpha;
  if (p.epi & SK_EPI_BIAS) v += p.bias[col];
  if (p.epi & SK_EPI_RELU) v = fmaxf(v, 0.f);
  int64_t off = (int64_t)row * p.ldc + col;
  if (p.epi & SK_EPI_RELU_MASK) {
    float a = Cvt<TIn>::to_f32(((const TIn*)p.aux)[off]);
    v = a > 0.f ? v : 0.f;
  }
  TOut* C = (TOut*)p.C;
  if (p.epi & SK_EPI_ACCUM) v += Cvt<TOut>::to_f32(C[off]);
  C[off] = Cvt<TOut>::from_f32(v);
}

template <typename T, int NC>
__device__ __forceinline__ void load_cols(const T* src, float (&out)[NC]) {
  constexpr int VN = Vec16<T>::N;
  if constexpr (NC >= VN) {
#pragma unroll
    for (int i = 0; i < NC / VN; ++i) {
      Vec16<T> v;
      v.v = ((const decltype(v.v)*)src)[i];
#pragma unroll
      for (int j = 0; j < VN; ++j) out[i * VN + j] = v.get(j);
    }
  } else {  // 4 bf16 = 8 bytes
    uint2 raw = *(const uint2*)src;
    const bf16_t* e = (const bf16_t*)&raw;
#pragma unroll
    for (int j = 0; j < NC; ++j) out[j] = bf16_to_f32(e[j]);
  }
}

template <typename T, int NC>
__device__ __forceinline__ void store_cols(T* dst, const float (&v)[NC]) {
  constexpr int VN = Vec16<T>::N;
  if constexpr (NC >= VN) {
#pragma unroll
    for (int i = 0; i < NC / VN; ++i) {
      Vec16<T> o;
#pragma unroll
      for (int j = 0; j < VN; ++j) o.set(j, v[i * VN + j]);
      ((decltype(o.v)*)dst)[i] = o.v;
    }
  } else {
    uint2 raw;
    bf16_t* e = (bf16_t*)&raw;
#pragma unroll
    for (int j = 0; j < NC; ++j) e[j] = f32_to_bf16(v[j]);
    *(uint2*)dst = raw;
  }
}

// NC consecutive columns of one output row, held by one lane.
template <typename TOut, typename TIn, int NC>
__device__ __forceinline__ void write_cols(const GemmParams& p, float (&v)[NC], int row, int col, int slab) {
  if (col >= p.N) return;
  const bool full = col + NC <= p.N;
  if (slab >= 0) {  // split-K: raw partial sums to the slab of this split
    float* w = p.ws + ((int64_t)slab * p.M + row) * p.N + col;
    if (full && (p.N & 3) == 0) store_cols<float, NC>(w, v);
    else
      for (int j = 0; j < NC; ++j) if (col + j < p.N) w[j] = v[j];
    return;
  }
  if (!(full && p.vec_c)) {
    for (int j = 0; j < NC; ++j) if (col + j < p.N) epilogue_store<TOut, TIn>(p, v[j], row, col + j);
    return;
  }
  const int64_t off = (int64_t)row * p.ldc + col;
#pragma unroll
  for (int j = 0; j < NC; ++j) v[j] *= p.alpha;
  if (p.epi & SK_EPI_BIAS) {
    float b[NC];
    load_cols<float, NC>(p.bias + col, b);
#pragma unroll
    for (int j = 0; j < NC; ++j) v[j] += b[j];
  }
  if (p.epi & SK_EPI_RELU) {
#pragma unroll
    for (int j = 0; j < NC; ++j) v[j] = fmaxf(v[j], 0.f);
  }
  if (p.epi & SK_EPI_RELU_MASK) {
    float a[NC];
    load_cols<TIn, NC>((const TIn*)p.aux + off, a);
#pragma unroll
    for (int j = 0; j < NC; ++j) v[j] = a[j] > 0.f ? v[j] : 0.f;
  }
  if constexpr (NC == 16) {  // sign patterns (the host admits them only on this path)
    // the four lanes l, l+16, l+32, l+48 of a wave hold the same row and 64 consecutive columns
    // (every kernel's wave tile starts at a multiple of 64 columns): one 8-byte word per row
    const int q = (col >> 4) & 3;
    uint64_t* word = (uint64_t*)p.aux + (((int64_t)row * p.N + col) >> 6);
    if (p.epi & SK_EPI_RELU_BITS_OUT) {
      uint32_t bits = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) bits |= (v[j] > 0.f ? 1u : 0u) << j;
      const uint32_t other = __shfl_xor(bits, 16, 64);
      const uint32_t pair = (q & 1) ? (other | (bits << 16)) : (bits | (other << 16));
      const uint32_t pair2 = __shfl_xor(pair, 32, 64);
      if (q == 0) *word = (uint64_t)pair | ((uint64_t)pair2 << 32);
    }
    if (p.epi & SK_EPI_RELU_MASK_BITS) {
      const uint32_t bits = (uint32_t)(*word >> (16 * q));
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = (bits >> j) & 1u ? v[j] : 0.f;
    }
  }
  if (p.epi & SK_EPI_ACCUM) {
    float c[NC];
    load_cols<TOut, NC>((const TOut*)p.C + off, c);
#pragma unroll
    for (int j = 0; j < NC; ++j) v[j] += c[j];
  }
  store_cols<TOut, NC>((TOut*)p.C + off, v);
}

// Slab index of this block's partial sums under split-K over gridDim.z, -1 without a split.
__device__ __forceinline__ int grid_slab() { return gridDim.z > 1 ? (int)blockIdx.z : -1; }

// acc[mi][ni] holds C^T tiles: lane l owns output row 16*mi + (l & 15); register r of tile ni
// is output column  PERM ? 16*(l>>4) + 4*ni + r  :  16*ni + 4*(l>>4) + r   (wave-local).
template <typename TOut, typename TIn, bool PERM, int NG = 4>
__device__ __forceinline__ void write_tile(const GemmParams& p, const f32x4_t (*acc)[4], int row0, int col0, int lane,
                                           int slab, int row_end) {
  const int g = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < NG; ++mi) {
    const int row = row0 + mi * 16 + (lane & 15);
    if (row >= row_end) continue;
    if (PERM) {
      float v[16];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[ni * 4 + r] = acc[mi][ni][r];
      // fp32 rows of a plain product, with or without statistics (the decode step's 90 MB of log-probabilities): the four lanes
      // l, l + 16, l + 32, l + 48 hold 64 consecutive columns of one row, each lane 16 of them, so a store
      // instruction of 16 bytes per lane leaves four 16-byte pieces 64 bytes apart in every row and a 128-byte line
      // is put together from eight separate writes in the L2 — the stores were 30 of the product's 70 us.  A 4 x 4
      // transpose of the 16-byte pieces across those lanes first (two exchange rounds) makes every instruction
      // write 64 consecutive bytes per row.  (Going on to whole 128-byte lines — neighbouring rows trading half of
      // their pieces, eight lanes per line — measured the same 56-57 us.)
      bool stored = false;
      if constexpr (sizeof(TOut) == 4) {
        if ((p.epi & ~SK_EPI_ROWSTATS) == 0 && p.alpha == 1.0f && slab < 0 && p.vec_c && col0 + 64 <= p.N) {
          float t[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) t[j] = v[j];
#pragma unroll
          for (int jb = 0; jb < 4; jb += 2)   // partner g ^ 1: 2 x 2 blocks of pieces
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float send = (g & 1) ? t[4 * jb + e] : t[4 * (jb + 1) + e];
              const float recv = __shfl_xor(send, 16, 64);
              if (g & 1) t[4 * jb + e] = recv; else t[4 * (jb + 1) + e] = recv;
            }
#pragma unroll
          for (int jl = 0; jl < 2; ++jl)      // partner g ^ 2: the blocks themselves
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float send = (g & 2) ? t[4 * jl + e] : t[4 * (jl + 2) + e];
              const float recv = __shfl_xor(send, 32, 64);
              if (g & 2) t[4 * jl + e] = recv; else t[4 * (jl + 2) + e] = recv;
            }
          float* dst = (float*)p.C + (int64_t)row * p.ldc + col0 + 4 * g;   // piece j: columns 16 j + 4 g ..
#pragma unroll
          for (int j = 0; j < 4; ++j) *(float4*)(dst + 16 * j) = make_float4(t[4 * j], t[4 * j + 1], t[4 * j + 2], t[4 * j + 3]);
          stored = true;
        }
      }
      if (!stored) write_cols<TOut, TIn, 16>(p, v, row, col0 + 16 * g, slab);
      if constexpr (sizeof(TOut) == 4) {
        // SK_EPI_ROWSTATS (plain fp32 products only: v is what was stored): maximum and sum of exponentials
        // against it over the 64 columns this wave holds of the row (the four lanes l, l + 16, l + 32, l + 48)
        if (p.epi & SK_EPI_ROWSTATS) {
          const int c0 = col0 + 16 * g;
          float m = -INFINITY;
#pragma unroll
          for (int j = 0; j < 16; ++j) m = fmaxf(m, c0 + j < p.N ? v[j] : -INFINITY);
          float se = 0.f;
#pragma unroll
          for (int j = 0; j < 16; ++j) se += c0 + j < p.N ? __expf(v[j] - m) : 0.f;
          if (m == -INFINITY) se = 0.f;
#pragma unroll
          for (int o = 16; o < 64; o <<= 1) {
            const float om = __shfl_xor(m, o, 64), os = __shfl_xor(se, o, 64);
            const float nm = fmaxf(m, om);
            se = (m == -INFINITY ? 0.f : se * __expf(m - nm)) + (om == -INFINITY ? 0.f : os * __expf(om - nm));
            m = nm;
          }
          if (g == 0 && col0 < p.N)
            *(float2*)((float*)p.aux + 2 * ((int64_t)row * ((p.N + 63) >> 6) + (col0 >> 6))) = make_float2(m, se);
        }
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
        write_cols<TOut, TIn, 4>(p, v, row, col0 + 16 * ni + 4 * g, slab);
      }
    }
  }
}

// Epilogue of a FULL interior wave tile (NMI 16-row groups x 64 columns) of the 256-tile kernels in the permuted
// layout (a lane owns 16 consecutive columns of one row in each of the groups; 16-byte aligned C, bias and pattern).
// write_tile above handles one row group at a time: load what the epilogue reads, wait, store, eight times in
// a row, each time exposing a full memory round trip.  Here the bias is fetched once per tile and the old C
// values (accumulate form) and sign-pattern words of a PAIR of row groups are requested together, so a tile
// exposes one round trip (plain form) or four instead of eight or more.
// Supported: alpha, bias, ReLU, sign pattern out / in, accumulate (bf16 output only).  The caller checks
// full_epilogue_ok() and otherwise takes write_tile.
//
// Each batch of loads is one inline-assembly statement that ends in its own s_waitcnt vmcnt(0): the loads of a
// batch are in flight together whatever the compiler would have made of the surrounding run-time branches (it
// orders a load, its wait and the store of the previous row group strictly one after the other otherwise).
__device__ __forceinline__ bool full_epilogue_ok(const GemmParams& p, bool out_is_bf16) {
  return p.vec_c && !(p.epi & SK_EPI_RELU_MASK) && (out_is_bf16 || !(p.epi & SK_EPI_ACCUM));
}

__device__ float g_zero_bias[16];  // stands in for the bias of a lane's 16 columns when there is none

__device__ __forceinline__ void load4x16_wait(const void* ptr, uint4& a, uint4& b, uint4& c, uint4& d) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off\n\t"
      "global_load_dwordx4 %1, %4, off offset:16\n\t"
      "global_load_dwordx4 %2, %4, off offset:32\n\t"
      "global_load_dwordx4 %3, %4, off offset:48\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
      : "v"(ptr)
      : "memory");
}
// two rows x 32 bytes
__device__ __forceinline__ void load2x32_wait(const void* p0, const void* p1, uint4& a0, uint4& a1, uint4& b0, uint4& b1) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off\n\t"
      "global_load_dwordx4 %1, %4, off offset:16\n\t"
      "global_load_dwordx4 %2, %5, off\n\t"
      "global_load_dwordx4 %3, %5, off offset:16\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1)
      : "v"(p0), "v"(p1)
      : "memory");
}
__device__ __forceinline__ void load2x8_wait(const void* p0, const void* p1, uint2& a, uint2& b) {
  asm volatile(
      "global_load_dwordx2 %0, %2, off\n\t"
      "global_load_dwordx2 %1, %3, off\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a), "=&v"(b)
      : "v"(p0), "v"(p1)
      : "memory");
}
// two rows x (32 bytes + an 8-byte word), one wait
__device__ __forceinline__ void load2x32_2x8_wait(const void* p0, const void* p1, const void* w0, const void* w1, uint4& a0,
                                                  uint4& a1, uint4& b0, uint4& b1, uint2& x, uint2& y) {
  asm volatile(
      "global_load_dwordx4 %0, %6, off\n\t"
      "global_load_dwordx4 %1, %6, off offset:16\n\t"
      "global_load_dwordx4 %2, %7, off\n\t"
      "global_load_dwordx4 %3, %7, off offset:16\n\t"
      "global_load_dwordx2 %4, %8, off\n\t"
      "global_load_dwordx2 %5, %9, off\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1), "=&v"(x), "=&v"(y)
      : "v"(p0), "v"(p1), "v"(w0), "v"(w1)
      : "memory");
}

template <typename TOut, int NMI>
__device__ __forceinline__ void write_wave_tile_full(const GemmParams& p, const f32x4_t (*acc)[4], int row0, int col0,
                                                     int lane, char* scratch) {
  const int g = lane >> 4, col = col0 + 16 * g;
  const int epi = p.epi;
  const float alpha = p.alpha;
  const bool accum = sizeof(TOut) == 2 && (epi & SK_EPI_ACCUM), maskb = epi & SK_EPI_RELU_MASK_BITS;
  float bias[16];
  {
    uint4 b[4];
    load4x16_wait((epi & SK_EPI_BIAS) ? (const void*)(p.bias + col) : (const void*)g_zero_bias, b[0], b[1], b[2], b[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bias[4 * i] = __uint_as_float(b[i].x); bias[4 * i + 1] = __uint_as_float(b[i].y);
      bias[4 * i + 2] = __uint_as_float(b[i].z); bias[4 * i + 3] = __uint_as_float(b[i].w);
    }
  }
  const int q = (col >> 4) & 3;  // which 16-column quarter of the 64-column pattern word this lane holds
#pragma unroll
  for (int pair = 0; pair < 4; ++pair) {
    if (2 * pair >= NMI) break;  // NMI: the wave's 16-row groups
    const bool second = 2 * pair + 1 < NMI;
    uint4 oldc[2][2];
    uint2 words[2];
    const int64_t r0 = row0 + (2 * pair) * 16 + (lane & 15), r1 = second ? r0 + 16 : r0;
    const void* c0 = (const bf16_t*)p.C + r0 * p.ldc + col;
    const void* c1 = (const bf16_t*)p.C + r1 * p.ldc + col;
    const void* w0 = (const uint64_t*)p.aux + ((r0 * p.N + col) >> 6);
    const void* w1 = (const uint64_t*)p.aux + ((r1 * p.N + col) >> 6);
    if (accum && maskb) load2x32_2x8_wait(c0, c1, w0, w1, oldc[0][0], oldc[0][1], oldc[1][0], oldc[1][1], words[0], words[1]);
    else if (accum) load2x32_wait(c0, c1, oldc[0][0], oldc[0][1], oldc[1][0], oldc[1][1]);
    else if (maskb) load2x8_wait(w0, w1, words[0], words[1]);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int mi = (m == 1 && !second) ? 0 : 2 * pair + m;  // (index kept in range; the iteration is skipped)
      if (m == 1 && !second) continue;
      const int64_t row = m ? r1 : r0;
      float v[16];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[ni * 4 + r] = acc[mi][ni][r] * alpha + bias[ni * 4 + r];
      if (epi & SK_EPI_RELU) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (epi & SK_EPI_RELU_BITS_OUT) {
        uint32_t bits = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) bits |= (v[j] > 0.f ? 1u : 0u) << j;
        const uint32_t other = __shfl_xor(bits, 16, 64);
        const uint32_t pr = (q & 1) ? (other | (bits << 16)) : (bits | (other << 16));
        const uint32_t pr2 = __shfl_xor(pr, 32, 64);
        if (q == 0) *((uint64_t*)p.aux + ((row * p.N + col) >> 6)) = (uint64_t)pr | ((uint64_t)pr2 << 32);
      }
      if (maskb) {
        const uint64_t word = (uint64_t)words[m].x | ((uint64_t)words[m].y << 32);
        const uint32_t bits = (uint32_t)(word >> (16 * q));
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = (bits >> j) & 1u ? v[j] : 0.f;
      }
      if (accum) {
        const bf16_t* e0 = (const bf16_t*)&oldc[m][0];
        const bf16_t* e1 = (const bf16_t*)&oldc[m][1];
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] += bf16_to_f32(e0[j]); v[8 + j] += bf16_to_f32(e1[j]); }
      }
      if constexpr (sizeof(TOut) == 2) {
        // Through the wave's 2 KiB of LDS scratch so that eight consecutive lanes store one whole 128-byte line:
        // stored straight from the accumulator layout, consecutive lanes are different ROWS, every 16-byte piece
        // is a memory request of its own (64 per instruction instead of 8), and the loads of the next tile queue
        // behind them (measured on 25.6 k x 2048 x 512: 7 -> 5 thousand cycles per tile epilogue, 84 -> 75 us per launch).
        uint4 h0, h1;
        {
          bf16_t* e0 = (bf16_t*)&h0;
          bf16_t* e1 = (bf16_t*)&h1;
#pragma unroll
          for (int j = 0; j < 8; ++j) { e0[j] = f32_to_bf16(v[j]); e1[j] = f32_to_bf16(v[8 + j]); }
        }
        const int r15 = lane & 15;
        *(uint4*)(scratch + r15 * 128 + (((2 * g) ^ (r15 & 7)) << 4)) = h0;
        *(uint4*)(scratch + r15 * 128 + (((2 * g + 1) ^ (r15 & 7)) << 4)) = h1;
        const int c = lane & 7;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int rho = (lane >> 3) + 8 * j;
          const uint4 o = *(const uint4*)(scratch + rho * 128 + ((c ^ (rho & 7)) << 4));
          bf16_t* dst = (bf16_t*)p.C + (int64_t)(row0 + mi * 16 + rho) * p.ldc + col0 + 8 * c;
          *(uint4*)dst = o;
        }
      } else {
        store_cols<TOut, 16>((TOut*)p.C + row * p.ldc + col, v);
      }
    }
  }
}

// ============================================================================ bf16
namespace b16 {
constexpr int BK = 64;
constexpr int OPER_BYTES = 128 * BK * 2;   // 16384
constexpr int SMEM_BYTES = 4 * OPER_BYTES; // 2 operands x 2 buffers = 64 KiB

__device__ __forceinline__ int key_a(int row) { return row & 7; }
__device__ __forceinline__ int key_b(int row) { return (((row >> 4) & 3) << 1) | ((row >> 1) & 1); }
__device__ __forceinline__ int key_t(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
// byte offset of element (k, col) in a [64 k][128 rows] image
__device__ __forceinline__ int t_off(int k, int col) { return k * 256 + ((((col >> 4) ^ key_t(k)) & 7) << 5) + (col & 15) * 2; }

// ---- global -> registers
template <bool T_LAYOUT, bool FAST>
__device__ __forceinline__ void load_tile(uint4 (&reg)[4], const bf16_t* base, int64_t ld, int r0, int R, int k0,
                                          int kend, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 256 * i;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (T_LAYOUT) {
      const int kk = k0 + (id >> 4), rr = r0 + (id & 15) * 8;
      if (FAST) {
        if (kk < kend && rr + 8 <= R) v = *(const uint4*)(base + (int64_t)kk * ld + rr);
      } else if (kk < kend) {
        const bf16_t* src = base + (int64_t)kk * ld + rr;
        bf16_t* e = (bf16_t*)&v;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (rr + j < R) e[j] = src[j];
      }
    } else {
      const int kk = k0 + (id & 7) * 8;
      if (FAST) {
        const int rr = min(r0 + (id >> 3), R - 1);  // clamped rows feed outputs that are never stored
        if (kk + 8 <= kend) v = *(const uint4*)(base + (int64_t)rr * ld + kk);
      } else {
        const int rr = r0 + (id >> 3);
        if (rr < R) {
          const bf16_t* src = base + (int64_t)rr * ld + kk;
          bf16_t* e = (bf16_t*)&v;
#pragma unroll
          for (int j = 0; j < 8; ++j) if (kk + j < kend) e[j] = src[j];
        }
      }
    }
    reg[i] = v;
  }
}

// ---- registers -> LDS
template <bool T_LAYOUT, bool IS_B>
__device__ __forceinline__ void store_tile(const uint4 (&reg)[4], char* lds, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 256 * i;
    if (T_LAYOUT) {
      const int k = id >> 4, c = id & 15;
      *(uint4*)(lds + k * 256 + ((((c >> 1) ^ key_t(k)) & 7) << 5) + (c & 1) * 16) = reg[i];
    } else {
      const int row = id >> 3, c = id & 7;
      const int key = IS_B ? key_b(row) : key_a(row);
      *(uint4*)(lds + row * 128 + ((c ^ key) << 4)) = reg[i];
    }
  }
}

__device__ __forceinline__ bf16x8_t tr_pair(const char* lds, int k, int col) {
  typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds + t_off(k, col)));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds + t_off(k + 4, col)));
  s16x8_t f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, f);
}

// A fragment (second MFMA operand): lane l holds A[row0 + (l&15)][k = 32*kk + 8*(l>>4) + j]
template <bool T_LAYOUT>
__device__ __forceinline__ bf16x8_t read_frag_a(const char* lds, int row0, int kk, int lane) {
  if (T_LAYOUT) {
    const int i = lane & 15, g = lane >> 4;
    return tr_pair(lds, kk * 32 + g * 8 + (i >> 2), row0 + (i & 3) * 4);
  } else {
    const int row = row0 + (lane & 15), chunk = kk * 4 + (lane >> 4);
    return *(const bf16x8_t*)(lds + row * 128 + ((chunk ^ key_a(row)) << 4));
  }
}

// B fragment (first MFMA operand) of tile ni inside the wave's 64 columns starting at col0.
// K-contiguous B: lane rho = l&15 reads row 16*(rho>>2) + 4*ni + (rho&3)  (permuted, see write_tile).
// T-layout B:     standard rows 16*ni + rho.
template <bool T_LAYOUT>
__device__ __forceinline__ bf16x8_t read_frag_b(const char* lds, int col0, int ni, int kk, int lane) {
  if (T_LAYOUT) {
    const int i = lane & 15, g = lane >> 4;
    return tr_pair(lds, kk * 32 + g * 8 + (i >> 2), col0 + 16 * ni + (i & 3) * 4);
  } else {
    const int rho = lane & 15;
    const int row = col0 + 16 * (rho >> 2) + 4 * ni + (rho & 3), chunk = kk * 4 + (lane >> 4);
    return *(const bf16x8_t*)(lds + row * 128 + ((chunk ^ key_b(row)) << 4));
  }
}

template <bool A_T, bool B_T, typename TOut, bool FAST>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int tile_m, tile_n;
  tile_of(p, blockIdx.x, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int wrow = (wave >> 1) * 64, wcol = (wave & 1) * 64;

  const int kbeg = blockIdx.z * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;

  const bf16_t* A = (const bf16_t*)p.A;
  const bf16_t* B = (const bf16_t*)p.B;
  auto ldsA = [&](int buf) { return smem + buf * OPER_BYTES; };
  auto ldsB = [&](int buf) { return smem + (2 + buf) * OPER_BYTES; };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  if (nk > 0) {
    load_tile<A_T, FAST>(ra, A, p.lda, m0, p.M, kbeg, kend, tid);
    load_tile<B_T, FAST>(rb, B, p.ldb, n0, p.N, kbeg, kend, tid);
    store_tile<A_T, false>(ra, ldsA(0), tid);
    store_tile<B_T, true>(rb, ldsB(0), tid);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) {
      load_tile<A_T, FAST>(ra, A, p.lda, m0, p.M, kbeg + (kt + 1) * BK, kend, tid);
      load_tile<B_T, FAST>(rb, B, p.ldb, n0, p.N, kbeg + (kt + 1) * BK, kend, tid);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = read_frag_a<A_T>(ldsA(cur), wrow + i * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) fb[i] = read_frag_b<B_T>(ldsB(cur), wcol, i, kk, lane);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ni], fa[mi], acc[mi][ni], 0, 0, 0);
    }
    if (more) {
      store_tile<A_T, false>(ra, ldsA(cur ^ 1), tid);
      store_tile<B_T, true>(rb, ldsB(cur ^ 1), tid);
    }
    __syncthreads();
  }
  write_tile<TOut, bf16_t, !B_T>(p, acc, m0 + wrow, n0 + wcol, lane, grid_slab(), p.M);
}
}  // namespace b16

// ============================================================================ bf16, direct-to-LDS
// The fast path: operand tiles go global -> LDS with global_load_lds_dwordx4 (no staging
// registers, no ds_write pass), three LDS stages of BK = 32, loads of tile kt+2 are issued
// while tile kt is multiplied, and a COUNTED s_waitcnt vmcnt(4) + one raw s_barrier per
// K-step lets the newer tile's loads stay in flight across the barrier.  48 KiB of LDS and
// ~120 VGPRs per block give three resident blocks per CU.
//
// A direct-to-LDS load writes lane l's 16 bytes at (wave-uniform LDS base) + 16*l, so the LDS
// image is lane-linear and every swizzle is applied to the per-lane SOURCE address:
//   K-contiguous operand: [128 rows][32 k], 64-byte rows; chunk c of row r is stored at chunk
//     position c ^ key(r); key = (-(r>>2)) & 3 for A and (-(r>>4)) & 3 for B (whose fragment
//     rows are permuted, see write_tile) make every ds_read_b128 fragment read conflict-free.
//   row-contiguous operand: [32 k][128 rows], 256-byte rows, 32-byte segments XOR key_t(k).
// Out-of-range chunks are sourced from a zero page; out-of-range rows of a K-contiguous
// operand are clamped (their products land in output rows/columns that are never stored).
namespace b16g {
constexpr int BK = 32, STAGES = 3;
constexpr int OPER_BYTES = 128 * BK * 2;        // 8192
constexpr int STAGE_BYTES = 2 * OPER_BYTES;     // A + B
constexpr int SMEM_BYTES = STAGES * STAGE_BYTES; // 48 KiB

__device__ uint4 g_zero_page[2];

__device__ __forceinline__ int key_ka(int row) { return (-(row >> 2)) & 3; }
__device__ __forceinline__ int key_kb(int row) { return (-(row >> 4)) & 3; }
using b16::key_t;
using b16::t_off;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// Source pointers of this lane's two direct-to-LDS loads per operand tile, for tile 0.
// K-contiguous: instruction idx covers rows 16*idx..+15; T layout: k rows 4*idx..+3.
template <bool T_LAYOUT, bool IS_B>
struct Loader {
  const char* ptr[2];
  int64_t step;       // bytes per K-step
  int kofs[2];        // first k of this lane's chunk relative to the tile start
  bool rows_ok[2];    // T layout: the 8 rows of the chunk exist
  __device__ __forceinline__ void init(const bf16_t* base, int64_t ld, int r0, int R, int kbeg, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = wave * 2 + j;
      if (T_LAYOUT) {
        const int k = 4 * idx + (lane >> 4), pos = lane & 15;
        const int col = (((pos >> 1) ^ key_t(k)) & 7) * 16 + (pos & 1) * 8;
        kofs[j] = k;
        rows_ok[j] = r0 + col + 8 <= R;
        ptr[j] = (const char*)(base + (int64_t)(kbeg + k) * ld + r0 + col);
        step = (int64_t)BK * ld * 2;
      } else {
        const int row = 16 * idx + (lane >> 2), cpos = lane & 3;
        const int c = cpos ^ (IS_B ? key_kb(row) : key_ka(row));
        kofs[j] = 8 * c;
        rows_ok[j] = true;
        ptr[j] = (const char*)(base + (int64_t)min(r0 + row, R - 1) * ld + kbeg + 8 * c);
        step = BK * 2;
      }
    }
  }
  // issue the two loads of tile kt into the operand image at `lds` (wave-uniform)
  __device__ __forceinline__ void issue(char* lds, int wave_u, int k0, int kend) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = T_LAYOUT ? (rows_ok[j] && k0 + kofs[j] < kend) : (k0 + kofs[j] + 8 <= kend);
      const char* src = ok ? ptr[j] : (const char*)g_zero_page;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(lds + (wave_u * 2 + j) * 1024), 16, 0, 0);
      ptr[j] += step;
    }
  }
};

template <bool T_LAYOUT>
__device__ __forceinline__ bf16x8_t read_frag_a(const char* lds, int row0, int lane) {
  if (T_LAYOUT) {
    const int i = lane & 15, g = lane >> 4;
    return b16::tr_pair(lds, g * 8 + (i >> 2), row0 + (i & 3) * 4);
  } else {
    const int row = row0 + (lane & 15), g = lane >> 4;
    return *(const bf16x8_t*)(lds + row * 64 + ((g ^ key_ka(row)) << 4));
  }
}

template <bool T_LAYOUT>
__device__ __forceinline__ bf16x8_t read_frag_b(const char* lds, int col0, int ni, int lane) {
  if (T_LAYOUT) {
    const int i = lane & 15, g = lane >> 4;
    return b16::tr_pair(lds, g * 8 + (i >> 2), col0 + 16 * ni + (i & 3) * 4);
  } else {
    const int rho = lane & 15, g = lane >> 4;
    const int row = col0 + 16 * (rho >> 2) + 4 * ni + (rho & 3);
    return *(const bf16x8_t*)(lds + row * 64 + ((g ^ key_kb(row)) << 4));
  }
}

template <bool A_T, bool B_T, typename TOut>
__global__ __launch_bounds__(256, 3) void gemm_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile_m, tile_n;
  tile_of(p, blockIdx.x, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int wrow = (wave >> 1) * 64, wcol = (wave & 1) * 64;

  const int kbeg = blockIdx.z * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;

  Loader<A_T, false> la;
  Loader<B_T, true> lb;
  la.init((const bf16_t*)p.A, p.lda, m0, p.M, kbeg, wave, lane);
  lb.init((const bf16_t*)p.B, p.ldb, n0, p.N, kbeg, wave, lane);

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int stage, int kt) {
    char* base = smem + stage * STAGE_BYTES;
    la.issue(base, wave, kbeg + kt * BK, kend);
    lb.issue(base + OPER_BYTES, wave, kbeg + kt * BK, kend);
  };
  if (nk > 0) issue(0, 0);
  if (nk > 1) issue(1, 1);

  int cs = 0, is = 2;  // stage being computed / stage being filled
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt's four loads are older than tile kt+1's four: wait for all but the newest four
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 2 < nk) issue(is, kt + 2);
    const char* sa = smem + cs * STAGE_BYTES;
    const char* sb = sa + OPER_BYTES;
    bf16x8_t fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = read_frag_a<A_T>(sa, wrow + i * 16, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) fb[i] = read_frag_b<B_T>(sb, wcol, i, lane);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ni], fa[mi], acc[mi][ni], 0, 0, 0);
    cs = cs == STAGES - 1 ? 0 : cs + 1;
    is = is == STAGES - 1 ? 0 : is + 1;
  }
  write_tile<TOut, bf16_t, !B_T>(p, acc, m0 + wrow, n0 + wcol, lane, grid_slab(), p.M);
}
}  // namespace b16g

// ============================================================================ bf16, 256x256 tile
// Large-tile variant of the direct-to-LDS kernel (all three operand layouts).  A 128x128
// tile needs 64 B of operand fill per clock per CU at full MFMA rate, which is all the
// global -> LDS path delivers; a 256x256 tile halves that.  It pays when there are many more
// tiles than CUs (the generator projection); with about one tile per CU the 128x128 kernels'
// three resident blocks per CU hide the prologue and epilogue better.
//
// 512 threads = 8 waves as 2 (M) x 4 (N), each wave owns 128x64 outputs = 8x4 MFMA tiles
// (128 fp32 accumulators per lane).  BK = 32, four LDS stages of 32 KiB (A 16 KiB + B 16 KiB),
// one block per CU, two waves per SIMD.  The waves of the two M halves run half a K-step apart
// (the upper half executes one extra barrier before the loop, the lower half one after it):
// while one half multiplies tile t, the other reads its fragments of the next tile from LDS and
// issues the loads of tile t+3, so each SIMD's MFMA pipe alternates between its two waves.
// (A BK = 64, two-stage form of the same schedule measured 3-15 % slower.)
//
// Per K-step and wave:   R: barrier; 12 x ds_read_b128 (tile t); 4 x global_load_lds (tile t+3)
//                        M: vmcnt (tile t+1 landed); lgkmcnt(0); barrier; 32 MFMAs
// Ordering argument (barrier instances are counted per workgroup; the lower half passes
// instance 2t entering R(t) and 2t+1 entering M(t), the upper half 2t+1 and 2t+2):
//   read-after-load:  every wave has waited for its own loads of tile t+1 before the barrier
//     that opens its M(t), i.e. before instance 2t+2 at the latest, and tile t+1 is first read
//     after instance 2t+2;
//   load-after-read:  tile t+3 overwrites the stage of tile t-1.  It is issued after instance 2t
//     at the earliest; every wave's reads of tile t-1 were completed (lgkmcnt(0)) before the
//     barrier that opens its M(t-1), i.e. before instance 2t at the latest.
namespace b16x {
constexpr int XM = 256, XN = 256, BK = 32, STAGES = 4;
constexpr int OPER_BYTES = 256 * BK * 2;         // 16384
constexpr int STAGE_BYTES = 2 * OPER_BYTES;      // A + B
constexpr int SMEM_BYTES = STAGES * STAGE_BYTES;  // 128 KiB
constexpr int EPI_SCRATCH_BYTES = 8 * 2048;       // per wave 16 rows x 128 bytes behind the ring: see write_wave_tile_full
constexpr int LDS_BYTES = SMEM_BYTES + EPI_SCRATCH_BYTES;

using b16::key_t;
using b16g::g_zero_page;
using b16g::gbl_ptr_t;
using b16g::key_ka;
using b16g::key_kb;
using b16g::lds_ptr_t;

// Row-contiguous ("T") operand image of a 256-wide tile: [32 k][256 rows], 512-byte k-rows whose
// sixteen 32-byte segments are XOR-swizzled (low three bits) with key_t(k), so that the transposing
// ds_read_b64_tr_b16 fragment reads (eight k-rows x 32 bytes per 32-lane half) spread over all banks.
__device__ __forceinline__ int t_off256(int k, int col) {
  return k * 512 + ((((col >> 4) ^ key_t(k)) & 15) << 5) + (col & 15) * 2;
}

// This lane's two direct-to-LDS loads per operand tile.  K-contiguous operand: instruction idx covers
// rows 16*idx..+15 (64-byte rows, chunk swizzle on the source address, as in the 128-tile kernel);
// T operand: instruction idx covers k-rows 2*idx, 2*idx+1 (32 sixteen-byte pieces each).
template <bool T_LAYOUT, bool IS_B>
struct Loader {
  const char* ptr[2];
  int64_t step;
  int kofs[2];
  bool rows_ok[2];
  // rows_used: rows of the 256-row image that the tile reads (the others are filled with zeros)
  __device__ __forceinline__ void init(const bf16_t* base, int64_t ld, int r0, int R, int kbeg, int wave, int lane,
                                       int rows_used) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = wave * 2 + j;
      if (T_LAYOUT) {
        const int k = 2 * idx + (lane >> 5), pos = lane & 31;
        const int col = (((pos >> 1) ^ key_t(k)) & 15) * 16 + (pos & 1) * 8;
        kofs[j] = k;
        rows_ok[j] = r0 + col + 8 <= R;
        ptr[j] = (const char*)(base + (int64_t)(kbeg + k) * ld + r0 + col);
        step = (int64_t)BK * ld * 2;
      } else {
        const int row = 16 * idx + (lane >> 2), cpos = lane & 3;
        const int c = cpos ^ (IS_B ? key_kb(row) : key_ka(row));
        kofs[j] = 8 * c;
        rows_ok[j] = 16 * idx < rows_used;
        ptr[j] = (const char*)(base + (int64_t)min(r0 + row, R - 1) * ld + kbeg + 8 * c);
        step = BK * 2;
      }
    }
  }
  __device__ __forceinline__ void issue(char* lds, int wave_u, int k0, int kend) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = rows_ok[j] && (T_LAYOUT ? k0 + kofs[j] < kend : k0 + kofs[j] + 8 <= kend);
      const char* src = ok ? ptr[j] : (const char*)g_zero_page;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(lds + (wave_u * 2 + j) * 1024), 16, 0, 0);
      ptr[j] += step;
    }
  }
};

// One MFMA operand fragment (8 bf16 per lane) from an operand image at per-lane byte offset `off`.
template <bool T_LAYOUT>
__device__ __forceinline__ bf16x8_t read_frag(const char* image, int off) {
  if (T_LAYOUT) {
    typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(image + off));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(image + off + 4 * 512));  // k + 4: same key
    const s16x8_t f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, f);
  } else {
    return *(const bf16x8_t*)(image + off);
  }
}

// One 256 x 256 output tile over K range [kbeg, kend): the whole block's work.  slab >= 0 sends the
// raw partial sums to that split-K slab instead of the epilogue.
// With a K-contiguous A the tile may be lower than 256 rows (p.tile_h, a multiple of 16, at least 144): one
// block per CU means a launch takes whole rounds of the chip, and 25.6 k rows are 100.2 tiles of 256 — with
// N = 512 that is 202 blocks on 256 CUs, with N = 2048 three rounds and an eighth.  Tiles of 208 rows make it 248
// blocks, or four rounds that are full, of proportionally shorter tiles.  The rows above tile_h are loaded as
// zeros and the upper wave half skips their MFMAs and stores.
// NMI: the 16-row groups this wave computes, wrow: its first row inside the tile, th: the tile's rows.
template <bool A_T, bool B_T, typename TOut, int NMI>
__device__ __forceinline__ void gemm_tile_part(const GemmParams& p, int tile_m, int tile_n, int kbeg, int kend, int slab,
                                               int wave, int wrow, int th) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int m0 = tile_m * th, n0 = tile_n * XN;
  const int wcol = (wave & 3) * 64;
  const int nk = (kend - kbeg + BK - 1) / BK;

  Loader<A_T, false> la;
  Loader<B_T, true> lb;
  la.init((const bf16_t*)p.A, p.lda, m0, p.M, kbeg, wave, lane, th);
  lb.init((const bf16_t*)p.B, p.ldb, n0, p.N, kbeg, wave, lane, XN);

  f32x4_t acc[NMI][4];
#pragma unroll
  for (int i = 0; i < NMI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int stage, int kt) {
    char* base = smem + stage * STAGE_BYTES;
    la.issue(base, wave, kbeg + kt * BK, kend);
    lb.issue(base + OPER_BYTES, wave, kbeg + kt * BK, kend);
  };
  if (nk > 0) issue(0, 0);
  if (nk > 1) issue(1, 1);
  if (nk > 2) issue(2, 2);
  // tile 0 must have landed before the first R phase
  if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (wave >= 4) __builtin_amdgcn_s_barrier();  // upper half runs half a K-step behind

  // per-lane fragment offsets inside a stage
  int offa[NMI], offb[4];
  {
    const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < NMI; ++i) {
      if (A_T) {
        offa[i] = t_off256(8 * g + (l15 >> 2), wrow + 16 * i + 4 * (l15 & 3));
      } else {
        const int row = wrow + i * 16 + l15;
        offa[i] = row * 64 + ((g ^ key_ka(row)) << 4);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (B_T) {  // standard column order (see write_tile, PERM = false)
        offb[i] = OPER_BYTES + t_off256(8 * g + (l15 >> 2), wcol + 16 * i + 4 * (l15 & 3));
      } else {    // permuted rows: a lane ends up with 16 consecutive output columns
        const int row = wcol + 16 * (l15 >> 2) + 4 * i + (l15 & 3);
        offb[i] = OPER_BYTES + row * 64 + ((g ^ key_kb(row)) << 4);
      }
    }
  }

  // Column sums of the A operand (TN products: A is [k][m], so these are sums over the contraction index) ride on
  // the matrix pipe: a product of an all-ones fragment with the A fragment is a 16 x 16 tile whose every row holds
  // the sums of the fragment's 16 columns.  Only the blocks of tile column 0 do it, and each of the four waves
  // that share the block's rows takes two of the eight row groups: two more MFMAs per wave and K-step (+6 %).
  bool do_colsum = false;
  if constexpr (A_T && B_T) do_colsum = p.colsum != nullptr && tile_n == 0;
  f32x4_t csum[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
  const s16x8_t ones_bits = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};  // bf16 1.0
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_bits);
  const int cs_mi = 2 * (wave & 3);

  int cs = 0, is = 3;  // stage being computed / stage being filled
  for (int kt = 0; kt < nk; ++kt) {
    // ---- R: fragments of tile kt, loads of tile kt+3
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const char* st = smem + cs * STAGE_BYTES;
    bf16x8_t fb[4], fa[NMI];
#pragma unroll
    for (int i = 0; i < 4; ++i) fb[i] = read_frag<B_T>(st, offb[i]);
#pragma unroll
    for (int i = 0; i < NMI; ++i) fa[i] = read_frag<A_T>(st, offa[i]);
    if (kt + 3 < nk) issue(is, kt + 3);
    // ---- M: tile kt+1 landed (this wave's part), own LDS reads done, then multiply
    if (kt + 3 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ni], fa[mi], acc[mi][ni], 0, 0, 0);
    if constexpr (A_T && B_T && NMI == 8) {
      if (do_colsum) {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)  // static register indices: the wave's pair is picked by a uniform compare
          if (mi >> 1 == (wave & 3))
            csum[mi & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[mi], csum[mi & 1], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    cs = cs == STAGES - 1 ? 0 : cs + 1;
    is = is == STAGES - 1 ? 0 : is + 1;
  }
  if (wave < 4) __builtin_amdgcn_s_barrier();  // balances the upper half's extra barrier

  if constexpr (A_T && B_T && NMI == 8) {
    if (do_colsum && lane < 16) {  // lanes 0..15, register 0: the sums of columns 16 * mi + lane
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = m0 + wrow + 16 * (cs_mi + j) + lane;
        if (m < p.M) {
          if (slab >= 0) p.colsum_ws[(int64_t)slab * p.M + m] = csum[j][0];
          else p.colsum[m] = p.colsum_acc ? p.colsum[m] + csum[j][0] : csum[j][0];
        }
      }
    }
  }

  if constexpr (!B_T) {
    if (slab < 0 && m0 + th <= p.M && n0 + XN <= p.N && full_epilogue_ok(p, sizeof(TOut) == 2)) {
      write_wave_tile_full<TOut, NMI>(p, acc, m0 + wrow, n0 + wcol, lane, smem + SMEM_BYTES + wave * 2048);
      return;
    }
  }
  const int row_end = min(p.M, m0 + th);
  write_tile<TOut, bf16_t, !B_T, (NMI < 4 ? NMI : 4)>(p, acc, m0 + wrow, n0 + wcol, lane, slab, row_end);
  if constexpr (NMI > 4)
    write_tile<TOut, bf16_t, !B_T, NMI - 4>(p, acc + 4, m0 + wrow + 64, n0 + wcol, lane, slab, row_end);
}

// The block's tile: the lower wave half takes the first ceil(n / 2) of the tile's n = tile_h / 16 row groups, the
// upper half the rest; each count is a compile-time constant of its own copy of the loop, so the accumulators
// stay in registers under static indices.  Both copies execute the same sequence of barriers.
template <bool A_T, bool B_T, typename TOut>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, int tile_m, int tile_n, int kbeg, int kend, int slab) {
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  if constexpr (A_T) {
    gemm_tile_part<A_T, B_T, TOut, 8>(p, tile_m, tile_n, kbeg, kend, slab, wave, (wave >> 2) * 128, XM);
  } else {
    const int th = p.tile_h, n = th >> 4, nlo = (n + 1) >> 1;
    const int mine = wave >= 4 ? n - nlo : nlo, wrow = wave >= 4 ? 16 * nlo : 0;
    switch (mine) {
      case 8: gemm_tile_part<A_T, B_T, TOut, 8>(p, tile_m, tile_n, kbeg, kend, slab, wave, wrow, th); break;
      case 7: gemm_tile_part<A_T, B_T, TOut, 7>(p, tile_m, tile_n, kbeg, kend, slab, wave, wrow, th); break;
      case 6: gemm_tile_part<A_T, B_T, TOut, 6>(p, tile_m, tile_n, kbeg, kend, slab, wave, wrow, th); break;
      case 5: gemm_tile_part<A_T, B_T, TOut, 5>(p, tile_m, tile_n, kbeg, kend, slab, wave, wrow, th); break;
      default: gemm_tile_part<A_T, B_T, TOut, 4>(p, tile_m, tile_n, kbeg, kend, slab, wave, wrow, th); break;
    }
  }
}

template <bool A_T, bool B_T, typename TOut>
__global__ __launch_bounds__(512, 1) void gemm_kernel(GemmParams p) {
  int tile_m, tile_n;
  tile_of(p, blockIdx.x, tile_m, tile_n);
  const int kbeg = blockIdx.z * p.k_per_split;
  gemm_tile<A_T, B_T, TOut>(p, tile_m, tile_n, kbeg, min(p.K, kbeg + p.k_per_split), grid_slab());
}

// ---- grouped weight gradients: many independent C_i (M_i x N_i, fp32) (+)= A_i^T B_i products with
// row-contiguous bf16 operands (A_i is K_i x M_i, B_i is K_i x N_i) in ONE launch.  Every block
// computes one 256-tile of one problem (optionally one K split of it); the problem table travels in
// the kernel arguments.  Small outputs (a 512 x 512 weight is four tiles) then share the chip with
// the others instead of each being cut into dozens of K splits with their slab traffic.
struct GroupDesc {
  const void* A;
  const void* B;
  float* C;
  float* ws;         // slabs [splits][M][N] when splits > 1, followed by [splits][M] column-sum partials
  float* colsum;     // optional: out[m] (+)= sum_k A[k][m]
  int M, N, K;
  int lda, ldb, ldc;
  int first_block;   // blocks of the problems before this one
  int splits, kps;   // K splits and K per split
  int flags;         // bit 0: accumulate into C, bit 1: C allows 16-byte access, bit 2: accumulate into colsum
};
constexpr int MAX_GROUP = 44;
struct GroupArgs {
  int n, pad;
  GroupDesc d[MAX_GROUP];
};
static_assert(sizeof(GroupArgs) <= 4096, "kernel arguments are limited to 4 KB");

__device__ __forceinline__ GemmParams params_of(const GroupDesc& d) {
  GemmParams p;
  p.A = d.A; p.B = d.B; p.C = d.C; p.bias = nullptr; p.aux = nullptr; p.ws = d.ws;
  p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
  p.M = d.M; p.N = d.N; p.K = d.K;
  p.k_per_split = d.kps;
  p.tiles_m = (d.M + XM - 1) / XM;
  p.tiles_n = (d.N + XN - 1) / XN;
  p.epi = (d.flags & 1) ? SK_EPI_ACCUM : 0;
  p.vec_c = (d.flags >> 1) & 1;
  p.alpha = 1.f;
  p.group = 1;
  p.tile_h = XM;
  p.colsum = d.colsum;
  p.colsum_ws = d.ws ? d.ws + (int64_t)d.splits * d.M * d.N : nullptr;
  p.colsum_acc = (d.flags >> 2) & 1;
  return p;
}

__global__ __launch_bounds__(512, 1) void gemm_grouped_tn_kernel(GroupArgs g) {
  // blocks that share an XCD (and its L2) take consecutive work items: the tiles of one problem that
  // re-read each other's operand panels
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  int i = 0;
  while (i + 1 < g.n && g.d[i + 1].first_block <= bid) ++i;  // uniform: a few scalar loads
  const GemmParams p = params_of(g.d[i]);
  const int local = bid - g.d[i].first_block, tiles = p.tiles_m * p.tiles_n;
  const int split = local / tiles, tile = local - split * tiles;
  const int kbeg = split * p.k_per_split;
  gemm_tile<true, true, float>(p, tile / p.tiles_n, tile % p.tiles_n, kbeg, min(p.K, kbeg + p.k_per_split),
                               g.d[i].splits > 1 ? split : -1);
}

// Adds the K-split slabs of the split problems of a group (blockIdx.y = problem) in split order.
__global__ __launch_bounds__(256) void grouped_reduce_kernel(GroupArgs g) {
  const GroupDesc& d = g.d[blockIdx.y];
  if (d.splits <= 1) return;
  const int64_t total = (int64_t)d.M * d.N;
  const bool acc = d.flags & 1;
  if (d.colsum && blockIdx.x == 0) {  // the column-sum partials, in split order
    const float* part = d.ws + (int64_t)d.splits * total;
    for (int m = threadIdx.x; m < d.M; m += blockDim.x) {
      float sum = 0.f;
      for (int z = 0; z < d.splits; ++z) sum += part[(int64_t)z * d.M + m];
      d.colsum[m] = (d.flags & 4) ? d.colsum[m] + sum : sum;
    }
  }
  if ((d.N & 3) == 0 && (d.flags & 2)) {
    for (int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * blockDim.x * 4) {
      float4 s = *(const float4*)(d.ws + i);
      for (int z = 1; z < d.splits; ++z) {
        const float4 v = *(const float4*)(d.ws + (int64_t)z * total + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      float* c = d.C + (i / d.N) * d.ldc + (i % d.N);
      if (acc) { const float4 o = *(const float4*)c; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
      *(float4*)c = s;
    }
  } else {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      float s = 0.f;
      for (int z = 0; z < d.splits; ++z) s += d.ws[(int64_t)z * total + i];
      float* c = d.C + (i / d.N) * d.ldc + (i % d.N);
      *c = acc ? *c + s : s;
    }
  }
}
}  // namespace b16x

// ============================================================================ bf16, few rows, whole K in LDS
// The decode step's projections: a few hundred rows (sentences x beams) against a 512-deep weight.  With a
// handful of tiles per launch every tiled kernel above is a serial chain of K-steps, each exposing a load
// latency (12 us per launch whatever the width).  Here a block owns a 32 x 64 output tile and requests ALL of
// its operands at once — 32 + 64 rows of K elements, at most 96 KiB, straight into LDS — waits once, and then
// multiplies out of LDS: one memory latency per launch instead of one per K-step.
// LDS image: [row][K], row r's 16-byte chunk c at chunk position c ^ (r & 15) (applied to the source address;
// the direct-to-LDS destination is lane-linear), so that the 16 rows of a fragment read hit 16 different
// 16-byte slots.  Four waves; wave w owns output columns 16 w .. 16 w + 15 of the tile (two MFMA tiles).
namespace b16s {
constexpr int SM = 32, SN = 64, MAX_K = 512;
using b16g::gbl_ptr_t;
using b16g::lds_ptr_t;

template <typename TOut>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_n = blockIdx.x % p.tiles_n, tile_m = blockIdx.x / p.tiles_n;
  const int m0 = tile_m * SM, n0 = tile_n * SN;
  const int K = p.K, cr = K >> 3;          // 16-byte chunks per row (a power of two >= 16: host)
  const int row_bytes = K * 2;
  char* ldsA = smem;
  char* ldsB = smem + SM * row_bytes;
  const bf16_t* A = (const bf16_t*)p.A;
  const bf16_t* B = (const bf16_t*)p.B;
  // all loads: linear chunk index i = 256 * it + tid covers first A's SM * cr chunks, then B's SN * cr
  const int total = (SM + SN) * cr;
  for (int it = 0; it * 256 < total; ++it) {
    const int i = it * 256 + tid;            // total is a multiple of 256 (cr >= 16): no partial wave
    const bool isB = i >= SM * cr;
    const int j = isB ? i - SM * cr : i;
    const int r = j / cr, pos = j - r * cr;
    const int c = pos ^ (r & 15);
    const int64_t grow = isB ? min(n0 + r, p.N - 1) : min(m0 + r, p.M - 1);  // clamped rows are never stored
    const bf16_t* src = (isB ? B + grow * p.ldb : A + grow * p.lda) + 8 * c;
    char* dst = (isB ? ldsB : ldsA) + (int64_t)(j - lane) * 16;  // wave-uniform base: this wave's 64 chunks are contiguous
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  const int l15 = lane & 15, g = lane >> 4;
  f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
  const char* a0 = ldsA + l15 * row_bytes;
  const char* a1 = ldsA + (16 + l15) * row_bytes;
  const char* b0 = ldsB + (16 * wave + l15) * row_bytes;
  const int nk = K >> 5;
#pragma unroll 4
  for (int kk = 0; kk < nk; ++kk) {
    const int off = ((4 * kk + g) ^ l15) << 4;   // rows l15, 16 + l15 and 16 w + l15 share (row & 15) = l15
    const bf16x8_t fa0 = *(const bf16x8_t*)(a0 + off);
    const bf16x8_t fa1 = *(const bf16x8_t*)(a1 + off);
    const bf16x8_t fb = *(const bf16x8_t*)(b0 + off);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa1, acc[1], 0, 0, 0);
  }
  // C^T tiles: lane owns output row 16 mi + l15 and columns 16 w + 4 g + r
  const int col = n0 + 16 * wave + 4 * g;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int row = m0 + 16 * mi + l15;
    if (row >= p.M) continue;
    float v[4] = {acc[mi][0], acc[mi][1], acc[mi][2], acc[mi][3]};
    write_cols<TOut, bf16_t, 4>(p, v, row, col, -1);
  }
}
}  // namespace b16s

// ============================================================================ fp32
namespace f32 {
constexpr int BK = 16;
constexpr int LDK = 20;    // [rows][16 k] image, padded row (floats)
constexpr int LDT = 144;   // [16 k][rows] image, padded row (floats)
constexpr int OPER_BYTES = 128 * LDK * 4;  // 10240 >= 16*144*4
constexpr int SMEM_BYTES = 4 * OPER_BYTES;

template <bool T_LAYOUT, bool FAST>
__device__ __forceinline__ void load_tile(float4 (&reg)[2], const float* base, int64_t ld, int r0, int R, int k0,
                                          int kend, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 256 * i;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (T_LAYOUT) {
      const int kk = k0 + (id >> 5), rr = r0 + (id & 31) * 4;
      if (FAST) {
        if (kk < kend && rr + 4 <= R) v = *(const float4*)(base + (int64_t)kk * ld + rr);
      } else if (kk < kend) {
        const float* src = base + (int64_t)kk * ld + rr;
        float* e = (float*)&v;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (rr + j < R) e[j] = src[j];
      }
    } else {
      const int kk = k0 + (id & 3) * 4;
      if (FAST) {
        const int rr = min(r0 + (id >> 2), R - 1);
        if (kk + 4 <= kend) v = *(const float4*)(base + (int64_t)rr * ld + kk);
      } else {
        const int rr = r0 + (id >> 2);
        if (rr < R) {
          const float* src = base + (int64_t)rr * ld + kk;
          float* e = (float*)&v;
#pragma unroll
          for (int j = 0; j < 4; ++j) if (kk + j < kend) e[j] = src[j];
        }
      }
    }
    reg[i] = v;
  }
}

template <bool T_LAYOUT>
__device__ __forceinline__ void store_tile(const float4 (&reg)[2], float* lds, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 256 * i;
    if (T_LAYOUT) *(float4*)(lds + (id >> 5) * LDT + (id & 31) * 4) = reg[i];
    else          *(float4*)(lds + (id >> 2) * LDK + (id & 3) * 4) = reg[i];
  }
}

// lane l holds X[row0 + (l&15)][k = 4*kk + (l>>4)]
template <bool T_LAYOUT>
__device__ __forceinline__ float read_frag(const float* lds, int row0, int kk, int lane) {
  const int row = row0 + (lane & 15), k = kk * 4 + (lane >> 4);
  return T_LAYOUT ? lds[k * LDT + row] : lds[row * LDK + k];
}

template <bool A_T, bool B_T, bool FAST>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int tile_m, tile_n;
  tile_of(p, blockIdx.x, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int wrow = (wave >> 1) * 64, wcol = (wave & 1) * 64;

  const int kbeg = blockIdx.z * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;

  const float* A = (const float*)p.A;
  const float* B = (const float*)p.B;
  auto ldsA = [&](int buf) { return (float*)(smem + buf * OPER_BYTES); };
  auto ldsB = [&](int buf) { return (float*)(smem + (2 + buf) * OPER_BYTES); };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  float4 ra[2], rb[2];
  if (nk > 0) {
    load_tile<A_T, FAST>(ra, A, p.lda, m0, p.M, kbeg, kend, tid);
    load_tile<B_T, FAST>(rb, B, p.ldb, n0, p.N, kbeg, kend, tid);
    store_tile<A_T>(ra, ldsA(0), tid);
    store_tile<B_T>(rb, ldsB(0), tid);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) {
      load_tile<A_T, FAST>(ra, A, p.lda, m0, p.M, kbeg + (kt + 1) * BK, kend, tid);
      load_tile<B_T, FAST>(rb, B, p.ldb, n0, p.N, kbeg + (kt + 1) * BK, kend, tid);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = read_frag<A_T>(ldsA(cur), wrow + i * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) fb[i] = read_frag<B_T>(ldsB(cur), wcol + i * 16, kk, lane);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[ni], fa[mi], acc[mi][ni], 0, 0, 0);
    }
    if (more) {
      store_tile<A_T>(ra, ldsA(cur ^ 1), tid);
      store_tile<B_T>(rb, ldsB(cur ^ 1), tid);
    }
    __syncthreads();
  }
  write_tile<float, float, false>(p, acc, m0 + wrow, n0 + wcol, lane, grid_slab(), p.M);
}
}  // namespace f32

// ---- split-K second pass: sum the per-split fp32 slabs in a fixed order, then epilogue.
template <typename TOut, typename TIn>
__global__ void splitk_reduce_kernel(GemmParams p, int splits) {
  int64_t total = (int64_t)p.M * p.N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += p.ws[(int64_t)z * total + i];
    epilogue_store<TOut, TIn>(p, s, (int)(i / p.N), (int)(i % p.N));
  }
}

// ---- column sums (bias gradients): out[n] (+)= sum_m X[m][n]
// block = 64 columns x 4 row-lanes; grid.x over column groups, grid.y over row chunks whose
// partial sums go to partial[chunk][N]; a second kernel adds the chunks in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* X, float* partial, int64_t M, int64_t N, int64_t ldx,
                                                     int rows_per_block) {
  __shared__ float part[4][64];
  int c = blockIdx.x * 64 + (threadIdx.x & 63);
  int rl = threadIdx.x >> 6;
  int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  int64_t r1 = min(M, r0 + rows_per_block);
  float s = 0.f;
  if (c < N)
    for (int64_t r = r0 + rl; r < r1; r += 4) s += Cvt<T>::to_f32(X[r * ldx + c]);
  part[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < N)
    partial[(int64_t)blockIdx.y * N + c] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// Vectorised variant: 16 row-lanes x 16 column chunks of 16 bytes per block.
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* X, float* partial, int64_t M, int64_t N, int64_t ldx,
                                                         int rows_per_block) {
  constexpr int VN = Vec16<T>::N;
  __shared__ float part[16][16 * VN + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int64_t c0 = ((int64_t)blockIdx.x * 16 + tx) * VN;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float acc[VN];
#pragma unroll
  for (int j = 0; j < VN; ++j) acc[j] = 0.f;
  if (c0 < N)
    for (int64_t r = r0 + ty; r < r1; r += 16) {
      Vec16<T> v;
      v.v = *(const decltype(v.v)*)(X + r * ldx + c0);
#pragma unroll
      for (int j = 0; j < VN; ++j) acc[j] += v.get(j);
    }
#pragma unroll
  for (int j = 0; j < VN; ++j) part[ty][tx * VN + j] = acc[j];
  __syncthreads();
  const int col = threadIdx.x;  // one thread per column of the block (16*VN <= 128 columns)
  if (col < 16 * VN) {
    const int64_t c = (int64_t)blockIdx.x * 16 * VN + col;
    if (c < N) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += part[k][col];
      partial[(int64_t)blockIdx.y * N + c] = s;
    }
  }
}

// Adds the chunk partials in a fixed order: block = 64 columns x 4 chunk-lanes; lane q sums
// chunks q, q+4, ... (independent loads in flight), then the four lane sums are combined as
// (0 + 1) + (2 + 3).
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* partial, float* out, int64_t N, int chunks, int accumulate) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t c = blockIdx.x * 64ll + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < N) {
    int k = q;
    for (; k + 4 < chunks; k += 8) {
      s0 += partial[(int64_t)k * N + c];
      s1 += partial[(int64_t)(k + 4) * N + c];
    }
    if (k < chunks) s0 += partial[(int64_t)k * N + c];
  }
  part[q][cl] = s0 + s1;
  __syncthreads();
  if (q == 0 && c < N) {
    const float s = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
    out[c] = accumulate ? out[c] + s : s;
  }
}

template <typename K>
int launch_gemm(K kernel, dim3 grid, size_t smem, hipStream_t stream, const GemmParams& p) {
  hipLaunchKernelGGL(kernel, grid, dim3(256), smem, stream, p);
  return SK_OK;
}

}  // namespace

// ============================================================================ host side
static int gemm_one(int layout, const void* A, const void* B, void* C, const void* bias, const void* aux,
                    int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, float alpha,
                    int epilogue, int in_dtype, int out_dtype, void* workspace, int64_t workspace_bytes,
                    void* stream_);

// Entry point: validates nothing itself (gemm_one does) but cuts a long-K product whose 256-tiles
// spill a few tiles into a further, mostly empty round of the chip into two row ranges: the tile
// rows that fill whole rounds, and the remaining rows, which alone are fewer tiles than CUs and
// therefore get split-K.  (35000 x 512 x 25k weight gradient: 274 tiles = one full round plus 18
// tiles that would otherwise occupy 18 of 256 CUs for a second full-length round.)
extern "C" int skhip_gemm(int layout, const void* A, const void* B, void* C, const void* bias, const void* aux,
                          int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, float alpha,
                          int epilogue, int in_dtype, int out_dtype, void* workspace, int64_t workspace_bytes,
                          void* stream_) {
  const char* env_cut = getenv("SKHIP_GEMM_ROWCUT");
  if (in_dtype == SK_BF16 && workspace && K >= 8192 && (layout == SK_GEMM_NT || layout == SK_GEMM_TN) && A && C &&
      M > 0 && N > 0 &&
      M < (1ll << 31) && N < (1ll << 31) && !(epilogue & (SK_EPI_RELU_BITS_OUT | SK_EPI_RELU_MASK_BITS)) &&
      !(env_cut && env_cut[0] == '0')) {
    const int64_t tm = (M + 255) / 256, tn = (N + 255) / 256, ncu = sk_sched_cus();
    const int64_t t256 = tm * tn, rounds = t256 / ncu, tail = t256 % ncu;
    if (rounds >= 1 && rounds < 8 && tail > 0 && 2 * tail < ncu) {
      const int64_t rows_full = rounds * ncu / tn;  // whole tile rows inside the full rounds
      const int64_t M0 = rows_full * 256;
      if (M0 > 0 && M0 < M) {
        const int esz = 2, osz = out_dtype == SK_BF16 ? 2 : 4;
        const char* A1 = (const char*)A + (layout == SK_GEMM_TN ? M0 : M0 * lda) * esz;
        char* C1 = (char*)C + M0 * ldc * osz;
        const char* aux1 = aux ? (const char*)aux + M0 * ldc * esz : nullptr;
        int rc = gemm_one(layout, A, B, C, bias, aux, M0, N, K, lda, ldb, ldc, alpha, epilogue, in_dtype, out_dtype,
                          workspace, workspace_bytes, stream_);
        if (rc) return rc;
        return gemm_one(layout, A1, B, C1, bias, aux1, M - M0, N, K, lda, ldb, ldc, alpha, epilogue, in_dtype,
                        out_dtype, workspace, workspace_bytes, stream_);
      }
    }
  }
  return gemm_one(layout, A, B, C, bias, aux, M, N, K, lda, ldb, ldc, alpha, epilogue, in_dtype, out_dtype, workspace,
                  workspace_bytes, stream_);
}

static int gemm_one(int layout, const void* A, const void* B, void* C, const void* bias, const void* aux,
                    int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, float alpha,
                    int epilogue, int in_dtype, int out_dtype, void* workspace, int64_t workspace_bytes,
                    void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(layout >= 0 && layout <= 2, "skhip_gemm: bad layout %d", layout);
  SK_CHECK_ARG(in_dtype == SK_F32 || in_dtype == SK_BF16, "skhip_gemm: bad in_dtype %d", in_dtype);
  SK_CHECK_ARG(out_dtype == SK_F32 || (out_dtype == SK_BF16 && in_dtype == SK_BF16),
               "skhip_gemm: unsupported dtype pair in=%d out=%d", in_dtype, out_dtype);
  SK_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31),
               "skhip_gemm: bad shape %lld x %lld x %lld", (long long)M, (long long)N, (long long)K);
  if (M == 0 || N == 0) return SK_OK;
  SK_CHECK_ARG(A && B && C, "skhip_gemm: null operand");
  SK_CHECK_ARG(!(epilogue & SK_EPI_BIAS) || bias, "skhip_gemm: SK_EPI_BIAS without bias");
  SK_CHECK_ARG(!(epilogue & SK_EPI_RELU_MASK) || aux, "skhip_gemm: SK_EPI_RELU_MASK without aux");
  constexpr int kBits = SK_EPI_RELU_BITS_OUT | SK_EPI_RELU_MASK_BITS;
  if (epilogue & kBits) {
    SK_CHECK_ARG(aux && (epilogue & kBits) != kBits && !(epilogue & SK_EPI_RELU_MASK),
                 "skhip_gemm: a sign-pattern epilogue needs aux and excludes the other mask forms");
    SK_CHECK_ARG(!(epilogue & SK_EPI_RELU_BITS_OUT) || (epilogue & SK_EPI_RELU), "skhip_gemm: SK_EPI_RELU_BITS_OUT needs SK_EPI_RELU");
    SK_CHECK_ARG(in_dtype == SK_BF16 && layout == SK_GEMM_NT && N % 64 == 0 && (uintptr_t)aux % 8 == 0,
                 "skhip_gemm: sign-pattern epilogues are for bf16 NT products with N %% 64 == 0");
    workspace = nullptr;  // the pattern is written / applied in the epilogue: no split-K
  }
  if (epilogue & SK_EPI_ROWSTATS) {
    SK_CHECK_ARG(epilogue == SK_EPI_ROWSTATS && alpha == 1.0f && aux && (uintptr_t)aux % 8 == 0 && in_dtype == SK_BF16 &&
                 out_dtype == SK_F32 && layout == SK_GEMM_NT,
                 "skhip_gemm: SK_EPI_ROWSTATS is for plain bf16 NT products with fp32 output (alpha 1, no other epilogue)");
    workspace = nullptr;  // the statistics are taken in the epilogue: no split-K
  }
  const bool a_t = layout == SK_GEMM_TN, b_t = layout != SK_GEMM_NT;
  SK_CHECK_ARG(lda >= (a_t ? M : K) && ldb >= (b_t ? N : K) && ldc >= N, "skhip_gemm: leading dimension too small");

  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.bias = (const float*)bias; p.aux = aux;
  p.ws = nullptr;
  p.colsum = p.colsum_ws = nullptr;
  p.colsum_acc = 0;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.tiles_m = (int)((M + BM - 1) / BM);
  p.tiles_n = (int)((N + BN - 1) / BN);
  p.epi = epilogue;
  p.alpha = alpha;
  // Grouped rasterisation pays when both operands are long panels that do not fit the L2s
  // (measured on 8192^3: 870 -> 1317 TFLOP/s with the 256-tile kernel); with a short K the
  // row-major order is better (its concurrent tiles write adjacent pieces of the same C rows).
  p.group = (K >= 4096 && N >= 2048 && M >= 2048) ? 8 : 1;
  // a very wide output with a short K (generator: 25 k x 35 000 x 512): pairs of tile rows halve the
  // re-reads of the weight panels (measured 1141 -> 1085 us; 4 rows 1108, 8 rows 1167, 32 rows 1507)
  if (p.group == 1 && N >= 8192 && M >= 2048) p.group = 2;
  { const char* e = getenv("SKHIP_GEMM_GROUP"); if (e) p.group = atoi(e); }
  const int esz = in_dtype == SK_BF16 ? 2 : 4;
  const int vecn = 16 / esz;
  const int osz = out_dtype == SK_BF16 ? 2 : 4;
  // FAST loads: aligned operands; the contiguous dimension of each operand a multiple of the
  // 16-byte chunk (K for K-contiguous operands, the row count for row-contiguous ones).
  bool fast = ((uintptr_t)A % 16 == 0) && (lda % vecn == 0) && ((uintptr_t)B % 16 == 0) && (ldb % vecn == 0);
  fast = fast && ((a_t ? M : K) % vecn == 0) && ((b_t ? N : K) % vecn == 0);
  p.vec_c = ((uintptr_t)C % 16 == 0) && (ldc % (16 / osz) == 0) &&
            (!(epilogue & SK_EPI_RELU_MASK) || ((uintptr_t)aux % 16 == 0 && ldc % vecn == 0)) &&
            (!(epilogue & SK_EPI_BIAS) || (uintptr_t)bias % 16 == 0);

  SK_CHECK_ARG(!(epilogue & kBits) || (fast && p.vec_c), "skhip_gemm: sign-pattern epilogues need 16-byte aligned operands");
  // SKHIP_GEMM_STAGED=1 selects the register-staged bf16 kernel for A/B comparisons.
  // Measured on the Transformer-base shapes (M ~ 33k, K, N in {512, 2048, 35000}): the
  // direct-to-LDS kernel wins for NT and short-K NN, the register-staged BK=64 kernel for TN
  // (weight gradients) and long-K NN (generator input gradient).  SKHIP_GEMM_STAGED=0/1 forces one.
  const char* env_staged = getenv("SKHIP_GEMM_STAGED");
  bool use_staged = layout == SK_GEMM_TN || (layout == SK_GEMM_NN && K >= 8192);
  // a few hundred rows against a long contraction with split-K (decode step, 640 x 512 x 2048): the
  // register-staged kernel's BK = 64 halves the serial K-steps of each split (18 vs 24 us per call)
  if (layout == SK_GEMM_NT && M <= 1024 && K >= 2048 && workspace) use_staged = true;
  if (env_staged && (env_staged[0] == '0' || env_staged[0] == '1')) use_staged = env_staged[0] == '1';
  // 256x256-tile kernel: K-contiguous operands (NT).  One block per CU, so what decides is how
  // well the 256-tiles fill whole rounds of the chip (measured, 256 CUs, M = 25k..33k rows):
  // it wins with many more tiles than CUs (generator: 835 vs 670 TFLOP/s), with a long K when the
  // last round is mostly full (N = 512, K = 2048: 1000 vs 695; K = 35000: 711 vs 408), and for a
  // short K only when everything fits a single round (200 tiles: 580 vs 479; 260 tiles: 460 vs 584).
  // SKHIP_GEMM_BIG=0/1 forces the choice (1 still requires NT + aligned operands).
  const char* env_big = getenv("SKHIP_GEMM_BIG");
  bool use_big = false;
  // A long-K NN product (generator input gradient: 25k x 512 x 35000 against the weight in its own
  // (V, d) layout) also runs on it, and there beats the NT form against a transposed weight copy
  // (1060 vs 1290 us: the 512-byte weight rows load in longer pieces); for K <= 2048 NT stays ahead.
  const bool nn_long = layout == SK_GEMM_NN && K >= 8192;
  if (in_dtype == SK_BF16 && fast && ((layout == SK_GEMM_NT && !use_staged) || (nn_long && !(env_staged && env_staged[0] == '1')))) {
    const int64_t t256 = ((M + 255) / 256) * ((N + 255) / 256), ncu = sk_sched_cus();
    const int64_t rounds = (t256 + ncu - 1) / ncu;
    const double fill = (double)t256 / (double)(rounds * ncu);
    use_big = t256 >= 8 * ncu || (K >= 1024 && fill >= 0.7) || (rounds == 1 && fill >= 0.7);
    // whole 256-column tiles of a tall product: the 256-tile kernel's full-tile epilogue stores whole 128-byte
    // lines (through LDS) where the 128-tile kernel stores 16-byte pieces; inside the training step that outweighs
    // the 128-tile kernel's better round fill (17.93 -> 17.73 ms per step, interleaved in one process)
    // (not below three quarters of a round of tiles: those products would be K-split, with another summation order)
    if (layout == SK_GEMM_NT && M >= 8192 && N % 256 == 0 && out_dtype == SK_BF16 && 4 * t256 >= 3 * ncu) use_big = true;
    if (use_big && nn_long) use_staged = false;
  }
  // Weight gradients (TN, long K, split-K): the 256-tile kernel when the output has enough 256-tiles
  // to keep the split count, and with it the slab traffic, moderate (measured at K = 25k: 2048x512
  // 81 vs 98 us, 1536x512 66 vs 72, 35000x512 1365 vs 1461; 1024x512 and 512x512 are slower).
  if (in_dtype == SK_BF16 && fast && layout == SK_GEMM_TN && workspace && K >= 4096)
    use_big = ((M + 255) / 256) * ((N + 255) / 256) >= 12;
  if (env_big && env_big[0] == '0') use_big = false;
  if (env_big && env_big[0] == '1') use_big = in_dtype == SK_BF16 && fast;
  if (epilogue & SK_EPI_ROWSTATS) {  // the statistics hook lives in the 128-tile kernel's epilogue
    SK_CHECK_ARG(fast, "skhip_gemm: SK_EPI_ROWSTATS needs 16-byte aligned operands with K %% 8 == 0");
    use_big = false;
    use_staged = false;
  }
  p.tile_h = b16x::XM;
  if (use_big) {
    p.tiles_n = (int)((N + b16x::XN - 1) / b16x::XN);
    // One block per CU: a launch takes whole rounds of the chip.  With a K-contiguous A the kernel can work on
    // lower tiles, so pick the height (a multiple of 16) whose rounds cost least.  Measured at M = 25.6 k (256 ->
    // 208 rows): a tile's time follows (rows + 256) rather than its rows (the B half of every K-step, the
    // barriers and the fragment reads stay): K = 35000 NN -9 %, K = 2048 -5..-8 %, N = 1536 / 2048 with K = 512
    // (three rounds and a bit -> three or four full ones) -3..-4 %; with K = 512 and one or two rounds the tile
    // time is latency, not work, and more blocks were 3-7 % slower, so those keep 256 rows.  Not below 0.6 of a
    // round of 256-tiles (split-K fills the chip there) nor from 16 rounds on (the last round's fill no longer
    // matters).  SKHIP_GEMM_TILE_H forces a height.
    const int64_t ncu = sk_sched_cus(), t256 = ((M + 255) / 256) * p.tiles_n;
    if (!a_t && 10 * t256 >= 6 * ncu && t256 < 16 * ncu) {
      if (K >= 1024 || t256 > 2 * ncu) {
        double best = 0;
        for (int th = 256; th >= 144; th -= 16) {
          const int64_t blocks = ((M + th - 1) / th) * p.tiles_n;
          const double cost = (double)((blocks + ncu - 1) / ncu) * (th + 256);
          if (th == 256 || cost < best) { best = cost; p.tile_h = th; }
        }
      }
      const char* e = getenv("SKHIP_GEMM_TILE_H");
      if (e && atoi(e) >= 144 && atoi(e) <= 256 && atoi(e) % 16 == 0) p.tile_h = atoi(e);
    }
    p.tiles_m = (int)((M + p.tile_h - 1) / p.tile_h);
  }
  const int bk = in_dtype == SK_BF16 ? (use_big ? b16x::BK : (fast && !use_staged) ? b16g::BK : b16::BK) : f32::BK;
  const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n;
  SK_CHECK_ARG(tiles < (1ll << 31), "skhip_gemm: too many tiles");
  // deterministic split-K when the output alone cannot fill the chip
  int splits = 1;
  const int cus = sk_sched_cus();
  if (workspace && tiles < cus && K >= 8 * bk) {
    // as many splits as fill the resident block slots of the chosen kernel WITHOUT spilling into a
    // further, mostly empty round (48 tiles x 11 splits on 512 slots ran 3 rounds instead of 2)
    const int64_t slots = (int64_t)cus * (in_dtype != SK_BF16 ? 2 : use_big ? 1 : (fast && !use_staged) ? 3 : 2);
    int64_t want = slots / tiles;
    int64_t max_by_k = K / (4 * bk);
    int64_t max_by_ws = workspace_bytes / (int64_t)(M * N * 4);
    int64_t s = want < max_by_k ? want : max_by_k;
    if (s > max_by_ws) s = max_by_ws;
    if (s > 64) s = 64;
    if (s > 1) splits = (int)s;
  }
  int kps = (int)((K + splits - 1) / splits);
  kps = (kps + bk - 1) / bk * bk;
  splits = (int)((K + kps - 1) / kps);
  if (splits < 1) splits = 1;
  p.k_per_split = kps;
  if (splits > 1) p.ws = (float*)workspace;

  // Few rows against a short contraction (the decode step's projections): the whole-K-in-LDS kernel.
  {
    const char* e = getenv("SKHIP_GEMM_SKINNY");
    const bool allowed = !(e && e[0] == '0') && !(env_big && env_big[0] == '1') && !(env_staged && env_staged[0] == '1');
    if (allowed && in_dtype == SK_BF16 && layout == SK_GEMM_NT && fast && M <= 1024 && K >= 128 && K <= b16s::MAX_K &&
        (K & (K - 1)) == 0 && N % b16s::SN == 0 && !(epilogue & ~(SK_EPI_BIAS | SK_EPI_RELU))) {
      p.tiles_m = (int)((M + b16s::SM - 1) / b16s::SM);
      p.tiles_n = (int)(N / b16s::SN);
      p.k_per_split = (int)K;
      const size_t lds = (size_t)(b16s::SM + b16s::SN) * K * 2;
      const dim3 sgrid((unsigned)(p.tiles_m * p.tiles_n));
      if (out_dtype == SK_BF16) {
        int rc = sk_allow_big_lds(b16s::gemm_kernel<bf16_t>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((b16s::gemm_kernel<bf16_t>), sgrid, dim3(256), lds, stream, p);
      } else {
        int rc = sk_allow_big_lds(b16s::gemm_kernel<float>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((b16s::gemm_kernel<float>), sgrid, dim3(256), lds, stream, p);
      }
      SK_CHECK_LAUNCH("skhip_gemm(few rows)");
      return SK_OK;
    }
  }
  dim3 grid((unsigned)tiles, 1, (unsigned)splits);
#define LAUNCH_B16(AT, BT, TO)                                                                              \
  (fast ? (use_staged ? launch_gemm(b16::gemm_kernel<AT, BT, TO, true>, grid, b16::SMEM_BYTES, stream, p)   \
                      : launch_gemm(b16g::gemm_kernel<AT, BT, TO>, grid, b16g::SMEM_BYTES, stream, p))      \
        : launch_gemm(b16::gemm_kernel<AT, BT, TO, false>, grid, b16::SMEM_BYTES, stream, p))
#define LAUNCH_F32(AT, BT)                                                                                  \
  (fast ? launch_gemm(f32::gemm_kernel<AT, BT, true>, grid, f32::SMEM_BYTES, stream, p)                     \
        : launch_gemm(f32::gemm_kernel<AT, BT, false>, grid, f32::SMEM_BYTES, stream, p))
  if (use_big) {
#define LAUNCH_BIG(AT, BT, TO)                                                                          \
  do {                                                                                                  \
    int rc = sk_allow_big_lds(b16x::gemm_kernel<AT, BT, TO>, b16x::LDS_BYTES);                         \
    if (rc) return rc;                                                                                  \
    hipLaunchKernelGGL((b16x::gemm_kernel<AT, BT, TO>), grid, dim3(512), b16x::LDS_BYTES, stream, p);  \
  } while (0)
    if (out_dtype == SK_BF16) {
      if (layout == SK_GEMM_NT) LAUNCH_BIG(false, false, bf16_t);
      else if (layout == SK_GEMM_NN) LAUNCH_BIG(false, true, bf16_t);
      else LAUNCH_BIG(true, true, bf16_t);
    } else {
      if (layout == SK_GEMM_NT) LAUNCH_BIG(false, false, float);
      else if (layout == SK_GEMM_NN) LAUNCH_BIG(false, true, float);
      else LAUNCH_BIG(true, true, float);
    }
#undef LAUNCH_BIG
  } else if (in_dtype == SK_BF16) {
    if (out_dtype == SK_BF16) {
      if (layout == SK_GEMM_NT) LAUNCH_B16(false, false, bf16_t);
      else if (layout == SK_GEMM_NN) LAUNCH_B16(false, true, bf16_t);
      else LAUNCH_B16(true, true, bf16_t);
    } else {
      if (layout == SK_GEMM_NT) LAUNCH_B16(false, false, float);
      else if (layout == SK_GEMM_NN) LAUNCH_B16(false, true, float);
      else LAUNCH_B16(true, true, float);
    }
  } else {
    if (layout == SK_GEMM_NT) LAUNCH_F32(false, false);
    else if (layout == SK_GEMM_NN) LAUNCH_F32(false, true);
    else LAUNCH_F32(true, true);
  }
  SK_CHECK_LAUNCH("skhip_gemm");
  if (splits > 1) {
    int64_t total = M * N;
    int rb = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (in_dtype == SK_BF16 && out_dtype == SK_BF16)
      hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t, bf16_t>), dim3(rb), dim3(256), 0, stream, p, splits);
    else if (in_dtype == SK_BF16)
      hipLaunchKernelGGL((splitk_reduce_kernel<float, bf16_t>), dim3(rb), dim3(256), 0, stream, p, splits);
    else
      hipLaunchKernelGGL((splitk_reduce_kernel<float, float>), dim3(rb), dim3(256), 0, stream, p, splits);
    SK_CHECK_LAUNCH("skhip_gemm(split-K reduce)");
  }
  return SK_OK;
}

// Grouped weight gradients: n independent products C_i (M_i x N_i fp32, row stride ldc_i) (+)= A_i^T B_i
// with row-contiguous bf16 operands A_i (K_i x M_i, row stride lda_i), B_i (K_i x N_i, ldb_i), all in as
// few launches as the 4 KB argument limit allows (48 sub-problems each).  `problems` is a HOST array
// of n x 12 int64: {A, B, C, M, N, K, lda, ldb, ldc, accumulate, colsum, colsum_accumulate} (colsum: optional
// fp32 [M] that receives the column sums of A_i, i.e. the bias gradient that goes with the weight gradient).  Scheduling: the 256-tiles of all
// problems are laid out in order; those that fill whole rounds of the chip's CUs run unsplit, the
// remainder (fewer tiles than CUs) is cut off at a tile-row boundary and K-split so that it, too,
// occupies every CU once, with slabs in `workspace` and a reduction in split order (deterministic).
constexpr int kProblemFields = 12;
extern "C" int skhip_gemm_grouped_tn(const int64_t* problems, int n, void* workspace, int64_t workspace_bytes,
                                     void* stream_) {
  using b16x::GroupArgs;
  using b16x::GroupDesc;
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(n >= 0 && (n == 0 || problems), "skhip_gemm_grouped_tn: bad problem table");
  if (n == 0) return SK_OK;
  const int ncu = sk_sched_cus();
  int64_t total_tiles = 0;
  for (int i = 0; i < n; ++i) {
    const int64_t* q = problems + kProblemFields * i;
    const int64_t M = q[3], N = q[4], K = q[5], lda = q[6], ldb = q[7], ldc = q[8];
    SK_CHECK_ARG(q[0] && q[1] && q[2], "skhip_gemm_grouped_tn: null operand in problem %d", i);
    SK_CHECK_ARG(M > 0 && N > 0 && K > 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31) &&
                 lda < (1ll << 31) && ldb < (1ll << 31) && ldc < (1ll << 31),
                 "skhip_gemm_grouped_tn: bad shape in problem %d", i);
    SK_CHECK_ARG(lda >= M && ldb >= N && ldc >= N, "skhip_gemm_grouped_tn: leading dimension too small in problem %d", i);
    SK_CHECK_ARG(q[0] % 16 == 0 && q[1] % 16 == 0 && M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && q[2] % 4 == 0,
                 "skhip_gemm_grouped_tn: problem %d is not 16-byte aligned (operands, M, N, lda, ldb in elements of 8)", i);
    total_tiles += ((M + 255) / 256) * ((N + 255) / 256);
  }
  SK_CHECK_ARG(total_tiles < (1ll << 30), "skhip_gemm_grouped_tn: too many tiles");

  // ---- plan: sub-problems (row ranges) with their split counts
  struct Sub { int prob; int64_t r0, r1; int splits; };
  std::vector<Sub> subs;
  const int64_t full_tiles = total_tiles / ncu * ncu;
  auto max_splits_of = [&](int64_t K) { int64_t m = K / (4 * b16x::BK); return (int)(m < 1 ? 1 : m > 32 ? 32 : m); };
  // K splits for `t` tiles that have the chip to themselves: the count that minimises rounds / splits
  // (time in units of one unsplit tile), with a small charge per split for its slab traffic.
  auto best_splits = [&](int64_t t) {
    int best = 1;
    double best_cost = 1e30;
    for (int sp = 1; sp <= 32; ++sp) {
      const double cost = (double)((t * sp + ncu - 1) / ncu) / sp + 0.02 * sp;
      if (cost < best_cost - 1e-9) { best_cost = cost; best = sp; }
    }
    return best;
  };
  if (total_tiles < ncu) {  // not even one round: split everything alike
    const int sp = best_splits(total_tiles);
    for (int i = 0; i < n; ++i) subs.push_back({i, 0, problems[kProblemFields * i + 3], sp});
  } else {
    int64_t used = 0;
    bool in_tail = false;
    int64_t tail_tiles = 0;
    std::vector<Sub> tail;
    for (int i = 0; i < n; ++i) {
      const int64_t M = problems[kProblemFields * i + 3], tn = (problems[kProblemFields * i + 4] + 255) / 256, tm = (M + 255) / 256;
      if (in_tail) { tail.push_back({i, 0, M, 1}); tail_tiles += tm * tn; continue; }
      if (used + tm * tn <= full_tiles) { subs.push_back({i, 0, M, 1}); used += tm * tn; continue; }
      const int64_t rows_here = (full_tiles - used) / tn;  // whole tile rows that still fit the full rounds
      if (rows_here > 0) subs.push_back({i, 0, rows_here * 256, 1});
      tail.push_back({i, rows_here * 256, M, 1});
      tail_tiles += (tm - rows_here) * tn;
      used += rows_here * tn;
      in_tail = true;
    }
    const int sp = tail_tiles > 0 ? best_splits(tail_tiles) : 1;
    for (auto& t : tail) { t.splits = sp; subs.push_back(t); }
  }

  // one split count for all split sub-problems (they run as equal work items): lower it until the
  // slabs of all of them fit the workspace
  {
    int sp = 1;
    for (const Sub& sb : subs) sp = std::max(sp, sb.splits);
    auto slab_bytes = [&](int spl) {
      int64_t b = 0;
      for (const Sub& sb : subs)
        if (sb.splits > 1) {
          const int64_t K = problems[kProblemFields * sb.prob + 5];
          b += (int64_t)std::min(spl, max_splits_of(K)) * (sb.r1 - sb.r0) * (problems[kProblemFields * sb.prob + 4] + 1) * 4;
        }
      return b;
    };
    while (sp > 1 && (!workspace || slab_bytes(sp) > workspace_bytes)) --sp;
    for (Sub& sb : subs) sb.splits = std::min(sb.splits, sp);
  }

  // ---- descriptors, slabs
  std::vector<GroupDesc> descs;
  int64_t ws_used = 0;
  for (const Sub& sb : subs) {
    const int64_t* q = problems + kProblemFields * sb.prob;
    GroupDesc d;
    d.M = (int)(sb.r1 - sb.r0); d.N = (int)q[4]; d.K = (int)q[5];
    d.lda = (int)q[6]; d.ldb = (int)q[7]; d.ldc = (int)q[8];
    d.A = (const char*)q[0] + sb.r0 * 2;
    d.B = (const void*)q[1];
    d.C = (float*)q[2] + sb.r0 * d.ldc;
    int splits = sb.splits < max_splits_of(d.K) ? sb.splits : max_splits_of(d.K);
    const int64_t slab = (int64_t)d.M * (d.N + 1) * 4;  // C partials and one row of column-sum partials
    if (splits > 1 && ws_used + splits * slab > workspace_bytes) splits = 1;  // cannot happen after the uniform cut above
    int kps = (d.K + splits - 1) / splits;
    kps = (kps + b16x::BK - 1) / b16x::BK * b16x::BK;
    splits = (d.K + kps - 1) / kps;
    d.splits = splits; d.kps = kps;
    d.ws = nullptr;
    if (splits > 1) { d.ws = (float*)((char*)workspace + ws_used); ws_used += splits * slab; }
    d.colsum = q[10] ? (float*)q[10] + sb.r0 : nullptr;
    d.flags = (q[9] ? 1 : 0) | ((((uintptr_t)d.C % 16 == 0) && d.ldc % 4 == 0) ? 2 : 0) | (q[11] ? 4 : 0);
    d.first_block = 0;
    descs.push_back(d);
  }

  // ---- launches, at most MAX_GROUP descriptors each
  int rc = sk_allow_big_lds(b16x::gemm_grouped_tn_kernel, b16x::LDS_BYTES);
  if (rc) return rc;
  for (size_t c0 = 0; c0 < descs.size(); c0 += b16x::MAX_GROUP) {
    GroupArgs g;
    g.n = (int)std::min<size_t>(b16x::MAX_GROUP, descs.size() - c0);
    g.pad = 0;
    int blocks = 0, max_red = 0;
    for (int i = 0; i < g.n; ++i) {
      g.d[i] = descs[c0 + i];
      g.d[i].first_block = blocks;
      const int tiles = ((g.d[i].M + 255) / 256) * ((g.d[i].N + 255) / 256);
      blocks += tiles * g.d[i].splits;
      if (g.d[i].splits > 1) {
        const int64_t need = ((int64_t)g.d[i].M * g.d[i].N / 4 + 255) / 256;
        max_red = (int)std::max<int64_t>(max_red, need < 1 ? 1 : need > 512 ? 512 : need);
      }
    }
    hipLaunchKernelGGL(b16x::gemm_grouped_tn_kernel, dim3(blocks), dim3(512), b16x::LDS_BYTES, stream, g);
    SK_CHECK_LAUNCH("skhip_gemm_grouped_tn");
    if (max_red > 0) {
      hipLaunchKernelGGL(b16x::grouped_reduce_kernel, dim3(max_red, g.n), dim3(256), 0, stream, g);
      SK_CHECK_LAUNCH("skhip_gemm_grouped_tn(reduce)");
    }
  }
  return SK_OK;
}

// Column sums with a fixed summation order (bitwise reproducible): the rows are cut into up
// to 64 chunks whose partial sums go to the caller's workspace and are added in chunk order
// by a second kernel.  Without a workspace a single chunk per column group is used.
// (>= 128*N floats of workspace allow the full chunk count.)
extern "C" int skhip_colsum(const void* X, float* out, int64_t M, int64_t N, int64_t ldx, int accumulate,
                            int dtype, void* workspace, int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(dtype == SK_F32 || dtype == SK_BF16, "skhip_colsum: bad dtype");
  SK_CHECK_ARG(M >= 0 && N > 0 && ldx >= N, "skhip_colsum: bad shape");
  SK_CHECK_ARG(X && out, "skhip_colsum: null pointer");
  float* partial = (float*)workspace;
  const int vn = dtype == SK_BF16 ? 8 : 4;
  const bool vec = ((uintptr_t)X % 16 == 0) && (ldx % vn == 0) && (N % vn == 0);
  int col_groups = vec ? (int)((N + 16 * vn - 1) / (16 * vn)) : (int)((N + 63) / 64);
  int chunks = 1;
  const int cus = sk_sched_cus();
  while (col_groups * chunks < 4 * cus && chunks < 128 && M / (chunks * 2) >= 64) chunks *= 2;
  while (chunks > 1 && (!partial || (int64_t)chunks * N * 4 > workspace_bytes)) chunks /= 2;
  int rows_per_block = (int)((M + chunks - 1) / chunks);
  dim3 grid(col_groups, chunks), block(256);
  if (chunks == 1 && !accumulate) partial = out;  // single chunk: write the result directly
  SK_CHECK_ARG(partial, "skhip_colsum: accumulate needs a workspace of N floats");
  if (vec && dtype == SK_BF16)
    hipLaunchKernelGGL((colsum_vec_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)X, partial, M, N, ldx, rows_per_block);
  else if (vec)
    hipLaunchKernelGGL((colsum_vec_kernel<float>), grid, block, 0, stream, (const float*)X, partial, M, N, ldx, rows_per_block);
  else if (dtype == SK_BF16)
    hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)X, partial, M, N, ldx, rows_per_block);
  else
    hipLaunchKernelGGL((colsum_kernel<float>), grid, block, 0, stream, (const float*)X, partial, M, N, ldx, rows_per_block);
  SK_CHECK_LAUNCH("skhip_colsum");
  if (partial != out) {
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, stream, partial, out, N, chunks, accumulate);
    SK_CHECK_LAUNCH("skhip_colsum(finish)");
  }
  return SK_OK;
}

// ---- column sums for callers that cannot allocate (HIP graph capture, serving loops, harnesses that
// forbid hipMalloc): the chunk partials live in a file-scope device array instead of a caller workspace.
// Every GPU holds its own copy of the array at its own address; the address is looked up once per
// device (hipGetSymbolAddress with that device current) and only that pointer is ever handed to
// kernels.  Every partial that the finishing kernel reads has been written by the first kernel of the
// same call, so nothing has to be cleared between calls; two calls that overlap on one GPU (different
// streams) would share the array and race.
constexpr int64_t kColsumScratchFloats = 512 * 1024;  // 2 MiB per GPU, held until the process ends
static __device__ __align__(256) float g_colsum_scratch[kColsumScratchFloats];
constexpr int kMaxDevices = 64;
static float* g_colsum_scratch_ptr[kMaxDevices];  // per device; nullptr until looked up
static std::mutex g_colsum_scratch_mutex;

// Device address of the current device's copy of the scratch array (looked up on first use).
int sk_colsum_scratch(float** ptr) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    sk_set_error("skhip_colsum_noalloc: no usable current device (%s)", e == hipSuccess ? "index out of range" : hipGetErrorString(e));
    return SK_ERR_LAUNCH;
  }
  std::lock_guard<std::mutex> lock(g_colsum_scratch_mutex);
  if (!g_colsum_scratch_ptr[dev]) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_colsum_scratch));
    if (e != hipSuccess || !p) {
      sk_set_error("skhip_colsum_noalloc: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
      return SK_ERR_LAUNCH;
    }
    g_colsum_scratch_ptr[dev] = (float*)p;
  }
  *ptr = g_colsum_scratch_ptr[dev];
  return SK_OK;
}

extern "C" int skhip_colsum_noalloc(const void* X, float* out, int64_t M, int64_t N, int64_t ldx, int accumulate,
                                    int dtype, void* stream_) {
  SK_CHECK_ARG(N > 0 && N <= SK_COLSUM_NOALLOC_MAX_N,
               "skhip_colsum_noalloc: N = %lld is outside 1..%d (the built-in scratch holds %lld floats of partial sums)",
               (long long)N, SK_COLSUM_NOALLOC_MAX_N, (long long)kColsumScratchFloats);
  float* scratch = nullptr;
  int rc = sk_colsum_scratch(&scratch);
  if (rc) return rc;
  return skhip_colsum(X, out, M, N, ldx, accumulate, dtype, scratch, kColsumScratchFloats * 4, stream_);
}
