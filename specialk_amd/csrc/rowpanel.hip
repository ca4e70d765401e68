// Row-panel GEMMs for d_model = 512 (gfx950): the activation-sized products of the Transformer layers with a
// 512-deep contraction (lib/transformer/SubLayers.py:46-48, 62-63, 73, 80 and their autograd backward) with the token
// rows, not output tiles, as the unit of work.
//
// A workgroup owns a PANEL of 104 token rows: 25.6 k rows / 104 = 247 panels, ONE round of the 256 CUs, where
// 256-row output tiles are 100.2 tile rows (78 % of two rounds) and a K = 512 tile spends a third of its life in
// prologue and epilogue.  The panel ([104][512] bf16, 1 KB rows) is brought into LDS once and multiplied by the
// N / 64 column groups of the weight in turn: feed-forward w_1 (+ bias, ReLU, sign pattern), its backward (dz w_2,
// masked by the pattern), the stacked q | k | v and key | value projections; with N = 512 the epilogue can be the
// whole sub-layer tail: bias, dropout, residual, LayerNorm and pad-row zeroing on the full 512-wide rows (the
// attention blocks' output projection).
//
// Shape of the computation.  512 threads = 8 waves, two per SIMD.  The activation operand is in LDS, 16-byte chunks
// of a row XOR-swizzled with row & 15 so that the ds_read_b128 fragment reads are bank-conflict free.  A wave
// computes all rows x one 64-COLUMN GROUP at a time (7 x 4 MFMA tiles, 112 accumulator registers).  It therefore
// needs its own slice of the weights only, and streams it from global memory (the L2: every panel reads the same
// weights) STRAIGHT INTO MFMA operand registers — no LDS staging of weights, no barrier inside a K loop, each wave
// paced by its own counted waits.  The weights are stored fragment-major for that (see skhip_cast2d_multi): one
// load instruction of a wave fetches 1 KB of consecutive bytes, a K-step of a wave 4 KB, and four half-K-steps
// ("units") are in flight per wave.  The two waves of a SIMD run staggered — one multiplies while the other
// converts and stores — and the first (which the arbiter prefers) takes 5 of every 8 column groups of the pair.
//
// MFMA operand order: v_mfma_f32_16x16x32_bf16(weight fragment, activation fragment): lane (l15 = lane & 15,
// g = lane >> 4) supplies k = 8 g .. 8 g + 7 of weight-tile row l15 and of activation row l15 and receives
// out[activation row l15][weight-tile rows 4 g .. 4 g + 3].  The fragment-major layout permutes the weight rows
// inside a 64-column group so that a lane ends up with 16 consecutive output columns: tile ni, tile row i
// <->  column  16 (i >> 2) + 4 ni + (i & 3)  of the group.
#include "common.h"
#include <algorithm>

namespace {
namespace rp {

constexpr int D = 512;               // columns of a panel (K of panel_kernel, N of both)
constexpr int R = SK_ROWPANEL_ROWS;  // rows per panel
constexpr int NMI = 7;               // 16-row groups computed (rows 104 .. 111 are never stored)
constexpr int XROW = D * 2;
constexpr int X_BYTES = R * XROW;    // 106496: the panel
constexpr int PART_BYTES = R * 8 * 4;  // LayerNorm partial sums [row][wave]
constexpr int LDS_BYTES = X_BYTES + 2 * PART_BYTES;
static_assert(R <= 16 * NMI && R % 8 == 0, "panel rows");

enum { E_STORE = 0, E_RELU_BITS = 1, E_MASK_BITS = 2, E_ACCUM = 3, E_LN = 4 };

struct Params {
  const bf16_t* A; int64_t lda;      // activation rows [T][K]
  const bf16_t* W;                   // fragment-major weights, 4 tiles per group: [N / 64][K / 32][4][64][8]
  const float* bias;                 // [N] (kernels instantiated with BIAS)
  bf16_t* C; int64_t ldc;            // result [T][N] (E_LN: the sub-layer output; E_ACCUM: added to in place)
  void* signs;                       // E_RELU_BITS / E_MASK_BITS: ReLU pattern (see pattern_index)
  const bf16_t* res; int64_t ldres;  // E_LN: residual rows
  const float* gamma; const float* beta; const uint8_t* row_pad;
  bf16_t* Y; float* mean; float* rstd;
  float eps; uint32_t drop_thr; float keep_scale; uint64_t seed;
  int T, N, K;
  unsigned long long* stamps;  // diagnostics (SKHIP_RP_STAMPS): per workgroup and wave 32 clock readings, else null
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void lds_barrier() {
  // LDS traffic of this wave done, then the workgroup barrier; vector-memory operations (the weight stream, stores)
  // stay in flight across it
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  uint4 o;
  bf16_t* e = (bf16_t*)&o;
#pragma unroll
  for (int j = 0; j < 8; ++j) e[j] = f32_to_bf16(v[j]);
  return o;
}

// Diagnostics: clock reading number `slot` of this wave (only when a stamp buffer was passed).
__device__ __forceinline__ void stamp(const Params& p, int wave, int lane, int slot) {
#ifdef SK_RP_STAMPS  // (a diagnostics build, `make stamps`: the branches and stores cost registers in the loops)
  if (p.stamps && lane == 0 && slot < 31) {
    unsigned long long* q = p.stamps + ((int64_t)blockIdx.x * 8 + wave) * 32;
    q[slot] = __builtin_amdgcn_s_memtime();
    if (slot == 0) q[31] = __builtin_amdgcn_s_memrealtime();  // (100 MHz: with the first and last reading, the clock)
    if (slot == 30) q[31] = __builtin_amdgcn_s_memrealtime() - q[31];
  }
#endif
}

// What every thread knows about its place.
struct Lane {
  int lane, l15, g, wave, r0, nrows;
  __device__ __forceinline__ void init(int T) {
    lane = threadIdx.x & 63; l15 = lane & 15; g = lane >> 4;
    wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    r0 = blockIdx.x * R;
    nrows = min(R, T - r0);  // rows of this panel that exist
  }
};

// Packed 16-bit operations by name: written as vector arithmetic the compiler takes them apart again (a compare and
// a select per element where one packed minimum does the job).
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  uint32_t d;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi));
  return d;
}
__device__ __forceinline__ uint32_t pk_max_i16_zero(uint32_t a) {  // ReLU of a bf16 pair: negative values have the top bit set
  uint32_t d;
  asm("v_pk_max_i16 %0, %1, 0" : "=v"(d) : "v"(a));
  return d;
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
  uint32_t d;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ uint32_t pk_neg_u16(uint32_t a) {  // 0 / 1 per half -> 0 / 0xFFFF
  uint32_t d;
  asm("v_pk_sub_u16 %0, 0, %1" : "=v"(d) : "v"(a));
  return d;
}

// 16 fp32 values -> 8 dwords of bf16 pairs (element 2 k in the low half of dword k).
__device__ __forceinline__ void pack16(const f32x4_t (&a)[4], uint32_t (&d)[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) d[k] = cvt_pk_bf16(a[k >> 1][2 * (k & 1)], a[k >> 1][2 * (k & 1) + 1]);
}

// The ReLU pattern in the row-panel kernels' own layout (written by E_RELU_BITS, read by E_MASK_BITS of a launch
// with the same T and N): one 32-bit word per lane, row group and 64-column group,
//   word[((workgroup * NMI + mi) * (N / 64) + column group) * 64 + lane],
// bit 2 k = element 2 k of the lane's 16 columns is positive, bit 2 k + 16 = element 2 k + 1: what falls out of the
// packed bf16 pairs without moving bits across lanes, in 256-byte pieces per store instruction.
__device__ __forceinline__ uint32_t pattern_index(const Params& p, const Lane& L, int mi, int cbase) {
  return (((uint32_t)blockIdx.x * NMI + mi) * (uint32_t)(p.N >> 6) + (uint32_t)(cbase >> 6)) * 64u + L.lane;
}

// acc[mi][ni][r]: row 16 mi + l15 of the panel, column col0 + 4 ni + r with col0 = cbase + 16 g (cbase: the first
// column of the wave's 64-column group); the bias is in the accumulators already (they start from it).  Epilogues on the packed bf16 pairs: ReLU as a signed 16-bit
// maximum with zero, its pattern as a 16-bit minimum with one, the backward mask as an AND.
template <int EPI>
__device__ __forceinline__ void store_rows(const Params& p, const Lane& L, f32x4_t (&acc)[NMI][4], int cbase) {
  const int col = cbase + 16 * L.g;
  // byte offsets of this lane's piece in row group 0 (32 bits: the host checks the sizes), made opaque so that they are
  // derived here each time and not kept as seven 64-bit row addresses across the K loop
  uint32_t rl = (uint32_t)(L.r0 + L.l15);
  asm volatile("" : "+v"(rl));
  const uint32_t coff = (rl * (uint32_t)p.ldc + (uint32_t)col) * 2u, cstep = 32u * (uint32_t)p.ldc;
  const uint32_t cfirst = ((uint32_t)L.r0 * (uint32_t)p.ldc + (uint32_t)col) * 2u;
  uint4 old[NMI][2];
  uint32_t pat[NMI];
  uint32_t ones = 0x00010001u;
  asm volatile("" : "+v"(ones));
  if (EPI == E_ACCUM) {
#pragma unroll
    for (int mi = 0; mi < NMI; ++mi) {  // (rows past the end read the panel's first row: never used)
      const char* src = (const char*)p.C + (16 * mi + L.l15 < L.nrows ? coff + mi * cstep : cfirst);
      old[mi][0] = *(const uint4*)src;
      old[mi][1] = *(const uint4*)(src + 16);
    }
  }
  if (EPI == E_MASK_BITS) {
#pragma unroll
    for (int mi = 0; mi < NMI; ++mi) pat[mi] = ((const uint32_t*)p.signs)[pattern_index(p, L, mi, cbase)];
  }
#pragma unroll
  for (int mi = 0; mi < NMI; ++mi) {
    const int row = 16 * mi + L.l15;
    if (EPI == E_ACCUM) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bf16_t* e = (const bf16_t*)&old[mi][h];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[mi][(8 * h + j) >> 2][j & 3] += bf16_to_f32(e[j]);
      }
    }
    uint32_t d[8];
    pack16(acc[mi], d);
    if (EPI == E_RELU_BITS) {
      uint32_t r = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        d[k] = pk_max_i16_zero(d[k]);
        r |= pk_min_u16(d[k], ones) << (2 * k);  // 0 / 1 per half: the pair's two pattern bits
      }
      ((uint32_t*)p.signs)[pattern_index(p, L, mi, cbase)] = r;
    }
    if (EPI == E_MASK_BITS) {
#pragma unroll
      for (int k = 0; k < 8; ++k) d[k] &= pk_neg_u16((pat[mi] >> (2 * k)) & ones);
    }
    if (row < L.nrows) {
      char* dst = (char*)p.C + (coff + mi * cstep);
      *(uint4*)dst = make_uint4(d[0], d[1], d[2], d[3]);
      *(uint4*)(dst + 16) = make_uint4(d[4], d[5], d[6], d[7]);
    }
  }
}

// Accumulators of a column group start from the bias of the lane's 16 columns (or zero): load_bias is called ahead of time
// (before the epilogue of the column group before), init_acc when the group starts.
template <bool BIAS>
__device__ __forceinline__ void load_bias(const Params& p, const Lane& L, f32x4_t (&b)[4], int cbase) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (BIAS) {
      const float4 v = *(const float4*)(p.bias + cbase + 16 * L.g + 4 * q);
      b[q] = f32x4_t{v.x, v.y, v.z, v.w};
    } else {
      b[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
  }
}
__device__ __forceinline__ void init_acc(f32x4_t (&acc)[NMI][4], const f32x4_t (&b)[4]) {
#pragma unroll
  for (int i = 0; i < NMI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = b[j];
}

// The sub-layer tail on full rows: sub-layer output = acc + bias (stored, rounded to bf16 as the separate kernels
// see it; the accumulators started from the bias), y = LayerNorm(res + dropout(sub-layer output)) gamma + beta with
// pad rows zeroed; row statistics in two
// passes (mean, then centred squares) through LDS partial sums [row][wave].
__device__ __forceinline__ void ln_rows(const Params& p, const Lane& L, f32x4_t (&acc)[NMI][4], float* part1, float* part2) {
  const int col = 64 * L.wave + 16 * L.g;
  const SkSeed sd = sk_seed(p.seed);
  uint4 resv[NMI][2];
#pragma unroll
  for (int mi = 0; mi < NMI; ++mi) {
    const int64_t grow = L.r0 + min(16 * mi + L.l15, L.nrows - 1);
    const bf16_t* src = p.res + grow * p.ldres + col;
    resv[mi][0] = *(const uint4*)src;
    resv[mi][1] = *(const uint4*)(src + 8);
  }
#pragma unroll
  for (int mi = 0; mi < NMI; ++mi) {
    const int row = 16 * mi + L.l15;
    const int64_t grow = L.r0 + min(row, L.nrows - 1);
    float v[2][8];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[ni >> 1][4 * (ni & 1) + r] = acc[mi][ni][r];
    const uint4 x0 = pack8(v[0]), x1 = pack8(v[1]);
    if (row < L.nrows) {
      bf16_t* dst = p.C + grow * p.ldc + col;
      *(uint4*)dst = x0;
      *(uint4*)(dst + 8) = x1;
    }
    const uint64_t base = (uint64_t)(grow * D + col);  // even: element pairs share a hash
    float sum = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bf16_t* xe = (const bf16_t*)(h ? &x1 : &x0);
      const bf16_t* re = (const bf16_t*)&resv[mi][h];
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        bool k0 = true, k1 = true;
        if (p.drop_thr) {
          const uint32_t rnd = sk_hash_pair(sd, (base >> 1) + 4 * h + jp);
          k0 = sk_keep_half(rnd, 0, p.drop_thr);
          k1 = sk_keep_half(rnd, 1, p.drop_thr);
        }
        const float z0 = bf16_to_f32(re[2 * jp]) + (k0 ? bf16_to_f32(xe[2 * jp]) * p.keep_scale : 0.f);
        const float z1 = bf16_to_f32(re[2 * jp + 1]) + (k1 ? bf16_to_f32(xe[2 * jp + 1]) * p.keep_scale : 0.f);
        const int j = 8 * h + 2 * jp;
        acc[mi][j >> 2][j & 3] = z0;
        acc[mi][(j + 1) >> 2][(j + 1) & 3] = z1;
        sum += z0 + z1;
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (L.g == 0 && row < R) part1[row * 8 + L.wave] = sum;
  }
  lds_barrier();
  float mean[NMI];
#pragma unroll
  for (int mi = 0; mi < NMI; ++mi) {
    const int row = min(16 * mi + L.l15, R - 1);
    const float4 a = *(const float4*)(part1 + row * 8), b = *(const float4*)(part1 + row * 8 + 4);
    mean[mi] = (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float cdev = acc[mi][ni][r] - mean[mi]; sq += cdev * cdev; }
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    if (L.g == 0 && 16 * mi + L.l15 < R) part2[row * 8 + L.wave] = sq;
  }
  lds_barrier();
  float gam[16], bet[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 a = *(const float4*)(p.gamma + col + 4 * q), b = *(const float4*)(p.beta + col + 4 * q);
    gam[4 * q] = a.x; gam[4 * q + 1] = a.y; gam[4 * q + 2] = a.z; gam[4 * q + 3] = a.w;
    bet[4 * q] = b.x; bet[4 * q + 1] = b.y; bet[4 * q + 2] = b.z; bet[4 * q + 3] = b.w;
  }
#pragma unroll
  for (int mi = 0; mi < NMI; ++mi) {
    const int row = 16 * mi + L.l15;
    const int rr = min(row, R - 1);
    const float4 a = *(const float4*)(part2 + rr * 8), b = *(const float4*)(part2 + rr * 8 + 4);
    const float var = (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) / (float)D;
    const float rstd = 1.0f / sqrtf(var + p.eps);
    if (row >= L.nrows) continue;
    const bool pad = p.row_pad && p.row_pad[L.r0 + row];
    if (L.wave == 0 && L.g == 0) { p.mean[L.r0 + row] = mean[mi]; p.rstd[L.r0 + row] = rstd; }
    float o[2][8];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float val = (acc[mi][j >> 2][j & 3] - mean[mi]) * rstd * gam[j] + bet[j];
      o[j >> 3][j & 7] = pad ? 0.f : val;
    }
    bf16_t* dst = p.Y + (int64_t)(L.r0 + row) * D + col;
    *(uint4*)dst = pack8(o[0]);
    *(uint4*)(dst + 8) = pack8(o[1]);
  }
}

// ---------------------------------------------------------------------------------------- K = 512, panel resident
template <int EPI, bool BIAS>
__global__ __launch_bounds__(512, 1) void panel_kernel(Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Lane L;
  L.init(p.T);
  stamp(p, L.wave, L.lane, 0);
  char* Xs = smem;
  // ---- this wave's column groups.  SIMD s (waves s and s + 4: a workgroup's waves go round the SIMDs) takes the
  // groups s, s + 4, ..; of those n = N / 256 the first wave takes the first ceil(5 n / 8), the second the rest.
  // The arbiter prefers the older wave, so the first wave runs its K loops at nearly full rate and the second fills
  // the gaps and the first one's epilogues; with equal shares the second is left to finish two groups alone, reading,
  // waiting and multiplying in turn.  (The sub-layer tail needs whole rows: one group per wave, group = wave.)
  const int per_simd = EPI == E_LN ? 2 : p.N / 256;
  const int n_first = EPI == E_LN ? 1 : (5 * per_simd + 4) / 8;
  const int item0 = L.wave < 4 ? 0 : n_first, nitems = L.wave < 4 ? n_first : per_simd - n_first;
  auto group_of = [&](int i) { return (L.wave & 3) + 4 * (item0 + min(i, nitems - 1)); };

  // ---- weight stream: a UNIT is two fragments (2 KB); a K-step takes two units; unit v of an item, v = 0 .. 31
  // (32 .. : the next item's; past the last item the last one's are fetched again and never used: no branch around
  // a load).  Wave-uniform byte address plus one 32-bit lane offset; a column group is 16 K-steps of 4 KB.
  const uint32_t lane_off = L.lane * 16;
  bf16x8_t bq[4][2];  // four units in flight
  auto issue = [&](int slot, int i, int v) {
    if (v >= 32) { v -= 32; i += 1; }
    const char* src = (const char*)p.W + (int64_t)group_of(i) * (16 * 4096) + v * 2048;
    bq[slot][0] = *(const bf16x8_t*)(src + lane_off);
    bq[slot][1] = *(const bf16x8_t*)(src + 1024 + lane_off);
  };

  // ---- the panel: one direct-to-LDS instruction per row (64 lanes x 16 bytes = the row's 1 KB), then the first four
  // weight units, which stay in flight across the wait for the panel
#pragma unroll
  for (int i = 0; i < R / 8; ++i) {
    const int row = L.wave + 8 * i;
    const int c = (L.lane & ~15) | ((L.lane ^ row) & 15);  // the logical chunk stored at chunk position `lane`
    const bf16_t* src = p.A + (int64_t)(L.r0 + min(row, L.nrows - 1)) * p.lda + 8 * c;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(Xs + row * XROW), 16, 0, 0);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) issue(s, 0, s);
  // the wait for the panel is the compiler's (a read of the rows this wave requested makes it count the eight
  // weight loads behind them): a wait it does not know of leaves it assuming that the transfers are still under
  // way at the head of the column-group loop, where it then drains the weight queue every time round
  {
    int probe = 0;
#pragma unroll
    for (int i = 0; i < R / 8; ++i) probe |= *(volatile const int*)(Xs + (L.wave + 8 * i) * XROW + L.lane * 4);
    asm volatile("" :: "v"(probe));
  }
  lds_barrier();

  // per-lane fragment addressing: chunk 4 ks + g of row 16 mi + l15
  int offs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) offs[j] = ((4 * j + L.g) ^ L.l15) << 4;
  const char* xa = Xs + L.l15 * XROW;
  auto read_x = [&](int ks, int mi) {
    return *(const bf16x8_t*)(xa + mi * 16 * XROW + (ks >> 2) * 256 + offs[ks & 3]);
  };
  f32x4_t bias[4];
  load_bias<BIAS>(p, L, bias, 64 * group_of(0));
  bf16x8_t ring[4];
  ring[0] = read_x(0, 0);
  ring[1] = read_x(0, 1);
  ring[2] = read_x(0, 2);
  stamp(p, L.wave, L.lane, 1);

  for (int i = 0; i < nitems; ++i) {
    const int cbase = 64 * group_of(i);
    f32x4_t acc[NMI][4];
    init_acc(acc, bias);
    __builtin_amdgcn_sched_barrier(0);
    // Activation fragments are requested three (row group, K-step) items ahead of the MFMAs that use them, through
    // a ring of four register sets: a wave that reads, waits and multiplies in turn leaves the matrix pipe idle a
    // third of the time whenever its partner on the SIMD is not there to fill in (it is in its epilogue, or has
    // finished).  Item it = (K-step it / 7, row group it % 7); a column group is 112 items, and the items past its
    // end are the first ones of the next group (the same panel).
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int s0 = (2 * ks) & 3;
#pragma unroll
      for (int mi = 0; mi < NMI; ++mi) {
        const int it = ks * NMI + mi, nx = (it + 3) % (16 * NMI);
        ring[(it + 3) & 3] = read_x(nx / NMI, nx % NMI);
        const bf16x8_t a = ring[it & 3];
        acc[mi][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[s0][0], a, acc[mi][0], 0, 0, 0);
        acc[mi][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[s0][1], a, acc[mi][1], 0, 0, 0);
        acc[mi][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[s0 + 1][0], a, acc[mi][2], 0, 0, 0);
        acc[mi][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[s0 + 1][1], a, acc[mi][3], 0, 0, 0);
        if (mi == NMI - 1) {
          issue(s0, i, 2 * ks + 4);
          issue(s0 + 1, i, 2 * ks + 5);
        }
        __builtin_amdgcn_sched_barrier(0);  // program order is the schedule: nothing moves across an item
      }
    }
    stamp(p, L.wave, L.lane, 2 + 2 * i);
    if constexpr (EPI == E_LN) {
      ln_rows(p, L, acc, (float*)(smem + X_BYTES), (float*)(smem + X_BYTES + PART_BYTES));
    } else {
      load_bias<BIAS>(p, L, bias, 64 * group_of(i + 1));  // the next group's, under the epilogue
      store_rows<EPI>(p, L, acc, cbase);
    }
    stamp(p, L.wave, L.lane, 3 + 2 * i);
  }
  stamp(p, L.wave, L.lane, 30);
}

template <typename K>
int launch(K kernel, int id, const Params& p, hipStream_t stream, const char* who) {
  static bool allowed[16] = {};  // per kernel instantiation (id)
  if (!allowed[id]) {
    int rc = sk_allow_big_lds(kernel, LDS_BYTES);
    if (rc) return rc;
    allowed[id] = true;
  }
  const int blocks = (p.T + R - 1) / R;
  hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(512), LDS_BYTES, stream, p);
  SK_CHECK_LAUNCH(who);
  return SK_OK;
}

bool aligned16(const void* q) { return ((uintptr_t)q & 15) == 0; }

unsigned long long* g_stamps = nullptr;  // diagnostics buffer (skhip_rowpanel_set_stamps)

// Common argument checks; fills the shape fields of p.
int check_common(const char* who, Params& p, const void* a, int64_t lda, const void* w_fm, const float* bias, void* c,
                 int64_t ldc, int64_t T, int64_t N, int64_t K) {
  SK_CHECK_ARG(T > 0 && N > 0 && K > 0 && N % rp::D == 0, "%s: T=%lld N=%lld K=%lld (N must be a multiple of 512)", who,
               (long long)T, (long long)N, (long long)K);
  SK_CHECK_ARG(a && w_fm && c, "%s: null pointer", who);
  SK_CHECK_ARG(lda % 8 == 0 && lda >= K && ldc % 8 == 0 && ldc >= N && aligned16(a) && aligned16(w_fm) && aligned16(c) &&
                   aligned16(bias),
               "%s: operands must be 16-byte aligned, row strides multiples of 8 elements", who);
  // the kernels address rows with 32-bit byte offsets
  SK_CHECK_ARG((T + 16) * std::max(std::max(lda, ldc), N) * 2 < ((int64_t)1 << 32), "%s: operand too large for 32-bit offsets", who);
  p.A = (const bf16_t*)a; p.lda = lda;
  p.W = (const bf16_t*)w_fm; p.bias = bias;
  p.C = (bf16_t*)c; p.ldc = ldc;
  p.T = (int)T; p.N = (int)N; p.K = (int)K;
  p.stamps = g_stamps;
  return SK_OK;
}

int fill_ln(const char* who, Params& p, const void* res, int64_t ldres, const float* gamma, const float* beta,
            const uint8_t* row_pad, void* y, float* mean, float* rstd, float eps, float p_drop, uint64_t seed) {
  SK_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "%s: p_drop=%f", who, p_drop);
  SK_CHECK_ARG(res && gamma && beta && y && mean && rstd && p.bias, "%s: null pointer", who);
  SK_CHECK_ARG(ldres % 8 == 0 && ldres >= rp::D && aligned16(res) && aligned16(y) && aligned16(gamma) && aligned16(beta),
               "%s: operands must be 16-byte aligned, row strides multiples of 8 elements", who);
  p.res = (const bf16_t*)res; p.ldres = ldres;
  p.gamma = gamma; p.beta = beta; p.row_pad = row_pad;
  p.Y = (bf16_t*)y; p.mean = mean; p.rstd = rstd;
  p.eps = eps; p.drop_thr = sk_drop_threshold(p_drop); p.keep_scale = 1.0f / (1.0f - p_drop); p.seed = seed;
  return SK_OK;
}

}  // namespace rp
}  // namespace

extern "C" int skhip_rowpanel_rows(void) { return rp::R; }

extern "C" int64_t skhip_rowpanel_pattern_bytes(int64_t T, int64_t N) {
  return ((T + rp::R - 1) / rp::R) * rp::NMI * N * 4;
}

extern "C" int skhip_rowpanel_set_stamps(void* buffer) {
  rp::g_stamps = (unsigned long long*)buffer;
  return SK_OK;
}

extern "C" int skhip_rowpanel_gemm(const void* a, int64_t lda, const void* w_fm, const float* bias, void* c, int64_t ldc,
                                   void* signs, int64_t T, int64_t N, int epilogue, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (T == 0) return SK_OK;
  rp::Params p{};
  int rc = rp::check_common("skhip_rowpanel_gemm", p, a, lda, w_fm, bias, c, ldc, T, N, rp::D);
  if (rc) return rc;
  p.signs = signs;
  const bool has_bias = bias != nullptr;
  SK_CHECK_ARG(has_bias == ((epilogue & SK_EPI_BIAS) != 0), "skhip_rowpanel_gemm: SK_EPI_BIAS and the bias pointer disagree");
  switch (epilogue & ~SK_EPI_BIAS) {
    case 0:
      if (has_bias) return rp::launch(rp::panel_kernel<rp::E_STORE, true>, 0, p, stream, "skhip_rowpanel_gemm");
      return rp::launch(rp::panel_kernel<rp::E_STORE, false>, 1, p, stream, "skhip_rowpanel_gemm");
    case SK_EPI_RELU | SK_EPI_RELU_BITS_OUT:
      SK_CHECK_ARG(has_bias && signs && rp::aligned16(signs), "skhip_rowpanel_gemm: the ReLU form takes a bias and a pattern buffer");
      return rp::launch(rp::panel_kernel<rp::E_RELU_BITS, true>, 2, p, stream, "skhip_rowpanel_gemm");
    case SK_EPI_RELU_MASK_BITS:
      SK_CHECK_ARG(!has_bias && signs && rp::aligned16(signs), "skhip_rowpanel_gemm: the masked form takes a pattern buffer and no bias");
      return rp::launch(rp::panel_kernel<rp::E_MASK_BITS, false>, 3, p, stream, "skhip_rowpanel_gemm");
    case SK_EPI_ACCUM:
      SK_CHECK_ARG(!has_bias, "skhip_rowpanel_gemm: the accumulating form takes no bias");
      return rp::launch(rp::panel_kernel<rp::E_ACCUM, false>, 4, p, stream, "skhip_rowpanel_gemm");
    default:
      sk_set_error("skhip_rowpanel_gemm: unsupported epilogue %d", epilogue);
      return SK_ERR_ARG;
  }
}

extern "C" int skhip_rowpanel_gemm_ln(const void* a, int64_t lda, const void* w_fm, const float* bias, const void* res,
                                      int64_t ldres, void* sub_out, const float* gamma, const float* beta,
                                      const uint8_t* row_pad, void* y, float* mean, float* rstd, int64_t T, float eps,
                                      float p_drop, uint64_t seed, void* stream_) {
  if (T == 0) return SK_OK;
  rp::Params p{};
  int rc = rp::check_common("skhip_rowpanel_gemm_ln", p, a, lda, w_fm, bias, sub_out, rp::D, T, rp::D, rp::D);
  if (rc) return rc;
  rc = rp::fill_ln("skhip_rowpanel_gemm_ln", p, res, ldres, gamma, beta, row_pad, y, mean, rstd, eps, p_drop, seed);
  if (rc) return rc;
  return rp::launch(rp::panel_kernel<rp::E_LN, true>, 5, p, (hipStream_t)stream_, "skhip_rowpanel_gemm_ln");
}
