// One decode step of the whole decoder stack in ONE launch (bf16; d = 512, 8 heads of 64, d_inner = 2048).
//
// A beam-search step runs the decoder on the newest position of every beam row only
// (lib/transformer/Translator.py:121-140 re-runs the whole prefix; the incremental form keeps key / value
// caches), and for one position every operation of the stack is ROW-LOCAL: projections, attention over the row's
// own history and its sentence's encoder output, residuals, LayerNorms, feed-forward.  Launched op by op that is
// about a hundred dependent launches of a few microseconds each, none of which can use the chip.
//
// Here a TEAM of 8 workgroups owns 32 rows for the whole stack.  Every member holds the 32 current activation
// rows in LDS; a projection is split over the members by output columns (member m: head m of q / k / v and of
// the attention, columns 64 m .. 64 m + 63 of the 512-wide outputs, columns 256 m .. of the hidden layer), so a
// member streams one eighth of every weight matrix — straight from global memory into MFMA operand registers,
// requested while the member waits at the team barrier before their use — and the members exchange their 32-row
// slices through global memory with a team barrier (five per layer).  The second feed-forward product is split
// over the members along K instead: a member multiplies its own slice of the hidden layer, which never leaves its
// LDS, and every member adds the eight partial products up inside the LayerNorm that follows.  Teams never wait
// for each other.  All summation orders are fixed: results are bit-reproducible.
//
// Team barrier: every wave waits for its stores (they are then in the L2), workgroup barrier, one atomic
// increment, the next phase's weight requests, a BOUNDED spin on the counter (a member that gives up raises the error word and stops waiting from
// then on, so every wave reaches the end of the kernel whatever happens), workgroup barrier, invalidation of the
// CU's L1.  The members of a team have the same block index modulo 8, which the dispatcher places on ONE XCD:
// they share an L2, so the exchange needs no L2 write-back or invalidation (agent-scope fences cost 20-80 us
// per barrier here: 160 workgroups flushing eight L2s thirty-six times a step).  The exchange is only correct under
// that placement (the counter's agent-scope atomics are served past the L2 and would advance all the same, while
// the slices of a member on another XCD would be read stale), so the kernel VERIFIES it: every member ORs the bit
// of the XCC it runs on (hardware register XCC_ID) into a word of its team before the first barrier, and the
// team's leaders look at the word after it; more than one bit raises the error word (value 2) and the caller
// must discard the step.  Members of a team are
// 8 workgroups of one launch of at most one workgroup per CU (the host checks teams x 8 <= CUs), so all of
// them are resident together.  The counters alternate between two sets by step parity: a launch zeroes the set
// of the other parity, which nothing touches meanwhile.
#include "common.h"
#include <stdlib.h>
#include <algorithm>

namespace {
constexpr int D = 512, H = 8, DH = 64, F = 2048, TEAM = 8, ROWS = 32, NTHREADS = 512;
constexpr int MAX_LAYERS = 8, LAYER_FIELDS = 22, MAX_TEAMS = 32;
constexpr int SPIN_LIMIT = 1 << 18;
constexpr int XCC_WORD0 = 192;   // first of the [2][MAX_TEAMS] XCC-mask words in the workspace head (the diagnostics' time stamps sit at words 128 .. 161)

struct LayerPtrs {
  const bf16_t* Wqkv; const float* bqkv;              // rows q | k | v of [1536][512], fragment-major (see load_w); [1536]
  const bf16_t* Wo;   const float* bo;   const float* g1; const float* be1;
  const bf16_t* Wqc;  const float* bqc;
  const bf16_t* Woc;  const float* boc;  const float* g2; const float* be2;
  const bf16_t* W1;   const float* b1;                // [2048][512]
  const bf16_t* W2;   const float* b2;   const float* g3; const float* be3;   // [512][2048]
  bf16_t* Kc; bf16_t* Vc;                             // [R][Tc][512]
  const bf16_t* crossK; const bf16_t* crossV;         // [n_sent][Ls][512]
};

struct StackArgs {
  LayerPtrs layer[MAX_LAYERS];
  const int64_t* tok; const int64_t* pos; const float* word_emb; const float* pos_tab;
  const int32_t* cur_len; const uint8_t* row_pad; const uint8_t* pad_hist; const int32_t* hist;
  const uint8_t* src_key_pad;
  bf16_t* dec_out;  // the stack's output rows, [R][512] or fragment-major (out_fm)
  int out_fm;
  int32_t* slots;   // [2][MAX_TEAMS][16] arrival words (slot form of the team barrier) or null
  int scalar_poll;  // poll the arrival words with scalar loads
  int32_t* flags;   // [2][MAX_TEAMS] team counters, one error word (index 2 MAX_TEAMS), [2][MAX_TEAMS] XCC masks from index XCC_WORD0
  bf16_t* att;      // [R32][512]
  float* zbuf;      // [R32][512]
  float* pbuf;      // [8][R32][512]
  int n_layers, R, R32, bm, Tc, Ls, parity, teams_pad, rpt;   // rpt: rows per team (<= 32)
  int fuse_red;     // the partial feed-forward products summed by every member (1) or by columns with one more team barrier (0)
  int TcP;          // row length of the history table in LDS (Tc rounded up to 32), 0: self-attention by attend()
  int xshare;       // encoder attention per (sentence, segment) with the beams sharing the loads (1) or row by row (0)
  int stamp_block, stamp_tid;   // who writes the stamps
  long long* stamps;   // diagnostics (SKHIP_STACK_STAMPS=1): 100 MHz clock at the phase boundaries of layer 3, team 0 member 0
  float scale, eps;
};

// LDS map (bytes)
constexpr int L_X = 0;                    // [32][512] bf16, 16-byte chunk c of row r at chunk position c ^ (r & 15)
constexpr int L_A = 32768;                // same layout: attention rows gathered from the team
constexpr int L_HS = 65536;               // [32][256] bf16 hidden slice, same swizzle
constexpr int L_QKV = 81920;              // q | k | v [3][32][64] bf16, plain rows
constexpr int L_BYTES = 94208;
constexpr int L_JT = L_BYTES;             // [32][TcP] history table of self_attend_pipe (when the launch carries it)
constexpr int L_MAX = 160 * 1024;
// encoder attention by (sentence, 32-position segment): a segment's un-normalised P.V sums [XSEG_MAX][ROWS][64] fp32 and
// (maximum, sum) pairs [XSEG_MAX][ROWS][2] lie over the A and HS images, which nobody uses during that phase
constexpr int XSEG = 32, XSEG_MAX = 5;    // source sentences of at most 160 positions take this path
constexpr int L_XACC = L_A, L_XML = L_A + XSEG_MAX * ROWS * DH * 4;
static_assert(L_XML + XSEG_MAX * ROWS * 2 * 4 <= L_QKV, "segment results must fit over the A and HS images");

struct TeamCtx {
  int32_t* flag;    // the team's arrival counter (counter form)
  int32_t* slots;   // the team's eight arrival words, one 64-byte line (slot form); null: counter form
  int32_t* err;
  int count;   // barriers passed
  int member;
  bool dead;
  bool scalar_poll;
};

// prefetch(): loads of the NEXT phase's weights, issued by every wave between announcing its arrival and waiting for
// the others, so that their trip to memory runs beside the barrier's own round trips.  (Issued before the barrier,
// they sit in front of the wait for the stores, and the team meets two microseconds later.)
// Arrival, slot form: a member STORES the number of the barrier into its own word of the team's line and polls the
// line until all eight words have reached it — no atomic whose result has to come back before the first poll can
// be issued (one L2 round trip less per barrier than the counter form, which the 30 barriers of a step feel).
template <class F>
__device__ __forceinline__ void team_sync(TeamCtx& c, F&& prefetch) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's stores have reached the L2
  __syncthreads();
  c.count += 1;
  const bool leader = threadIdx.x == 0 && !c.dead;
  if (c.slots) {
    if (leader) __hip_atomic_store(c.slots + c.member, c.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prefetch();
    if (leader) {
      int spins = 0;
      while (true) {
        int lo = c.count;
        if (c.scalar_poll) {
          // The eight words by ONE scalar load that bypasses the scalar cache (glc): scalar loads have a counter of
          // their own, so the poll does not queue behind the weight prefetch that this wave has just issued (vector
          // loads return in the order of their issue: a vector poll waited out the whole prefetch).
          typedef uint32_t u32x8_t __attribute__((ext_vector_type(8)));
          u32x8_t w;
          const uint64_t pa = (uint64_t)c.slots;
          const uint32_t plo = __builtin_amdgcn_readfirstlane((uint32_t)pa), phi = __builtin_amdgcn_readfirstlane((uint32_t)(pa >> 32));
          const uint64_t sp = ((uint64_t)phi << 32) | plo;
          asm volatile("s_load_dwordx8 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(sp) : "memory");
#pragma unroll
          for (int i = 0; i < TEAM; ++i) lo = min(lo, (int)w[i]);
        } else {
#pragma unroll
          for (int i = 0; i < TEAM; ++i) lo = min(lo, __hip_atomic_load(c.slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        if (lo >= c.count) break;
        if (++spins > SPIN_LIMIT) {
          __hip_atomic_store(c.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          c.dead = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
  } else {
    const int target = TEAM * c.count;
    int seen = target;
    // the value the increment returns already tells the last arriver that everybody is here: no poll for it
    if (leader) seen = __hip_atomic_fetch_add(c.flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    prefetch();
    if (leader) {
      int spins = 0;
      while (seen < target) {
        __builtin_amdgcn_s_sleep(1);
        seen = __hip_atomic_load(c.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (++spins > SPIN_LIMIT) {
          __hip_atomic_store(c.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          c.dead = true;
          break;
        }
      }
    }
  }
  __syncthreads();
  asm volatile("buffer_inv sc0" ::: "memory");   // forget this CU's L1 lines: the team's slices are read from the L2
}
__device__ __forceinline__ void team_sync(TeamCtx& c) { team_sync(c, [] {}); }

// ---- MFMA pieces.  v_mfma_f32_16x16x32_bf16(w, a): lane (l15 = lane & 15, g = lane >> 4) supplies k = 8 g .. 8 g + 7
// of weight row l15 and of activation row l15, and receives out[activation row l15][weight rows 4 g .. 4 g + 3].
template <int NK>
struct WFrag {
  bf16x8_t w[NK];
};

// Weight matrices are stored FRAGMENT-MAJOR, [rows / 16][K / 32][64 lanes][8]: the 8 elements lane (l15, g) needs for
// row tile nt and K-step kt at ((nt * K / 32 + kt) * 64 + l15 + 16 g) * 8, so that one load instruction of a wave covers
// 1 KB of consecutive bytes (whole 128-byte lines).  From the matrices' own [rows][K] layout an instruction asks for
// sixteen 64-byte pieces a row apart, and a CU gets such requests served at half the rate: a member streams 0.9 MB of
// weights per layer, and the step took 0.78 instead of 0.71 ms.  ldw: K of the matrix; n_row: first row of the tile + l15.
template <int NK>
__device__ __forceinline__ void load_w(WFrag<NK>& f, const bf16_t* W, int64_t ldw, int n_row, int k0, int g) {
  const int l15 = n_row & 15;   // (row tiles start at multiples of 16)
  const bf16_t* p = W + (((int64_t)(n_row >> 4) * (ldw >> 5) + (k0 >> 5)) * 64 + l15 + 16 * g) * 8;
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) f.w[kk] = *(const bf16x8_t*)(p + 512 * kk);
}

// acc[mi] += W-tile x A rows (16 mi + l15) over NK K-steps; A image in LDS with `row_bytes` per row, first chunk kc0.
template <int NK>
__device__ __forceinline__ void mma_rows(f32x4_t (&acc)[2], const WFrag<NK>& f, const char* A, int row_bytes, int kc0,
                                         int l15, int g) {
  const char* a0 = A + l15 * row_bytes;
  const char* a1 = A + (16 + l15) * row_bytes;
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const int off = ((kc0 + 4 * kk + g) ^ l15) << 4;
    const bf16x8_t fa0 = *(const bf16x8_t*)(a0 + off);
    const bf16x8_t fa1 = *(const bf16x8_t*)(a1 + off);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w[kk], fa0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w[kk], fa1, acc[1], 0, 0, 0);
    if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep the fragment reads of later K-steps out of the registers
  }
}

// The same over NT weight tiles at once: an activation fragment is read from LDS once and multiplies all NT tiles.
// (Tile by tile every tile re-reads the whole image, and with eight waves at it the LDS port — 128 bytes a clock —
// sets the pace: four tiles of the [32][256] hidden slice per wave are 512 KB a phase, 2 us.)
// every4(kk), kk = 3, 7, ..: a place for requests that use the registers of the weight fragments consumed so far.
template <int NT, int NK, class H>
__device__ __forceinline__ void mma_rows_tiles(f32x4_t (&acc)[NT][2], const WFrag<NK> (&f)[NT], const char* A, int row_bytes,
                                               int kc0, int l15, int g, H&& every4) {
  const char* a0 = A + l15 * row_bytes;
  const char* a1 = A + (16 + l15) * row_bytes;
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const int off = ((kc0 + 4 * kk + g) ^ l15) << 4;
    const bf16x8_t fa0 = *(const bf16x8_t*)(a0 + off);
    const bf16x8_t fa1 = *(const bf16x8_t*)(a1 + off);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[t].w[kk], fa0, acc[t][0], 0, 0, 0);
      acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[t].w[kk], fa1, acc[t][1], 0, 0, 0);
    }
    if ((kk & 3) == 3) {
      __builtin_amdgcn_sched_barrier(0);
      every4(kk);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
template <int NT, int NK>
__device__ __forceinline__ void mma_rows_tiles(f32x4_t (&acc)[NT][2], const WFrag<NK> (&f)[NT], const char* A, int row_bytes,
                                               int kc0, int l15, int g) {
  mma_rows_tiles<NT, NK>(acc, f, A, row_bytes, kc0, l15, g, [](int) {});
}

__device__ __forceinline__ float round_bf16(float v) { return bf16_to_f32(f32_to_bf16(v)); }

// four consecutive bf16 of the swizzled [32][512] image: row r, columns col .. col + 3 (col % 4 == 0)
__device__ __forceinline__ void x_read4(const char* X, int r, int col, float (&v)[4]) {
  const uint2 raw = *(const uint2*)(X + r * 1024 + (((col >> 3) ^ (r & 15)) << 4) + (col & 7) * 2);
  v[0] = bf16_to_f32((bf16_t)(raw.x & 0xffff));
  v[1] = bf16_to_f32((bf16_t)(raw.x >> 16));
  v[2] = bf16_to_f32((bf16_t)(raw.y & 0xffff));
  v[3] = bf16_to_f32((bf16_t)(raw.y >> 16));
}

__device__ __forceinline__ uint2 pack4(const float (&v)[4]) {
  uint2 o;
  o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
  o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
  return o;
}

// ---- LayerNorm of the team's rows of the fp32 exchange buffer into the LDS image X (wave w: rows w, w + 8, w + 16,
// w + 24 — with 20 rows per team every wave then has two or three of them instead of five waves four each; a lane
// owns 8 consecutive columns; same arithmetic as add_ln_fwd_kernel of layernorm.hip).  Rows beyond the team's last
// one are left as they are in X: the products run over all 32 image rows and their results are never stored.
// after_loads(): requests of the phase that may trail the rows' own (a prefetch for later in the phase).
template <class F>
__device__ __forceinline__ void ln_rows(char* X, const float* z, int row0, int R, const float* gamma, const float* beta,
                                        const uint8_t* row_pad, float eps, int wave, int lane, F&& after_loads) {
  float4 zz[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {   // all rows' loads in flight together
    const float4* zp = (const float4*)(z + (int64_t)min(row0 + wave + 8 * i, R - 1) * D + 8 * lane);
    zz[i][0] = zp[0];
    zz[i][1] = zp[1];
  }
  const float4* gp = (const float4*)(gamma + 8 * lane);
  const float4* bp = (const float4*)(beta + 8 * lane);
  const float4 g0 = gp[0], g1 = gp[1], b0 = bp[0], b1 = bp[1];
  uint8_t pads[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) pads[i] = row_pad[min(row0 + wave + 8 * i, R - 1)];
  __builtin_amdgcn_sched_barrier(0);
  after_loads();
  __builtin_amdgcn_sched_barrier(0);
  const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
  const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave + 8 * i;
    if (row0 + r >= R) break;   // wave-uniform
    const float4 z0 = zz[i][0], z1 = zz[i][1];
    const float v[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += v[j];
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float c = v[j] - mean; sq += c * c; }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
    const bool pad = pads[i] != 0;
    Vec16<bf16_t> o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.set(j, pad ? 0.f : (v[j] - mean) * rstd * gg[j] + bb[j]);
    *(uint4*)(X + r * 1024 + ((lane ^ (r & 15)) << 4)) = o.v;
  }
}
__device__ __forceinline__ void ln_rows(char* X, const float* z, int row0, int R, const float* gamma, const float* beta,
                                        const uint8_t* row_pad, float eps, int wave, int lane) {
  ln_rows(X, z, row0, R, gamma, beta, row_pad, eps, wave, lane, [] {});
}

// ---- the same LayerNorm fed from the team's 8 partial second feed-forward products: every member sums ALL columns of
// its team's rows itself (partials in member order, + bias, rounded to bf16, + the residual held in X — the arithmetic
// of a reduce into the exchange buffer followed by ln_rows) instead of summing its 64 columns, publishing them and
// meeting the team once more.  8 x the reads of the partials (they sit in the L2) against one team barrier less:
// 0.663 against 0.677 ms per step.  (The attention blocks' output projections do not gain from the same trade — each
// member multiplying its own head's outputs by a 64-column slice of the matrix, K = 64, and every member summing the
// eight partial products: one barrier instead of two per block, but 0.674 - 0.684 ms per step.)
__device__ __forceinline__ void ln_rows_partials(char* X, const float* pbuf, int R32, int row0, int R, const float* b2,
                                                 const float* gamma, const float* beta, const uint8_t* row_pad, float eps,
                                                 int wave, int lane) {
  const float4* gp = (const float4*)(gamma + 8 * lane);
  const float4* bp = (const float4*)(beta + 8 * lane);
  const float4* cp = (const float4*)(b2 + 8 * lane);
  const float4 g0 = gp[0], g1 = gp[1], b0 = bp[0], b1 = bp[1], c0 = cp[0], c1 = cp[1];
  const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
  const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
  for (int ip = 0; ip < 4; ip += 2) {   // two rows' sixteen partial slices in flight together
    float4 pp[2][TEAM][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int mm = 0; mm < TEAM; ++mm) {
        const float4* q = (const float4*)(pbuf + ((int64_t)mm * R32 + min(row0 + wave + 8 * (ip + i), R - 1)) * D + 8 * lane);
        pp[i][mm][0] = q[0];
        pp[i][mm][1] = q[1];
      }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = wave + 8 * (ip + i);
      if (row0 + r >= R) break;   // wave-uniform
      const int rg = row0 + r;
      float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mm = 0; mm < TEAM; ++mm) {
        s[0] += pp[i][mm][0].x; s[1] += pp[i][mm][0].y; s[2] += pp[i][mm][0].z; s[3] += pp[i][mm][0].w;
        s[4] += pp[i][mm][1].x; s[5] += pp[i][mm][1].y; s[6] += pp[i][mm][1].z; s[7] += pp[i][mm][1].w;
      }
      const uint4 raw = *(const uint4*)(X + r * 1024 + ((lane ^ (r & 15)) << 4));
      const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w};
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float res = bf16_to_f32((bf16_t)((j & 1) ? (rw[j >> 1] >> 16) : (rw[j >> 1] & 0xffff)));
        v[j] = round_bf16(s[j] + cc[j]) + res;
      }
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[j];
      const float mean = wave_sum(sum) / (float)D;
      float sq = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float c = v[j] - mean; sq += c * c; }
      const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
      const bool pad = row_pad[rg] != 0;
      Vec16<bf16_t> o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o.set(j, pad ? 0.f : (v[j] - mean) * rstd * gg[j] + bb[j]);
      *(uint4*)(X + r * 1024 + ((lane ^ (r & 15)) << 4)) = o.v;
    }
  }
}

// ---- single-query attention of NR (row, head) pairs by one wave, their memory requests in flight together.
// Lane = (position group rg = lane / 8, 16-byte chunk ch = lane % 8 of the 64-wide head): in a sweep of 64 positions
// the lane handles positions j0 + rg + 8 i, i = 0 .. 7, and of each the chunk ch of the key AND of the value, so that
// all sixteen loads of a sweep depend on the history-row table only and travel together, every load instruction
// covers eight whole 128-byte rows, and the probabilities are already in the lanes that multiply the values.  A
// score is summed over the 8 lanes of a group; maxima and sums over the groups are taken across the wave.  hr:
// history-row table of the query row (self-attention) or null; newk / newv: LDS rows holding the newest position's
// key / value, which are also appended to the caches here.  n == 0: nothing to do for the row.
template <int NR>
struct AttRows {
  const bf16_t* q[NR];      // LDS
  const int32_t* hr[NR];
  const bf16_t* newk[NR];   // LDS or null
  const bf16_t* newv[NR];
  bf16_t* out[NR];
  int n[NR], kr_row[NR];
};

// Lane exchanges without the LDS crossbar where the hardware has a cheaper form: DPP within rows of 16 lanes
// (quad permutes, half-row mirror, row rotate), ds_swizzle with an immediate pattern within 32 lanes; only the
// exchange between the wave's halves needs ds_bpermute.
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float group8_sum(float v) {   // sum over the 8 lanes of a position group, in all of them
  v += dpp<0xB1>(v);    // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp<0x4E>(v);    // quad_perm [2,3,0,1]: lane ^ 2
  v += dpp<0x141>(v);   // row_half_mirror: the other quad of the 8 (every lane of a quad holds the quad's sum)
  return v;
}
__device__ __forceinline__ float xor8(float v) { return dpp<0x128>(v); }   // row_ror:8 == lane ^ 8
__device__ __forceinline__ float xor16(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)); }
__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {   // c + a.lo * b.lo + a.hi * b.hi (bf16 pairs)
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

// The arithmetic stays on packed bf16 pairs (v_dot2c_f32_bf16: two products and an fp32 sum per instruction): with
// two waves per SIMD nothing hides the instruction stream, and converting every key / value element to fp32 first
// took more time than the memory traffic.  Scores: four dot2 per 8-element chunk, scaled after the group sum.
// P.V: the probabilities of two positions are rounded to a bf16 pair (as the training kernels' P is) and multiply
// the two positions' value elements, paired up by a byte permute; the normaliser sums the same rounded pairs.
template <int NR>
__device__ __forceinline__ void attend(const AttRows<NR>& rw, float scale, const bf16_t* Kh, const bf16_t* Vh,
                                       bf16_t* Kw, bf16_t* Vw, int Lmax, const uint8_t* key_pad, int lane) {
  const int ch = lane & 7, rg = lane >> 3;
  uint4 qv[NR];
  float m[NR], l[NR], accv[NR][8];
  int nmax = 0;
#pragma unroll
  for (int x = 0; x < NR; ++x) {
    qv[x] = *(const uint4*)(rw.q[x] + 8 * ch);
#pragma unroll
    for (int e = 0; e < 8; ++e) accv[x][e] = 0.f;
    m[x] = -INFINITY;
    l[x] = 0.f;
    nmax = max(nmax, rw.n[x]);
  }
  const uint32_t ones = 0x3f803f80u;   // bf16 (1, 1)
  for (int j0 = 0; j0 < nmax; j0 += 64) {
    int jr[NR][8];
    uint4 kq[NR][8], vq[NR][8];
    uint8_t pad[NR][8];
#pragma unroll
    for (int x = 0; x < NR; ++x)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int pos = j0 + rg + 8 * i;
        const int pc = min(pos, max(rw.n[x] - 1, 0));   // clamped: loads stay inside the row's history
        jr[x][i] = (rw.hr[x] ? rw.hr[x][pc] : rw.kr_row[x]) * Lmax + pc;
      }
#pragma unroll
    for (int x = 0; x < NR; ++x)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int pos = j0 + rg + 8 * i;
        const bool fresh = rw.newk[x] && pos == rw.n[x] - 1;
        // (positions past the history re-read the newest one, a line the wave has anyway: a conditional load
        // would make the fragments conditionally defined values, which the register allocator keeps in memory)
        pad[x][i] = key_pad[jr[x][i]];
        if (fresh) {
          kq[x][i] = *(const uint4*)(rw.newk[x] + 8 * ch);
          vq[x][i] = *(const uint4*)(rw.newv[x] + 8 * ch);
          *(uint4*)(Kw + (int64_t)jr[x][i] * D + 8 * ch) = kq[x][i];
          *(uint4*)(Vw + (int64_t)jr[x][i] * D + 8 * ch) = vq[x][i];
          pad[x][i] = 0;
        } else {
          kq[x][i] = *(const uint4*)(Kh + (int64_t)jr[x][i] * D + 8 * ch);
          vq[x][i] = *(const uint4*)(Vh + (int64_t)jr[x][i] * D + 8 * ch);
        }
      }
#pragma unroll
    for (int x = 0; x < NR; ++x) {
      float sc[8], mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float d = dot2(kq[x][i].x, qv[x].x, 0.f);
        d = dot2(kq[x][i].y, qv[x].y, d);
        d = dot2(kq[x][i].z, qv[x].z, d);
        d = dot2(kq[x][i].w, qv[x].w, d);
        d = group8_sum(d) * scale;
        const int pos = j0 + rg + 8 * i;
        sc[i] = (pos < rw.n[x] && !pad[x][i]) ? d : -INFINITY;
        mx = fmaxf(mx, sc[i]);
      }
      mx = fmaxf(mx, xor8(mx));
      mx = fmaxf(mx, xor16(mx));
      mx = fmaxf(mx, xor32(mx));
      const float mn = fmaxf(m[x], mx);
      const bool any = mn != -INFINITY;
      const float alpha = any ? __expf(m[x] - mn) : 1.f;
      m[x] = mn;
#pragma unroll
      for (int e = 0; e < 8; ++e) accv[x][e] *= alpha;
      float ls = 0.f;
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const float p0 = any ? __expf(sc[i] - mn) : 0.f, p1 = any ? __expf(sc[i + 1] - mn) : 0.f;
        const uint32_t pp = pack_bf16x2(p0, p1);
        ls = dot2(pp, ones, ls);
        const uint32_t va[4] = {vq[x][i].x, vq[x][i].y, vq[x][i].z, vq[x][i].w};
        const uint32_t vb[4] = {vq[x][i + 1].x, vq[x][i + 1].y, vq[x][i + 1].z, vq[x][i + 1].w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          accv[x][2 * d] = dot2(__builtin_amdgcn_perm(vb[d], va[d], 0x05040100u), pp, accv[x][2 * d]);
          accv[x][2 * d + 1] = dot2(__builtin_amdgcn_perm(vb[d], va[d], 0x07060302u), pp, accv[x][2 * d + 1]);
        }
      }
      l[x] = l[x] * alpha + ls;   // this group's share; the groups are added up at the end
    }
  }
#pragma unroll
  for (int x = 0; x < NR; ++x) {
    if (rw.n[x] == 0) continue;
    float lt = l[x];
    lt += xor8(lt);
    lt += xor16(lt);
    lt += xor32(lt);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = accv[x][e];
      v += xor8(v);
      v += xor16(v);
      v += xor32(v);
      accv[x][e] = v;
    }
    const float inv = 1.0f / lt;
    if (rg == 0) {
      Vec16<bf16_t> ov;
#pragma unroll
      for (int e = 0; e < 8; ++e) ov.set(e, accv[x][e] * inv);
      *(uint4*)(rw.out[x] + ch * 8) = ov.v;
    }
  }
}

// ---- self-attention of a wave's rows as a software pipeline.  attend() above runs a sweep as: history-row table ->
// keys / values -> arithmetic, two memory round trips and the arithmetic one after the other, twice per row pair at
// history lengths 65 .. 128, and for a missing row of a pair it still issues that row's loads.  Here
//   - the history-row table of the team's rows is staged in LDS once per launch (all layers index the caches alike):
//     entry (r, p) = cache row of position p of row r's history * Tc + p, bit 31 = that position is padding;
//   - the newest position — whose key / value exist in LDS only — is appended to the caches up front and folded into
//     the softmax at the end, so that the sweeps see cached positions only and issue their loads unconditionally;
//   - a stage is 32 positions of a row pair (sixteen 16-byte loads per lane); the loads of stage s + 1 are issued
//     before the arithmetic of stage s, across the boundary between the wave's two row pairs as well.
// Results are left in the rows' query slots in LDS (a row's query is read by its own wave only) and copied out at
// the end.  Lane = (position group lane / 8, 16-byte chunk lane % 8), positions 32 w + group + 8 i, i = 0 .. 3.
struct KVStage {
  uint4 k[2][4], v[2][4];
  uint32_t ok;   // bit 4 x + i: position i of row x is inside the history and not padding
};

struct SelfRows {
  int n[4];    // positions of the wave's rows wave + 8 i, newest included (0: no such row)
  int ns[2];   // stages of the two row pairs
};

__device__ __forceinline__ void self_issue(KVStage& st, const SelfRows& sr, const int32_t* JT, int TcP, const bf16_t* Kh,
                                           const bf16_t* Vh, int wave, int s, int lane) {
  const int ch = lane & 7, rg = lane >> 3;
  const int total = sr.ns[0] + sr.ns[1];
  const int sc = max(min(s, total - 1), 0);   // past the end: the last stage once more
  const int p = sc >= sr.ns[0] ? 1 : 0, w = sc - (p ? sr.ns[0] : 0);
  uint32_t ok = 0;
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int nh = max((p ? sr.n[2 + x] : sr.n[x]) - 1, 0);   // cached positions
    const int32_t* jt = JT + (wave + 8 * (2 * p + x)) * TcP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pos = 32 * w + rg + 8 * i;
      const int32_t e = jt[min(pos, max(nh - 1, 0))];   // clamped: loads stay inside the row's history
      const int64_t jr = e & 0x7fffffff;
      st.k[x][i] = *(const uint4*)(Kh + jr * D + 8 * ch);
      st.v[x][i] = *(const uint4*)(Vh + jr * D + 8 * ch);
      if (pos < nh && e >= 0) ok |= 1u << (4 * x + i);
    }
  }
  st.ok = ok;
}

struct SelfState {
  float m[2], l[2], acc[2][8];
};

__device__ __forceinline__ void self_compute(const KVStage& st, SelfState& S, const SelfRows& sr, bf16_t* QKV, int wave,
                                             int s, float scale, int lane) {
  const int ch = lane & 7, rg = lane >> 3;
  const int p = s >= sr.ns[0] ? 1 : 0, w = s - (p ? sr.ns[0] : 0);
  const bool last = w == (p ? sr.ns[1] : sr.ns[0]) - 1;
  const uint32_t ones = 0x3f803f80u;   // bf16 (1, 1)
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int r = wave + 8 * (2 * p + x);
    const int n = p ? sr.n[2 + x] : sr.n[x];
    const uint4 qv = *(const uint4*)(QKV + r * DH + 8 * ch);
    float sc[4], mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float d = dot2(st.k[x][i].x, qv.x, 0.f);
      d = dot2(st.k[x][i].y, qv.y, d);
      d = dot2(st.k[x][i].z, qv.z, d);
      d = dot2(st.k[x][i].w, qv.w, d);
      d = group8_sum(d) * scale;
      sc[i] = (st.ok >> (4 * x + i)) & 1 ? d : -INFINITY;
      mx = fmaxf(mx, sc[i]);
    }
    mx = fmaxf(mx, xor8(mx));
    mx = fmaxf(mx, xor16(mx));
    mx = fmaxf(mx, xor32(mx));
    const float mn = fmaxf(S.m[x], mx);
    const bool any = mn != -INFINITY;
    const float alpha = any ? __expf(S.m[x] - mn) : 1.f;
    S.m[x] = mn;
#pragma unroll
    for (int e = 0; e < 8; ++e) S.acc[x][e] *= alpha;
    float ls = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
      const float p0 = any ? __expf(sc[i] - mn) : 0.f, p1 = any ? __expf(sc[i + 1] - mn) : 0.f;
      const uint32_t pp = pack_bf16x2(p0, p1);
      ls = dot2(pp, ones, ls);
      const uint32_t va[4] = {st.v[x][i].x, st.v[x][i].y, st.v[x][i].z, st.v[x][i].w};
      const uint32_t vb[4] = {st.v[x][i + 1].x, st.v[x][i + 1].y, st.v[x][i + 1].z, st.v[x][i + 1].w};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        S.acc[x][2 * d] = dot2(__builtin_amdgcn_perm(vb[d], va[d], 0x05040100u), pp, S.acc[x][2 * d]);
        S.acc[x][2 * d + 1] = dot2(__builtin_amdgcn_perm(vb[d], va[d], 0x07060302u), pp, S.acc[x][2 * d + 1]);
      }
    }
    S.l[x] = S.l[x] * alpha + ls;   // this group's share
    if (last) {   // wave-uniform: the row's cached positions are done — add the groups up, fold the newest position in
      if (n > 0) {
        float lt = S.l[x];
        lt += xor8(lt);
        lt += xor16(lt);
        lt += xor32(lt);
        const uint4 kn = *(const uint4*)(QKV + (ROWS + r) * DH + 8 * ch);
        const uint4 vn = *(const uint4*)(QKV + (2 * ROWS + r) * DH + 8 * ch);
        float d = dot2(kn.x, qv.x, 0.f);
        d = dot2(kn.y, qv.y, d);
        d = dot2(kn.z, qv.z, d);
        d = dot2(kn.w, qv.w, d);
        d = group8_sum(d) * scale;
        const float mf = fmaxf(S.m[x], d);
        const float af = __expf(S.m[x] - mf);   // no cached position counted: exp(-inf) = 0
        const float pn = round_bf16(__expf(d - mf));
        const float inv = 1.0f / (lt * af + pn);
        const uint32_t vw[4] = {vn.x, vn.y, vn.z, vn.w};
        Vec16<bf16_t> ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = S.acc[x][e];
          v += xor8(v);
          v += xor16(v);
          v += xor32(v);
          const float vne = bf16_to_f32((bf16_t)((e & 1) ? (vw[e >> 1] >> 16) : (vw[e >> 1] & 0xffff)));
          ov.set(e, (v * af + pn * vne) * inv);
        }
        if (rg == 0) *(uint4*)(QKV + r * DH + 8 * ch) = ov.v;
      }
      S.m[x] = -INFINITY;
      S.l[x] = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) S.acc[x][e] = 0.f;
    }
  }
}

// att: the team's first row of the attention buffer, head m's columns; Kc / Vc: the layer's caches at head m's columns
__device__ __forceinline__ void self_rows(SelfRows& sr, const int32_t* cur_len, int Tc, int row0, int R, int wave) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rgl = row0 + wave + 8 * i;
    sr.n[i] = rgl < R ? min(cur_len[min(rgl, R - 1)], Tc) : 0;
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int nmax = max(sr.n[2 * p], sr.n[2 * p + 1]);
    sr.ns[p] = nmax > 0 ? max((nmax - 1 + 31) >> 5, 1) : 0;   // (a row with the newest position only still takes a stage)
  }
}

// A: stage 0, already requested (self_issue(A, .., 0, ..) — before the q | k | v product of the layer, which hides its
// trip to memory; with no rows at all, total == 0, it holds the loads of a clamped address and is dropped)
__device__ __forceinline__ void self_attend_pipe(KVStage& A, const SelfRows& sr, const int32_t* JT, int TcP, bf16_t* QKV,
                                                 bf16_t* Kc, bf16_t* Vc, bf16_t* att, int row0, int R, float scale,
                                                 int wave, int lane) {
  const int ch = lane & 7, rg = lane >> 3;
  const int total = sr.ns[0] + sr.ns[1];
  if (total == 0) return;
  // the newest position joins the caches: groups 0 / 1 store the key / value chunks
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (sr.n[i] == 0) continue;
    const int r = wave + 8 * i;
    const int64_t jr = JT[r * TcP + sr.n[i] - 1] & 0x7fffffff;
    if (rg == 0) *(uint4*)(Kc + jr * D + 8 * ch) = *(const uint4*)(QKV + (ROWS + r) * DH + 8 * ch);
    if (rg == 1) *(uint4*)(Vc + jr * D + 8 * ch) = *(const uint4*)(QKV + (2 * ROWS + r) * DH + 8 * ch);
  }
  SelfState S;
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    S.m[x] = -INFINITY;
    S.l[x] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) S.acc[x][e] = 0.f;
  }
  KVStage B;
#pragma unroll 1
  for (int s = 0; s < total; s += 2) {
    self_issue(B, sr, JT, TcP, Kc, Vc, wave, s + 1, lane);
    __builtin_amdgcn_sched_barrier(0);
    self_compute(A, S, sr, QKV, wave, s, scale, lane);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 >= total) break;
    self_issue(A, sr, JT, TcP, Kc, Vc, wave, s + 2, lane);
    __builtin_amdgcn_sched_barrier(0);
    self_compute(B, S, sr, QKV, wave, s + 1, scale, lane);
    __builtin_amdgcn_sched_barrier(0);
  }
  // the wave's rows -> the attention buffer (lanes 0 .. 31: row lane / 8, chunk lane % 8)
  if (lane < 32 && sr.n[0] > 0) {   // (n[i] > 0 only if n[0] > 0: rows are dealt out in order)
    const int i = lane >> 3, r = wave + 8 * i;
    if (row0 + r < R) *(uint4*)(att + (int64_t)r * D + 8 * ch) = *(const uint4*)(QKV + r * DH + 8 * ch);
  }
}

// ---- encoder attention with the beams of a sentence sharing its keys / values.  The rows of a sentence (its beams)
// attend to the SAME encoder keys and values, so the unit of work is (sentence, segment of 32 source positions): a wave
// loads the segment once (four keys and four values per lane, all in flight together) and runs every row of the
// sentence that belongs to the team against it, leaving per (row, segment) the maximum score, the sum of
// exponentials against it and the un-normalised P.V sums in LDS; merge_xsegments() combines a row's segments.
// (Row by row, attend() loads the sentence's keys / values once per beam: five times the loads and their address
// arithmetic, and with 20 rows on 8 waves half of the waves make two passes.)
// Lane = (position group rg = lane / 8, 16-byte chunk ch = lane % 8): positions j0 + rg + 8 i, i = 0 .. 3.
// The segment's keys / values do not depend on the queries: a wave requests those of its first (sentence, segment) pair
// before the LayerNorm and the query projection of the phase, which then hide the trip to memory.
struct XSeg {
  uint4 kq[4], vq[4];
  uint32_t ok;   // bit i: position i is inside the sentence and not padding
};

__device__ __forceinline__ void xseg_load(XSeg& xs, int sent, int j0, int Ls, const bf16_t* Kh, const bf16_t* Vh,
                                          const uint8_t* key_pad, int lane) {
  const int ch = lane & 7, rg = lane >> 3;
  uint8_t pad[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pc = min(j0 + rg + 8 * i, Ls - 1);   // clamped: loads stay inside the sentence
    const int64_t jr = (int64_t)sent * Ls + pc;
    pad[i] = key_pad[jr];
    xs.kq[i] = *(const uint4*)(Kh + jr * D + 8 * ch);
    xs.vq[i] = *(const uint4*)(Vh + jr * D + 8 * ch);
  }
  uint32_t ok = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (j0 + rg + 8 * i < Ls && !pad[i]) ok |= 1u << i;
  xs.ok = ok;
}

__device__ __forceinline__ void attend_shared(const XSeg& xs, const bf16_t* q_lds, int r_lo, int r_hi, int seg, float scale,
                                              float* xacc, float* xml, int lane) {
  const int ch = lane & 7, rg = lane >> 3;
  const uint32_t ones = 0x3f803f80u;   // bf16 (1, 1)
  uint4 kq[4], vq[4];
  bool ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    kq[i] = xs.kq[i];
    vq[i] = xs.vq[i];
    ok[i] = (xs.ok >> i) & 1;
  }
#pragma unroll 1
  for (int r = r_lo; r < r_hi; ++r) {
    const uint4 qv = *(const uint4*)(q_lds + r * DH + 8 * ch);
    float sc[4], mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float d = dot2(kq[i].x, qv.x, 0.f);
      d = dot2(kq[i].y, qv.y, d);
      d = dot2(kq[i].z, qv.z, d);
      d = dot2(kq[i].w, qv.w, d);
      d = group8_sum(d) * scale;
      sc[i] = ok[i] ? d : -INFINITY;
      mx = fmaxf(mx, sc[i]);
    }
    mx = fmaxf(mx, xor8(mx));
    mx = fmaxf(mx, xor16(mx));
    mx = fmaxf(mx, xor32(mx));
    const bool any = mx != -INFINITY;
    float acc[8], ls = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
      const float p0 = any ? __expf(sc[i] - mx) : 0.f, p1 = any ? __expf(sc[i + 1] - mx) : 0.f;
      const uint32_t pp = pack_bf16x2(p0, p1);
      ls = dot2(pp, ones, ls);
      const uint32_t va[4] = {vq[i].x, vq[i].y, vq[i].z, vq[i].w};
      const uint32_t vb[4] = {vq[i + 1].x, vq[i + 1].y, vq[i + 1].z, vq[i + 1].w};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        acc[2 * d] = dot2(__builtin_amdgcn_perm(vb[d], va[d], 0x05040100u), pp, acc[2 * d]);
        acc[2 * d + 1] = dot2(__builtin_amdgcn_perm(vb[d], va[d], 0x07060302u), pp, acc[2 * d + 1]);
      }
    }
    ls += xor8(ls);
    ls += xor16(ls);
    ls += xor32(ls);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = acc[e];
      v += xor8(v);
      v += xor16(v);
      v += xor32(v);
      acc[e] = v;
    }
    if (rg == 0) {
      float* o = xacc + (seg * ROWS + r) * DH + 8 * ch;
      *(float4*)o = make_float4(acc[0], acc[1], acc[2], acc[3]);
      *(float4*)(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
      if (ch == 0) *(float2*)(xml + 2 * (seg * ROWS + r)) = make_float2(mx, ls);
    }
  }
}

// a row's segments -> its attention output (bf16, head m's 64 columns of att): thread = (row tid / 16, 4 columns)
__device__ __forceinline__ void merge_xsegments(const float* xacc, const float* xml, bf16_t* att, int nseg, int row0, int R,
                                                int m, int tid) {
  const int r = tid >> 4, c = 4 * (tid & 15), rg = row0 + r;
  if (rg >= R) return;
  float M = -INFINITY;
  for (int sg = 0; sg < nseg; ++sg) M = fmaxf(M, xml[2 * (sg * ROWS + r)]);
  float den = 0.f, num[4] = {0.f, 0.f, 0.f, 0.f};
  for (int sg = 0; sg < nseg; ++sg) {
    const int it = sg * ROWS + r;
    const float2 ml = *(const float2*)(xml + 2 * it);
    const float w = ml.x == -INFINITY ? 0.f : __expf(ml.x - M);
    const float4 a = *(const float4*)(xacc + it * DH + c);
    den += ml.y * w;
    num[0] += a.x * w; num[1] += a.y * w; num[2] += a.z * w; num[3] += a.w * w;
  }
  const float inv = 1.0f / den;   // a fully masked row: 0 / 0, as in the row-by-row form
  const float o[4] = {num[0] * inv, num[1] * inv, num[2] * inv, num[3] * inv};
  *(uint2*)(att + (int64_t)rg * D + DH * m + c) = pack4(o);
}

// gather the team's 32 attention rows (bf16 [.][512]) into the swizzled LDS image
__device__ __forceinline__ void load_att(char* A, const bf16_t* att, int row0, int R, int tid) {
  uint4 v[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int i = tid + NTHREADS * it;
    v[it] = *(const uint4*)(att + (int64_t)min(row0 + (i >> 6), R - 1) * D + 8 * (i & 63));
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int i = tid + NTHREADS * it;
    const int r = i >> 6, c = i & 63;
    *(uint4*)(A + r * 1024 + ((c ^ (r & 15)) << 4)) = v[it];
  }
}

// z[row][col .. col + 3] = bf16(acc + bias) + residual, for the two 16-row halves of a 16-column tile
__device__ __forceinline__ void store_z(float* zbuf, const char* X, const f32x4_t (&acc)[2], const float4 b, int col,
                                        int row0, int R, int l15) {
  const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int r = 16 * mi + l15;
    float res[4];
    x_read4(X, r, col, res);
    float4 o;
    o.x = round_bf16(acc[mi][0] + bb[0]) + res[0];
    o.y = round_bf16(acc[mi][1] + bb[1]) + res[1];
    o.z = round_bf16(acc[mi][2] + bb[2]) + res[2];
    o.w = round_bf16(acc[mi][3] + bb[3]) + res[3];
    if (row0 + r < R) *(float4*)(zbuf + (int64_t)(row0 + r) * D + col) = o;
  }
}

template <bool PIPE>   // self-attention by self_attend_pipe() over the history table in LDS / by attend()
__global__ __launch_bounds__(NTHREADS) void decode_stack_kernel(StackArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int team = blockIdx.x % a.teams_pad, m0 = blockIdx.x / a.teams_pad;   // members of a team: same blockIdx mod 8
  const int n_teams = (a.R + a.rpt - 1) / a.rpt;
  if (team >= n_teams) return;
  const int tid_ = threadIdx.x, row0_ = team * a.rpt, m_ = m0, R = min(a.R, (team + 1) * a.rpt);   // R: end of this team's rows
  // Every phase derives its lane / row / member indices afresh from values the optimiser cannot see through:
  // otherwise it hoists the address arithmetic of ALL phases out of the layer loop and keeps it in registers
  // (and, past 256 of them, in scratch memory) for the whole kernel.
#define IDS()                                                                  \
  int tid = tid_, row0 = row0_, m = m_;                                        \
  asm volatile("" : "+v"(tid));                                                \
  asm volatile("" : "+s"(row0));                                               \
  asm volatile("" : "+s"(m));                                                  \
  const int lane = tid & 63, l15 = lane & 15, g = lane >> 4;                   \
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                   \
  (void)l15; (void)g; (void)wave; (void)row0; (void)m
  char* X = smem + L_X;
  char* A = smem + L_A;
  char* HS = smem + L_HS;
  bf16_t* QKV = (bf16_t*)(smem + L_QKV);

#define GP(T, x) ((T*)(__attribute__((address_space(1))) T*)(x))
  typedef const __attribute__((address_space(4))) int64_t* kargs_t;   // LAYER_FIELDS addresses per layer
  const kargs_t kargs = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();   // StackArgs::layer is the first member
  TeamCtx ctx;
  ctx.flag = a.flags + a.parity * MAX_TEAMS + team;
  ctx.err = a.flags + 2 * MAX_TEAMS;
  ctx.count = 0;
  ctx.dead = false;
  ctx.member = m0;
  ctx.scalar_poll = a.scalar_poll;
  // slot form: [2 parities][MAX_TEAMS] lines of 16 words behind the exchange buffers; a member clears its word of
  // the other parity's line for the next step
  ctx.slots = a.slots ? a.slots + (a.parity * MAX_TEAMS + team) * 16 : nullptr;
  if (a.slots && tid_ == 0) a.slots[((a.parity ^ 1) * MAX_TEAMS + team) * 16 + m0] = 0;
  int32_t* const xcc_mask = a.flags + XCC_WORD0 + a.parity * MAX_TEAMS + team;
  if (m0 == 0 && tid_ == 0) {   // the next step's counter and XCC mask
    a.flags[(a.parity ^ 1) * MAX_TEAMS + team] = 0;
    a.flags[XCC_WORD0 + (a.parity ^ 1) * MAX_TEAMS + team] = 0;
  }
  if (tid_ == 0) {   // (the first barrier waits for this request like for every other one of the wave)
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    __hip_atomic_fetch_or(xcc_mask, 1 << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // q | k | v of head m are 12 column tiles of 16: tile t = (which = t / 4, sub = t % 4); wave w takes tile w, and
  // waves 0..3 tile 8 + w as well
  auto qkv_row = [](int t, int m) { return (t >> 2) * D + DH * m + 16 * (t & 3); };
  WFrag<16> wq[2];
  {
  IDS();
  load_w<16>(wq[0], GP(const bf16_t, kargs[0]), D, qkv_row(wave, m) + l15, 0, g);
  load_w<16>(wq[1], GP(const bf16_t, kargs[0]), D, qkv_row(8 + (wave & 3), m) + l15, 0, g);
  // embedding of the newest token of each row: word_emb[tok] + pos_tab[pos]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 4 * wave + i;
    const int rg = min(row0 + r, R - 1);
    const float4* wp = (const float4*)(a.word_emb + a.tok[rg] * D + 8 * lane);
    const float4* pp = (const float4*)(a.pos_tab + a.pos[rg] * D + 8 * lane);
    const float4 w0 = wp[0], w1 = wp[1], p0 = pp[0], p1 = pp[1];
    Vec16<bf16_t> o;
    o.set(0, w0.x + p0.x); o.set(1, w0.y + p0.y); o.set(2, w0.z + p0.z); o.set(3, w0.w + p0.w);
    o.set(4, w1.x + p1.x); o.set(5, w1.y + p1.y); o.set(6, w1.z + p1.z); o.set(7, w1.w + p1.w);
    *(uint4*)(X + r * 1024 + ((lane ^ (r & 15)) << 4)) = o.v;
  }
  // history table: per thread 8 entries at a time, their two dependent loads in flight together
  if constexpr (PIPE) {
    int32_t* JT = (int32_t*)(smem + L_JT);
    const int n_ent = ROWS * a.TcP;
#pragma unroll 1
    for (int base = 0; base < n_ent; base += 8 * NTHREADS) {
      int32_t h[8], pc[8];
      uint8_t pd[8];
      int64_t hrow[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = min(base + tid + NTHREADS * u, n_ent - 1);
        const int r = idx / a.TcP, pp = idx - r * a.TcP;
        const int rgl = min(row0 + r, R - 1);
        pc[u] = min(pp, a.Tc - 1);
        hrow[u] = (int64_t)rgl * a.Tc;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) h[u] = a.hist[hrow[u] + pc[u]];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        // (entries past a row's length are never used: the table there holds whatever the history table does,
        // clamped into the cache so that even a stray use would stay inside it)
        h[u] = min(max(h[u], 0), a.R - 1) * a.Tc + pc[u];
        pd[u] = a.pad_hist[h[u]];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + tid + NTHREADS * u;
        if (idx < n_ent) JT[idx] = h[u] | (pd[u] ? (int32_t)0x80000000 : 0);
      }
    }
  }
  }
  __syncthreads();

  SelfRows sr;
  if constexpr (PIPE) {
    IDS();
    self_rows(sr, a.cur_len, a.Tc, row0, R, wave);
  }
  int stamp_i = 0;
#define STAMP() do { if (a.stamps && (int)blockIdx.x == a.stamp_block && tid_ == a.stamp_tid && li == min(3, a.n_layers - 1)) a.stamps[stamp_i++] = wall_clock64(); } while (0)
  for (int li = 0; li < a.n_layers; ++li) {
    // Layer addresses are read from the kernel-argument segment (indexing the by-value copy would go through
    // scratch) and declared GLOBAL: from a plain integer cast the compiler could only assume the flat address
    // space, whose loads count as LDS traffic too and serialise against every LDS wait.
    LayerPtrs P;
    {
      const kargs_t q = kargs + (int64_t)LAYER_FIELDS * li;
      P.Wqkv = GP(const bf16_t, q[0]); P.bqkv = GP(const float, q[1]);
      P.Wo = GP(const bf16_t, q[2]); P.bo = GP(const float, q[3]); P.g1 = GP(const float, q[4]); P.be1 = GP(const float, q[5]);
      P.Wqc = GP(const bf16_t, q[6]); P.bqc = GP(const float, q[7]);
      P.Woc = GP(const bf16_t, q[8]); P.boc = GP(const float, q[9]); P.g2 = GP(const float, q[10]); P.be2 = GP(const float, q[11]);
      P.W1 = GP(const bf16_t, q[12]); P.b1 = GP(const float, q[13]);
      P.W2 = GP(const bf16_t, q[14]); P.b2 = GP(const float, q[15]); P.g3 = GP(const float, q[16]); P.be3 = GP(const float, q[17]);
      P.Kc = GP(bf16_t, q[18]); P.Vc = GP(bf16_t, q[19]); P.crossK = GP(const bf16_t, q[20]); P.crossV = GP(const bf16_t, q[21]);
    }
    STAMP();  // 0
    // ---- A: q | k | v of head m for the 32 rows
    // (loads return in the order of their issue: whatever a phase needs early is requested BEFORE a long-latency
    // prefetch, never behind it)
    KVStage st0;
    float4 bqv[2];
    {
      IDS();
      bqv[0] = *(const float4*)(P.bqkv + qkv_row(wave, m) + 4 * g);
      bqv[1] = *(const float4*)(P.bqkv + qkv_row(8 + (wave & 3), m) + 4 * g);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PIPE) {
      IDS();
      self_issue(st0, sr, (const int32_t*)(smem + L_JT), a.TcP, P.Kc + DH * m, P.Vc + DH * m, wave, 0, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
    {
    IDS();
    // (both tiles in one pass over the image; waves 4..7 multiply a second tile too and drop it)
    f32x4_t accq[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) accq[u][0] = accq[u][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    mma_rows_tiles<2, 16>(accq, wq, X, 1024, 0, l15, g);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && wave >= 4) break;
      const int t = u == 0 ? wave : 8 + wave;
      const f32x4_t (&acc)[2] = accq[u];
      const float4 b = bqv[u];
      const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const float v[4] = {acc[mi][0] + bb[0], acc[mi][1] + bb[1], acc[mi][2] + bb[2], acc[mi][3] + bb[3]};
        *(uint2*)(QKV + ((t >> 2) * ROWS + 16 * mi + l15) * DH + 16 * (t & 3) + 4 * g) = pack4(v);
      }
    }
    }
    __syncthreads();
    STAMP();  // 1: after A
    // ---- B: self-attention of head m (wave w: the team's rows w, w + 8, w + 16, w + 24, two at a time), appending
    // the new key / value
    if constexpr (PIPE) {
      IDS();
      self_attend_pipe(st0, sr, (const int32_t*)(smem + L_JT), a.TcP, QKV, P.Kc + DH * m, P.Vc + DH * m,
                       a.att + (int64_t)row0 * D + DH * m, row0, R, a.scale, wave, lane);
    } else
    {
    IDS();
#pragma unroll 1
    for (int i = 0; i < 4; i += 2) {
      AttRows<2> rw;
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int r = wave + 8 * (i + x), rg = row0 + r;
        const bool on = rg < R;
        const int rc = min(rg, R - 1);
        rw.q[x] = QKV + r * DH;
        rw.hr[x] = a.hist + (int64_t)rc * a.Tc;
        rw.newk[x] = QKV + (ROWS + r) * DH;
        rw.newv[x] = QKV + (2 * ROWS + r) * DH;
        rw.out[x] = a.att + (int64_t)rc * D + DH * m;
        rw.n[x] = on ? min(a.cur_len[rc], a.Tc) : 0;
        rw.kr_row[x] = rc;
      }
      attend<2>(rw, a.scale, P.Kc + DH * m, P.Vc + DH * m, P.Kc + DH * m, P.Vc + DH * m, a.Tc, a.pad_hist, lane);
    }
    }
    STAMP();  // 2: after self-attention
    // weights are requested where the registers are free and the wait for the team hides their latency
    __builtin_amdgcn_sched_barrier(0);
    WFrag<16> wo;
    {
    IDS();
    // (waves 4..7 load what they do not use: a value defined on one path only is kept in memory)
    team_sync(ctx, [&] { load_w<16>(wo, P.Wo, D, DH * m + 16 * (wave & 3) + l15, 0, g); });
    }
    STAMP();  // 3: after sync 1
    // ---- C: output projection columns 64 m .., + residual -> z
    WFrag<16> wc;
    {
    IDS();
    // (small vectors are requested before the workgroup barrier of their phase: behind it their latency would show)
    const float4 bias = *(const float4*)(P.bo + DH * m + 16 * (wave & 3) + 4 * g);
    load_att(A, a.att, row0, R, tid);
    __syncthreads();
    if (wave < 4) {
      f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
      mma_rows<16>(acc, wo, A, 1024, 0, l15, g);
      store_z(a.zbuf, X, acc, bias, DH * m + 16 * wave + 4 * g, row0, R, l15);
    }
    }
    STAMP();  // 4: after C
    {
    IDS();
    team_sync(ctx, [&] { load_w<16>(wc, P.Wqc, D, DH * m + 16 * (wave & 3) + l15, 0, g); });
    }
    STAMP();  // 5: after sync 2
    // ---- D: LayerNorm, query projection of head m, attention over the sentence's encoder keys / values
    const bool xshared = a.Ls <= XSEG * XSEG_MAX && a.xshare;
    XSeg xs;
    {
    IDS();
    const float4 b = *(const float4*)(P.bqc + DH * m + 16 * (wave & 3) + 4 * g);
    ln_rows(X, a.zbuf, row0, R, P.g1, P.be1, a.row_pad, a.eps, wave, lane, [&] {
      // the wave's first (sentence, segment) pair (or the team's last one again, which nobody then uses)
      const int s_lo = row0 / a.bm, s_hi = (R - 1) / a.bm, nseg = (a.Ls + XSEG - 1) / XSEG;
      const int it = min(wave, (s_hi - s_lo + 1) * nseg - 1);
      xseg_load(xs, s_lo + it / nseg, XSEG * (it - (it / nseg) * nseg), a.Ls, P.crossK + DH * m, P.crossV + DH * m,
                a.src_key_pad, lane);
    });
    __syncthreads();
    if (wave < 4) {
      f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
      mma_rows<16>(acc, wc, X, 1024, 0, l15, g);
      const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const float v[4] = {acc[mi][0] + bb[0], acc[mi][1] + bb[1], acc[mi][2] + bb[2], acc[mi][3] + bb[3]};
        *(uint2*)(QKV + (16 * mi + l15) * DH + 16 * wave + 4 * g) = pack4(v);
      }
    }
    }
    __syncthreads();
    STAMP();  // 6: after LN + cross-q
    if (xshared) {
      // the team's (sentence, segment) pairs, dealt out to the waves round-robin; then the rows' segments are merged
      IDS();
      const int s_lo = row0 / a.bm, s_hi = (R - 1) / a.bm, nseg = (a.Ls + XSEG - 1) / XSEG;
      const int total = (s_hi - s_lo + 1) * nseg;
      float* xacc = (float*)(smem + L_XACC);
      float* xml = (float*)(smem + L_XML);
      if (wave < total) {
        const int sent = s_lo + wave / nseg, seg = wave - (wave / nseg) * nseg;
        const int r_lo = max(sent * a.bm, row0) - row0, r_hi = min((sent + 1) * a.bm, R) - row0;
        attend_shared(xs, QKV, r_lo, r_hi, seg, a.scale, xacc, xml, lane);
      }
#pragma unroll 1
      for (int it = wave + NTHREADS / 64; it < total; it += NTHREADS / 64) {
        const int sent = s_lo + it / nseg, seg = it - (it / nseg) * nseg;
        const int r_lo = max(sent * a.bm, row0) - row0, r_hi = min((sent + 1) * a.bm, R) - row0;
        XSeg xn;
        xseg_load(xn, sent, XSEG * seg, a.Ls, P.crossK + DH * m, P.crossV + DH * m, a.src_key_pad, lane);
        attend_shared(xn, QKV, r_lo, r_hi, seg, a.scale, xacc, xml, lane);
      }
      __syncthreads();
      merge_xsegments(xacc, xml, a.att, nseg, row0, R, m, tid);
    } else {
    IDS();
#pragma unroll 1
    for (int i = 0; i < 4; i += 2) {
      AttRows<2> rw;
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int r = wave + 8 * (i + x), rg = row0 + r;
        const int rc = min(rg, R - 1);
        rw.q[x] = QKV + r * DH;
        rw.hr[x] = nullptr;
        rw.newk[x] = nullptr;
        rw.newv[x] = nullptr;
        rw.out[x] = a.att + (int64_t)rc * D + DH * m;
        rw.n[x] = rg < R ? a.Ls : 0;
        rw.kr_row[x] = rc / a.bm;
      }
      attend<2>(rw, a.scale, P.crossK + DH * m, P.crossV + DH * m, nullptr, nullptr, a.Ls, a.src_key_pad, lane);
    }
    }
    STAMP();  // 7: after cross attention
    __builtin_amdgcn_sched_barrier(0);
    WFrag<16> wo2;
    {
    IDS();
    team_sync(ctx, [&] { load_w<16>(wo2, P.Woc, D, DH * m + 16 * (wave & 3) + l15, 0, g); });
    }
    STAMP();  // 8: after sync 3
    // ---- E: second output projection + residual -> z
    WFrag<16> w1[2];
    {
    IDS();
    const float4 bias = *(const float4*)(P.boc + DH * m + 16 * (wave & 3) + 4 * g);
    load_att(A, a.att, row0, R, tid);
    __syncthreads();
    if (wave < 4) {
      f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
      mma_rows<16>(acc, wo2, A, 1024, 0, l15, g);
      store_z(a.zbuf, X, acc, bias, DH * m + 16 * wave + 4 * g, row0, R, l15);
    }
    }
    STAMP();  // 9: after E
    {
    IDS();
    // hidden columns 256 m .. 256 m + 255 are 16 tiles: wave w takes tiles 2 w, 2 w + 1
    team_sync(ctx, [&] {
#pragma unroll
      for (int t = 0; t < 2; ++t) load_w<16>(w1[t], P.W1, D, 256 * m + 32 * wave + 16 * t + l15, 0, g);
    });
    }
    STAMP();  // 10: after sync 4
    // ---- F: LayerNorm, hidden slice, ReLU, kept in LDS
    WFrag<8> w2[4];
    {
    IDS();
    float4 b1v[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) b1v[t] = *(const float4*)(P.b1 + 256 * m + 32 * wave + 16 * t + 4 * g);
    // G operands: output columns 64 w + 16 t .. (rows of W2), K range 256 m .. 256 m + 255; one of the four tiles
    // behind the LayerNorm's own loads, the others as the hidden product consumes its weight fragments and leaves
    // them registers (both hidden tiles in one pass over the image)
    ln_rows(X, a.zbuf, row0, R, P.g2, P.be2, a.row_pad, a.eps, wave, lane,
            [&] { load_w<8>(w2[0], P.W2, F, 64 * wave + l15, 256 * m, g); });
    __syncthreads();
    f32x4_t acch[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) acch[t][0] = acch[t][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    mma_rows_tiles<2, 16>(acch, w1, X, 1024, 0, l15, g, [&](int kk) {
      if (kk == 3) load_w<8>(w2[1], P.W2, F, 64 * wave + 16 + l15, 256 * m, g);
      if (kk == 7) load_w<8>(w2[2], P.W2, F, 64 * wave + 32 + l15, 256 * m, g);
      if (kk == 11) load_w<8>(w2[3], P.W2, F, 64 * wave + 48 + l15, 256 * m, g);
    });
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f32x4_t (&acc)[2] = acch[t];
      const int cl = 32 * wave + 16 * t + 4 * g;   // column within the slice
      const float4 b = b1v[t];
      const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int r = 16 * mi + l15;
        const float v[4] = {fmaxf(acc[mi][0] + bb[0], 0.f), fmaxf(acc[mi][1] + bb[1], 0.f), fmaxf(acc[mi][2] + bb[2], 0.f),
                            fmaxf(acc[mi][3] + bb[3], 0.f)};
        *(uint2*)(HS + r * 512 + (((cl >> 3) ^ (r & 15)) << 4) + (cl & 7) * 2) = pack4(v);
      }
    }
    }
    __syncthreads();
    STAMP();  // 11: after LN + F
    // ---- G: partial second feed-forward product over this member's 256 hidden columns -> pbuf[m]
    {
    IDS();
    f32x4_t acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t][0] = acc[t][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    mma_rows_tiles<4, 8>(acc, w2, HS, 512, 0, l15, g);
    // The accumulators hold, per lane, four columns of each 16-column tile: stored as they are, an instruction writes
    // 64-byte pieces of sixteen rows, partial lines that the barrier's wait for the stores then sits out (6 us).
    // Through a wave-private 4 KB of LDS (the attention image, idle here; rows padded by 16 bytes) they leave as the
    // wave's whole 256 bytes of four rows per instruction.
    float* scr = (float*)(wave < 6 ? A + wave * 4352 : (char*)QKV + (wave - 6) * 4352);   // [16][68] floats (the q | k | v image is idle as well)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int t = 0; t < 4; ++t) *(f32x4_t*)(scr + l15 * 68 + 16 * t + 4 * g) = acc[t][mi];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * i + (lane >> 4), rg = row0 + 16 * mi + r;
        const f32x4_t v = *(const f32x4_t*)(scr + r * 68 + 4 * (lane & 15));
        if (rg < R) *(f32x4_t*)(a.pbuf + ((int64_t)m * a.R32 + rg) * D + 64 * wave + 4 * (lane & 15)) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    }
    STAMP();  // 12: after G
    // the next layer's q | k | v weights (loaded unconditionally — the last layer re-reads its own — so that the
    // old fragments are dead here and need no registers)
    {
    IDS();
    team_sync(ctx, [&] {
      const kargs_t qn = kargs + (int64_t)LAYER_FIELDS * min(li + 1, a.n_layers - 1);
      load_w<16>(wq[0], GP(const bf16_t, qn[0]), D, qkv_row(wave, m) + l15, 0, g);
      load_w<16>(wq[1], GP(const bf16_t, qn[0]), D, qkv_row(8 + (wave & 3), m) + l15, 0, g);
    });
    }
    STAMP();  // 13: after sync 5
    if (a.fuse_red) {
      // ---- the partial products summed by every member for all columns, + bias, + residual, LayerNorm -> X
      IDS();
      ln_rows_partials(X, a.pbuf, a.R32, row0, R, P.b2, P.g3, P.be3, a.row_pad, a.eps, wave, lane);
      STAMP();  // 14
      STAMP();  // 15
    } else {
    // ---- reduce the 8 partial products for columns 64 m .. (fixed order), + bias, + residual -> z
    {
      IDS();
      const int r = tid >> 4, col = DH * m + 4 * (tid & 15);
      const int rg = row0 + r;
      if (rg < R) {
        float4 pp[TEAM];
#pragma unroll
        for (int mm = 0; mm < TEAM; ++mm) pp[mm] = *(const float4*)(a.pbuf + ((int64_t)mm * a.R32 + rg) * D + col);
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int mm = 0; mm < TEAM; ++mm) { s4.x += pp[mm].x; s4.y += pp[mm].y; s4.z += pp[mm].z; s4.w += pp[mm].w; }
        const float4 b = *(const float4*)(P.b2 + col);
        float res[4];
        x_read4(X, r, col, res);
        float4 o;
        o.x = round_bf16(s4.x + b.x) + res[0];
        o.y = round_bf16(s4.y + b.y) + res[1];
        o.z = round_bf16(s4.z + b.z) + res[2];
        o.w = round_bf16(s4.w + b.w) + res[3];
        *(float4*)(a.zbuf + (int64_t)rg * D + col) = o;
      }
    }
    STAMP();  // 14: after reduce
    team_sync(ctx);
    STAMP();  // 15: after sync 6
    {
    IDS();
    ln_rows(X, a.zbuf, row0, R, P.g3, P.be3, a.row_pad, a.eps, wave, lane);
    }
    }
    __syncthreads();
    STAMP();  // 16: after LN
  }
#undef STAMP
  // every member's bit has been in the team's XCC mask since the first barrier: exactly one bit, or the exchanges
  // above read another L2's stale lines
  if (tid_ == 0) {
    const int mk = __hip_atomic_load(xcc_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__popc(mk) != 1) __hip_atomic_store(ctx.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // the stack's output rows, columns 64 m .. 64 m + 63 by member m
  {
    IDS();
    const int r = (tid >> 3) & 31, c = 8 * m + (tid & 7);
    const int rg = row0 + r;
    // (fragment-major: the order the generator product reads its MFMA operands in, see skhip_generator_rowstats)
    const int64_t at = a.out_fm ? ((int64_t)((rg >> 4) * 16 + (c >> 2)) * 64 + 16 * (c & 3) + (rg & 15)) * 8 : (int64_t)rg * D + 8 * c;
    if (tid < 256 && rg < R) *(uint4*)(a.dec_out + at) = *(const uint4*)(X + r * 1024 + ((c ^ (r & 15)) << 4));
  }
}

#undef IDS
#undef GP
}  // namespace

extern "C" int64_t skhip_decode_stack_ws_bytes(int R) {
  const int64_t R32 = (R + 31) / 32 * 32;
  return 1024 + R32 * 512 * 2 + R32 * 512 * 4 + 8 * R32 * 512 * 4 + 2 * MAX_TEAMS * 64;
}

extern "C" int skhip_decode_stack(const int64_t* layers, int n_layers, const int64_t* tok, const int64_t* pos,
                                  const float* word_emb, const float* pos_tab, const int32_t* cur_len,
                                  const uint8_t* row_pad, const uint8_t* pad_hist, const int32_t* hist,
                                  const uint8_t* src_key_pad, void* dec_out, void* workspace, int64_t workspace_bytes,
                                  int R, int bm, int Tc, int Ls, int d_model, int n_head, int d_head, int d_inner,
                                  float scale, float eps, int parity, int out_layout, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(d_model == D && n_head == H && d_head == DH && d_inner == F,
               "skhip_decode_stack: built for d_model 512, 8 heads of 64, d_inner 2048 (got %d, %d x %d, %d)", d_model, n_head,
               d_head, d_inner);
  SK_CHECK_ARG(n_layers >= 1 && n_layers <= MAX_LAYERS, "skhip_decode_stack: n_layers=%d not in 1..%d", n_layers, MAX_LAYERS);
  SK_CHECK_ARG(R >= 1 && bm >= 1 && Tc >= 1 && Ls >= 1 && (parity == 0 || parity == 1), "skhip_decode_stack: bad shape");
  SK_CHECK_ARG(out_layout == SK_ROWS_PLAIN || out_layout == SK_ROWS_FRAGMENT_MAJOR, "skhip_decode_stack: out_layout=%d", out_layout);
  // as many teams as fit the chip at one workgroup per CU (every CU then pulls its share of the keys / values
  // and of the weights), each with the same number of rows, at most 32
  const int R32 = (R + 31) / 32 * 32;
  const int max_teams = std::min(MAX_TEAMS, sk_num_cus() / TEAM);
  SK_CHECK_ARG(max_teams >= 1 && (R + max_teams - 1) / max_teams <= ROWS,
               "skhip_decode_stack: %d rows need more than the %d teams of 32 rows that are resident together on %d CUs", R,
               max_teams, sk_num_cus());
  const int rpt = (R + max_teams - 1) / max_teams, n_teams = (R + rpt - 1) / rpt;
  SK_CHECK_ARG(layers && tok && pos && word_emb && pos_tab && cur_len && row_pad && pad_hist && hist && src_key_pad &&
               dec_out && workspace, "skhip_decode_stack: null pointer");
  SK_CHECK_ARG(workspace_bytes >= skhip_decode_stack_ws_bytes(R) && (uintptr_t)workspace % 256 == 0,
               "skhip_decode_stack: workspace of %lld bytes, 256-byte aligned, needed", (long long)skhip_decode_stack_ws_bytes(R));
  StackArgs a;
  for (int i = 0; i < n_layers; ++i) {
    const int64_t* q = layers + (int64_t)LAYER_FIELDS * i;
    for (int f = 0; f < LAYER_FIELDS; ++f)
      SK_CHECK_ARG(q[f] != 0 && q[f] % 16 == 0, "skhip_decode_stack: layer %d field %d is null or not 16-byte aligned", i, f);
    LayerPtrs& P = a.layer[i];
    P.Wqkv = (const bf16_t*)q[0]; P.bqkv = (const float*)q[1];
    P.Wo = (const bf16_t*)q[2]; P.bo = (const float*)q[3]; P.g1 = (const float*)q[4]; P.be1 = (const float*)q[5];
    P.Wqc = (const bf16_t*)q[6]; P.bqc = (const float*)q[7];
    P.Woc = (const bf16_t*)q[8]; P.boc = (const float*)q[9]; P.g2 = (const float*)q[10]; P.be2 = (const float*)q[11];
    P.W1 = (const bf16_t*)q[12]; P.b1 = (const float*)q[13];
    P.W2 = (const bf16_t*)q[14]; P.b2 = (const float*)q[15]; P.g3 = (const float*)q[16]; P.be3 = (const float*)q[17];
    P.Kc = (bf16_t*)q[18]; P.Vc = (bf16_t*)q[19]; P.crossK = (const bf16_t*)q[20]; P.crossV = (const bf16_t*)q[21];
  }
  a.tok = tok; a.pos = pos; a.word_emb = word_emb; a.pos_tab = pos_tab;
  a.cur_len = cur_len; a.row_pad = row_pad; a.pad_hist = pad_hist; a.hist = hist; a.src_key_pad = src_key_pad;
  a.dec_out = (bf16_t*)dec_out; a.out_fm = out_layout == SK_ROWS_FRAGMENT_MAJOR;
  char* ws = (char*)workspace;
  a.flags = (int32_t*)ws;
  a.att = (bf16_t*)(ws + 1024);
  a.zbuf = (float*)(ws + 1024 + (int64_t)R32 * 512 * 2);
  a.pbuf = (float*)(ws + 1024 + (int64_t)R32 * 512 * 2 + (int64_t)R32 * 512 * 4);
  a.slots = (int32_t*)(ws + 1024 + (int64_t)R32 * 512 * 2 + (int64_t)R32 * 512 * 4 + 8 * (int64_t)R32 * 512 * 4);
  { const char* e = getenv("SKHIP_STACK_SLOTS"); if (e && e[0] == '0') a.slots = nullptr; }
  { const char* e = getenv("SKHIP_STACK_SPOLL"); a.scalar_poll = !(e && e[0] == '0'); }
  a.n_layers = n_layers; a.R = R; a.R32 = R32; a.bm = bm; a.Tc = Tc; a.Ls = Ls; a.parity = parity; a.rpt = rpt;
  a.teams_pad = (n_teams + 7) / 8 * 8;   // members of a team sit at the same block index modulo 8
  a.scale = scale; a.eps = eps;
  { const char* e = getenv("SKHIP_STACK_XSHARE"); a.xshare = !(e && e[0] == '0'); }
  { const char* e = getenv("SKHIP_STACK_FUSERED"); a.fuse_red = !(e && e[0] == '0'); }
  a.stamps = getenv("SKHIP_STACK_STAMPS") ? (long long*)(ws + 512) : nullptr;
  { const char* e = getenv("SKHIP_STACK_STAMP_BLOCK"); a.stamp_block = e ? atoi(e) : 0; }
  { const char* e = getenv("SKHIP_STACK_STAMP_TID"); a.stamp_tid = e ? atoi(e) : 0; }
  // the history table rides in LDS when it fits beside the images (history rows of up to 512 positions)
  a.TcP = (Tc + 31) / 32 * 32;
  { const char* e = getenv("SKHIP_STACK_PIPE"); if ((e && e[0] == '0') || L_JT + ROWS * a.TcP * 4 > L_MAX) a.TcP = 0; }
  const int lds_bytes = a.TcP ? L_JT + ROWS * a.TcP * 4 : L_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)decode_stack_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, L_MAX);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)decode_stack_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, L_BYTES);
    if (e != hipSuccess) { sk_set_error("skhip_decode_stack: LDS attribute: %s", hipGetErrorString(e)); return SK_ERR_LAUNCH; }
    attr_set = true;
  }
  if (a.TcP) hipLaunchKernelGGL(decode_stack_kernel<true>, dim3((unsigned)(a.teams_pad * TEAM)), dim3(NTHREADS), lds_bytes, stream, a);
  else hipLaunchKernelGGL(decode_stack_kernel<false>, dim3((unsigned)(a.teams_pad * TEAM)), dim3(NTHREADS), lds_bytes, stream, a);
  SK_CHECK_LAUNCH("skhip_decode_stack");
  return SK_OK;
}
