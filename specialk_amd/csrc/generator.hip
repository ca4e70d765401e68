// The generator product of a decode step (gfx950): log-probability inputs of all beam rows against the whole
// vocabulary, `self.model.generator(dec_output)` of Translator.predict_word (lib/transformer/Translator.py:136-137)
// for d_model = 512, with the vocabulary, not output tiles, as the unit of work.
//
// A few hundred rows against 35 k vocabulary entries is 5 x 274 tiles of 128 x 128: 1.8 rounds of the chip, every
// tile re-reading its operands.  Here a workgroup owns a SLAB of 144 consecutive vocabulary entries — 35 000 / 144 =
// 244 slabs, ONE round of the 256 CUs — whose weight rows ([144][512] bf16, 1 KB each, 144 KB) it brings into LDS
// once (so the weight matrix is read from memory exactly once per step), and multiplies every row of the
// activations by it: the activation rows (a few hundred KB, shared by all workgroups: L2) stream straight into MFMA
// operand registers.  512 threads = 8 waves; wave w takes the 16-row blocks w, w + 8, .. up to three at a time (3 x 9
// MFMA tiles, 108 accumulator registers), and after each pass writes the fp32 products and, per row and 48-column
// CHUNK (three chunks per slab), the chunk's maximum and its sum of exponentials against that maximum — what the beam
// top-k needs to find the normaliser and the few chunks that can hold a best token without reading the rows.
//
// The activations are best handed over FRAGMENT-MAJOR (the order the MFMA operands are read in: a load instruction
// of a wave fetches 1 KB of consecutive bytes).  Row-major, the same instruction takes 64 bytes from each of 16 rows
// 1 KB apart, and those loads — not the cache they come from: it made no difference whether every wave read the same
// 16 rows — cost the loop as much again as its MFMAs (640 rows x 35 000: 48 us against 41).
//
// MFMA operand order as in the row-panel kernels: v_mfma_f32_16x16x32_bf16(weight fragment, activation fragment):
// lane (l15 = lane & 15, g = lane >> 4) supplies k = 8 g .. 8 g + 7 of slab row 16 ni + l15 and of activation row
// l15, and receives out[activation row l15][slab rows 16 ni + 4 g .. + 3].
#include "common.h"

namespace {
namespace gs {

constexpr int D = 512;             // contraction length
constexpr int NT = 9;              // 16-column tiles of a slab
constexpr int SLAB = 16 * NT;      // 144 vocabulary entries
constexpr int CW = 48;             // columns of a statistics chunk (three tiles)
constexpr int WROW = D * 2;        // bytes of a weight row
constexpr int LDS_BYTES = SLAB * WROW;
constexpr int MAXB = 3;            // 16-row blocks of the activations a wave multiplies at a time
constexpr int RING = 4;            // weight fragments in flight from LDS (RING - 1 items ahead)
constexpr int XD = 4;              // K-steps of activation fragments in flight
static_assert(SLAB % 8 == 0 && SLAB % CW == 0, "slab shape");

struct Params {
  const bf16_t* X; int64_t ldx;   // activation rows [R][512], or fragment-major (the XFM kernel; ldx unused)
  const bf16_t* W;                // generator weight [V][512], row-major
  float* C;                       // products [R][V]
  float2* stats;                  // [R][nchunk]: (maximum, sum of exponentials against it) per 48-column chunk
  int R, V, nchunk, vec;          // vec: rows of C start on 16-byte boundaries
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef const __attribute__((address_space(3))) bf16x8_t* lds_frag_t;

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// What a wave keeps across its passes over the activations.
struct WaveCtx {
  int lane, l15, g, c0;
  int offs[4];       // byte offsets of the lane's weight-fragment chunk in the four K-steps of a 256-byte group
  uint32_t wa32;     // LDS address of the lane's slab row of tile 0
};

// Activation fragments of the first XD K-steps of the blocks b0, b0 + 8, .. (rows past the end read the last row).
template <bool XFM>
__device__ __forceinline__ void x_pointers(const Params& p, const WaveCtx& c, int b0, const bf16_t* (&xp)[MAXB]) {
#pragma unroll
  for (int nb = 0; nb < MAXB; ++nb)
    xp[nb] = XFM ? p.X + ((int64_t)min(b0 + 8 * nb, ((p.R + 15) >> 4) - 1) * 16 * 64 + c.lane) * 8
                 : p.X + (int64_t)min(16 * (b0 + 8 * nb) + c.l15, p.R - 1) * p.ldx + 8 * c.g;
}
__device__ __forceinline__ void x_first(const bf16_t* const (&xp)[MAXB], bf16x8_t (&xr)[MAXB][XD], int xstep) {
#pragma unroll
  for (int s = 0; s < XD; ++s)
#pragma unroll
    for (int nb = 0; nb < MAXB; ++nb) xr[nb][s] = *(const bf16x8_t*)(xp[nb] + xstep * s);
}

// NB 16-row blocks against the slab.  Program order is the schedule (a scheduling barrier after every item): the
// weight fragment of item (K-step, tile) is read from LDS three items ahead through a ring of four, the activation
// fragments of a K-step are requested XD K-steps ahead; left to itself the compiler moves every read next to its
// use and waits for it there.
template <int NB>
__device__ __forceinline__ void products(const WaveCtx& c, const bf16_t* const (&xp)[MAXB], bf16x8_t (&xr)[MAXB][XD],
                                         f32x4_t (&acc)[MAXB][NT], int xstep) {
  // LDS addresses: four per-lane bases (one per K-step of a 256-byte group) made opaque here, so that they are
  // derived in every pass instead of being kept, one register per (tile, K-step) pair, across the whole kernel; the
  // rest of an address is a constant the instruction carries (below 64 KB: tiles 4 .. 7 and 8 add 64 KB steps)
  uint32_t wb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    wb[j] = c.wa32 + (uint32_t)c.offs[j];
    asm volatile("" : "+v"(wb[j]));
  }
  auto read_w = [&](int it) {
    const int ks = it / NT, ni = it % NT;
    const uint32_t a = wb[ks & 3] + (uint32_t)(ni >> 2) * 65536u;
    return *(lds_frag_t)(uintptr_t)(a + (uint32_t)((ni & 3) * 16 * WROW + (ks >> 2) * 256));
  };
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) acc[nb][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t ring[RING];
#pragma unroll
  for (int r = 0; r < RING - 1; ++r) ring[r] = read_w(r);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const int it = ks * NT + ni;
      if (it + RING - 1 < 16 * NT) ring[(it + RING - 1) % RING] = read_w(it + RING - 1);
      const bf16x8_t w = ring[it % RING];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[nb][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xr[nb][ks % XD], acc[nb][ni], 0, 0, 0);
      if (ni == NT - 1 && ks + XD < 16) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) xr[nb][ks % XD] = *(const bf16x8_t*)(xp[nb] + xstep * (ks + XD));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Products and chunk statistics of NB blocks: the lanes l15, l15 + 16, l15 + 32, l15 + 48 hold a row's 16 columns of
// a tile.  FULL: the whole slab lies inside the vocabulary and the rows of C start on 16-byte boundaries (every
// workgroup but the last): no column is masked, every store is 16 bytes.
template <int NB, bool FULL>
__device__ __forceinline__ void write_out(const Params& p, const WaveCtx& c, int b0, f32x4_t (&acc)[MAXB][NT]) {
  const int g = c.g, c0 = c.c0;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int row = 16 * (b0 + 8 * nb) + c.l15;
    const bool live = row < p.R;
    float* crow = p.C + (int64_t)(live ? row : 0) * p.V;
#pragma unroll
    for (int q = 0; q < SLAB / CW; ++q) {
      float m = -INFINITY;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int col = c0 + 16 * (3 * q + t) + 4 * g;
        const f32x4_t v = acc[nb][3 * q + t];
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, FULL || col + r < p.V ? v[r] : -INFINITY);
        if (live) {
          if (FULL || (p.vec && col + 3 < p.V)) {
            *(float4*)(crow + col) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (col + r < p.V) crow[col + r] = v[r];
          }
        }
      }
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float se = 0.f;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int col = c0 + 16 * (3 * q + t) + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) se += FULL || col + r < p.V ? __expf(acc[nb][3 * q + t][r] - m) : 0.f;
      }
      if (!FULL && m == -INFINITY) se = 0.f;
      se += __shfl_xor(se, 16, 64);
      se += __shfl_xor(se, 32, 64);
      const int chunk = (c0 + CW * q) / CW;
      if (live && g == 0 && chunk < p.nchunk) p.stats[(int64_t)row * p.nchunk + chunk] = make_float2(m, se);
    }
  }
}

template <bool XFM>
__global__ __launch_bounds__(512, 1) void slab_kernel(Params p) {
  constexpr int xstep = XFM ? 512 : 32;   // elements between a lane's fragments of consecutive K-steps
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int c0 = blockIdx.x * SLAB;
  // ---- the slab: one direct-to-LDS instruction per weight row (64 lanes x 16 bytes); entries past the vocabulary
  // read its last row and are masked where they come out
#pragma unroll
  for (int i = 0; i < SLAB / 8; ++i) {
    const int s = wave + 8 * i;
    const int c = (lane & ~15) | ((lane ^ s) & 15);  // the logical chunk stored at chunk position `lane`
    const bf16_t* src = p.W + (int64_t)min(c0 + s, p.V - 1) * D + 8 * c;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(smem + s * WROW), 16, 0, 0);
  }
  {
    // a read of the rows this wave requested: the compiler places the wait for them
    int probe = 0;
#pragma unroll
    for (int i = 0; i < SLAB / 8; ++i) probe |= *(volatile const int*)(smem + (wave + 8 * i) * WROW + lane * 4);
    asm volatile("" :: "v"(probe));
  }
  lds_barrier();
  WaveCtx c;
  c.lane = lane; c.l15 = lane & 15; c.g = lane >> 4; c.c0 = c0;
#pragma unroll
  for (int j = 0; j < 4; ++j) c.offs[j] = ((4 * j + c.g) ^ c.l15) << 4;
  c.wa32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + c.l15 * WROW;
  // wave w takes the blocks w, w + 8, .. up to three at a time; the second wave of a SIMD (waves 4 .. 7) starts with
  // the short pass where the first one ends with it, so that the two are not in their epilogues together.  The first
  // activation fragments of a pass are requested before the stores of the pass before it (the memory counter is in
  // order: behind them, the first product of the pass would wait for every one of those stores)
  const bool full = p.vec && c0 + SLAB <= p.V;
  const int nblk = (p.R + 15) >> 4;
  const int n = wave < nblk ? (nblk - wave + 7) >> 3 : 0;   // blocks of this wave
  auto pass_size = [&](int j) { return wave >= 4 && j == 0 && n % MAXB ? n % MAXB : min(MAXB, n - j); };
  const bf16_t* xp[MAXB];
  bf16x8_t xr[MAXB][XD];
  f32x4_t acc[MAXB][NT];
  if (n) {
    x_pointers<XFM>(p, c, wave, xp);
    x_first(xp, xr, xstep);
  }
  for (int j = 0; j < n;) {
    const int nb = pass_size(j), b0 = wave + 8 * j;
    if (nb == 3) products<3>(c, xp, xr, acc, xstep);
    else if (nb == 2) products<2>(c, xp, xr, acc, xstep);
    else products<1>(c, xp, xr, acc, xstep);
    j += nb;
    if (j < n) {
      x_pointers<XFM>(p, c, wave + 8 * j, xp);
      x_first(xp, xr, xstep);
    }
    if (full) {
      if (nb == 3) write_out<3, true>(p, c, b0, acc);
      else if (nb == 2) write_out<2, true>(p, c, b0, acc);
      else write_out<1, true>(p, c, b0, acc);
    } else {
      if (nb == 3) write_out<3, false>(p, c, b0, acc);
      else if (nb == 2) write_out<2, false>(p, c, b0, acc);
      else write_out<1, false>(p, c, b0, acc);
    }
  }
}

}  // namespace gs
}  // namespace

extern "C" int skhip_generator_chunk_cols(void) { return gs::CW; }

extern "C" int skhip_generator_rowstats(const void* x, int64_t ldx, int x_layout, const void* w, float* logits, float* stats,
                                        int64_t R, int64_t V, int64_t K, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SK_CHECK_ARG(R >= 0 && V > 0 && K == gs::D, "skhip_generator_rowstats: R=%lld V=%lld K=%lld (K must be 512)", (long long)R,
               (long long)V, (long long)K);
  SK_CHECK_ARG(x_layout == SK_ROWS_PLAIN || x_layout == SK_ROWS_FRAGMENT_MAJOR, "skhip_generator_rowstats: x_layout=%d", x_layout);
  if (R == 0) return SK_OK;
  SK_CHECK_ARG(x && w && logits && stats, "skhip_generator_rowstats: null pointer");
  SK_CHECK_ARG((x_layout == SK_ROWS_FRAGMENT_MAJOR || (ldx % 8 == 0 && ldx >= K)) && (uintptr_t)x % 16 == 0 &&
                   (uintptr_t)w % 16 == 0 && (uintptr_t)logits % 16 == 0 && (uintptr_t)stats % 8 == 0,
               "skhip_generator_rowstats: operands must be 16-byte aligned, the row stride a multiple of 8 elements");
  SK_CHECK_ARG(R < (1 << 24) && V < (1ll << 31) - gs::SLAB, "skhip_generator_rowstats: shape too large");
  gs::Params p{};
  p.X = (const bf16_t*)x; p.ldx = ldx; p.W = (const bf16_t*)w;
  p.C = logits; p.stats = (float2*)stats;
  p.R = (int)R; p.V = (int)V; p.nchunk = (int)((V + gs::CW - 1) / gs::CW); p.vec = V % 4 == 0;
  static bool allowed = false;
  if (!allowed) {
    int rc = sk_allow_big_lds(gs::slab_kernel<false>, gs::LDS_BYTES);
    if (rc == SK_OK) rc = sk_allow_big_lds(gs::slab_kernel<true>, gs::LDS_BYTES);
    if (rc) return rc;
    allowed = true;
  }
  const unsigned blocks = (unsigned)((V + gs::SLAB - 1) / gs::SLAB);
  if (x_layout == SK_ROWS_FRAGMENT_MAJOR)
    hipLaunchKernelGGL(gs::slab_kernel<true>, dim3(blocks), dim3(512), gs::LDS_BYTES, stream, p);
  else
    hipLaunchKernelGGL(gs::slab_kernel<false>, dim3(blocks), dim3(512), gs::LDS_BYTES, stream, p);
  SK_CHECK_LAUNCH("skhip_generator_rowstats");
  return SK_OK;
}
