// Memory-only model of the fused attention backward: per (sentence, head) read the q, k, v, o, do rows of the head
// (128 bytes of a 3072- or 1024-byte row) and write dq, dk, dv rows; no arithmetic.  Variants of the block -> work
// mapping, to see what the access pattern alone costs.   hipcc --offload-arch=gfx950 -O3 attn_mem_probe.hip -o attn_mem_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int L = 41, B = 640, H = 8, T = L * B;

// variant 0: block = (b, h), 256 threads; each thread moves 16-byte chunks: 8 chunks per 128-byte head row
__global__ __launch_bounds__(256) void per_head(const uint16_t* qkv, const uint16_t* o, const uint16_t* dO, uint16_t* dqkv, int items_per_block) {
  for (int it = 0; it < items_per_block; ++it) {
    const int item = blockIdx.x * items_per_block + it;
    const int b = item / H, h = item % H;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 v[5][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = threadIdx.x + 256 * i, r = min(e >> 3, L - 1), c = e & 7;
      const int64_t row = (int64_t)b * L + r;
      v[0][i] = *(const u32x4*)(qkv + row * 1536 + h * 64 + c * 8);
      v[1][i] = *(const u32x4*)(qkv + row * 1536 + 512 + h * 64 + c * 8);
      v[2][i] = *(const u32x4*)(qkv + row * 1536 + 1024 + h * 64 + c * 8);
      v[3][i] = *(const u32x4*)(o + row * 512 + h * 64 + c * 8);
      v[4][i] = *(const u32x4*)(dO + row * 512 + h * 64 + c * 8);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      acc = v[0][i] ^ v[1][i] ^ v[2][i] ^ v[3][i] ^ v[4][i];
      const int e = threadIdx.x + 256 * i, r = e >> 3, c = e & 7;
      if (r < L) {
        const int64_t row = (int64_t)b * L + r;
        *(u32x4*)(dqkv + row * 1536 + h * 64 + c * 8) = acc;
        *(u32x4*)(dqkv + row * 1536 + 512 + h * 64 + c * 8) = acc;
        *(u32x4*)(dqkv + row * 1536 + 1024 + h * 64 + c * 8) = acc;
      }
    }
  }
}

// variant 1: block = sentence, all heads: a thread moves chunks of whole 1024-byte row segments
__global__ __launch_bounds__(256) void per_sentence(const uint16_t* qkv, const uint16_t* o, const uint16_t* dO, uint16_t* dqkv) {
  const int b = blockIdx.x;
  for (int e = threadIdx.x; e < L * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    const int64_t row = (int64_t)b * L + r;
    u32x4 a0 = *(const u32x4*)(qkv + row * 1536 + c * 8);
    u32x4 a1 = *(const u32x4*)(qkv + row * 1536 + 512 + c * 8);
    u32x4 a2 = *(const u32x4*)(qkv + row * 1536 + 1024 + c * 8);
    u32x4 a3 = *(const u32x4*)(o + row * 512 + c * 8);
    u32x4 a4 = *(const u32x4*)(dO + row * 512 + c * 8);
    u32x4 acc = a0 ^ a1 ^ a2 ^ a3 ^ a4;
    *(u32x4*)(dqkv + row * 1536 + c * 8) = acc;
    *(u32x4*)(dqkv + row * 1536 + 512 + c * 8) = acc;
    *(u32x4*)(dqkv + row * 1536 + 1024 + c * 8) = acc;
  }
}

int main() {
  uint16_t *qkv, *o, *dO, *dqkv;
  const int NB = 4;
  CK(hipMalloc(&qkv, (size_t)NB * T * 1536 * 2)); CK(hipMalloc(&dqkv, (size_t)NB * T * 1536 * 2));
  CK(hipMalloc(&o, (size_t)NB * T * 512 * 2)); CK(hipMalloc(&dO, (size_t)NB * T * 512 * 2));
  CK(hipMemset(qkv, 1, (size_t)NB * T * 1536 * 2)); CK(hipMemset(o, 1, (size_t)NB * T * 512 * 2)); CK(hipMemset(dO, 1, (size_t)NB * T * 512 * 2));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = 8.0 * T * 512 * 2;
  for (int variant = 0; variant < 4; ++variant) {
    const int iters = 40;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < iters; ++i) {
        const size_t k = i % NB;
        const uint16_t *a = qkv + k * T * 1536, *b = o + k * T * 512, *c = dO + k * T * 512; uint16_t* d = dqkv + k * T * 1536;
        if (variant == 0) hipLaunchKernelGGL(per_head, dim3(B * H), dim3(256), 0, 0, a, b, c, d, 1);
        else if (variant == 1) hipLaunchKernelGGL(per_sentence, dim3(B), dim3(256), 0, 0, a, b, c, d);
        else if (variant == 2) hipLaunchKernelGGL(per_head, dim3(B * H / 4), dim3(256), 0, 0, a, b, c, d, 4);
        else hipLaunchKernelGGL(per_head, dim3(B * H / 8), dim3(256), 0, 0, a, b, c, d, 8);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    printf("variant %d: %7.1f us  %5.2f TB/s\n", variant, us, bytes / us / 1e6);
  }
  return 0;
}
