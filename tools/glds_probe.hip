#include <hip/hip_runtime.h>
#include <stdint.h>
// Each wave copies 1 KiB chunks global -> LDS with global_load_lds_dwordx4, then dumps LDS to out.
__global__ void k(const uint4* src, uint4* out, const int* perm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // wave w fills LDS [w*2048, w*2048+2048) with two instructions; lane reads src[perm[...]]
  for (int i = 0; i < 2; ++i) {
    const uint4* g = src + perm[wave * 128 + i * 64 + lane];
    __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)(smem + wave * 2048 + i * 1024);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, l, 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * 128; i += 256) out[i] = ((uint4*)smem)[i];
}
int main() {
  const int n = 512;
  uint4 h[n]; int perm[n];
  for (int i = 0; i < n; ++i) { h[i] = make_uint4(i, i * 10, i * 100, i * 1000); perm[i] = (i * 37 + 5) % n; }
  uint4 *d, *o; int* p;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h)); hipMalloc(&p, sizeof(perm));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice); hipMemcpy(p, perm, sizeof(perm), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 8192, 0, d, o, p);
  uint4 r[n]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) if (r[i].x != (unsigned)perm[i] || r[i].w != (unsigned)perm[i] * 1000) { if (bad < 5) printf("mismatch at %d: got %u want %d\n", i, r[i].x, perm[i]); ++bad; }
  printf("glds test: %d mismatches of %d\n", bad, n);
  return bad != 0;
}
