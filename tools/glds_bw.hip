// Throughput of direct-to-LDS loads (global_load_lds_dwordx4) per CU as a function of the bytes requested per matrix
// row and instruction: an NT GEMM with BK = 32 asks for 64 bytes of each operand row per K-step (half a cache line),
// BK = 64 for whole 128-byte lines.  One block of 512 threads per CU streams `rows` rows of `K` bf16 elements in
// K-steps, `depth` K-steps in flight, no arithmetic.  shared = 1: all blocks read the same panel (served by the L2s);
// shared = 0: every block its own panel.
//   hipcc --offload-arch=gfx950 -O3 tools/glds_bw.hip -o /tmp/glds_bw && /tmp/glds_bw
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef const __attribute__((address_space(1))) void* gptr;
typedef __attribute__((address_space(3))) void* lptr;

template <int SEG, int STAGES>  // bytes per row and K-step: 64 or 128; K-steps in flight + 1
__global__ __launch_bounds__(512) void stream(const char* base, int64_t panel_stride, int rows, int row_bytes, int reps,
                                              int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const char* panel = base + (int64_t)blockIdx.x * panel_stride;
  constexpr int LPR = SEG / 16;             // lanes per row
  constexpr int RPI = 64 / LPR;             // rows per instruction
  const int ksteps = row_bytes / SEG;
  const int inst_per_step = rows / RPI;      // spread over 8 waves
  const int per_wave = inst_per_step / 8;
  const int stage_bytes = rows * SEG;        // <= 32 KiB
  int issued = 0;
  for (int rep = 0; rep < reps; ++rep)
    for (int ks = 0; ks < ksteps; ++ks) {
      char* st = smem + (issued % STAGES) * stage_bytes;
      for (int i = 0; i < per_wave; ++i) {
        const int inst = wave * per_wave + i;
        const int row = inst * RPI + lane / LPR;
        const char* src = panel + (int64_t)row * row_bytes + ks * SEG + (lane % LPR) * 16;
        __builtin_amdgcn_global_load_lds((gptr)src, (lptr)(st + inst * 1024), 16, 0, 0);
      }
      ++issued;
      // keep STAGES - 1 K-steps in flight
      // keep STAGES - 1 K-steps in flight (per_wave = 4 for 64-byte, 8 for 128-byte segments of 512 rows)
      if (STAGES == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (STAGES == 2 && SEG == 128) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0 && smem[5] == 123) sink[0] = 1;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const int rows = 512, K = 512, row_bytes = K * 2, reps = 64;
  const int64_t panel = (int64_t)rows * row_bytes;
  char* buf;
  int* sink;
  hipMalloc(&buf, panel * cus);
  hipMemset(buf, 1, panel * cus);
  hipMalloc(&sink, 4);
  hipFuncSetAttribute((const void*)stream<64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)stream<64, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)stream<128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int shared = 1; shared >= 0; --shared)
    for (int var = 0; var < 3; ++var) {
      const int seg = var == 2 ? 128 : 64, stages = var == 0 ? 4 : 2;
      const int lds = stages * rows * seg;
      float best = 1e30f;
      for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        if (var == 0) hipLaunchKernelGGL((stream<64, 4>), dim3(cus), dim3(512), lds, 0, buf, shared ? 0 : panel, rows, row_bytes, reps, sink);
        else if (var == 1) hipLaunchKernelGGL((stream<64, 2>), dim3(cus), dim3(512), lds, 0, buf, shared ? 0 : panel, rows, row_bytes, reps, sink);
        else hipLaunchKernelGGL((stream<128, 2>), dim3(cus), dim3(512), lds, 0, buf, shared ? 0 : panel, rows, row_bytes, reps, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double bytes = (double)panel * reps;  // per CU
      printf("%s panel, %3d bytes per row and K-step, %d stages (%3d KB in flight): %7.1f us, %6.1f GB/s per CU, %5.1f B/clk/CU at 2.1 GHz, %5.2f TB/s chip\n",
             shared ? "shared " : "private", seg, stages, (stages - 1) * rows * seg / 1024, best * 1e3, bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 2.1e9,
             bytes * cus / (best * 1e-3) / 1e12);
    }
  return 0;
}
