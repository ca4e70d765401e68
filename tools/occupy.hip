// Stand-in for a communication kernel that holds some CUs while the training step runs: n_blocks workgroups of
// 256 threads with `lds_bytes` of LDS each spin until `us` microseconds of wall-clock time have passed (every wave
// leaves on its own clock reading, so the grid always drains).  Built by tools/bench_comm_shadow.py's recipe:
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/occupy.hip -o tools/liboccupy.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void occupy_kernel(long long ticks, int* sink) {
  extern __shared__ int lds[];
  const long long t0 = wall_clock64();   // 100 MHz
  int acc = 0;
  while (wall_clock64() - t0 < ticks) {
    lds[threadIdx.x] = acc;
    acc += lds[(threadIdx.x + 1) & 255];
    __builtin_amdgcn_s_sleep(8);
  }
  if (acc == 0x7fffffff) *sink = acc;
}

extern "C" int occupy(int n_blocks, int us, int lds_bytes, int* sink, void* stream) {
  if (n_blocks < 1 || us < 1 || us > 5000 || lds_bytes < 1024 || lds_bytes > 65536) return -1;
  hipLaunchKernelGGL(occupy_kernel, dim3(n_blocks), dim3(256), lds_bytes, (hipStream_t)stream, (long long)us * 100, sink);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
