// How many 256-thread workgroups share a CU for a given dynamic-LDS size (gfx950): every workgroup spins for a fixed time, the
// launch time over 12 workgroups per CU shows the number that ran together.   hipcc --offload-arch=gfx950 -O2 -o lds_occ lds_occupancy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void spin(long long ticks, int* sink) {
  extern __shared__ char smem[];
  smem[threadIdx.x] = (char)threadIdx.x;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (smem[(threadIdx.x + 1) & 255] == 77 && ticks < 0) sink[0] = 1;
}
int main(int argc, char** argv) {
  int* sink;
  hipMalloc(&sink, 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const long long ticks = 2000;   // 20 us at 100 MHz
  for (int a = 1; a < argc; ++a) {
    const int lds = atoi(argv[a]);
    hipLaunchKernelGGL(spin, dim3(256 * 12), dim3(256), lds, 0, ticks, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(spin, dim3(256 * 12), dim3(256), lds, 0, ticks, sink);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("LDS %6d B: %.1f us for 12 workgroups per CU of 20 us each -> %.1f together per CU (%s)\n", lds, ms * 1e3, 12 * 20.0 / (ms * 1e3),
           hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
