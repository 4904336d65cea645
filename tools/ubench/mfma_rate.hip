// Micro-benchmark (diagnostic): cycles per v_mfma_f32_32x32x16_f16 / 16x16x32_f16, one wave per SIMD, registers only and
// with ds_read_b128 operands.  hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 65536 / 16; i += 256) ((u32x4*)lds)[i] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  __syncthreads();
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * lane + j); b[j] = (_Float16)(0.5f - 0.001f * lane); }
  floatx16 acc[9];
  floatx4 acc4[9];
  for (int q = 0; q < 9; ++q) { for (int i = 0; i < 16; ++i) acc[q][i] = 0.f; acc4[q] = floatx4{0, 0, 0, 0}; }
  const long long t0 = clock64();
  for (int it = 0; it < (MODE == 3 ? 0 : iters); ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 9; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q], 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int q = 0; q < 9; ++q) acc4[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc4[q], 0, 0, 0);
    } else {  // 3 A + 3 B fragments from LDS per 9 MFMAs (the head kernel's stage A shape)
      half8 af[3], bf[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        af[q] = __builtin_bit_cast(half8, *(const u32x4*)(lds + ((it * 3 + q) & 31) * 1024 + lane * 16));
        bf[q] = __builtin_bit_cast(half8, *(const u32x4*)(lds + 32768 + (q * 7 + it) % 16 * 1040 + (lane & 31) * 80 + (lane >> 5) * 16));
      }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int p = 0; p < 3; ++p) acc[r * 3 + p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[r], bf[p], acc[r * 3 + p], 0, 0, 0);
    }
  }
  if (MODE == 3) {  // mode 2 with the fragments of step it+1 requested before the MFMAs of step it (two named register sets)
    auto ld = [&](half8 (&af)[3], half8 (&bf)[3], int it) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        af[q] = __builtin_bit_cast(half8, *(const u32x4*)(lds + ((it * 3 + q) & 31) * 1024 + lane * 16));
        bf[q] = __builtin_bit_cast(half8, *(const u32x4*)(lds + 32768 + (q * 7 + it) % 16 * 1040 + (lane & 31) * 80 + (lane >> 5) * 16));
      }
    };
    auto mm = [&](const half8 (&af)[3], const half8 (&bf)[3]) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int p = 0; p < 3; ++p) acc[r * 3 + p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[r], bf[p], acc[r * 3 + p], 0, 0, 0);
    };
    half8 a0[3], b0[3], a1[3], b1[3];
    ld(a0, b0, 0);
    for (int it = 0; it < iters; it += 2) {
      ld(a1, b1, it + 1);
      mm(a0, b0);
      ld(a0, b0, it + 2);
      mm(a1, b1);
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int q = 0; q < 9; ++q) { for (int i = 0; i < 16; ++i) s += acc[q][i]; s += acc4[q][0]; }
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
  long long* d; float* s;
  hipMalloc(&d, 8 * 1024); hipMalloc(&s, 4 * 1024 * 256);
  const int iters = 2000;
  for (int mode = 0; mode < 4; ++mode)
    for (int grid : {1, 256, 1024}) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, s, iters);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, s, iters);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d, s, iters);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, d, s, iters);
        hipDeviceSynchronize();
      }
      long long h[1024];
      hipMemcpy(h, d, 8 * grid, hipMemcpyDeviceToHost);
      double sum = 0;
      for (int i = 0; i < grid; ++i) sum += h[i];
      printf("mode %d grid %4d: %.1f cycles per MFMA (avg over workgroups)\n", mode, grid, sum / grid / iters / 9.0);
    }
  return 0;
}
