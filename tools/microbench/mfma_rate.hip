// How fast ONE wave per SIMD can issue v_mfma_f32_32x32x2_f32: cycles per instruction for 1, 2 and 4 independent
// accumulator chains, at 1, 2 and 3 waves per SIMD (one block of 256 / 512 / 768 threads on one CU, no memory
// traffic).  Settles what the 64x64 GEMM tile's K loop waits for (GEMM section).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_rate.hip -o tools/microbench/mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void mfma_loop(float* out, long long* cyc, int iters, float a0, float b0) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16 / CHAINS; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int e = 0; e < 16; ++e) s += acc[c][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int CHAINS>
void run(int threads, int blocks) {
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * threads * blocks);
  hipMalloc(&cyc, sizeof(long long) * 16 * blocks);
  const int iters = 2000;
  hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.f, 2.f);
  hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.f, 2.f);
  hipDeviceSynchronize();
  long long h[16];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("chains %d, %d waves per SIMD, %4d blocks: %.1f cycles per MFMA per wave -> %.1f cycles per MFMA per SIMD\n", CHAINS,
         threads / 256, blocks, (double)h[0] / (16.0 * iters), (double)h[0] / (16.0 * iters) / (threads / 256));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int blocks : {1, 256}) {
    for (int threads : {256, 512, 768}) {
      run<1>(threads, blocks);
      run<2>(threads, blocks);
      run<4>(threads, blocks);
    }
  }
  return 0;
}
