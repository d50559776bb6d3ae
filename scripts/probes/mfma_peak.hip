// Probe: sustained v_mfma_f32_32x32x2_f32 rate per SIMD and per chip (register-only loop, no memory).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_peak.hip -o scripts/probes/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 12345.f) out[threadIdx.x] = s;
}

template <int NACC>
static void run(const char* name, int blocks, int threads, int iters) {
  float* out;
  hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  mfma_loop<NACC><<<blocks, threads>>>(out, iters, 1.0f, 2.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_loop<NACC><<<blocks, threads>>>(out, iters, 1.0f, 2.0f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * threads / 64;
  const double mfmas = (double)iters * 8 * NACC;
  const double flops = waves * mfmas * 4096.0;
  printf("%-44s %8.3f ms  %7.1f ns/MFMA/wave  %8.1f TFLOP/s\n", name, ms, ms * 1e6 / mfmas, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  run<4>("1 block x 4 waves (1 wave/SIMD), 4 acc", 1, 256, 200000);
  run<1>("1 block x 4 waves (1 wave/SIMD), 1 acc (dependent)", 1, 256, 200000);
  run<4>("256 blocks x 4 waves (1 wave/SIMD), 4 acc", 256, 256, 100000);
  run<4>("512 blocks x 4 waves (2 waves/SIMD), 4 acc", 512, 256, 100000);
  run<4>("768 blocks x 4 waves (3 waves/SIMD), 4 acc", 768, 256, 100000);
  run<4>("1024 blocks x 4 waves (4 waves/SIMD), 4 acc", 1024, 256, 100000);
  return 0;
}
