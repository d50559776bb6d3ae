// Probe: sustained v_mfma_f32_32x32x16_bf16 rate of the chip on NON-ZERO operands (the clock follows the power budget:
// MI355X_MICROARCH.md "DVFS give-back"), register-only loop, 1 / 2 / 3 waves per SIMD.  The exact-split recurrence
// (csrc/split_core.h) can at best run at this rate.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_bf16_peak.hip -o scripts/probes/mfma_bf16_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float seed) {
  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8 a[3], b[4];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int i = 0; i < 8; ++i) a[p][i] = (__bf16)(seed * (float)((threadIdx.x * 37 + i * 11 + p * 5) % 97 - 48) * 0.01f);
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int i = 0; i < 8; ++i) b[p][i] = (__bf16)(seed * (float)((threadIdx.x * 53 + i * 7 + p * 3) % 89 - 44) * 0.01f);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 9; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u % 3], b[(u + j) & 3], acc[j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 12345.f) out[threadIdx.x] = s;
}

// the same for v_mfma_f32_32x32x2_f32 (4096 FLOP per instruction, 64 cycles per SIMD)
template <int NACC>
__global__ void __launch_bounds__(256) mfma_f32_loop(float* out, int iters, float seed) {
  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a[4], b[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    a[p] = seed * (float)((threadIdx.x * 37 + p * 5) % 97 - 48) * 0.0123f;
    b[p] = seed * (float)((threadIdx.x * 53 + p * 3) % 89 - 44) * 0.0117f;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u & 3], b[(u + j) & 3], acc[j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 12345.f) out[threadIdx.x] = s;
}
static void run_f32(const char* name, int blocks, int iters, float seed) {
  float* out;
  (void)hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  mfma_f32_loop<4><<<blocks, 256>>>(out, iters, seed);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  mfma_f32_loop<4><<<blocks, 256>>>(out, iters, seed);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)iters * 8 * 4;
  printf("%-52s %8.3f ms  %6.1f cycles@2.4GHz/MFMA/SIMD  %8.1f TFLOP/s\n", name, ms, ms * 1e-3 * 2.4e9 / (mfmas * blocks / 256.0), (double)blocks * 4 * mfmas * 4096.0 / ms / 1e9);
  (void)hipFree(out);
}

template <int NACC>
static void run(const char* name, int blocks, int iters, float seed) {
  float* out;
  (void)hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  mfma_loop<NACC><<<blocks, 256>>>(out, iters, seed);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  mfma_loop<NACC><<<blocks, 256>>>(out, iters, seed);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * 4;
  const double mfmas = (double)iters * 9 * NACC;
  printf("%-52s %8.3f ms  %6.1f cycles@2.4GHz/MFMA/SIMD  %8.1f TFLOP/s\n", name, ms, ms * 1e-3 * 2.4e9 / (mfmas * blocks / 256.0), waves * mfmas * 32768.0 / ms / 1e9);
  (void)hipFree(out);
}

int main() {
  run<8>("256 blocks (1 wave/SIMD), 8 acc, zero operands", 256, 40000, 0.f);
  run<8>("256 blocks (1 wave/SIMD), 8 acc, random operands", 256, 40000, 1.f);
  run<8>("512 blocks (2 waves/SIMD), 8 acc, random operands", 512, 20000, 1.f);
  run<4>("768 blocks (3 waves/SIMD), 4 acc, random operands", 768, 20000, 1.f);
  run<8>("256 blocks (1 wave/SIMD), 8 acc, random, long (0.4 s)", 256, 400000, 1.f);
  run_f32("f32 32x32x2: 256 blocks (1 wave/SIMD), zero operands", 256, 20000, 0.f);
  run_f32("f32 32x32x2: 256 blocks (1 wave/SIMD), random operands", 256, 20000, 1.f);
  run_f32("f32 32x32x2: 768 blocks (3 waves/SIMD), random operands", 768, 10000, 1.f);
  run_f32("f32 32x32x2: 256 blocks, random, long (0.4 s)", 256, 200000, 1.f);
  return 0;
}
