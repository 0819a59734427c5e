// What does the matrix pipe of this MI355X sustain?  Waves that do nothing but independent v_mfma_f32_32x32x16_bf16 out of registers.
//   grid = number of workgroups (one per CU up to 256), waves per workgroup 4 (one per SIMD, 16 accumulators) or 8 (two per SIMD, 8 accumulators each)
// hipcc --offload-arch=gfx950 -O3 -o probe_mfma probe_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int NACC>
__global__ __launch_bounds__(NACC == 16 ? 256 : 512) void mfma_only(float* out, int iters) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const uint32_t tv = 0x3c003c00u | (threadIdx.x & 0x7f);       // normal bf16 values (denormal operands ran ~40x slower in the first version)
  const uint4 a = make_uint4(tv, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u), b = make_uint4(0x3c003c00u, tv, 0x3c003c00u, 0x3c003c00u);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
  }
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) v += acc[i][e];
  if (v == 12345.f) out[threadIdx.x] = v;
}

int main() {
  float* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int waves : {4, 8})
    for (int grid : {32, 64, 128, 256, 512}) {
      auto launch = [&]() {
        if (waves == 4) hipLaunchKernelGGL(mfma_only<16>, dim3(grid), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL(mfma_only<8>, dim3(grid), dim3(512), 0, 0, out, iters * 2);
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)grid * 4 * 16 * iters * 32768.0;      // per SIMD: 16 MFMAs x iters (two waves: 8 x 2 iters)
      printf("waves/CU %d  workgroups %3d: %.2f ms  %.0f TFLOP/s  = %.0f GFLOP/s per CU in use (nominal 9766)\n", waves, grid, ms, flop / (ms * 1e-3) / 1e12,
             flop / (ms * 1e-3) / 1e9 / (grid < 256 ? grid : 256));
    }
  return 0;
}
