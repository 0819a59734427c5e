// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (used to design the bf16 WGRAD fragment reads).
// LDS holds element value = its element index; every lane reads with a chosen address; prints what each lane got.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(uint16_t* out, int pitch_elems) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  // lane i of each 16-lane group points at row (i>>2), cols 4*(i&3).. of block g (4 rows x 16 cols), block g at rows 4g
  const int g = l >> 4, i = l & 15;
  const int elem = (4 * g + (i >> 2)) * pitch_elems + 4 * (i & 3);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + elem));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  for (int pitch : {16, 64}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pitch);
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pitch %d\n", pitch);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l*4+j] / pitch, h[l*4+j] % pitch); printf("\n"); }
  }
  return 0;
}
