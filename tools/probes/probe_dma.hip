// Micro-benchmark (MI355X): per-CU rate of L2-resident operand staging, the quantity that bounds the conv GEMM loop.
//   mode 0: global_load_lds_dwordx4 (LDS-DMA), W waves issuing, counted vmcnt ring
//   mode 1: global_load_dwordx4 to VGPRs (results consumed by a trivial XOR), same addresses
//   mode 2: half of the bytes by LDS-DMA and the same amount again by VGPR loads (are the two paths additive?)
// Every workgroup (1 per CU, `waves` waves) streams the same `kb` KiB region (L2 resident) `iters` times.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void dma16(const void* g, uint32_t l) {
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(l)), "v"(g) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE>
__global__ __launch_bounds__(1024) void probe(const char* src, int kb, int iters, uint32_t* sink) {
  __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
  const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)lds;
  const int chunks = kb * 1024 / 1024;             // 1 KiB pieces (one wave instruction each)
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    for (int c = wid; c < chunks; c += nw * 4) {   // 4 pieces per wave per round
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cc = c + u * nw;
        if (cc >= chunks) break;
        const char* g = src + (size_t)cc * 1024 + lane * 16;
        if (MODE == 0 || (MODE == 2 && (u & 1) == 0)) dma16(g, lbase + ((cc * 1024) & 0xffff));
        else {
          const uint4 v = *reinterpret_cast<const uint4*>(g);
          acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
      }
      if (MODE != 1) wait_vm<4>();
    }
  }
  wait_vm<0>();
  if (acc.x == 0x12345 && acc.y == 7) sink[tid] = acc.z ^ acc.w ^ lds[tid];
}

int main(int argc, char** argv) {
  const int kb = 256, iters = 200;
  char* d; uint32_t* sink;
  hipMalloc(&d, (size_t)kb * 1024 + 4096); hipMalloc(&sink, 4096 * 4);
  hipMemset(d, 1, (size_t)kb * 1024 + 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode)
    for (int waves : {1, 2, 4, 8, 12, 16}) {
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(waves * 64), 0, 0, d, kb, iters, sink);
        else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(waves * 64), 0, 0, d, kb, iters, sink);
        else hipLaunchKernelGGL(probe<2>, dim3(256), dim3(waves * 64), 0, 0, d, kb, iters, sink);
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)kb * 1024 * iters;           // per CU
      printf("mode %d waves %2d: %.3f ms  %.1f GB/s per CU  %.2f TB/s chip  (%.1f B/clk/CU at 2.1 GHz)\n", mode, waves, ms,
             bytes / ms / 1e6, bytes * 256 / ms / 1e9, bytes / (ms * 1e-3) / 2.1e9);
    }
  return 0;
}
