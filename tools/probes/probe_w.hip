// Micro-benchmark for the next wide-GEMM design: dense bf16 GEMM C[M][N] = A[M][K] . B[N][K]^T with a 256 x 256 tile per
// 256-thread workgroup, 4 waves x (128 x 128) accumulators (one wave per SIMD, accumulators in AGPRs), operands global -> VGPR -> LDS
// (ds_write_b128, XOR-swizzled 128-byte rows), two 64 KB LDS buffers, ONE barrier per 64-deep K tile.  No conv addressing, trivial epilogue: it
// answers "what does the loop structure reach", nothing else.   hipcc --offload-arch=gfx950 -O3 -o probe_w probe_w.hip ; ./probe_w [M K N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int MODE>   // 0: everything; 1: no MFMA; 2: no global loads / LDS writes (first tile only); 3: MFMAs only (fragments read once per tile)
__global__ __launch_bounds__(256) void probe_w(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, uint16_t* __restrict__ C, int M, int K, int N) {
  constexpr int BM = 256, BN = 256, KT = 64;
  constexpr int ABUF = BM * 128, BUF = (BM + BN) * 128;      // bytes
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm0 = (wid >> 1) * 128, wn0 = (wid & 1) * 128;
  const int nn = N / BN, nm = M / BM, total = nm * nn, nkt = K / KT;
  const int lc = tid & 7, lr = tid >> 3;                    // load role: chunk lc of rows lr + 32 i
  const int frow = lane & 31, fh = lane >> 5;
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    const int mt = t / nn, nt = t % nn;
    const uint16_t* ga = A + (int64_t)(mt * BM + lr) * K + lc * 8;
    const uint16_t* gb = B + (int64_t)(nt * BN + lr) * K + lc * 8;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    uint4 ra[8], rb[8];
    auto gload = [&](int kt) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ra[i] = *reinterpret_cast<const uint4*>(ga + (int64_t)(32 * i) * K + kt * KT);
        rb[i] = *reinterpret_cast<const uint4*>(gb + (int64_t)(32 * i) * K + kt * KT);
      }
    };
    auto lstore = [&](int buf) {
      char* base = smem + buf * BUF;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = lr + 32 * i;
        *reinterpret_cast<uint4*>(base + row * 128 + ((lc ^ (row & 7)) << 4)) = ra[i];
        *reinterpret_cast<uint4*>(base + ABUF + row * 128 + ((lc ^ (row & 7)) << 4)) = rb[i];
      }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    for (int p = 0; p < nkt; ++p) {
      const char* ab = smem + (p & 1) * BUF;
      const char* bb = ab + ABUF;
      if (MODE < 2 && p + 1 < nkt) gload(p + 1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        uint4 af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (MODE == 3) { af[i] = make_uint4(p, s, i, lane); bf[i] = make_uint4(lane, i, s, p); continue; }
          const int ra_ = wm0 + i * 32 + frow, rb_ = wn0 + i * 32 + frow;
          af[i] = *reinterpret_cast<const uint4*>(ab + ra_ * 128 + (((2 * s + fh) ^ (ra_ & 7)) << 4));
          bf[i] = *reinterpret_cast<const uint4*>(bb + rb_ * 128 + (((2 * s + fh) ^ (rb_ & 7)) << 4));
        }
        if (MODE != 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) { acc[i][0][0] += __builtin_bit_cast(float, af[i].x ^ bf[i].y); }
        }
      }
      if (MODE < 2 && p + 1 < nkt) lstore((p + 1) & 1);
      if (MODE != 3) __syncthreads();
    }
    // trivial epilogue: one value per accumulator register pair (keeps the accumulators alive, writes 1/8 of C)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) v += acc[i][j][e];
        C[(int64_t)(mt * BM + wm0 + i * 32 + (lane & 31)) * N + nt * BN + wn0 + j * 32 + (lane >> 5)] = (uint16_t)(__builtin_bit_cast(uint32_t, v) >> 16);
      }
  }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 247296, K = argc > 2 ? atoi(argv[2]) : 1280, N = argc > 3 ? atoi(argv[3]) : 256;
  uint16_t *A, *B, *C;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2);
  hipMemset(A, 0x3c, (size_t)M * K * 2); hipMemset(B, 0x3c, (size_t)N * K * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 4; ++mode) {
    auto launch = [&]() {
      if (mode == 0) hipLaunchKernelGGL(probe_w<0>, dim3(256), dim3(256), 0, 0, A, B, C, M, K, N);
      else if (mode == 1) hipLaunchKernelGGL(probe_w<1>, dim3(256), dim3(256), 0, 0, A, B, C, M, K, N);
      else if (mode == 2) hipLaunchKernelGGL(probe_w<2>, dim3(256), dim3(256), 0, 0, A, B, C, M, K, N);
      else hipLaunchKernelGGL(probe_w<3>, dim3(256), dim3(256), 0, 0, A, B, C, M, K, N);
    };
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("probe_w mode %d  M %d K %d N %d: %.1f us  %.0f TFLOP/s  (%s)\n", mode, M, K, N, ms * 1e3, 2.0 * M * K * N / (ms * 1e-3) / 1e12,
           mode == 0 ? "all" : mode == 1 ? "no MFMA" : mode == 2 ? "no loads" : "MFMAs only, operands from registers");
  }
  return 0;
}
