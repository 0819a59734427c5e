// Fused STFT kernel: the HBM-bound front end of every model (reference ConvSTFT, tools_for_model.py:54-61, computes the same
// numbers as a 99 MMAC/utterance conv1d; as an FFT it is 2.8 MMAC and the kernel is bound by reading 4 B/sample and writing
// 2 x 257 x 4 B per frame).  One 64-lane wavefront per frame, 8 points per lane:
//   512 = 8 (registers) x 8 x 8 (two LDS transposes inside the wave's private 4 KiB LDS slice; no workgroup barrier).
//   n = l + 64 j :  Y_l[k0] = DFT8_j ;  Z = Y * W512^(l k0)
//   l = l0 + 8 l1:  A[ma] = DFT8_l1(Z) * W64^(l0 ma) ;  X[k0 + 8 ma + 64 mb] = DFT8_l0(A)[mb]
#include <hip/hip_runtime.h>
#include "sefd_desc.h"
#include "dev_common.h"

namespace sefd {

struct cf { float x, y; };
__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cf cmul(cf a, cf b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cf mulmi(cf a) { return {a.y, -a.x}; }          // a * (-i)

__device__ __forceinline__ void dft4(cf y0, cf y1, cf y2, cf y3, cf* o) {
  const cf t0 = cadd(y0, y2), t1 = csub(y0, y2), t2 = cadd(y1, y3), t3 = mulmi(csub(y1, y3));
  o[0] = cadd(t0, t2); o[1] = cadd(t1, t3); o[2] = csub(t0, t2); o[3] = csub(t1, t3);
}
// X[k] = sum_j x[j] exp(-2 pi i j k / 8)
__device__ __forceinline__ void dft8(const cf* x, cf* X) {
  cf E[4], O[4];
  dft4(x[0], x[2], x[4], x[6], E);
  dft4(x[1], x[3], x[5], x[7], O);
  const float h = 0.70710678118654752f;
  const cf w1 = {h, -h}, w3 = {-h, -h};
  const cf o1 = cmul(O[1], w1), o2 = mulmi(O[2]), o3 = cmul(O[3], w3);
  X[0] = cadd(E[0], O[0]); X[4] = csub(E[0], O[0]);
  X[1] = cadd(E[1], o1);   X[5] = csub(E[1], o1);
  X[2] = cadd(E[2], o2);   X[6] = csub(E[2], o2);
  X[3] = cadd(E[3], o3);   X[7] = csub(E[3], o3);
}

__global__ __launch_bounds__(256) void stft_fft_kernel(const StftFft d, const ArenaBases ab) {
  __shared__ float2 lds[4][512];
  __shared__ float2 twl[512];
  const float* src = reinterpret_cast<const float*>(rp(ab, d.src));
  const float* win = reinterpret_cast<const float*>(rp(ab, d.win));
  const float2* tw = reinterpret_cast<const float2*>(rp(ab, d.tw));
  float2* spec = reinterpret_cast<float2*>(rp(ab, d.spec));
  for (int i = threadIdx.x; i < 512; i += 256) twl[i] = tw[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t fr = (int64_t)blockIdx.x * 4 + wv;
  if (fr >= (int64_t)d.B * d.T) return;                      // wave-uniform
  const int64_t b = fr / d.T;
  const int t = (int)(fr - b * d.T);
  float2* buf = lds[wv];
  // ---- stage 1: 8 windowed samples per lane, stride 64
  cf x[8], X[8];
  const int p0 = t * d.hop - d.off;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = lane + 64 * j, p = p0 + n;
    const float v = (p >= 0 && p < d.L) ? src[b * d.L + p] * win[n] : 0.f;
    x[j] = {v, 0.f};
  }
  dft8(x, X);
#pragma unroll
  for (int k0 = 0; k0 < 8; ++k0) {
    const float2 w = twl[(lane * k0) & 511];
    const cf z = cmul(X[k0], cf{w.x, -w.y});                 // W512^(l k0) = cos - i sin
    buf[k0 * 64 + lane] = make_float2(z.x, z.y);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // wave-private LDS slice: ordering inside the wave is enough
  // ---- stage 2: DFT over l1 for (k0, l0)
  const int k0 = lane >> 3, l0 = lane & 7;
#pragma unroll
  for (int l1 = 0; l1 < 8; ++l1) { const float2 v = buf[k0 * 64 + l0 + 8 * l1]; x[l1] = {v.x, v.y}; }
  dft8(x, X);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int ma = 0; ma < 8; ++ma) {
    const float2 w = twl[(8 * l0 * ma) & 511];
    const cf z = cmul(X[ma], cf{w.x, -w.y});                 // W64^(l0 ma)
    buf[k0 * 64 + ma * 8 + l0] = make_float2(z.x, z.y);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // ---- stage 3: DFT over l0 for (k0, ma) -> bins k0 + 8 ma + 64 mb
  const int ma = lane & 7;
#pragma unroll
  for (int q = 0; q < 8; ++q) { const float2 v = buf[k0 * 64 + ma * 8 + q]; x[q] = {v.x, v.y}; }
  dft8(x, X);
  float2* out = spec + fr * 258;
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) out[1 + k0 + 8 * ma + 64 * mb] = make_float2(X[mb].x, X[mb].y);
  if (lane == 0) { out[1 + 256] = make_float2(X[4].x, X[4].y); out[0] = make_float2(0.f, 0.f); }
}

void launch_stft_fft(const StftFft& d, const ArenaBases& ab, hipStream_t st) {
  const int64_t frames = (int64_t)d.B * d.T;
  hipLaunchKernelGGL(stft_fft_kernel, dim3((unsigned)((frames + 3) / 4)), dim3(256), 0, st, d, ab);
}

}  // namespace sefd
