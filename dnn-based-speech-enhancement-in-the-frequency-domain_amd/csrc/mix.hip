// On-GPU SNR mixing of a batch (reference generate_noisy_data.py:46-67, done there once, offline, with numpy):
//   noise segment = noise[start : start + L];  powers after removing the DC bias;  alpha = sqrt(10^(-snr/10) * P_speech / (P_noise + 1e-6));
//   noisy = speech + alpha * segment, optionally through the reference's int16 file round trip ((x * 32768).astype(int16) / 32768).
// Two HBM-bound passes: per-utterance sums (one workgroup per utterance, doubles), then the mix.
#include <hip/hip_runtime.h>
#include "../../include/sefd.h"
#include "dev_common.h"

namespace {
__global__ __launch_bounds__(256) void mix_stats_kernel(const float* speech, const float* noise, const int64_t* nstart, int L, double* stats) {
  const int b = blockIdx.x;
  const float* s = speech + (int64_t)b * L;
  const float* n = noise + nstart[b];
  double a0 = 0, a1 = 0, b0 = 0, b1 = 0;
  for (int i = threadIdx.x; i < L; i += 256) {
    const double x = s[i], y = n[i];
    a0 += x; a1 += x * x; b0 += y; b1 += y * y;
  }
  __shared__ double sh[4][4];
  double v[4] = {a0, a1, b0, b1};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < 4) stats[b * 4 + threadIdx.x] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}
__global__ __launch_bounds__(256) void mix_apply_kernel(const float* speech, const float* noise, const int64_t* nstart, const float* snr_db, int B, int L,
                                                        const double* stats, int quantize, float* noisy) {
  const int64_t total = (int64_t)B * L;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / L);
    const int j = (int)(i - (int64_t)b * L);
    const double ms = stats[b * 4] / L, mn = stats[b * 4 + 2] / L;
    const double ps = stats[b * 4 + 1] / L - ms * ms, pn = stats[b * 4 + 3] / L - mn * mn;      // mean((x - mean)^2)
    const double alpha = sqrt(pow(10.0, -(double)snr_db[b] / 10.0) * ps / (pn + 1e-6));
    double v = (double)speech[i] + alpha * (double)noise[nstart[b] + j];
    if (quantize) {                                   // (v * 32768).astype(np.int16): truncation toward zero, two's-complement wrap
      const long long q = (long long)(v * 32768.0);
      v = (double)(short)(q & 0xffff) / 32768.0;
    }
    noisy[i] = (float)v;
  }
}
}  // namespace

extern "C" int32_t sefd_mix_snr(const float* speech, const float* noise, const int64_t* noise_start, const float* snr_db, int32_t B, int32_t L,
                                int32_t quantize, double* ws, float* noisy, void* stream) {
  if (!speech || !noise || !noise_start || !snr_db || !ws || !noisy || B <= 0 || L <= 0) return -1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(mix_stats_kernel, dim3(B), dim3(256), 0, st, speech, noise, noise_start, L, ws);
  const int64_t total = (int64_t)B * L;
  hipLaunchKernelGGL(mix_apply_kernel, dim3((unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256)), dim3(256), 0, st, speech, noise, noise_start,
                     snr_db, B, L, ws, quantize, noisy);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
