// LMS log-mel perceptual loss (reference tools_for_loss.py:120-249, models.py:306-312) as two fused kernels.
// One workgroup per (utterance, row): the reference re-views the contiguous [257][T] magnitude array as [T][257]
// WITHOUT transposing, so "row j" is simply the 257 consecutive floats at flat offset j*257 - reproduced as is.
// Mel banks are sparse triangles (<= ~80 taps per band, 112 bands over the 3 scales): staged in LDS as (start, len, offset).
#include <hip/hip_runtime.h>
#include "../../include/sefd.h"
#include "dev_common.h"

namespace {
using namespace sefd;
constexpr int kMaxBands = 256, kMaxNF = 1024, kMaxScales = 8;

struct LmsArgs {
  const float *cr, *ci, *er, *ei;      // clean / estimate spectra [B][NF][T]; ci / ei may be null (magnitudes given)
  const int32_t* bands;                // [nbands][4] = start, len, weight offset, scale index
  const float* weights;
  float* rowloss;                      // [B*T]
  const float* gscale;                 // backward: upstream gradient (may be null -> 1)
  float *ger, *gei;                    // backward outputs [B][NF][T]
  int B, NF, T, nbands, nscales, nfft;
  int nb[kMaxScales];
};

template <bool BWD>
__global__ __launch_bounds__(128) void lms_kernel(const LmsArgs a) {
  __shared__ float pc[kMaxNF], pe[kMaxNF], fc[kMaxBands], fe[kMaxBands], qe[kMaxBands], dq[kMaxBands], rm[kMaxScales];
  __shared__ int32_t bnd[kMaxBands][4];
  const int j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int64_t base = ((int64_t)b * a.NF) * a.T + (int64_t)j * a.NF;      // flat re-view: row j of [T][NF]
  const float inv = 1.f / (float)a.nfft;
  for (int n = tid; n < a.nbands; n += 128) { bnd[n][0] = a.bands[4 * n]; bnd[n][1] = a.bands[4 * n + 1]; bnd[n][2] = a.bands[4 * n + 2]; bnd[n][3] = a.bands[4 * n + 3]; }
  for (int c = tid; c < a.NF; c += 128) {
    const float cr = a.cr[base + c], er = a.er[base + c];
    pc[c] = (a.ci ? sqrtf(cr * cr + a.ci[base + c] * a.ci[base + c] + 1e-7f) : cr) * inv;
    pe[c] = (a.ei ? sqrtf(er * er + a.ei[base + c] * a.ei[base + c] + 1e-7f) : er) * inv;
  }
  __syncthreads();
  for (int n = tid; n < a.nbands; n += 128) {
    float sc = 0.f, se = 0.f;
    const float* w = a.weights + bnd[n][2];
    for (int k = 0; k < bnd[n][1]; ++k) { sc += pc[bnd[n][0] + k] * w[k]; se += pe[bnd[n][0] + k] * w[k]; }
    fc[n] = logf(sc + 1e-7f);
    fe[n] = logf(se + 1e-7f);
    qe[n] = se;
  }
  __syncthreads();
  if (tid < a.nscales) {
    float s = 0.f;
    for (int n = 0; n < a.nbands; ++n)
      if (bnd[n][3] == tid) { const float e = fe[n] - fc[n]; s += e * e; }
    rm[tid] = sqrtf(s / (float)a.nb[tid] + 1e-7f);
  }
  __syncthreads();
  if (!BWD) {
    if (tid == 0) {
      float v = 0.f;
      for (int s = 0; s < a.nscales; ++s) v += rm[s];
      a.rowloss[(int64_t)b * a.T + j] = v / (float)a.nscales;
    }
    return;
  }
  const float gs = (a.gscale ? a.gscale[0] : 1.f) / ((float)a.nscales * (float)a.T * (float)a.B);
  for (int n = tid; n < a.nbands; n += 128) {
    const int s = bnd[n][3];
    dq[n] = gs * (fe[n] - fc[n]) / ((float)a.nb[s] * rm[s]) / (qe[n] + 1e-7f);
  }
  __syncthreads();
  for (int c = tid; c < a.NF; c += 128) {
    float dp = 0.f;
    for (int n = 0; n < a.nbands; ++n) {
      const int k = c - bnd[n][0];
      if (k >= 0 && k < bnd[n][1]) dp += a.weights[bnd[n][2] + k] * dq[n];
    }
    const float dm = dp * inv;
    if (a.ei) {
      const float er = a.er[base + c], ei = a.ei[base + c];
      const float mag = sqrtf(er * er + ei * ei + 1e-7f);
      a.ger[base + c] = dm * er / mag;
      a.gei[base + c] = dm * ei / mag;
    } else {
      a.ger[base + c] = dm;
    }
  }
}

__global__ void lms_reduce_kernel(const float* rowloss, int64_t n, float* out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += rowloss[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] = (float)(red[0] / (double)n);
}

int fill(LmsArgs& a, const float* cr, const float* ci, const float* er, const float* ei, int B, int NF, int T, const int32_t* bands,
         const float* weights, int nbands, const int32_t* scale_sizes, int nscales, int nfft) {
  if (NF > kMaxNF || nbands > kMaxBands || nscales > kMaxScales || (NF * T) % NF != 0) return -1;
  a.cr = cr; a.ci = ci; a.er = er; a.ei = ei; a.bands = bands; a.weights = weights;
  a.B = B; a.NF = NF; a.T = T; a.nbands = nbands; a.nscales = nscales; a.nfft = nfft;
  for (int s = 0; s < nscales; ++s) a.nb[s] = scale_sizes[s];
  return 0;
}
}  // namespace

extern "C" {
int32_t sefd_lms_forward(const float* clean_r, const float* clean_i, const float* est_r, const float* est_i, int32_t B, int32_t NF, int32_t T,
                         const int32_t* bands, const float* weights, int32_t nbands, const int32_t* scale_sizes_host, int32_t nscales,
                         int32_t nfft, float* rowloss_ws, float* loss_out, void* stream) {
  LmsArgs a{};
  if (fill(a, clean_r, clean_i, est_r, est_i, B, NF, T, bands, weights, nbands, scale_sizes_host, nscales, nfft)) return -1;
  a.rowloss = rowloss_ws;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL((lms_kernel<false>), dim3(T, B), dim3(128), 0, st, a);
  hipLaunchKernelGGL(lms_reduce_kernel, dim3(1), dim3(256), 0, st, rowloss_ws, (int64_t)B * T, loss_out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
int32_t sefd_lms_backward(const float* clean_r, const float* clean_i, const float* est_r, const float* est_i, int32_t B, int32_t NF, int32_t T,
                          const int32_t* bands, const float* weights, int32_t nbands, const int32_t* scale_sizes_host, int32_t nscales,
                          int32_t nfft, const float* grad_scale, float* grad_est_r, float* grad_est_i, void* stream) {
  LmsArgs a{};
  if (fill(a, clean_r, clean_i, est_r, est_i, B, NF, T, bands, weights, nbands, scale_sizes_host, nscales, nfft)) return -1;
  a.gscale = grad_scale; a.ger = grad_est_r; a.gei = grad_est_i;
  hipLaunchKernelGGL((lms_kernel<true>), dim3(T, B), dim3(128), 0, reinterpret_cast<hipStream_t>(stream), a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
}
