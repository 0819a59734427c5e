// ComplexBatchNorm (training and eval) + PReLU on channels-last rows [R][C]: DCCRN(use_cbn=True), reference tools_for_model.py:430-607
// (whitening of every complex channel by the inverse square root of its 2 x 2 covariance, then a 2 x 2 affine map W, B) followed by the layer's
// nn.PReLU (models.py:76-78, 120-122).  Channel k < h = C / 2 is the real part and k + h the imaginary part of complex channel k.
//
// Forward:   STATS    y -> per-block sums of xr, xi, xr^2, xi^2, xr xi                                  (one read of y)
//            FINALIZE sums -> M, V (+eps), U = V^-1/2, Z = W U, b' = B - Z M; running statistics lerp    (fp64, one workgroup per complex channel)
//            APPLY    z = prelu(Z x + b')                                                              (read y, write z)
// Backward:  with dbn = prelu'(bn) dz (bn recomputed from y), x~ = x - M, N rows:
//            REDUCE   per-block sums of dbn (2), dbn x~^T (4), slope gradient share
//            FINALIZE dB = sum dbn; dZ = sum dbn x~^T; dW = dZ U (symmetric W: the two off-diagonal entries add); dU = W dZ (symmetric U likewise);
//                     dV through U = f(V) (square root of a 2 x 2 matrix and its inverse, the reference's closed form); coefficients of
//            APPLY    dx = Z^T (dbn - mean dbn) + (1 / N) [[2 dVrr, dVri], [dVri, 2 dVii]] x~
// All passes move 8-byte (bf16) / 16-byte (fp32) pieces of 4 complex channels per lane; h % 4 == 0 is required by the planner.
#include <hip/hip_runtime.h>
#include "sefd_desc.h"
#include "dev_common.h"

namespace sefd {

namespace {

template <typename T> struct Io4;
template <> struct Io4<float> {
  static __device__ __forceinline__ void load(const char* base, int64_t i, float* o) {
    const float4 v = *reinterpret_cast<const float4*>(base + i * 4);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  static __device__ __forceinline__ void store(char* base, int64_t i, const float* o) { *reinterpret_cast<float4*>(base + i * 4) = make_float4(o[0], o[1], o[2], o[3]); }
};
template <> struct Io4<bf16_t> {
  static __device__ __forceinline__ void load(const char* base, int64_t i, float* o) {
    const uint2 v = *reinterpret_cast<const uint2*>(base + i * 2);
    o[0] = bf2f(v.x & 0xffff); o[1] = bf2f(v.x >> 16); o[2] = bf2f(v.y & 0xffff); o[3] = bf2f(v.y >> 16);
  }
  static __device__ __forceinline__ void store(char* base, int64_t i, const float* o) {
    *reinterpret_cast<uint2*>(base + i * 2) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
  }
};

__device__ __forceinline__ void ld4(const float* p, float* o) { const float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }

// the 2 x 2 affine map of 4 complex channels starting at k0
struct Coef4 {
  float zrr[4], zri[4], zir[4], zii[4], br[4], bi[4], mr[4], mi[4];
  __device__ __forceinline__ void load(const float* coef, int h, int k0, bool want_mean) {
    ld4(coef + 0 * h + k0, zrr); ld4(coef + 1 * h + k0, zri); ld4(coef + 2 * h + k0, zir); ld4(coef + 3 * h + k0, zii);
    ld4(coef + 4 * h + k0, br); ld4(coef + 5 * h + k0, bi);
    if (want_mean) { ld4(coef + 6 * h + k0, mr); ld4(coef + 7 * h + k0, mi); }
  }
};

// upstream gradient of y row (b, ql) (ql = row inside the batch item), 4 channels at element column c: BnBwdReduce's convention
template <typename T>
__device__ __forceinline__ void load_dz4(const CbnBwd& d, const char* dz0, const char* dz1, int64_t b, int ql, int c, float* g) {
#pragma unroll
  for (int e = 0; e < 4; ++e) g[e] = 0.f;
  if (ql >= d.skip) Io4<T>::load(dz0, (b * (d.rpb - d.skip) + ql - d.skip) * d.C + c, g);
  if (dz1) {
    float t[4];
    Io4<T>::load(dz1, (b * d.rpb + ql) * d.C + c, t);
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] += t[e];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cbn_stats_kernel(const CbnFwd d, const ArenaBases ab) {
  __shared__ float red[5 * 1024];
  const int h = d.C / 2, hq = h / 4;
  const int nrl = 256 / hq > 0 ? 256 / hq : 1;
  const int rl = threadIdx.x / hq, cc = threadIdx.x - rl * hq;
  const char* y = rp(ab, d.y);
  const int64_t row0 = (int64_t)blockIdx.x * d.rows_per_blk, row1 = min(d.R, row0 + d.rows_per_blk);
  float s[5][4];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) s[j][e] = 0.f;
  if (rl < nrl) {
    for (int64_t r = row0 + rl; r < row1; r += nrl) {
      float xr[4], xi[4];
      Io4<T>::load(y, r * d.C + 4 * cc, xr);
      Io4<T>::load(y, r * d.C + h + 4 * cc, xi);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[0][e] += xr[e]; s[1][e] += xi[e]; s[2][e] += xr[e] * xr[e]; s[3][e] += xi[e] * xi[e]; s[4][e] += xr[e] * xi[e]; }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[(j * nrl + rl) * h + 4 * cc + e] = s[j][e];
  }
  __syncthreads();
  float* part = reinterpret_cast<float*>(rp(ab, d.part)) + (int64_t)blockIdx.x * 5 * h;
  for (int i = threadIdx.x; i < 5 * h; i += 256) {
    const int j = i / h, k = i - j * h;
    float t = 0.f;
    for (int q = 0; q < nrl; ++q) t += red[(j * nrl + q) * h + k];
    part[i] = t;
  }
}

// per-block rows (part[b][j][k], j < NS, `ld` floats per block) -> fp64 totals tot[j] of complex channel k, fixed order
template <int NS>
__device__ __forceinline__ void block_totals(const float* part, int nblk, int ld, int h, int k, double* tot) {
  __shared__ double r[NS][256];
  double s[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) s[j] = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256)
#pragma unroll
    for (int j = 0; j < NS; ++j) s[j] += part[(int64_t)b * ld + j * h + k];
#pragma unroll
  for (int j = 0; j < NS; ++j) r[j][threadIdx.x] = s[j];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o)
#pragma unroll
      for (int j = 0; j < NS; ++j) r[j][threadIdx.x] += r[j][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0)
#pragma unroll
    for (int j = 0; j < NS; ++j) tot[j] = r[j][0];
  __syncthreads();
}

// U = V^-1/2 of the symmetric positive definite [[vrr, vri], [vri, vii]] (tools_for_model.py:567-576)
__device__ __forceinline__ void inv_sqrt_2x2(double vrr, double vri, double vii, double& urr, double& uri, double& uii, double& s, double& t, double& rst) {
  const double tau = vrr + vii, delta = vrr * vii - vri * vri;
  s = sqrt(delta);
  t = sqrt(tau + 2 * s);
  rst = 1.0 / (s * t);
  urr = (s + vii) * rst;
  uii = (s + vrr) * rst;
  uri = -vri * rst;
}

__global__ __launch_bounds__(256) void cbn_finalize_kernel(const CbnFwd d, const ArenaBases ab) {
  const int k = blockIdx.x, h = d.C / 2;
  __shared__ double tot[5];
  double mr, mi, vrr, vri, vii;
  float* rmr = d.RM[0].arena >= 0 ? reinterpret_cast<float*>(rp(ab, d.RM[0])) : nullptr;
  float* rmi = d.RM[1].arena >= 0 ? reinterpret_cast<float*>(rp(ab, d.RM[1])) : nullptr;
  float* rv0 = d.RV[0].arena >= 0 ? reinterpret_cast<float*>(rp(ab, d.RV[0])) : nullptr;
  float* rv1 = d.RV[1].arena >= 0 ? reinterpret_cast<float*>(rp(ab, d.RV[1])) : nullptr;
  float* rv2 = d.RV[2].arena >= 0 ? reinterpret_cast<float*>(rp(ab, d.RV[2])) : nullptr;
  if (d.training) {
    block_totals<5>(reinterpret_cast<const float*>(rp(ab, d.part)), d.nblk, 5 * h, h, k, tot);
    if (threadIdx.x != 0) return;
    mr = tot[0] / d.count; mi = tot[1] / d.count;
    vrr = tot[2] / d.count - mr * mr; vii = tot[3] / d.count - mi * mi; vri = tot[4] / d.count - mr * mi;
    if (vrr < 0) vrr = 0;
    if (vii < 0) vii = 0;
    if (rmr) {                                   // Tensor.lerp_(new, momentum); the covariance goes in WITHOUT eps and biased (tools_for_model.py:541-556)
      rmr[k] += d.momentum * ((float)mr - rmr[k]); rmi[k] += d.momentum * ((float)mi - rmi[k]);
      rv0[k] += d.momentum * ((float)vrr - rv0[k]); rv1[k] += d.momentum * ((float)vri - rv1[k]); rv2[k] += d.momentum * ((float)vii - rv2[k]);
    }
  } else {
    if (threadIdx.x != 0) return;
    mr = rmr[k]; mi = rmi[k]; vrr = rv0[k]; vri = rv1[k]; vii = rv2[k];
  }
  vrr += d.eps; vii += d.eps;
  double urr, uri, uii, s, t, rst;
  inv_sqrt_2x2(vrr, vri, vii, urr, uri, uii, s, t, rst);
  const double wrr = reinterpret_cast<const float*>(rp(ab, d.W[0]))[k], wri = reinterpret_cast<const float*>(rp(ab, d.W[1]))[k],
               wii = reinterpret_cast<const float*>(rp(ab, d.W[2]))[k];
  const double zrr = wrr * urr + wri * uri, zri = wrr * uri + wri * uii, zir = wri * urr + wii * uri, zii = wri * uri + wii * uii;
  const double br = reinterpret_cast<const float*>(rp(ab, d.Bv[0]))[k], bi = reinterpret_cast<const float*>(rp(ab, d.Bv[1]))[k];
  float* coef = reinterpret_cast<float*>(rp(ab, d.coef));
  coef[0 * h + k] = (float)zrr; coef[1 * h + k] = (float)zri; coef[2 * h + k] = (float)zir; coef[3 * h + k] = (float)zii;
  coef[4 * h + k] = (float)(br - zrr * mr - zri * mi); coef[5 * h + k] = (float)(bi - zir * mr - zii * mi);
  coef[6 * h + k] = (float)mr; coef[7 * h + k] = (float)mi;
  coef[8 * h + k] = (float)urr; coef[9 * h + k] = (float)uri; coef[10 * h + k] = (float)uii;
  coef[11 * h + k] = (float)vrr; coef[12 * h + k] = (float)vri; coef[13 * h + k] = (float)vii;
}

template <typename T>
__global__ __launch_bounds__(256) void cbn_apply_kernel(const CbnFwd d, const ArenaBases ab) {
  const int h = d.C / 2, hq = h / 4;
  const char* y = rp(ab, d.y);
  char* z = rp(ab, d.z);
  const float* coef = reinterpret_cast<const float*>(rp(ab, d.coef));
  const float a = *reinterpret_cast<const float*>(rp(ab, d.slope));
  const int64_t n = d.R * hq, stride = (int64_t)gridDim.x * 256;
  const bool fixed = stride % hq == 0;           // a thread meets the same 4 complex channels on every trip: coefficients stay in registers
  Coef4 cf;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (fixed && i < n) cf.load(coef, h, 4 * (int)(i % hq), false);
  for (; i < n; i += stride) {
    const int64_t r = i / hq;
    const int k0 = 4 * (int)(i - r * hq);
    if (!fixed) cf.load(coef, h, k0, false);
    float xr[4], xi[4], zr[4], zi[4];
    Io4<T>::load(y, r * d.C + k0, xr);
    Io4<T>::load(y, r * d.C + h + k0, xi);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float yr = cf.zrr[e] * xr[e] + cf.zri[e] * xi[e] + cf.br[e], yi = cf.zir[e] * xr[e] + cf.zii[e] * xi[e] + cf.bi[e];
      zr[e] = yr > 0.f ? yr : a * yr;
      zi[e] = yi > 0.f ? yi : a * yi;
    }
    Io4<T>::store(z, r * d.C + k0, zr);
    Io4<T>::store(z, r * d.C + h + k0, zi);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cbn_bwd_reduce_kernel(const CbnBwd d, const ArenaBases ab) {
  __shared__ float red[6 * 1024];
  __shared__ float reds[4];
  const int h = d.C / 2, hq = h / 4;
  const int nrl = 256 / hq > 0 ? 256 / hq : 1;
  const int rl = threadIdx.x / hq, cc = threadIdx.x - rl * hq;
  const char* y = rp(ab, d.y);
  const char* dz0 = rp(ab, d.dz0);
  const char* dz1 = d.dz1.arena >= 0 ? rp(ab, d.dz1) : nullptr;
  const float a = *reinterpret_cast<const float*>(rp(ab, d.slope));
  const int64_t row0 = (int64_t)blockIdx.x * d.rows_per_blk, row1 = min(d.R, row0 + d.rows_per_blk);
  float s[6][4], sa = 0.f;
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) s[j][e] = 0.f;
  if (rl < nrl) {
    Coef4 cf;
    cf.load(reinterpret_cast<const float*>(rp(ab, d.coef)), h, 4 * cc, true);
    for (int64_t r = row0 + rl; r < row1; r += nrl) {
      const int64_t b = r / d.rpb;
      const int ql = (int)(r - b * d.rpb);
      float xr[4], xi[4], gr[4], gi[4];
      Io4<T>::load(y, r * d.C + 4 * cc, xr);
      Io4<T>::load(y, r * d.C + h + 4 * cc, xi);
      load_dz4<T>(d, dz0, dz1, b, ql, 4 * cc, gr);
      load_dz4<T>(d, dz0, dz1, b, ql, h + 4 * cc, gi);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float yr = cf.zrr[e] * xr[e] + cf.zri[e] * xi[e] + cf.br[e], yi = cf.zir[e] * xr[e] + cf.zii[e] * xi[e] + cf.bi[e];
        const float dr = yr > 0.f ? gr[e] : a * gr[e], di = yi > 0.f ? gi[e] : a * gi[e];
        sa += (yr > 0.f ? 0.f : yr * gr[e]) + (yi > 0.f ? 0.f : yi * gi[e]);
        const float cr = xr[e] - cf.mr[e], ci = xi[e] - cf.mi[e];
        s[0][e] += dr; s[1][e] += di; s[2][e] += dr * cr; s[3][e] += dr * ci; s[4][e] += di * cr; s[5][e] += di * ci;
      }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[(j * nrl + rl) * h + 4 * cc + e] = s[j][e];
  }
  __syncthreads();
  float* part = reinterpret_cast<float*>(rp(ab, d.part)) + (int64_t)blockIdx.x * 7 * h;
  for (int i = threadIdx.x; i < 6 * h; i += 256) {
    const int j = i / h, k = i - j * h;
    float t = 0.f;
    for (int q = 0; q < nrl; ++q) t += red[(j * nrl + q) * h + k];
    part[i] = t;
  }
  sa = wave_sum(sa);
  if ((threadIdx.x & 63) == 0) reds[threadIdx.x >> 6] = sa;
  __syncthreads();
  if (threadIdx.x == 0) part[6 * h] = reds[0] + reds[1] + reds[2] + reds[3];
  for (int k = 1 + threadIdx.x; k < h; k += 256) part[6 * h + k] = 0.f;
}

__global__ __launch_bounds__(256) void cbn_bwd_finalize_kernel(const CbnBwd d, const ArenaBases ab) {
  const int k = blockIdx.x, h = d.C / 2;
  __shared__ double tot[7];
  const float* part = reinterpret_cast<const float*>(rp(ab, d.part));
  block_totals<6>(part, d.nblk, 7 * h, h, k, tot);
  if (k == 0) {                                  // the PReLU slope gradient is one scalar: block 0 adds the shares of all blocks (column 0 of row 6)
    __shared__ double rs[256];
    double sa = 0.0;
    for (int b = threadIdx.x; b < d.nblk; b += 256) sa += part[(int64_t)b * 7 * h + 6 * h];
    rs[threadIdx.x] = sa;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) rs[threadIdx.x] += rs[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) reinterpret_cast<float*>(rp(ab, d.dslope))[0] = (float)rs[0];
  }
  if (threadIdx.x != 0) return;
  const float* coef = reinterpret_cast<const float*>(rp(ab, d.coef));
  const double N = d.count;
  const double dbr = tot[0], dbi = tot[1], dzrr = tot[2], dzri = tot[3], dzir = tot[4], dzii = tot[5];
  const double urr = coef[8 * h + k], uri = coef[9 * h + k], uii = coef[10 * h + k];
  const double wrr = reinterpret_cast<const float*>(rp(ab, d.W[0]))[k], wri = reinterpret_cast<const float*>(rp(ab, d.W[1]))[k],
               wii = reinterpret_cast<const float*>(rp(ab, d.W[2]))[k];
  reinterpret_cast<float*>(rp(ab, d.dB[0]))[k] = (float)dbr;
  reinterpret_cast<float*>(rp(ab, d.dB[1]))[k] = (float)dbi;
  reinterpret_cast<float*>(rp(ab, d.dW[0]))[k] = (float)(dzrr * urr + dzri * uri);
  reinterpret_cast<float*>(rp(ab, d.dW[1]))[k] = (float)(dzrr * uri + dzri * uii + dzir * urr + dzii * uri);
  reinterpret_cast<float*>(rp(ab, d.dW[2]))[k] = (float)(dzir * uri + dzii * uii);
  // dU = W^T dZ with W = [[wrr, wri], [wri, wii]]; U is symmetric: its off-diagonal entry collects both positions
  const double durr = wrr * dzrr + wri * dzir, duii = wri * dzri + wii * dzii, duri = (wrr * dzri + wri * dzii) + (wri * dzrr + wii * dzir);
  // back through U(V + eps I): the forward scalars again from the saved covariance
  const double vrr = coef[11 * h + k], vri = coef[12 * h + k], vii = coef[13 * h + k];
  double u0, u1, u2, s, t, rst;
  inv_sqrt_2x2(vrr, vri, vii, u0, u1, u2, s, t, rst);
  double dvrr = 0, dvri = 0, dvii = 0;
  double d_rst = durr * (s + vii) + duii * (s + vrr) + duri * (-vri);
  double d_s = (durr + duii) * rst;
  dvii += durr * rst; dvrr += duii * rst; dvri += -duri * rst;
  const double d_st = -d_rst * rst * rst;         // rst = 1 / (s t)
  d_s += d_st * t;
  const double d_t = d_st * s;
  const double d_u = d_t / (2 * t);               // t = sqrt(tau + 2 s)
  d_s += 2 * d_u;
  const double d_tau = d_u;
  const double d_delta = d_s / (2 * s);           // s = sqrt(delta)
  dvrr += d_delta * vii + d_tau; dvii += d_delta * vrr + d_tau; dvri += -2 * vri * d_delta;
  float* cb = reinterpret_cast<float*>(rp(ab, d.coefb));
  cb[0 * h + k] = coef[0 * h + k]; cb[1 * h + k] = coef[1 * h + k]; cb[2 * h + k] = coef[2 * h + k]; cb[3 * h + k] = coef[3 * h + k];
  cb[4 * h + k] = (float)(dbr / N); cb[5 * h + k] = (float)(dbi / N);
  cb[6 * h + k] = (float)(2 * dvrr / N); cb[7 * h + k] = (float)(dvri / N); cb[8 * h + k] = (float)(2 * dvii / N);
}

// grid: x over the chunks of one batch item, y = batch item
template <typename T>
__global__ __launch_bounds__(256) void cbn_bwd_apply_kernel(const CbnBwd d, const ArenaBases ab) {
  const int h = d.C / 2, hq = h / 4;
  const char* y = rp(ab, d.y);
  const char* dz0 = rp(ab, d.dz0);
  const char* dz1 = d.dz1.arena >= 0 ? rp(ab, d.dz1) : nullptr;
  char* dy = rp(ab, d.dy);
  const float* coef = reinterpret_cast<const float*>(rp(ab, d.coef));
  const float* cb = reinterpret_cast<const float*>(rp(ab, d.coefb));
  const float a = *reinterpret_cast<const float*>(rp(ab, d.slope));
  const int64_t b = blockIdx.y;
  const int n = (int)(d.rpb * hq), stride = gridDim.x * 256;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int ql = i / hq, k0 = 4 * (i - ql * hq);
    Coef4 cf;
    cf.load(coef, h, k0, true);
    float mdr[4], mdi[4], qrr[4], qri[4], qii[4];
    ld4(cb + 4 * h + k0, mdr); ld4(cb + 5 * h + k0, mdi); ld4(cb + 6 * h + k0, qrr); ld4(cb + 7 * h + k0, qri); ld4(cb + 8 * h + k0, qii);
    const int64_t r = b * d.rpb + ql;
    float xr[4], xi[4], gr[4], gi[4], or_[4], oi[4];
    Io4<T>::load(y, r * d.C + k0, xr);
    Io4<T>::load(y, r * d.C + h + k0, xi);
    load_dz4<T>(d, dz0, dz1, b, ql, k0, gr);
    load_dz4<T>(d, dz0, dz1, b, ql, h + k0, gi);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float yr = cf.zrr[e] * xr[e] + cf.zri[e] * xi[e] + cf.br[e], yi = cf.zir[e] * xr[e] + cf.zii[e] * xi[e] + cf.bi[e];
      const float dr = (yr > 0.f ? gr[e] : a * gr[e]) - mdr[e], di = (yi > 0.f ? gi[e] : a * gi[e]) - mdi[e];
      const float cr = xr[e] - cf.mr[e], ci = xi[e] - cf.mi[e];
      or_[e] = cf.zrr[e] * dr + cf.zir[e] * di + qrr[e] * cr + qri[e] * ci;
      oi[e] = cf.zri[e] * dr + cf.zii[e] * di + qri[e] * cr + qii[e] * ci;
    }
    Io4<T>::store(dy, r * d.C + k0, or_);
    Io4<T>::store(dy, r * d.C + h + k0, oi);
  }
}

inline int gridcap(int64_t n, int cap = 16384) {
  const int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

void launch_cbn(const Op& op, const ArenaBases& ab, hipStream_t st) {
  const bool bf = (op.kind == OP_CBN_STATS || op.kind == OP_CBN_FINALIZE || op.kind == OP_CBN_APPLY ? op.cbf.dt : op.cbb.dt) == DT_BF16;
  switch (op.kind) {
    case OP_CBN_STATS:
      if (bf) hipLaunchKernelGGL((cbn_stats_kernel<bf16_t>), dim3(op.cbf.nblk), dim3(256), 0, st, op.cbf, ab);
      else hipLaunchKernelGGL((cbn_stats_kernel<float>), dim3(op.cbf.nblk), dim3(256), 0, st, op.cbf, ab);
      break;
    case OP_CBN_FINALIZE:
      hipLaunchKernelGGL(cbn_finalize_kernel, dim3(op.cbf.C / 2), dim3(256), 0, st, op.cbf, ab); break;
    case OP_CBN_APPLY: {
      const int g = gridcap(op.cbf.R * (op.cbf.C / 8));
      if (bf) hipLaunchKernelGGL((cbn_apply_kernel<bf16_t>), dim3(g), dim3(256), 0, st, op.cbf, ab);
      else hipLaunchKernelGGL((cbn_apply_kernel<float>), dim3(g), dim3(256), 0, st, op.cbf, ab);
      break;
    }
    case OP_CBN_BWD_REDUCE:
      if (bf) hipLaunchKernelGGL((cbn_bwd_reduce_kernel<bf16_t>), dim3(op.cbb.nblk), dim3(256), 0, st, op.cbb, ab);
      else hipLaunchKernelGGL((cbn_bwd_reduce_kernel<float>), dim3(op.cbb.nblk), dim3(256), 0, st, op.cbb, ab);
      break;
    case OP_CBN_BWD_FINALIZE:
      hipLaunchKernelGGL(cbn_bwd_finalize_kernel, dim3(op.cbb.C / 2), dim3(256), 0, st, op.cbb, ab); break;
    case OP_CBN_BWD_APPLY: {
      const int nb = (int)(op.cbb.R / op.cbb.rpb);
      const int gx = gridcap(op.cbb.rpb * (op.cbb.C / 8), 16384 / (nb > 0 ? nb : 1) + 1);
      if (bf) hipLaunchKernelGGL((cbn_bwd_apply_kernel<bf16_t>), dim3(gx, nb), dim3(256), 0, st, op.cbb, ab);
      else hipLaunchKernelGGL((cbn_bwd_apply_kernel<float>), dim3(gx, nb), dim3(256), 0, st, op.cbb, ab);
      break;
    }
    default: break;
  }
}

}  // namespace sefd
