// BatchNorm2d(train) + PReLU on channels-last rows [R][C] - the HBM-bound passes of the conv stack.
// 16-byte accesses for both storage dtypes (4 fp32 / 8 bf16 per lane), per-channel parameters staged once per
// workgroup in LDS, row -> (batch item, frame) decode kept incremental (no 64-bit divisions in the loops).
// Statistics themselves come for free from the producing GEMM's epilogue (rungemm.hip) -> bn_finalize (kernels.hip).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "sefd_desc.h"
#include "tuning.h"
#include "dev_common.h"

namespace sefd {

template <typename T> struct VecIO;
template <> struct VecIO<float> {
  static constexpr int V = 4;
  static __device__ __forceinline__ void load(const char* base, int64_t i, float* o) {
    const float4 v = *reinterpret_cast<const float4*>(base + i * 4);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  static __device__ __forceinline__ void store(char* base, int64_t i, const float* o) {
    *reinterpret_cast<float4*>(base + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
};
template <> struct VecIO<bf16_t> {
  static constexpr int V = 8;
  static __device__ __forceinline__ void load(const char* base, int64_t i, float* o) {
    const uint4 v = *reinterpret_cast<const uint4*>(base + i * 2);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { o[2 * k] = bf2f(w[k] & 0xffff); o[2 * k + 1] = bf2f(w[k] >> 16); }
  }
  static __device__ __forceinline__ void store(char* base, int64_t i, const float* o) {
    uint4 v;
    v.x = pack_bf16x2(o[0], o[1]); v.y = pack_bf16x2(o[2], o[3]);
    v.z = pack_bf16x2(o[4], o[5]); v.w = pack_bf16x2(o[6], o[7]);
    *reinterpret_cast<uint4*>(base + i * 2) = v;
  }
};

constexpr int kMaxC = 1024;

// LDS parameter block: mean | invstd | gamma | beta  (each C floats)
__device__ __forceinline__ void stage_params(float* sp, int C, const float* mi, const float* gamma, const float* beta) {
  for (int c = threadIdx.x; c < C; c += blockDim.x) { sp[c] = mi[c]; sp[C + c] = mi[C + c]; sp[2 * C + c] = gamma[c]; sp[3 * C + c] = beta[c]; }
  __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel_v(const BnApply d, const ArenaBases ab, const int rev) {
  constexpr int V = VecIO<T>::V;
  __shared__ float sp[4 * kMaxC];
  const int C = d.C;
  stage_params(sp, C, reinterpret_cast<const float*>(rp(ab, d.mean_invstd)), reinterpret_cast<const float*>(rp(ab, d.gamma)),
               reinterpret_cast<const float*>(rp(ab, d.beta)));
  const char* y = rp(ab, d.y);
  char* z = rp(ab, d.z);
  const float a = *reinterpret_cast<const float*>(rp(ab, d.slope));
  const int cmask = (C & (C - 1)) == 0 ? C - 1 : -1;
  const int64_t nq = d.R * C / V;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (cmask >= 0 && ((stride * V) & cmask) == 0) {
    // the grid stride is a whole number of rows: a thread always meets the same V channels, so their parameters live in
    // registers (the LDS look-ups, 4 per element, were costing more issue slots than the 16-byte HBM accesses they serve)
    // rev: walk the tensor from its END - the producer GEMM wrote it front to back, so its tail is what the 256 MB Infinity Cache
    // still holds (the tensors here are 32-126 MB)
    const int64_t idx0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int c = (int)(((rev ? nq - 1 - idx0 : idx0) * V) & cmask);
    float pm[V], pis[V], pg[V], pb[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { pm[e] = sp[c + e]; pis[e] = sp[C + c + e]; pg[e] = sp[2 * C + c + e]; pb[e] = sp[3 * C + c + e]; }
    for (int64_t qq = idx0; qq < nq; qq += stride) {
      const int64_t q = rev ? nq - 1 - qq : qq;
      float v[V];
      VecIO<T>::load(y, q * V, v);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float bn = pg[e] * ((v[e] - pm[e]) * pis[e]) + pb[e];
        v[e] = bn > 0.f ? bn : a * bn;
      }
      VecIO<T>::store(z, q * V, v);
    }
    return;
  }
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nq; q += stride) {
    const int64_t i = q * V;
    const int c = cmask >= 0 ? (int)(i & cmask) : (int)(i % C);
    float v[V];
    VecIO<T>::load(y, i, v);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float bn = sp[2 * C + c + e] * ((v[e] - sp[c + e]) * sp[C + c + e]) + sp[3 * C + c + e];
      v[e] = bn > 0.f ? bn : a * bn;
    }
    VecIO<T>::store(z, i, v);
  }
}

// dz for y-row (b, ql) where ql = row inside the batch item; chunk of V channels at c
template <typename T>
__device__ __forceinline__ void load_dz_v(const BnBwdReduce& d, const char* dz0, const char* dz1, int64_t b, int ql, int c, float* g) {
  constexpr int V = VecIO<T>::V;
#pragma unroll
  for (int e = 0; e < V; ++e) g[e] = 0.f;
  if (ql >= d.skip) VecIO<T>::load(dz0, (b * (d.rpb - d.skip) + ql - d.skip) * d.C + c, g);
  if (dz1) {
    float h[V];
    VecIO<T>::load(dz1, (b * d.rpb + ql) * d.C + c, h);
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] += h[e];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel_v(const BnBwdReduce d, const ArenaBases ab, const int rev) {
  constexpr int V = VecIO<T>::V;
  __shared__ float sp[4 * kMaxC];
  __shared__ float red[512 * V + 8];
  const int C = d.C;
  stage_params(sp, C, reinterpret_cast<const float*>(rp(ab, d.mean_invstd)), reinterpret_cast<const float*>(rp(ab, d.gamma)),
               reinterpret_cast<const float*>(rp(ab, d.beta)));
  const char* y = rp(ab, d.y);
  const char* dz0 = rp(ab, d.dz0);
  const char* dz1 = d.dz1.arena >= 0 ? rp(ab, d.dz1) : nullptr;
  const float a = *reinterpret_cast<const float*>(rp(ab, d.slope));
  const int CV = C / V;                         // chunk columns (<= 256 for C <= 1024 fp32 / 2048 bf16)
  const int nrl = 256 / CV > 0 ? 256 / CV : 1;  // row lanes
  const int rl = threadIdx.x / CV, cc = threadIdx.x - rl * CV;
  float s0[V], s1[V], sa = 0.f;
#pragma unroll
  for (int e = 0; e < V; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
  const int blk = rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;      // rev: last rows first (the dgrad GEMM wrote them last)
  const int64_t row0 = (int64_t)blk * d.rows_per_blk;
  const int64_t row1 = min(d.R, row0 + d.rows_per_blk);
  if (rl < nrl && cc < CV) {
    const int c = cc * V;
    float pm[V], pis[V], pg[V], pb[V];              // this thread's channels never change: parameters in registers
#pragma unroll
    for (int e = 0; e < V; ++e) { pm[e] = sp[c + e]; pis[e] = sp[C + c + e]; pg[e] = sp[2 * C + c + e]; pb[e] = sp[3 * C + c + e]; }
    int64_t r = row0 + rl;
    int64_t b = r / d.rpb;
    int ql = (int)(r - b * d.rpb);
    auto accumulate = [&](const float* yv, const float* gz) {
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float xh = (yv[e] - pm[e]) * pis[e];
        const float bn = pg[e] * xh + pb[e];
        const float dbn = bn > 0.f ? gz[e] : a * gz[e];
        sa += bn > 0.f ? 0.f : bn * gz[e];
        s0[e] += dbn;
        s1[e] += dbn * xh;
      }
    };
    auto next_row = [&]() { r += nrl; ql += nrl; while (ql >= d.rpb) { ql -= (int)d.rpb; ++b; } };
    // two rows per trip: four independent 16-byte loads in flight per lane (one row per trip left the pass latency-bound)
    while (r + nrl < row1) {
      float yv0[V], gz0[V], yv1[V], gz1[V];
      VecIO<T>::load(y, r * C + c, yv0);
      load_dz_v<T>(d, dz0, dz1, b, ql, c, gz0);
      next_row();
      VecIO<T>::load(y, r * C + c, yv1);
      load_dz_v<T>(d, dz0, dz1, b, ql, c, gz1);
      next_row();
      accumulate(yv0, gz0);
      accumulate(yv1, gz1);
    }
    if (r < row1) {
      float yv[V], gz[V];
      VecIO<T>::load(y, r * C + c, yv);
      load_dz_v<T>(d, dz0, dz1, b, ql, c, gz);
      accumulate(yv, gz);
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      red[(rl * C + c + e) * 2 + 0] = s0[e];
      red[(rl * C + c + e) * 2 + 1] = s1[e];
    }
  }
  __syncthreads();
  float* part = reinterpret_cast<float*>(rp(ab, d.part)) + (int64_t)blk * 3 * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    float t0 = 0.f, t1 = 0.f;
    for (int k = 0; k < nrl; ++k) { t0 += red[(k * C + c) * 2 + 0]; t1 += red[(k * C + c) * 2 + 1]; }
    part[c] = t0;
    part[C + c] = t1;
  }
  __syncthreads();
  sa = wave_sum(sa);
  if ((threadIdx.x & 63) == 0) red[512 * V + (threadIdx.x >> 6)] = sa;
  __syncthreads();
  if (threadIdx.x == 0) part[2 * C] = red[512 * V] + red[512 * V + 1] + red[512 * V + 2] + red[512 * V + 3];
  for (int c = 1 + threadIdx.x; c < C; c += 256) part[2 * C + c] = 0.f;      // row 2 holds per-channel shares of the slope gradient: all in channel 0 here
}

// grid: x over the chunks of one batch item, y = batch item
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel_v(const BnBwdApply d, const ArenaBases ab) {
  constexpr int V = VecIO<T>::V;
  __shared__ float sp[4 * kMaxC];
  __shared__ float st[2 * kMaxC];
  const BnBwdReduce& r = d.r;
  const int C = r.C;
  stage_params(sp, C, reinterpret_cast<const float*>(rp(ab, r.mean_invstd)), reinterpret_cast<const float*>(rp(ab, r.gamma)),
               reinterpret_cast<const float*>(rp(ab, r.beta)));
  const float* tot = reinterpret_cast<const float*>(rp(ab, d.totals));
  const float inv_n = (float)(1.0 / d.count);
  for (int c = threadIdx.x; c < C; c += blockDim.x) { st[c] = tot[c] * inv_n; st[C + c] = tot[C + c] * inv_n; }
  __syncthreads();
  const char* y = rp(ab, r.y);
  const char* dz0 = rp(ab, r.dz0);
  const char* dz1 = r.dz1.arena >= 0 ? rp(ab, r.dz1) : nullptr;
  char* dy = rp(ab, d.dy);
  const float a = *reinterpret_cast<const float*>(rp(ab, r.slope));
  const int64_t b = blockIdx.y;
  const int nq = (int)(r.rpb * C / V);
  const int cmask = (C & (C - 1)) == 0 ? C - 1 : -1;
  const int csh = cmask >= 0 ? __ffs(C) - 1 : 0;
  const int qstride = gridDim.x * blockDim.x;
  if (cmask >= 0 && ((qstride * V) & cmask) == 0) {       // same channels on every iteration: parameters in registers
    const int c = ((blockIdx.x * blockDim.x + threadIdx.x) * V) & cmask;
    float pm[V], pis[V], pg[V], pb[V], t0[V], t1[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
      pm[e] = sp[c + e]; pis[e] = sp[C + c + e]; pg[e] = sp[2 * C + c + e]; pb[e] = sp[3 * C + c + e];
      t0[e] = st[c + e]; t1[e] = st[C + c + e];
    }
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += qstride) {
      const int il = q * V;
      const int ql = il >> csh;
      float yv[V], gz[V], o[V];
      const int64_t gi = b * r.rpb * C + il;
      VecIO<T>::load(y, gi, yv);
      load_dz_v<T>(r, dz0, dz1, b, ql, c, gz);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float xh = (yv[e] - pm[e]) * pis[e];
        const float bn = pg[e] * xh + pb[e];
        const float dbn = bn > 0.f ? gz[e] : a * gz[e];
        o[e] = pg[e] * pis[e] * (dbn - t0[e] - xh * t1[e]);
      }
      VecIO<T>::store(dy, gi, o);
    }
    return;
  }
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += qstride) {
    const int il = q * V;
    const int ql = cmask >= 0 ? (il >> csh) : il / C;
    const int c = il - ql * C;
    float yv[V], gz[V], o[V];
    const int64_t gi = b * r.rpb * C + il;
    VecIO<T>::load(y, gi, yv);
    load_dz_v<T>(r, dz0, dz1, b, ql, c, gz);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float xh = (yv[e] - sp[c + e]) * sp[C + c + e];
      const float bn = sp[2 * C + c + e] * xh + sp[3 * C + c + e];
      const float dbn = bn > 0.f ? gz[e] : a * gz[e];
      o[e] = sp[2 * C + c + e] * sp[C + c + e] * (dbn - st[c + e] - xh * st[C + c + e]);
    }
    VecIO<T>::store(dy, gi, o);
  }
}

static inline int gridcap(int64_t n, int cap = 16384) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

void launch_bn(const Op& op, const ArenaBases& ab, hipStream_t st) {
  static const int rev = tune_str("BN_REV") ? atoi(tune_str("BN_REV")) : 1;   // measured: bn_bwd_reduce 97 -> 80 us, bn_bwd_apply 96 -> 86 us per launch (average)
  if (op.kind == OP_BN_APPLY) {
    const BnApply& d = op.bna;
    if (d.dt == DT_BF16) hipLaunchKernelGGL((bn_apply_kernel_v<bf16_t>), dim3(gridcap(d.R * d.C / 8)), dim3(256), 0, st, d, ab, rev);
    else hipLaunchKernelGGL((bn_apply_kernel_v<float>), dim3(gridcap(d.R * d.C / 4)), dim3(256), 0, st, d, ab, rev);
  } else if (op.kind == OP_BN_BWD_REDUCE) {
    const BnBwdReduce& d = op.bnr;
    if (d.dt == DT_BF16) hipLaunchKernelGGL((bn_bwd_reduce_kernel_v<bf16_t>), dim3(d.nblk), dim3(256), 0, st, d, ab, rev);
    else hipLaunchKernelGGL((bn_bwd_reduce_kernel_v<float>), dim3(d.nblk), dim3(256), 0, st, d, ab, rev);
  } else if (op.kind == OP_BN_BWD_APPLY) {
    const BnBwdApply& d = op.bnb;
    const int nb = (int)(d.r.R / d.r.rpb);
    const int v = d.r.dt == DT_BF16 ? 8 : 4;
    int gx = gridcap(d.r.rpb * d.r.C / v, 16384 / (nb > 0 ? nb : 1) + 1);
    if (d.r.dt == DT_BF16) hipLaunchKernelGGL((bn_bwd_apply_kernel_v<bf16_t>), dim3(gx, nb), dim3(256), 0, st, d, ab);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel_v<float>), dim3(gx, nb), dim3(256), 0, st, d, ab);
  }
}

}  // namespace sefd
