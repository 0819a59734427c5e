// HBM-bound and recurrent kernels of the speech-enhancement hot path (gfx950).
//   pack / splitsum / unpack   weight re-layout between the reference state_dict layout and the MFMA-friendly packing
//   bn_*                       BatchNorm2d(train) + PReLU forward / backward on channels-last rows
//   lstm_fwd / lstm_bwd        persistent recurrence, W_hh resident in VGPRs as MFMA B-fragments, h exchanged through LDS
//   combine, mask, ola, specout  complex-LSTM glue, cRM application (E/C/R), overlap-add + clamp, layout conversion
#include <hip/hip_runtime.h>
#include <algorithm>
#include <mutex>
#include <unordered_map>
#include "sefd_desc.h"
#include "tuning.h"
#include "dev_common.h"

namespace sefd {

static inline int cdiv64(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int grid_for(int64_t n, int block = 256, int cap = 8192) {
  int64_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ------------------------------------------------------------------------------------------------ pack / unpack
__global__ void pack_kernel(const Pack d, const ArenaBases ab) {
  const int32_t* tab = reinterpret_cast<const int32_t*>(rp(ab, d.tab));
  const float* src = reinterpret_cast<const float*>(rp(ab, d.src));
  char* dst = rp(ab, d.dst);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int e = 0; e < d.width; ++e) {
      const int32_t t = tab[i * d.width + e];
      if (t > 0) v += src[t - 1];
      else if (t < 0) v -= src[-t - 1];
    }
    st_elem(dst, d.ddt, i, v);
  }
}

__global__ void packmulti_kernel(const PackMulti m, const ArenaBases ab) {
  const Pack d = reinterpret_cast<const Pack*>(rp(ab, m.entries))[blockIdx.y];
  const int32_t* tab = reinterpret_cast<const int32_t*>(rp(ab, d.tab));
  const float* src = reinterpret_cast<const float*>(rp(ab, d.src));
  char* dst = rp(ab, d.dst);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int e = 0; e < d.width; ++e) {
      const int32_t t = tab[i * d.width + e];
      if (t > 0) v += src[t - 1];
      else if (t < 0) v -= src[-t - 1];
    }
    st_elem(dst, d.ddt, i, v);
  }
}

// Sum of the nsplit row-split partials of a WGRAD (deterministic order).  A workgroup owns 64 consecutive elements:
// 16 lanes x float4 along the elements, 16 lanes along the splits (thin layers have 768 splits of only 8 K elements, so
// parallelism has to come from the split axis), then a fixed-order 16-way combine through LDS.
__global__ __launch_bounds__(256) void splitsum_kernel(Unpack d, const ArenaBases ab) {
  __shared__ float4 red[16][16];
  float* part = reinterpret_cast<float*>(rp(ab, d.part));
  if (d.nseg > 0) {                                  // table form: blockIdx.y picks the segment (wave-uniform)
    const int64_t* seg = reinterpret_cast<const int64_t*>(rp(ab, d.start)) + 3 * (int64_t)blockIdx.y;
    part += seg[0];
    d.n = d.sstride = seg[1];
    d.nsplit = (int32_t)seg[2];
  }
  const int ex = threadIdx.x & 15, sy = threadIdx.x >> 4;
  const bool vec = (d.n & 3) == 0 && (d.sstride & 3) == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0;
  for (int64_t base = (int64_t)blockIdx.x * 64; base < d.n; base += (int64_t)gridDim.x * 64) {
    const int64_t i = base + ex * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec) {
      if (i < d.n)
        for (int k = sy; k < d.nsplit; k += 16) {
          const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)k * d.sstride + i);
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    } else {
      float* sp = &s.x;
      for (int e = 0; e < 4; ++e)
        if (i + e < d.n)
          for (int k = sy; k < d.nsplit; k += 16) sp[e] += part[(int64_t)k * d.sstride + i + e];
    }
    red[sy][ex] = s;
    __syncthreads();
    if (sy == 0) {
      float4 t = red[0][ex];
#pragma unroll
      for (int j = 1; j < 16; ++j) { const float4 v = red[j][ex]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
      if (vec) { if (i < d.n) *reinterpret_cast<float4*>(part + i) = t; }
      else { const float* tp = &t.x; for (int e = 0; e < 4; ++e) if (i + e < d.n) part[i + e] = tp[e]; }
    }
    __syncthreads();
  }
}

__global__ void unpack_kernel(const Unpack d, const ArenaBases ab) {
  const int32_t* start = reinterpret_cast<const int32_t*>(rp(ab, d.start));
  const int32_t* ent = reinterpret_cast<const int32_t*>(rp(ab, d.ent));
  const float* part = reinterpret_cast<const float*>(rp(ab, d.part));
  float* dst = reinterpret_cast<float*>(rp(ab, d.dst));
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < d.n; j += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int e = start[j]; e < start[j + 1]; ++e) {
      const int32_t t = ent[e];
      if (t > 0) v += part[t - 1];
      else if (t < 0) v -= part[-t - 1];
    }
    if (start[j + 1] > start[j]) dst[j] = v;       // BatchNorm / PReLU gradients are written by their own kernels
  }
}

// ------------------------------------------------------------------------------------------------ BatchNorm + PReLU
__global__ __launch_bounds__(256) void bn_finalize_kernel(const BnFinalize d, const ArenaBases ab) {
  const int c = blockIdx.x;
  if (d.nblk < 0) {                       // eval mode: normalise with the running statistics, no update
    if (threadIdx.x == 0) {
      float* mi = reinterpret_cast<float*>(rp(ab, d.mean_invstd));
      mi[c] = reinterpret_cast<const float*>(rp(ab, d.running_mean))[c];
      mi[d.C + c] = 1.f / sqrtf(reinterpret_cast<const float*>(rp(ab, d.running_var))[c] + d.eps);
    }
    return;
  }
  const float* part = reinterpret_cast<const float*>(rp(ab, d.part));
  __shared__ double r1[256], r2[256];
  if (d.mode == 2) {                      // SyncBN: the totals of all ranks are already in place
    if (threadIdx.x == 0) {
      const double* tot = reinterpret_cast<const double*>(rp(ab, d.totals));
      r1[0] = tot[c];
      r2[0] = tot[d.C + c];
    }
  } else {
    double s1 = 0.0, s2 = 0.0;
    const int nsub = d.nsub > 1 ? d.nsub : 1;
    for (int b = threadIdx.x; b < d.nblk; b += 256)
      for (int u = 0; u < nsub; ++u) {
        s1 += part[((int64_t)b * 2 + 0) * d.Cpad + u * d.substride + c];
        s2 += part[((int64_t)b * 2 + 1) * d.Cpad + u * d.substride + c];
      }
    r1[threadIdx.x] = s1;
    r2[threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
      __syncthreads();
    }
    if (d.mode == 1) {                    // SyncBN: publish this rank's sums, the caller all-reduces them
      if (threadIdx.x == 0) {
        double* tot = reinterpret_cast<double*>(rp(ab, d.totals));
        tot[c] = r1[0];
        tot[d.C + c] = r2[0];
      }
      return;
    }
  }
  if (threadIdx.x == 0) {
    const double mean = r1[0] / d.count;
    double var = r2[0] / d.count - mean * mean;
    if (var < 0) var = 0;
    float* mi = reinterpret_cast<float*>(rp(ab, d.mean_invstd));
    mi[c] = (float)mean;
    mi[d.C + c] = (float)(1.0 / sqrt(var + (double)d.eps));
    if (d.running_mean.arena >= 0) {
      float* rm = reinterpret_cast<float*>(rp(ab, d.running_mean));
      float* rv = reinterpret_cast<float*>(rp(ab, d.running_var));
      const double unb = var * (d.count / (d.count > 1 ? d.count - 1 : 1));
      rm[c] = (float)((1.0 - d.momentum) * rm[c] + d.momentum * mean);
      rv[c] = (float)((1.0 - d.momentum) * rv[c] + d.momentum * unb);
    }
  }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const BnApply d, const ArenaBases ab) {
  const char* y = rp(ab, d.y);
  char* z = rp(ab, d.z);
  const float* mi = reinterpret_cast<const float*>(rp(ab, d.mean_invstd));
  const float* gamma = reinterpret_cast<const float*>(rp(ab, d.gamma));
  const float* beta = reinterpret_cast<const float*>(rp(ab, d.beta));
  const float a = *reinterpret_cast<const float*>(rp(ab, d.slope));
  const int64_t n4 = d.R * d.C / 4;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = q * 4;
    const int c = (int)(i % d.C);
    float4 v = ld4(y, d.dt, i);
    float* pv = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float bn = gamma[c + e] * ((pv[e] - mi[c + e]) * mi[d.C + c + e]) + beta[c + e];
      pv[e] = bn > 0.f ? bn : a * bn;
    }
    st4(z, d.dt, i, v);
  }
}

// upstream gradient wrt z for y-row r, 4 channels starting at c
__device__ __forceinline__ float4 load_dz(const BnBwdReduce& d, const char* dz0, const char* dz1, int64_t r, int c) {
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t b = r / d.rpb, q = r - b * d.rpb;
  if (q >= d.skip) {
    const int64_t dr = b * (d.rpb - d.skip) + q - d.skip;
    g = ld4(dz0, d.dt, dr * d.C + c);
  }
  if (dz1) {
    const float4 h = ld4(dz1, d.dt, r * d.C + c);
    g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w;
  }
  return g;
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BnBwdReduce d, const ArenaBases ab) {
  const char* y = rp(ab, d.y);
  const char* dz0 = rp(ab, d.dz0);
  const char* dz1 = d.dz1.arena >= 0 ? rp(ab, d.dz1) : nullptr;
  const float* mi = reinterpret_cast<const float*>(rp(ab, d.mean_invstd));
  const float* gamma = reinterpret_cast<const float*>(rp(ab, d.gamma));
  const float* beta = reinterpret_cast<const float*>(rp(ab, d.beta));
  const float a = *reinterpret_cast<const float*>(rp(ab, d.slope));
  const int C4 = d.C / 4;
  const int nrl = 256 / C4 > 0 ? 256 / C4 : 1;
  __shared__ float red[2048 + 256];
  float s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, sa = 0.f;
  const int64_t row0 = (int64_t)blockIdx.x * d.rows_per_blk;
  const int64_t row1 = min(d.R, row0 + d.rows_per_blk);
  for (int cc0 = 0; cc0 < C4; cc0 += 256) {              // C4 <= 256 for every supported model; loop keeps it general
    const int rl = C4 >= 256 ? 0 : threadIdx.x / C4;
    const int cc = C4 >= 256 ? cc0 + threadIdx.x : threadIdx.x % C4;
    if (cc < C4 && rl < nrl) {
      const int c = cc * 4;
      for (int64_t r = row0 + rl; r < row1; r += nrl) {
        const float4 yv = ld4(y, d.dt, r * d.C + c);
        const float4 gz = load_dz(d, dz0, dz1, r, c);
        const float* py = reinterpret_cast<const float*>(&yv);
        const float* pg = reinterpret_cast<const float*>(&gz);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (py[e] - mi[c + e]) * mi[d.C + c + e];
          const float bn = gamma[c + e] * xh + beta[c + e];
          const float dbn = bn > 0.f ? pg[e] : a * pg[e];
          sa += bn > 0.f ? 0.f : bn * pg[e];
          s0[e] += dbn;
          s1[e] += dbn * xh;
        }
      }
    }
  }
  // reduce over row-lanes (C4 < 256 case) through LDS: red[rl][C][2]
  float* part = reinterpret_cast<float*>(rp(ab, d.part)) + (int64_t)blockIdx.x * 3 * d.C;
  if (C4 < 256) {
    const int rl = threadIdx.x / C4, cc = threadIdx.x % C4;
    if (rl < nrl) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[(rl * d.C + cc * 4 + e) * 2 + 0] = s0[e];
        red[(rl * d.C + cc * 4 + e) * 2 + 1] = s1[e];
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d.C; c += 256) {
      float t0 = 0.f, t1 = 0.f;
      for (int k = 0; k < nrl; ++k) { t0 += red[(k * d.C + c) * 2 + 0]; t1 += red[(k * d.C + c) * 2 + 1]; }
      part[c] = t0;
      part[d.C + c] = t1;
    }
  } else {
    const int cc = threadIdx.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) { part[cc * 4 + e] = s0[e]; part[d.C + cc * 4 + e] = s1[e]; }
  }
  __syncthreads();
  sa = wave_sum(sa);
  if ((threadIdx.x & 63) == 0) red[2048 + (threadIdx.x >> 6)] = sa;
  __syncthreads();
  if (threadIdx.x == 0) part[2 * d.C] = red[2048] + red[2049] + red[2050] + red[2051];
  for (int c = 1 + threadIdx.x; c < d.C; c += blockDim.x) part[2 * d.C + c] = 0.f;     // row 2 holds per-channel shares (all in channel 0 here)
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const BnBwdApply d, const ArenaBases ab) {
  const int c = blockIdx.x;
  const int C = d.r.C;
  const float* part = reinterpret_cast<const float*>(rp(ab, d.r.part));
  const int ld = d.r.ldp > 0 ? d.r.ldp : C;
  double s0 = 0.0, s1 = 0.0, sa = 0.0;
  for (int b = threadIdx.x; b < d.r.nblk; b += 256) {
    s0 += part[((int64_t)b * 3 + 0) * ld + c];
    s1 += part[((int64_t)b * 3 + 1) * ld + c];
    if (c == 0)                                      // the slope gradient is one scalar: the first workgroup adds every channel's share
      for (int cc = 0; cc < C; ++cc) sa += part[((int64_t)b * 3 + 2) * ld + cc];
  }
  __shared__ double r0[256], r1[256], r2[256];
  r0[threadIdx.x] = s0; r1[threadIdx.x] = s1; r2[threadIdx.x] = sa;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      r0[threadIdx.x] += r0[threadIdx.x + o]; r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float* tot = reinterpret_cast<float*>(rp(ab, d.totals));
    tot[c] = (float)r0[0];
    tot[C + c] = (float)r1[0];
    reinterpret_cast<float*>(rp(ab, d.dbeta))[c] = (float)r0[0];
    reinterpret_cast<float*>(rp(ab, d.dgamma))[c] = (float)r1[0];
    if (c == 0) reinterpret_cast<float*>(rp(ab, d.dslope))[0] = (float)r2[0];
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdApply d, const ArenaBases ab) {
  const BnBwdReduce& r = d.r;
  const char* y = rp(ab, r.y);
  const char* dz0 = rp(ab, r.dz0);
  const char* dz1 = r.dz1.arena >= 0 ? rp(ab, r.dz1) : nullptr;
  char* dy = rp(ab, d.dy);
  const float* mi = reinterpret_cast<const float*>(rp(ab, r.mean_invstd));
  const float* gamma = reinterpret_cast<const float*>(rp(ab, r.gamma));
  const float* beta = reinterpret_cast<const float*>(rp(ab, r.beta));
  const float* tot = reinterpret_cast<const float*>(rp(ab, d.totals));
  const float a = *reinterpret_cast<const float*>(rp(ab, r.slope));
  const float inv_n = (float)(1.0 / d.count);
  const int C = r.C;
  const int64_t n4 = r.R * C / 4;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = q * 4;
    const int64_t row = i / C;
    const int c = (int)(i - row * C);
    const float4 yv = ld4(y, r.dt, i);
    const float4 gz = load_dz(r, dz0, dz1, row, c);
    const float* py = reinterpret_cast<const float*>(&yv);
    const float* pg = reinterpret_cast<const float*>(&gz);
    float4 o;
    float* po = reinterpret_cast<float*>(&o);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (py[e] - mi[c + e]) * mi[C + c + e];
      const float bn = gamma[c + e] * xh + beta[c + e];
      const float dbn = bn > 0.f ? pg[e] : a * pg[e];
      po[e] = gamma[c + e] * mi[C + c + e] * (dbn - tot[c + e] * inv_n - xh * tot[C + c + e] * inv_n);
    }
    st4(dy, r.dt, i, o);
  }
}

// ------------------------------------------------------------------------------------------------ LSTM recurrence
// One workgroup = 16 sequences (rows) of one group for all T steps; wave w owns hidden units [16w, 16w+16) for all four
// gates, so the 16x16 MFMA accumulators of gates i,f,g,o for one (row, unit) live in the same lane and the cell update is
// lane-local.  W_hh stays in VGPRs for the whole sequence as fp32 MFMA B-fragments (4 gates x H/4 k-steps = H registers);
// only h_t crosses waves, through a double-buffered LDS tile.  v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
// D[row=4*(l>>4)+r][col=l&15].
template <int HMAX>
__global__ __launch_bounds__(HMAX * 4) void lstm_fwd_kernel(const LstmRec d, const ArenaBases ab) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int H = d.H, T = d.T;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = blockIdx.y, b0 = blockIdx.x * 16;
  const int set = g % d.nset;
  const float* whh = reinterpret_cast<const float*>(rp(ab, d.whh[set]));
  const float* gx = reinterpret_cast<const float*>(rp(ab, d.gx)) + d.gx_goff[g];
  char* hout = rp(ab, d.h);
  float* gates = reinterpret_cast<float*>(rp(ab, d.gates));
  float* cs = reinterpret_cast<float*>(rp(ab, d.c));
  const int hs = H + 2;                      // LDS row stride (floats): conflict-free A-fragment reads
  const int KS = H / 4;
  const int unit = 16 * w + (lane & 15);
  const int kq = lane >> 4;

  float wreg[4][HMAX / 4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int ks = 0; ks < HMAX / 4; ++ks)
      wreg[q][ks] = ks < KS ? whh[(int64_t)(q * H + unit) * H + 4 * ks + kq] : 0.f;

  for (int i = threadIdx.x; i < 2 * 16 * hs; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();

  bool rvalid[4];
  int64_t rowbt[4];                          // (b*T) for the lane's 4 rows
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + 4 * kq + r;
    rvalid[r] = b < d.B;
    rowbt[r] = (int64_t)(rvalid[r] ? b : 0) * T;
  }
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  float gxv[4][4];
  auto load_gx = [&](int t) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        gxv[q][r] = gx[(rowbt[r] + t) * d.gx_ld + gate_col(q, unit)];     // rows >= B alias row 0, never stored
  };
  load_gx(0);
  const int64_t GBT = (int64_t)d.B * T;
  for (int t = 0; t < T; ++t) {
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][r] = gxv[q][r];
    if (t + 1 < T) load_gx(t + 1);
    const int hp = (t & 1) * 16 * hs;          // LDS offsets, not pointers: keeps the accesses ds_* instead of flat_*
    if (t > 0) {
#pragma unroll
      for (int ks = 0; ks < HMAX / 4; ++ks) {
        if (ks < KS) {
          const float a = lds[hp + (lane & 15) * hs + 4 * ks + kq];
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[q][ks], acc[q], 0, 0, 0);
        }
      }
    }
    const int hn = ((t + 1) & 1) * 16 * hs;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ig = sigmoidf_(acc[0][r]), fg = sigmoidf_(acc[1][r]), gg = tanhf_(acc[2][r]), og = sigmoidf_(acc[3][r]);
      c[r] = fg * c[r] + ig * gg;
      const float h = og * tanhf_(c[r]);
      lds[hn + (4 * kq + r) * hs + unit] = h;
      if (rvalid[r]) {
        const int64_t row = (int64_t)g * GBT + rowbt[r] + t;
        st_elem(hout, d.hdt, row * H + unit, h);
        *reinterpret_cast<float4*>(gates + (row * H + unit) * 4) = make_float4(ig, fg, gg, og);
        cs[row * H + unit] = c[r];
      }
    }
    lds_barrier();
  }
}

// Backward through time.  Per step: lane-local gate gradients -> LDS [16][4H] -> dh_{t-1} = dgates_t . W_hh on the MFMA
// (contraction over all 4H gate columns, W_hh^T slice of this wave's 16 units resident in H VGPRs).
template <int HMAX>
__global__ __launch_bounds__(HMAX * 4) void lstm_bwd_kernel(const LstmRec d, const ArenaBases ab) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int H = d.H, T = d.T;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = blockIdx.y, b0 = blockIdx.x * 16;
  const int set = g % d.nset;
  const float* whh = reinterpret_cast<const float*>(rp(ab, d.whh[set]));
  const float* gates = reinterpret_cast<const float*>(rp(ab, d.gates));
  const float* cs = reinterpret_cast<const float*>(rp(ab, d.c));
  const float* dh = reinterpret_cast<const float*>(rp(ab, d.dh));
  char* dgo = rp(ab, d.dgates);
  const int gs = 4 * H + 2;
  const int unit = 16 * w + (lane & 15);
  const int kq = lane >> 4;

  float wreg[HMAX];                          // B[k = n][j = unit] = W_hh[n][unit], n = 4*ks + kq
#pragma unroll
  for (int ks = 0; ks < HMAX; ++ks) wreg[ks] = ks < H ? whh[(int64_t)(4 * ks + kq) * H + unit] : 0.f;

  bool rvalid[4];
  int64_t rowbt[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + 4 * kq + r;
    rvalid[r] = b < d.B;
    rowbt[r] = (int64_t)(rvalid[r] ? b : 0) * T;
  }
  const int64_t GBT = (int64_t)d.B * T;
  float dcarry[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 dhrec = {0.f, 0.f, 0.f, 0.f};
  // software prefetch: the saved activations of step t-1 are fetched while step t computes (all addresses are known)
  float pg[4][4], pct[4], pcp[4], pdh[4];       // gates i,f,g,o ; c_t ; c_{t-1} ; upstream dh   for the current step
  float ng[4][4], ncp[4], ndh[4];               // the same for the next (t-1) step
  auto fetch = [&](int t, float (&g4)[4][4], float (&cp)[4], float (&dhv)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = (int64_t)g * GBT + rowbt[r] + t;
#pragma unroll
      for (int q = 0; q < 4; ++q) g4[r][q] = gates[(row * H + unit) * 4 + q];      // rows >= B alias row 0, results unused
      cp[r] = cs[(row - (t > 0 ? 1 : 0)) * H + unit];
      dhv[r] = dh[row * H + unit];
    }
  };
  fetch(T - 1, pg, pcp, pdh);
#pragma unroll
  for (int r = 0; r < 4; ++r) pct[r] = cs[((int64_t)g * GBT + rowbt[r] + T - 1) * H + unit];
  for (int t = T - 1; t >= 0; --t) {
    if (t > 0) fetch(t - 1, ng, ncp, ndh);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float di = 0.f, df = 0.f, dg = 0.f, dog = 0.f;
      if (rvalid[r]) {
        const float ig = pg[r][0], fg = pg[r][1], gg = pg[r][2], og = pg[r][3];
        const float ct = pct[r];
        const float cp = t > 0 ? pcp[r] : 0.f;
        const float dht = pdh[r] + dhrec[r];
        const float tc = tanhf_(ct);
        dog = dht * tc * og * (1.f - og);
        const float dc = dht * og * (1.f - tc * tc) + dcarry[r];
        di = dc * gg * ig * (1.f - ig);
        df = dc * cp * fg * (1.f - fg);
        dg = dc * ig * (1.f - gg * gg);
        dcarry[r] = dc * fg;
        const int64_t o = d.gx_goff[g] + (rowbt[r] + t) * d.gx_ld;
        st_elem(dgo, d.gdt, o + gate_col(0, unit), di);
        st_elem(dgo, d.gdt, o + gate_col(1, unit), df);
        st_elem(dgo, d.gdt, o + gate_col(2, unit), dg);
        st_elem(dgo, d.gdt, o + gate_col(3, unit), dog);
      }
      float* lrow = lds + (4 * kq + r) * gs;
      lrow[unit] = di; lrow[H + unit] = df; lrow[2 * H + unit] = dg; lrow[3 * H + unit] = dog;
    }
    lds_barrier();
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    if (t > 0) {
#pragma unroll
      for (int ks = 0; ks < HMAX; ks += 2) {
        if (ks < H) {
          const float x0 = lds[(lane & 15) * gs + 4 * ks + kq];
          const float x1 = lds[(lane & 15) * gs + 4 * (ks + 1) + kq];
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, wreg[ks], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, wreg[ks + 1], a1, 0, 0, 0);
        }
      }
    }
    dhrec = a0 + a1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pct[r] = pcp[r]; pcp[r] = ncp[r]; pdh[r] = ndh[r];
#pragma unroll
      for (int q = 0; q < 4; ++q) pg[r][q] = ng[r][q];
    }
    lds_barrier();
  }
}

// ------------------------------------------------------------------------------------------------ complex-LSTM glue
__global__ void combine_fwd_kernel(const Combine d, const ArenaBases ab) {
  const char* h = rp(ab, d.h);
  char* out = rp(ab, d.out);
  const int64_t gsz = d.rows * d.H;
  const int Tc = d.t1 > 0 ? d.t1 - d.t0 : 0;                 // chunked: rows (b, t0 <= t < t1)
  const int64_t n = Tc > 0 ? (d.rows / d.T) * Tc * d.H : gsz;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    int64_t row = q / d.H;
    const int j = (int)(q - row * d.H);
    if (Tc > 0) { const int64_t b = row / Tc; row = b * d.T + d.t0 + (row - b * Tc); }
    const int64_t i = row * d.H + j;
    const float h0 = ld_elem(h, d.dt, i), h1 = ld_elem(h, d.dt, gsz + i), h2 = ld_elem(h, d.dt, 2 * gsz + i), h3 = ld_elem(h, d.dt, 3 * gsz + i);
    st_elem(out, d.dt, row * 2 * d.H + j, h0 - h3);
    st_elem(out, d.dt, row * 2 * d.H + d.H + j, h2 + h1);
  }
}
__global__ void combine_bwd_kernel(const Combine d, const ArenaBases ab) {   // h := dh (fp32, out), out := dout (fp32, in)
  float* dh = reinterpret_cast<float*>(rp(ab, d.h));
  const float* dout = reinterpret_cast<const float*>(rp(ab, d.out));
  const int64_t n = d.rows * d.H, gsz = d.rows * d.H;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / d.H;
    const int j = (int)(i - row * d.H);
    const float dr = dout[row * 2 * d.H + j], di = dout[row * 2 * d.H + d.H + j];
    dh[i] = dr; dh[gsz + i] = di; dh[2 * gsz + i] = di; dh[3 * gsz + i] = -dr;
  }
}

// ------------------------------------------------------------------------------------------------ mask (E / C / R)
__global__ void mask_fwd_kernel(const Mask d, const ArenaBases ab) {
  const float* spec = reinterpret_cast<const float*>(rp(ab, d.spec));
  const char* mask = rp(ab, d.mask);
  float* est = reinterpret_cast<float*>(rp(ab, d.est));
  const int NS = d.NF + 1;
  const int64_t n = d.frames * NS;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = i / NS;
    const int slot = (int)(i - f * NS);
    float er = 0.f, ei = 0.f, emag = 0.f;
    if (slot >= 2) {                           // bins >= 1 ; bin 0 (slot 1) has a zero mask (models.py:255-256) -> est = 0
      const int64_t b = f / d.T, t = f - b * d.T;
      const int64_t mo = b * d.mask_bstride + t * d.mask_fstride + d.mask_base + (slot - 2) * d.mch;
      const float mr = ld_elem(mask, d.mdt, mo), mi = d.mch >= 2 ? ld_elem(mask, d.mdt, mo + 1) : 0.f;
      const float sr = spec[i * 2], si = spec[i * 2 + 1];
      if (d.mode == 3) {                         // CRN (models.py:519-526): est_mags = tanh(out) * mags, noisy phase re-attached
        const float em = tanhf(mr) * sqrtf(sr * sr + si * si);
        float sn, cs;
        sincosf(atan2f(si, sr), &sn, &cs);
        er = em * cs; ei = em * sn;
        emag = em;
      } else if (d.mode == 5) {                  // CRN spectral mapping: magnitude = decoder output, noisy phase
        float sn, cs;
        sincosf(atan2f(si, sr), &sn, &cs);
        er = mr * cs; ei = mr * sn;
        emag = mr;
      } else if (d.mode == 0) {
        const float mag = sqrtf(sr * sr + si * si + 1e-8f);
        const float ph = atan2f(si, sr);
        const float mm = sqrtf(mr * mr + mi * mi);
        const float mph = atan2f(mi / (mm + 1e-8f), mr / (mm + 1e-8f));
        const float em = tanhf(mm) * mag;
        float sn, cs;
        sincosf(ph + mph, &sn, &cs);
        er = em * cs; ei = em * sn;
      } else if (d.mode == 1) {
        er = sr * mr - si * mi; ei = sr * mi + si * mr;
      } else if (d.mode == 4) {                  // 'Direct(None make)': spectral mapping, the decoder output IS the spectrum (models.py:232-250)
        er = mr; ei = mi;
      } else {
        er = sr * mr; ei = si * mi;
      }
    }
    est[i * 2] = er;
    est[i * 2 + 1] = ei;
    if ((d.mode == 3 || d.mode == 5) && slot >= 1) reinterpret_cast<float*>(rp(ab, d.estm))[f * d.NF + slot - 1] = emag;
  }
}

__global__ void mask_bwd_kernel(const Mask d, const ArenaBases ab) {
  const float* spec = reinterpret_cast<const float*>(rp(ab, d.spec));
  const char* mask = rp(ab, d.mask);
  const float* dest = reinterpret_cast<const float*>(rp(ab, d.dest));
  char* dmask = rp(ab, d.dmask);
  const int NB = d.NF - 1;
  const int NS = d.NF + 1;
  const int lead = (int)(d.mask_base / d.mask_fstride);      // dropped leading decoder frames (1)
  const int TT = d.T + lead;
  const int64_t B = d.frames / d.T;
  const int64_t n = B * TT * NB;
  float cr = 0.f, ci = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % NB);
    const int64_t bu = i / NB;
    const int64_t b = bu / TT;
    const int u = (int)(bu - b * TT);
    const int64_t mo = b * d.mask_bstride + (int64_t)u * d.mask_fstride + k * d.mch;   // includes the leading frame
    float gr = 0.f, gi = 0.f;
    if (u >= lead) {
      const int64_t f = b * d.T + (u - lead);
      const int64_t si_ = (f * NS + k + 2) * 2;
      const float sr = spec[si_], si = spec[si_ + 1];
      const float der = dest[si_], dei = dest[si_ + 1];
      const float mr = ld_elem(mask, d.mdt, mo), mi = d.mch >= 2 ? ld_elem(mask, d.mdt, mo + 1) : 0.f;
      if (d.mode == 3 || d.mode == 5) {
        float sn, cs;
        sincosf(atan2f(si, sr), &sn, &cs);
        float d_em = der * cs + dei * sn;
        if (d.destm.arena >= 0) d_em += reinterpret_cast<const float*>(rp(ab, d.destm))[f * d.NF + k + 1];
        if (d.mode == 3) {
          const float tm = tanhf(mr);
          gr = d_em * sqrtf(sr * sr + si * si) * (1.f - tm * tm);
        } else {
          gr = d_em;
        }
      } else if (d.mode == 0) {
        const float mag = sqrtf(sr * sr + si * si + 1e-8f);
        const float ph = atan2f(si, sr);
        const float mm = sqrtf(mr * mr + mi * mi);
        const float den = mm + 1e-8f;
        const float rpv = mr / den, ipv = mi / den;
        const float mph = atan2f(ipv, rpv);
        const float tm = tanhf(mm);
        const float em = tm * mag;
        float sn, cs;
        sincosf(ph + mph, &sn, &cs);
        const float d_em = der * cs + dei * sn;
        const float d_ph = em * (-der * sn + dei * cs);
        float d_mm = d_em * mag * (1.f - tm * tm);
        const float q = rpv * rpv + ipv * ipv;
        float d_rp = 0.f, d_ip = 0.f;
        if (q > 0.f) { d_ip = d_ph * rpv / q; d_rp = -d_ph * ipv / q; }
        gr = d_rp / den; gi = d_ip / den;
        d_mm += -(d_rp * mr + d_ip * mi) / (den * den);
        if (mm > 0.f) { gr += d_mm * mr / mm; gi += d_mm * mi / mm; }
      } else if (d.mode == 1) {
        gr = der * sr + dei * si; gi = -der * si + dei * sr;
      } else if (d.mode == 4) {
        gr = der; gi = dei;
      } else {
        gr = der * sr; gi = dei * si;
      }
    }
    st_elem(dmask, d.mdt, mo, gr);
    if (d.mch >= 2) st_elem(dmask, d.mdt, mo + 1, gi);
    if (d.mdt == DT_BF16) { gr = bf2f(f2bf(gr)); gi = bf2f(f2bf(gi)); }     // the sums are those of the stored values
    cr += gr;
    if (d.mch >= 2) ci += gi;
  }
  if (d.colsum_rows > 0) {                       // fixed-order reduction: lanes of a wave, then the waves
    __shared__ float red[2][16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cr += __shfl_xor(cr, o); ci += __shfl_xor(ci, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = cr; red[1][threadIdx.x >> 6] = ci; }
    __syncthreads();
    if (threadIdx.x < 8) {
      float v = 0.f;
      if (threadIdx.x < 2)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) v += red[threadIdx.x][w];
      reinterpret_cast<float*>(rp(ab, d.colsum))[(int64_t)blockIdx.x * 8 + threadIdx.x] = v;
    }
  }
}

__global__ void mags_kernel(const Mags d, const ArenaBases ab) {
  const float* spec = reinterpret_cast<const float*>(rp(ab, d.spec));
  char* mags = rp(ab, d.mags);
  const int NS = d.NF + 1;
  const int64_t n = d.frames * d.MS;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = i / d.MS;
    const int k = (int)(i - f * d.MS) - d.MO;
    float v = 0.f;
    if (k >= 0 && k < d.NF) {
      const float sr = spec[(f * NS + k + 1) * 2], si = spec[(f * NS + k + 1) * 2 + 1];
      v = sqrtf(sr * sr + si * si);
    }
    st_elem(mags, d.dt, i, v);
  }
}

// Channel-padded copy of the spectrogram for the first encoder layer: C = 2 would make every run of that layer a 4/8-byte
// fragment; padded to 8 channels the layer uses the same 16-byte LDS-DMA path as all the others (weights of the pad are 0).
__global__ void specpad_kernel(const Mags d, const ArenaBases ab) {
  const float* spec = reinterpret_cast<const float*>(rp(ab, d.spec));
  char* out = rp(ab, d.mags);
  const int64_t n = d.frames * d.NF * d.MS;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % d.MS);
    const int64_t slot = i / d.MS;
    st_elem(out, d.dt, i, ch < 2 ? spec[slot * 2 + ch] : 0.f);
  }
}

// ------------------------------------------------------------------------------------------------ overlap-add
__global__ void ola_fwd_kernel(const Ola d, const ArenaBases ab) {
  const float* fr = reinterpret_cast<const float*>(rp(ab, d.frames));
  const float* coff = reinterpret_cast<const float*>(rp(ab, d.coff));
  float* wav = reinterpret_cast<float*>(rp(ab, d.wav));
  const int64_t n = (int64_t)d.B * d.L;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / d.L;
    const int p = (int)(i - b * d.L) + d.trim;
    int t1 = p / d.hop; if (t1 > d.T - 1) t1 = d.T - 1;
    int t0 = (p - d.win + d.hop) / d.hop; if (p - d.win + 1 <= 0) t0 = 0;
    float s = 0.f;
    for (int t = t0; t <= t1; ++t) s += fr[((int64_t)b * d.T + t) * d.win + (p - t * d.hop)];
    s = s / (coff[p] + 1e-8f);
    wav[i] = d.noclamp ? s : fminf(1.f, fmaxf(-1.f, s));
  }
}
__global__ void ola_bwd_kernel(const Ola d, const ArenaBases ab) {
  const float* coff = reinterpret_cast<const float*>(rp(ab, d.coff));
  const float* wav = reinterpret_cast<const float*>(rp(ab, d.wav));
  const float* dwav = reinterpret_cast<const float*>(rp(ab, d.dwav));
  float* dpad = reinterpret_cast<float*>(rp(ab, d.dpad));
  const int Lp = (d.T - 1) * d.hop + d.win;
  const int64_t n = (int64_t)d.B * Lp;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / Lp;
    const int p = (int)(i - b * Lp);
    float v = 0.f;
    const int s = p - d.trim;
    if (s >= 0 && s < d.L) {
      const float w = wav[b * d.L + s];
      // clamp_ passes the gradient where the un-clamped value lies in [-1, 1]; a clamped sample equals +-1 exactly
      if (w > -1.f && w < 1.f) v = dwav[b * d.L + s] / (coff[p] + 1e-8f);
    }
    dpad[i] = v;
  }
}

// ------------------------------------------------------------------------------------------------ spec layout conversion
// est [B][T][NS][2]  <->  out_real / out_imag [B][NF][T]     (32x32 LDS transpose tiles over (t, bin))
__global__ __launch_bounds__(256) void specout_fwd_kernel(const SpecOut d, const ArenaBases ab) {
  __shared__ float tr[32][33], ti[32][33];
  const float* est = reinterpret_cast<const float*>(rp(ab, d.est));
  float* outr = reinterpret_cast<float*>(rp(ab, d.out_real));
  float* outi = d.mode == 0 ? reinterpret_cast<float*>(rp(ab, d.out_imag)) : nullptr;
  const int NS = d.NF + 1;
  const int b = blockIdx.z, t0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int t = t0 + r, k = k0 + tx;
    float vr = 0.f, vi = 0.f;
    if (t < d.T && k < d.NF) {
      if (d.mode == 2) {
        vr = est[((int64_t)b * d.T + t) * d.NF + k];
      } else {
        const int64_t o = (((int64_t)b * d.T + t) * NS + k + 1) * 2;
        vr = est[o]; vi = est[o + 1];
        if (d.mode == 1) vr = sqrtf(vr * vr + vi * vi);
      }
    }
    tr[r][tx] = vr; ti[r][tx] = vi;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, t = t0 + tx;
    if (t < d.T && k < d.NF) {
      const int64_t o = ((int64_t)b * d.NF + k) * d.T + t;
      if (d.mode == 3) { outr[2 * o] = tr[tx][r]; outr[2 * o + 1] = ti[tx][r]; }
      else {
        outr[o] = tr[tx][r];
        if (d.mode == 0) outi[o] = ti[tx][r];
      }
    }
  }
}
__global__ __launch_bounds__(256) void specout_bwd_kernel(const SpecOut d, const ArenaBases ab) {
  __shared__ float tr[32][33], ti[32][33];
  float* dest = reinterpret_cast<float*>(rp(ab, d.est));
  const float* gr = reinterpret_cast<const float*>(rp(ab, d.out_real));
  const float* gi = (d.mode == 2 || d.mode == 3) ? nullptr : reinterpret_cast<const float*>(rp(ab, d.out_imag));
  const int NS = d.NF + 1;
  const int b = blockIdx.z, t0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, t = t0 + tx;
    float vr = 0.f, vi = 0.f;
    if (t < d.T && k < d.NF) {
      const int64_t o = ((int64_t)b * d.NF + k) * d.T + t;
      if (d.mode == 3) { vr = gr[2 * o]; vi = gr[2 * o + 1]; }
      else { vr = gr[o]; vi = d.mode == 2 ? 0.f : gi[o]; }
    }
    tr[r][tx] = vr; ti[r][tx] = vi;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int t = t0 + r, k = k0 + tx;
    if (t < d.T && k < d.NF) {
      if (d.mode == 2) { dest[((int64_t)b * d.T + t) * d.NF + k] = tr[tx][r]; continue; }     // d est_mags [B][NF][T] -> [B*T][NF]
      const int64_t o = (((int64_t)b * d.T + t) * NS + k + 1) * 2;
      if (d.accumulate) { dest[o] += tr[tx][r]; dest[o + 1] += ti[tx][r]; }
      else { dest[o] = tr[tx][r]; dest[o + 1] = ti[tx][r]; }
    }
  }
}

// ------------------------------------------------------------------------------------------------ dispatch
template <int HMAX>
static void launch_lstm(const Op& op, const ArenaBases& ab, hipStream_t st, bool fwd) {
  const LstmRec& d = op.lstm;
  dim3 grid((d.B + 15) / 16, d.G);
  dim3 block(64 * (d.H / 16));
  if (fwd) {
    const size_t sh = 2 * 16 * (d.H + 2) * sizeof(float);
    hipLaunchKernelGGL((lstm_fwd_kernel<HMAX>), grid, block, sh, st, d, ab);
  } else {
    const size_t sh = 16 * (4 * d.H + 2) * sizeof(float);
    hipLaunchKernelGGL((lstm_bwd_kernel<HMAX>), grid, block, sh, st, d, ab);
  }
}

// ---- two-level BatchNorm finalize (training, per-rank statistics; forward and backward) ------------------------------------------------
// The one-workgroup-per-channel kernels above read one float per 128-byte line of the partial-sum rows (a channel's partials are Cpad
// floats apart): 18-40 us per layer, 22 launches per step.  Here a workgroup owns 32 adjacent channels x one chunk of partial rows
// (coalesced 128-byte reads, 8 row lanes), writes its fp64 chunk sums to a library-owned scratch, and the LAST workgroup of a channel
// group to finish (self-resetting ticket counter) adds the chunk sums in chunk order - the result does not depend on the order in which
// workgroups ran - and does the per-channel finalize.
constexpr int kFinCG = 32, kFinMaxChunks = 64, kFinMaxC = 2048;
struct FinScratch { double* sums; unsigned* tickets; };

static FinScratch fin_scratch(hipStream_t st) {
  // one scratch per stream: finalize launches on one stream are ordered, launches on different streams never share a buffer
  static std::mutex mu;
  static std::unordered_map<StreamKey, FinScratch, StreamKeyHash> map;        // per (device, stream): dev_common.h StreamKey
  std::lock_guard<std::mutex> lk(mu);
  const StreamKey key = stream_key(st);
  auto it = map.find(key);
  if (it != map.end()) return it->second;
  FinScratch f{};
  // + one double per channel group and one ticket for the second level of the slope gradient (bn_bwd_finalize2_kernel)
  const size_t nsum = (size_t)(kFinMaxC / kFinCG) * kFinMaxChunks * 3 * kFinCG + kFinMaxC / kFinCG, ntick = kFinMaxC / kFinCG + 1;
  if (hipMalloc(reinterpret_cast<void**>(&f.sums), nsum * sizeof(double)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&f.tickets), ntick * sizeof(unsigned)) != hipSuccess ||
      hipMemset(f.tickets, 0, ntick * sizeof(unsigned)) != hipSuccess) {
    f.sums = nullptr; f.tickets = nullptr;
    return f;                                      // not cached: the caller falls back to the one-level kernel
  }
  map.emplace(key, f);
  return f;
}

// rows [nblk][NS][ld] floats; sums NS (2 forward, 2 + slope backward) per channel
template <int NS, bool BWD>
__device__ __forceinline__ bool fin_stage1(const float* part, int nblk, int C, int ld, int rowstride, FinScratch fs, double* out /* LDS [3][32] */,
                                           int nsub = 1, int substride = 0) {
  __shared__ double red[8][3][kFinCG];
  __shared__ unsigned ticket;
  const int g = blockIdx.x, k = blockIdx.y, nch = gridDim.y;
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5, c = g * kFinCG + cl;
  const int per = (nblk + nch - 1) / nch, b0 = k * per, b1 = min(nblk, b0 + per);
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  if (c < C) {
    // four partial rows per trip, all their loads in flight before the first add: as one row per trip the loop was a chain of HBM round
    // trips (242 rows per workgroup at the first encoder layer: 26 us for 4 MB)
    int b = b0 + rl;
    if (nsub == 1) {
      for (; b + 24 < b1; b += 32) {
        float v0[4], v1[4], v2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* row = part + (int64_t)(b + 8 * u) * rowstride;
          v0[u] = row[c]; v1[u] = row[ld + c]; v2[u] = BWD ? row[2 * ld + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s0 += v0[u]; s1 += v1[u]; if (BWD) s2 += v2[u]; }
      }
    }
    for (; b < b1; b += 8)
      for (int u = 0; u < nsub; ++u) {
        const float* row = part + (int64_t)b * rowstride + u * substride;
        s0 += row[c];
        s1 += row[ld + c];
        if (BWD) s2 += row[2 * ld + c];
      }
  }
  red[rl][0][cl] = s0; red[rl][1][cl] = s1; red[rl][2][cl] = s2;
  __syncthreads();
  double* mine = fs.sums + ((int64_t)g * kFinMaxChunks + k) * 3 * kFinCG;
  if (threadIdx.x < 3 * kFinCG) {
    const int q = threadIdx.x / kFinCG, cc = threadIdx.x % kFinCG;
    double v = 0.0;
    for (int r = 0; r < 8; ++r) v += red[r][q][cc];
    // write-through store (sc0 sc1) + wait: visible to every XCD before the ticket is taken.  No fence: an agent-scope fence writes back and
    // invalidates this XCD's whole L2 (the GEMM output the next kernel is about to read) - measured +0.5 ms per step
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(mine + q * kFinCG + cc), "v"(v) : "memory");
  }
  __syncthreads();
  if (threadIdx.x == 0) ticket = atomicInc(fs.tickets + g, (unsigned)nch - 1);     // wraps to 0 after the last one: ready for the next launch
  __syncthreads();
  if (ticket != (unsigned)nch - 1) return false;
  if (threadIdx.x < 3 * kFinCG) {
    // cache-bypassing loads (sc0 sc1: the other workgroups' sums come from memory, not from a stale line), all in flight before the first is used
    const int q = threadIdx.x / kFinCG, cc = threadIdx.x % kFinCG;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(fs.sums + (int64_t)g * kFinMaxChunks * 3 * kFinCG, 0,
                                                                       kFinMaxChunks * 3 * kFinCG * 8, 0x00020000);
    double v = 0.0;
    for (int k0 = 0; k0 < nch; k0 += 16) {
      double x[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int kk = k0 + u < nch ? k0 + u : nch - 1;
        x[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (uint32_t)((kk * 3 * kFinCG + q * kFinCG + cc) * 8), 0, 17));
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (k0 + u < nch) v += x[u];
    }
    out[q * kFinCG + cc] = v;
  }
  __syncthreads();
  return true;
}

__global__ __launch_bounds__(256) void bn_finalize2_kernel(const BnFinalize d, const ArenaBases ab, FinScratch fs) {
  __shared__ double tot[3 * kFinCG];
  if (!fin_stage1<2, false>(reinterpret_cast<const float*>(rp(ab, d.part)), d.nblk, d.C, d.Cpad, 2 * d.Cpad, fs, tot, d.nsub > 1 ? d.nsub : 1, d.substride)) return;
  const int c = blockIdx.x * kFinCG + threadIdx.x;
  if (threadIdx.x < kFinCG && c < d.C) {
    const double mean = tot[threadIdx.x] / d.count;
    double var = tot[kFinCG + threadIdx.x] / d.count - mean * mean;
    if (var < 0) var = 0;
    float* mi = reinterpret_cast<float*>(rp(ab, d.mean_invstd));
    mi[c] = (float)mean;
    mi[d.C + c] = (float)(1.0 / sqrt(var + (double)d.eps));
    if (d.running_mean.arena >= 0) {
      float* rm = reinterpret_cast<float*>(rp(ab, d.running_mean));
      float* rv = reinterpret_cast<float*>(rp(ab, d.running_var));
      const double unb = var * (d.count / (d.count > 1 ? d.count - 1 : 1));
      rm[c] = (float)((1.0 - d.momentum) * rm[c] + d.momentum * mean);
      rv[c] = (float)((1.0 - d.momentum) * rv[c] + d.momentum * unb);
    }
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize2_kernel(const BnBwdApply d, const ArenaBases ab, FinScratch fs) {
  __shared__ double tot[3 * kFinCG];
  __shared__ unsigned last;
  const int C = d.r.C, ld = d.r.ldp > 0 ? d.r.ldp : C;
  if (!fin_stage1<3, true>(reinterpret_cast<const float*>(rp(ab, d.r.part)), d.r.nblk, C, ld, 3 * ld, fs, tot)) return;
  const int c = blockIdx.x * kFinCG + threadIdx.x;
  if (threadIdx.x < kFinCG && c < C) {
    float* t = reinterpret_cast<float*>(rp(ab, d.totals));
    t[c] = (float)tot[threadIdx.x];
    t[C + c] = (float)tot[kFinCG + threadIdx.x];
    reinterpret_cast<float*>(rp(ab, d.dbeta))[c] = (float)tot[threadIdx.x];
    reinterpret_cast<float*>(rp(ab, d.dgamma))[c] = (float)tot[kFinCG + threadIdx.x];
  }
  // PReLU slope gradient = sum over ALL channels: this group's share goes to the scratch (write-through), the last group to arrive
  // (second-level ticket, self-resetting) adds the shares in group order
  const int ng = gridDim.x;
  double* gsl = fs.sums + (size_t)(kFinMaxC / kFinCG) * kFinMaxChunks * 3 * kFinCG;
  if (threadIdx.x == 0) {
    double v = 0.0;
    for (int cc = 0; cc < kFinCG; ++cc)
      if (blockIdx.x * kFinCG + cc < C) v += tot[2 * kFinCG + cc];
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(gsl + blockIdx.x), "v"(v) : "memory");
    last = atomicInc(fs.tickets + kFinMaxC / kFinCG, (unsigned)ng - 1) == (unsigned)ng - 1;
    if (last) {
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(gsl, 0, (kFinMaxC / kFinCG) * 8, 0x00020000);
      double s = 0.0;
      for (int g = 0; g < ng; ++g) s += __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (uint32_t)(g * 8), 0, 17));
      reinterpret_cast<float*>(rp(ab, d.dslope))[0] = (float)s;
    }
  }
}

static bool fin_two_level(int nblk, int C, hipStream_t st, FinScratch* fs, dim3* grid) {
  static const bool on = !(tune_str("BN_FIN2") && atoi(tune_str("BN_FIN2")) == 0);
  if (!on || nblk < 64 || C > kFinMaxC) return false;
  *fs = fin_scratch(st);
  if (!fs->sums) return false;
  const int nch = std::min(kFinMaxChunks, (nblk + 31) / 32);
  *grid = dim3((C + kFinCG - 1) / kFinCG, nch);
  return true;
}

void launch_misc(const Op& op, const ArenaBases& ab, hipStream_t st) {
  switch (op.kind) {
    case OP_PACK:
      hipLaunchKernelGGL(pack_kernel, dim3(grid_for(op.pack.n)), dim3(256), 0, st, op.pack, ab); break;
    case OP_PACKMULTI:
      hipLaunchKernelGGL(packmulti_kernel, dim3(128, op.packm.count), dim3(256), 0, st, op.packm, ab); break;
    case OP_SPLITSUM:
      hipLaunchKernelGGL(splitsum_kernel, dim3((unsigned)std::min<int64_t>((op.unpack.n + 63) / 64, op.unpack.nseg > 0 ? 1024 : 8192), std::max(1, op.unpack.nseg)),
                         dim3(256), 0, st, op.unpack, ab);
      break;
    case OP_UNPACK:
      hipLaunchKernelGGL(unpack_kernel, dim3(grid_for(op.unpack.n)), dim3(256), 0, st, op.unpack, ab); break;
    case OP_BN_FINALIZE: {
      FinScratch fs; dim3 grid;
      if (op.bnf.nblk >= 0 && op.bnf.mode == 0 && fin_two_level(op.bnf.nblk, op.bnf.C, st, &fs, &grid))
        hipLaunchKernelGGL(bn_finalize2_kernel, grid, dim3(256), 0, st, op.bnf, ab, fs);
      else
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(op.bnf.C), dim3(256), 0, st, op.bnf, ab);
      break;
    }
    case OP_BN_APPLY:
    case OP_BN_BWD_REDUCE:
    case OP_BN_BWD_APPLY:
      launch_bn(op, ab, st); break;
    case OP_CBN_STATS: case OP_CBN_FINALIZE: case OP_CBN_APPLY: case OP_CBN_BWD_REDUCE: case OP_CBN_BWD_FINALIZE: case OP_CBN_BWD_APPLY:
      launch_cbn(op, ab, st); break;
    case OP_BN_BWD_FINALIZE: {
      FinScratch fs; dim3 grid;
      if (fin_two_level(op.bnb.r.nblk, op.bnb.r.C, st, &fs, &grid))
        hipLaunchKernelGGL(bn_bwd_finalize2_kernel, grid, dim3(256), 0, st, op.bnb, ab, fs);
      else
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(op.bnb.r.C), dim3(256), 0, st, op.bnb, ab);
      break;
    }
    case OP_LSTM_FWD:
    case OP_LSTM_BWD: {
      const bool fwd = op.kind == OP_LSTM_FWD;
      if (op.lstm.hdt == DT_BF16 && op.lstm.impl == 1) launch_lstm_rows(op.lstm, ab, st, fwd);
      else if (op.lstm.hdt == DT_BF16 && op.lstm.H > 128) launch_lstm_cluster(op.lstm, ab, st, fwd);
      else if (op.lstm.hdt == DT_BF16) launch_lstm_bf16(op.lstm, ab, st, fwd);
      else if (op.lstm.H <= 64) launch_lstm<64>(op, ab, st, fwd);
      else launch_lstm<128>(op, ab, st, fwd);
      break;
    }
    case OP_COMBINE_FWD:
      hipLaunchKernelGGL(combine_fwd_kernel, dim3(grid_for(op.comb.rows * op.comb.H)), dim3(256), 0, st, op.comb, ab); break;
    case OP_COMBINE_BWD:
      hipLaunchKernelGGL(combine_bwd_kernel, dim3(grid_for(op.comb.rows * op.comb.H)), dim3(256), 0, st, op.comb, ab); break;
    case OP_MASK_FWD:
      hipLaunchKernelGGL(mask_fwd_kernel, dim3(grid_for(op.mask.frames * (op.mask.NF + 1))), dim3(256), 0, st, op.mask, ab); break;
    case OP_MASK_BWD:
      hipLaunchKernelGGL(mask_bwd_kernel, dim3(op.mask.colsum_rows > 0 ? op.mask.colsum_rows : grid_for(op.mask.frames * 2 * op.mask.NF)), dim3(256), 0, st, op.mask, ab); break;
    case OP_OLA_FWD:
      hipLaunchKernelGGL(ola_fwd_kernel, dim3(grid_for((int64_t)op.ola.B * op.ola.L)), dim3(256), 0, st, op.ola, ab); break;
    case OP_OLA_BWD:
      hipLaunchKernelGGL(ola_bwd_kernel, dim3(grid_for((int64_t)op.ola.B * op.ola.L)), dim3(256), 0, st, op.ola, ab); break;
    case OP_SPECOUT_FWD:
    case OP_SPECOUT_BWD: {
      dim3 grid((op.so.T + 31) / 32, (op.so.NF + 31) / 32, op.so.B);
      if (op.kind == OP_SPECOUT_FWD) hipLaunchKernelGGL(specout_fwd_kernel, grid, dim3(256), 0, st, op.so, ab);
      else hipLaunchKernelGGL(specout_bwd_kernel, grid, dim3(256), 0, st, op.so, ab);
      break;
    }
    case OP_SPECPAD:
      hipLaunchKernelGGL(specpad_kernel, dim3(grid_for(op.mags.frames * op.mags.NF * op.mags.MS)), dim3(256), 0, st, op.mags, ab); break;
    case OP_MAGS:
      hipLaunchKernelGGL(mags_kernel, dim3(grid_for(op.mags.frames * op.mags.MS)), dim3(256), 0, st, op.mags, ab); break;
    case OP_MEMSET:
      (void)hipMemsetAsync(rp(ab, op.ms.dst), 0, op.ms.bytes, st); break;
    default: launch_fsn(op, ab, st); break;
  }
}

}  // namespace sefd
