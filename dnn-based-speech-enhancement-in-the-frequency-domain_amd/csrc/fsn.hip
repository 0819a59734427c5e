// FullSubNet glue kernels (reference models.py:626-672, tools_for_model.py:806-837, 997-1011): everything around the
// per-time-step LSTM GEMMs is HBM-bound element-wise / gather work on time-major tensors.
#include <hip/hip_runtime.h>
#include "sefd_desc.h"
#include "dev_common.h"

namespace sefd {

constexpr float kNormEps = 1.1920928955078125e-07f;    // EPSILON = np.finfo(np.float32).eps (tools_for_model.py: cumulative norms)

static inline int gridn(int64_t n, int cap = 16384) { int64_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > cap ? cap : g)); }
#define GSL(i, n) for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ---------------------------------------------------------------------------------------------- LSTM cell (one time step)
// element offset of (row r, buffer k) and the gate-column helper for both layouts of LstmCell
struct CellAddr {
  int64_t o[5];
};
__device__ __forceinline__ CellAddr cell_addr(const LstmCell& d, int64_t r) {
  CellAddr a;
  if (d.G > 0) {
    const int g = (int)(r / d.Bg);
    const int64_t b = r - (int64_t)g * d.Bg;
#pragma unroll
    for (int k = 0; k < 5; ++k) a.o[k] = d.go[k][g] + b * d.rs[k];
  } else {
    a.o[0] = r * 4 * d.H; a.o[1] = r * d.H; a.o[2] = r * d.H; a.o[3] = r * d.H; a.o[4] = r * 4 * d.H;
  }
  return a;
}
__device__ __forceinline__ int cell_col(const LstmCell& d, int q, int j) { return d.unit_major ? gate_col(q, j) : q * d.H + j; }

__global__ __launch_bounds__(256) void cell_fwd_kernel(const LstmCell d, const ArenaBases ab) {
  float* g = reinterpret_cast<float*>(rp(ab, d.gates));
  const float* cp = d.first ? nullptr : reinterpret_cast<const float*>(rp(ab, d.c_prev));
  float* c = reinterpret_cast<float*>(rp(ab, d.c));
  char* h = rp(ab, d.h);
  const int H = d.H;
  GSL(i, d.rows * H) {
    const int64_t r = i / H;
    const int j = (int)(i - r * H);
    const CellAddr a = cell_addr(d, r);
    float* gr = g + a.o[0];
    const int ci = cell_col(d, 0, j), cf = cell_col(d, 1, j), cg = cell_col(d, 2, j), co = cell_col(d, 3, j);
    const float ig = sigmoidf_(gr[ci]), fg = sigmoidf_(gr[cf]), gg = tanhf_(gr[cg]), og = sigmoidf_(gr[co]);
    const float cn = fg * (cp ? cp[a.o[1] + j] : 0.f) + ig * gg;
    gr[ci] = ig; gr[cf] = fg; gr[cg] = gg; gr[co] = og;
    c[a.o[1] + j] = cn;
    st_elem(h, d.hdt, a.o[2] + j, og * tanhf_(cn));
  }
}

// GRU cell, one time step (LstmCell kind 1)
__global__ __launch_bounds__(256) void gru_fwd_kernel(const LstmCell d, const ArenaBases ab) {
  float* g = reinterpret_cast<float*>(rp(ab, d.gates));
  const float* gh = reinterpret_cast<const float*>(rp(ab, d.gh));
  const char* hp = d.first ? nullptr : rp(ab, d.c_prev);
  char* h = rp(ab, d.h);
  const int H = d.H;
  GSL(i, d.rows * H) {
    const int64_t r = i / H;
    const int j = (int)(i - r * H);
    float* gr = g + r * 4 * H;
    const float* hr = gh + r * 3 * H;
    const float rg = sigmoidf_(gr[j] + hr[j]), zg = sigmoidf_(gr[H + j] + hr[H + j]);
    const float hn = hr[2 * H + j];
    const float ng = tanhf_(gr[2 * H + j] + rg * hn);
    const float hprev = hp ? ld_elem(hp, d.hdt, r * H + j) : 0.f;
    gr[j] = rg; gr[H + j] = zg; gr[2 * H + j] = ng; gr[3 * H + j] = hn;
    st_elem(h, d.hdt, r * H + j, (1.f - zg) * ng + zg * hprev);
  }
}
__global__ __launch_bounds__(256) void gru_bwd_kernel(const LstmCell d, const ArenaBases ab) {
  const float* g = reinterpret_cast<const float*>(rp(ab, d.gates));
  const char* hp = d.c_prev.arena >= 0 ? rp(ab, d.c_prev) : nullptr;
  const float* dh = reinterpret_cast<const float*>(rp(ab, d.dh));
  float* dhp = d.dc.arena >= 0 ? reinterpret_cast<float*>(rp(ab, d.dc)) : nullptr;
  char* dgi = rp(ab, d.dgates);
  char* dgh = rp(ab, d.gh);
  const int H = d.H;
  GSL(i, d.rows * H) {
    const int64_t r = i / H;
    const int j = (int)(i - r * H);
    const float* gr = g + r * 4 * H;
    const float rg = gr[j], zg = gr[H + j], ng = gr[2 * H + j], hn = gr[3 * H + j];
    const float hprev = hp ? ld_elem(hp, d.hdt, r * H + j) : 0.f;
    const float dht = dh[r * H + j];
    const float dn = dht * (1.f - zg) * (1.f - ng * ng);
    const float dz = dht * (hprev - ng) * zg * (1.f - zg);
    const float dr = dn * hn * rg * (1.f - rg);
    st_elem(dgi, d.gdt, r * 3 * H + j, dr); st_elem(dgi, d.gdt, r * 3 * H + H + j, dz); st_elem(dgi, d.gdt, r * 3 * H + 2 * H + j, dn);
    st_elem(dgh, d.gdt, r * 3 * H + j, dr); st_elem(dgh, d.gdt, r * 3 * H + H + j, dz); st_elem(dgh, d.gdt, r * 3 * H + 2 * H + j, dn * rg);
    if (dhp) dhp[r * H + j] += dht * zg;
  }
}

__global__ __launch_bounds__(256) void cell_bwd_kernel(const LstmCell d, const ArenaBases ab) {
  const float* g = reinterpret_cast<const float*>(rp(ab, d.gates));
  const float* cp = d.c_prev.arena >= 0 ? reinterpret_cast<const float*>(rp(ab, d.c_prev)) : nullptr;
  const float* c = reinterpret_cast<const float*>(rp(ab, d.c));
  const float* dh = reinterpret_cast<const float*>(rp(ab, d.dh));
  float* dc = reinterpret_cast<float*>(rp(ab, d.dc));
  char* dg = rp(ab, d.dgates);
  const int H = d.H;
  GSL(i, d.rows * H) {
    const int64_t r = i / H;
    const int j = (int)(i - r * H);
    const CellAddr a = cell_addr(d, r);
    const float* gr = g + a.o[0];
    const int ci = cell_col(d, 0, j), cf = cell_col(d, 1, j), cg = cell_col(d, 2, j), co = cell_col(d, 3, j);
    const float ig = gr[ci], fg = gr[cf], gg = gr[cg], og = gr[co];
    const float tc = tanhf_(c[a.o[1] + j]);
    const float dht = dh[a.o[3] + j];
    const float dcv = dht * og * (1.f - tc * tc) + (d.first ? 0.f : dc[i]);
    st_elem(dg, d.gdt, a.o[4] + ci, dcv * gg * ig * (1.f - ig));
    st_elem(dg, d.gdt, a.o[4] + cf, dcv * (cp ? cp[a.o[1] + j] : 0.f) * fg * (1.f - fg));
    st_elem(dg, d.gdt, a.o[4] + cg, dcv * ig * (1.f - gg * gg));
    st_elem(dg, d.gdt, a.o[4] + co, dht * tc * og * (1.f - og));
    dc[i] = dcv * fg;
  }
}

// ---------------------------------------------------------------------------------------------- dropout
__device__ __forceinline__ float keep_scale(const Dropout& d, const uint32_t* seed, int64_t i) { return drop_scale(seed[0], seed[1], d.layer, d.keep, i); }
__global__ __launch_bounds__(256) void dropout_kernel(const Dropout d, const ArenaBases ab) {   // forward and backward are the same map
  const char* x = rp(ab, d.x);
  char* y = rp(ab, d.y);
  const uint32_t* seed = reinterpret_cast<const uint32_t*>(rp(ab, d.seed));
  GSL(i, d.n) st_elem(y, d.dt, i, ld_elem(x, d.dt, i) * keep_scale(d, seed, i));
}

// ---------------------------------------------------------------------------------------------- input / normalisation
__global__ __launch_bounds__(256) void fsn_in_kernel(const Fsn d, const ArenaBases ab) {       // one workgroup per (b, f-chunk)
  const float* in = reinterpret_cast<const float*>(rp(ab, d.in));       // [B][F][T]
  float* mt = reinterpret_cast<float*>(rp(ab, d.out));                 // [TP][B][F]
  float* sums = reinterpret_cast<float*>(rp(ab, d.sums));              // [B][F]  partial sums (one per (b, f))
  const int b = blockIdx.y;
  for (int f = blockIdx.x; f < d.F; f += gridDim.x) {
    float s = 0.f;
    for (int t = threadIdx.x; t < d.TP; t += 256) {
      const float v = t < d.T ? in[((int64_t)b * d.F + f) * d.T + t] : 0.f;
      mt[((int64_t)t * d.B + b) * d.F + f] = v;
      s += v;
    }
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[b * d.F + f] = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void fsn_scale_kernel(const Fsn d, const ArenaBases ab) {
  const float* mt = reinterpret_cast<const float*>(rp(ab, d.in));
  char* out = rp(ab, d.out);                                             // [TP][B][FP]
  const float* mu = reinterpret_cast<const float*>(rp(ab, d.sums));     // [B] means (finalised)
  GSL(i, (int64_t)d.TP * d.B * d.FP) {
    const int f = (int)(i % d.FP);
    const int64_t tb = i / d.FP;
    const int b = (int)(tb % d.B);
    float v = 0.f;
    if (f < d.F) {
      const float x = mt[tb * d.F + f];
      if (d.mode == 0) v = x / (mu[b] + 1e-5f);
      else {
        const float* stt = reinterpret_cast<const float*>(rp(ab, d.stat));
        if (d.mode == 2) v = (x - stt[b]) / (stt[d.B + b] + 1e-5f);
        else if (d.mode == 1) v = x / (stt[tb * 2] + kNormEps);
        else v = (x - stt[tb * 2]) / stt[tb * 2 + 1];
      }
    }
    st_elem(out, d.dt, i, v);
  }
}
// means: sums [B][nper] -> aux2[B] = sum / count
// one wave per batch item: lane l adds partials l, l + 64, ... in double, then a fixed-order butterfly (one thread per item walking the 257
// partials was a chain of 257 dependent loads: 60 us per launch, three launches per step on the critical path)
__global__ __launch_bounds__(64) void fsn_mean_kernel(const Fsn d, const ArenaBases ab, double count, int nper) {
  const float* sums = reinterpret_cast<const float*>(rp(ab, d.sums));
  float* mu = reinterpret_cast<float*>(rp(ab, d.aux2));
  const int b = blockIdx.x, lane = threadIdx.x;
  double s = 0.0;
  for (int f = lane; f < nper; f += 64) s += sums[b * nper + f];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) mu[b] = (float)(s / count);
}

__device__ __forceinline__ int reflect_idx(int f, int F) { return f < 0 ? -f : (f >= F ? 2 * (F - 1) - f : f); }

// un-normalised sub-band input element (tools_for_model.py:806-837 + models.py:649-655), k < NB: neighbour f - n + k, k == NB: full band
__device__ __forceinline__ float sb_raw(const Fsn& d, const float* mt, const float* fbo, int t, int b, int f, int k) {
  const int n = (d.NB - 1) / 2;
  if (k < d.NB) return mt[((int64_t)t * d.B + b) * d.F + reflect_idx(f - n + k, d.F)];
  return fbo[((int64_t)t * d.B + b) * d.FP + f];
}
__global__ __launch_bounds__(256) void fsn_sbsum_kernel(const Fsn d, const ArenaBases ab) {    // grid (F, B): sums[b][f] over (k, t)
  const float* mt = reinterpret_cast<const float*>(rp(ab, d.in));
  const float* fbo = reinterpret_cast<const float*>(rp(ab, d.aux));
  float* sums = reinterpret_cast<float*>(rp(ab, d.sums));
  const int f = blockIdx.x, b = blockIdx.y, W = d.NB + 1;
  float s = 0.f;
  for (int i = threadIdx.x; i < d.TP * W; i += 256) s += sb_raw(d, mt, fbo, i / W, b, f, i % W);
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) sums[b * d.F + f] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void fsn_sbbuild_kernel(const Fsn d, const ArenaBases ab) {
  const float* mt = reinterpret_cast<const float*>(rp(ab, d.in));
  const float* fbo = reinterpret_cast<const float*>(rp(ab, d.aux));
  const float* mu = reinterpret_cast<const float*>(rp(ab, d.sums));
  char* out = rp(ab, d.out);                                            // [TP][B*F][NB+1]
  const int W = d.NB + 1;
  if (d.mode == 0 && d.dt == DT_BF16 && (W & 7) == 0 && (int64_t)d.TP * d.B * d.F * (W / 8) < (1LL << 31)) {
    // the default norm in bf16 plans (round 6): a thread forms 8 consecutive features of a row and stores them as ONE 16-byte chunk, its index split with
    // 32-bit divisions once per chunk (one thread per element: four 64-bit divisions and a 2-byte store each - 250 us for 203 MB at B = 64; same values)
    const unsigned CH = (unsigned)W / 8, n8 = (unsigned)d.TP * d.B * d.F * CH;
    const int nn = (d.NB - 1) / 2;
    for (unsigned j = blockIdx.x * 256u + threadIdx.x; j < n8; j += gridDim.x * 256u) {
      const unsigned c = j % CH, r = j / CH;
      const unsigned f = r % (unsigned)d.F, tb = r / (unsigned)d.F, b = tb % (unsigned)d.B;
      const float den = mu[b] + 1e-5f;
      const float* mrow = mt + (int64_t)tb * d.F;
      uint32_t pk[4];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        float x[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int k = (int)(8 * c) + e + u;
          x[u] = k < d.NB ? mrow[reflect_idx((int)f - nn + k, d.F)] : fbo[(int64_t)tb * d.FP + f];
        }
        pk[e >> 1] = (uint32_t)f2bf(x[0] / den) | ((uint32_t)f2bf(x[1] / den) << 16);
      }
      *reinterpret_cast<uint4*>(out + ((int64_t)r * W + 8 * c) * 2) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
    return;
  }
  GSL(i, (int64_t)d.TP * d.B * d.F * W) {
    const int k = (int)(i % W);
    const int64_t r = i / W;
    const int f = (int)(r % d.F);
    const int64_t tb = r / d.F;
    const int b = (int)(tb % d.B), t = (int)(tb / d.B);
    const float x = sb_raw(d, mt, fbo, t, b, f, k);
    float v;
    if (d.mode == 0) v = x / (mu[b] + 1e-5f);
    else {
      const float* stt = reinterpret_cast<const float*>(rp(ab, d.stat));
      if (d.mode == 2) v = (x - stt[b]) / (stt[d.B + b] + 1e-5f);
      else if (d.mode == 1) v = x / (stt[r * 2] + kNormEps);
      else v = (x - stt[r * 2]) / stt[r * 2 + 1];
    }
    st_elem(out, d.dt, i, v);
  }
}
__global__ __launch_bounds__(256) void fsn_out_kernel(const Fsn d, const ArenaBases ab) {      // crm [B][F][T][2] <- sbo [TP][B*F][2]
  const float* sbo = reinterpret_cast<const float*>(rp(ab, d.in));
  float* crm = reinterpret_cast<float*>(rp(ab, d.out));
  GSL(i, (int64_t)d.B * d.F * d.T * 2) {
    const int cch = (int)(i & 1);
    const int64_t q = i >> 1;
    const int t = (int)(q % d.T);
    const int64_t bf = q / d.T;
    crm[i] = sbo[(((int64_t)(t + d.LA)) * d.B * d.F + bf) * 2 + cch];
  }
}
__global__ __launch_bounds__(256) void fsn_out_bwd_kernel(const Fsn d, const ArenaBases ab) {  // d_sbo [TP][B*F][2] (dtype dt) <- grad_crm
  const float* g = reinterpret_cast<const float*>(rp(ab, d.in));
  char* ds = rp(ab, d.out);
  GSL(i, (int64_t)d.TP * d.B * d.F * 2) {
    const int cch = (int)(i & 1);
    const int64_t q = i >> 1;
    const int64_t bf = q % ((int64_t)d.B * d.F);
    const int t = (int)(q / ((int64_t)d.B * d.F));
    st_elem(ds, d.dt, i, t >= d.LA ? g[(bf * d.T + (t - d.LA)) * 2 + cch] : 0.f);
  }
}
// S[b][f] = sum_{k,t} d_sbin * sbin   (in = d_sbin fp32 [TP][B*F][W], aux = sbin dtype dt)
__global__ __launch_bounds__(256) void fsn_sbbwd_sum_kernel(const Fsn d, const ArenaBases ab) {
  const float* dsb = reinterpret_cast<const float*>(rp(ab, d.in));
  const char* sb = rp(ab, d.aux);
  float* sums = reinterpret_cast<float*>(rp(ab, d.sums));
  const int f = blockIdx.x, b = blockIdx.y, W = d.NB + 1;
  float s = 0.f;
  for (int i = threadIdx.x; i < d.TP * W; i += 256) {
    const int t = i / W, k = i % W;
    const int64_t o = (((int64_t)t * d.B + b) * d.F + f) * W + k;
    s += dsb[o] * ld_elem(sb, d.dt, o);
  }
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) sums[b * d.F + f] = red[0] + red[1] + red[2] + red[3];
}
// d_fb_pre [TP][B][FP] (dtype dt) = relu'(fbo) * ( d_sbin[..][NB] / (mu+eps) - S_b / (N (mu+eps)) )     aux = fbo, aux2 = mu_sb, sums = S_b (mean form)
__global__ __launch_bounds__(256) void fsn_sbbwd_apply_kernel(const Fsn d, const ArenaBases ab) {
  const float* dsb = reinterpret_cast<const float*>(rp(ab, d.in));
  const float* fbo = reinterpret_cast<const float*>(rp(ab, d.aux));
  const float* mu = reinterpret_cast<const float*>(rp(ab, d.aux2));
  const float* Sm = reinterpret_cast<const float*>(rp(ab, d.sums));     // S_b / N  (finalised as a "mean")
  char* out = rp(ab, d.out);
  const int W = d.NB + 1;
  GSL(i, (int64_t)d.TP * d.B * d.FP) {
    const int f = (int)(i % d.FP);
    const int64_t tb = i / d.FP;
    const int b = (int)(tb % d.B);
    float v = 0.f;
    if (f < d.F) {
      if (d.mode == 0) {
        const float den = mu[b] + 1e-5f;
        v = dsb[(tb * d.F + f) * W + d.NB] / den - Sm[b] / den;
      } else {
        v = dsb[tb * d.F + f];                                   // FSN_NORMBWD already went through the normalisation
      }
      const float y = fbo[i];
      if (d.act == 1) v = y > 0.f ? v : 0.f;                     // ReLU
      else if (d.act == 2) v *= (1.f - y * y);                   // Tanh
      else if (d.act == 3) v = (y > 0.f && y < 6.f) ? v : 0.f;   // ReLU6
    }
    st_elem(out, d.dt, i, v);
  }
}

// ---------------------------------------------------------------------------------------------- other norm_type choices
// block sum of doubles for 64- or 256-thread blocks (result valid in every thread)
__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int nw = blockDim.x >> 6;
  if (nw == 1) return v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0;
  for (int i = 0; i < nw; ++i) t += sh[i];
  return t;
}
// value j of frame t of row (b, f): src 0 -> mag_t[t][b][j] (f unused), src 1 -> the un-normalised sub-band input
__device__ __forceinline__ float norm_val(const Fsn& d, const float* mt, const float* fbo, int t, int b, int f, int j) {
  return d.src ? sb_raw(d, mt, fbo, t, b, f, j) : mt[((int64_t)t * d.B + b) * d.F + j];
}
// mode 2: one workgroup per utterance, (mean, unbiased std) over all its values (torch.mean / torch.std, tools_for_model.py:1047-1061)
__global__ __launch_bounds__(256) void fsn_normstat_utt_kernel(const Fsn d, const ArenaBases ab) {
  const float* mt = reinterpret_cast<const float*>(rp(ab, d.in));
  const float* fbo = d.src ? reinterpret_cast<const float*>(rp(ab, d.aux)) : nullptr;
  float* st = reinterpret_cast<float*>(rp(ab, d.stat));
  __shared__ double sh[4];
  const int b = blockIdx.x, W = d.src ? d.NB + 1 : d.F, nf = d.src ? d.F : 1;
  const int64_t n = (int64_t)d.TP * nf * W;
  double s = 0, q = 0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const int j = (int)(i % W);
    const int64_t r = i / W;
    const float x = norm_val(d, mt, fbo, (int)(r / nf), b, (int)(r % nf), j);
    s += x; q += (double)x * x;
  }
  s = block_sum(s, sh); q = block_sum(q, sh);
  if (threadIdx.x == 0) {
    const double mu = s / n, var = (q - n * mu * mu) / (n - 1);
    st[b] = (float)mu;
    st[d.B + b] = (float)sqrt(var > 0 ? var : 0.0);
  }
}
// modes 1 / 3: one workgroup per row (grid (F or 1, B)), frames in order: running mean and sqrt(running variance + eps) (:1014-1044, 1064-1104)
__global__ void fsn_normstat_cum_kernel(const Fsn d, const ArenaBases ab) {
  const float* mt = reinterpret_cast<const float*>(rp(ab, d.in));
  const float* fbo = d.src ? reinterpret_cast<const float*>(rp(ab, d.aux)) : nullptr;
  float* st = reinterpret_cast<float*>(rp(ab, d.stat));
  __shared__ double sh[4];
  const int f = blockIdx.x, b = blockIdx.y, W = d.src ? d.NB + 1 : d.F;
  const int64_t rows = d.src ? (int64_t)d.B * d.F : d.B, row = d.src ? (int64_t)b * d.F + f : b;
  double cs = 0, cq = 0;
  for (int t = 0; t < d.TP; ++t) {
    double s = 0, q = 0;
    for (int j = threadIdx.x; j < W; j += blockDim.x) { const float x = norm_val(d, mt, fbo, t, b, f, j); s += x; q += (double)x * x; }
    cs += block_sum(s, sh); cq += block_sum(q, sh);
    if (threadIdx.x == 0) {
      const double n = (double)W * (t + 1), m = cs / n;
      st[((int64_t)t * rows + row) * 2] = (float)m;
      st[((int64_t)t * rows + row) * 2 + 1] = d.mode == 3 ? (float)sqrt((cq - 2 * m * cs) / n + m * m + (double)kNormEps) : 0.f;
    }
  }
}
// backward of the sub-band normalisation w.r.t. its full-band column k = NB.  in = d_sbin fp32 [TP][B*F][W], aux2 = sb_in (dtype dt),
// aux = fbo (raw full-band output, fp32 [TP][B][FP]), stat as above, out = fp32 [TP][B][F].
// modes 1 / 3, one wave per row, frames last to first (suffix sums of the statistics' gradients):
//   cumulative_laplace : dx_t = g_t / (m_t + eps) - sum_{u >= t} S_u / ((m_u + eps) n_u)                          S_u = sum_k g y at frame u
//   cumulative_layer   : dx_t = g_t / sd_t - sum_{u >= t} [ G_u / (n_u sd_u) + (x_t - m_u) S_u / (n_u sd_u^2) ]   G_u = sum_k g
__global__ __launch_bounds__(64) void fsn_normbwd_cum_kernel(const Fsn d, const ArenaBases ab) {
  const float* dsb = reinterpret_cast<const float*>(rp(ab, d.in));
  const char* sb = rp(ab, d.aux2);
  const float* fbo = reinterpret_cast<const float*>(rp(ab, d.aux));
  const float* st = reinterpret_cast<const float*>(rp(ab, d.stat));
  float* out = reinterpret_cast<float*>(rp(ab, d.out));
  const int f = blockIdx.x, b = blockIdx.y, W = d.NB + 1, k = threadIdx.x;
  const int64_t rows = (int64_t)d.B * d.F, row = (int64_t)b * d.F + f;
  double accA = 0, accB = 0, accBM = 0;
  for (int t = d.TP - 1; t >= 0; --t) {
    const int64_t o = ((int64_t)t * rows + row) * W;
    const float g = k < W ? dsb[o + k] : 0.f, y = k < W ? ld_elem(sb, d.dt, o + k) : 0.f;
    double G = g, S = (double)g * y;
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) { G += __shfl_xor(G, sft); S += __shfl_xor(S, sft); }
    if (k == 0) {
      const double m = st[((int64_t)t * rows + row) * 2], sd = st[((int64_t)t * rows + row) * 2 + 1], n = (double)W * (t + 1);
      const double gk = dsb[o + d.NB];
      double v;
      if (d.mode == 1) {
        const double den = m + (double)kNormEps;
        accA += S / (den * n);
        v = gk / den - accA;
      } else {
        const double bb = S / (n * sd * sd);
        accA += G / (n * sd); accB += bb; accBM += bb * m;
        v = gk / sd - accA - (double)fbo[((int64_t)t * d.B + b) * d.FP + f] * accB + accBM;
      }
      out[((int64_t)t * d.B + b) * d.F + f] = (float)v;
    }
  }
}
// mode 2 (offline_gaussian_norm): dx = (g - G_b / N) / (sd_b + 1e-5) - y S_b / ((N - 1) sd_b);  sums[2][B][F] partial (S, G) per (b, f)
__global__ __launch_bounds__(256) void fsn_normbwd_utt_part_kernel(const Fsn d, const ArenaBases ab) {
  const float* dsb = reinterpret_cast<const float*>(rp(ab, d.in));
  const char* sb = rp(ab, d.aux2);
  float* part = reinterpret_cast<float*>(rp(ab, d.sums));
  __shared__ double sh[4];
  const int f = blockIdx.x, b = blockIdx.y, W = d.NB + 1;
  double S = 0, G = 0;
  for (int i = threadIdx.x; i < d.TP * W; i += 256) {
    const int64_t o = (((int64_t)(i / W) * d.B + b) * d.F + f) * W + i % W;
    const float g = dsb[o];
    G += g; S += (double)g * ld_elem(sb, d.dt, o);
  }
  S = block_sum(S, sh); G = block_sum(G, sh);
  if (threadIdx.x == 0) { part[b * d.F + f] = (float)S; part[(d.B + b) * d.F + f] = (float)G; }
}
__global__ __launch_bounds__(256) void fsn_normbwd_utt_kernel(const Fsn d, const ArenaBases ab) {
  const float* dsb = reinterpret_cast<const float*>(rp(ab, d.in));
  const char* sb = rp(ab, d.aux2);
  const float* part = reinterpret_cast<const float*>(rp(ab, d.sums));
  const float* st = reinterpret_cast<const float*>(rp(ab, d.stat));
  float* out = reinterpret_cast<float*>(rp(ab, d.out));
  __shared__ double tot[2];
  const int b = blockIdx.y, W = d.NB + 1;
  if (threadIdx.x < 2) {                                        // serial sum of F partials: deterministic, tiny
    double s = 0;
    for (int f = 0; f < d.F; ++f) s += part[(threadIdx.x * d.B + b) * d.F + f];
    tot[threadIdx.x] = s;
  }
  __syncthreads();
  const double N = (double)d.F * W * d.TP, sdv = st[d.B + b], sden = sdv + 1e-5;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < (int64_t)d.TP * d.F; i += (int64_t)gridDim.x * 256) {
    const int f = (int)(i % d.F), t = (int)(i / d.F);
    const int64_t o = (((int64_t)t * d.B + b) * d.F + f) * W + d.NB;
    out[((int64_t)t * d.B + b) * d.F + f] = (float)((dsb[o] - tot[1] / N) / sden - (double)ld_elem(sb, d.dt, o) * tot[0] / ((N - 1) * sdv));
  }
}

__global__ __launch_bounds__(256) void reflectpad_kernel(const ReflectPad d, const ArenaBases ab) {
  const float* src = reinterpret_cast<const float*>(rp(ab, d.src));
  float* dst = reinterpret_cast<float*>(rp(ab, d.dst));
  const int Lp = d.L + 2 * d.pad;
  GSL(i, (int64_t)d.B * Lp) {
    const int64_t b = i / Lp;
    int j = (int)(i - b * Lp) - d.pad;
    j = j < 0 ? -j : (j >= d.L ? 2 * (d.L - 1) - j : j);
    dst[i] = src[b * d.L + j];
  }
}

void launch_fsn(const Op& op, const ArenaBases& ab, hipStream_t st) {
  switch (op.kind) {
    case OP_STFT_FFT: launch_stft_fft(op.fft, ab, st); break;
    case OP_ISTFT_FFT: launch_istft_fft(op.ifft, ab, st); break;
    case OP_REFLECTPAD: hipLaunchKernelGGL(reflectpad_kernel, dim3(gridn((int64_t)op.rpad.B * (op.rpad.L + 2 * op.rpad.pad))), dim3(256), 0, st, op.rpad, ab); break;
    case OP_CELL_FWD:
      if (op.cell.kind == 1) hipLaunchKernelGGL(gru_fwd_kernel, dim3(gridn(op.cell.rows * op.cell.H)), dim3(256), 0, st, op.cell, ab);
      else hipLaunchKernelGGL(cell_fwd_kernel, dim3(gridn(op.cell.rows * op.cell.H)), dim3(256), 0, st, op.cell, ab);
      break;
    case OP_CELL_BWD:
      if (op.cell.kind == 1) hipLaunchKernelGGL(gru_bwd_kernel, dim3(gridn(op.cell.rows * op.cell.H)), dim3(256), 0, st, op.cell, ab);
      else hipLaunchKernelGGL(cell_bwd_kernel, dim3(gridn(op.cell.rows * op.cell.H)), dim3(256), 0, st, op.cell, ab);
      break;
    case OP_DROPOUT_FWD:
    case OP_DROPOUT_BWD: hipLaunchKernelGGL(dropout_kernel, dim3(gridn(op.drop.n)), dim3(256), 0, st, op.drop, ab); break;
    case OP_FSN_IN: {
      const Fsn& d = op.fsn;
      hipLaunchKernelGGL(fsn_in_kernel, dim3(d.F, d.B), dim3(256), 0, st, d, ab);
      hipLaunchKernelGGL(fsn_mean_kernel, dim3(d.B), dim3(64), 0, st, d, ab, (double)d.F * d.TP, d.F);
      break;
    }
    case OP_FSN_SCALE: hipLaunchKernelGGL(fsn_scale_kernel, dim3(gridn((int64_t)op.fsn.TP * op.fsn.B * op.fsn.FP)), dim3(256), 0, st, op.fsn, ab); break;
    case OP_FSN_SBSUM: {
      const Fsn& d = op.fsn;
      hipLaunchKernelGGL(fsn_sbsum_kernel, dim3(d.F, d.B), dim3(256), 0, st, d, ab);
      hipLaunchKernelGGL(fsn_mean_kernel, dim3(d.B), dim3(64), 0, st, d, ab, (double)d.F * d.TP * (d.NB + 1), d.F);
      break;
    }
    case OP_FSN_SBBUILD: hipLaunchKernelGGL(fsn_sbbuild_kernel, dim3(gridn((int64_t)op.fsn.TP * op.fsn.B * op.fsn.F * (op.fsn.NB + 1))), dim3(256), 0, st, op.fsn, ab); break;
    case OP_FSN_OUT: hipLaunchKernelGGL(fsn_out_kernel, dim3(gridn((int64_t)op.fsn.B * op.fsn.F * op.fsn.T * 2)), dim3(256), 0, st, op.fsn, ab); break;
    case OP_FSN_OUT_BWD: hipLaunchKernelGGL(fsn_out_bwd_kernel, dim3(gridn((int64_t)op.fsn.TP * op.fsn.B * op.fsn.F * 2)), dim3(256), 0, st, op.fsn, ab); break;
    case OP_FSN_SBBWD_SUM: {
      const Fsn& d = op.fsn;
      hipLaunchKernelGGL(fsn_sbbwd_sum_kernel, dim3(d.F, d.B), dim3(256), 0, st, d, ab);
      hipLaunchKernelGGL(fsn_mean_kernel, dim3(d.B), dim3(64), 0, st, d, ab, (double)d.F * d.TP * (d.NB + 1), d.F);
      break;
    }
    case OP_FSN_NORMSTAT: {
      const Fsn& d = op.fsn;
      if (d.mode == 2) hipLaunchKernelGGL(fsn_normstat_utt_kernel, dim3(d.B), dim3(256), 0, st, d, ab);
      else if (d.src) hipLaunchKernelGGL(fsn_normstat_cum_kernel, dim3(d.F, d.B), dim3(64), 0, st, d, ab);
      else hipLaunchKernelGGL(fsn_normstat_cum_kernel, dim3(1, d.B), dim3(256), 0, st, d, ab);
      break;
    }
    case OP_FSN_NORMBWD: {
      const Fsn& d = op.fsn;
      if (d.mode == 2) {
        hipLaunchKernelGGL(fsn_normbwd_utt_part_kernel, dim3(d.F, d.B), dim3(256), 0, st, d, ab);
        hipLaunchKernelGGL(fsn_normbwd_utt_kernel, dim3(64, d.B), dim3(256), 0, st, d, ab);
      } else {
        hipLaunchKernelGGL(fsn_normbwd_cum_kernel, dim3(d.F, d.B), dim3(64), 0, st, d, ab);
      }
      break;
    }
    case OP_FSN_SBBWD_APPLY: hipLaunchKernelGGL(fsn_sbbwd_apply_kernel, dim3(gridn((int64_t)op.fsn.TP * op.fsn.B * op.fsn.FP)), dim3(256), 0, st, op.fsn, ab); break;
    default: break;
  }
}

// ---------------------------------------------------------------------------------------------- training targets
// trainer.py:100-104: noisy_mag = |noisy| ; cIRM = compress(complex ratio mask)   (tools_for_model.py:683-717), interleaved complex in.
__global__ __launch_bounds__(256) void fsn_targets_kernel(const float* noisy, const float* clean, int64_t n, float* mag, float* phase, float* cirm) {
  GSL(i, n) {
    const float nr = noisy[2 * i], ni = noisy[2 * i + 1];
    if (mag) mag[i] = sqrtf(nr * nr + ni * ni);
    if (phase) phase[i] = atan2f(ni, nr);
    if (cirm) {
      const float cr = clean[2 * i], ci = clean[2 * i + 1];
      const float den = nr * nr + ni * ni + 1.1920929e-07f;          // np.finfo(np.float32).eps
      float m[2] = {(nr * cr + ni * ci) / den, (nr * ci - ni * cr) / den};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float v = m[e] <= -100.f ? -100.f : m[e];
        const float ex = expf(-0.1f * v);
        cirm[2 * i + e] = 10.f * (1.f - ex) / (1.f + ex);
      }
    }
  }
}

}  // namespace sefd

extern "C" int32_t sefd_fsn_targets(const float* noisy_c64, const float* clean_c64, int64_t n, float* mag, float* phase, float* cirm, void* stream) {
  using namespace sefd;
  hipLaunchKernelGGL(fsn_targets_kernel, dim3(gridn(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), noisy_c64, clean_c64, n, mag, phase, cirm);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
