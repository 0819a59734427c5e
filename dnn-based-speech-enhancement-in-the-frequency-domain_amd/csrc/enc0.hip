// First encoder layer of the bf16 DCCRN plans, read straight from the fp32 spectrum (round 6; VERDICT r5 item 8, north_star: the degenerate
// Cin/2 = 1 layer - models.py:69-75, ComplexConv2d(2 -> kn[0], kernel (5, 2), stride (2, 1)), tools_for_model.py:199-269).
//
// Through round 5 the STFT kernel wrote, beside the [B][T][258][2] fp32 spectrum, a channel-padded bf16 copy [B][T][258][8] (16 bytes per bin for
// 4 bytes of data: 63.8 of the forward launch's 95.7 MB at B = 32) so that the first layer's runs were whole 16-byte chunks for the LDS-DMA GEMMs;
// forward and weight gradient then multiplied K = 128 columns of which 20 are not padding.  The planner now describes the layer on the spectrum
// itself (RunGemm with xdt = fp32, ydt = bf16, runs of 5 bins x 2 = 10 floats, flag kRunEnc0) and these two kernels execute that descriptor:
//   * enc0_fwd_kernel<CO>: a workgroup (4 waves) walks a strip of frames of one utterance; each frame is staged ONCE in LDS (coalesced 16-byte loads,
//     de-interleaved by float index mod 4 so that the per-tap reads of 32 consecutive output bins are conflict-free), reused by the two output frames
//     that read it; a wave multiplies 32 output bins x CO channels x K = 20 with ten exact fp32 MFMAs (32x32x2: A = one input value per lane from
//     LDS, B = the layer's weights, resident in 10 registers per 32 channels); epilogue as the wide GEMMs (bias, BatchNorm partial sums per 128 rows =
//     one frame, quad transpose, 16-byte stores).  HBM: 4 B/sample-bin in, CO x 2 B per output bin out - no intermediate copy;
//   * enc0_wgrad_kernel<CO>: dW[n][k] = sum over rows of dy[row][n] * x[row][k] on the same fp32 MFMA (contraction over row pairs), dy read as bf16
//     rows, x gathered from the spectrum (L2-resident), one workgroup per row split of the planner's flat partition (the partial-sum layout and the
//     SPLITSUM / UNPACK tables of every other weight gradient), its four waves folded through LDS in wave order: deterministic.
// The arithmetic is fp32 x fp32 -> fp32 (closer to the reference than the bf16 operands it replaces); the host simulator interprets the same descriptor.
#include <hip/hip_runtime.h>
#include "sefd_desc.h"
#include "dev_common.h"

namespace sefd {

namespace {

constexpr int kE0Frames = 8;       // output frames per workgroup of the forward kernel
constexpr int kE0Pitch = 132;      // floats per de-interleaved quarter of a staged frame (516 / 4 = 129 used + zero tail)

// the descriptor has the form these kernels execute: one fp32 source, two runs (frame t - 1, frame t) of 10 floats at 4 floats per output bin
__host__ __device__ inline bool enc0_form(const RunGemm& d) {
  return (d.flags & kRunEnc0) && d.xdt == DT_F32 && d.nseg == 2 && d.seg[0].src == 0 && d.seg[1].src == 0 && d.seg[0].len == 10 && d.seg[1].len == 10 &&
         d.seg[0].dt == -1 && d.seg[1].dt == 0 && d.seg[0].off == -4 && d.seg[1].off == -4 && d.fstride[0] == 4 && d.base[0] == 4 && d.rowlen[0] == 512 &&
         d.Fo == 128 && d.tstride[0] == 516 && (d.N == 16 || d.N == 32 || d.N == 64) && d.n2 == 0;
}

}  // namespace

template <int CO>
__global__ __launch_bounds__(256) void enc0_fwd_kernel(const RunGemm d, const ArenaBases ab) {
  constexpr int NB = (CO + 31) / 32;                         // 32-channel blocks
  __shared__ float fr[2][4][kE0Pitch];                       // two staged frames (ring), de-interleaved: fr[slot][i & 3][i >> 2] = frame[i]
  __shared__ float st[2][4][NB * 32][2];                     // per-wave BatchNorm sums of a frame, double-buffered by frame parity
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int T = d.Tout;
  const int strips = (T + kE0Frames - 1) / kE0Frames;
  const int b = blockIdx.x / strips, t0 = (blockIdx.x % strips) * kE0Frames;
  const int t1 = min(T, t0 + kE0Frames);
  const float* x = reinterpret_cast<const float*>(rp(ab, d.x[0])) + (int64_t)b * d.bstride[0];
  const float* w = reinterpret_cast<const float*>(rp(ab, d.w));
  const float* biasp = d.bias.arena >= 0 ? reinterpret_cast<const float*>(rp(ab, d.bias)) : nullptr;
  uint16_t* y = reinterpret_cast<uint16_t*>(rp(ab, d.y));
  float* part = d.stats.arena >= 0 ? reinterpret_cast<float*>(rp(ab, d.stats)) : nullptr;
  // B operand: lane (k = 2 s + (lane >> 5), n = lane & 31) of k-step s; k = kw * 10 + j -> packed column seg[kw].koff + j
  float wb[NB][10];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int s = 0; s < 10; ++s) {
      const int k = 2 * s + (lane >> 5), kw = k / 10, j = k - kw * 10, n = nb * 32 + (lane & 31);
      wb[nb][s] = n < d.N ? w[(int64_t)n * d.ldw + d.seg[kw].koff + j] : 0.f;
    }
  float bv[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) { const int n = nb * 32 + (lane & 31); bv[nb] = (biasp && n < d.N) ? biasp[n] : 0.f; }
  // stage frame tt into ring slot `sl` (zero frame for tt < 0; floats 0..3 = zero pad slot + the DC bin the model drops, floats >= 516 do not exist)
  auto stage = [&](int tt, int sl) {
    if (tid < 33 * 4) {
      const int p = tid;                                     // float4 number p of the frame: floats 4 p .. 4 p + 3 -> position p of the four quarters
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tt >= 0 && p >= 1 && p < 129) v = *reinterpret_cast<const float4*>(x + (int64_t)tt * d.tstride[0] + 4 * p);
      fr[sl][0][p] = v.x; fr[sl][1][p] = v.y; fr[sl][2][p] = v.z; fr[sl][3][p] = v.w;
    }
  };
  stage(t0 - 1, 0);
  const QuadT qt(lane);
  const OctW ow(lane);
  const int fo = wid * 32 + (lane & 31);                     // this lane's output bin as the A operand's row
  for (int t = t0; t < t1; ++t) {
    const int cur = (t - t0 + 1) & 1, prv = cur ^ 1;         // ring slots of frame t and of frame t - 1
    stage(t, cur);
    __syncthreads();
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;
#pragma unroll
    for (int s = 0; s < 10; ++s) {
      const int k = 2 * s + (lane >> 5), kw = k / 10, j = k - kw * 10;        // frame t - 1 + kw, float 4 fo + j of it
      const float a = fr[kw ? cur : prv][j & 3][fo + (j >> 2)];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[nb][s], acc[nb], 0, 0, 0);
    }
    // ---- epilogue: rows (b, t, wid * 32 + r), channel = lane & 31 of block nb
    const int64_t rowbase = (int64_t)b * d.y_bstride + (int64_t)t * d.y_tstride + d.y_off;
    int64_t ro[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ro[q] = rowbase + (int64_t)(wid * 32 + 8 * q + 4 * (lane >> 5) + (lane & 3)) * d.y_fstride;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = nb * 32 + (lane & 31);
      float v[16], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        v[e] = acc[nb][e] + bv[nb];
        if (d.flags & kRunRelu) v[e] = fmaxf(v[e], 0.f);
        s1 += v[e]; s2 += v[e] * v[e];
      }
      if (part) {
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (lane < 32) { st[t & 1][wid][n][0] = s1; st[t & 1][wid][n][1] = s2; }
      }
      uint2 pk[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) pk[q] = qt.pack(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
      uint4 wide[2];
      ow.widen(pk, wide);
      const int n8 = (nb * 32 + (lane & 28)) & ~7;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int64_t r = ow.hi4 ? ro[2 * h + 1] : ro[2 * h];
        if (n8 < d.N) *reinterpret_cast<uint4*>(y + r + n8) = wide[h];
      }
    }
    __syncthreads();                                         // the frame's sums are in LDS; every wave is done with ring slot `prv` (refilled next)
    if (part && tid < d.N) {                                 // one row of partial sums per 128 output rows = this frame; waves added in order
      const float a1 = st[t & 1][0][tid][0] + st[t & 1][1][tid][0] + st[t & 1][2][tid][0] + st[t & 1][3][tid][0];
      const float a2 = st[t & 1][0][tid][1] + st[t & 1][1][tid][1] + st[t & 1][2][tid][1] + st[t & 1][3][tid][1];
      const int64_t blk = (int64_t)b * T + t;
      part[(blk * 2 + 0) * d.Npad + tid] = a1;
      part[(blk * 2 + 1) * d.Npad + tid] = a2;
    }
  }
}

// operands of step s (32 rows) of the weight-gradient kernel, as whole 16-byte chunks: dy rows (32 x CO bf16; FUSE: dz0, dz1 and the forward output y,
// same index) and float4 number fo0 + lane of the frames t - 1, t (lanes 0..34: floats 4 fo0 .. 4 fo0 + 139)
template <int CO, bool FUSE> struct Enc0Regs { uint4 dy[CO / 16]; uint4 dz1[FUSE ? CO / 16 : 1]; uint4 yf[FUSE ? CO / 16 : 1]; float4 x0, x1; };
template <int CO, bool FUSE>
__device__ __forceinline__ Enc0Regs<CO, FUSE> enc0_wg_fetch(const RunGemm& d, const float* x, const uint16_t* dy, const uint16_t* dz1, const uint16_t* yf, int T, int s, int lane) {
  constexpr int CPR = CO / 8;
  Enc0Regs<CO, FUSE> r;
  const int frame = s >> 2, fo0 = (s & 3) * 32;
  const int b = frame / T, t = frame - b * T;
  const int64_t rowbase = (int64_t)b * d.y_bstride + (int64_t)t * d.y_tstride + d.y_off;
#pragma unroll
  for (int i = 0; i < CO / 16; ++i) {
    const int c = lane + 64 * i, row = c / CPR, c8 = c - row * CPR;
    const int64_t o = rowbase + (int64_t)(fo0 + row) * d.y_fstride + c8 * 8;
    r.dy[i] = *reinterpret_cast<const uint4*>(dy + o);
    if constexpr (FUSE) {
      r.dz1[i] = dz1 ? *reinterpret_cast<const uint4*>(dz1 + o) : make_uint4(0, 0, 0, 0);
      r.yf[i] = *reinterpret_cast<const uint4*>(yf + o);
    }
  }
  const int q4 = fo0 + lane;
  r.x0 = make_float4(0.f, 0.f, 0.f, 0.f);
  r.x1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane < 35 && q4 >= 1 && q4 < 129) {
    const float* xb = x + (int64_t)b * d.bstride[0] + 4 * q4;
    if (t >= 1) r.x0 = *reinterpret_cast<const float4*>(xb + (int64_t)(t - 1) * d.tstride[0]);
    r.x1 = *reinterpret_cast<const float4*>(xb + (int64_t)t * d.tstride[0]);
  }
  return r;
}

// FUSE (kRunDyFromBn): the step's dy tile is formed from dz0 (+ dz1) and the layer's forward output through the BatchNorm + PReLU backward and kept in
// fp32; otherwise the stored bf16 dy is the operand
template <int CO, bool FUSE>
__global__ __launch_bounds__(256) void enc0_wgrad_kernel(const RunGemm d, const ArenaBases ab) {
  constexpr int NB = (CO + 31) / 32;
  constexpr int NDY = CO / 16;                               // 16-byte chunks of a 32-row bf16 tile per lane (32 rows x CO x 2 B / 1 KB)
  constexpr int CPR = CO / 8;                                // chunks per row
  // per wave: the step's 32 rows of dy as fp32; after the loop the same bytes carry the accumulators of waves 1..3 to wave 0 (red[3][NB][16][64])
  constexpr int SMF = 128 * CO > 3072 * NB ? 128 * CO : 3072 * NB;
  __shared__ __attribute__((aligned(16))) float smf[SMF];
  float (*dys)[32 * CO] = reinterpret_cast<float (*)[32 * CO]>(smf);
  float (*red)[NB][16][64] = reinterpret_cast<float (*)[NB][16][64]>(smf);
  __shared__ __attribute__((aligned(16))) float xs[4][2][144];           // per wave: floats 4 fo0 .. 4 fo0 + 139 of frames t - 1 and t
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int sp = blockIdx.x;
  const int T = d.Tout;
  const int nsteps = (d.M + kWgRows - 1) / kWgRows;
  const int per = (nsteps + d.nsplit - 1) / d.nsplit;
  const int st0 = sp * per, st1 = min(nsteps, (sp + 1) * per);
  const float* x = reinterpret_cast<const float*>(rp(ab, d.x[0]));
  const uint16_t* dy = reinterpret_cast<const uint16_t*>(rp(ab, d.y));
  const uint16_t* dz1 = (FUSE && d.bnb_dz1.arena >= 0) ? reinterpret_cast<const uint16_t*>(rp(ab, d.bnb_dz1)) : nullptr;
  const uint16_t* yf = FUSE ? reinterpret_cast<const uint16_t*>(rp(ab, d.bnb_y)) : nullptr;
  float* part = reinterpret_cast<float*>(rp(ab, d.w)) + (int64_t)sp * d.Npad * d.ldw;
  // FUSE: the 8 channels of this lane's chunks are the same in every step (64 % CPR == 0): their BatchNorm constants live in registers
  // per channel: bn = ca y + cc (BatchNorm output, only its sign is used), dy = ca dbn - c0 - c1 y, with ca = gamma invstd, cc = beta - ca mean,
  // c1 = ca invstd t1, c0 = ca t0 - c1 mean (t0, t1: the layer's two backward totals / count) - the apply kernel's formula with the constants folded
  float ca[8], cc[8], c0[8], c1[8], slope = 0.f;
  if constexpr (FUSE) {
    const float* mi = reinterpret_cast<const float*>(rp(ab, d.bnb_mi));
    const float* ga = reinterpret_cast<const float*>(rp(ab, d.bnb_gamma));
    const float* be = reinterpret_cast<const float*>(rp(ab, d.bnb_beta));
    const float* tot = reinterpret_cast<const float*>(rp(ab, d.bnb_totals));
    slope = *reinterpret_cast<const float*>(rp(ab, d.bnb_slope));
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int n = (lane % CPR) * 8 + e;
      const float mean = mi[n], is = mi[d.N + n];
      ca[e] = ga[n] * is; cc[e] = be[n] - ca[e] * mean;
      c1[e] = ca[e] * is * (tot[d.N + n] * d.bnb_inv_count); c0[e] = ca[e] * (tot[n] * d.bnb_inv_count) - c1[e] * mean;
    }
  }
  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;
  // B operand column of this lane: k' = lane & 31 -> (kw, j); columns 20..31 multiply zeros
  const int kq = lane & 31, kw = kq >= 10 ? 1 : 0, j = kq - 10 * kw, hh = lane >> 5;
  const bool kok = kq < 20;
  // A step = 32 consecutive rows = a quarter of one frame (Fo = 128).  Its operands travel global -> registers as whole 16-byte chunks (the first
  // version gathered every MFMA operand with its own 2- / 4-byte load: 32 load instructions per 16 MFMAs, 156 us at B = 32) -> the wave's private
  // LDS tile -> MFMA operand reads; the next step's chunks are in flight while this one multiplies.  (The last iteration re-fetches a step it does
  // not use: a conditional copy of the register struct went to scratch.)
  Enc0Regs<CO, FUSE> rg = enc0_wg_fetch<CO, FUSE>(d, x, dy, dz1, yf, T, min(st0 + wid, max(st1, 1) - 1), lane);
  for (int s = st0 + wid; s < st1; s += 4) {
#pragma unroll
    for (int i = 0; i < NDY; ++i) {
      const uint32_t gw[4] = {rg.dy[i].x, rg.dy[i].y, rg.dy[i].z, rg.dy[i].w};
      float o[8];
      if constexpr (FUSE) {
        const uint32_t zw[4] = {rg.dz1[i].x, rg.dz1[i].y, rg.dz1[i].z, rg.dz1[i].w}, yw[4] = {rg.yf[i].x, rg.yf[i].y, rg.yf[i].z, rg.yf[i].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int sh = 16 * (e & 1);
          float dz = bf2f((uint16_t)(gw[e >> 1] >> sh));
          if (dz1) dz += bf2f((uint16_t)(zw[e >> 1] >> sh));
          const float yv = bf2f((uint16_t)(yw[e >> 1] >> sh));
          const float dbn = fmaf(ca[e], yv, cc[e]) > 0.f ? dz : slope * dz;
          o[e] = fmaf(-c1[e], yv, fmaf(ca[e], dbn, -c0[e]));
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = bf2f((uint16_t)(gw[e >> 1] >> (16 * (e & 1))));
      }
      float* dst = &dys[wid][(lane + 64 * i) * 8];
      *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
    if (lane < 35) {
      *reinterpret_cast<float4*>(&xs[wid][0][4 * lane]) = rg.x0;
      *reinterpret_cast<float4*>(&xs[wid][1][4 * lane]) = rg.x1;
    }
    rg = enc0_wg_fetch<CO, FUSE>(d, x, dy, dz1, yf, T, min(s + 4, st1 - 1), lane);
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int r = 2 * p + hh;                              // row of the step this lane feeds into the contraction
      const float bv = kok ? xs[wid][kw][4 * r + j] : 0.f;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = nb * 32 + (lane & 31);
        const float av = n < CO ? dys[wid][r * CO + n] : 0.f;
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[nb], 0, 0, 0);
      }
    }
  }
  __syncthreads();                                           // every wave is done with its dy tile: the bytes change hands
  if (wid > 0) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) red[wid - 1][nb][e][lane] = acc[nb][e];
  }
  __syncthreads();
  // the split's whole [Npad][ldw] block is written: zeros where no run lands
  for (int i = tid; i < d.Npad * d.ldw; i += 256) part[i] = 0.f;
  __syncthreads();
  if (wid == 0) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = ((acc[nb][e] + red[0][nb][e][lane]) + red[1][nb][e][lane]) + red[2][nb][e][lane];
        const int n = nb * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);            // D row = output channel
        if (kok && n < d.N) part[(int64_t)n * d.ldw + d.seg[kw].koff + j] = v;      // D column = lane & 31 = k'
      }
  }
}

bool launch_enc0_fwd(const RunGemm& d, const ArenaBases& ab, hipStream_t st) {
  if (!enc0_form(d) || d.ydt != DT_BF16 || !(d.flags & kRunYAligned) || (d.flags & (kRunAccum | kRunBnBwd))) return false;
  const int B = d.M / (d.Tout * d.Fo);
  const dim3 grid((unsigned)(B * ((d.Tout + kE0Frames - 1) / kE0Frames)));
  if (d.N <= 16) hipLaunchKernelGGL(enc0_fwd_kernel<16>, grid, dim3(256), 0, st, d, ab);
  else if (d.N <= 32) hipLaunchKernelGGL(enc0_fwd_kernel<32>, grid, dim3(256), 0, st, d, ab);
  else hipLaunchKernelGGL(enc0_fwd_kernel<64>, grid, dim3(256), 0, st, d, ab);
  return true;
}

bool launch_enc0_wgrad(const RunGemm& d, const ArenaBases& ab, hipStream_t st) {
  if (!enc0_form(d) || d.ydt != DT_BF16 || d.y_fstride % 8 != 0 || d.y_off % 8 != 0) return false;
  const dim3 grid((unsigned)d.nsplit);
  const bool fuse = (d.flags & kRunDyFromBn) != 0;
#define SEFD_E0W(CO)                                                                                              \
  do {                                                                                                            \
    if (fuse) hipLaunchKernelGGL((enc0_wgrad_kernel<CO, true>), grid, dim3(256), 0, st, d, ab);                   \
    else hipLaunchKernelGGL((enc0_wgrad_kernel<CO, false>), grid, dim3(256), 0, st, d, ab);                       \
  } while (0)
  if (d.N <= 16) SEFD_E0W(16);
  else if (d.N <= 32) SEFD_E0W(32);
  else SEFD_E0W(64);
#undef SEFD_E0W
  return true;
}

}  // namespace sefd
