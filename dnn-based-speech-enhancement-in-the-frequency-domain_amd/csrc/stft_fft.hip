// Fused STFT kernel: the HBM-bound front end of every model (reference ConvSTFT, tools_for_model.py:54-61, computes the same
// numbers as a 99 MMAC/utterance conv1d; as an FFT it is 2.8 MMAC and the kernel is bound by reading 4 B/sample and writing
// 2 x 257 x 4 B per frame).  One 64-lane wavefront per frame, 8 points per lane:
//   512 = 8 (registers) x 8 x 8 (two LDS transposes inside the wave's private 4 KiB LDS slice; no workgroup barrier).
//   n = l + 64 j :  Y_l[k0] = DFT8_j ;  Z = Y * W512^(l k0)
//   l = l0 + 8 l1:  A[ma] = DFT8_l1(Z) * W64^(l0 ma) ;  X[k0 + 8 ma + 64 mb] = DFT8_l0(A)[mb]
#include <hip/hip_runtime.h>
#include "sefd_desc.h"
#include "tuning.h"
#include <cstdlib>
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

// LDS slice of a wave: 8 rows of 64 complex values, row pitch 72 (a half wave's four rows then start in different bank octets: the 64-pitch
// layout served both transposes 4- to 8-way conflicted, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.5 in profiles/r04_sq_counters_single_lane.json);
// the second transpose pitches its 8-value groups by 9 for the same reason.
constexpr int kFftRow = 72, kFftSlice = 8 * kFftRow;
// 512-point forward DFT of the wave's 512 values: lane holds x[j] = value n = lane + 64 j on entry and
// X[mb] = bin k = (lane >> 3) + 8 (lane & 7) + 64 mb on exit.  buf: this wave's private 4 KiB of LDS; twl: (cos, sin)(2 pi k / 512).
__device__ __forceinline__ void fft512_wave(cf* x, cf* X, float2* buf, const float2* twl, int lane) {
  dft8(x, X);
#pragma unroll
  for (int k0 = 0; k0 < 8; ++k0) {
    const float2 w = twl[(lane * k0) & 511];
    const cf z = cmul(X[k0], cf{w.x, -w.y});                 // W512^(l k0) = cos - i sin
    buf[k0 * kFftRow + lane] = make_float2(z.x, z.y);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // wave-private LDS slice: ordering inside the wave is enough
  const int k0 = lane >> 3, l0 = lane & 7;
#pragma unroll
  for (int l1 = 0; l1 < 8; ++l1) { const float2 v = buf[k0 * kFftRow + l0 + 8 * l1]; x[l1] = {v.x, v.y}; }
  dft8(x, X);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int ma = 0; ma < 8; ++ma) {
    const float2 w = twl[(8 * l0 * ma) & 511];
    const cf z = cmul(X[ma], cf{w.x, -w.y});                 // W64^(l0 ma)
    buf[k0 * kFftRow + ma * 9 + l0] = make_float2(z.x, z.y);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int ma = lane & 7;
#pragma unroll
  for (int q = 0; q < 8; ++q) { const float2 v = buf[k0 * kFftRow + ma * 9 + q]; x[q] = {v.x, v.y}; }
  dft8(x, X);
}

// ONE real frame per complex transform (rounds 2-4): what the fp32 plans keep (StftFft::pair == 0).  The pair transform below is as accurate in the
// max norm (8e-8 of the largest bin against the host simulator), but a frame's small bins then carry the OTHER frame's rounding noise as well, and
// the fp32 goldens of mask mode E - whose backward divides by |mask| - sit within 2.4e-3 instead of 1e-3 on one BatchNorm bias gradient of the
// no-skip model with it (GPU suite, round 5).  fp32 is the parity dtype: it keeps the transform its goldens were captured against.
constexpr int kStftFPW = 2;

__global__ __launch_bounds__(256) void stft_fft_single_kernel(const StftFft d, const ArenaBases ab) {
  constexpr int FPW = kStftFPW;
  __shared__ float2 lds[4][kFftSlice];
  __shared__ float2 twl[512];
  const float* src = reinterpret_cast<const float*>(rp(ab, d.src));
  const float* win = reinterpret_cast<const float*>(rp(ab, d.win));
  const float2* tw = reinterpret_cast<const float2*>(rp(ab, d.tw));
  float2* spec = reinterpret_cast<float2*>(rp(ab, d.spec));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t nfr = (int64_t)d.B * d.T;
  const int64_t fr0 = ((int64_t)blockIdx.x * 4 + wv) * FPW;
  float raw[FPW][8], wn[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) wn[j] = win[lane + 64 * j];
#pragma unroll
  for (int f = 0; f < FPW; ++f) {
    const int64_t fr = fr0 + f;
    const int64_t b = fr < nfr ? fr / d.T : 0;
    const int t = fr < nfr ? (int)(fr - b * d.T) : 0;
    const int p0 = t * d.hop - d.off;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int p = p0 + lane + 64 * j;
      raw[f][j] = (fr < nfr && p >= 0 && p < d.L) ? src[b * d.L + p] : 0.f;
    }
  }
  for (int i = threadIdx.x; i < 512; i += 256) twl[i] = tw[i];
  __syncthreads();
  const int k0 = lane >> 3, ma = lane & 7;
#pragma unroll
  for (int f = 0; f < FPW; ++f) {
    const int64_t fr = fr0 + f;
    if (fr >= nfr) break;                                    // wave-uniform
    cf x[8], X[8];
    float vs = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float v = raw[f][j] * wn[j]; x[j] = {v, 0.f}; vs += v; }
    fft512_wave(x, X, lds[wv], twl, lane);
    // Bins leave in bin order: the transform ends with lane l holding bins (l >> 3) + 8 (l & 7) + 64 mb - stored from there every store
    // instruction touched 64 different 64-byte segments (and 64 different lines of the padded copy).  One more pass through the wave's LDS
    // slice (conflict-free: (l >> 3) + 8 (l & 7) is a permutation of 0..63) and lane l stores bins l + 64 j: whole lines per instruction.
    float2* buf = lds[wv];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) buf[k0 + 8 * ma + 64 * mb] = make_float2(X[mb].x, X[mb].y);
    const cf x256 = X[4];                                    // bin 256: lane 0 (k0 = ma = 0, mb = 4)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float2 Y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) Y[j] = buf[lane + 64 * j];
    float2* out = spec + fr * 258;
    if (d.corr.arena >= 0) {                                 // backward of the pinv synthesis (see sefd_desc.h)
      const float* cr = reinterpret_cast<const float*>(rp(ab, d.corr));
      const float ge = wave_sum((lane & 1) ? 0.f : vs), go = wave_sum((lane & 1) ? vs : 0.f);   // n = lane + 64 j has the parity of lane
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        out[1 + k] = make_float2(d.scale * (Y[j].x - cr[k] * ge - cr[2 * 257 + k] * go),
                                 d.scale * (Y[j].y - cr[257 + k] * ge - cr[3 * 257 + k] * go));
      }
      if (lane == 0) {
        out[1 + 256] = make_float2(d.scale * (x256.x - cr[256] * ge - cr[2 * 257 + 256] * go),
                                   d.scale * (x256.y - cr[257 + 256] * ge - cr[3 * 257 + 256] * go));
        out[0] = make_float2(0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) out[1 + lane + 64 * j] = Y[j];
      if (lane == 0) { out[1 + 256] = make_float2(x256.x, x256.y); out[0] = make_float2(0.f, 0.f); }
      if (d.lp.arena >= 0) {                                 // channel-padded copy for the first encoder layer: one 16 / 32-byte slot per bin
        char* lp = rp(ab, d.lp);
        auto put = [&](int slot, float re, float im) {
          const int64_t s = fr * 258 + slot;
          if (d.lp_dt == DT_BF16) *reinterpret_cast<uint4*>(lp + s * 16) = make_uint4(pack_bf16x2(re, im), 0u, 0u, 0u);
          else {
            *reinterpret_cast<float4*>(lp + s * 32) = make_float4(re, im, 0.f, 0.f);
            *reinterpret_cast<float4*>(lp + s * 32 + 16) = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) put(1 + lane + 64 * j, Y[j].x, Y[j].y);
        if (lane == 0) { put(1 + 256, x256.x, x256.y); put(0, 0.f, 0.f); }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's LDS slice is reused by its next frame
  }
}


// TWO real frames per complex transform (round 5): z[n] = a[n] + i b[n] for a wave's two frames a, b; Z = DFT512(z); the frames' spectra are
// A[k] = (Z[k] + conj Z[N - k]) / 2 and B[k] = (Z[k] - conj Z[N - k]) / (2 i).  The kernel is bound by the ~450 VALU / LDS instructions of one
// transform per frame, not by bandwidth (profiles/r03_tuning_notes.md section 5): one transform per PAIR of frames halves them; the split costs one
// more pass through the wave's LDS slice (bins in order, lane l reads Z[l + 64 j] and Z[512 - l - 64 j]: whole lines, no conflicts).
// PPW pairs per wave: the samples of all of a wave's frames are requested before the twiddle table is staged and before the first transform starts.
template <int PPW>
__global__ __launch_bounds__(256) void stft_fft_kernel(const StftFft d, const ArenaBases ab) {
  __shared__ float2 lds[4][kFftSlice];
  __shared__ float2 twl[512];
  const float* src = reinterpret_cast<const float*>(rp(ab, d.src));
  const float* win = reinterpret_cast<const float*>(rp(ab, d.win));
  const float2* tw = reinterpret_cast<const float2*>(rp(ab, d.tw));
  float2* spec = reinterpret_cast<float2*>(rp(ab, d.spec));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t nfr = (int64_t)d.B * d.T;
  const int64_t fr0 = ((int64_t)blockIdx.x * 4 + wv) * (2 * PPW);
  float raw[2 * PPW][8], wn[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) wn[j] = win[lane + 64 * j];
#pragma unroll
  for (int f = 0; f < 2 * PPW; ++f) {
    const int64_t fr = fr0 + f;
    const int64_t b = fr < nfr ? fr / d.T : 0;
    const int t = fr < nfr ? (int)(fr - b * d.T) : 0;
    const int p0 = t * d.hop - d.off;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int p = p0 + lane + 64 * j;
      raw[f][j] = (fr < nfr && p >= 0 && p < d.L) ? src[b * d.L + p] : 0.f;
    }
  }
  for (int i = threadIdx.x; i < 512; i += 256) twl[i] = tw[i];
  __syncthreads();
  const int k0 = lane >> 3, ma = lane & 7;
  const float* cr = d.corr.arena >= 0 ? reinterpret_cast<const float*>(rp(ab, d.corr)) : nullptr;
#pragma unroll
  for (int pr = 0; pr < PPW; ++pr) {
    const int64_t fra = fr0 + 2 * pr;
    if (fra >= nfr) break;                                   // wave-uniform
    const bool has_b = fra + 1 < nfr;                        // (an odd frame count: the last pair's second frame is all zeros, not stored)
    cf x[8], X[8];
    float vsa = 0.f, vsb = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float va = raw[2 * pr][j] * wn[j], vb = raw[2 * pr + 1][j] * wn[j];
      x[j] = {va, vb}; vsa += va; vsb += vb;
    }
    fft512_wave(x, X, lds[wv], twl, lane);
    // all 512 bins to the wave's LDS slice in bin order (conflict-free: (l >> 3) + 8 (l & 7) is a permutation of 0..63)
    float2* buf = lds[wv];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) buf[k0 + 8 * ma + 64 * mb] = make_float2(X[mb].x, X[mb].y);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float2 Ya[4], Yb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = lane + 64 * j;
      const float2 z = buf[k], n = buf[(512 - k) & 511];
      Ya[j] = make_float2(0.5f * (z.x + n.x), 0.5f * (z.y - n.y));
      Yb[j] = make_float2(0.5f * (z.y + n.y), 0.5f * (n.x - z.x));
    }
    const float2 z256 = buf[256];                            // bin 256 (its own mirror): A = (re, 0), B = (im, 0); stored by lane 0
    float gea = 0.f, goa = 0.f, geb = 0.f, gob = 0.f;
    if (cr) {                                                // backward of the pinv synthesis (see sefd_desc.h): n = lane + 64 j has the parity of lane
      gea = wave_sum((lane & 1) ? 0.f : vsa); goa = wave_sum((lane & 1) ? vsa : 0.f);
      geb = wave_sum((lane & 1) ? 0.f : vsb); gob = wave_sum((lane & 1) ? vsb : 0.f);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h == 1 && !has_b) break;
      const int64_t fr = fra + h;
      const float2* Y = h ? Yb : Ya;
      const float x256 = h ? z256.y : z256.x;
      float2* out = spec + fr * 258;
      if (cr) {
        const float ge = h ? geb : gea, go = h ? gob : goa;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = lane + 64 * j;
          out[1 + k] = make_float2(d.scale * (Y[j].x - cr[k] * ge - cr[2 * 257 + k] * go),
                                   d.scale * (Y[j].y - cr[257 + k] * ge - cr[3 * 257 + k] * go));
        }
        if (lane == 0) {
          out[1 + 256] = make_float2(d.scale * (x256 - cr[256] * ge - cr[2 * 257 + 256] * go),
                                     d.scale * (0.f - cr[257 + 256] * ge - cr[3 * 257 + 256] * go));
          out[0] = make_float2(0.f, 0.f);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[1 + lane + 64 * j] = Y[j];
        if (lane == 0) { out[1 + 256] = make_float2(x256, 0.f); out[0] = make_float2(0.f, 0.f); }
        if (d.lp.arena >= 0) {                               // channel-padded copy for the first encoder layer: one 16 / 32-byte slot per bin
          char* lp = rp(ab, d.lp);
          auto put = [&](int slot, float re, float im) {
            const int64_t s = fr * 258 + slot;
            if (d.lp_dt == DT_BF16) *reinterpret_cast<uint4*>(lp + s * 16) = make_uint4(pack_bf16x2(re, im), 0u, 0u, 0u);
            else {
              *reinterpret_cast<float4*>(lp + s * 32) = make_float4(re, im, 0.f, 0.f);
              *reinterpret_cast<float4*>(lp + s * 32 + 16) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
          };
#pragma unroll
          for (int j = 0; j < 4; ++j) put(1 + lane + 64 * j, Y[j].x, Y[j].y);
          if (lane == 0) { put(1 + 256, x256, 0.f); put(0, 0.f, 0.f); }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's LDS slice is reused by its next pair
  }
}

__global__ __launch_bounds__(256) void istft_fft_kernel(const IstftFft d, const ArenaBases ab) {
  __shared__ float2 lds[4][kFftSlice];
  __shared__ float2 twl[512];
  const float2* est = reinterpret_cast<const float2*>(rp(ab, d.est));
  const float* win = reinterpret_cast<const float*>(rp(ab, d.win));
  const float* cr = reinterpret_cast<const float*>(rp(ab, d.corr));
  const float2* tw = reinterpret_cast<const float2*>(rp(ab, d.tw));
  float* frames = reinterpret_cast<float*>(rp(ab, d.frames));
  for (int i = threadIdx.x; i < 512; i += 256) twl[i] = tw[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t fr = (int64_t)blockIdx.x * 4 + wv;
  if (fr >= d.nframes) return;                               // wave-uniform
  const float2* in = est + fr * 258 + 1;
  cf x[8], X[8];
  float ce = 0.f, co = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = lane + 64 * j;
    float2 v = make_float2(0.f, 0.f);
    if (k <= 256) {
      v = in[k];
      ce += v.x * cr[k] + v.y * cr[257 + k];
      co += v.x * cr[2 * 257 + k] + v.y * cr[3 * 257 + k];
    }
    x[j] = {v.x, -v.y};                                      // inverse transform = conj(DFT(conj Y)); only the real part is needed
  }
  ce = wave_sum(ce);
  co = wave_sum(co);
  fft512_wave(x, X, lds[wv], twl, lane);
  const int k0 = lane >> 3, ma = lane & 7;
  float* out = frames + fr * d.W;
  // samples leave in sample order (see stft_fft_kernel): real parts through the wave's LDS slice, lane l stores samples l + 64 i
  float* buf = reinterpret_cast<float*>(lds[wv]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) buf[k0 + 8 * ma + 64 * mb] = X[mb].x;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = lane + 64 * i;
    if (j < d.W) out[j] = (buf[j] - ((j & 1) ? co : ce)) * win[j] * (1.f / 256.f);
  }
}

void launch_stft_fft(const StftFft& d, const ArenaBases& ab, hipStream_t st) {
  const int64_t frames = (int64_t)d.B * d.T;
  if (!d.pair) { hipLaunchKernelGGL(stft_fft_single_kernel, dim3((unsigned)((frames + 4 * kStftFPW - 1) / (4 * kStftFPW))), dim3(256), 0, st, d, ab); return; }
  static const int ppw = tune_str("STFT_PPW") ? atoi(tune_str("STFT_PPW")) : 1;
  if (ppw == 2) hipLaunchKernelGGL(stft_fft_kernel<2>, dim3((unsigned)((frames + 15) / 16)), dim3(256), 0, st, d, ab);
  else hipLaunchKernelGGL(stft_fft_kernel<1>, dim3((unsigned)((frames + 7) / 8)), dim3(256), 0, st, d, ab);
}
void launch_istft_fft(const IstftFft& d, const ArenaBases& ab, hipStream_t st) {
  hipLaunchKernelGGL(istft_fft_kernel, dim3((unsigned)((d.nframes + 3) / 4)), dim3(256), 0, st, d, ab);
}

}  // namespace sefd
