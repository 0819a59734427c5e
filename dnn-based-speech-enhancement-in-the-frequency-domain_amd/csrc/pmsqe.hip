// PMSQE perceptual loss, forward and gradient w.r.t. the estimated waveform (reference call chain tools_for_loss.py:253-269 and
// models.py:313-314; the arithmetic itself is third-party `asteroid` code that is not under the reference tree: PARITY UNPINNED, this
// follows the published algorithm - Martin-Donas et al., IEEE SPL 2018 - as oracle/pmsqe.py restates it).
//
//   waves [B][L], L = S * 16000: every second is one "source" (view(N, -1, fs));  per second 61 frames of 512 samples, hop 256
//   K1 pmsqe_stft:   frames x sqrt-Hann DFT tables -> spectrum X = sqrt(re^2 + im^2 + 1e-8) (or the power) -> raw Bark spectrum
//                    (P.862.2 frequency warping: band sums x power-density correction) + the speech-band sum of the SLL mean
//   K2 pmsqe_pair:   one workgroup per (estimate second i, clean second j): SLL scale, Bark frequency equalisation, gain equalisation,
//                    Zwicker loudness, symmetric / asymmetric disturbance, per-frame norms, audible-power weight -> pair loss [B][S][S]
//   K3 pmsqe_pit:    per utterance the permutation of the seconds with the smallest mean pair loss (PITLossWrapper 'pw_pt'), batch mean
//   backward: K2 again for the S chosen pairs of each utterance with the hand-derived chain rule down to the raw Bark spectrum and the
//   SLL mean, then K4 pmsqe_istft: Bark / SLL gradients -> spectrum -> frames (transposed DFT tables) -> overlap-add into d(est wave).
// All of it is a few hundred small workgroups (B = 32: 288 pairs x 61 frames x 49 bands); the DFTs run four frames per workgroup so
// that the 1 MB of tables is read from L2 once per four frames.
#include <hip/hip_runtime.h>
#include "../../include/sefd.h"
#include "dev_common.h"

namespace {
using namespace sefd;
constexpr int FS = 16000, NFFT = 512, HOP = 256, NBINS = 257, NB = 49, T = 61, FR = 4, MAXS = 6;
constexpr float SP = 6.910853e-006f;
// float table offsets (built by the host side, sefd_amd/tools_for_loss.py:_pmsqe_tables)
constexpr int O_THR = 0, O_ZP = 49, O_WIDTH = 98, O_CORR = 147, O_ATERM = 196, O_MASK = 245, O_C = 512, O_S = O_C + NFFT * NBINS,
              O_CT = O_S + NFFT * NBINS, O_ST = O_CT + NFFT * NBINS, TAB_FLOATS = O_ST + NFFT * NBINS;
// int table: band_lo[50] (prefix sums of the bins per band), band_of[257] (-1: bin in no band)
constexpr int IO_LO = 0, IO_OF = 64;

struct Ws {
  float *spec, *bark0, *msum, *pw, *gbark, *gm, *perloss;
  int32_t* sel;
};
__host__ __device__ inline int64_t ws_floats(int B, int S) {
  const int64_t BS = (int64_t)B * S;
  return BS * T * NBINS * 2 + 2 * BS * T * NB + 2 * BS * T + BS * S + BS * T * NB + BS + B + BS + 64;
}
__host__ __device__ inline Ws carve(float* w, int B, int S) {
  const int64_t BS = (int64_t)B * S;
  Ws r;
  r.spec = w; w += BS * T * NBINS * 2;
  r.bark0 = w; w += 2 * BS * T * NB;
  r.msum = w; w += 2 * BS * T;
  r.pw = w; w += BS * S;
  r.gbark = w; w += BS * T * NB;
  r.gm = w; w += BS;
  r.perloss = w; w += B;
  r.sel = reinterpret_cast<int32_t*>(w);
  return r;
}

__global__ __launch_bounds__(320) void pmsqe_stft_kernel(const float* est, const float* clean, int BS, int power, const float* tab,
                                                         const int32_t* itab, Ws ws) {
  __shared__ float x[FR][NFFT], X[FR][NBINS + 3], red[FR][5];
  const int seg = blockIdx.y, t0 = blockIdx.x * FR, tid = threadIdx.x;
  const bool is_est = seg < BS;
  const float* src = (is_est ? est : clean) + (int64_t)(is_est ? seg : seg - BS) * FS;
  for (int i = tid; i < FR * NFFT; i += 320) {
    const int fr = i / NFFT, n = i % NFFT, t = t0 + fr;
    x[fr][n] = t < T ? src[t * HOP + n] : 0.f;
  }
  __syncthreads();
  float part[FR] = {};
  if (tid < NBINS) {
    float re[FR] = {}, im[FR] = {};
    const float *C = tab + O_C + tid, *Sm = tab + O_S + tid;
    for (int n = 0; n < NFFT; ++n) {
      const float c = C[n * NBINS], s = Sm[n * NBINS];
#pragma unroll
      for (int fr = 0; fr < FR; ++fr) { re[fr] += x[fr][n] * c; im[fr] += x[fr][n] * s; }
    }
    const float m = tab[O_MASK + tid];
#pragma unroll
    for (int fr = 0; fr < FR; ++fr) {
      const float p = re[fr] * re[fr] + im[fr] * im[fr];
      const float v = power ? p : sqrtf(p + 1e-8f);
      X[fr][tid] = v;
      part[fr] = v * m;
      if (is_est && t0 + fr < T) {
        float* o = ws.spec + (((int64_t)seg * T + t0 + fr) * NBINS + tid) * 2;
        o[0] = re[fr]; o[1] = im[fr];
      }
    }
  }
#pragma unroll
  for (int fr = 0; fr < FR; ++fr) {
    const float s = wave_sum(part[fr]);
    if ((tid & 63) == 0) red[fr][tid >> 6] = s;
  }
  __syncthreads();
  if (tid < FR && t0 + tid < T) ws.msum[(int64_t)seg * T + t0 + tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3] + red[tid][4];
  if (tid < FR * NB) {
    const int fr = tid / NB, k = tid % NB;
    if (t0 + fr < T) {
      float s = 0.f;
      for (int f = itab[IO_LO + k]; f < itab[IO_LO + k + 1]; ++f) s += X[fr][f];
      ws.bark0[((int64_t)seg * T + t0 + fr) * NB + k] = s * tab[O_CORR + k];
    }
  }
}

__device__ __forceinline__ float loud(float b, float thr, float zp, float aterm) {   // aterm = Sl * (thr / 0.5)^zp
  return b < thr ? 0.f : aterm * (powf(0.5f + 0.5f * b / thr, zp) - 1.f);
}

// BWD = false: grid B*S*S, pair (e = b*S + i, c = b*S + j) -> pw.   BWD = true: grid B*S, pair (e, sel[e]) -> gbark[e], gm[e].
template <bool BWD>
__global__ __launch_bounds__(256) void pmsqe_pair_kernel(int B, int S, const float* tab, Ws ws) {
  __shared__ float BR[T * NB], BD[T * NB], G[BWD ? T * NB : 1];
  __shared__ float thr[NB], zp[NB], wid[NB], at[NB], eq[NB], avr[NB], avd[NB], apr[T], apd[T], gain[T], wgt[T], lossf[T], red[4];
  __shared__ unsigned char ns[T];
  const int tid = threadIdx.x, BS = B * S;
  int e, c;
  if (BWD) { e = blockIdx.x; c = (e / S) * S + ws.sel[e]; }
  else { const int b = blockIdx.x / (S * S), r = blockIdx.x % (S * S); e = b * S + r / S; c = b * S + r % S; }
  float md = 0.f, mr = 0.f;
  for (int t = 0; t < T; ++t) { md += ws.msum[(int64_t)e * T + t]; mr += ws.msum[(int64_t)(BS + c) * T + t]; }
  md /= (float)(NBINS * T); mr /= (float)(NBINS * T);
  const float sd = 1e7f * SP / md, sr = 1e7f * SP / mr;
  if (tid < NB) { thr[tid] = tab[O_THR + tid]; zp[tid] = tab[O_ZP + tid]; wid[tid] = tab[O_WIDTH + tid]; at[tid] = tab[O_ATERM + tid]; }
  for (int i = tid; i < T * NB; i += 256) {
    BD[i] = sd * ws.bark0[(int64_t)e * T * NB + i];
    BR[i] = sr * ws.bark0[(int64_t)(BS + c) * T * NB + i];
  }
  __syncthreads();
  if (tid < T) {                                   // audible power of the reference, x1 and x100 thresholds
    float a1 = 0.f, a100 = 0.f;
    for (int k = 0; k < NB; ++k) { const float v = BR[tid * NB + k]; if (v > thr[k]) a1 += v; if (v > 100.f * thr[k]) a100 += v; }
    apr[tid] = a1;
    ns[tid] = a100 >= 1e7f;
  }
  __syncthreads();
  if (tid < NB) {                                  // Bark frequency equaliser over the speech-active frames
    float ar = 0.f, ad = 0.f;
    for (int t = 0; t < T; ++t)
      if (ns[t] && BR[t * NB + tid] >= 100.f * thr[tid]) { ar += BR[t * NB + tid]; ad += BD[t * NB + tid]; }
    avr[tid] = ar; avd[tid] = ad;
    eq[tid] = fminf(fmaxf((ar + 1000.f) / (ad + 1000.f), 0.01f), 100.f);
  }
  __syncthreads();
  for (int i = tid; i < T * NB; i += 256) BD[i] *= eq[i % NB];              // BD1
  __syncthreads();
  if (tid < T) {                                   // gain equaliser, then the frame's disturbances (BD stays BD1: BD2 = gain * BD1)
    float a = 0.f;
    for (int k = 0; k < NB; ++k) { const float v = BD[tid * NB + k]; if (v > thr[k]) a += v; }
    apd[tid] = a;
    const float g = fminf(fmaxf((apr[tid] + 5e3f) / (a + 5e3f), 3e-4f), 5.f);
    gain[tid] = g;
    float s2 = 0.f, da = 0.f;
    for (int k = 0; k < NB; ++k) {
      const float d = g * BD[tid * NB + k], r = BR[tid * NB + k];
      const float lr = loud(r, thr[k], zp[k], at[k]), ld = loud(d, thr[k], zp[k], at[k]);
      const float sym = fmaxf(fabsf(ld - lr) - 0.25f * fminf(lr, ld), 0.f);
      const float as = powf((d + 50.f) / (r + 50.f), 1.2f);
      const float af = as < 3.f ? 0.f : fminf(as, 12.f);
      const float sw = sym * wid[k];
      s2 += sw * sw + 1e-8f;
      da += af * sw;
    }
    float stw = 0.f;
    for (int k = 0; k < NB; ++k) stw += wid[k];
    stw = sqrtf(stw);
    const float w = powf((apr[tid] + 1e5f) / 1e7f, 0.04f);
    wgt[tid] = w;
    const float df = sqrtf(s2) * stw;
    lossf[tid] = (0.1f * fminf(df / w, 45.f) + 0.0309f * fminf(da / w, 45.f)) / (float)T;
    if (BWD) {
      // chain rule of the frame: d loss / d BD2[k] -> G, then through the gain equaliser -> d loss / d BD1[k]
      const float g_df = df / w < 45.f ? 0.1f / ((float)T * w) : 0.f, g_da = da / w < 45.f ? 0.0309f / ((float)T * w) : 0.f;
      float gg = 0.f;
      for (int k = 0; k < NB; ++k) {
        const float b1 = BD[tid * NB + k], d = g * b1, r = BR[tid * NB + k];
        const float lr = loud(r, thr[k], zp[k], at[k]), ld = loud(d, thr[k], zp[k], at[k]);
        const float diff = fabsf(ld - lr) - 0.25f * fminf(lr, ld);
        const float sym = fmaxf(diff, 0.f);
        const float as = powf((d + 50.f) / (r + 50.f), 1.2f);
        const float af = as < 3.f ? 0.f : fminf(as, 12.f);
        const float g_sym = g_df * stw * sym * wid[k] * wid[k] / sqrtf(s2) + g_da * wid[k] * af;
        const float g_as = (as >= 3.f && as < 12.f) ? g_da * wid[k] * sym : 0.f;
        float g_ld = 0.f;
        if (diff > 0.f) g_ld = g_sym * ((ld > lr ? 1.f : (ld < lr ? -1.f : 0.f)) - (ld < lr ? 0.25f : 0.f));
        float g_d = g_as * 1.2f * as / (d + 50.f);
        if (d >= thr[k]) g_d += g_ld * at[k] * zp[k] * powf(0.5f + 0.5f * d / thr[k], zp[k] - 1.f) * 0.5f / thr[k];
        G[tid * NB + k] = g_d;
        gg += g_d * b1;
      }
      const float num = apr[tid] + 5e3f, den = a + 5e3f, raw = num / den;
      const float g_apd = (raw > 3e-4f && raw < 5.f) ? -gg * num / (den * den) : 0.f;
      for (int k = 0; k < NB; ++k) G[tid * NB + k] = g * G[tid * NB + k] + (BD[tid * NB + k] > thr[k] ? g_apd : 0.f);
    }
  }
  __syncthreads();
  if (!BWD) {
    if (tid < 64) {
      float v = lossf[tid < T ? tid : 0];
      if (tid >= T) v = 0.f;
      v = wave_sum(v);
      if (tid == 0) ws.pw[blockIdx.x] = v;
    }
    return;
  }
  if (BWD) {
    if (tid < NB) {                                // through the frequency equaliser -> d loss / d BD (before equalisation)
      float ge = 0.f;
      for (int t = 0; t < T; ++t) ge += G[t * NB + tid] * BD[t * NB + tid];
      ge /= eq[tid];                               // BD (pre) = BD1 / eq
      const float raw = (avr[tid] + 1000.f) / (avd[tid] + 1000.f);
      const float g_avd = (raw > 0.01f && raw < 100.f) ? -ge * (avr[tid] + 1000.f) / ((avd[tid] + 1000.f) * (avd[tid] + 1000.f)) : 0.f;
      for (int t = 0; t < T; ++t)
        G[t * NB + tid] = eq[tid] * G[t * NB + tid] + ((ns[t] && BR[t * NB + tid] >= 100.f * thr[tid]) ? g_avd : 0.f);
    }
    __syncthreads();
    float gs = 0.f;                                // d loss / d (SLL scale) = sum G * bark0;  d loss / d bark0 = scale * G
    for (int i = tid; i < T * NB; i += 256) {
      const float b0 = ws.bark0[(int64_t)e * T * NB + i];
      gs += G[i] * b0;
      ws.gbark[(int64_t)e * T * NB + i] = sd * G[i];
    }
    gs = wave_sum(gs);
    if ((tid & 63) == 0) red[tid >> 6] = gs;
    __syncthreads();
    if (tid == 0) ws.gm[e] = -(red[0] + red[1] + red[2] + red[3]) * sd / md;
  }
}

__global__ __launch_bounds__(256) void pmsqe_pit_kernel(int B, int S, Ws ws, float* loss_out) {
  __shared__ float red[4];
  int nperm = 1;
  for (int i = 2; i <= S; ++i) nperm *= i;
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    float best = 0.f; int bestp = -1;
    for (int p = 0; p < nperm; ++p) {            // lexicographic order of the permutations (factorial number system)
      int pool[MAXS], rem = p, f = nperm;
      for (int i = 0; i < S; ++i) pool[i] = i;
      float v = 0.f;
      for (int i = 0; i < S; ++i) {
        f /= (S - i);
        const int d = rem / f; rem %= f;
        v += ws.pw[((int64_t)b * S + i) * S + pool[d]];
        for (int q = d; q < S - 1 - i; ++q) pool[q] = pool[q + 1];
      }
      v /= (float)S;
      if (bestp < 0 || v < best) { best = v; bestp = p; }
    }
    int pool[MAXS], rem = bestp, f = nperm;
    for (int i = 0; i < S; ++i) pool[i] = i;
    for (int i = 0; i < S; ++i) {
      f /= (S - i);
      const int d = rem / f; rem %= f;
      ws.sel[b * S + i] = pool[d];
      for (int q = d; q < S - 1 - i; ++q) pool[q] = pool[q + 1];
    }
    ws.perloss[b] = best;
    acc += best;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *loss_out = (red[0] + red[1] + red[2] + red[3]) / (float)B;
}

__global__ __launch_bounds__(512) void pmsqe_istft_kernel(int B, int S, int power, const float* tab, const int32_t* itab, Ws ws,
                                                          const float* gscale, float* gwav) {
  __shared__ float gre[FR][NBINS + 3], gim[FR][NBINS + 3];
  const int e = blockIdx.y, t0 = blockIdx.x * FR, tid = threadIdx.x;
  const float up = (gscale ? *gscale : 1.f) / (float)(B * S);                 // mean over the S seconds and the batch
  if (tid < NBINS) {
    const int band = itab[IO_OF + tid];
    const float gmean = ws.gm[e] * tab[O_MASK + tid] / (float)(NBINS * T), corr = band >= 0 ? tab[O_CORR + band] : 0.f;
#pragma unroll
    for (int fr = 0; fr < FR; ++fr) {
      const int t = t0 + fr;
      float a = 0.f, b = 0.f;
      if (t < T) {
        const float* sp = ws.spec + (((int64_t)e * T + t) * NBINS + tid) * 2;
        const float re = sp[0], im = sp[1];
        float gx = gmean + (band >= 0 ? corr * ws.gbark[((int64_t)e * T + t) * NB + band] : 0.f);
        if (!power) gx *= 0.5f / sqrtf(re * re + im * im + 1e-8f);
        a = 2.f * re * gx * up; b = 2.f * im * gx * up;
      }
      gre[fr][tid] = a; gim[fr][tid] = b;
    }
  }
  __syncthreads();
  float acc[FR] = {};
  const float *Ct = tab + O_CT + tid, *St = tab + O_ST + tid;
  for (int f = 0; f < NBINS; ++f) {
    const float c = Ct[f * NFFT], s = St[f * NFFT];
#pragma unroll
    for (int fr = 0; fr < FR; ++fr) acc[fr] += gre[fr][f] * c + gim[fr][f] * s;
  }
#pragma unroll
  for (int fr = 0; fr < FR; ++fr)
    if (t0 + fr < T) atomicAdd(gwav + (int64_t)e * FS + (t0 + fr) * HOP + tid, acc[fr]);   // two frames per sample: the sum is order-independent
}

int check(int B, int L, int* S) {
  if (B <= 0 || L <= 0 || L % FS) return -1;       // view(N, -1, fs) of the reference needs whole seconds
  *S = L / FS;
  return *S > MAXS ? -1 : 0;
}
}  // namespace

extern "C" {
int64_t sefd_pmsqe_table_floats(void) { return TAB_FLOATS; }
int64_t sefd_pmsqe_ws_floats(int32_t B, int32_t L) {
  int S;
  return check(B, L, &S) ? -1 : ws_floats(B, S);
}
int32_t sefd_pmsqe_forward(const float* est, const float* clean, int32_t B, int32_t L, int32_t power, const float* tab, const int32_t* itab,
                           float* ws_mem, float* loss_out, void* stream) {
  int S;
  if (check(B, L, &S) || !est || !clean || !tab || !itab || !ws_mem || !loss_out) return -1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const Ws ws = carve(ws_mem, B, S);
  hipLaunchKernelGGL(pmsqe_stft_kernel, dim3((T + FR - 1) / FR, 2 * B * S), dim3(320), 0, st, est, clean, B * S, power, tab, itab, ws);
  hipLaunchKernelGGL((pmsqe_pair_kernel<false>), dim3(B * S * S), dim3(256), 0, st, B, S, tab, ws);
  hipLaunchKernelGGL(pmsqe_pit_kernel, dim3(1), dim3(256), 0, st, B, S, ws, loss_out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
int32_t sefd_pmsqe_backward(int32_t B, int32_t L, int32_t power, const float* tab, const int32_t* itab, float* ws_mem, const float* grad_scale,
                            float* grad_est, void* stream) {
  int S;
  if (check(B, L, &S) || !tab || !itab || !ws_mem || !grad_est) return -1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const Ws ws = carve(ws_mem, B, S);
  if (hipMemsetAsync(grad_est, 0, sizeof(float) * (size_t)B * L, st) != hipSuccess) return -2;
  hipLaunchKernelGGL((pmsqe_pair_kernel<true>), dim3(B * S), dim3(256), 0, st, B, S, tab, ws);
  hipLaunchKernelGGL(pmsqe_istft_kernel, dim3((T + FR - 1) / FR, B * S), dim3(512), 0, st, B, S, power, tab, itab, ws, grad_scale, grad_est);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
}
