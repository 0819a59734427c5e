// RUNGEMM, direct-operand variant for the THIN conv layers (bf16, N <= 64: enc0 / enc1 / dec3-dec5 and the input-gradient GEMMs next to
// them - < 5 % of the step's FLOPs, ~25 % of its GEMM time on the tiled kernel).
//
// What bounds those layers on rungemm_kernel (measured, profiles/r03_tuning_notes.md): not HBM and not the MFMAs - the implicit-GEMM
// A tile (128 rows x 64 k) is re-fetched through LDS-DMA once per K tile with a workgroup barrier behind it, one tile of prefetch,
// 25 % zero-page padding of the 96-long runs, and with N = 32 a K tile feeds 4 MFMAs per wave: every K tile exposes the L2 latency
// (~2400 cycles per tile observed) and deeper rings lose more occupancy than they hide (S = 3 / 4: 20-60 % slower).  Here:
//   * no LDS round trip and no barrier for the activation operand: lane (row r, half h) of a wave loads its 32x32x16 MFMA fragment -
//     8 consecutive channels of one tap, 16 bytes - straight from L2 into registers; out-of-range taps are a predicated zero, so runs
//     are walked in 16-element steps with NO padding to the 64-element K tile;
//   * waves are independent: a wave owns 128 consecutive rows (= one BatchNorm statistics block, 4 row blocks of 32), keeps D fragment
//     loads in flight (plain global loads, counted vmcnt by the compiler) and never synchronises with the other waves after the
//     weights are staged;
//   * the weights (N x K <= 96 KB, shared by the 8 waves of a workgroup) are staged once into LDS with a 16-byte row pad
//     (conflict-free ds_read_b128 of 32 different rows);
//   * epilogue per wave: bias, BatchNorm partial sums from the rounded values, 32 x 32 piece transposed through 2 KB of wave-private
//     LDS into 16-byte row-chunk stores.
// Same descriptor, same results as rungemm_kernel (the MFMA accumulation order over k differs only by the skipped zero padding).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "sefd_desc.h"
#include "tuning.h"
#include "dev_common.h"

namespace sefd {

namespace {

__device__ __forceinline__ int fdiv3(int x, uint32_t m, uint32_t s) { return m ? (int)(__umulhi((uint32_t)x, m) >> s) : x; }

constexpr int kThinWaves = 8;          // waves per workgroup
constexpr int kThinRows = 128;         // rows per wave (one statistics block)
constexpr int kThinDepth = 8;          // fragment loads in flight per wave

// NI: 32-column tiles (Npad / 32)
template <int NI, bool BNB>
__global__ __launch_bounds__(kThinWaves * 64) void rundirect_kernel(const RunGemm d, const ArenaBases ab, const int wpitch /* bytes per LDS weight row */) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = kThinDepth;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Npad = NI * 32;
  // ---- stage the packed weights [Npad][ldw] into LDS rows of `wpitch` bytes
  {
    const uint4* w = reinterpret_cast<const uint4*>(rp(ab, d.w));
    const int cpr = d.ldw / 8;                                  // 16-byte chunks per row
    for (int i = tid; i < Npad * cpr; i += kThinWaves * 64) {
      const int n = i / cpr, c = i - n * cpr;
      *reinterpret_cast<uint4*>(smem + n * wpitch + c * 16) = w[i];
    }
  }
  __syncthreads();
  char* stg = smem + Npad * wpitch + wid * (32 * 32 * 2);       // wave-private 32 x 32 bf16 staging piece

  const uint16_t* x0 = reinterpret_cast<const uint16_t*>(rp(ab, d.x[0]));
  const uint16_t* x1 = d.x[1].arena >= 0 ? reinterpret_cast<const uint16_t*>(rp(ab, d.x[1])) : x0;
  const uint16_t* zp = reinterpret_cast<const uint16_t*>(rp(ab, d.zero));
  const float* biasp = d.bias.arena >= 0 ? reinterpret_cast<const float*>(rp(ab, d.bias)) : nullptr;
  uint16_t* yb = reinterpret_cast<uint16_t*>(rp(ab, d.y));
  uint16_t* yb2 = d.n2 > 0 ? reinterpret_cast<uint16_t*>(rp(ab, d.y2)) - d.n2 : yb;      // second destination of the columns n >= n2
  const bool want_stats = d.stats.arena >= 0;
  const int TF = d.Tout * d.Fo;
  const int Tin0 = d.Tin[0], Tin1 = d.Tin[1], fs0 = d.fstride[0], fs1 = d.fstride[1], rl0 = d.rowlen[0], rl1 = d.rowlen[1];
  const int64_t ts0 = d.tstride[0], ts1 = d.tstride[1];
  const int frow = lane & 31, fh = lane >> 5;
  const int nseg = d.nseg;
  int total_steps = 0;
  for (int s = 0; s < nseg; ++s) total_steps += (d.seg[s].len + 15) >> 4;

  const int blk = blockIdx.x * kThinWaves + wid;               // this wave's 128-row block
  const int m0 = blk * kThinRows;
  if (m0 >= d.M) return;
  float bv[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) { const int n = j * 32 + frow; bv[j] = (biasp && n < d.N) ? biasp[n] : 0.f; }
  float s1[NI], s2[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) s1[j] = s2[j] = 0.f;
  // kRunBnBwd: the three backward sums are formed in the chunk loop of the epilogue (lane -> chunk column lane & 3 of every column tile:
  // 8 fixed columns per tile), against coalesced 16-byte chunks of the BatchNorm layer's forward output
  constexpr bool bnb = BNB;
  const uint16_t* ybn = bnb ? reinterpret_cast<const uint16_t*>(rp(ab, d.bnb_y)) : nullptr;
  const float bslope = bnb ? *reinterpret_cast<const float*>(rp(ab, d.bnb_slope)) : 0.f;
  constexpr int NB = BNB ? NI : 1;                             // (no state in the plain instantiation)
  float pm[NB][8], pis[NB][8], pg[NB][8], pb[NB][8], t0[NB][8], t1[NB][8], t2[NB][8];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      pm[j][e] = pis[j][e] = pg[j][e] = pb[j][e] = 0.f;
      t0[j][e] = t1[j][e] = t2[j][e] = 0.f;
      const int n = j * 32 + (lane & 3) * 8 + e;
      if (bnb && n < d.N) {
        const float* mi = reinterpret_cast<const float*>(rp(ab, d.bnb_mi));
        pm[j][e] = mi[n]; pis[j][e] = mi[d.N + n];
        pg[j][e] = reinterpret_cast<const float*>(rp(ab, d.bnb_gamma))[n];
        pb[j][e] = reinterpret_cast<const float*>(rp(ab, d.bnb_beta))[n];
      }
    }
  const char* wl = smem + frow * wpitch + fh * 16;             // this lane's weight row (column n = frow of tile 0) at k half fh

  for (int rb = 0; rb < kThinRows / 32; ++rb) {
    const int mrow0 = m0 + rb * 32;
    if (mrow0 >= d.M) break;
    const int m = mrow0 + frow;
    const bool rv = m < d.M;
    const int mm = rv ? m : 0;
    const int b = fdiv3(mm, d.div_tf_m, d.div_tf_s), rem = mm - b * TF;
    const int u = fdiv3(rem, d.div_fo_m, d.div_fo_s), fo = rem - u * d.Fo;
    const int64_t rb0 = (int64_t)b * d.bstride[0] + d.base[0], rb1 = (int64_t)b * d.bstride[1] + d.base[1];
    // fetch cursor (runs D steps ahead of the multiply cursor): run, its pointer / valid range for this lane's row, position
    int fseg = 0, fj = 0, flen = 0, flo = 0, fhi = 0;
    const uint16_t* fptr = x0;
    auto enter = [&](int sgi) {
      const Seg sg = d.seg[sgi];
      flen = sg.len; flo = 0; fhi = 0; fptr = x0;
      if (sg.src >= 0 && rv) {
        const int s = sg.src;
        const int tt = u + sg.dt;
        if (tt >= 0 && tt < (s ? Tin1 : Tin0)) {
          const int rr = sg.off + fo * (s ? fs1 : fs0);
          flo = rr < 0 ? -rr : 0;
          fhi = min(sg.len, (s ? rl1 : rl0) - rr);
          fptr = (s ? x1 : x0) + (s ? rb1 : rb0) + (int64_t)tt * (s ? ts1 : ts0) + rr;
        }
      }
    };
    // Every load is issued unconditionally (an invalid chunk reads the zero page, steps past the end too): with loads inside
    // conditional regions hipcc cannot count them and drains the queue (vmcnt(0)) in front of every MFMA group.
    auto fetch = [&]() -> uint4 {
      const int j0 = fj + fh * 8;
      const uint16_t* p = (j0 >= flo && j0 + 8 <= fhi) ? fptr + j0 : zp;
      const uint4 v = *reinterpret_cast<const uint4*>(p);
      fj += 16;
      if (fj >= flen) {
        if (fseg + 1 < nseg) { ++fseg; fj = 0; enter(fseg); }
        else { flo = 0; fhi = 0; }                               // past the last run: zero page from here on
      }
      return v;
    };
    enter(0);
    uint4 abuf[D];
#pragma unroll
    for (int i = 0; i < D; ++i) abuf[i] = fetch();
    f32x16 acc[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    // multiply cursor: run, position -> weight column koff + position (parks on the last step once past the end: zero A x finite B)
    int cseg = 0, cj = 0, clen = d.seg[0].len, ckoff = d.seg[0].koff;
    for (int st = 0; st < total_steps; st += D) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const uint4 a = abuf[i];
        abuf[i] = fetch();
        const char* wk = wl + (ckoff + cj) * 2;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const uint4 bw = *reinterpret_cast<const uint4*>(wk + j * 32 * wpitch);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bw), acc[j], 0, 0, 0);
        }
        cj += 16;
        if (cj >= clen) {
          if (cseg + 1 < nseg) { ++cseg; cj = 0; clen = d.seg[cseg].len; ckoff = d.seg[cseg].koff; }
          else cj -= 16;
        }
      }
    }
    // ---- epilogue of this 32-row block: lane = column n, 16 rows; transposed through the wave's staging piece
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = j * 32 + frow;
      const bool nok = n < d.N;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * fh;
        const float v = acc[j][e] + bv[j];
        *reinterpret_cast<uint16_t*>(stg + row * 64 + (((frow >> 3) ^ ((row >> 1) & 3)) << 4) + (frow & 7) * 2) = f2bf(v);
        if (!bnb && mrow0 + row < d.M && nok) { s1[j] += v; s2[j] += v * v; }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {                          // 32 rows x 4 chunks of 16 B = 128 chunks, 2 per lane
        const int c = lane + 64 * c2, row = c >> 2, ch = c & 3;
        const uint4 v = *reinterpret_cast<const uint4*>(stg + row * 64 + ((ch ^ ((row >> 1) & 3)) << 4));
        const int mo = mrow0 + row, n0 = j * 32 + ch * 8;
        if (mo < d.M && n0 < d.N) {
          const int bb = fdiv3(mo, d.div_tf_m, d.div_tf_s), rem2 = mo - bb * TF, uu = fdiv3(rem2, d.div_fo_m, d.div_fo_s), ff = rem2 - uu * d.Fo;
          const int64_t o = (int64_t)bb * d.y_bstride + (int64_t)uu * d.y_tstride + (int64_t)ff * d.y_fstride + d.y_off;
          *reinterpret_cast<uint4*>((d.n2 > 0 && n0 >= d.n2 ? yb2 : yb) + o + n0) = v;
          if constexpr (BNB) {
            const uint4 yv = *reinterpret_cast<const uint4*>(ybn + (int64_t)bb * d.bnb_bstride + (int64_t)uu * d.bnb_tstride + (int64_t)ff * d.bnb_fstride + d.bnb_off + n0);
            const uint32_t dw[4] = {v.x, v.y, v.z, v.w}, yw[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float dz = bf2f((uint16_t)(dw[e >> 1] >> (16 * (e & 1)))), yy = bf2f((uint16_t)(yw[e >> 1] >> (16 * (e & 1))));
              const float xh = (yy - pm[j][e]) * pis[j][e];
              const float bn = pg[j][e] * xh + pb[j][e];
              const float dbn = bn > 0.f ? dz : bslope * dz;
              t0[j][e] += dbn;
              t1[j][e] += dbn * xh;
              t2[j][e] += bn > 0.f ? 0.f : bn * dz;
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the piece is rewritten by the next column tile / row block
    }
  }
  if constexpr (BNB) {
   if (want_stats) {
    float* part = reinterpret_cast<float*>(rp(ab, d.stats));
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = 32; o >= 4; o >>= 1) { t0[j][e] += __shfl_xor(t0[j][e], o); t1[j][e] += __shfl_xor(t1[j][e], o); t2[j][e] += __shfl_xor(t2[j][e], o); }
        const int n = j * 32 + (lane & 3) * 8 + e;
        if (lane < 4) {
          part[((int64_t)blk * 3 + 0) * d.Npad + n] = t0[j][e];
          part[((int64_t)blk * 3 + 1) * d.Npad + n] = t1[j][e];
          part[((int64_t)blk * 3 + 2) * d.Npad + n] = t2[j][e];
        }
      }
   }
  } else if (want_stats) {
    float* part = reinterpret_cast<float*>(rp(ab, d.stats));
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const float t1 = s1[j] + __shfl_xor(s1[j], 32), t2 = s2[j] + __shfl_xor(s2[j], 32);
      const int n = j * 32 + frow;
      if (lane < 32) {
        part[((int64_t)blk * 2 + 0) * d.Npad + n] = t1;
        part[((int64_t)blk * 2 + 1) * d.Npad + n] = t2;
      }
    }
  }
}

}  // namespace

// Chosen by shape: bf16 in / out, LDS-DMA-able (aligned) runs, no accumulate / ReLU, N <= 64 and the packed weights fit LDS next to the
// staging pieces.  SEFD_DIRECT=0 keeps the tiled kernel (A/B runs); SEFD_DIRECT_MAXK bounds ldw.
bool launch_rundirect(const RunGemm& d, const ArenaBases& ab, hipStream_t st) {
  static const bool on = !(tune_str("DIRECT") && atoi(tune_str("DIRECT")) == 0);
  // measured (profiles/r03_tuning_notes.md): fragment-shaped loads run at the line rate of the vector memory pipe (32 rows x 32 B per
  // instruction: ~6 TB/s of operand bytes), so the kernel wins only where K is tiny (enc0, the mask layer's input gradient: ldw = 128,
  // 92 -> 64-72 us) and loses from K = 320 on (dec4: 139 -> 255 us); larger K stays on the tiled LDS-DMA kernel
  static const int maxk = tune_str("DIRECT_MAXK") ? atoi(tune_str("DIRECT_MAXK")) : 128;
  const char* em = tune_str("DIRECT_MINM");               // read per launch: the per-op test lowers it for one small case
  const int minm = em ? atoi(em) : 65536;
  if (!on || d.xdt != DT_BF16 || d.ydt != DT_BF16 || !(d.flags & kRunAligned) || !(d.flags & kRunYAligned)) return false;
  if ((d.flags & (kRunAccum | kRunRelu | kRunWTile32)) || d.Npad > 64 || d.ldw > maxk || d.M < minm || d.ldw % 8 != 0) return false;
  for (int s = 0; s < d.nseg; ++s) if (d.seg[s].src < 0 || d.seg[s].len % 8 != 0) return false;
  const int wpitch = d.ldw * 2 + 16;
  const size_t lds = (size_t)d.Npad * wpitch + kThinWaves * 2048;
  if (lds > 100 * 1024) return false;
  const int blocks = (d.M + kThinRows - 1) / kThinRows;
  const dim3 grid((blocks + kThinWaves - 1) / kThinWaves);
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL(kern, grid, dim3(kThinWaves * 64), lds, st, d, ab, wpitch);
  };
  const bool bnb = (d.flags & kRunBnBwd) != 0;
  if (d.Npad == 32) { if (bnb) go(&rundirect_kernel<1, true>); else go(&rundirect_kernel<1, false>); }
  else { if (bnb) go(&rundirect_kernel<2, true>); else go(&rundirect_kernel<2, false>); }
  return true;
}

}  // namespace sefd
