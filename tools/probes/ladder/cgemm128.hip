// RUNGEMM, persistent 256 x 128 tile for the bf16 layers whose width is an odd multiple of 128 (round 6).
//
// What bounds the 128 x 128 kernel of rungemm.hip on these layers (enc1 / dec3 / dec4 and their input gradients: N = 128, M = 250 000 ... 990 000,
// 0.17-0.20 of the bf16 roof, 1.36 ms exclusive on the step's timeline for three rounds): a workgroup has ONE K tile in flight behind a 2-stage ring,
// the LDS-DMA round trip under load is 2400-6000 cycles (profiles/r03_tuning_notes.md) and a K tile holds 512 MFMA cycles per SIMD of work: two
// co-resident workgroups keep the matrix pipe ~28 % busy whatever else is tuned.  A deeper ring cost co-residency (r03: 20-60 % slower).  This kernel
// keeps the cgemm256.hip recipe - one workgroup per CU, persistent over output tiles, whole-line DMAs, weights K-tile major (kRunWTile32), one
// barrier per 64-deep K tile, DMAs behind MFMAs - on a 256 x 128 tile:
//   * 8 waves as 4 (M) x 2 (N), 64 x 64 accumulators each (two per SIMD); 48 KB of operands per K tile = 0.75 of the bytes per FLOP of the 128 x 128 tile;
//   * the stage stream runs TWO K tiles ahead of the multiply and is continuous across the output tiles a workgroup walks: A ring of 3 K tiles (96 KB),
//     B ring of 3 (6 sub-slots of 8 KB).  While K tile p is multiplied a thread issues B(p + 2) (steps 0-1, one DMA each) and A(p + 2) (steps 2-3, two
//     DMAs each) into the slots of K tile p - 1; at the top of K tile p + 1 it waits with `s_waitcnt vmcnt(6)`: only what was issued during K tile p may
//     still be in flight - every operand has more than one K tile (1024 MFMA cycles per SIMD) of cover and the queue is never drained;
//   * fragments of k16 step s + 1 are read while step s multiplies (two register sets: the 64 x 64 accumulators leave room);
//   * epilogue in registers as cgemm256.hip (quad transpose, 16-byte row pieces); the BatchNorm partial sums are per 128 rows = two waves: the upper
//     wave of a pair hands its 64-row sums to the lower one through 2 KB of LDS (lower + upper, the order in which rungemm.hip adds its two wave rows);
//     two-destination outputs (RunGemm::n2) supported.
// Same descriptor and same arithmetic as rungemm_kernel: k16 steps in run order into the same 32 x 32 accumulators, bias added to the sum, the same
// statistics order - outputs and partial sums are bit-identical to the kernel it replaces.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "sefd_desc.h"
#include "dev_common.h"

namespace sefd {

namespace {

__device__ __forceinline__ int fdiv1(int x, uint32_t m, uint32_t s) { return m ? (int)(__umulhi((uint32_t)x, m) >> s) : x; }

__device__ __forceinline__ int xcd_remap1(int bid, int nwg) {
  const int xcd = bid & 7, local = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + local;
}

__device__ __forceinline__ void bar1() { asm volatile("s_barrier" ::: "memory"); }

}  // namespace

__global__ __launch_bounds__(512) void cgemm128_kernel(const RunGemm d, const ArenaBases ab) {
  constexpr int BM = 256, BN = 128, NW = 8, KT = 64;
  constexpr int A_SLOT = BM * 128, B_SLOT = BN * 64;        // 32 KB (64-deep), 8 KB (32-deep sub-slot)
  constexpr int NAS = 3, NBK = 3;                           // K tiles in the A ring / in the B ring (two sub-slots each)
  constexpr int B_BASE = NAS * A_SLOT;
  constexpr int X_BASE = B_BASE + NBK * 2 * B_SLOT;         // 2 KB: statistics hand-over between the two waves of a 128-row block
  constexpr int WN_ = 2, MI = 2, NI = 2;
  constexpr int SMEM = X_BASE + 2048;
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = d.Npad / BN;
  const int nm = (d.M + BM - 1) / BM;
  const int total = nm * nn;
  const int TF = d.Tout * d.Fo;
  int nkt = 0;
  for (int s = 0; s < d.nseg; ++s) nkt += (d.seg[s].len + KT - 1) / KT;

  const uint16_t* x0 = reinterpret_cast<const uint16_t*>(rp(ab, d.x[0]));
  const uint16_t* x1 = d.x[1].arena >= 0 ? reinterpret_cast<const uint16_t*>(rp(ab, d.x[1])) : x0;
  const uint16_t* w = reinterpret_cast<const uint16_t*>(rp(ab, d.w));
  const uint16_t* zp = reinterpret_cast<const uint16_t*>(rp(ab, d.zero));
  // DMA roles.  A: instruction q (0..3) covers tile rows (q * 8 + wid) * 8 .. + 8; lane -> (row la, position pa); swizzle on the source chunk.
  const int la = lane >> 3, pa = lane & 7;
  const int csa = pa ^ ((((wid & 1) * 4) + (la >> 1)) & 7);
  // B: ONE instruction per 32-deep sub-tile covers columns wid * 16 .. + 16; lane -> (row lb, position pb)
  const int lb = lane >> 2, pb = lane & 3;
  const int csb = pb ^ ((lb >> 2) & 3);
  const int64_t wtile = (int64_t)d.Npad * 32;               // elements per 32-deep weight tile (K-tile major layout)
  const int Tin0 = d.Tin[0], Tin1 = d.Tin[1], fs0 = d.fstride[0], fs1 = d.fstride[1], rl0 = d.rowlen[0], rl1 = d.rowlen[1];
  const int64_t ts0 = d.tstride[0], ts1 = d.tstride[1];
  const uint32_t lds0 = lds_addr(smem);
  // fragment roles: wave tile (2 x 32) x (2 x 32) at (wm0, wn0)
  const int wr4 = wid / WN_, wc = wid % WN_;
  const int wm0 = wr4 * 64, wn0 = wc * 64;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int ca0 = (fhalf ^ ((frow >> 1) & 7)) * 16, cb0 = (fhalf ^ ((frow >> 2) & 3)) * 16;
  const int aoff = (wm0 + frow) * 128 + ca0, boff = B_BASE + (wn0 + frow) * 64 + cb0;
  const float* biasp = d.bias.arena >= 0 ? reinterpret_cast<const float*>(rp(ab, d.bias)) : nullptr;
  char* yb = rp(ab, d.y);
  const int n2 = d.n2 > 0 ? d.n2 : 0x7fffffff;             // columns >= n2 leave for the second destination, at column n - n2
  char* yb2 = d.n2 > 0 ? rp(ab, d.y2) - (int64_t)d.n2 * (d.ydt == DT_BF16 ? 2 : 4) : yb;
  const bool want_stats = d.stats.arena >= 0;
  const bool staged = (d.flags & kRunYAligned) && !(d.flags & kRunAccum);
  const bool ylin = d.y_tstride == d.Fo * d.y_fstride && d.y_bstride == (int64_t)d.Tout * d.y_tstride;

  // ---- stage stream: A cursor and B cursor, two K tiles ahead of the multiply, walking the workgroup's output tiles
  int a_t = blockIdx.x, b_t = blockIdx.x;                   // output tile (persistent walk) of the cursors; >= total: stream exhausted (zero-page DMAs)
  int a_g = 0, b_g = 0;                                     // ring slots of the cursors' K tiles (mod 3)
  int a_kl = 0, b_kl = 0;                                   // K tile index inside the output tile
  int aseg = 0, ak0 = 0, aseglen = 0;
  int bseg = 0, bk0 = 0, bseglen = 0, bkoff = 0;
  int a_mtile = 0;
  const uint16_t* rptr[4];                                  // this thread's 4 operand rows (instruction q) of the current run
  uint32_t lohi[4];                                         // valid element range [lo, hi) of the run for that row, lo | hi << 16
  const uint16_t* wb0 = w;
  auto enter_run = [&]() {
    const Seg sg = d.seg[aseg];
    aseglen = sg.len;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = a_mtile * BM + (q * NW + wid) * 8 + la;
      int lo = 0, hi = 0;
      const uint16_t* ptr = x0;
      if (sg.src >= 0 && m < d.M) {
        const int b = fdiv1(m, d.div_tf_m, d.div_tf_s), rem = m - b * TF;
        const int u = fdiv1(rem, d.div_fo_m, d.div_fo_s), fo = rem - u * d.Fo;
        const int s = sg.src;
        const int tt = u + sg.dt;
        if (tt >= 0 && tt < (s ? Tin1 : Tin0)) {
          const int rr = sg.off + fo * (s ? fs1 : fs0);
          lo = rr < 0 ? -rr : 0;
          hi = max(min(sg.len, (s ? rl1 : rl0) - rr), 0);
          ptr = (s ? x1 : x0) + (int64_t)b * d.bstride[s] + d.base[s] + (int64_t)tt * (s ? ts1 : ts0) + rr;
        }
      }
      rptr[q] = ptr;
      lohi[q] = (uint32_t)lo | ((uint32_t)hi << 16);
    }
  };
  auto a_begin_tile = [&]() {
    if (a_t < total) {
      a_mtile = xcd_remap1(a_t, total) / nn;
      aseg = 0; ak0 = 0; a_kl = 0;
      enter_run();
    }
  };
  auto b_begin_tile = [&]() {
    if (b_t < total) {
      const int ntile = xcd_remap1(b_t, total) % nn;
      wb0 = w + ((int64_t)ntile * BN + wid * 16 + lb) * 32 + csb * 8;
      bseg = 0; bk0 = 0; b_kl = 0; bseglen = d.seg[0].len; bkoff = d.seg[0].koff;
    }
  };
  // half hf (instructions 2 hf, 2 hf + 1) of the A cursor's K tile; the cursor moves on behind the second half
  auto stage_a = [&](int hf) {
    const uint32_t dst = lds0 + a_g * A_SLOT + wid * 1024;
    const int j0 = ak0 + csa * 8;
#pragma unroll
    for (int q = 2 * hf; q < 2 * hf + 2; ++q) {
      const uint32_t lh = lohi[q];
      const bool ok = a_t < total && j0 >= (int)(lh & 0xffffu) && j0 + 8 <= (int)(lh >> 16);
      const uint16_t* src = ok ? rptr[q] + j0 : zp;
      dma16(src, dst + q * (NW * 1024));
    }
    if (hf == 1) {
      a_g = a_g == NAS - 1 ? 0 : a_g + 1;
      if (a_t < total) {
        ak0 += KT;
        if (++a_kl == nkt) { a_t += gridDim.x; a_begin_tile(); }
        else if (ak0 >= aseglen) { ak0 = 0; ++aseg; enter_run(); }
      }
    }
  };
  // 32-deep half h of the B cursor's K tile; the cursor moves on behind the second half
  auto stage_b = [&](int h) {
    const uint32_t dst = lds0 + B_BASE + (b_g * 2 + h) * B_SLOT + wid * 1024;
    const uint16_t* src = b_t < total ? wb0 + (int64_t)(((bkoff + bk0) >> 5) + h) * wtile : zp;
    dma16(src, dst);
    if (h == 1) {
      b_g = b_g == NBK - 1 ? 0 : b_g + 1;
      if (b_t < total) {
        bk0 += KT;
        if (++b_kl == nkt) { b_t += gridDim.x; b_begin_tile(); }
        else if (bk0 >= bseglen) { bk0 = 0; ++bseg; bseglen = d.seg[bseg].len; bkoff = d.seg[bseg].koff; }
      }
    }
  };

  // ---- prologue: the first two K tiles of the stream
  a_begin_tile();
  b_begin_tile();
  stage_b(0); stage_b(1); stage_a(0); stage_a(1);
  stage_b(0); stage_b(1); stage_a(0); stage_a(1);

  int c_g = 0;                                              // ring slot of the K tile being multiplied
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    const int tile = xcd_remap1(t, total);
    const int ntile = tile % nn, mtile = tile / nn;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int p = 0; p < nkt; ++p) {
      const char* abase = smem + c_g * A_SLOT;
      const char* bbase = smem + c_g * 2 * B_SLOT;
      c_g = c_g == NAS - 1 ? 0 : c_g + 1;
      wait_vm<6>();                                        // this thread's part of K tile p has landed: only the six DMAs of the previous K tile's steps may be in flight
      bar1();                                              // ... everyone's part; and every wave has finished K tile p - 1: its slots may be refilled
      uint4 af[2][MI], bf[2][NI];
#pragma unroll
      for (int ii = 0; ii < MI; ++ii) af[0][ii] = *reinterpret_cast<const uint4*>(abase + ii * (32 * 128) + aoff);
#pragma unroll
      for (int j = 0; j < NI; ++j) bf[0][j] = *reinterpret_cast<const uint4*>(bbase + j * (32 * 64) + boff);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < 3) {                                       // fragments of the next k16 step, read while this one multiplies
#pragma unroll
          for (int ii = 0; ii < MI; ++ii) af[(s + 1) & 1][ii] = *reinterpret_cast<const uint4*>(abase + ii * (32 * 128) + (aoff ^ (32 * (s + 1))));
#pragma unroll
          for (int j = 0; j < NI; ++j) bf[(s + 1) & 1][j] = *reinterpret_cast<const uint4*>(bbase + ((s + 1) >> 1) * B_SLOT + j * (32 * 64) + (boff ^ (32 * ((s + 1) & 1))));
        }
#pragma unroll
        for (int ii = 0; ii < MI; ++ii) {
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[s & 1][ii]), __builtin_bit_cast(bf16x8, bf[s & 1][j]),
                                                                 acc[ii][j], 0, 0, 0);
          if (ii == 0) {                                   // this step's DMAs behind its first two MFMAs
            __builtin_amdgcn_sched_barrier(0);
            if (s == 0) stage_b(0);
            else if (s == 1) stage_b(1);
            else if (s == 2) stage_a(0);
            else stage_a(1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    // ---- epilogue, wave local, no LDS for the tile (as cgemm256.hip); the stream's DMAs for the next tile are in flight meanwhile
    float bv[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = ntile * BN + wn0 + j * 32 + (lane & 31);
      bv[j] = (biasp && n < d.N) ? biasp[n] : 0.f;
    }
    const QuadT qt(lane);
    const OctW ow(lane);
    float s1[NI], s2[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row0 = mtile * BM + wm0 + i * 32;
      if (staged) {
        const bool full = row0 + 32 <= d.M;
        int64_t ro[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = row0 + 8 * q + 4 * (lane >> 5) + (lane & 3);
          ro[q] = -1;
          if (m < d.M) {
            if (ylin) ro[q] = (int64_t)m * d.y_fstride + d.y_off;
            else {
              const int b = fdiv1(m, d.div_tf_m, d.div_tf_s), rem = m - b * TF, u = fdiv1(rem, d.div_fo_m, d.div_fo_s), fo = rem - u * d.Fo;
              ro[q] = (int64_t)b * d.y_bstride + (int64_t)u * d.y_tstride + (int64_t)fo * d.y_fstride + d.y_off;
            }
          }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int col = j * 32 + (lane & 31);
          const bool nok = ntile * BN + wn0 + col < d.N;
          const int n0 = ntile * BN + wn0 + j * 32 + (lane & 28);
          float v[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            v[e] = acc[i][j][e] + bv[j];
            if (d.flags & kRunRelu) v[e] = fmaxf(v[e], 0.f);
          }
          if (want_stats && nok) {
            if (full) {
#pragma unroll
              for (int e = 0; e < 16; ++e) { s1[j] += v[e]; s2[j] += v[e] * v[e]; }
            } else {
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row0 + row < d.M) { s1[j] += v[e]; s2[j] += v[e] * v[e]; }
              }
            }
          }
          uint2 pk[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) pk[q] = qt.pack(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
          uint4 wide[2];
          ow.widen(pk, wide);
          const int n8 = n0 & ~7;
          uint16_t* ydst = reinterpret_cast<uint16_t*>(n8 >= n2 ? yb2 : yb);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int64_t r = ow.hi4 ? ro[2 * h + 1] : ro[2 * h];
            if (r >= 0 && n8 < d.N) *reinterpret_cast<uint4*>(ydst + r + n8) = wide[h];
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = row0 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          if (m >= d.M) continue;
          const int b = fdiv1(m, d.div_tf_m, d.div_tf_s), rem = m - b * TF, u = fdiv1(rem, d.div_fo_m, d.div_fo_s), fo = rem - u * d.Fo;
          const int64_t o = (int64_t)b * d.y_bstride + (int64_t)u * d.y_tstride + (int64_t)fo * d.y_fstride + d.y_off;
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const int n = ntile * BN + wn0 + j * 32 + (lane & 31);
            if (n >= d.N) continue;
            char* yd = n >= n2 ? yb2 : yb;
            float v = acc[i][j][e] + bv[j];
            if (d.flags & kRunAccum) v += reinterpret_cast<float*>(yd)[o + n];
            if (d.flags & kRunRelu) v = fmaxf(v, 0.f);
            if (d.ydt == DT_BF16) reinterpret_cast<uint16_t*>(yd)[o + n] = f2bf(v);
            else reinterpret_cast<float*>(yd)[o + n] = v;
            s1[j] += v;
            s2[j] += v * v;
          }
        }
      }
    }
    if (want_stats) {                                        // partial sums per 128 rows = this wave pair (wr4 even: rows 0-63, odd: rows 64-127 of the block)
      float* part = reinterpret_cast<float*>(rp(ab, d.stats));
      float* xch = reinterpret_cast<float*>(smem + X_BASE) + ((wr4 >> 1) * WN_ + wc) * (NI * 2 * 32);
      float t1[NI], t2[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        t1[j] = s1[j] + __shfl_xor(s1[j], 32);
        t2[j] = s2[j] + __shfl_xor(s2[j], 32);
        if ((wr4 & 1) && lane < 32) { xch[(j * 2 + 0) * 32 + lane] = t1[j]; xch[(j * 2 + 1) * 32 + lane] = t2[j]; }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      bar1();
      const int srow = mtile * 2 + (wr4 >> 1);
      const int nrows = (d.M + kBM - 1) / kBM;
      if (!(wr4 & 1) && lane < 32 && srow < nrows) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int n = ntile * BN + wn0 + j * 32 + lane;
          part[((int64_t)srow * 2 + 0) * d.Npad + n] = t1[j] + xch[(j * 2 + 0) * 32 + lane];
          part[((int64_t)srow * 2 + 1) * d.Npad + n] = t2[j] + xch[(j * 2 + 1) * 32 + lane];
        }
      }
    }
  }
  wait_vm<0>();                                             // nothing may still be on its way into this workgroup's LDS when it ends
}

// The planner marks the GEMMs of this kernel (and packs their weights K-tile major) with kRunWTile32; widths that are a multiple of 256 take cgemm256.hip.
bool launch_cgemm128(const RunGemm& d, const ArenaBases& ab, hipStream_t st) {
  if (!(d.flags & kRunWTile32) || d.Npad % 256 == 0 || d.Npad % 128 != 0 || (d.flags & kRunBnBwd)) return false;
  static const int ncu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
  const int total = ((d.M + 255) / 256) * (d.Npad / 128);
  hipLaunchKernelGGL(cgemm128_kernel, dim3(total < ncu ? total : ncu), dim3(512), 0, st, d, ab);
  return true;
}

}  // namespace sefd
