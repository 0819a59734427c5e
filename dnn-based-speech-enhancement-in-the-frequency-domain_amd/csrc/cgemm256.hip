// RUNGEMM, wide-tile variant for the layers that carry the FLOPs (N >= 128, bf16, LDS-DMA-able runs).
//
// Why a second kernel: the 128 x 128 kernel of rungemm.hip moves 32 KB of operands through L2 -> LDS per 2.1 MFLOP
// (64 FLOP/B); at the 2.5 PFLOP/s MFMA peak that would be 39 TB/s, more than the L2s deliver, and its one-barrier-per-K-tile
// loop exposes the DMA / fragment-read latency of every K tile (measured: MFMA busy 23 %, SQ_WAIT_ANY 45 %).  Here:
//   * 256 x 256 (or 256 x 128) output tile per 512-thread workgroup: 128 FLOP per operand byte, and for the N = 256 layers
//     the activation operand is streamed ONCE per M tile instead of once per 128-column tile;
//   * the 8 waves form two groups of 4 (one wave of each group per SIMD) that run half a K tile apart: while group A
//     multiplies K tile i out of registers, group B issues its LDS-DMAs and reads its fragments of tile i, then they swap.
//     The matrix pipe of every SIMD always has one wave in its MFMA phase;
//   * LDS-DMA ring of S stages with counted `s_waitcnt vmcnt(N)`: S-2 K tiles stay in flight across every barrier, the
//     queue is never drained inside the loop;
//   * XOR-swizzled K rows (64 or 128 bytes) so that the ds_read_b128 fragment reads are bank-conflict free; the swizzle is
//     applied on the DMA *source* chunk index (the DMA destination is lane-linear).
// Same descriptor (sefd_desc.h RunGemm), same epilogue contract (bias, ReLU, accumulate, BatchNorm partial sums per 128 rows,
// bf16 tile staged through LDS and stored as 16-byte row chunks) as rungemm_kernel.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "sefd_desc.h"
#include "dev_common.h"

namespace sefd {

namespace {

__device__ __forceinline__ int fdiv2(int x, uint32_t m, uint32_t s) { return m ? (int)(__umulhi((uint32_t)x, m) >> s) : x; }

__device__ __forceinline__ int xcd_remap2(int bid, int nwg) {
  const int xcd = bid & 7, local = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + local;
}

__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }

// wait until at most `rem` tiles (NL DMAs each) of this thread's DMAs are still in flight; rem is wave-uniform, 0 <= rem <= S-2
template <int NL, int S>
__device__ __forceinline__ void wait_tiles(int rem) {
  if constexpr (S >= 4) { if (rem >= 2) { wait_vm<2 * NL>(); return; } }
  if constexpr (S >= 3) { if (rem >= 1) { wait_vm<NL>(); return; } }
  wait_vm<0>();
}

}  // namespace

// BM = 256 rows, BN columns, BK elements per K tile (32 or 64), S ring stages.  8 waves.
// dbg (tuning runs only, SEFD_CG256_DBG): 1 skip the MFMAs, 2 skip the DMAs, 4 skip the fragment reads, 8 skip the epilogue, 16 no s_setprio,
// 32 every A chunk from the zero page (no activation traffic)
template <int BN, int BK, int S, bool STAG>
__global__ __launch_bounds__(512) void cgemm256_kernel(const RunGemm d, const ArenaBases ab, const int dbg) {
  constexpr int BM = 256, NW = 8;
  constexpr int RB = BK * 2;                   // bytes per K row in LDS
  constexpr int CPR = RB / 16;                 // 16-byte chunks per row (4 or 8)
  constexpr int RPI = 1024 / RB;               // rows covered by one wave-level DMA instruction (16 or 8)
  constexpr int NA = BM / RPI / NW;            // A DMAs per thread per K tile
  constexpr int NB = BN / RPI / NW;            // B DMAs per thread per K tile
  constexpr int NL = NA + NB;
  constexpr int WN_ = BN == 256 ? 4 : 2;       // waves along N
  constexpr int WM_ = NW / WN_;                // waves along M
  constexpr int MI = BM / (32 * WM_);          // 32-row MFMA tiles per wave
  constexpr int NI = BN / (32 * WN_);
  constexpr int KS = BK / 16;                  // k16 steps per K tile
  constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, STAGE = A_BYTES + B_BYTES;
  constexpr int OS = BN + 8;                   // bf16 staging row stride (elements)
  constexpr int EPI = BM * 8 + WM_ * BN * 8 + BM * OS * 2;
  constexpr int SMEM = S * STAGE > EPI ? S * STAGE : EPI;
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  static_assert(NA >= 1 && NB >= 1, "tile too small for 8 waves");

  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;                    // 0: group A, 1: group B (half a K tile behind)
  const int nn = d.Npad / BN;
  const int nm = (d.M + BM - 1) / BM;
  const int swz = xcd_remap2(blockIdx.x, nm * nn);
  const int ntile = swz % nn, mtile = swz / nn;

  const uint16_t* x0 = reinterpret_cast<const uint16_t*>(rp(ab, d.x[0]));
  const uint16_t* x1 = d.x[1].arena >= 0 ? reinterpret_cast<const uint16_t*>(rp(ab, d.x[1])) : x0;
  const uint16_t* w = reinterpret_cast<const uint16_t*>(rp(ab, d.w));
  const uint16_t* zp = reinterpret_cast<const uint16_t*>(rp(ab, d.zero));

  // ---- DMA assignment: instruction q of this wave covers LDS rows (q*8 + wid)*RPI .. +RPI; lane -> (row lr, position pos)
  const int lr = lane / CPR, pos = lane % CPR;
  int csrc;                                    // source chunk of the K tile this lane fetches (swizzle on the source side)
  if constexpr (CPR == 4) csrc = pos ^ ((lr >> 2) & 3);
  else csrc = pos ^ (((4 * (wid & 1)) + (lr >> 1)) & 7);
  const int TF = d.Tout * d.Fo;
  int64_t rb0[NA], rb1[NA];
  int ru[NA], rfo[NA];
  bool rv[NA];
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int m = mtile * BM + (q * NW + wid) * RPI + lr;
    rv[q] = m < d.M;
    const int mm = rv[q] ? m : 0;
    const int b = fdiv2(mm, d.div_tf_m, d.div_tf_s), rem = mm - b * TF;
    ru[q] = fdiv2(rem, d.div_fo_m, d.div_fo_s);
    rfo[q] = rem - ru[q] * d.Fo;
    rb0[q] = (int64_t)b * d.bstride[0] + d.base[0];
    rb1[q] = (int64_t)b * d.bstride[1] + d.base[1];
  }
  const uint16_t* wrow[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) wrow[q] = w + (int64_t)(ntile * BN + (q * NW + wid) * RPI + lr) * d.ldw + csrc * 8;

  int ntiles = 0;
  for (int s = 0; s < d.nseg; ++s) ntiles += (d.seg[s].len + BK - 1) / BK;

  const int Tin0 = d.Tin[0], Tin1 = d.Tin[1], fs0 = d.fstride[0], fs1 = d.fstride[1], rl0 = d.rowlen[0], rl1 = d.rowlen[1];
  const int64_t ts0 = d.tstride[0], ts1 = d.tstride[1];
  const uint16_t* rptr[NA];
  int jlo[NA], jhi[NA];
  int seglen = 0, wseg = 0, seg = 0, k0 = 0;
  auto enter_run = [&](int sgi) {
    const Seg sg = d.seg[sgi];
    seglen = sg.len;
    wseg = sg.koff;
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      int lo = 0, hi = 0;
      const uint16_t* ptr = x0;
      if (sg.src >= 0 && rv[q]) {
        const int s = sg.src;
        const int tt = ru[q] + sg.dt;
        if (tt >= 0 && tt < (s ? Tin1 : Tin0)) {
          const int rr = sg.off + rfo[q] * (s ? fs1 : fs0);
          lo = rr < 0 ? -rr : 0;
          hi = min(sg.len, (s ? rl1 : rl0) - rr);
          ptr = (s ? x1 : x0) + (s ? rb1[q] : rb0[q]) + (int64_t)tt * (s ? ts1 : ts0) + rr;
        }
      }
      rptr[q] = ptr; jlo[q] = lo; jhi[q] = hi;
    }
  };
  const uint32_t lds0 = lds_addr(smem);
  int issued = 0, istage = 0;
  auto issue_tile = [&]() {                    // this thread's NL DMAs of the next K tile
    const uint32_t A = lds0 + istage * STAGE, B = A + A_BYTES;
    const int j0 = k0 + csrc * 8;
    if (!(dbg & 2)) {
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        const uint16_t* src = (j0 >= jlo[q] && j0 + 8 <= jhi[q] && !(dbg & 32)) ? rptr[q] + j0 : zp;
        dma16(src, A + (q * NW + wid) * 1024);
      }
#pragma unroll
      for (int q = 0; q < NB; ++q) dma16(wrow[q] + wseg + k0, B + (q * NW + wid) * 1024);
    }
    istage = istage + 1 == S ? 0 : istage + 1;
    ++issued;
    k0 += BK;
    if (k0 >= seglen && issued < ntiles) { k0 = 0; ++seg; enter_run(seg); }
  };

  // ---- fragment addressing: wave tile (MI x 32) x (NI x 32) at (wm0, wn0)
  const int wm0 = (wid / WN_) * (MI * 32), wn0 = (wid % WN_) * (NI * 32);
  const int frow = lane & 31, fhalf = lane >> 5;
  int fsw;                                     // swizzle term of this lane's fragment rows (tile bases are multiples of 32)
  if constexpr (CPR == 4) fsw = (frow >> 2) & 3; else fsw = (frow >> 1) & 7;
  int choff[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) choff[s] = ((2 * s + fhalf) ^ fsw) * 16;
  const int aoff = (wm0 + frow) * RB, boff = A_BYTES + (wn0 + frow) * RB;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  uint4 af[KS][MI], bf[KS][NI];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
#pragma unroll
    for (int i = 0; i < MI; ++i) af[s][i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NI; ++j) bf[s][j] = make_uint4(0, 0, 0, 0);
  }
  auto read_frags = [&](int stage) {
    if (dbg & 4) return;
    const char* base = smem + stage * STAGE;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int i = 0; i < MI; ++i) af[s][i] = *reinterpret_cast<const uint4*>(base + aoff + i * (32 * RB) + choff[s]);
#pragma unroll
      for (int j = 0; j < NI; ++j) bf[s][j] = *reinterpret_cast<const uint4*>(base + boff + j * (32 * RB) + choff[s]);
    }
  };
  auto multiply = [&]() {
    if (dbg & 1) return;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[s][i]), __builtin_bit_cast(bf16x8, bf[s][j]),
                                                              acc[i][j], 0, 0, 0);
  };

  // ---- prologue: S-1 tiles in flight, tile 0 landed everywhere
  enter_run(0);
  for (int i = 0; i < S - 1; ++i)
    if (issued < ntiles) issue_tile();
  wait_tiles<NL, S>(min(issued - 1, S - 2));
  wg_barrier();
  int cstage = 0;
  if constexpr (STAG) {
    if (grp == 1) wg_barrier();                // group B idles one phase: from here on it runs half a K tile behind group A
    for (int i = 0; i < ntiles; ++i) {
      // load phase: refill the slot tile i-1 was read from (both groups have finished reading it), fragments of tile i -> registers
      read_frags(cstage);                      // fragment reads first: their LDS latency runs under the DMA issue, which holds the wave
      if (issued < ntiles) issue_tile();
      if (grp == 1 && i + 1 < ntiles) wait_tiles<NL, S>(min(issued - (i + 2), S - 2));   // my part of tile i+1 has landed
      lds_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // multiply phase (the other group is in its load phase now)
      if (!(dbg & 16)) __builtin_amdgcn_s_setprio(1);
      multiply();
      if (!(dbg & 16)) __builtin_amdgcn_s_setprio(0);
      if (grp == 0 && i + 1 < ntiles) wait_tiles<NL, S>(min(issued - (i + 2), S - 2));
      wg_barrier();
      cstage = cstage + 1 == S ? 0 : cstage + 1;
    }
    if (grp == 0) wg_barrier();                // group A waits for B's last multiply phase: every LDS read is done, the ring is free
  } else {
    // lock-step variant (A/B reference): one barrier per K tile, all 8 waves load then multiply
    for (int i = 0; i < ntiles; ++i) {
      read_frags(cstage);
      if (issued < ntiles) issue_tile();
      if (i + 1 < ntiles) wait_tiles<NL, S>(min(issued - (i + 2), S - 2));
      lds_barrier();
      __builtin_amdgcn_sched_barrier(0);
      multiply();
      cstage = cstage + 1 == S ? 0 : cstage + 1;
    }
    wg_barrier();
  }
  if (dbg & 8) return;

  // ---- epilogue (same contract as rungemm_kernel): row address table, bias, store, BatchNorm partial statistics
  int64_t* rowoff = reinterpret_cast<int64_t*>(smem);              // [BM]
  float* stat = reinterpret_cast<float*>(smem + BM * 8);           // [WM_][BN][2]
  uint16_t* otile = reinterpret_cast<uint16_t*>(smem + BM * 8 + WM_ * BN * 8);   // [BM][OS] bf16 staging tile
  const bool staged = (d.flags & kRunYAligned) && !(d.flags & kRunAccum);
  if (tid < BM) {
    const int m = mtile * BM + tid;
    int64_t o = -1;
    if (m < d.M) {
      const int b = fdiv2(m, d.div_tf_m, d.div_tf_s), rem = m - b * TF, u = fdiv2(rem, d.div_fo_m, d.div_fo_s), fo = rem - u * d.Fo;
      o = (int64_t)b * d.y_bstride + (int64_t)u * d.y_tstride + (int64_t)fo * d.y_fstride + d.y_off;
    }
    rowoff[tid] = o;
  }
  __syncthreads();
  const float* bias = d.bias.arena >= 0 ? reinterpret_cast<const float*>(rp(ab, d.bias)) : nullptr;
  char* yb = rp(ab, d.y);
  const bool want_stats = d.stats.arena >= 0;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int nl = wn0 + j * 32 + (lane & 31);
    const int n = ntile * BN + nl;
    const float bv = (bias && n < d.N) ? bias[n] : 0.f;
    float s1 = 0.f, s2 = 0.f;                  // statistics of this wave's MI x 32 rows (all inside ONE 128-row half) of column n
    if (staged) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          float v = acc[i][j][e] + bv;
          if (d.flags & kRunRelu) v = fmaxf(v, 0.f);
          otile[row * OS + nl] = f2bf(v);
          if (mtile * BM + row < d.M && n < d.N) { s1 += v; s2 += v * v; }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          const int64_t o = rowoff[row];
          float v = acc[i][j][e] + bv;
          if (o >= 0 && n < d.N) {
            if (d.flags & kRunAccum) v += reinterpret_cast<float*>(yb)[o + n];
            if (d.flags & kRunRelu) v = fmaxf(v, 0.f);
            if (d.ydt == DT_BF16) reinterpret_cast<uint16_t*>(yb)[o + n] = f2bf(v);
            else reinterpret_cast<float*>(yb)[o + n] = v;
            s1 += v;
            s2 += v * v;
          }
        }
      }
    }
    if (want_stats) {
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (lane < 32) {
        stat[((wid / WN_) * BN + nl) * 2 + 0] = s1;
        stat[((wid / WN_) * BN + nl) * 2 + 1] = s2;
      }
    }
  }
  if (staged) {
    __syncthreads();
    constexpr int CPRO = BN / 8;                                   // 16-byte chunks per output tile row
    for (int q = tid; q < BM * CPRO; q += NW * 64) {
      const int row = q / CPRO, cc = q - row * CPRO;
      const int64_t o = rowoff[row];
      const int n0 = ntile * BN + cc * 8;
      if (o >= 0 && n0 < d.N)
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(yb) + o + n0) = *reinterpret_cast<const uint4*>(otile + row * OS + cc * 8);
    }
  }
  if (want_stats) {
    __syncthreads();
    if (tid < BN) {
      constexpr int HALVES = BM / kBM, WMH = WM_ / HALVES;         // wave rows per 128-row half
      float* part = reinterpret_cast<float*>(rp(ab, d.stats));
      const int nrows = (d.M + kBM - 1) / kBM;
#pragma unroll
      for (int h = 0; h < HALVES; ++h) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int wmi = h * WMH; wmi < (h + 1) * WMH; ++wmi) {
          s1 += stat[(wmi * BN + tid) * 2 + 0];
          s2 += stat[(wmi * BN + tid) * 2 + 1];
        }
        const int srow = mtile * HALVES + h;
        if (srow < nrows) {
          part[((int64_t)srow * 2 + 0) * d.Npad + ntile * BN + tid] = s1;
          part[((int64_t)srow * 2 + 1) * d.Npad + ntile * BN + tid] = s2;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Wave-specialised variant (256 x 256 tile): 8 consumer waves (2 x 4 grid, 128 x 64 each, two staggered groups of 4 as above)
// never touch global memory in the K loop, 4 producer waves (one per SIMD) do nothing but issue the LDS-DMAs.
// Measured on the non-specialised kernel (profiles/r02_tuning_notes.md): with the DMAs switched off the loop runs at 1090
// cycles per K tile (the MFMA rate is 1024), with the MFMAs switched off at 1250 (the LDS-DMA path delivers ~26 B/clk/CU),
// but together they take 2060-2600: a wave that issues a `global_load_lds` is held at issue for ~250 cycles while the
// DMA queue drains, and an in-order wave cannot multiply meanwhile.  Separate waves make the two streams independent.
template <int S>
__global__ __launch_bounds__(768) void cgemm_ws_kernel(const RunGemm d, const ArenaBases ab, const int dbg) {
  constexpr int BM = 256, BN = 256, BK = 32, NC = 8, NP = 4;
  constexpr int RB = BK * 2, CPR = 4, RPI = 16;
  constexpr int NA = BM / RPI / NP, NB = BN / RPI / NP;     // DMAs per producer thread per K tile (4 + 4)
  constexpr int NL = NA + NB;
  constexpr int WN_ = 4, MI = 4, NI = 2, KS = BK / 16;
  constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, STAGE = A_BYTES + B_BYTES;
  constexpr int OS = BN + 8;
  constexpr int EPI = BM * 8 + 2 * BN * 8 + BM * OS * 2;
  constexpr int SMEM = S * STAGE > EPI ? S * STAGE : EPI;
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = d.Npad / BN;
  const int nm = (d.M + BM - 1) / BM;
  const int swz = xcd_remap2(blockIdx.x, nm * nn);
  const int ntile = swz % nn, mtile = swz / nn;
  const int TF = d.Tout * d.Fo;
  int ntiles = 0;
  for (int s = 0; s < d.nseg; ++s) ntiles += (d.seg[s].len + BK - 1) / BK;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int wm0 = ((wid & 7) / WN_) * (MI * 32), wn0 = ((wid & 7) % WN_) * (NI * 32);

  if (wid >= NC) {
    // =============================================================================== producers
    const int pw = wid - NC;
    const uint16_t* x0 = reinterpret_cast<const uint16_t*>(rp(ab, d.x[0]));
    const uint16_t* x1 = d.x[1].arena >= 0 ? reinterpret_cast<const uint16_t*>(rp(ab, d.x[1])) : x0;
    const uint16_t* w = reinterpret_cast<const uint16_t*>(rp(ab, d.w));
    const uint16_t* zp = reinterpret_cast<const uint16_t*>(rp(ab, d.zero));
    const int lr = lane / CPR, pos = lane % CPR;
    const int csrc = pos ^ ((lr >> 2) & 3);
    int64_t rb0[NA], rb1[NA];
    int ru[NA], rfo[NA];
    bool rv[NA];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int m = mtile * BM + (q * NP + pw) * RPI + lr;
      rv[q] = m < d.M;
      const int mm = rv[q] ? m : 0;
      const int b = fdiv2(mm, d.div_tf_m, d.div_tf_s), rem = mm - b * TF;
      ru[q] = fdiv2(rem, d.div_fo_m, d.div_fo_s);
      rfo[q] = rem - ru[q] * d.Fo;
      rb0[q] = (int64_t)b * d.bstride[0] + d.base[0];
      rb1[q] = (int64_t)b * d.bstride[1] + d.base[1];
    }
    const uint16_t* wrow[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) wrow[q] = w + (int64_t)(ntile * BN + (q * NP + pw) * RPI + lr) * d.ldw + csrc * 8;
    const int Tin0 = d.Tin[0], Tin1 = d.Tin[1], fs0 = d.fstride[0], fs1 = d.fstride[1], rl0 = d.rowlen[0], rl1 = d.rowlen[1];
    const int64_t ts0 = d.tstride[0], ts1 = d.tstride[1];
    const uint16_t* rptr[NA];
    int jlo[NA], jhi[NA];
    int seglen = 0, wseg = 0, seg = 0, k0 = 0;
    auto enter_run = [&](int sgi) {
      const Seg sg = d.seg[sgi];
      seglen = sg.len;
      wseg = sg.koff;
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        int lo = 0, hi = 0;
        const uint16_t* ptr = x0;
        if (sg.src >= 0 && rv[q]) {
          const int s = sg.src;
          const int tt = ru[q] + sg.dt;
          if (tt >= 0 && tt < (s ? Tin1 : Tin0)) {
            const int rr = sg.off + rfo[q] * (s ? fs1 : fs0);
            lo = rr < 0 ? -rr : 0;
            hi = min(sg.len, (s ? rl1 : rl0) - rr);
            ptr = (s ? x1 : x0) + (s ? rb1[q] : rb0[q]) + (int64_t)tt * (s ? ts1 : ts0) + rr;
          }
        }
        rptr[q] = ptr; jlo[q] = lo; jhi[q] = hi;
      }
    };
    const uint32_t lds0 = lds_addr(smem);
    int issued = 0, istage = 0;
    // the NL DMAs of a tile are spread over the 2 * KS phases of a K tile (2 per phase), the K position advances with the last part
    constexpr int NPH = 2 * KS, PER = NL / NPH;
    static_assert(NL % NPH == 0, "DMAs must divide evenly over the phases");
    auto issue_part = [&](int ph) {
      if (!(dbg & 2)) {
        const uint32_t A = lds0 + istage * STAGE, B = A + A_BYTES;
        const int j0 = k0 + csrc * 8;
#pragma unroll
        for (int q = 0; q < NL; ++q) {
          if (q / PER != ph) continue;
          if (q < NA) {
            const uint16_t* src = (j0 >= jlo[q] && j0 + 8 <= jhi[q] && !(dbg & 32)) ? rptr[q] + j0 : zp;
            dma16(src, A + (q * NP + pw) * 1024);
          } else {
            dma16(wrow[q - NA] + wseg + k0, B + ((q - NA) * NP + pw) * 1024);
          }
        }
      }
      if (ph == NPH - 1) {
        istage = istage + 1 == S ? 0 : istage + 1;
        ++issued;
        k0 += BK;
        if (k0 >= seglen && issued < ntiles) { k0 = 0; ++seg; enter_run(seg); }
      }
    };
    enter_run(0);
    for (int i = 0; i < S - 1; ++i)
      if (issued < ntiles) {
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) issue_part(ph);
      }
    wait_tiles<NL, S>(min(issued - 1, S - 2));
    wg_barrier();
    for (int i = 0; i < ntiles; ++i) {
      const bool more = issued < ntiles;
#pragma unroll
      for (int ph = 0; ph < NPH; ++ph) {
        if (more) issue_part(ph);
        if (ph == NPH - 1 && i + 1 < ntiles) wait_tiles<NL, S>(min(issued - (i + 2), S - 2));   // tile i+1 has landed (this thread's part)
        wg_barrier();
      }
    }
    wg_barrier();
  } else {
    // =============================================================================== consumers
    const int grp = wid >> 2;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int fsw = (frow >> 2) & 3;
    int choff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) choff[s] = ((2 * s + fhalf) ^ fsw) * 16;
    const int aoff = (wm0 + frow) * RB, boff = A_BYTES + (wn0 + frow) * RB;
    uint4 af[MI], bf[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) af[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NI; ++j) bf[j] = make_uint4(0, 0, 0, 0);
    wg_barrier();                                            // tile 0 has landed
    if (grp == 1) wg_barrier();                              // group B runs one phase (half a k16 step) behind group A
    int cstage = 0;
    for (int i = 0; i < ntiles; ++i) {
      const char* base = smem + cstage * STAGE;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        // read phase: this step's 6 fragments (the other group multiplies meanwhile)
        if (!(dbg & 4)) {
#pragma unroll
          for (int ii = 0; ii < MI; ++ii) af[ii] = *reinterpret_cast<const uint4*>(base + aoff + ii * (32 * RB) + choff[s]);
#pragma unroll
          for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const uint4*>(base + boff + j * (32 * RB) + choff[s]);
        }
        lds_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // multiply phase
        if (!(dbg & 1)) {
#pragma unroll
          for (int ii = 0; ii < MI; ++ii)
#pragma unroll
            for (int j = 0; j < NI; ++j)
              acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[ii]), __builtin_bit_cast(bf16x8, bf[j]),
                                                                   acc[ii][j], 0, 0, 0);
        }
        wg_barrier();
      }
      cstage = cstage + 1 == S ? 0 : cstage + 1;
    }
    if (grp == 0) wg_barrier();
  }
  if (dbg & 8) return;

  // ---- epilogue: consumers hold the accumulators; all 12 waves take part in the barriers and in the staged store pass
  int64_t* rowoff = reinterpret_cast<int64_t*>(smem);              // [BM]
  float* stat = reinterpret_cast<float*>(smem + BM * 8);           // [2][BN][2]
  uint16_t* otile = reinterpret_cast<uint16_t*>(smem + BM * 8 + 2 * BN * 8);   // [BM][OS]
  const bool staged = (d.flags & kRunYAligned) && !(d.flags & kRunAccum);
  if (tid < BM) {
    const int m = mtile * BM + tid;
    int64_t o = -1;
    if (m < d.M) {
      const int b = fdiv2(m, d.div_tf_m, d.div_tf_s), rem = m - b * TF, u = fdiv2(rem, d.div_fo_m, d.div_fo_s), fo = rem - u * d.Fo;
      o = (int64_t)b * d.y_bstride + (int64_t)u * d.y_tstride + (int64_t)fo * d.y_fstride + d.y_off;
    }
    rowoff[tid] = o;
  }
  __syncthreads();
  const float* bias = d.bias.arena >= 0 ? reinterpret_cast<const float*>(rp(ab, d.bias)) : nullptr;
  char* yb = rp(ab, d.y);
  const bool want_stats = d.stats.arena >= 0;
  if (wid < NC) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int nl = wn0 + j * 32 + (lane & 31);
      const int n = ntile * BN + nl;
      const float bv = (bias && n < d.N) ? bias[n] : 0.f;
      float s1 = 0.f, s2 = 0.f;
      if (staged) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            float v = acc[i][j][e] + bv;
            if (d.flags & kRunRelu) v = fmaxf(v, 0.f);
            otile[row * OS + nl] = f2bf(v);
            if (mtile * BM + row < d.M && n < d.N) { s1 += v; s2 += v * v; }
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            const int64_t o = rowoff[row];
            float v = acc[i][j][e] + bv;
            if (o >= 0 && n < d.N) {
              if (d.flags & kRunAccum) v += reinterpret_cast<float*>(yb)[o + n];
              if (d.flags & kRunRelu) v = fmaxf(v, 0.f);
              if (d.ydt == DT_BF16) reinterpret_cast<uint16_t*>(yb)[o + n] = f2bf(v);
              else reinterpret_cast<float*>(yb)[o + n] = v;
              s1 += v;
              s2 += v * v;
            }
          }
        }
      }
      if (want_stats) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (lane < 32) {
          stat[((wid / WN_) * BN + nl) * 2 + 0] = s1;
          stat[((wid / WN_) * BN + nl) * 2 + 1] = s2;
        }
      }
    }
  }
  __syncthreads();
  if (staged) {
    constexpr int CPRO = BN / 8;
    for (int q = tid; q < BM * CPRO; q += (NC + NP) * 64) {
      const int row = q / CPRO, cc = q - row * CPRO;
      const int64_t o = rowoff[row];
      const int n0 = ntile * BN + cc * 8;
      if (o >= 0 && n0 < d.N)
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(yb) + o + n0) = *reinterpret_cast<const uint4*>(otile + row * OS + cc * 8);
    }
  }
  if (want_stats && tid < BN) {
    float* part = reinterpret_cast<float*>(rp(ab, d.stats));
    const int nrows = (d.M + kBM - 1) / kBM;
#pragma unroll
    for (int h = 0; h < 2; ++h) {                                  // wave row h of the consumer grid = 128-row half h
      const int srow = mtile * 2 + h;
      if (srow < nrows) {
        part[((int64_t)srow * 2 + 0) * d.Npad + ntile * BN + tid] = stat[(h * BN + tid) * 2 + 0];
        part[((int64_t)srow * 2 + 1) * d.Npad + ntile * BN + tid] = stat[(h * BN + tid) * 2 + 1];
      }
    }
  }
}

// SEFD_CG256=0 disables the wide kernel (A/B runs); bit 0: N % 256 == 0 layers, bit 1: N % 128 == 0 layers.
// SEFD_CG256_MINM: smallest M it is used for.  SEFD_CG256_VAR / SEFD_CG256_DBG: tuning variants (see the kernel).
bool launch_cgemm256(const RunGemm& d, const ArenaBases& ab, hipStream_t st) {
  static const int enabled = getenv("SEFD_CG256") ? atoi(getenv("SEFD_CG256")) : 1;
  static const int minm = getenv("SEFD_CG256_MINM") ? atoi(getenv("SEFD_CG256_MINM")) : 4096;
  static const int var = getenv("SEFD_CG256_VAR") ? atoi(getenv("SEFD_CG256_VAR")) : 0;
  static const int dbg = getenv("SEFD_CG256_DBG") ? atoi(getenv("SEFD_CG256_DBG")) : 0;
  if (!enabled || d.xdt != DT_BF16 || !(d.flags & kRunAligned) || d.M < minm) return false;
  const int nm = (d.M + 255) / 256;
  if (d.Npad % 256 == 0 && (enabled & 1)) {
    const dim3 grid(nm * (d.Npad / 256));
    if (var & 4) hipLaunchKernelGGL((cgemm_ws_kernel<4>), grid, dim3(768), 0, st, d, ab, dbg);
    else if (var & 1) hipLaunchKernelGGL((cgemm256_kernel<256, 32, 4, false>), grid, dim3(512), 0, st, d, ab, dbg);
    else hipLaunchKernelGGL((cgemm256_kernel<256, 32, 4, true>), grid, dim3(512), 0, st, d, ab, dbg);
    return true;
  }
  if (d.Npad % 128 == 0 && (enabled & 2)) {
    const dim3 grid(nm * (d.Npad / 128));
    if (var & 2) hipLaunchKernelGGL((cgemm256_kernel<128, 32, 4, true>), grid, dim3(512), 0, st, d, ab, dbg);
    else hipLaunchKernelGGL((cgemm256_kernel<128, 64, 3, true>), grid, dim3(512), 0, st, d, ab, dbg);
    return true;
  }
  return false;
}

}  // namespace sefd
