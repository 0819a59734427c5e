// RUNGEMM, wide-tile variant for the layers that carry the FLOPs (bf16, N a multiple of 256, LDS-DMA-able runs).
//
// What bounds the 128 x 128 kernel of rungemm.hip (measured, profiles/r02_tuning_notes.md): it moves 32 KB of operands through
// L2 -> LDS per 2.1 MFLOP and the LDS-DMA stream it sustains next to the MFMAs (~24 B/clk/CU) caps it near 1 PFLOP/s; every
// K tile ends in a barrier behind which DMA issue, fragment reads and MFMAs run one after the other.  This kernel:
//   * 256 x 256 output tile per 512-thread workgroup: 128 FLOP per operand byte, the activation operand is streamed once per M tile;
//   * 8 waves in a 2 x 4 grid, 128 x 64 accumulators each, two per SIMD.  ONE workgroup barrier per 64-deep K tile; inside it a wave runs its
//     four k16 steps (6 fragment reads, 8 MFMAs, two LDS-DMAs behind the fourth MFMA) at its own pace and the two waves of a SIMD
//     interleave on their own.  (Rounds 2-3 ran the waves as two groups one phase apart with two barriers per k16 step: the slowest wave -
//     the one held longest at a DMA issue - then paced all eight twice per step.  Round 4, same box: 10.55 -> 10.44 ms per step, 19.0 -> 18.8
//     at B = 64, DCCRN-large 54.2 -> 54.1, FullSubNet 66.8 -> 66.7; two fragment register sets inside the loop measured no better.)
//   * every DMA instruction moves whole 128-byte lines: the A operand is staged in 64-deep K tiles (8 rows x 128 B per
//     instruction), the weights are packed K-tile major (kRunWTile32) so that a 32-deep B tile is one contiguous 16 KB block.
//     With 64-byte row pieces (first version) every line was fetched twice from L2 and the DMA stream alone took 1250-1480
//     cycles per 32-deep K tile against 1024 cycles of MFMAs; with whole lines ~800;
//   * LDS ring: 3 A slots of 32 KB + 4 B sub-slots of 16 KB = exactly 160 KB.  While K tile p is multiplied, B(p+1) is issued in its steps 0-1
//     (into the sub-slots of K tile p-1) and A(p+2) in steps 2-3 (into the slot of K tile p-1): at the top of K tile p+1 a thread waits with
//     `s_waitcnt vmcnt(4)` - only its four A(p+2) instructions may still be in flight - and the barrier behind that wait both publishes
//     everyone's part of A(p+1), B(p+1) and frees the slots of K tile p.  The queue is never drained inside the loop;
//   * persistent: workgroup b works on output tiles b, b + gridDim, ...; the ring positions carry over, so the first K tiles of
//     the next output tile are already in flight during the epilogue, which touches neither LDS nor a barrier (quad transpose in
//     registers, 8-byte row-piece stores; dev_common.h QuadT).
// Same descriptor (sefd_desc.h RunGemm) and epilogue contract (bias, ReLU, accumulate, BatchNorm partial sums per 128 rows) as
// rungemm_kernel.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "sefd_desc.h"
#include "tuning.h"
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

}  // namespace

// dbg (tuning BUILDS only, -DSEFD_TUNING + SEFD_CG256_DBG; the product library instantiates dbg = 0 only): 1 skip the MFMAs, 2 skip the DMAs, 4 skip the fragment reads, 8 skip the epilogue,
// 32 every A chunk from the zero page (no activation traffic), 128 / 256 the step's DMAs at a wave-dependent position between the MFMAs (2 / 4 positions)
// BNB: the kRunBnBwd epilogue as its own instantiation - the kernel sits at the 256-register cap of two waves per SIMD, and with the
// extra epilogue state in the one body the allocator spilled inside the K loop of EVERY launch (15x slower)
// (dbg is a template argument; only the switch values the tuning runs use are instantiated)
template <bool BNB, int dbg = 0>
__global__ __launch_bounds__(512) void cgemm256_kernel(const RunGemm d, const ArenaBases ab) {
  constexpr int BM = 256, BN = 256, NW = 8;
  constexpr int KT = 64;                                     // K tile of the A operand and of the loop (two 32-deep B tiles)
  constexpr int A_SLOT = BM * 128, B_SLOT = BN * 64;        // 32 KB, 16 KB
  constexpr int NAS = 3, NBS = 4;
  constexpr int B_BASE = NAS * A_SLOT;
  constexpr int NA = BM / 8 / NW;                            // A DMAs per thread per K tile: 4 (8 rows x 128 B each)
  constexpr int NBH = BN / 16 / NW;                          // B DMAs per thread per 32-deep half tile: 2 (16 rows x 64 B each)
  constexpr int WN_ = 4, MI = 4, NI = 2, KS = KT / 16;
  constexpr int SMEM = NAS * A_SLOT + NBS * B_SLOT;
  static_assert(SMEM <= 160 * 1024 && NW * 4096 <= A_SLOT && KS == 4, "LDS budget / schedule");
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
  // DMA roles.  A: instruction q covers rows (q*8 + wid)*8 .. +8 ; lane -> (row la, position pa); swizzle on the source chunk.
  const int la = lane >> 3, pa = lane & 7;
  const int csa = pa ^ ((((wid & 1) * 4) + (la >> 1)) & 7);
  // B: instruction q covers rows (q*8 + wid)*16 .. +16 of a 32-deep tile ; lane -> (row lb, position pb)
  const int lb = lane >> 2, pb = lane & 3;
  const int csb = pb ^ ((lb >> 2) & 3);
  const int64_t wtile = (int64_t)d.Npad * 32;                // elements per 32-deep weight tile (K-tile major layout)
  const int Tin0 = d.Tin[0], Tin1 = d.Tin[1], fs0 = d.fstride[0], fs1 = d.fstride[1], rl0 = d.rowlen[0], rl1 = d.rowlen[1];
  const int64_t ts0 = d.tstride[0], ts1 = d.tstride[1];
  const uint32_t lds0 = lds_addr(smem);
  // fragment roles: wave tile (4 x 32) x (2 x 32) at (wm0, wn0)
  const int wm0 = (wid / WN_) * (MI * 32), wn0 = (wid % WN_) * (NI * 32);
  const int frow = lane & 31, fhalf = lane >> 5;
  // chunk of k16 step s: ((2s + half) ^ swizzle) * 16 = (c0 * 16) ^ (32 s)
  const int ca0 = (fhalf ^ ((frow >> 1) & 7)) * 16, cb0 = (fhalf ^ ((frow >> 2) & 3)) * 16;
  const int aoff = (wm0 + frow) * 128, boff = B_BASE + (wn0 + frow) * 64;
  const float* biasp = d.bias.arena >= 0 ? reinterpret_cast<const float*>(rp(ab, d.bias)) : nullptr;
  char* yb = rp(ab, d.y);
  const bool want_stats = d.stats.arena >= 0;
  const bool staged = (d.flags & kRunYAligned) && !(d.flags & kRunAccum);
  // the output rows are one dense [M][row pitch] array (conv outputs: frames and batch items follow each other without gaps): row offset = m * y_fstride + y_off,
  // no (b, u, fo) decode - the epilogue is VALU-bound (~1400 instructions per lane and tile, 20-25 % of a launch: profiles/r05_tuning_notes.md section 7)
  const bool ylin = d.y_tstride == d.Fo * d.y_fstride && d.y_bstride == (int64_t)d.Tout * d.y_tstride;
  const uint16_t* ybn = BNB ? reinterpret_cast<const uint16_t*>(rp(ab, d.bnb_y)) : nullptr;   // the BatchNorm layer's forward output (bf16)
  const float bslope = BNB ? *reinterpret_cast<const float*>(rp(ab, d.bnb_slope)) : 0.f;

  int gk = 0;                                                // K tiles finished by this workgroup so far (ring positions carry over)
  // ---- DMA state of the output tile being loaded (set up, and its first K tiles issued, before the previous tile's epilogue)
  int d_ntile = 0, d_mtile = 0;
  int64_t rb0[NA], rb1[NA];
  int ru[NA], rfo[NA];
  bool rv[NA];
  const uint16_t* wb0 = w;
  const uint16_t* rptr[NA];
  int jlo[NA], jhi[NA];
  int aseg = 0, ak0 = 0, aseglen = 0, a_local = 0;           // A cursor: run, position inside the run, run length, K tiles issued
  int a_slot = 0, c_slot = 0;                                // ring slots (gk + a_local) % NAS of the issue cursor / (gk + p) % NAS of the K loop, kept incrementally
  int bseg = 0, bk0 = 0, bseglen = 0, bkoff = 0;             // B cursor (advances per 64-deep K tile)
  int kt32[3];                                               // 32-deep weight tile index of local K tiles p, p+1, p+2
  auto enter_run = [&](int sgi) {
    const Seg sg = d.seg[sgi];
    aseglen = sg.len;
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
  // half `hf` (0, 1) of the next A tile: 2 of this thread's 4 instructions; the cursor advances with the second half
  auto issue_a = [&](int hf) {
    if (!(dbg & 2)) {
      const uint32_t A = lds0 + a_slot * A_SLOT;
      const int j0 = ak0 + csa * 8;
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        if (q / 2 != hf) continue;
        if ((dbg & 64) && q != 0) continue;                    // 64: one of the four A instructions only (traffic of an LDS-resident input slab)
        const uint16_t* src = (j0 >= jlo[q] && j0 + 8 <= jhi[q] && !(dbg & 32)) ? rptr[q] + j0 : zp;
        dma16(src, A + (q * NW + wid) * 1024);
      }
    }
    if (hf == 1) {
      ++a_local;
      a_slot = a_slot + 1 == NAS ? 0 : a_slot + 1;
      ak0 += KT;
      if (ak0 >= aseglen && a_local < nkt) { ak0 = 0; ++aseg; enter_run(aseg); }
    }
  };
  // this thread's 2 instructions of half h of the B tile of K tile kt (local index), whose first 32-deep tile index is koff32
  auto issue_b = [&](int kt, int h, int koff32) {
    if (!(dbg & 2)) {
      const uint32_t B = lds0 + B_BASE + (((gk + kt) & 1) * 2 + h) * B_SLOT;
      const uint16_t* src = wb0 + ((int64_t)koff32 + h) * wtile;
#pragma unroll
      for (int q = 0; q < NBH; ++q) dma16(src + q * (NW * 16 * 32), B + (q * NW + wid) * 1024);
    }
  };
  auto b_tile32 = [&]() { return (bkoff + bk0) >> 5; };
  auto b_advance = [&]() { bk0 += KT; if (bk0 >= bseglen && bseg + 1 < d.nseg) { bk0 = 0; ++bseg; bseglen = d.seg[bseg].len; bkoff = d.seg[bseg].koff; } };
  // set up output tile t and put its first K tiles in flight: A(0), Bh0(0), Bh1(0) [, A(1)].  Called with gk = the ring
  // position the tile starts at: its slots (gk, gk + 1; both B pairs) are free as soon as the previous tile's K loop is over.
  auto begin_tile = [&](int t) {
    const int tile = xcd_remap2(t, total);
    d_ntile = tile % nn; d_mtile = tile / nn;
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int m = d_mtile * BM + (q * NW + wid) * 8 + la;
      rv[q] = m < d.M;
      const int mm = rv[q] ? m : 0;
      const int b = fdiv2(mm, d.div_tf_m, d.div_tf_s), rem = mm - b * TF;
      ru[q] = fdiv2(rem, d.div_fo_m, d.div_fo_s);
      rfo[q] = rem - ru[q] * d.Fo;
      rb0[q] = (int64_t)b * d.bstride[0] + d.base[0];
      rb1[q] = (int64_t)b * d.bstride[1] + d.base[1];
    }
    wb0 = w + ((int64_t)d_ntile * BN + wid * 16 + lb) * 32 + csb * 8;   // instruction q adds q * NW * 16 rows
    aseg = 0; ak0 = 0; a_local = 0;
    a_slot = gk % NAS;
    bseg = 0; bk0 = 0; bseglen = d.seg[0].len; bkoff = d.seg[0].koff;
    enter_run(0);
    issue_a(0); issue_a(1);
    kt32[0] = b_tile32(); b_advance();
    kt32[1] = b_tile32(); b_advance();
    kt32[2] = b_tile32(); b_advance();
    issue_b(0, 0, kt32[0]);
    issue_b(0, 1, kt32[0]);
    if (nkt > 1) { issue_a(0); issue_a(1); }
  };
  if ((int)blockIdx.x < total) begin_tile(blockIdx.x);
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    const int ntile = d_ntile, mtile = d_mtile;
    float bv[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = ntile * BN + wn0 + j * 32 + (lane & 31);
      bv[j] = (biasp && n < d.N) ? biasp[n] : 0.f;
    }
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;      // (the bias is NOT summed in here: bias + sum rounds differently from sum + bias, and the kernels of
                                                              // this library produce bit-identical outputs for the same GEMM - eval-mode batch independence test)
    uint4 af[MI], bf[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) af[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NI; ++j) bf[j] = make_uint4(0, 0, 0, 0);
    c_slot = gk % NAS;
    for (int p = 0; p < nkt; ++p) {
      const bool has1 = p + 1 < nkt, has2 = p + 2 < nkt;
      const char* abase = smem + c_slot * A_SLOT + aoff;
      c_slot = c_slot + 1 == NAS ? 0 : c_slot + 1;
      const char* bbase = smem + ((gk + p) & 1) * 2 * B_SLOT + boff;
      if (has1 && !(dbg & 2)) wait_vm<(dbg & 64) ? 1 : NA>(); else wait_vm<0>();   // A(p), B(p) have landed (this thread's part); only A(p+1) may be in flight
      wg_barrier();                                        // ... everyone's part; and every wave has finished K tile p - 1: its slots may be refilled
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (!(dbg & 4)) {
#pragma unroll
          for (int ii = 0; ii < MI; ++ii) af[ii] = *reinterpret_cast<const uint4*>(abase + ii * (32 * 128) + (ca0 ^ (32 * s)));
#pragma unroll
          for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const uint4*>(bbase + (s >> 1) * B_SLOT + j * (32 * 64) + (cb0 ^ (32 * (s & 1))));
        }
        if (!(dbg & 1)) {
#pragma unroll
          for (int ii = 0; ii < MI; ++ii) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
              acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[ii]), __builtin_bit_cast(bf16x8, bf[j]),
                                                                   acc[ii][j], 0, 0, 0);
            if (ii == ((dbg & 128) ? ((wid & 1) ? 0 : 2) : ((dbg & 256) ? (wid & 3) : 1))) {   // this step's two DMAs behind the first four MFMAs: the wave is held at their issue while its
              __builtin_amdgcn_sched_barrier(0);           // own MFMAs drain (in front of / behind all eight: +0.04 ms per step, same box)
              if (s == 0) { if (has1) issue_b(p + 1, 0, kt32[1]); }
              else if (s == 1) { if (has1) issue_b(p + 1, 1, kt32[1]); }
              else if (s == 2) { if (has2) issue_a(0); }
              else { if (has2) issue_a(1); }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        } else {
          if (s == 0) { if (has1) issue_b(p + 1, 0, kt32[1]); }
          else if (s == 1) { if (has1) issue_b(p + 1, 1, kt32[1]); }
          else if (s == 2) { if (has2) issue_a(0); }
          else { if (has2) issue_a(1); }
        }
      }
      kt32[0] = kt32[1]; kt32[1] = kt32[2]; kt32[2] = b_tile32(); b_advance();
    }
    if (BNB) wg_barrier();                             // the kRunBnBwd epilogue writes LDS: every wave must be past the last K tile
    gk += nkt;
    // the next tile's first K tiles are in flight during this epilogue - except in the BNB instantiation, where the epilogue needs the
    // registers of that DMA state (it starts the next tile after the epilogue instead)
    if (!BNB && t + (int)gridDim.x < total) begin_tile(t + gridDim.x);
    if constexpr (BNB) {
      // ---- kRunBnBwd epilogue (bf16, 16-byte aligned rows - what the planner gives this kernel).  Every wave is past the last K tile and
      // nothing is in flight: the whole LDS is free.  Phase 1: the wave's 128 x 64 tile goes to 16 KB of LDS as bf16 (the accumulators
      // die here).  Phase 2: lane (chunk column ch = lane & 7, row lane >> 3 + 8k) moves 16-byte row chunks LDS -> global and, beside each,
      // reads the same chunk of the BatchNorm layer's forward output (coalesced) and accumulates the three backward sums of its 8 columns.
      if (!(dbg & 8)) {
        char* wt = smem + wid * 16384;
        const QuadT qt(lane);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const int col = j * 32 + (lane & 28);                  // quad transpose: this lane writes 4 consecutive columns of ONE row per group
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int row = i * 32 + 8 * q + 4 * (lane >> 5) + (lane & 3);
              const uint2 pk = qt.pack(acc[i][j][4 * q] + bv[j], acc[i][j][4 * q + 1] + bv[j], acc[i][j][4 * q + 2] + bv[j], acc[i][j][4 * q + 3] + bv[j]);
              *reinterpret_cast<uint2*>(wt + row * 128 + (((col >> 3) ^ (row & 7)) << 4) + (col & 7) * 2) = pk;
            }
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int ch = lane & 7, n0 = ntile * BN + wn0 + ch * 8;
        const bool cok = n0 < d.N;
        float pm[8], pis[8], pg[8], pb[8], t0[8], t1[8], t2[8];
        {
          const float* mi = reinterpret_cast<const float*>(rp(ab, d.bnb_mi));
          const float* ga = reinterpret_cast<const float*>(rp(ab, d.bnb_gamma));
          const float* be = reinterpret_cast<const float*>(rp(ab, d.bnb_beta));
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int n = cok ? n0 + e : 0;
            pm[e] = mi[n]; pis[e] = mi[d.N + n]; pg[e] = ga[n]; pb[e] = be[n];
            t0[e] = t1[e] = t2[e] = 0.f;
          }
        }
        // two rounds of 8 rows: the 8 forward-output chunks of a round are all in flight before the first is used
        for (int k8 = 0; k8 < 16; k8 += 8) {
          uint4 ypre[8];
          int64_t oo[8];
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            const int m = mtile * BM + wm0 + (lane >> 3) + 8 * (k8 + kk);
            oo[kk] = -1;
            ypre[kk] = make_uint4(0, 0, 0, 0);
            if (m < d.M && cok) {
              const int b = fdiv2(m, d.div_tf_m, d.div_tf_s), rem = m - b * TF, u = fdiv2(rem, d.div_fo_m, d.div_fo_s), fo = rem - u * d.Fo;
              oo[kk] = (int64_t)b * d.y_bstride + (int64_t)u * d.y_tstride + (int64_t)fo * d.y_fstride + d.y_off;
              ypre[kk] = *reinterpret_cast<const uint4*>(ybn + (int64_t)b * d.bnb_bstride + (int64_t)u * d.bnb_tstride + (int64_t)fo * d.bnb_fstride + d.bnb_off + n0);
            }
          }
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
          const int row = (lane >> 3) + 8 * (k8 + kk);
          const int64_t o = oo[kk];
          if (o < 0) continue;
          const uint4 dzv = *reinterpret_cast<const uint4*>(wt + row * 128 + ((ch ^ (row & 7)) << 4));
          const uint4 yv = ypre[kk];
          *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(yb) + o + n0) = dzv;
          const uint32_t dw[4] = {dzv.x, dzv.y, dzv.z, dzv.w}, yw[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float dz = bf2f((uint16_t)(dw[e >> 1] >> (16 * (e & 1)))), yy = bf2f((uint16_t)(yw[e >> 1] >> (16 * (e & 1))));
            const float xh = (yy - pm[e]) * pis[e];
            const float bn = pg[e] * xh + pb[e];
            const float dbn = bn > 0.f ? dz : bslope * dz;
            t0[e] += dbn;
            t1[e] += dbn * xh;
            t2[e] += bn > 0.f ? 0.f : bn * dz;
          }
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
          for (int o = 32; o >= 8; o >>= 1) { t0[e] += __shfl_xor(t0[e], o); t1[e] += __shfl_xor(t1[e], o); t2[e] += __shfl_xor(t2[e], o); }
        }
        const int srow = mtile * 2 + wid / WN_;
        if (lane < 8 && srow < (d.M + kBM - 1) / kBM) {              // this wave's 128 rows are one 128-row block of partial sums
          float* part = reinterpret_cast<float*>(rp(ab, d.stats));
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            part[((int64_t)srow * 3 + 0) * d.Npad + n0 + e] = t0[e];
            part[((int64_t)srow * 3 + 1) * d.Npad + n0 + e] = t1[e];
            part[((int64_t)srow * 3 + 2) * d.Npad + n0 + e] = t2[e];
          }
        }
      }
      if (t + (int)gridDim.x < total) {
        wg_barrier();                                          // every wave has left its 16 KB before the next tile's DMAs land in LDS
        begin_tile(t + gridDim.x);
      }
      continue;
    }
    if (dbg & 8) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(acc[i][j][e]));      // every accumulator element stays live: nothing upstream is dead code
      continue;
    }
    // ---- epilogue, wave local, no LDS: bias / ReLU / statistics on the accumulators as they sit (one column per lane), then the quad transpose
    // of dev_common.h QuadT turns every group of 4 rows x 4 lanes into 8-byte row pieces that are stored directly (a wave instruction covers 8 rows x
    // 64 contiguous bytes).  The LDS-staged version (128 ds_write_b16 + 16 ds_read_b128 per lane) took 28-34 % of the kernel (SEFD_CG256_DBG=8).
    const QuadT qt(lane);
    const OctW ow(lane);
    float s1[NI], s2[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row0 = mtile * BM + wm0 + i * 32;
      if (staged) {
        const bool full = row0 + 32 <= d.M;                  // every row of this 32-row block exists (all tiles but the last)
        int64_t ro[4];                                       // output offsets of this lane's 4 rows after the transpose (-1: beyond M)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = row0 + 8 * q + 4 * (lane >> 5) + (lane & 3);
          ro[q] = -1;
          if (m < d.M) {
            if (ylin) ro[q] = (int64_t)m * d.y_fstride + d.y_off;
            else {
              const int b = fdiv2(m, d.div_tf_m, d.div_tf_s), rem = m - b * TF, u = fdiv2(rem, d.div_fo_m, d.div_fo_s), fo = rem - u * d.Fo;
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
          // 16-byte stores: the lane pair (l, l ^ 4) splits the four row groups between them (dev_common.h OctW); N % 8 == 0: 8 columns are all valid or all padding
          uint4 wide[2];
          ow.widen(pk, wide);
          const int n8 = n0 & ~7;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int64_t r = ow.hi4 ? ro[2 * h + 1] : ro[2 * h];
            if (r >= 0 && n8 < d.N) *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(yb) + r + n8) = wide[h];
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = row0 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          if (m >= d.M) continue;
          const int b = fdiv2(m, d.div_tf_m, d.div_tf_s), rem = m - b * TF, u = fdiv2(rem, d.div_fo_m, d.div_fo_s), fo = rem - u * d.Fo;
          const int64_t o = (int64_t)b * d.y_bstride + (int64_t)u * d.y_tstride + (int64_t)fo * d.y_fstride + d.y_off;
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const int n = ntile * BN + wn0 + j * 32 + (lane & 31);
            if (n >= d.N) continue;
            float v = acc[i][j][e] + bv[j];
            if (d.flags & kRunAccum) v += reinterpret_cast<float*>(yb)[o + n];
            if (d.flags & kRunRelu) v = fmaxf(v, 0.f);
            if (d.ydt == DT_BF16) reinterpret_cast<uint16_t*>(yb)[o + n] = f2bf(v);
            else reinterpret_cast<float*>(yb)[o + n] = v;
            s1[j] += v;
            s2[j] += v * v;
          }
        }
      }
    }
    if (want_stats) {                                        // this wave's 128 rows are ONE 128-row statistics block of its 64 columns
      float* part = reinterpret_cast<float*>(rp(ab, d.stats));
      const int srow = mtile * 2 + wid / WN_;
      const int nrows = (d.M + kBM - 1) / kBM;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const float t1 = s1[j] + __shfl_xor(s1[j], 32), t2 = s2[j] + __shfl_xor(s2[j], 32);
        const int n = ntile * BN + wn0 + j * 32 + (lane & 31);
        if (lane < 32 && srow < nrows) {
          part[((int64_t)srow * 2 + 0) * d.Npad + n] = t1;
          part[((int64_t)srow * 2 + 1) * d.Npad + n] = t2;
        }
      }
    }

  }
}

#ifdef SEFD_TUNING
int g_cgemm256_dbg = -1;       // tuning builds (-DSEFD_TUNING: tools/probes, A/B libraries) only: kernel variant; -1 = read SEFD_CG256_DBG once
#endif

// The planner decides which GEMMs take this kernel: it marks them (and packs their weights) with kRunWTile32.
bool launch_cgemm256(const RunGemm& d, const ArenaBases& ab, hipStream_t st) {
  if (!(d.flags & kRunWTile32)) return false;
  static const int ncu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
  const int total = ((d.M + 255) / 256) * (d.Npad / 256);
  const dim3 grid(total < ncu ? total : ncu);
  const bool bnb = (d.flags & kRunBnBwd) != 0;
#define SEFD_CG256_LAUNCH(DBG)                                                                                              \
  do {                                                                                                                      \
    if (bnb) hipLaunchKernelGGL((cgemm256_kernel<true, DBG>), grid, dim3(512), 0, st, d, ab);                               \
    else hipLaunchKernelGGL((cgemm256_kernel<false, DBG>), grid, dim3(512), 0, st, d, ab);                                  \
  } while (0)
#ifdef SEFD_TUNING
  // wrong-result / experimental arms exist in tuning builds only: the product library has no switch that changes what a launch computes
  if (g_cgemm256_dbg < 0) g_cgemm256_dbg = tune_str("CG256_DBG") ? atoi(tune_str("CG256_DBG")) : 0;
  switch (g_cgemm256_dbg) {
    case 1: SEFD_CG256_LAUNCH(1); return true;
    case 2: SEFD_CG256_LAUNCH(2); return true;
    case 4: SEFD_CG256_LAUNCH(4); return true;
    case 8: SEFD_CG256_LAUNCH(8); return true;
    case 32: SEFD_CG256_LAUNCH(32); return true;
    case 64: SEFD_CG256_LAUNCH(64); return true;
    case 128: SEFD_CG256_LAUNCH(128); return true;
    case 256: SEFD_CG256_LAUNCH(256); return true;
    default: break;
  }
#endif
  SEFD_CG256_LAUNCH(0);
#undef SEFD_CG256_LAUNCH
  return true;
}

}  // namespace sefd
