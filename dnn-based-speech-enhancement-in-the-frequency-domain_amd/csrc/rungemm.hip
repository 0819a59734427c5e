// RUNGEMM / WGRAD: the two MFMA kernels that carry >96 % of the DCCRN train-step FLOPs.
//
// gfx950 only.  64-wide wavefronts, 4 waves per workgroup, 32x32 MFMA tiles:
//   fp32 path  : v_mfma_f32_32x32x2_f32   (exact fp32, used for the parity mode and the fp32-only front end)
//   bf16 path  : v_mfma_f32_32x32x16_bf16 (fp32 accumulate)
// Both dtypes use 128-byte K-rows in LDS (32 fp32 / 64 bf16), written as 16-byte chunks with an XOR swizzle on
// the chunk index so that the ds_read_b128 fragment reads of 16 different rows hit 16 different 16-byte slots.
// The A operand is never materialised (no im2col): every A row is a handful of contiguous runs of the
// channels-last activation tensor (see sefd_desc.h), gathered straight from HBM/L2 with 16-byte loads.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "sefd_desc.h"
#include "tuning.h"
#include "dev_common.h"

namespace sefd {

template <typename TA>
struct RowInfo {
  int64_t base0, base1;   // element offset of (b, u=0-relative) row start in x[0] / x[1] (without time)
  int32_t u, fo;
  bool valid;
};

// One 16-byte chunk (VEC elements) of run `sg` for row `ri`, starting at run position j0.
template <typename TA>
__device__ __forceinline__ uint4 load_a_chunk(const RunGemm& d, const TA* x0, const TA* x1, const Seg& sg,
                                              int64_t base0, int64_t base1, int u, int fo, bool rvalid, int j0) {
  constexpr int VEC = 16 / sizeof(TA);
  uint4 z = make_uint4(0, 0, 0, 0);
  if (!rvalid) return z;
  if (sg.src < 0) {                       // "ones" run
    if (j0 == 0) {
      if (sizeof(TA) == 4) z.x = 0x3f800000u; else z.x = 0x3f80u;
    }
    return z;
  }
  const int s = sg.src;
  const int tt = u + sg.dt;
  if (tt < 0 || tt >= d.Tin[s]) return z;
  const int r = sg.off + fo * d.fstride[s] + j0;
  int lo = r < 0 ? -r : 0;
  int hi = VEC;
  if (sg.len - j0 < hi) hi = sg.len - j0;
  if (d.rowlen[s] - r < hi) hi = d.rowlen[s] - r;
  if (hi <= lo) return z;
  const TA* src = (s ? x1 : x0) + (s ? base1 : base0) + (int64_t)tt * d.tstride[s] + r;
  if (lo == 0 && hi == VEC && (reinterpret_cast<uintptr_t>(src) & 15) == 0) return *reinterpret_cast<const uint4*>(src);
  // partially valid or unaligned chunk: element-wise
  if (sizeof(TA) == 4) {
    uint32_t v[4] = {0, 0, 0, 0};
    const uint32_t* p = reinterpret_cast<const uint32_t*>(src);
    for (int e = lo; e < hi; ++e) v[e] = p[e];
    return make_uint4(v[0], v[1], v[2], v[3]);
  } else {
    uint16_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint16_t* p = reinterpret_cast<const uint16_t*>(src);
    for (int e = lo; e < hi; ++e) v[e] = p[e];
    return make_uint4(v[0] | (uint32_t)v[1] << 16, v[2] | (uint32_t)v[3] << 16, v[4] | (uint32_t)v[5] << 16,
                      v[6] | (uint32_t)v[7] << 16);
  }
}

// x / d for 0 <= x < 2^31 with the planner's (m, s) of sefd_desc.h fastdiv_make
__device__ __forceinline__ int fdiv(int x, uint32_t m, uint32_t s) { return m ? (int)(__umulhi((uint32_t)x, m) >> s) : x; }

__device__ __forceinline__ int swz_off(int row, int chunk) {   // byte offset of a 16-byte chunk inside a [rows][128 B] tile
  return (row * 8 + (chunk ^ ((row >> 1) & 7))) * 16;
}

template <typename TA>
__device__ __forceinline__ void mfma_step(f32x16& acc, const uint4& a, const uint4& b) {
  if constexpr (sizeof(TA) == 4) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
}

// XCD-aware bijective remap of the linear workgroup id (8 XCDs, private L2s): each XCD gets a contiguous
// range of tiles so that neighbouring M-tiles (which share input rows) and all N-tiles of one M-tile share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, local = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + local;
}

// ------------------------------------------------------------------------------------------------------------
// GLDS = true: both operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass,
// which otherwise costs more LDS cycles than the fragment reads); padding chunks are fetched from a zero page, and the
// XOR swizzle is applied on the SOURCE chunk index because the DMA destination is lane-linear (wave base + lane*16).
// BNB: the kRunBnBwd epilogue as its own instantiation (its state costs ~25 VGPRs, which the plain kernel needs for occupancy)
template <typename TA, int BN, bool GLDS, int S = 2, int BM = kBM, int NW = 4, bool ILV = false, bool BNB = false>
__global__ __launch_bounds__(NW * 64) void rungemm_kernel(const RunGemm d, const ArenaBases ab) {
  constexpr int VEC = 16 / sizeof(TA);
  constexpr int BK = 8 * VEC;
  constexpr int PR = NW * 8;                   // rows per load pass (8 threads per 128-byte row)
  constexpr int RP = BM / PR;                  // A-row passes per thread (rows r0 + PR*p)
  constexpr int WN_ = BN == 128 ? 2 : 1;      // waves along N
  constexpr int WM_ = NW / WN_;                // waves along M
  constexpr int MI = BM / (32 * WM_);
  constexpr int NI = BN / (32 * WN_);
  constexpr int BPASS = BN / PR;
  constexpr int TILE_BYTES = (BM + BN) * 128;

  __shared__ __attribute__((aligned(16))) char smem[S * TILE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int nn = d.Npad / BN;
  const int nm = (d.M + BM - 1) / BM;
  const int swz = xcd_remap(blockIdx.x, nm * nn);
  const int ntile = swz % nn, mtile = swz / nn;

  const TA* x0 = reinterpret_cast<const TA*>(rp(ab, d.x[0]));
  const TA* x1 = d.x[1].arena >= 0 ? reinterpret_cast<const TA*>(rp(ab, d.x[1])) : x0;
  const TA* w = reinterpret_cast<const TA*>(rp(ab, d.w));

  // ---- per-thread load assignment: chunk column c, rows r0 + 32p
  const int c = tid & 7, r0 = tid >> 3;
  int64_t rb0[RP], rb1[RP];
  int ru[RP], rfo[RP];
  bool rv[RP];
  const int TF = d.Tout * d.Fo;
#pragma unroll
  for (int p = 0; p < RP; ++p) {
    const int m = mtile * BM + r0 + PR * p;
    rv[p] = m < d.M;
    const int mm = rv[p] ? m : 0;
    const int b = fdiv(mm, d.div_tf_m, d.div_tf_s), rem = mm - b * TF;
    ru[p] = fdiv(rem, d.div_fo_m, d.div_fo_s);
    rfo[p] = rem - ru[p] * d.Fo;
    rb0[p] = (int64_t)b * d.bstride[0] + d.base[0];
    rb1[p] = (int64_t)b * d.bstride[1] + d.base[1];
  }
  const int csrc = GLDS ? (c ^ ((r0 >> 1) & 7)) : c;       // chunk of the K-tile this thread fetches
  const TA* wrow[BPASS];
#pragma unroll
  for (int p = 0; p < BPASS; ++p) wrow[p] = w + (int64_t)(ntile * BN + r0 + PR * p) * d.ldw + csrc * VEC;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wm0 = (wid / WN_) * (MI * 32), wn0 = (wid % WN_) * (NI * 32);

  // ---- K-tile iteration state.  Everything that depends only on (row, run) - run start pointer, valid j-range,
  // alignment - is hoisted to run entry; a K-tile then costs two compares and one 16-byte load per chunk.
  int seg = 0, k0 = 0;
  int ntiles = 0;
  for (int s = 0; s < d.nseg; ++s) ntiles += (d.seg[s].len + BK - 1) / BK;

  const TA* rptr[RP];
  int jlo[RP], jhi[RP];
  int seglen = 0, wseg = 0;
  // per-source geometry in registers: indexing the by-value kernel argument with a run-time source id would turn into
  // global loads from the kernarg segment inside the K loop
  const int Tin0 = d.Tin[0], Tin1 = d.Tin[1], fs0 = d.fstride[0], fs1 = d.fstride[1], rl0 = d.rowlen[0], rl1 = d.rowlen[1];
  const int64_t ts0 = d.tstride[0], ts1 = d.tstride[1];
  auto enter_run = [&](int sgi) {
    const Seg sg = d.seg[sgi];
    seglen = sg.len;
    wseg = sg.koff;
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      int lo = 0, hi = 0;
      const TA* ptr = x0;
      if (sg.src >= 0 && rv[p]) {
        const int s = sg.src;
        const int tt = ru[p] + sg.dt;
        if (tt >= 0 && tt < (s ? Tin1 : Tin0)) {
          const int rr = sg.off + rfo[p] * (s ? fs1 : fs0);
          lo = rr < 0 ? -rr : 0;
          hi = min(sg.len, (s ? rl1 : rl0) - rr);
          ptr = (s ? x1 : x0) + (s ? rb1[p] : rb0[p]) + (int64_t)tt * (s ? ts1 : ts0) + rr;
          if ((reinterpret_cast<uintptr_t>(ptr) & 15) != 0) { lo |= 0x40000000; }   // unaligned run: element-wise path
        }
      }
      rptr[p] = ptr; jlo[p] = lo; jhi[p] = hi;
    }
  };
  uint4 aReg[RP], bReg[BPASS];
  auto issue_loads = [&](int kk) {
    const int j0 = kk + c * VEC;
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (j0 >= jlo[p] && j0 + VEC <= jhi[p]) {
        v = *reinterpret_cast<const uint4*>(rptr[p] + j0);
      } else {
        const int lo = jlo[p] & 0x3fffffff;
        if (j0 + VEC > lo && j0 < jhi[p]) {              // partially valid or unaligned chunk (never on the DCCRN shapes)
          const int e0 = lo > j0 ? lo - j0 : 0, e1 = jhi[p] - j0 < VEC ? jhi[p] - j0 : VEC;
          if constexpr (sizeof(TA) == 4) {
            uint32_t t[4] = {0, 0, 0, 0};
            const uint32_t* q = reinterpret_cast<const uint32_t*>(rptr[p] + j0);
            for (int e = e0; e < e1; ++e) t[e] = q[e];
            v = make_uint4(t[0], t[1], t[2], t[3]);
          } else {
            uint16_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const uint16_t* q = reinterpret_cast<const uint16_t*>(rptr[p] + j0);
            for (int e = e0; e < e1; ++e) t[e] = q[e];
            v = make_uint4(t[0] | (uint32_t)t[1] << 16, t[2] | (uint32_t)t[3] << 16, t[4] | (uint32_t)t[5] << 16, t[6] | (uint32_t)t[7] << 16);
          }
        }
      }
      aReg[p] = v;
    }
#pragma unroll
    for (int p = 0; p < BPASS; ++p) bReg[p] = *reinterpret_cast<const uint4*>(wrow[p] + wseg + kk);
  };
  // LDS offsets: the swizzle term ((row >> 1) & 7) is the same for all of a thread's rows (row strides are multiples of 16)
  int woff[RP > BPASS ? RP : BPASS];
#pragma unroll
  for (int p = 0; p < (RP > BPASS ? RP : BPASS); ++p) woff[p] = swz_off(r0 + PR * p, c);
  const int rsw = ((lane & 31) >> 1) & 7;
  int chs[4];
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) chs[kc] = ((2 * kc + (lane >> 5)) ^ rsw) * 16;
  const int arow = (wm0 + (lane & 31)) * 128, brow = (wn0 + (lane & 31)) * 128;

  auto compute = [&](const char* As, const char* Bs) {
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      uint4 af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const uint4*>(As + arow + i * 4096 + chs[kc]);
#pragma unroll
      for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const uint4*>(Bs + brow + j * 4096 + chs[kc]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) mfma_step<TA>(acc[i][j], af[i], bf[j]);
    }
  };

  enter_run(0);
  if constexpr (GLDS) {
    const TA* zp = reinterpret_cast<const TA*>(rp(ab, d.zero));
    const uint32_t lbase = lds_addr(smem) + __builtin_amdgcn_readfirstlane(wid) * 1024;   // this wave's 8 rows x 128 B of each pass
    constexpr int NL = RP + BPASS;                      // DMAs per thread per stage
    // part q of 4 of a stage's DMAs (A passes q, q+4, ..; B passes likewise): the interleaved loop issues one part
    // after each of the four MFMA groups of the tile being multiplied
    auto dma_part = [&](int stage, int kk, int q) {
      const uint32_t A = lbase + stage * TILE_BYTES, B = A + BM * 128;
      const int j0 = kk + csrc * VEC;
#pragma unroll
      for (int p = 0; p < RP; ++p) {
        if ((p & 3) != q) continue;
        const TA* src = (j0 >= jlo[p] && j0 + VEC <= jhi[p]) ? rptr[p] + j0 : zp;
        dma16(src, A + p * (PR * 128));
      }
#pragma unroll
      for (int p = 0; p < BPASS; ++p) {
        if ((p & 3) != q) continue;
        dma16(wrow[p] + wseg + kk, B + p * (PR * 128));
      }
    };
    auto dma = [&](int stage, int kk) {
#pragma unroll
      for (int q = 0; q < 4; ++q) dma_part(stage, kk, q);
    };
    // S-stage ring: the issue pointer runs S-1 K-tiles ahead of the MFMA loop; one LDS-only barrier per tile
    int issued = 0, istage = 0;
    auto issue_next = [&]() {
      dma(istage, k0);
      istage = istage + 1 == S ? 0 : istage + 1;
      ++issued;
      k0 += BK;
      if (k0 >= seglen && issued < ntiles) { k0 = 0; ++seg; enter_run(seg); }
    };
    for (int i = 0; i < S - 1; ++i)
      if (issued < ntiles) issue_next();
    int cstage = 0;
    for (int kt = 0; kt < ntiles; ++kt) {
      wait_stage<NL, S>(ntiles - 1 - kt);              // tile kt has landed (this thread's part)
      lds_barrier();                                   // ... everyone's part; and stage kt-1 is no longer being read
      const char* As = smem + cstage * TILE_BYTES;
      if constexpr (ILV) {
        // software pipeline inside the wave: fragments of group kc+1 are read while group kc multiplies, and the next
        // stage's DMAs (address math included) are spread over the four MFMA groups instead of preceding them
        const char* Bs = As + BM * 128;
        const bool more = issued < ntiles;
        uint4 af[2][MI], bf[2][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[0][i] = *reinterpret_cast<const uint4*>(As + arow + i * 4096 + chs[0]);
#pragma unroll
        for (int j = 0; j < NI; ++j) bf[0][j] = *reinterpret_cast<const uint4*>(Bs + brow + j * 4096 + chs[0]);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          if (kc < 3) {
#pragma unroll
            for (int i = 0; i < MI; ++i) af[(kc + 1) & 1][i] = *reinterpret_cast<const uint4*>(As + arow + i * 4096 + chs[kc + 1]);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[(kc + 1) & 1][j] = *reinterpret_cast<const uint4*>(Bs + brow + j * 4096 + chs[kc + 1]);
          }
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) mfma_step<TA>(acc[i][j], af[kc & 1][i], bf[kc & 1][j]);
          if (more) dma_part(istage, k0, kc);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (more) {
          istage = istage + 1 == S ? 0 : istage + 1;
          ++issued;
          k0 += BK;
          if (k0 >= seglen && issued < ntiles) { k0 = 0; ++seg; enter_run(seg); }
        }
      } else {
        if (issued < ntiles) issue_next();             // refills the stage read in kt-1
        compute(As, As + BM * 128);
      }
      cstage = cstage + 1 == S ? 0 : cstage + 1;
    }
  } else {
    issue_loads(0);
    for (int kt = 0; kt < ntiles; ++kt) {
      char* As = smem + (kt & 1) * TILE_BYTES;
      char* Bs = As + BM * 128;
#pragma unroll
      for (int p = 0; p < RP; ++p) *reinterpret_cast<uint4*>(As + woff[p]) = aReg[p];
#pragma unroll
      for (int p = 0; p < BPASS; ++p) *reinterpret_cast<uint4*>(Bs + woff[p]) = bReg[p];
      __syncthreads();
      // advance (run, k0) and prefetch the next tile into registers while this one is consumed from LDS
      k0 += BK;
      if (kt + 1 < ntiles) {
        if (k0 >= seglen) { k0 = 0; ++seg; enter_run(seg); }
        issue_loads(k0);
      }
      compute(As, Bs);
    }
  }
  __syncthreads();

  // ---- epilogue: row address table in LDS, bias, store, BatchNorm partial statistics
  int64_t* rowoff = reinterpret_cast<int64_t*>(smem);              // [BM]
  int64_t* rowoffb = rowoff + BM;                                  // [BM] row offsets into the BatchNorm layer's forward output (kRunBnBwd)
  float* stat = reinterpret_cast<float*>(smem + BM * 16);          // [NW][BN][3]
  uint16_t* otile = reinterpret_cast<uint16_t*>(smem + BM * 16 + NW * BN * 12);    // [BM][OS] bf16 staging tile (kRunYAligned)
  // row stride in elements: 16-byte aligned rows; in dwords = 16 (mod 32), so that the 4 rows x 64 bytes a half wave writes per ds_write_b64
  // of the quad-transposed tile (dev_common.h QuadT) fall into disjoint banks, and the 16-byte chunk reads of a row stay conflict-free
  constexpr int OS = BN % 64 == 0 ? BN + 32 : BN;
  constexpr bool kCanStage = BM * 16 + NW * BN * 12 + BM * OS * 2 <= S * TILE_BYTES;
  const bool staged = kCanStage && (d.flags & kRunYAligned) && !(d.flags & kRunAccum);
  constexpr bool bnb = BNB;
  // kRunBnBwd on the staged path: the sums are formed in the CHUNK loop below (a thread meets the same 8 columns on every row it stores:
  // the layer's forward output is read as coalesced 16-byte chunks beside the stores, not as 2-byte gathers in the accumulator layout)
  const bool bnb_chunk = bnb && staged && BM == kBM;
  if (tid < BM) {
    const int m = mtile * BM + tid;
    int64_t o = -1, ob = 0;
    if (m < d.M) {
      const int b = fdiv(m, d.div_tf_m, d.div_tf_s), rem = m - b * TF, u = fdiv(rem, d.div_fo_m, d.div_fo_s), fo = rem - u * d.Fo;
      o = (int64_t)b * d.y_bstride + (int64_t)u * d.y_tstride + (int64_t)fo * d.y_fstride + d.y_off;
      ob = (int64_t)b * d.bnb_bstride + (int64_t)u * d.bnb_tstride + (int64_t)fo * d.bnb_fstride + d.bnb_off;
    }
    rowoff[tid] = o;
    rowoffb[tid] = ob;
  }
  __syncthreads();
  const char* ybn = bnb ? rp(ab, d.bnb_y) : nullptr;
  const float bslope = bnb ? *reinterpret_cast<const float*>(rp(ab, d.bnb_slope)) : 0.f;
  // chunk loop geometry: thread -> chunk column cc (fixed), rows tid / CPR + k * (threads / CPR), k < KCH
  constexpr int CPR = BN / 8;                                      // 16-byte chunks per tile row
  constexpr int KCH = BM * CPR / (NW * 64);
  uint4 ypre[BNB ? KCH : 1];                                                 // the forward-output chunks of this thread's rows, in flight while the tile is staged
  if (bnb_chunk) {
    const int cc = tid % CPR, n0 = ntile * BN + cc * 8;
    const uint16_t* yf = reinterpret_cast<const uint16_t*>(ybn);
#pragma unroll
    for (int k = 0; k < KCH; ++k) {
      const int row = tid / CPR + k * (NW * 64 / CPR);
      ypre[k] = (rowoff[row] >= 0 && n0 < d.N) ? *reinterpret_cast<const uint4*>(yf + rowoffb[row] + n0) : make_uint4(0, 0, 0, 0);
    }
  }
  const float* bias = d.bias.arena >= 0 ? reinterpret_cast<const float*>(rp(ab, d.bias)) : nullptr;
  char* yb = rp(ab, d.y);
  const int n2 = d.n2 > 0 ? d.n2 : 0x7fffffff;                    // columns >= n2 leave for the second destination
  char* yb2 = d.n2 > 0 ? rp(ab, d.y2) - (int64_t)d.n2 * (d.ydt == DT_BF16 ? 2 : 4) : yb;
  const bool want_stats = d.stats.arena >= 0;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int nl = wn0 + j * 32 + (lane & 31);
    const int n = ntile * BN + nl;
    const float bv = (bias && n < d.N) ? bias[n] : 0.f;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    BnbCol bc{0.f, 0.f, 0.f, 0.f};
    if (bnb) bc = bnb_col(d, ab, n);
    if (staged) {
      // bf16 output through LDS: the 32x32 accumulator layout gives every lane ONE column, i.e. 2-byte global stores in
      // 64-byte pieces; staged, the tile leaves as whole 16-byte chunks of contiguous rows (8x fewer store instructions)
      // (round 4: written as 8-byte row pieces after the quad transpose - 4 x fewer LDS stores than one 2-byte store per accumulator element)
      const QuadT qt(lane);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[i][j][4 * q + r] + bv;
            if (d.flags & kRunRelu) v[r] = fmaxf(v[r], 0.f);
          }
          const uint32_t p01 = pack_bf16x2(v[0], v[1]), p23 = pack_bf16x2(v[2], v[3]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = wm0 + i * 32 + r + 8 * q + 4 * (lane >> 5);
            if (!bnb_chunk && mtile * BM + row < d.M && n < d.N) {
              const uint16_t hv = (uint16_t)((r < 2 ? p01 : p23) >> (16 * (r & 1)));
              if (bnb) bnb_accum(bc, bslope, bf2f(hv), ld_elem(ybn, d.ydt, rowoffb[row] + n), s1, s2, s3);
              else { s1 += v[r]; s2 += v[r] * v[r]; }
            }
          }
          const uint2 pk = qt.pack(p01, p23);
          const int trow = wm0 + i * 32 + 8 * q + 4 * (lane >> 5) + (lane & 3);
          *reinterpret_cast<uint2*>(otile + trow * OS + wn0 + j * 32 + (lane & 28)) = pk;
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
            char* yd = n >= n2 ? yb2 : yb;
            if (d.flags & kRunAccum) v += reinterpret_cast<float*>(yd)[o + n];
            if (d.flags & kRunRelu) v = fmaxf(v, 0.f);
            if (d.ydt == DT_BF16) { const uint16_t hv = f2bf(v); reinterpret_cast<uint16_t*>(yd)[o + n] = hv; if (bnb) v = bf2f(hv); }
            else reinterpret_cast<float*>(yd)[o + n] = v;
            if (bnb) bnb_accum(bc, bslope, v, ld_elem(ybn, d.ydt, rowoffb[row] + n), s1, s2, s3);
            else { s1 += v; s2 += v * v; }
          }
        }
      }
    }
    if (want_stats && !bnb_chunk) {
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      s3 += __shfl_xor(s3, 32);
      if (lane < 32) {
        stat[((wid / WN_) * BN + nl) * 3 + 0] = s1;
        stat[((wid / WN_) * BN + nl) * 3 + 1] = s2;
        stat[((wid / WN_) * BN + nl) * 3 + 2] = s3;
      }
    }
  }
  if (staged) {
    __syncthreads();
    if (!bnb_chunk) {
      for (int q = tid; q < BM * CPR; q += NW * 64) {
        const int row = q / CPR, cc = q - row * CPR;
        const int64_t o = rowoff[row];
        const int n0 = ntile * BN + cc * 8;
        if (o >= 0 && n0 < d.N)                                    // N % 8 == 0 here: a chunk is all valid or all padding
          *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(n0 >= n2 ? yb2 : yb) + o + n0) = *reinterpret_cast<const uint4*>(otile + row * OS + cc * 8);
      }
    } else {
      static_assert((NW * 64) % CPR == 0, "a thread keeps its chunk column");
      const int cc = tid % CPR, n0 = ntile * BN + cc * 8;
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
#pragma unroll
      for (int k = 0; k < KCH; ++k) {
        const int row = tid / CPR + k * (NW * 64 / CPR);
        const int64_t o = rowoff[row];
        if (o < 0 || !cok) continue;
        const uint4 dzv = *reinterpret_cast<const uint4*>(otile + row * OS + cc * 8);
        const uint4 yv = ypre[k];
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(n0 >= n2 ? yb2 : yb) + o + n0) = dzv;
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
      // lanes of a wave with the same chunk column, then the 4 waves through `stat`
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = 32; o >= CPR; o >>= 1) { t0[e] += __shfl_xor(t0[e], o); t1[e] += __shfl_xor(t1[e], o); t2[e] += __shfl_xor(t2[e], o); }
      }
      if (lane < CPR) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float* sp = stat + ((tid >> 6) * BN + cc * 8 + e) * 3;
          sp[0] = t0[e]; sp[1] = t1[e]; sp[2] = t2[e];
        }
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    if (tid < BN) {
      // one row of partials per kBM (= 128) output rows, whatever the tile height: the planner sizes `stats` that way
      constexpr int HALVES = BM / kBM, WMH = WM_ / HALVES;
      float* part = reinterpret_cast<float*>(rp(ab, d.stats));
      const int nrows = (d.M + kBM - 1) / kBM;
      const int nst = bnb ? 3 : 2;                                   // partial rows per block: (sum, sum of squares) or the three backward sums
#pragma unroll
      for (int h = 0; h < HALVES; ++h) {
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (bnb_chunk) {                                               // every wave holds a share of all columns
#pragma unroll
          for (int wv = 0; wv < NW; ++wv) { s1 += stat[(wv * BN + tid) * 3 + 0]; s2 += stat[(wv * BN + tid) * 3 + 1]; s3 += stat[(wv * BN + tid) * 3 + 2]; }
        } else {
#pragma unroll
          for (int wmi = h * WMH; wmi < (h + 1) * WMH; ++wmi) {
            s1 += stat[(wmi * BN + tid) * 3 + 0];
            s2 += stat[(wmi * BN + tid) * 3 + 1];
            s3 += stat[(wmi * BN + tid) * 3 + 2];
          }
        }
        const int srow = mtile * HALVES + h;
        if (srow < nrows) {
          part[((int64_t)srow * nst + 0) * d.Npad + ntile * BN + tid] = s1;
          part[((int64_t)srow * nst + 1) * d.Npad + ntile * BN + tid] = s2;
          if (bnb) part[((int64_t)srow * nst + 2) * d.Npad + ntile * BN + tid] = s3;
        }
      }
    }
  }
}

// WGRAD workgroup order: all (n, k) tiles of one row split read the SAME activation / gradient rows, so they are placed on
// ONE XCD (private L2) back to back; the 8 XCDs work on 8 different splits.  Linear id L -> xcd = L % 8 (dispatch order).
__device__ __forceinline__ void wgrad_block(int gx, int gy, int nsplit, int& ntile, int& ktile, int& split) {
  const int tiles = gx * gy;
  const int L = blockIdx.x;
  const int full = (nsplit / 8) * 8;                       // splits that can be grouped 8 at a time
  if (L < full * tiles) {
    const int xcd = L & 7, idx = L >> 3;
    split = (idx / tiles) * 8 + xcd;
    const int t = idx - (idx / tiles) * tiles;
    ntile = t % gx; ktile = t / gx;
  } else {                                                 // remainder splits: plain order
    const int r = L - full * tiles;
    split = full + r / tiles;
    const int t = r % tiles;
    ntile = t % gx; ktile = t / gx;
  }
}

// ------------------------------------------------------------------------------------------------------------
// WGRAD (fp32 MFMA):  part[split][n][k] = sum_{rows m of the split} dy[m][n] * A[m][k]
// MFMA view: "A operand" = dy^T (i = n, kk = m), "B operand" = gathered activations (kk = m, j = k).
// Workgroup tile 64 (n) x 128 (k); 4 waves as 2x2, each 32 x 64; 32 reduction rows per step.
// LDS tiles are stored in their natural [m][n] / [m][k] layout: the 32x32x2 fp32 MFMA takes ONE scalar per
// lane per operand, so fragment reads are conflict-free ds_read_b32 of 32 consecutive floats per half-wave.
template <typename TA>
__global__ __launch_bounds__(256) void wgrad_kernel(const RunGemm d, const ArenaBases ab) {
  static_assert(sizeof(TA) == 4, "bf16 WGRAD uses the transposing variant");
  constexpr int TN = kWgTN, TK = kWgTK, RS = kWgRows;
  __shared__ __attribute__((aligned(16))) float dys[2][RS][TN + 4];
  __shared__ __attribute__((aligned(16))) float as[2][RS][TK + 4];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int ntile, ktile, split;
  wgrad_block((d.Npad + TN - 1) / TN, (d.ldw + TK - 1) / TK, d.nsplit, ntile, ktile, split);
  const float* x0 = reinterpret_cast<const float*>(rp(ab, d.x[0]));
  const float* x1 = d.x[1].arena >= 0 ? reinterpret_cast<const float*>(rp(ab, d.x[1])) : x0;
  const float* dy = reinterpret_cast<const float*>(rp(ab, d.y));
  float* part = reinterpret_cast<float*>(rp(ab, d.w)) + (int64_t)split * d.Npad * d.ldw;

  // rows of this split, in units of RS
  const int nsteps_total = (d.M + RS - 1) / RS;
  const int per = (nsteps_total + d.nsplit - 1) / d.nsplit;
  const int step0 = split * per;
  const int step1 = min(nsteps_total, step0 + per);
  const int TF = d.Tout * d.Fo;

  // A-gather assignment: chunk column ca (32 chunks of 4 floats per row), rows ra + 8p
  const int ca = tid & 31, ra = tid >> 5;
  const int kcol = ktile * TK + ca * 4;
  int sgi = -1, j0 = 0;
  for (int s = 0; s < d.nseg; ++s) {
    const int plen = (d.seg[s].len + 31) / 32 * 32;
    if (kcol >= d.seg[s].koff && kcol < d.seg[s].koff + plen) { sgi = s; j0 = kcol - d.seg[s].koff; }
  }
  Seg sg;
  if (sgi >= 0) sg = d.seg[sgi]; else { sg.src = 0; sg.dt = 0; sg.off = 0; sg.len = 0; sg.koff = 0; }
  // dy assignment: chunk column cd (16 chunks of 4 floats), rows rd + 16p
  const int cd = tid & 15, rd = tid >> 4;
  const int ncol = ntile * TN + cd * 4;

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  const int wn = (wid >> 1) * 32, wk = (wid & 1) * 64;

  uint4 aReg[4], dReg[2];
  auto issue = [&](int step) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int m = step * RS + ra + 8 * p;
      const bool v = m < d.M;
      const int mm = v ? m : 0;
      const int b = mm / TF, rem = mm - b * TF, u = rem / d.Fo, fo = rem - u * d.Fo;
      aReg[p] = load_a_chunk<float>(d, x0, x1, sg, (int64_t)b * d.bstride[0] + d.base[0], (int64_t)b * d.bstride[1] + d.base[1],
                                    u, fo, v && sgi >= 0, j0);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int m = step * RS + rd + 16 * p;
      uint4 z = make_uint4(0, 0, 0, 0);
      if (m < d.M) {
        const int b = m / TF, rem = m - b * TF, u = rem / d.Fo, fo = rem - u * d.Fo;
        const float* src = dy + (int64_t)b * d.y_bstride + (int64_t)u * d.y_tstride + (int64_t)fo * d.y_fstride + d.y_off + ncol;
        if (ncol + 4 <= d.N) z = *reinterpret_cast<const uint4*>(src);
        else {
          uint32_t v[4] = {0, 0, 0, 0};
          for (int e = 0; e < 4; ++e) if (ncol + e < d.N) v[e] = reinterpret_cast<const uint32_t*>(src)[e];
          z = make_uint4(v[0], v[1], v[2], v[3]);
        }
      }
      dReg[p] = z;
    }
  };
  if (step0 < step1) issue(step0);
  for (int st = step0; st < step1; ++st) {
    const int buf = (st - step0) & 1;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<uint4*>(&as[buf][ra + 8 * p][ca * 4]) = aReg[p];
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<uint4*>(&dys[buf][rd + 16 * p][cd * 4]) = dReg[p];
    __syncthreads();
    if (st + 1 < step1) issue(st + 1);
#pragma unroll
    for (int mm = 0; mm < RS; mm += 2) {
      const int row = mm + (lane >> 5);
      const float a = dys[buf][row][wn + (lane & 31)];
      const float b0 = as[buf][row][wk + (lane & 31)];
      const float b1 = as[buf][row][wk + 32 + (lane & 31)];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
    }
  }
  // store: C layout col = lane&31 -> k, row -> n
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int n = ntile * TN + wn + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      const int k = ktile * TK + wk + j * 32 + (lane & 31);
      if (n < d.Npad && k < d.ldw) part[(int64_t)n * d.ldw + k] = acc[j][e];
    }
}

// ------------------------------------------------------------------------------------------------------------
// WGRAD (bf16 MFMA).  The contraction index of a weight gradient is the pixel row m, which is the *strided* index of
// both channels-last operands, while v_mfma_f32_16x16x32_bf16 wants 8 consecutive k per lane.  gfx950's LDS transpose
// read does the turn for free: ds_read_b64_tr_b16 with lane i of a 16-lane group pointing at (row 4g + i/4, col 4(i%4))
// of a row-major [m][c] tile returns (rows 4g..4g+3, col i) - four consecutive m for one channel (probed on MI355X,
// tools/probe_tr16.hip).  Two such reads (rows +0 and +16) fill one 8-wide k fragment; A and B use the same m order.
// Workgroup tile TN (n) x 128 (k), 4 waves as 2x2, 32 reduction rows per step, fp32 partial sums per row split.
typedef __attribute__((ext_vector_type(4))) short s16x4;
struct Frag8 { s16x4 lo, hi; };

__device__ __forceinline__ bf16x8 tr_frag(const uint16_t* tile, int pitch, int col0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const uint16_t* p = tile + (4 * g + (i >> 2)) * pitch + col0 + 4 * (i & 3);
  Frag8 f;
  f.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  f.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 16 * pitch));
  return __builtin_bit_cast(bf16x8, f);
}

template <int TN>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const RunGemm d, const ArenaBases ab) {
  constexpr int TK = kWgTK, RS = kWgRows, PN = TN + 8, PK = TK + 8;
  constexpr int NT = TN / 32;                  // 16-wide n tiles per wave
  constexpr int DPASS = TN / 64;               // dy chunk passes per thread
  __shared__ __attribute__((aligned(16))) uint16_t dys[2][RS][PN];
  __shared__ __attribute__((aligned(16))) uint16_t as[2][RS][PK];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int ntile, ktile, split;
  wgrad_block((d.Npad + TN - 1) / TN, (d.ldw + TK - 1) / TK, d.nsplit, ntile, ktile, split);
  const bf16_t* x0 = reinterpret_cast<const bf16_t*>(rp(ab, d.x[0]));
  const bf16_t* x1 = d.x[1].arena >= 0 ? reinterpret_cast<const bf16_t*>(rp(ab, d.x[1])) : x0;
  const uint16_t* dy = reinterpret_cast<const uint16_t*>(rp(ab, d.y));
  float* part = reinterpret_cast<float*>(rp(ab, d.w)) + (int64_t)split * d.Npad * d.ldw;

  const int nsteps_total = (d.M + RS - 1) / RS;
  const int per = (nsteps_total + d.nsplit - 1) / d.nsplit;
  const int step0 = split * per;
  const int step1 = min(nsteps_total, step0 + per);
  const int TF = d.Tout * d.Fo;

  // A gather: 16 chunks (8 bf16) per row; rows ra + 16p
  const int ca = tid & 15, ra = tid >> 4;
  const int kcol = ktile * TK + ca * 8;
  int sgi = -1, j0 = 0;
  for (int s = 0; s < d.nseg; ++s) {
    const int plen = (d.seg[s].len + 63) / 64 * 64;
    if (kcol >= d.seg[s].koff && kcol < d.seg[s].koff + plen) { sgi = s; j0 = kcol - d.seg[s].koff; }
  }
  Seg sg;
  if (sgi >= 0) sg = d.seg[sgi]; else { sg.src = 0; sg.dt = 0; sg.off = 0; sg.len = 0; sg.koff = 0; }
  // dy: TN/8 chunks per row
  constexpr int DCH = TN / 8;
  const int cd = tid % DCH, rd = tid / DCH;     // rows rd + (256/DCH) p
  const int ncol = ntile * TN + cd * 8;

  f32x4 acc[NT][4];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wn = (wid >> 1) * (TN / 2), wk = (wid & 1) * 64;

  // Row decode is incremental: a thread's rows advance by RS per step, so (b, q = position inside the batch item) is
  // carried along and only the split u = q / Fo remains (a shift when Fo is a power of two, as in every conv layer).
  const int fsh = (d.Fo & (d.Fo - 1)) == 0 ? __ffs(d.Fo) - 1 : -1;
  struct RowPos { int b, q; };
  auto init_pos = [&](int m) { RowPos r; r.b = m / TF; r.q = m - r.b * TF; return r; };
  auto advance = [&](RowPos& r) { r.q += RS; while (r.q >= TF) { r.q -= TF; ++r.b; } };
  RowPos apos[2], dpos[DPASS];
#pragma unroll
  for (int p = 0; p < 2; ++p) apos[p] = init_pos(step0 * RS + ra + 16 * p);
#pragma unroll
  for (int p = 0; p < DPASS; ++p) dpos[p] = init_pos(step0 * RS + rd + (256 / DCH) * p);
  const bool a_full = sgi >= 0 && sg.src >= 0 && j0 + 8 <= sg.len;          // chunk entirely inside the run
  const bool a_ones = sgi >= 0 && sg.src < 0 && j0 == 0;
  const int asrc = sg.src > 0 ? 1 : 0;
  const bf16_t* xs = asrc ? x1 : x0;
  const bool d_full = ncol + 8 <= d.N;

  uint4 aReg[2], dReg[DPASS];
  auto issue = [&](int step) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int m = step * RS + ra + 16 * p;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m < d.M) {
        const int u = fsh >= 0 ? (apos[p].q >> fsh) : (apos[p].q / d.Fo);
        const int fo = apos[p].q - u * d.Fo;
        if (a_full) {
          const int tt = u + sg.dt;
          const int rr = sg.off + fo * d.fstride[asrc] + j0;
          if (tt >= 0 && tt < d.Tin[asrc]) {
            const bf16_t* src = xs + (int64_t)apos[p].b * d.bstride[asrc] + d.base[asrc] + (int64_t)tt * d.tstride[asrc] + rr;
            if (rr >= 0 && rr + 8 <= d.rowlen[asrc] && (reinterpret_cast<uintptr_t>(src) & 15) == 0) v = *reinterpret_cast<const uint4*>(src);
            else if (rr + 8 > 0 && rr < d.rowlen[asrc])
              v = load_a_chunk<bf16_t>(d, x0, x1, sg, (int64_t)apos[p].b * d.bstride[0] + d.base[0], (int64_t)apos[p].b * d.bstride[1] + d.base[1], u, fo, true, j0);
          }
        } else if (a_ones) {
          v.x = 0x3f80u;
        } else if (sgi >= 0 && sg.src >= 0) {
          v = load_a_chunk<bf16_t>(d, x0, x1, sg, (int64_t)apos[p].b * d.bstride[0] + d.base[0], (int64_t)apos[p].b * d.bstride[1] + d.base[1], u, fo, true, j0);
        }
      }
      aReg[p] = v;
      advance(apos[p]);
    }
#pragma unroll
    for (int p = 0; p < DPASS; ++p) {
      const int m = step * RS + rd + (256 / DCH) * p;
      uint4 z = make_uint4(0, 0, 0, 0);
      if (m < d.M) {
        const int u = fsh >= 0 ? (dpos[p].q >> fsh) : (dpos[p].q / d.Fo);
        const int fo = dpos[p].q - u * d.Fo;
        const uint16_t* src = dy + (int64_t)dpos[p].b * d.y_bstride + (int64_t)u * d.y_tstride + (int64_t)fo * d.y_fstride + d.y_off + ncol;
        if (d_full && (reinterpret_cast<uintptr_t>(src) & 15) == 0) z = *reinterpret_cast<const uint4*>(src);
        else {
          uint16_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          for (int e = 0; e < 8; ++e) if (ncol + e < d.N) v[e] = src[e];
          z = make_uint4(v[0] | (uint32_t)v[1] << 16, v[2] | (uint32_t)v[3] << 16, v[4] | (uint32_t)v[5] << 16, v[6] | (uint32_t)v[7] << 16);
        }
      }
      dReg[p] = z;
      advance(dpos[p]);
    }
  };
  if (step0 < step1) issue(step0);
  for (int st = step0; st < step1; ++st) {
    const int buf = (st - step0) & 1;
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<uint4*>(&as[buf][ra + 16 * p][ca * 8]) = aReg[p];
#pragma unroll
    for (int p = 0; p < DPASS; ++p) *reinterpret_cast<uint4*>(&dys[buf][rd + (256 / DCH) * p][cd * 8]) = dReg[p];
    __syncthreads();
    if (st + 1 < step1) issue(st + 1);
    bf16x8 af[NT], bfr[4];
#pragma unroll
    for (int a = 0; a < NT; ++a) af[a] = tr_frag(&dys[buf][0][0], PN, wn + a * 16, lane);
#pragma unroll
    for (int b = 0; b < 4; ++b) bfr[b] = tr_frag(&as[buf][0][0], PK, wk + b * 16, lane);
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
  }
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = ntile * TN + wn + a * 16 + 4 * (lane >> 4) + r;
        const int k = ktile * TK + wk + b * 16 + (lane & 15);
        if (n < d.Npad && k < d.ldw) part[(int64_t)n * d.ldw + k] = acc[a][b][r];
      }
}

// ------------------------------------------------------------------------------------------------------------
// WGRAD (bf16 MFMA), LDS-DMA variant for aligned layers: both operand tiles go HBM/L2 -> LDS with
// global_load_lds_dwordx4 (no VGPR staging, no ds_write pass - the write pass cost more LDS cycles than the MFMAs took).
// LDS rows are unpadded (lane-linear DMA destination); bank conflicts of the transpose reads are avoided by XOR-ing the
// 32-byte piece index with the row (applied on the SOURCE chunk index: position p of row r holds chunk p ^ ((r & m) << 1)).
// Padding / out-of-range rows come from a zero page, the bias "ones" run from a ones page (1.0, 0, 0, ...).
template <int TN>
__device__ __forceinline__ bf16x8 tr_frag_swz(const uint16_t* tile, int col0, int lane) {
  constexpr int MASK = TN >= 128 ? 7 : TN == 64 ? 3 : 0;   // 32-byte pieces per row: 8 (256-byte rows) or 4 (128-byte rows); narrow tiles are not swizzled
  const int g = lane >> 4, i = lane & 15;
  const int r = 4 * g + (i >> 2);
  const int piece = ((col0 >> 4) + 0);             // col0 is a multiple of 16 elements = one 32-byte piece
  const uint16_t* p0 = tile + r * TN + (((piece) ^ (r & MASK)) << 4) + 4 * (i & 3);
  const int r1 = r + 16;
  const uint16_t* p1 = tile + r1 * TN + (((piece) ^ (r1 & MASK)) << 4) + 4 * (i & 3);
  Frag8 f;
  f.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
  f.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
  return __builtin_bit_cast(bf16x8, f);
}

// TK x NTHR: 128 x 256 threads (tiles up to 128 x 128) or 256 x 512 threads (the 256 x 256 tile of the wide layers: half the operand
// bytes through L2 -> LDS per MFMA, which is what bounds the 128 x 128 tile - 64 B/clk/CU at the MFMA rate, the DMA stream's ceiling)
// DUAL (needs S == 4): two 32-row steps per barrier - stages i, i+1 are multiplied while i+2, i+3 fly; the barrier, the vmcnt wait and
// the DMA issue burst are paid once per 64 rows (the row splits, i.e. the partial sums, stay those of the 32-row steps)
// ONESM (kRunOnesMfma, 256 x 256 tile): the bias "ones" run - the LAST 64 packed columns - is not a k tile of its own (a whole tile's DMA stream and MFMAs for
// one column): the k tiles cover ldw - 64 columns and the workgroups of k tile 0 multiply their dy fragments with a constant ones operand, two extra
// MFMAs per wave and step, no extra bytes ([h1 | h2 | ones] = 832 columns of FullSubNet's upper sub-band layer: 3 tiles instead of 4)
template <int TN, int S, int TK = kWgTK, int NTHR = 256, bool DUAL = false, bool ONESM = false>
__global__ __launch_bounds__(NTHR) void wgrad_bf16_dma_kernel(const RunGemm d, const ArenaBases ab) {
  constexpr int RS = kWgRows, NW = NTHR / 64;
  // wave grid: 2 (n) x 2 (k) for the 128 / 64 / 32 wide n tiles, 1 x 4 for the 16 wide one (thin layers: N <= 16, e.g. the mask
  // layer's 2 -> 8 channels - a 64-wide tile there is 7/8 padding and the launch is bound by the activation stream instead)
  constexpr int WK = (TN >= 32 ? 2 : 4) * (NW / 4);
  constexpr int PWN = TN / (NW / WK), PWK = TK / WK;
  constexpr int NT = PWN / 16, KB = PWK / 16;
  constexpr int DMASK = TN >= 128 ? 7 : TN == 64 ? 3 : 0;
  __shared__ __attribute__((aligned(16))) uint16_t dys[S][RS * TN];
  __shared__ __attribute__((aligned(16))) uint16_t as[S][RS * TK];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int ntile, ktile, split;
  const int ldk = ONESM ? d.ldw - 64 : d.ldw;      // columns the k tiles cover
  wgrad_block((d.Npad + TN - 1) / TN, (ldk + TK - 1) / TK, d.nsplit, ntile, ktile, split);
  const uint16_t* x0 = reinterpret_cast<const uint16_t*>(rp(ab, d.x[0]));
  const uint16_t* x1 = d.x[1].arena >= 0 ? reinterpret_cast<const uint16_t*>(rp(ab, d.x[1])) : x0;
  const uint16_t* dy = reinterpret_cast<const uint16_t*>(rp(ab, d.y));
  const uint16_t* zp = reinterpret_cast<const uint16_t*>(rp(ab, d.zero));
  const uint16_t* onep = zp + 128;                 // second 256-byte page: bf16 (1, 0, 0, ...)
  float* part = reinterpret_cast<float*>(rp(ab, d.w)) + (int64_t)split * d.Npad * d.ldw;
  if (ntile * TN >= d.N) {                         // n tile made of padding rows only (N = 8 in a 32-row packed matrix): zeros
    for (int i = tid; i < TN * TK; i += NTHR) {
      const int n = ntile * TN + i / TK, k = ktile * TK + i % TK;
      if (n < d.Npad && k < ldk) part[(int64_t)n * d.ldw + k] = 0.f;
    }
    if (ONESM && ktile == 0)
      for (int i = tid; i < TN * 64; i += NTHR) {
        const int n = ntile * TN + i / 64;
        if (n < d.Npad) part[(int64_t)n * d.ldw + ldk + i % 64] = 0.f;
      }
    return;
  }

  // Reduction rows are walked per batch item in steps of RS rows: step (b, s) covers rows q = RS*s .. RS*s+RS-1 of item b
  // (q >= Tout*Fo reads the zero page).  With Fo a power of two the (frame, bin) of a row is then a per-thread constant plus
  // a wave-uniform term, so one DMA address costs a handful of VALU instructions next to the 4 MFMAs it feeds.
  const int Fo = d.Fo, Tout = d.Tout;
  const int TF = Tout * Fo;
  const int nb = d.M / TF;
  const int spb = (TF + RS - 1) / RS;
  const int nsteps_total = nb * spb;
  const int per = (nsteps_total + d.nsplit - 1) / d.nsplit;
  const int step0 = split * per;
  const int step1 = min(nsteps_total, step0 + per);
  const int fsh = (Fo & (Fo - 1)) == 0 ? __ffs(Fo) - 1 : -1;

  // A tile: 16 chunks per row, thread -> (row ra + 16p, LDS position pa); it fetches source chunk qa = pa ^ ((row & 7) << 1)
  constexpr int ACH = TK / 8;                      // 16-byte chunks per row of the A tile; NTHR / ACH = 16 rows per pass
  constexpr int AROWS = NTHR / ACH, APASS = RS / AROWS;   // rows per pass (16 or 8), passes (2 or 4)
  const int pa = tid % ACH, ra = tid / ACH;
  const int qa = pa ^ ((ra & 7) << 1);
  const int kcol = ktile * TK + qa * 8;
  int sgi = -1, j0 = 0;
  for (int s = 0; s < d.nseg; ++s) {
    const int plen = (d.seg[s].len + 63) / 64 * 64;
    if (kcol >= d.seg[s].koff && kcol < d.seg[s].koff + plen) { sgi = s; j0 = kcol - d.seg[s].koff; }
  }
  Seg sg;
  if (sgi >= 0) sg = d.seg[sgi]; else { sg.src = 0; sg.dt = 0; sg.off = 0; sg.len = 0; sg.koff = 0; }
  const bool a_real = sgi >= 0 && sg.src >= 0 && j0 + 8 <= sg.len;
  const bool a_ones = sgi >= 0 && sg.src < 0 && j0 == 0;
  const int asrc = sg.src > 0 ? 1 : 0;
  // geometry of this thread's source in registers (a run-time index into the by-value argument becomes kernarg loads in the loop)
  const int a_Tin = asrc ? d.Tin[1] : d.Tin[0];
  const int a_fstride = asrc ? d.fstride[1] : d.fstride[0];
  const int a_rowlen = asrc ? d.rowlen[1] : d.rowlen[0];
  const int64_t a_tstride = asrc ? d.tstride[1] : d.tstride[0];
  const int64_t bs0 = d.bstride[0], bs1 = d.bstride[1], ts0 = d.tstride[0], ts1 = d.tstride[1];
  const int fs0 = d.fstride[0], fs1 = d.fstride[1];
  const uint16_t* xs_base = (asrc ? x1 + d.base[1] : x0 + d.base[0]) + sg.off + j0;
  const int a_r0 = sg.off + j0;
  const int a_dt = sg.dt;
  // valid bins of this thread's chunk: 0 <= a_r0 + fo*fstride, a_r0 + fo*fstride + 8 <= rowlen
  int flo = 0, fhi = Fo - 1;
  if (a_fstride > 0) {
    if (a_r0 < 0) flo = (-a_r0 + a_fstride - 1) / a_fstride;
    const int room = a_rowlen - 8 - a_r0;
    fhi = room < 0 ? -1 : min(Fo - 1, room / a_fstride);
  } else if (a_r0 < 0 || a_r0 + 8 > a_rowlen) {
    fhi = -1;
  }
  const unsigned fspan = fhi >= flo ? (unsigned)(fhi - flo) : 0u;
  const bool a_any = a_real && fhi >= flo;
  // dy tile: TN/8 chunks per row
  constexpr int DCH = TN / 8;
  constexpr int DROWS = NTHR / DCH;                 // rows per pass (16 or 32; the narrow tiles need only the first 128 / 64 threads)
  constexpr int DPASS = RS >= DROWS ? RS / DROWS : 1;
  constexpr int DWAVES = RS >= DROWS ? NW : RS * DCH / 64;   // waves that move the dy tile
  constexpr int NL = APASS + DPASS;                    // DMAs per thread per stage (waves >= DWAVES: 2)
  const bool dwave = __builtin_amdgcn_readfirstlane(wid) < DWAVES;
  const int pd = tid % DCH, rd = tid / DCH;
  const int qd = pd ^ ((rd & DMASK) << 1);
  const int ncol = ntile * TN + qd * 8;
  const bool d_ok = ncol + 8 <= d.N;
  const int64_t y_bstride = d.y_bstride, y_tstride = d.y_tstride, y_fstride = d.y_fstride;
  const uint16_t* dy_base = dy + d.y_off + ncol;

  // per-thread constant part of each row (pow-2 Fo): row r of a step sits at frame u0 + ur, bin f0 + fr
  int a_r[APASS], a_ur[APASS], a_fr[APASS], y_r[DPASS];
  const uint16_t* a_tc[APASS];
  const uint16_t* y_tc[DPASS];
#pragma unroll
  for (int p = 0; p < APASS; ++p) {
    const int r = ra + AROWS * p;
    a_r[p] = r;
    a_ur[p] = (fsh >= 0 && Fo < RS) ? (r >> fsh) : 0;
    a_fr[p] = (fsh >= 0 && Fo < RS) ? (r & (Fo - 1)) : r;
    a_tc[p] = xs_base + (int64_t)(a_ur[p] + a_dt) * a_tstride + (int64_t)a_fr[p] * a_fstride;
  }
#pragma unroll
  for (int p = 0; p < DPASS; ++p) {
    const int r = rd + DROWS * p;
    y_r[p] = r;
    const int ur = (fsh >= 0 && Fo < RS) ? (r >> fsh) : 0, fr = (fsh >= 0 && Fo < RS) ? (r & (Fo - 1)) : r;
    y_tc[p] = dy_base + (int64_t)ur * y_tstride + (int64_t)fr * y_fstride;
  }
  const uint32_t wbase = __builtin_amdgcn_readfirstlane(wid) * 1024;   // bytes: this wave's 1 KiB inside each 4 KiB pass
  const uint32_t as_l = lds_addr(&as[0][0]) + wbase, dys_l = lds_addr(&dys[0][0]) + wbase;

  int ib = step0 / spb, is = step0 - ib * spb;     // issue pointer (wave-uniform)
  auto dma = [&](int stage) {
    const int q0 = is * RS;
    if (fsh >= 0) {
      const int u0 = q0 >> fsh, f0 = q0 & (Fo - 1);
      const int64_t o0 = (int64_t)ib * bs0 + (int64_t)u0 * ts0 + (int64_t)f0 * fs0;
      const int64_t o1 = (int64_t)ib * bs1 + (int64_t)u0 * ts1 + (int64_t)f0 * fs1;
      const int64_t oy = (int64_t)ib * y_bstride + (int64_t)u0 * y_tstride + (int64_t)f0 * y_fstride;
      const int64_t oa = asrc ? o1 : o0;
#pragma unroll
      for (int p = 0; p < APASS; ++p) {
        const bool rowv = q0 + a_r[p] < TF;
        const bool v = a_any && rowv && (unsigned)(u0 + a_ur[p] + a_dt) < (unsigned)a_Tin &&
                       (unsigned)(f0 + a_fr[p] - flo) <= fspan;
        const uint16_t* src = v ? a_tc[p] + oa : ((a_ones && rowv) ? onep : zp);
        dma16(src, as_l + stage * (RS * TK * 2) + p * (NTHR * 16));
      }
      if (dwave) {
#pragma unroll
        for (int p = 0; p < DPASS; ++p) {
          const uint16_t* src = (d_ok && q0 + y_r[p] < TF) ? y_tc[p] + oy : zp;
          dma16(src, dys_l + stage * (RS * TN * 2) + p * (NTHR * 16));
        }
      }
    } else {                                         // general Fo: one division per row
#pragma unroll
      for (int p = 0; p < APASS; ++p) {
        const int q = q0 + a_r[p];
        const int u = q / Fo, fo = q - u * Fo;
        const bool rowv = q < TF;
        const bool v = a_any && rowv && (unsigned)(u + a_dt) < (unsigned)a_Tin && (unsigned)(fo - flo) <= fspan;
        const uint16_t* src = v ? xs_base + (int64_t)ib * (asrc ? bs1 : bs0) + (int64_t)(u + a_dt) * a_tstride + (int64_t)fo * a_fstride
                                : ((a_ones && rowv) ? onep : zp);
        dma16(src, as_l + stage * (RS * TK * 2) + p * (NTHR * 16));
      }
      if (dwave) {
#pragma unroll
        for (int p = 0; p < DPASS; ++p) {
          const int q = q0 + y_r[p];
          const int u = q / Fo, fo = q - u * Fo;
          const uint16_t* src = (d_ok && q < TF) ? dy_base + (int64_t)ib * y_bstride + (int64_t)u * y_tstride + (int64_t)fo * y_fstride : zp;
          dma16(src, dys_l + stage * (RS * TN * 2) + p * (NTHR * 16));
        }
      }
    }
    if (++is == spb) { is = 0; ++ib; }
  };

  f32x4 acc[NT][KB];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < KB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wn = (wid / WK) * PWN, wk = (wid % WK) * PWK;
  // ONESM: the WK waves that share a dy fragment range split its NT blocks between them (NT / WK each)
  constexpr int NBIAS = ONESM ? NT / WK : 1;
  static_assert(!ONESM || (NT % WK == 0 && WK == 4 && NBIAS == 2), "ones-by-MFMA: 256 x 256 tile of 8 waves");
  f32x4 bacc[NBIAS];
#pragma unroll
  for (int u = 0; u < NBIAS; ++u) bacc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int kw = __builtin_amdgcn_readfirstlane(wid % WK);
  const bool bias_wg = ONESM && __builtin_amdgcn_readfirstlane(ktile) == 0;
  auto bias_mul = [&](const bf16x8* af) {
    if constexpr (ONESM) {
      if (bias_wg) {
        Frag8 o;
        o.lo = s16x4{0x3F80, 0x3F80, 0x3F80, 0x3F80};
        o.hi = o.lo;
        const bf16x8 ones = __builtin_bit_cast(bf16x8, o);
        switch (kw) {                                  // wave-uniform: a scalar branch, the fragment registers stay statically indexed
          case 0: bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], ones, bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], ones, bacc[1], 0, 0, 0); break;
          case 1: bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], ones, bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[3], ones, bacc[1], 0, 0, 0); break;
          case 2: bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[4], ones, bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[5], ones, bacc[1], 0, 0, 0); break;
          default: bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[6], ones, bacc[0], 0, 0, 0); bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[7], ones, bacc[1], 0, 0, 0); break;
        }
      }
    }
  };

  // S-stage ring: DMA runs S-1 row steps ahead of the MFMAs; per step one vmcnt wait for the oldest stage + one LDS barrier
  const int nst = step1 - step0;
  int issued = 0, istage = 0;
  if constexpr (DUAL) {
    static_assert(S == 4, "two steps per barrier: ring of 4");
    auto mul = [&](int slot) {
      bf16x8 af[NT], bfr[KB];
#pragma unroll
      for (int a = 0; a < NT; ++a) af[a] = tr_frag_swz<TN>(&dys[slot][0], wn + a * 16, lane);
#pragma unroll
      for (int b = 0; b < KB; ++b) bfr[b] = tr_frag_swz<TK>(&as[slot][0], wk + b * 16, lane);
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < KB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
      bias_mul(af);
    };
    for (int i = 0; i < 2; ++i)
      if (issued < nst) { dma(istage); istage = (istage + 1) & 3; ++issued; }
    for (int i = 0; i < nst; i += 2) {
      wait_vm<0>();                                  // stages i, i + 1 (everything issued so far) have landed
      lds_barrier();                                 // ... for every wave; the slots of stages i - 2, i - 1 are free again
      for (int u = 0; u < 2; ++u)
        if (issued < nst) { dma(istage); istage = (istage + 1) & 3; ++issued; }
      mul(i & 3);
      if (i + 1 < nst) mul((i + 1) & 3);
    }
  } else {
  for (int i = 0; i < S - 1; ++i)
    if (issued < nst) { dma(istage); istage = istage + 1 == S ? 0 : istage + 1; ++issued; }
  int cstage = 0;
  for (int i = 0; i < nst; ++i) {
    if (dwave) wait_stage<NL, S>(nst - 1 - i); else wait_stage<APASS, S>(nst - 1 - i);
    lds_barrier();
    if (issued < nst) { dma(istage); istage = istage + 1 == S ? 0 : istage + 1; ++issued; }
    bf16x8 af[NT], bfr[KB];
#pragma unroll
    for (int a = 0; a < NT; ++a) af[a] = tr_frag_swz<TN>(&dys[cstage][0], wn + a * 16, lane);
#pragma unroll
    for (int b = 0; b < KB; ++b) bfr[b] = tr_frag_swz<TK>(&as[cstage][0], wk + b * 16, lane);
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int b = 0; b < KB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    bias_mul(af);
    cstage = cstage + 1 == S ? 0 : cstage + 1;
  }
  }
  if (d.flags & (1 << 30)) {                        // tuning (SEFD_WG_DBG=8): no partial-sum stores; every accumulator element stays live
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int b = 0; b < KB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(acc[a][b][r]));
    return;
  }
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < KB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = ntile * TN + wn + a * 16 + 4 * (lane >> 4) + r;
        const int k = ktile * TK + wk + b * 16 + (lane & 15);
        if (n < d.Npad && k < ldk) part[(int64_t)n * d.ldw + k] = acc[a][b][r];
      }
  if constexpr (ONESM) {
    if (bias_wg) {                                   // every column of the ones product holds the row sum: column 0 is the bias, the run's other 63 are zeros
#pragma unroll
      for (int u = 0; u < NBIAS; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = ntile * TN + wn + (kw * NBIAS + u) * 16 + 4 * (lane >> 4) + r;
          if (n < d.Npad) {
            float* row = part + (int64_t)n * d.ldw + ldk;
#pragma unroll
            for (int c = 0; c < 4; ++c) row[16 * c + (lane & 15)] = (c == 0 && (lane & 15) == 0) ? bacc[u][r] : 0.f;
          }
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// WGRAD, rank-N form (kRunRank; round 6): N <= 4 outputs, K <= 512 inputs, millions of contiguous rows - FullSubNet's sub-band head (models.py:668-672 through
// SequenceModel's fc_output_layer: 384 -> 2).  On the tiled kernels N = 2 is padded to a 64-wide tile of a non-DMA kernel (dy rows of 4 bytes are no 16-byte
// chunks): 1.2 ms alone, 3.3 ms beside the recurrence, for a pass over 2.4 GB.  Here a thread owns one 16-byte chunk of the K axis and every RL-th row of the
// workgroup's split: one 16-byte load of x and one 2 N-byte load of dy per row, 8 N fmas; the row lanes fold through LDS in lane order.  Same split boundaries
// as the tiled kernels (32-row steps), whole [Npad][ldw] block written.
template <int NR>
__global__ __launch_bounds__(256) void wgrad_rank_kernel(const RunGemm d, const ArenaBases ab) {
  __shared__ float red[2048 * NR];                          // [RL][NR][K] floats: CH * RL <= 256 threads, K = 8 CH
  __shared__ float redb[256][NR];
  const int tid = threadIdx.x, split = blockIdx.x;
  const Seg sg = d.seg[0];
  const int K = sg.len, CH = K / 8, RL = 256 / CH;           // chunks per row, row lanes
  const int kc = tid % CH, rl = tid / CH;
  const bool act = rl < RL;
  const uint16_t* x = reinterpret_cast<const uint16_t*>(rp(ab, d.x[0])) + sg.off + 8 * kc;
  const uint16_t* dy = reinterpret_cast<const uint16_t*>(rp(ab, d.y)) + d.y_off;
  float* part = reinterpret_cast<float*>(rp(ab, d.w)) + (int64_t)split * d.Npad * d.ldw;
  const int nsteps = (d.M + kWgRows - 1) / kWgRows;
  const int per = (nsteps + d.nsplit - 1) / d.nsplit;
  const int64_t m0 = (int64_t)split * per * kWgRows, m1 = min((int64_t)d.M, (int64_t)(split + 1) * per * kWgRows);
  const int fs = d.fstride[0], ys = d.y_fstride;
  float acc[NR][8], bs[NR];
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    bs[n] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[n][e] = 0.f;
  }
  auto row = [&](const uint4 xv, const float (&g)[NR]) {
    const uint32_t w[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xe = bf2f((uint16_t)(w[e >> 1] >> (16 * (e & 1))));
#pragma unroll
      for (int n = 0; n < NR; ++n) acc[n][e] = fmaf(g[n], xe, acc[n][e]);
    }
#pragma unroll
    for (int n = 0; n < NR; ++n) bs[n] += g[n];
  };
  auto ldy = [&](int64_t m, float (&g)[NR]) {
#pragma unroll
    for (int n = 0; n < NR; ++n) g[n] = n < d.N ? bf2f(dy[m * ys + n]) : 0.f;
  };
  if (act) {
    int64_t m = m0 + rl;
    for (; m + 3 * RL < m1; m += 4 * RL) {                  // four rows in flight
      uint4 xv[4];
      float g[4][NR];
#pragma unroll
      for (int u = 0; u < 4; ++u) { xv[u] = *reinterpret_cast<const uint4*>(x + (m + u * RL) * fs); ldy(m + u * RL, g[u]); }
#pragma unroll
      for (int u = 0; u < 4; ++u) row(xv[u], g[u]);
    }
    for (; m < m1; m += RL) {
      float g[NR];
      ldy(m, g);
      row(*reinterpret_cast<const uint4*>(x + m * fs), g);
    }
#pragma unroll
    for (int n = 0; n < NR; ++n) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(rl * NR + n) * K + 8 * kc + e] = acc[n][e];
      if (kc == 0) redb[rl][n] = bs[n];
    }
  }
  // the split's whole [Npad][ldw] block: zeros where nothing lands
  for (int i = tid; i < d.Npad * d.ldw; i += 256) part[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < d.N * K; i += 256) {
    const int n = i / K, k = i - n * K;
    float v = red[n * K + k];
    for (int r = 1; r < RL; ++r) v += red[(r * NR + n) * K + k];
    part[(int64_t)n * d.ldw + sg.koff + k] = v;
  }
  if (d.nseg == 2 && tid < d.N) {
    float v = redb[0][tid];
    for (int r = 1; r < RL; ++r) v += redb[r][tid];
    part[(int64_t)tid * d.ldw + d.seg[1].koff] = v;
  }
}

static bool launch_wgrad_rank(const RunGemm& d, const ArenaBases& ab, hipStream_t st) {
  if (!(d.flags & kRunRank) || !wgrad_rank_form(d)) return false;
  const dim3 grid((unsigned)d.nsplit);
  if (d.N <= 2) hipLaunchKernelGGL((wgrad_rank_kernel<2>), grid, dim3(256), 0, st, d, ab);
  else hipLaunchKernelGGL((wgrad_rank_kernel<4>), grid, dim3(256), 0, st, d, ab);
  return true;
}

// ------------------------------------------------------------------------------------------------------------
// Ring depth of the LDS-DMA pipelines.  SEFD_RG_STAGES / SEFD_WG_STAGES (2..4) override the defaults for tuning runs.
static int env_stages(const char* name, int dflt) {
  const char* e = tune_str(name);
  if (!e) return dflt;
  const int v = atoi(e);
  return v >= 2 && v <= 4 ? v : dflt;
}

// Tile-shape note (measured on MI355X, DCCRN B=32, profiles/r01_tuning_notes.md): the kernel template also builds as a
// 256 x 128 tile (4 or 8 waves, 2- or 3-stage ring), as a 3/4-stage 128 x 128 ring (one workgroup per CU) and with the
// fragment reads / next-stage DMAs interleaved between the MFMA groups; none of them beat two co-resident 128 x 128
// workgroups with a 2-stage ring, so that is the only configuration launched.  SEFD_RG_STAGES keeps the ring depth tunable.
template <typename TA, int BN>
static void launch_rungemm_dma(const RunGemm& d, const ArenaBases& ab, hipStream_t st, int grid) {
  static const int stages = env_stages("RG_STAGES", 2);
  if (d.flags & kRunBnBwd) { hipLaunchKernelGGL((rungemm_kernel<TA, BN, true, 2, kBM, 4, false, true>), dim3(grid), dim3(256), 0, st, d, ab); return; }
  if (stages == 2) hipLaunchKernelGGL((rungemm_kernel<TA, BN, true, 2>), dim3(grid), dim3(256), 0, st, d, ab);
  else if (stages == 3) hipLaunchKernelGGL((rungemm_kernel<TA, BN, true, 3>), dim3(grid), dim3(256), 0, st, d, ab);
  else hipLaunchKernelGGL((rungemm_kernel<TA, BN, true, 4>), dim3(grid), dim3(256), 0, st, d, ab);
}

template <typename TA>
static void launch_rungemm_t(const RunGemm& d, const ArenaBases& ab, hipStream_t st) {
  const int bn = bn_of(d.N);
  const int nm = (d.M + kBM - 1) / kBM;
  const int grid = nm * (d.Npad / bn);
  if (d.flags & kRunAligned) {
    if (bn == 128) launch_rungemm_dma<TA, 128>(d, ab, st, grid);
    else if (bn == 64) launch_rungemm_dma<TA, 64>(d, ab, st, grid);
    else launch_rungemm_dma<TA, 32>(d, ab, st, grid);
    return;
  }
  if (d.flags & kRunBnBwd) {
    if (bn == 128) hipLaunchKernelGGL((rungemm_kernel<TA, 128, false, 2, kBM, 4, false, true>), dim3(grid), dim3(256), 0, st, d, ab);
    else if (bn == 64) hipLaunchKernelGGL((rungemm_kernel<TA, 64, false, 2, kBM, 4, false, true>), dim3(grid), dim3(256), 0, st, d, ab);
    else hipLaunchKernelGGL((rungemm_kernel<TA, 32, false, 2, kBM, 4, false, true>), dim3(grid), dim3(256), 0, st, d, ab);
    return;
  }
  if (bn == 128) hipLaunchKernelGGL((rungemm_kernel<TA, 128, false>), dim3(grid), dim3(256), 0, st, d, ab);
  else if (bn == 64) hipLaunchKernelGGL((rungemm_kernel<TA, 64, false>), dim3(grid), dim3(256), 0, st, d, ab);
  else hipLaunchKernelGGL((rungemm_kernel<TA, 32, false>), dim3(grid), dim3(256), 0, st, d, ab);
}

void launch_rungemm(const RunGemm& d, const ArenaBases& ab, hipStream_t st) {
  if (launch_enc0_fwd(d, ab, st)) return;                  // first encoder layer of the bf16 plans, on the fp32 spectrum (enc0.hip)
  if (launch_cgemm256(d, ab, st)) return;                  // wide-tile kernel for the N >= 128 bf16 layers (cgemm256.hip)
  if (launch_rundirect(d, ab, st)) return;                 // direct-operand kernel for the thin bf16 layers (thin.hip)
  if (d.xdt == DT_BF16) launch_rungemm_t<bf16_t>(d, ab, st);
  else launch_rungemm_t<float>(d, ab, st);
}

// 256 x 256 tile, 8 waves, 4-stage ring of 32 KB = 128 KB: one workgroup per CU (wgrad_tn == 256)
static void launch_wgrad_wide(const RunGemm& d, const ArenaBases& ab, hipStream_t st) {
  static const int stages = env_stages("WG256_STAGES", 2);   // 2, 3 and 4 stages measure the same (+-1 %): 2 x 32 KB leaves LDS to the other stream's kernels
  static const bool dual = !(tune_str("WG_DUAL") && atoi(tune_str("WG_DUAL")) == 0);
  if (d.flags & kRunOnesMfma) {                     // the ones run is not a k tile (plan.cpp wgrad())
    dim3 grid1(((d.Npad + 255) / 256) * ((d.ldw - 64 + 255) / 256) * d.nsplit);
    if (dual) hipLaunchKernelGGL((wgrad_bf16_dma_kernel<256, 4, 256, 512, true, true>), grid1, dim3(512), 0, st, d, ab);
    else hipLaunchKernelGGL((wgrad_bf16_dma_kernel<256, 2, 256, 512, false, true>), grid1, dim3(512), 0, st, d, ab);
    return;
  }
  dim3 grid(((d.Npad + 255) / 256) * ((d.ldw + 255) / 256) * d.nsplit);
  if (dual) { hipLaunchKernelGGL((wgrad_bf16_dma_kernel<256, 4, 256, 512, true>), grid, dim3(512), 0, st, d, ab); return; }
  if (stages == 2) hipLaunchKernelGGL((wgrad_bf16_dma_kernel<256, 2, 256, 512>), grid, dim3(512), 0, st, d, ab);
  else if (stages == 3) hipLaunchKernelGGL((wgrad_bf16_dma_kernel<256, 3, 256, 512>), grid, dim3(512), 0, st, d, ab);
  else hipLaunchKernelGGL((wgrad_bf16_dma_kernel<256, 4, 256, 512>), grid, dim3(512), 0, st, d, ab);
}

// 128 x 512 tile for the N = 128 layers (kRunWgWide with Npad == 128): same bytes per MFMA as 256 x 256 would need at N = 256
static void launch_wgrad_wide128(const RunGemm& d, const ArenaBases& ab, hipStream_t st) {
  dim3 grid(((d.Npad + 127) / 128) * ((d.ldw + 511) / 512) * d.nsplit);
  static const bool dual = tune_str("WG_DUAL") && atoi(tune_str("WG_DUAL")) == 2;   // measured no gain at N = 128 (783 vs 774 TFLOP/s) for 160 KB of LDS: opt-in
  if (dual) { hipLaunchKernelGGL((wgrad_bf16_dma_kernel<128, 4, 512, 512, true>), grid, dim3(512), 0, st, d, ab); return; }
  hipLaunchKernelGGL((wgrad_bf16_dma_kernel<128, 2, 512, 512>), grid, dim3(512), 0, st, d, ab);
}

template <int TN>
static void launch_wgrad_dma(const RunGemm& d, const ArenaBases& ab, hipStream_t st) {
  static const int stages = env_stages("WG_STAGES", 3);   // 3 stages = 48 KiB (128-wide) / 36 KiB: co-resides better with the other stream (13.63 -> 13.24 ms per step vs 4)
  dim3 grid(((d.Npad + TN - 1) / TN) * ((d.ldw + kWgTK - 1) / kWgTK) * d.nsplit);
  if (stages == 2) hipLaunchKernelGGL((wgrad_bf16_dma_kernel<TN, 2>), grid, dim3(256), 0, st, d, ab);
  else if (stages == 3) hipLaunchKernelGGL((wgrad_bf16_dma_kernel<TN, 3>), grid, dim3(256), 0, st, d, ab);
  else hipLaunchKernelGGL((wgrad_bf16_dma_kernel<TN, 4>), grid, dim3(256), 0, st, d, ab);
}

void launch_wgrad(const RunGemm& d0, const ArenaBases& ab, hipStream_t st) {
  if (launch_enc0_wgrad(d0, ab, st)) return;               // ... and its weight gradient
  if (launch_wgrad_rank(d0, ab, st)) return;               // N <= 4 outputs over a contiguous array
  RunGemm d = d0;
#ifdef SEFD_TUNING
  // wrong-result arm (no partial-sum stores), tuning builds only (-DSEFD_TUNING): the product library has no switch that changes what a launch computes
  static const int wdbg = tune_str("WG_DBG") ? atoi(tune_str("WG_DBG")) : 0;
  if (wdbg & 8) d.flags |= 1 << 30;
#endif
  if (d.xdt == DT_BF16 && (d.flags & kRunAligned)) {
    if (d.flags & kRunWgWide) { if (d.Npad % 256 == 0) launch_wgrad_wide(d, ab, st); else launch_wgrad_wide128(d, ab, st); return; }
    switch (wgrad_tn(d.xdt, d.N, d.Npad)) {        // sefd_desc.h: the planner sized nsplit for the same tile
      case 128: launch_wgrad_dma<128>(d, ab, st); break;
      case 64: launch_wgrad_dma<64>(d, ab, st); break;
      case 32: launch_wgrad_dma<32>(d, ab, st); break;
      default: launch_wgrad_dma<16>(d, ab, st); break;
    }
    return;
  }
  if (d.xdt == DT_BF16) {
    if (d.Npad >= 128) {
      dim3 grid(((d.Npad + 127) / 128) * ((d.ldw + kWgTK - 1) / kWgTK) * d.nsplit);
      hipLaunchKernelGGL((wgrad_bf16_kernel<128>), grid, dim3(256), 0, st, d, ab);
    } else {
      dim3 grid(((d.Npad + 63) / 64) * ((d.ldw + kWgTK - 1) / kWgTK) * d.nsplit);
      hipLaunchKernelGGL((wgrad_bf16_kernel<64>), grid, dim3(256), 0, st, d, ab);
    }
    return;
  }
  dim3 grid(((d.Npad + kWgTN - 1) / kWgTN) * ((d.ldw + kWgTK - 1) / kWgTK) * d.nsplit);
  hipLaunchKernelGGL((wgrad_kernel<float>), grid, dim3(256), 0, st, d, ab);
}

}  // namespace sefd
