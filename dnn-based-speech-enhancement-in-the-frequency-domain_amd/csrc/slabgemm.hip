// RUNGEMM for the (t, f) convolutions, LDS-resident input slab (descriptor flag kRunSlab, fields slab_*; bf16).
//
// What bounds rungemm.hip / cgemm256.hip on these layers (profiles/r05_tuning_notes.md): both stream the TAP-EXPANDED activation operand through
// LDS-DMA - every input element enters a CU 2 * ntap / S = 10 (strided conv), 6 or 4 (sub-pixel phases) times per 256-row tile - and the
// LDS-DMA path of a CU sustains ~16 B/clk next to the MFMAs: a 128 x 128 tile needs 32 KB per 64-deep K tile (0.21 of the matrix peak), a
// 256 x 256 tile 64 KB per 2048 MFMA cycles.  With three of the four activation instructions of cgemm256 switched off (SEFD_CG256_DBG=64: the
// traffic of this kernel) its launches take 17 % less.
//
// This kernel keeps the input of a 256-row tile in LDS ONCE per 32-channel chunk - the slab: (frames of the tile + 1) x (positions a frame's
// rows touch) x 64 bytes - and forms the MFMA A fragments of every (time tap, frequency tap) from it: the fragment row of output (frame fl,
// fo) for tap (dt, df) is the slab row of position fo * S + df of frame fl + dt.  K order = [source][chunk][run][tap][32 channels] (the packed
// weights are permuted to match, sefd_desc.h w_index_g); a 32-deep weight sub-tile (one tap of one chunk) is one contiguous block, 4 sub-slots.
//   * tile 256 x BN (BN = 256: 8 waves x (128 x 64); BN = 128: 8 waves x (64 x 64)), persistent over output tiles;
//   * ONE workgroup barrier per PAIR of taps (64 deep).  During pair P a thread issues the weights of pair P + 1 (steps 0-1) and its share of
//     the NEXT chunk's slab (steps 2-3 of every pair but a chunk's last) into the other slab buffer; `s_waitcnt vmcnt(n)` at the top of a pair
//     leaves only those slab pieces in flight.  The stream of chunks runs across output tiles: the first slab and weights of the next tile are in
//     flight during the epilogue (not in the kRunBnBwd instantiation, whose epilogue stages through LDS);
//   * frame slots: consecutive frames of the tile, plus ONE all-zero slot wherever the tile crosses from batch item b to b + 1 - it is frame
//     "T" of b and frame "-1" of b + 1 at once, so no fragment read needs a per-row validity test; padded positions and frames outside the
//     input are filled from the zero page by the DMA;
//   * 64-byte slab rows, 16-byte unit u of row R holds channel octet u ^ ((R >> swz) & 3); the planner picks the slot order (parity split for
//     the stride-2 layers), the padding and swz so that every fragment read is free of bank conflicts (plan.cpp slab_layout).
// Same epilogue contract as cgemm256_kernel (bias, ReLU, BatchNorm partial sums per 128 rows, kRunBnBwd) plus the two-destination form (n2).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "sefd_desc.h"
#include "dev_common.h"

namespace sefd {

namespace {

__device__ __forceinline__ int sdiv(int x, uint32_t m, uint32_t s) { return m ? (int)(__umulhi((uint32_t)x, m) >> s) : x; }

__device__ __forceinline__ int xcd_remap3(int bid, int nwg) {
  const int xcd = bid & 7, local = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + local;
}

__device__ __forceinline__ void wgb() { asm volatile("s_barrier" ::: "memory"); }

__device__ __forceinline__ void wait_vm_n(int n) {           // n is wave-uniform, 0 .. 6
  if (n < 2) { if (n < 1) wait_vm<0>(); else wait_vm<1>(); }
  else if (n < 4) { if (n < 3) wait_vm<2>(); else wait_vm<3>(); }
  else if (n < 5) wait_vm<4>();
  else if (n < 6) wait_vm<5>();
  else wait_vm<6>();
}

}  // namespace

// dbg (tuning runs only, SEFD_SLAB_DBG): 1 skip the MFMAs, 2 skip the weight DMAs, 4 skip the fragment reads, 8 skip the slab DMAs, 16 no fragment
// address arithmetic (every tap reads the rows of tap 0), 64 no epilogue
// NBS: weight sub-slots.  4: the weights of pair P + 1 are issued during pair P.  6: those of pair P + 2 - the barrier at the top of pair P then
// publishes pair P + 1 as well, and the first fragments of pair P + 1 are read in front of its barrier (no read bubble behind every barrier).
template <int BN, int WM, int WN, bool BNB, int NBS, int dbg = 0>
__global__ __launch_bounds__(512) void slabgemm_kernel(const RunGemm d, const ArenaBases ab) {
  constexpr int BM = 256, NW = 8;
  constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 32, NI = WTN / 32;
  constexpr int SLAB = kSlabMaxRows * 64;                    // bytes of one slab buffer
  constexpr int B_SUB = BN * 64;                             // one 32-deep weight sub-tile
  constexpr int B_BASE = 2 * SLAB;
  constexpr int NBH = BN / 16 / NW;                          // weight DMAs per thread per sub-tile (16 rows x 64 B each)
  constexpr int NQ = (kSlabMaxRows / 16 + NW - 1) / NW;      // slab DMAs per thread per chunk, at most
  constexpr int NRING = NBS / 2, AHEAD = NRING - 1;           // pairs in the weight ring; pairs the weight stream runs ahead
  constexpr bool PF = NBS == 6;                              // cross-barrier prefetch of a pair's first fragments
  constexpr int STAT_BASE = B_BASE + NBS * B_SUB;
  constexpr int SMEM = STAT_BASE + (WM == 4 ? NW * WTN * 2 * 4 : 0);
  static_assert(WM * WN == NW && NBH >= 1 && SMEM <= 160 * 1024 && (SLAB % 2048) == 0 && (!BNB || WM == 2) && (NBS == 4 || NBS == 6), "geometry");
  __shared__ __attribute__((aligned(2048))) char smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = d.Npad / BN;
  const int nm = (d.M + BM - 1) / BM;
  const int total = nm * nn;
  const int Fo = d.Fo, fosh = __builtin_ctz(Fo), Tout = d.Tout;
  const int NF = BM >> fosh;                                 // frames per tile
  const int nbatch = d.M / (Tout * Fo);
  const int ntap = d.slab_ntap, NT = 2 * ntap;               // taps per chunk; pairs per chunk = ntap
  const int S = d.slab_S, NPp = d.slab_np, NPh = d.slab_nph, swz = d.slab_swz + 6;
  const int NP = (Fo - 1) * S + ntap;                        // positions a frame's rows touch
  // planner contract (slab_geometry): runs 0, 1 belong to source 0, runs 2, 3 (if any) to source 1
  const int nsrc = d.nseg == 4 ? 2 : 1;
  const int dtr00 = d.seg[0].dt - d.slab_dtmin, dtr01 = d.seg[1].dt - d.slab_dtmin;   // frame-slot offset (dt - dtmin) of run `rank` of source s
  const int dtr10 = nsrc == 2 ? d.seg[2].dt - d.slab_dtmin : 0, dtr11 = nsrc == 2 ? d.seg[3].dt - d.slab_dtmin : 0;
  const int nch0 = d.slab_C[0] >> 5, NCH = nch0 + (nsrc == 2 ? d.slab_C[1] >> 5 : 0);
  // slab issue slots of a chunk: the four steps of every pair but the last two, steps 0-1 of the pair before the last - at the top of a chunk's last
  // pair `vmcnt(0)` then has nothing younger than two steps to wait for, and the barrier there publishes the next chunk's slab whole
  // (dbg 32: all four steps of every pair but the last two + steps 0-1 of the pair before the last - what the 6-sub-slot ring needs so that the
  // barrier in front of a chunk's last pair publishes the next slab whole; measured slower than the default: steps 2-3 of every pair but the last)
  const int nslots = (dbg & 32) ? 4 * (ntap - 2) + 2 : 2 * (ntap - 1);

  const uint16_t* x0 = reinterpret_cast<const uint16_t*>(rp(ab, d.x[0]));
  const uint16_t* x1 = nsrc == 2 ? reinterpret_cast<const uint16_t*>(rp(ab, d.x[1])) : x0;
  const uint16_t* w = reinterpret_cast<const uint16_t*>(rp(ab, d.w));
  const uint16_t* zp = reinterpret_cast<const uint16_t*>(rp(ab, d.zero));
  const uint32_t lds0 = lds_addr(smem);

  // weight DMA role: instruction q covers rows (q * 8 + wid) * 16 .. + 16 of a 32-deep sub-tile; lane -> (row lb, unit pb)
  const int lb = lane >> 2, pb = lane & 3;
  const int csb = pb ^ ((lb >> 2) & 3);
  // fragment roles
  const int wm0 = (wid / WN) * WTM, wn0 = (wid % WN) * WTN;
  const int frow = lane & 31, fhalf = lane >> 5;
  const uint32_t boff0 = lds0 + B_BASE + (wn0 + frow) * 64 + ((fhalf ^ ((frow >> 2) & 3)) << 4);   // step 0 of a sub-tile; step 1: ^ 32
  const float* biasp = d.bias.arena >= 0 ? reinterpret_cast<const float*>(rp(ab, d.bias)) : nullptr;
  char* yb = rp(ab, d.y);
  const int n2 = d.n2 > 0 ? d.n2 : 0x7fffffff;               // columns >= n2 leave for the second destination
  char* yb2 = d.n2 > 0 ? rp(ab, d.y2) - (int64_t)d.n2 * 2 : yb;
  const bool want_stats = d.stats.arena >= 0;
  const uint16_t* ybn = BNB ? reinterpret_cast<const uint16_t*>(rp(ab, d.bnb_y)) : nullptr;
  const float bslope = BNB ? *reinterpret_cast<const float*>(rp(ab, d.bnb_slope)) : 0.f;

  // ---- stream state (carried across output tiles)
  int gr = 0;                                                // ring position gp % NRING of the pair being multiplied: weight sub-slots gr * 2 + h
  int gch = 0;                                               // chunks finished: slab buffer gch & 1
  // slab DMA cursor: the chunk being loaded is chunk d_ch of output tile d_t (-1: none), into buffer d_buf
  int d_t = -1, d_ch = 0, d_buf = 0, d_nq = 0, d_src = 0;
  int poff[NQ];                                              // element offset of this thread's 16 bytes of piece q in the source (chunk 0), -1: zero page
  // weight DMA cursor: next sub-tile pair to load
  int b_t = -1, b_st = 0;
  const uint16_t* b_ptr = w;
  int pend = 0;                                              // slab pieces this wave issued in the previous pair

  auto tile_m = [&](int t) { return xcd_remap3(t, total) / nn; };
  auto tile_n = [&](int t) { return xcd_remap3(t, total) % nn; };

  // piece geometry of (output tile t, source sc): which input position / frame each of this thread's slab rows holds
  auto setup_pieces = [&](int t, int sc) {
    const int g0 = (tile_m(t) * BM) >> fosh;
    const int b0 = g0 / Tout, ub0 = g0 - b0 * Tout;
    const int kmax = (ub0 + NF - 1) / Tout;
    const int NR = (NF + 1 + kmax) * NPp;
    const int nri = (NR + 15) >> 4;
    d_nq = nri > wid ? (nri - wid + NW - 1) / NW : 0;
    const int L0 = Tout - ub0 + 1;
    const int C = d.slab_C[sc], Fin = d.rowlen[sc] / C, Tin = d.Tin[sc];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int Rr = (q * NW + wid) * 16 + (lane >> 2);
      const int lc = (lane & 3) ^ (((lds0 + (uint32_t)Rr * 64) >> swz) & 3);
      const int fsl = sdiv(Rr, d.slab_div_m, d.slab_div_s), ps = Rr - fsl * NPp;
      int p = ps;
      if (NPh > 0) { const int par = ps >= NPh ? 1 : 0; p = 2 * (ps - par * NPh) + par; }
      const int pa = p + d.slab_p0;
      int k = 0, tt = ub0 + d.slab_dtmin + fsl;
      if (fsl >= L0) { const int s2 = fsl - L0; k = 1 + s2 / (Tout + 1); tt = d.slab_dtmin + (s2 - (k - 1) * (Tout + 1)); }
      const bool ok = Rr < NR && p < NP && pa >= 0 && pa < Fin && tt >= 0 && tt < Tin && b0 + k < nbatch;
      poff[q] = ok ? (int)((int64_t)(b0 + k) * d.bstride[sc] + (int64_t)tt * d.tstride[sc] + d.base[sc] + pa * C + lc * 8) : -1;
    }
    d_src = sc;
  };
  auto issue_piece = [&](int q) {                            // piece q of the chunk under the cursor
    const int cc = d_ch - (d_src ? nch0 : 0);
    const uint16_t* src = poff[q] >= 0 ? (d_src ? x1 : x0) + poff[q] + cc * 32 : zp;
    if (!(dbg & 8)) dma16(src, lds0 + d_buf * SLAB + (q * NW + wid) * 1024);
  };
  // pieces of issue slot `slot` (q mod nslots == slot); returns how many this wave issued
  auto issue_slot = [&](int slot) {
    int n = 0;
    if (d_t < 0) return 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int qs = nslots >= NQ ? q : (nslots == 4 ? (q & 3) : (q & 1));   // (ntap 2: two slots)
      if (qs == slot && q < d_nq) { issue_piece(q); ++n; }
    }
    return n;
  };
  auto issue_all_pieces = [&]() {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (q < d_nq) issue_piece(q);
  };
  // move the slab cursor to the chunk after (t, ch): next chunk of the tile, or chunk 0 of the next tile (not BNB), or none
  auto advance_slab = [&](int t, int ch) {
    int nt = t, nc = ch + 1;
    if (nc == NCH) { nc = 0; nt = (!BNB && t + (int)gridDim.x < total) ? t + (int)gridDim.x : -1; }
    d_buf ^= 1;
    d_t = nt; d_ch = nc;
    if (nt >= 0 && (nc == 0 || nc == nch0)) setup_pieces(nt, nc >= nch0 ? 1 : 0);
  };
  const int64_t wstep = (int64_t)d.Npad * 32;                // elements per 32-deep sub-tile
  auto b_set = [&](int t) {                                  // weight cursor to sub-tile 0 of tile t
    b_t = t; b_st = 0;
    if (t >= 0) b_ptr = w + ((int64_t)tile_n(t) * BN + wid * 16 + lb) * 32 + csb * 8;
  };
  // half h of the weights of the pair under the cursor into sub-slot (par * 2 + h); the cursor advances with the second half
  auto issue_b = [&](int ring, int h) {
    if (b_t >= 0) {
      const uint32_t B = lds0 + B_BASE + (ring * 2 + h) * B_SUB;
#pragma unroll
      for (int q = 0; q < NBH; ++q) if (!(dbg & 2)) dma16(b_ptr + q * (NW * 16 * 32), B + (q * NW + wid) * 1024);
      b_ptr += wstep;
    }
    if (h == 1 && b_t >= 0) {
      b_st += 2;
      if (b_st == NCH * NT) b_set((!BNB && b_t + (int)gridDim.x < total) ? b_t + (int)gridDim.x : -1);
    }
  };
  // everything the first pair of tile t needs, issued at once (kernel start; every tile of the BNB instantiation)
  auto prologue = [&](int t) {
    d_t = t; d_ch = 0; d_buf = gch & 1;
    setup_pieces(t, 0);
    issue_all_pieces();
    b_set(t);
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) {
      const int ring = gr + a >= NRING ? gr + a - NRING : gr + a;
      issue_b(ring, 0); issue_b(ring, 1);
    }
    pend = 0;
  };

  if ((int)blockIdx.x < total) prologue(blockIdx.x);
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    const int mtile = tile_m(t), ntile = tile_n(t);
    float bv[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = ntile * BN + wn0 + j * 32 + (lane & 31);
      bv[j] = (biasp && n < d.N) ? biasp[n] : 0.f;
    }
    // slab row of this lane's fragment rows for tap (dt = dtmin, df = 0), byte offset inside a slab buffer
    int Rb[MI];
    {
      const int g0 = (mtile * BM) >> fosh;
      const int ub0 = g0 - (g0 / Tout) * Tout;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int tr = wm0 + i * 32 + frow;
        const int fl = tr >> fosh, fo = tr & (Fo - 1);
        const int k = (ub0 + fl) / Tout;
        Rb[i] = ((fl + k) * NPp + (NPh > 0 ? fo : fo * S)) * 64;
      }
    }
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- the tile's pairs of taps, one flat loop: chunk ch, pair pc of the chunk, (rk, df) = run rank and frequency tap of the pair's first tap
    const int NPT = NCH * ntap;
    int ch = 0, pc = 0, rk = 0, df = 0;
    uint32_t sbuf = lds0 + (gch & 1) * SLAB;
    uint32_t sb0 = sbuf + dtr00 * NPp * 64, sb1 = sbuf + dtr01 * NPp * 64;      // slab row 0 of the frame slots of the chunk's two runs
    auto tap_base = [&]() { return (rk ? sb1 : sb0) + (NPh > 0 ? (df & 1) * NPh + (df >> 1) : df) * 64; };
    auto tap_next = [&]() { if (++df == ntap) { df = 0; rk = 1; } };
    // fragment address of a tap, step 0 (step 1: ^ 32): unit = ((row >> swz) & 3) ^ octet, octet = 2 * step + half
    uint32_t aA[MI], aB[MI];
#define SLAB_ADDR(AX)                                                                                        \
  {                                                                                                          \
    const uint32_t tb_ = (dbg & 16) ? sbuf : tap_base();                                                     \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                         \
      const uint32_t r_ = tb_ + Rb[i];                                                                       \
      AX[i] = r_ + ((((r_ >> swz) & 3) ^ (uint32_t)fhalf) << 4) - lds0;                                      \
    }                                                                                                        \
  }
#define SLAB_READ(AF, BF, AA, BB, X)                                                                         \
  if (!(dbg & 4)) {                                                                                          \
  _Pragma("unroll") for (int i = 0; i < MI; ++i) AF[i] = *reinterpret_cast<const uint4*>(smem + (AA[i] ^ (X))); \
  _Pragma("unroll") for (int j = 0; j < NI; ++j) BF[j] = *reinterpret_cast<const uint4*>(smem + ((BB) ^ (X)) + j * (32 * 64)); }
#define SLAB_MFMA(AF, BF, DMA)                                                                               \
  _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                           \
    _Pragma("unroll") for (int j = 0; j < NI; ++j) if (!(dbg & 1))                                           \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, AF[i]), __builtin_bit_cast(bf16x8, BF[j]), acc[i][j], 0, 0, 0); \
    if (i == MI / 2 - 1) { __builtin_amdgcn_sched_barrier(0); DMA; __builtin_amdgcn_sched_barrier(0); }      \
  }
    SLAB_ADDR(aA)
    uint4 af0[MI], bf0[NI], af1[MI], bf1[NI];
    if (dbg & 4) {
#pragma unroll
      for (int i = 0; i < MI; ++i) af0[i] = af1[i] = make_uint4(aA[i], Rb[i], 0, 0);
#pragma unroll
      for (int j = 0; j < NI; ++j) bf0[j] = bf1[j] = make_uint4(boff0, lane, 0, 0);
    }
    bool pf = false;                                         // set 0 already holds step 0 of this pair (read in front of the barrier)
    for (int pp = 0; pp < NPT; ++pp) {
      if (pc == 0) advance_slab(t, ch);                      // the slab cursor moves to the chunk loaded during this one
      tap_next();
      SLAB_ADDR(aB)
      tap_next();
      const uint32_t bA = boff0 - lds0 + (gr * 2) * B_SUB, bB = bA + B_SUB;
      const int grn = gr + 1 == NRING ? 0 : gr + 1;                    // ring position of the next pair
      const int gri = gr + AHEAD >= NRING ? gr + AHEAD - NRING : gr + AHEAD;   // ... of the pair whose weights are issued now
      wait_vm_n(pend);                                       // the newest weights (and, in front of a chunk's last pair, its successor's slab) have landed - this thread's part
      wgb();                                                 // ... everyone's; and every wave is past the previous pair: its sub-slots / the other slab buffer may be refilled
      pend = 0;
      if (!PF || !pf) { SLAB_READ(af0, bf0, aA, bA, 0u) }
      SLAB_READ(af1, bf1, aA, bA, 32u)
      const int slot = (dbg & 32) ? 4 * pc : 2 * pc - 2;
      const bool more = (dbg & 32) ? pc + 2 < ntap : pc + 1 < ntap;                       // steps 2-3 carry slab pieces
      SLAB_MFMA(af0, bf0, issue_b(gri, 0); if ((dbg & 32) && pc + 1 < ntap) issue_slot(slot))
      SLAB_READ(af0, bf0, aB, bB, 0u)
      SLAB_MFMA(af1, bf1, issue_b(gri, 1); if ((dbg & 32) && pc + 1 < ntap) issue_slot(slot + 1))
      SLAB_READ(af1, bf1, aB, bB, 32u)
      SLAB_MFMA(af0, bf0, if (more) pend += issue_slot(slot + 2))
      // on to the next pair: its chunk's slab buffer / run bases, its first tap's addresses - and with the deep ring its first fragments
      if (++pc == ntap) {
        pc = 0; ++ch; ++gch; rk = 0; df = 0;
        sbuf = lds0 + (gch & 1) * SLAB;
        const bool s1 = ch >= nch0;
        sb0 = sbuf + (s1 ? dtr10 : dtr00) * NPp * 64; sb1 = sbuf + (s1 ? dtr11 : dtr01) * NPp * 64;
      }
      pf = false;
      if (pp + 1 < NPT) {
        SLAB_ADDR(aA)
        if (PF) { SLAB_READ(af0, bf0, aA, (boff0 - lds0 + (grn * 2) * B_SUB), 0u) pf = true; }
      }
      SLAB_MFMA(af1, bf1, if (more) pend += issue_slot(slot + 3))
      gr = grn;
    }
#undef SLAB_ADDR
#undef SLAB_READ
#undef SLAB_MFMA
    if (BNB) wgb();                                          // the kRunBnBwd epilogue writes LDS: every wave must be past the last pair

    if constexpr (BNB) {
      // ---- kRunBnBwd epilogue (as cgemm256_kernel<true>): the wave's 128 x 64 tile goes to 16 KB of LDS as bf16, then 16-byte row chunks leave
      // LDS -> global beside the same chunk of the BatchNorm layer's forward output, whose three backward sums are accumulated per column
      const int TF = Tout * Fo;
      char* wt = smem + wid * 16384;
      const QuadT qt(lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int col = j * 32 + (lane & 28);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = i * 32 + 8 * q + 4 * (lane >> 5) + (lane & 3);
            const uint2 pk = qt.pack(acc[i][j][4 * q] + bv[j], acc[i][j][4 * q + 1] + bv[j], acc[i][j][4 * q + 2] + bv[j], acc[i][j][4 * q + 3] + bv[j]);
            *reinterpret_cast<uint2*>(wt + row * 128 + (((col >> 3) ^ (row & 7)) << 4) + (col & 7) * 2) = pk;
          }
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int chn = lane & 7, n0 = ntile * BN + wn0 + chn * 8;
      const bool cok = n0 < d.N;
      float pm[8], pis[8], pg[8], pbt[8], t0[8], t1[8], t2[8];
      {
        const float* mi = reinterpret_cast<const float*>(rp(ab, d.bnb_mi));
        const float* ga = reinterpret_cast<const float*>(rp(ab, d.bnb_gamma));
        const float* be = reinterpret_cast<const float*>(rp(ab, d.bnb_beta));
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int n = cok ? n0 + e : 0;
          pm[e] = mi[n]; pis[e] = mi[d.N + n]; pg[e] = ga[n]; pbt[e] = be[n];
          t0[e] = t1[e] = t2[e] = 0.f;
        }
      }
      for (int k8 = 0; k8 < 16; k8 += 8) {
        uint4 ypre[8];
        int64_t oo[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const int m = mtile * BM + wm0 + (lane >> 3) + 8 * (k8 + kk);
          oo[kk] = -1;
          ypre[kk] = make_uint4(0, 0, 0, 0);
          if (m < d.M && cok) {
            const int b = sdiv(m, d.div_tf_m, d.div_tf_s), rem = m - b * TF, u = sdiv(rem, d.div_fo_m, d.div_fo_s), fo = rem - u * Fo;
            oo[kk] = (int64_t)b * d.y_bstride + (int64_t)u * d.y_tstride + (int64_t)fo * d.y_fstride + d.y_off;
            ypre[kk] = *reinterpret_cast<const uint4*>(ybn + (int64_t)b * d.bnb_bstride + (int64_t)u * d.bnb_tstride + (int64_t)fo * d.bnb_fstride + d.bnb_off + n0);
          }
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const int row = (lane >> 3) + 8 * (k8 + kk);
          const int64_t o = oo[kk];
          if (o < 0) continue;
          const uint4 dzv = *reinterpret_cast<const uint4*>(wt + row * 128 + ((chn ^ (row & 7)) << 4));
          const uint4 yv = ypre[kk];
          *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(n0 >= n2 ? yb2 : yb) + o + n0) = dzv;
          const uint32_t dw[4] = {dzv.x, dzv.y, dzv.z, dzv.w}, yw[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float dz = bf2f((uint16_t)(dw[e >> 1] >> (16 * (e & 1)))), yy = bf2f((uint16_t)(yw[e >> 1] >> (16 * (e & 1))));
            const float xh = (yy - pm[e]) * pis[e];
            const float bn = pg[e] * xh + pbt[e];
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
      const int srow = mtile * 2 + wid / WN;
      if (lane < 8 && srow < (d.M + kBM - 1) / kBM) {
        float* part = reinterpret_cast<float*>(rp(ab, d.stats));
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          part[((int64_t)srow * 3 + 0) * d.Npad + n0 + e] = t0[e];
          part[((int64_t)srow * 3 + 1) * d.Npad + n0 + e] = t1[e];
          part[((int64_t)srow * 3 + 2) * d.Npad + n0 + e] = t2[e];
        }
      }
      if (t + (int)gridDim.x < total) {
        wgb();                                               // every wave has left its 16 KB before the next tile's DMAs land in LDS
        prologue(t + gridDim.x);
      }
      continue;
    } else if (dbg & 64) {                                   // (tuning: no epilogue at all)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(acc[i][j][e]));      // every accumulator element stays live: nothing upstream is dead code
    } else {
      // ---- epilogue, wave local, no LDS for the tile: bias / ReLU / statistics on the accumulators as they sit, quad transpose (dev_common.h
      // QuadT), 8-byte row-piece stores.  The next tile's first slab and weights are in flight meanwhile.
      const int TF = Tout * Fo;
      const QuadT qt(lane);
      float s1[NI], s2[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row0 = mtile * BM + wm0 + i * 32;
        int64_t ro[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = row0 + 8 * q + 4 * (lane >> 5) + (lane & 3);
          ro[q] = -1;
          if (m < d.M) {
            const int b = sdiv(m, d.div_tf_m, d.div_tf_s), rem = m - b * TF, u = sdiv(rem, d.div_fo_m, d.div_fo_s), fo = rem - u * Fo;
            ro[q] = (int64_t)b * d.y_bstride + (int64_t)u * d.y_tstride + (int64_t)fo * d.y_fstride + d.y_off;
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
            const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            v[e] = acc[i][j][e] + bv[j];
            if (d.flags & kRunRelu) v[e] = fmaxf(v[e], 0.f);
            if (row0 + row < d.M && nok) { s1[j] += v[e]; s2[j] += v[e] * v[e]; }
          }
          uint16_t* dst = reinterpret_cast<uint16_t*>(n0 >= n2 ? yb2 : yb);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint2 pk = qt.pack(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            if (ro[q] >= 0 && n0 < d.N) *reinterpret_cast<uint2*>(dst + ro[q] + n0) = pk;
          }
        }
      }
      if (want_stats) {
        float* part = reinterpret_cast<float*>(rp(ab, d.stats));
        const int nrows = (d.M + kBM - 1) / kBM;
        if constexpr (WM == 2) {                             // this wave's 128 rows are ONE 128-row statistics block of its columns
          const int srow = mtile * 2 + wid / WN;
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const float t1 = s1[j] + __shfl_xor(s1[j], 32), t2 = s2[j] + __shfl_xor(s2[j], 32);
            const int n = ntile * BN + wn0 + j * 32 + (lane & 31);
            if (lane < 32 && srow < nrows) {
              part[((int64_t)srow * 2 + 0) * d.Npad + n] = t1;
              part[((int64_t)srow * 2 + 1) * d.Npad + n] = t2;
            }
          }
        } else {                                             // 64-row wave tiles: the odd wave row hands its sums to the even one through LDS
          float* sx = reinterpret_cast<float*>(smem + STAT_BASE);
          const int wmi = wid / WN;
          float t1[NI], t2[NI];
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            t1[j] = s1[j] + __shfl_xor(s1[j], 32); t2[j] = s2[j] + __shfl_xor(s2[j], 32);
            if ((wmi & 1) && lane < 32) { sx[(wid * WTN + j * 32 + lane) * 2] = t1[j]; sx[(wid * WTN + j * 32 + lane) * 2 + 1] = t2[j]; }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          wgb();
          const int srow = mtile * 2 + (wmi >> 1);
          if (!(wmi & 1) && lane < 32 && srow < nrows) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
              const int n = ntile * BN + wn0 + j * 32 + lane;
              part[((int64_t)srow * 2 + 0) * d.Npad + n] = t1[j] + sx[((wid + WN) * WTN + j * 32 + lane) * 2];
              part[((int64_t)srow * 2 + 1) * d.Npad + n] = t2[j] + sx[((wid + WN) * WTN + j * 32 + lane) * 2 + 1];
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          wgb();                                             // the hand-over region is free again before the next tile's epilogue
        }
      }
    }
  }
}

// The planner decides which GEMMs take this kernel: it marks them (and permutes their packed weights) with kRunSlab.
bool launch_slabgemm(const RunGemm& d, const ArenaBases& ab, hipStream_t st) {
  if (!(d.flags & kRunSlab)) return false;
  static const int ncu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
  const int bn = d.Npad % 256 == 0 ? 256 : 128;
  const int total = ((d.M + 255) / 256) * (d.Npad / bn);
  const dim3 grid(total < ncu ? total : ncu);
  static const int dbg = getenv("SEFD_SLAB_DBG") ? atoi(getenv("SEFD_SLAB_DBG")) : 0;
  // 6 sub-slots + reading a pair's first fragments in front of its barrier is only correct with the early slab schedule (SEFD_SLAB_DBG=32)
  static const int nbs = (getenv("SEFD_SLAB_NBS") && atoi(getenv("SEFD_SLAB_NBS")) == 6 && dbg == 32) ? 6 : 4;
#define SEFD_SLAB_LAUNCH(DBG)                                                                                                \
  do {                                                                                                                       \
    if (bn == 256) {                                                                                                         \
      if (d.flags & kRunBnBwd) hipLaunchKernelGGL((slabgemm_kernel<256, 2, 4, true, 4, DBG>), grid, dim3(512), 0, st, d, ab);   \
      else hipLaunchKernelGGL((slabgemm_kernel<256, 2, 4, false, 4, DBG>), grid, dim3(512), 0, st, d, ab);                      \
    } else {                                                                                                                 \
      if (nbs == 4) hipLaunchKernelGGL((slabgemm_kernel<128, 4, 2, false, 4, DBG>), grid, dim3(512), 0, st, d, ab);             \
      else hipLaunchKernelGGL((slabgemm_kernel<128, 4, 2, false, 6, DBG>), grid, dim3(512), 0, st, d, ab);                      \
    }                                                                                                                        \
  } while (0)
  switch (dbg) {
    case 1: SEFD_SLAB_LAUNCH(1); break;
    case 2: SEFD_SLAB_LAUNCH(2); break;
    case 4: SEFD_SLAB_LAUNCH(4); break;
    case 8: SEFD_SLAB_LAUNCH(8); break;
    case 10: SEFD_SLAB_LAUNCH(10); break;
    case 16: SEFD_SLAB_LAUNCH(16); break;
    case 32: SEFD_SLAB_LAUNCH(32); break;
    case 64: SEFD_SLAB_LAUNCH(64); break;
    case 74: SEFD_SLAB_LAUNCH(74); break;
    default: SEFD_SLAB_LAUNCH(0); break;
  }
#undef SEFD_SLAB_LAUNCH
  return true;
}

}  // namespace sefd
