// RUNGEMM, LDS-resident input-window variant for the thin conv layers with K of a few hundred (bf16, N <= 64, 64 or 128 output rows per
// frame: enc1, dec4, dec5 forward and the input-gradient GEMMs of dec4 / enc1).
//
// The tiled kernel (rungemm.hip) re-fetches the im2col-expanded A tile through LDS-DMA once per 64-deep K tile: the 3 (or 5) frequency taps
// of a run overlap from row to row, 96-long runs are padded to 128, and every K tile ends in a barrier with 4 MFMAs per wave behind it -
// dec4's forward phases stream 2.5 GB through LDS for 254 MB of activations and run at 2.5x their HBM floor.  Here
//   * a workgroup (4 waves) owns 128 consecutive output rows = one frame (Fo = 128) or two frames (Fo = 64), processed as two halves of 64;
//   * per half the WHOLE frame row of every (source, frame offset) pair a run refers to is DMA'd ONCE into an 8 KB LDS slot (whole
//     128-byte lines, 16-byte chunks XOR-swizzled on the source side so that the 32 rows of an MFMA fragment read 32 different slots);
//     the taps of all runs are then formed by shifted ds_read_b128 - no im2col, no padding, no barrier inside the K loop;
//   * the weights never touch LDS: the K range is split over wave pairs, every wave keeps the B fragments of its K half in REGISTERS for
//     the whole (persistent) kernel; the two partial 32 x 32 tiles of an output tile meet through 4 KB of LDS;
//   * N = 32: waves = (K half, 32-row block); N = 64: waves = (K half, 32-column tile), both row blocks per wave;
//   * epilogue by the owning wave: bias, BatchNorm partial sums (one row of partials per 128 rows, as everywhere), bf16 tile transposed
//     through 2 KB of wave-private LDS into 16-byte row-chunk stores.
// Same descriptor and results as rungemm_kernel (fp32 accumulation order over k differs).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "sefd_desc.h"
#include "dev_common.h"

namespace sefd {

namespace {

constexpr int kWcSlot = 8192;            // bytes per window slot: one frame row of one source (rowlen <= 4096 bf16)
constexpr int kWcMaxQ = 4;               // distinct (source, frame offset) pairs

struct WcGeom {
  int32_t nq, nsteps, spw;               // slots, 16-deep K steps in total, steps per K half
  int32_t qsrc[kWcMaxQ], qdt[kWcMaxQ];
  int32_t segq[kMaxSeg];                 // slot of every run
};

// NI = Npad / 32; SPW = K steps per wave (compile-time bound of the register-resident weight fragments)
template <int NI, int SPW>
__global__ __launch_bounds__(256) void winconv_kernel(const RunGemm d, const ArenaBases ab, const WcGeom gm, const int dbg) {
  // dbg (tuning runs, SEFD_WC_DBG): 1 no MFMA loop, 2 no window DMAs, 4 no reduction / epilogue
  constexpr int NRB = NI == 1 ? 1 : 2;                       // row blocks a wave multiplies per half
  constexpr int kWin = kWcMaxQ * kWcSlot + 256;              // one window buffer: the slots + 256 zero bytes (target of out-of-range taps)
  constexpr int kZero = kWcMaxQ * kWcSlot;                   // ... at this offset inside the buffer
  constexpr int NBUF = 3;                                    // window ring: frames i + 1, i + 2 in flight while frame i is multiplied
  constexpr int kScr = NBUF * kWin;                          // partial tiles of the non-owning waves: 4 x 4 KB
  constexpr int kStg = kScr + 4 * 4096;                      // wave-private 32 x 32 bf16 staging pieces: 4 x 2 KB
  constexpr int kStat = kStg + 4 * 2048;                     // [4 waves][32][2] floats
  constexpr int kSmem = kStat + 4 * 32 * 2 * 4;
  __shared__ __attribute__((aligned(16))) char smem[kSmem];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = wid >> 1, wsel = wid & 1;                   // K half; NI == 1: row block of the half, NI == 2: column tile
  const int frow = lane & 31, fh = lane >> 5;
  const int Fo = d.Fo, TF = d.Tout * Fo;
  const int nframes = d.M / Fo;
  const int ntiles = (d.M + 127) >> 7;
  const bool two_frames = Fo == 64;                          // a half = one frame (Fo = 64) or half a frame (Fo = 128)

  for (int i = tid; i < NBUF * 64; i += 256) reinterpret_cast<uint32_t*>(smem + (i >> 6) * kWin + kZero)[i & 63] = 0u;

  // ---- this wave's K steps: run, position; weight fragments in registers
  const uint16_t* w = reinterpret_cast<const uint16_t*>(rp(ab, d.w));
  int sq[SPW], se[SPW], sfs[SPW], srl[SPW];                  // slot, element offset in the frame row (off + j), row stride, row length
  bool sv[SPW];
  uint4 breg[SPW];                                           // one column tile per wave either way
  {
    int s0 = kh * gm.spw;
#pragma unroll
    for (int k = 0; k < SPW; ++k) {
      const int s = s0 + k;
      sv[k] = k < gm.spw && s < gm.nsteps;
      int seg = 0, j = 0, acc = 0;
      for (int g = 0; g < d.nseg; ++g) {
        const int ns = d.seg[g].len >> 4;
        if (s >= acc && s < acc + ns) { seg = g; j = (s - acc) << 4; }
        acc += ns;
      }
      const Seg sg = d.seg[seg];
      const int src = sg.src > 0 ? 1 : 0;
      sq[k] = gm.segq[seg]; se[k] = sg.off + j; sfs[k] = d.fstride[src]; srl[k] = d.rowlen[src];
      const int n = (NI == 1 ? 0 : wsel * 32) + frow;
      breg[k] = sv[k] ? *reinterpret_cast<const uint4*>(w + (int64_t)n * d.ldw + sg.koff + j + 8 * fh) : make_uint4(0, 0, 0, 0);
    }
  }
  // ---- LDS address of this lane's A fragment for (half, row block, step): tile independent
  constexpr int NH = NI == 1 ? 2 : 1;                        // (N = 64 runs with Fo = 64 only: both halves are whole frames, same table)
  uint32_t aaddr[NH][NRB][SPW];                              // byte offsets into smem
  const uint32_t lds0 = lds_addr(smem);
#pragma unroll
  for (int hf = 0; hf < NH; ++hf)
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int k = 0; k < SPW; ++k) {
        const int rblk = NI == 1 ? wsel : rb;
        const int fo = (two_frames ? 0 : hf * 64) + rblk * 32 + frow;
        const int e = se[k] + fo * sfs[k] + 8 * fh;
        const int p = e >> 3;
        const int mask = min(sfs[k] >> 3, 16) - 1;
        const int ps = p ^ ((p >> 4) & mask);
        aaddr[hf][rb][k] = (sv[k] && e >= 0 && e + 8 <= srl[k]) ? (uint32_t)(sq[k] * kWcSlot + ps * 16) : (uint32_t)kZero;   // relative to the frame's window buffer
      }
  const uint16_t* x0 = reinterpret_cast<const uint16_t*>(rp(ab, d.x[0]));
  const uint16_t* x1 = d.x[1].arena >= 0 ? reinterpret_cast<const uint16_t*>(rp(ab, d.x[1])) : x0;
  const uint16_t* zp = reinterpret_cast<const uint16_t*>(rp(ab, d.zero));
  const float* biasp = d.bias.arena >= 0 ? reinterpret_cast<const float*>(rp(ab, d.bias)) : nullptr;
  uint16_t* yb = reinterpret_cast<uint16_t*>(rp(ab, d.y));
  const bool want_stats = d.stats.arena >= 0;
  // owner of an output tile: the kh == 0 wave that multiplied its first K half (NI == 1: of that row block; NI == 2: of that column tile)
  const int ncol = (NI == 1 ? 0 : wsel * 32) + frow;         // this wave's column
  const float bv = (biasp && ncol < d.N) ? biasp[ncol] : 0.f;
  char* stg = smem + kStg + wid * 2048;
  const int tile0 = blockIdx.x;
  __syncthreads();

  // ---- frame pipeline: the window of frame i + 1 lands in the other buffer while frame i is multiplied
  const int fpt = two_frames ? 2 : 1;                        // frames per 128-row tile
  const int hpf = two_frames ? 1 : 2;                        // 64-row halves per frame
  const int my_tiles = tile0 < ntiles ? (ntiles - 1 - tile0) / (int)gridDim.x + 1 : 0;
  const int my_frames = my_tiles * fpt;
  auto frame_of = [&](int i, int& tile, int& g) { tile = tile0 + (i / fpt) * (int)gridDim.x; g = tile * fpt + (i % fpt); };
  auto issue = [&](int i) {                                  // every slot = one whole frame row, 512 chunks, 2 per thread
    int tile, g;
    frame_of(i, tile, g);
    const bool gvalid = g < nframes;
    const int gg = gvalid ? g : 0;
    const int b = gg / d.Tout, u = gg - b * d.Tout;
    if (kh == 0 || (dbg & 2)) return;                        // the two non-owning waves (they never store) move the windows: their vmcnt counts DMAs only
    const uint32_t wbase = lds0 + (i % NBUF) * kWin;
#pragma unroll
    for (int q = 0; q < kWcMaxQ; ++q) {
      if (q >= gm.nq) break;
      const int src = gm.qsrc[q];
      const int tt = u + gm.qdt[q];
      const bool ok = gvalid && tt >= 0 && tt < d.Tin[src];
      const uint16_t* fr = (src ? x1 : x0) + (int64_t)b * d.bstride[src] + d.base[src] + (int64_t)tt * d.tstride[src];
      const int mask = min(d.fstride[src] >> 3, 16) - 1;
      const int rl = d.rowlen[src];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ci = (c * 2 + wsel) * 64 + lane;                           // 512 chunks per slot, 128 threads
        const int sp = ci ^ ((ci >> 4) & mask);
        const uint16_t* sptr = (ok && sp * 8 + 8 <= rl) ? fr + sp * 8 : zp;   // invalid frames / chunks come from the zero page
        dma16(sptr, wbase + q * kWcSlot + (c * 2 + wsel) * 1024);
      }
    }
  };
  float s1 = 0.f, s2 = 0.f;
  if (my_frames > 0) issue(0);
  if (my_frames > 1) issue(1);
  const int dmas = 4 * gm.nq;                                // DMA instructions per window and loader thread
  for (int i = 0; i < my_frames; ++i) {
    int tile, g;
    frame_of(i, tile, g);
    const bool gvalid = g < nframes;
    const int gg = gvalid ? g : 0;
    const int b = gg / d.Tout, u = gg - b * d.Tout;          // frames per batch item = Tout (wave-uniform)
    if (kh == 1) {                                           // window i has landed: at most window i + 1 (issued one frame ago) still in flight
      if (i + 1 < my_frames) { if (dmas == 16) wait_vm<16>(); else if (dmas == 12) wait_vm<12>(); else if (dmas == 8) wait_vm<8>(); else wait_vm<4>(); }
      else wait_vm<0>();
    }
    lds_barrier();                                           // window i complete; everyone is done with frame i - 1 (its buffer is the one refilled next)
    if (i + 2 < my_frames) issue(i + 2);
    const char* win = smem + (i % NBUF) * kWin;
    for (int hh = 0; hh < hpf; ++hh) {
      if (!gvalid) break;                                    // (workgroup-uniform)
      if (hh > 0) lds_barrier();                             // the partial tiles of the first half have been read
      const int hf = two_frames ? (i % fpt) : hh;            // position of these 64 rows inside the 128-row tile
      // ---- this wave's K half
      f32x16 acc[NRB];
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[rb][e] = 0.f;
      const int hx = two_frames ? 0 : hh;
      // fragment reads run PF steps ahead of the MFMAs (a read followed by its own MFMA exposed the LDS latency on every step: 4 us per
      // frame); steps beyond this wave's share read the zero chunk against zero weights - no branch in the loop
      if (!(dbg & 1)) {
      constexpr int PF = 4;
      auto rd = [&](int k, int rb) { return *reinterpret_cast<const uint4*>(win + (NH == 2 && hx ? aaddr[NH - 1][rb][k] : aaddr[0][rb][k])); };
      uint4 abuf[PF][NRB];
#pragma unroll
      for (int k = 0; k < PF; ++k)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) abuf[k][rb] = rd(k, rb);
#pragma unroll
      for (int k = 0; k < SPW; ++k) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
          const uint4 a = abuf[k % PF][rb];
          if (k + PF < SPW) abuf[k % PF][rb] = rd(k + PF, rb);
          acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, breg[k]), acc[rb], 0, 0, 0);
        }
      }
      }
      if (dbg & 4) continue;
      // ---- partial tiles of the non-owners to LDS
      float* scr = reinterpret_cast<float*>(smem + kScr);
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        const bool mine = kh == 0;
        if (!mine) {
          const int slot = NI == 1 ? wsel : rb * 2 + wsel;   // output tile (row block, column tile)
          float4* dst = reinterpret_cast<float4*>(scr + slot * 1024 + lane * 16);
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) dst[e4] = make_float4(acc[rb][4 * e4], acc[rb][4 * e4 + 1], acc[rb][4 * e4 + 2], acc[rb][4 * e4 + 3]);
        }
      }
      lds_barrier();
      // ---- owners: add the partner's partial, epilogue
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        const bool mine = kh == 0;
        if (!mine) continue;
        const int slot = NI == 1 ? wsel : rb * 2 + wsel;
        const int rblk = NI == 1 ? wsel : rb;
        const float4* srcp = reinterpret_cast<const float4*>(scr + slot * 1024 + lane * 16);
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const float4 pv = srcp[e4];
          acc[rb][4 * e4] += pv.x; acc[rb][4 * e4 + 1] += pv.y; acc[rb][4 * e4 + 2] += pv.z; acc[rb][4 * e4 + 3] += pv.w;
        }
        const bool nok = ncol < d.N;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * fh;
          const float v = acc[rb][e] + bv;
          *reinterpret_cast<uint16_t*>(stg + row * 64 + (((frow >> 3) ^ ((row >> 1) & 3)) << 4) + (frow & 7) * 2) = f2bf(v);
          if (nok) { s1 += v; s2 += v * v; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const int c = lane + 64 * c2, row = c >> 2, ch = c & 3;
          const uint4 v = *reinterpret_cast<const uint4*>(stg + row * 64 + ((ch ^ ((row >> 1) & 3)) << 4));
          const int fo = (two_frames ? 0 : hh * 64) + rblk * 32 + row;
          const int n0 = (NI == 1 ? 0 : wsel * 32) + ch * 8;
          if (n0 < d.N) {
            const int64_t o = (int64_t)b * d.y_bstride + (int64_t)u * d.y_tstride + (int64_t)fo * d.y_fstride + d.y_off;
            *reinterpret_cast<uint4*>(yb + o + n0) = v;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      (void)hf;
    }
    // ---- end of a 128-row tile: BatchNorm partial sums; the two owner waves of a column tile meet through LDS
    if ((i % fpt) == fpt - 1) {
      if (want_stats) {
        float* st = reinterpret_cast<float*>(smem + kStat);
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (lane < 32) { st[(wid * 32 + lane) * 2] = s1; st[(wid * 32 + lane) * 2 + 1] = s2; }
        lds_barrier();
        // NI == 1: owners are waves 0, 1 (kh == 0), same 32 columns; NI == 2: column tile j is owned by waves j (kh 0) and 2 + j (kh 1)
        float* part = reinterpret_cast<float*>(rp(ab, d.stats));
        if (NI == 1) {
          if (wid == 0 && lane < 32) {
            part[((int64_t)tile * 2 + 0) * d.Npad + lane] = st[lane * 2] + st[(32 + lane) * 2];
            part[((int64_t)tile * 2 + 1) * d.Npad + lane] = st[lane * 2 + 1] + st[(32 + lane) * 2 + 1];
          }
        } else if (wid < 2 && lane < 32) {                     // wave j owns column tile j (both row blocks)
          const int n = wid * 32 + lane;
          part[((int64_t)tile * 2 + 0) * d.Npad + n] = st[(wid * 32 + lane) * 2];
          part[((int64_t)tile * 2 + 1) * d.Npad + n] = st[(wid * 32 + lane) * 2 + 1];
        }
      }
      s1 = 0.f; s2 = 0.f;
    }
  }
}

}  // namespace

// OPT-IN (SEFD_WINCONV=1; the per-op tests run it): parity-exact, but as measured it does not beat the tiled kernel yet
// (profiles/r03_tuning_notes.md section 6: dec4 even phase 189 us vs 140 us tiled; with SEFD_WC_DBG the time splits into ~45 us of loop /
// barrier / index overhead, ~80 us of window DMA, ~30 us of MFMA + fragment reads and 40-75 us of reduction + epilogue that one
// workgroup per CU runs one after the other).  What it needs is overlap between those phases (two co-resident workgroups - i.e. <= 72 KB
// of LDS - or loader / multiplier / storer wave roles), not fewer bytes.
bool launch_winconv(const RunGemm& d, const ArenaBases& ab, hipStream_t st) {
  const char* eo = getenv("SEFD_WINCONV");                   // read per launch (one test case of the suite switches it on)
  const bool on = eo && atoi(eo) != 0;
  const char* em = getenv("SEFD_WINCONV_MINM");
  const int minm = em ? atoi(em) : 65536;
  if (!on || d.xdt != DT_BF16 || d.ydt != DT_BF16 || !(d.flags & kRunAligned) || !(d.flags & kRunYAligned)) return false;
  if ((d.flags & (kRunAccum | kRunRelu | kRunWTile32 | kRunBnBwd)) || d.n2 > 0 || d.Npad > 64 || d.M < minm) return false;
  if (d.Fo != 64 && d.Fo != 128) return false;
  if (d.Npad == 64 && d.Fo != 64) return false;              // (the address table of the two-row-block variant assumes identical halves)
  if (d.M % d.Fo != 0 || d.ldw % 8 != 0) return false;
  WcGeom gm{};
  int nsteps = 0;
  for (int s = 0; s < d.nseg; ++s) {
    const Seg& sg = d.seg[s];
    if (sg.src < 0 || sg.len % 16 != 0 || sg.off % 8 != 0) return false;
    const int src = sg.src > 0 ? 1 : 0;
    if (d.rowlen[src] * 2 > kWcSlot || d.rowlen[src] % 8 != 0 || d.fstride[src] % 8 != 0 || d.fstride[src] < 8) return false;
    int q = -1;
    for (int i = 0; i < gm.nq; ++i) if (gm.qsrc[i] == src && gm.qdt[i] == sg.dt) q = i;
    if (q < 0) { if (gm.nq == kWcMaxQ) return false; q = gm.nq++; gm.qsrc[q] = src; gm.qdt[q] = sg.dt; }
    gm.segq[s] = q;
    nsteps += sg.len / 16;
  }
  if (nsteps < 8 || nsteps > 48) return false;               // tiny K: the direct kernel; beyond 24 steps per wave the weights do not fit registers
  gm.nsteps = nsteps;
  gm.spw = (nsteps + 1) / 2;
  static const int ncu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
  const int ntiles = (d.M + 127) / 128;
  const dim3 grid(ntiles < ncu ? ntiles : ncu);             // 126 KB of LDS: one workgroup per CU, the frame pipeline keeps it busy
  const int ni = d.Npad / 32;
  static const int dbg = getenv("SEFD_WC_DBG") ? atoi(getenv("SEFD_WC_DBG")) : 0;
#define SEFD_WC(NI_, SPW_) hipLaunchKernelGGL((winconv_kernel<NI_, SPW_>), grid, dim3(256), 0, st, d, ab, gm, dbg)
  if (ni == 1) {
    if (gm.spw <= 8) SEFD_WC(1, 8); else if (gm.spw <= 12) SEFD_WC(1, 12); else if (gm.spw <= 16) SEFD_WC(1, 16); else SEFD_WC(1, 24);
  } else {
    if (gm.spw <= 8) SEFD_WC(2, 8); else if (gm.spw <= 12) SEFD_WC(2, 12); else if (gm.spw <= 16) SEFD_WC(2, 16); else SEFD_WC(2, 24);
  }
#undef SEFD_WC
  return true;
}

}  // namespace sefd
