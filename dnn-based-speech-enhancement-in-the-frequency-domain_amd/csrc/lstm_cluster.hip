// LSTM recurrence for hidden sizes whose W_hh does not fit one CU (H = 192 .. 512, bf16 mode): a CLUSTER of H/64 workgroups
// shares one block of 16 sequences.  Same descriptor (LstmRec), buffers, gate-column order and arithmetic contract as the
// single-CU kernels of lstm_bf16.hip (reference cell: torch.nn.LSTM as used by tools_for_model.py:141-181 / models.py:96-105).
//
//   * workgroup (j, rb, g) = 4 waves; wave w owns the 16 hidden units 64 j + 16 w .. +16 of the 16 sequences of row block rb of
//     group g, with its slice of W_hh (4 gates x H inputs x 16 units, bf16) resident in VGPRs for the whole sequence;
//   * the recurrence h_{t-1} -> gates needs ALL H units of h_{t-1}: every wave gathers the 16 x H tile straight into MFMA A
//     fragments from the h output array itself, which its producers (the waves of the cluster, this one included) store
//     write-through (`sc1`) right after the cell update.  The array is pre-filled with the bf16 pattern 0xFFFF (a NaN no
//     conversion produces: the producer maps it to 0x7FC0), and a consumer re-reads a fragment until no half-word of it is
//     0xFFFF: the data is its own flag, no barrier and no separate flag round trip (MI355X_MICROARCH.md, hand-off price list:
//     data-tagged hand-off ~1 us against 1.7-1.9x that for payload + flag and >= 4 us for a grid barrier);
//   * nothing inside the step loop synchronises the four waves of a workgroup: they are independent members of the cluster;
//   * backward: the same with dgates_t (16 x 4H, bf16) as the exchanged tile and W_hh^T slices in registers.
// Per step a wave issues H/8 (forward) MFMAs of 16 cycles; the step time is the hand-off latency plus that, ~2-3 us, against
// ~50 us for the GEMM + cell launch pair per step this replaces (DCCRN-large: 203 ms of a 229 ms step).
// Dispatch order: cluster members are consecutive block ids, so a partially resident cluster only ever waits for blocks that
// are next in the dispatch queue; every spin is bounded (a latched budget) so a lost block cannot hang the device; a wave whose budget ran out reports it
// in the issuing plan's host-mapped status word: the guarded Adam skips the update and the plan's next run returns -5 (api.hip).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include "sefd_desc.h"
#include "tuning.h"
#include "dev_common.h"

#pragma clang fp contract(off)

namespace sefd {

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 exp2_2(f32x2 x) { return f32x2{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }
__device__ __forceinline__ f32x2 rcp_2(f32x2 x) { return f32x2{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
__device__ __forceinline__ f32x2 sigmoid2(f32x2 x) { return rcp_2(exp2_2(x * -1.4426950408889634f) + 1.f); }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 tanh2(f32x2 x) { return fma2(rcp_2(exp2_2(x * 2.8853900817779268f) + 1.f), f32x2{-2.f, -2.f}, f32x2{1.f, 1.f}); }

constexpr int kSc1 = 16;                         // aux bits of the raw buffer builtins: sc1 = write-through store / L1-bypassing load
constexpr int kSpinBudget = 1 << 18;             // polls a wave may spend waiting over the whole launch before it stops waiting

// OR-accumulated "some 16-bit half of x is 0xFFFF" test: bit 15 / 31 of ((~x - 0x00010001) & x) is set for such a half
__device__ __forceinline__ uint32_t unset_bits(uint32_t acc, uint32_t x) { return acc | ((~x - 0x00010001u) & x); }
__device__ __forceinline__ uint32_t unset4(uint32_t acc, u32x4 v) { return unset_bits(unset_bits(unset_bits(unset_bits(acc, v.x), v.y), v.z), v.w); }
// the pattern itself never leaves a producer
__device__ __forceinline__ uint32_t clean2(uint32_t p) {
  if ((p & 0xffffu) == 0xffffu) p = (p & 0xffff0000u) | 0x7fc0u;
  if ((p >> 16) == 0xffffu) p = (p & 0xffffu) | 0x7fc00000u;
  return p;
}

// Gather N 16-byte fragments (byte offsets off0 + stride * i of buffer r) until none holds an unwritten half-word.
template <int N>
__device__ __forceinline__ void gather(__amdgpu_buffer_rsrc_t r, uint32_t off0, uint32_t stride, u32x4 (&a)[N], int& budget) {
  for (;;) {
    uint32_t bad = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = __builtin_amdgcn_raw_buffer_load_b128(r, off0 + stride * i, 0, kSc1);
#pragma unroll
    for (int i = 0; i < N; ++i) bad = unset4(bad, a[i]);
    if (!__any((bad & 0x80008000u) != 0)) return;
    if (budget <= 0) return;                     // latched: after the budget is spent no later step waits either
    --budget;
    __builtin_amdgcn_s_sleep(1);
  }
}

// A wave that spent its whole budget consumed fragments its peers had not written: the launch's results are garbage.  It says so in a
// host-mapped status word of its plan (system-scope store): the guarded Adam kernel of the same step leaves the parameters untouched and the
// host raises at the plan's next run / before a checkpoint is written - the step fails loudly instead of training on NaNs.
__device__ __forceinline__ void report_timeout(int* status, int* dstatus, int budget) {
  if (budget <= 0 && (threadIdx.x & 63) == 0) set_status(status, dstatus);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- forward
// Work of one wave = the linear sequence of tiles (t, mt): frame t, row tile mt (16 sequences) of the workgroup's 16 * MT rows.
// MT = 1 is the latency-bound case (few sequences, e.g. DCCRN: B <= a few hundred); with thousands of sequences (FullSubNet's
// sub-band model: B * 257 rows) a workgroup walks MT row tiles per frame, so the hand-off latency of one tile hides behind the
// work on the others (the gather of the next tile is issued before the MFMAs of this one, template flag PF).
// Buffers may be batch-major [row][T][..] (DCCRN / CRN) or time-major [T][row][..] (FullSubNet, d.tmajor).
constexpr int kMaxMT = 8;

template <int H, bool PF>
__global__ __launch_bounds__(256) void lstm_fwd_cluster_kernel(const LstmRec d, const ArenaBases ab, const int MT, int* status, int* dstatus) {
  constexpr int KS = H / 32;
  constexpr int HAS = H + 8, HCPR = H / 8, HNCH = (16 * HCPR + 255) / 256;      // cooperative gather tile: row stride, chunks per row / thread
  __shared__ __attribute__((aligned(16))) uint16_t htile[PF ? 8 : 2 * 16 * HAS];
  __shared__ __attribute__((aligned(16))) uint16_t stage[4][16 * 16];
  __shared__ __attribute__((aligned(16))) float4 cst[4][kMaxMT][64];
  const int T = d.T;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = blockIdx.x, b0 = blockIdx.y * 16 * MT, g = blockIdx.z;
  const int64_t rstr = d.tmajor ? 1 : T, tstr = d.tmajor ? d.B : 1;       // row / frame strides, in rows
  const float* whh = reinterpret_cast<const float*>(rp(ab, d.whh[g % d.nset]));
  const float* gx = reinterpret_cast<const float*>(rp(ab, d.gx)) + d.gx_goff[g];
  const int64_t GBT = (int64_t)d.B * T;
  uint16_t* hgrp = reinterpret_cast<uint16_t*>(rp(ab, d.h)) + (int64_t)g * GBT * H;
  float* gates = reinterpret_cast<float*>(rp(ab, d.gates)) + (int64_t)g * GBT * H * 4;
  float* cs = reinterpret_cast<float*>(rp(ab, d.c)) + (int64_t)g * GBT * H;
  const int ubase = 64 * j + 16 * w;
  const int unit = ubase + (lane & 15);
  const int kq = lane >> 4;
  const int64_t gx_ld = d.gx_ld;
  const uint32_t hbytes = (uint32_t)((((int64_t)d.B - 1) * rstr + 1) * H * 2);
  auto hres = [&](int frame) {                   // frame `frame` of every row: row b at byte offset b * rstr * H * 2
    return __builtin_amdgcn_make_buffer_rsrc(hgrp + (int64_t)frame * tstr * H, 0, hbytes, 0x00020000);
  };

  uint4 wreg[4][KS];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float* p = whh + (int64_t)(q * H + unit) * H + 32 * ks + 8 * kq;
      const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
      wreg[q][ks] = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
    }
  const int tb = d.t0, te = d.t1 > 0 ? d.t1 : T;
  const int ntile = (te - tb) * MT;

  // cell (r = 0..3) of tile mt: row b0 + 16 mt + 4 kq + r, this lane's unit
  auto cell_row = [&](int mt, int r, bool& valid) -> int64_t {
    const int b = b0 + 16 * mt + 4 * kq + r;
    valid = b < d.B;
    return valid ? b : 0;                        // rows beyond the batch alias row 0 and are never stored
  };
  for (int mt = 0; mt < MT; ++mt) {              // cell state of frame tb - 1
    float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tb > 0) {
      float cc[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { bool v; const int64_t b = cell_row(mt, r, v); cc[r] = cs[(b * rstr + (int64_t)(tb - 1) * tstr) * H + unit]; }
      c0 = make_float4(cc[0], cc[1], cc[2], cc[3]);
    }
    cst[w][mt][lane] = c0;
  }
  const int srow = lane >> 1, shalf = lane & 1;
  uint16_t* stg = &stage[w][0];
  int budget = kSpinBudget;

  int pt = tb, pmt = 0;                          // prefetch cursor of the gate pre-activations (two tiles ahead)
  auto load_gx = [&](float4 (&dst)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bool v;
      const int64_t b = cell_row(pmt, r, v);
      dst[r] = *reinterpret_cast<const float4*>(gx + (b * rstr + (int64_t)pt * tstr) * gx_ld + gate_col(0, unit));
    }
    if (pt < te - 1 || pmt < MT - 1) { if (++pmt == MT) { pmt = 0; ++pt; } }      // the last prefetches re-read the last tile
  };
  auto a_off = [&](int mt) -> uint32_t {         // A fragment of tile mt: sequence lane & 15 of the tile, inputs 8 kq .. (+ 32 ks)
    const int b = b0 + 16 * mt + (lane & 15);
    return (uint32_t)(((int64_t)(b < d.B ? b : 0) * rstr * H + 8 * kq) * 2);
  };
  u32x4 an[PF ? KS : 1];
  auto issue_next = [&](int t, int mt) {         // start the gather of the tile after (t, mt), if it has a recurrent term
    if constexpr (PF) {
      int nt = t, nmt = mt + 1;
      if (nmt == MT) { nmt = 0; ++nt; }
      if (nt < te && nt > 0) {
        const auto r = hres(nt - 1);
        const uint32_t o = a_off(nmt);
#pragma unroll
        for (int i = 0; i < KS; ++i) an[i] = __builtin_amdgcn_raw_buffer_load_b128(r, o + 64u * i, 0, kSc1);
      }
    }
  };
  int t = tb, mt = 0, hpar = 0;
  auto tile = [&](const float4 (&cur)[4], float4 (&pre)[4]) {
    f32x4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc[0][r] = cur[r].x; acc[1][r] = cur[r].y; acc[2][r] = cur[r].z; acc[3][r] = cur[r].w; }
    load_gx(pre);
    if (t > 0) {
      u32x4 a[KS];
      bool have = false;
      if constexpr (PF) {
        if (t > tb || mt > 0) {                  // prefetched by the previous tile: usable if complete
          uint32_t bad = 0;
#pragma unroll
          for (int i = 0; i < KS; ++i) { a[i] = an[i]; bad = unset4(bad, a[i]); }
          have = !__any((bad & 0x80008000u) != 0);
        }
      }
      if constexpr (!PF) {
        // the 16 x H tile of h_{t-1} is gathered once per workgroup (thread i: chunks i, i + 256, ...: whole lines, a quarter of the loads and
        // checks per wave), shared through a ping-pong LDS tile - see the backward kernel
        const auto rs = hres(t - 1);
        uint16_t* at = htile + hpar * 16 * HAS;
        u32x4 ch[HNCH];
        uint32_t off[HNCH];
#pragma unroll
        for (int i = 0; i < HNCH; ++i) {
          const int c = (int)threadIdx.x + 256 * i, row = c / HCPR, col = c - row * HCPR;
          const int b = b0 + 16 * mt + row;
          off[i] = (uint32_t)(((int64_t)(b < d.B ? b : 0) * rstr * H + 8 * col) * 2);
        }
        for (;;) {
          uint32_t bad = 0;
#pragma unroll
          for (int i = 0; i < HNCH; ++i) if (HNCH * 256 == 16 * HCPR || (int)threadIdx.x + 256 * i < 16 * HCPR) ch[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[i], 0, kSc1);
#pragma unroll
          for (int i = 0; i < HNCH; ++i) if (HNCH * 256 == 16 * HCPR || (int)threadIdx.x + 256 * i < 16 * HCPR) bad = unset4(bad, ch[i]);
          if (!__any((bad & 0x80008000u) != 0) || budget <= 0) break;
          --budget;
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int i = 0; i < HNCH; ++i) {
          const int c = (int)threadIdx.x + 256 * i, row = c / HCPR, col = c - row * HCPR;
          if (HNCH * 256 == 16 * HCPR || c < 16 * HCPR) *reinterpret_cast<u32x4*>(at + row * HAS + 8 * col) = ch[i];
        }
        lds_barrier();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[ks] = *reinterpret_cast<const u32x4*>(at + (lane & 15) * HAS + 32 * ks + 8 * kq);
        hpar ^= 1;
      } else {
        if (!have) gather<KS>(hres(t - 1), a_off(mt), 64u, a, budget);
      }
      issue_next(t, mt);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[ks]), __builtin_bit_cast(bf16x8, wreg[q][ks]), acc[q], 0, 0, 0);
    } else {
      issue_next(t, mt);
    }
    const float4 cv = cst[w][mt][lane];
    const float c[4] = {cv.x, cv.y, cv.z, cv.w};
    float cnew[4];
#pragma unroll
    for (int rp2 = 0; rp2 < 4; rp2 += 2) {
      const f32x2 ig = sigmoid2(f32x2{acc[0][rp2], acc[0][rp2 + 1]}), fg = sigmoid2(f32x2{acc[1][rp2], acc[1][rp2 + 1]});
      const f32x2 gg = tanh2(f32x2{acc[2][rp2], acc[2][rp2 + 1]}), og = sigmoid2(f32x2{acc[3][rp2], acc[3][rp2 + 1]});
      const f32x2 cn = fma2(fg, f32x2{c[rp2], c[rp2 + 1]}, ig * gg);
      const f32x2 hv = og * tanh2(cn);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int r = rp2 + k;
        cnew[r] = cn[k];
        uint16_t hb = f2bf(hv[k]);
        if (hb == 0xffffu) hb = 0x7fc0u;
        stg[(4 * kq + r) * 16 + (lane & 15)] = hb;
        bool v;
        const int64_t b = cell_row(mt, r, v);
        if (v) {
          const int64_t so = (b * rstr + (int64_t)t * tstr) * H + unit;
          *reinterpret_cast<float4*>(gates + so * 4) = make_float4(ig[k], fg[k], gg[k], og[k]);
          cs[so] = cn[k];
        }
      }
    }
    cst[w][mt][lane] = make_float4(cnew[0], cnew[1], cnew[2], cnew[3]);
    // wave-local transpose through LDS (DS operations of one wave execute in order; the fences only stop the compiler)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 32) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(stg + srow * 16 + 8 * shalf);
      const int b = b0 + 16 * mt + srow;
      if (b < d.B) __builtin_amdgcn_raw_buffer_store_b128(v, hres(t), (uint32_t)(((int64_t)b * rstr * H + ubase + 8 * shalf) * 2), 0, kSc1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (++mt == MT) { mt = 0; ++t; }
  };
  float4 b0v[4], b1v[4], b2v[4];
  load_gx(b0v);
  load_gx(b1v);
  int n = 0;
  for (; n + 3 <= ntile; n += 3) {
    tile(b0v, b2v);
    tile(b1v, b0v);
    tile(b2v, b1v);
  }
  if (n < ntile) { tile(b0v, b2v); ++n; }
  if (n < ntile) { tile(b1v, b0v); ++n; }
  report_timeout(status, dstatus, budget);
}

// --------------------------------------------------------------------------------------------------------------- backward
// Tile (t, mt), frames last to first: dh_rec = dgates_{t+1}[tile] . W_hh (gathered first: the peers stored it MT tiles ago),
// then the cell backward of frame t, whose dgates_t leave write-through for the peers' (and this wave's) tile (t-1, mt).
template <int H>
__global__ __launch_bounds__(256) void lstm_bwd_cluster_kernel(const LstmRec d, const ArenaBases ab, const int MT, int* status, int* dstatus) {
  constexpr int KS = 4 * H / 32;                 // k32 steps over the 4H gate columns
  constexpr int AS = 4 * H + 8, CPR = 4 * H / 8, NCH = 16 * CPR / 256;   // LDS tile row stride, 16-byte chunks per row, chunks per thread
  constexpr bool DB = H <= 448;                  // two tiles (ping-pong, one barrier per tile) while 2 x 16 x 4H bf16 fits next to the rest
  __shared__ __attribute__((aligned(16))) uint16_t atile[(DB ? 2 : 1) * 16 * AS];
  __shared__ __attribute__((aligned(16))) uint16_t stage[4][16 * 64];
  __shared__ __attribute__((aligned(16))) float4 dcs[4][kMaxMT][64];
  const int T = d.T;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = blockIdx.x, b0 = blockIdx.y * 16 * MT, g = blockIdx.z;
  const int64_t rstr = d.tmajor ? 1 : T, tstr = d.tmajor ? d.B : 1;
  const int64_t GBT = (int64_t)d.B * T;
  const float* whh = reinterpret_cast<const float*>(rp(ab, d.whh[g % d.nset]));
  const float* gates = reinterpret_cast<const float*>(rp(ab, d.gates)) + (int64_t)g * GBT * H * 4;
  const float* cs = reinterpret_cast<const float*>(rp(ab, d.c)) + (int64_t)g * GBT * H;
  const float* dh = reinterpret_cast<const float*>(rp(ab, d.dh)) + (int64_t)g * GBT * H;
  uint16_t* dgrp = reinterpret_cast<uint16_t*>(rp(ab, d.dgates)) + d.gx_goff[g];          // this group's rows, ld gx_ld
  const int ubase = 64 * j + 16 * w;
  const int unit = ubase + (lane & 15);
  const int kq = lane >> 4;
  const int64_t gx_ld = d.gx_ld;
  const uint32_t gbytes = (uint32_t)(((((int64_t)d.B - 1) * rstr) * gx_ld + 4 * H) * 2);
  auto gres = [&](int frame) {
    return __builtin_amdgcn_make_buffer_rsrc(dgrp + (int64_t)frame * tstr * gx_ld, 0, gbytes, 0x00020000);
  };

  // B[k = gate column][n = unit] = W_hh[torch row of that column][unit]
  uint4 wreg[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = whh[(int64_t)gate_torch_row(32 * ks + 8 * kq + e, H) * H + unit];
    wreg[ks] = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
  }
  auto cell_row = [&](int mt, int r, bool& valid) -> int64_t {
    const int b = b0 + 16 * mt + 4 * kq + r;
    valid = b < d.B;
    return valid ? b : 0;
  };
  for (int mt = 0; mt < MT; ++mt) dcs[w][mt][lane] = make_float4(0.f, 0.f, 0.f, 0.f);
  uint16_t* stg = &stage[w][0];
  int budget = kSpinBudget, par = 0;
  const int ntile = T * MT;

  struct Sav { float4 g[4]; float cp[4], ct[4], dh[4]; };
  int pt = T - 1, pmt = 0;
  auto fetch = [&](Sav& s) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bool v;
      const int64_t b = cell_row(pmt, r, v);
      const int64_t o = (b * rstr + (int64_t)pt * tstr) * H + unit;
      s.g[r] = *reinterpret_cast<const float4*>(gates + o * 4);
      s.ct[r] = cs[o];
      s.cp[r] = pt > 0 ? cs[o - tstr * H] : 0.f;
      s.dh[r] = dh[o];
    }
    if (pt > 0 || pmt < MT - 1) { if (++pmt == MT) { pmt = 0; --pt; } }
  };
  int t = T - 1, mt = 0;
  auto tile = [&](const Sav& cur, Sav& pre) {
    fetch(pre);
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    if (t < T - 1) {
      // The 16 x 4H tile of dgates_{t+1} is gathered ONCE per workgroup: thread i takes the 16-byte chunks i, i + 256, ... (whole lines,
      // a quarter of the tile and of the validity checks per wave instead of all of it), re-reads them until none holds an unwritten
      // half-word, and puts them into the LDS tile; after the barrier every wave reads its A fragments from there.
      const auto r = gres(t + 1);
      uint16_t* at = atile + (DB ? par * 16 * AS : 0);
      u32x4 ch[NCH];
      uint32_t off[NCH];
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = (int)threadIdx.x + 256 * i, row = c / CPR, col = c - row * CPR;
        const int b = b0 + 16 * mt + row;
        off[i] = (uint32_t)(((int64_t)(b < d.B ? b : 0) * rstr * gx_ld + 8 * col) * 2);
      }
      for (;;) {
        uint32_t bad = 0;
#pragma unroll
        for (int i = 0; i < NCH; ++i) ch[i] = __builtin_amdgcn_raw_buffer_load_b128(r, off[i], 0, kSc1);
#pragma unroll
        for (int i = 0; i < NCH; ++i) bad = unset4(bad, ch[i]);
        if (!__any((bad & 0x80008000u) != 0) || budget <= 0) break;
        --budget;
        __builtin_amdgcn_s_sleep(1);
      }
      if (!DB) lds_barrier();                     // single tile (H = 512): the previous tile's readers must be done before it is overwritten
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = (int)threadIdx.x + 256 * i, row = c / CPR, col = c - row * CPR;
        *reinterpret_cast<u32x4*>(at + row * AS + 8 * col) = ch[i];
      }
      lds_barrier();
#pragma unroll
      for (int ks = 0; ks < KS; ks += 2) {
        const u32x4 x0 = *reinterpret_cast<const u32x4*>(at + (lane & 15) * AS + 32 * ks + 8 * kq);
        const u32x4 x1 = *reinterpret_cast<const u32x4*>(at + (lane & 15) * AS + 32 * (ks + 1) + 8 * kq);
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x0), __builtin_bit_cast(bf16x8, wreg[ks]), a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x1), __builtin_bit_cast(bf16x8, wreg[ks + 1]), a1, 0, 0, 0);
      }
      par ^= 1;
    }
    const f32x4 dhrec = a0 + a1;
    const float4 dcv = dcs[w][mt][lane];
    const float dcarry[4] = {dcv.x, dcv.y, dcv.z, dcv.w};
    float dcn4[4];
#pragma unroll
    for (int rp2 = 0; rp2 < 4; rp2 += 2) {
      const f32x2 ig = {cur.g[rp2].x, cur.g[rp2 + 1].x}, fg = {cur.g[rp2].y, cur.g[rp2 + 1].y};
      const f32x2 gg = {cur.g[rp2].z, cur.g[rp2 + 1].z}, og = {cur.g[rp2].w, cur.g[rp2 + 1].w};
      const f32x2 ct = {cur.ct[rp2], cur.ct[rp2 + 1]};
      const f32x2 cp = {cur.cp[rp2], cur.cp[rp2 + 1]};
      const f32x2 dht = f32x2{cur.dh[rp2], cur.dh[rp2 + 1]} + f32x2{dhrec[rp2], dhrec[rp2 + 1]};
      const f32x2 tc = tanh2(ct);
      const f32x2 dog = dht * tc * og * (1.f - og);
      const f32x2 dc = dht * og * (1.f - tc * tc) + f32x2{dcarry[rp2], dcarry[rp2 + 1]};
      const f32x2 di = dc * gg * ig * (1.f - ig);
      const f32x2 df = dc * cp * fg * (1.f - fg);
      const f32x2 dg = dc * ig * (1.f - gg * gg);
      const f32x2 dcn = dc * fg;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int r = rp2 + k;
        bool v;
        (void)cell_row(mt, r, v);
        dcn4[r] = v ? dcn[k] : 0.f;
        *reinterpret_cast<uint2*>(stg + (4 * kq + r) * 64 + 4 * (lane & 15)) =
            make_uint2(clean2(pack_bf16x2(di[k], df[k])), clean2(pack_bf16x2(dg[k], dog[k])));
      }
    }
    dcs[w][mt][lane] = make_float4(dcn4[0], dcn4[1], dcn4[2], dcn4[3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
      // dgates_t of this wave: 16 sequences x (16 units x 4 gates) = 16 x 128 bytes = 128 chunks of 16 bytes, two per lane
      const auto r = gres(t);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int cidx = lane + 64 * i, row = cidx >> 3, piece = cidx & 7;
        const u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * 64 + 8 * piece);
        const int b = b0 + 16 * mt + row;
        if (b < d.B) __builtin_amdgcn_raw_buffer_store_b128(v, r, (uint32_t)(((int64_t)b * rstr * gx_ld + 4 * ubase + 8 * piece) * 2), 0, kSc1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (++mt == MT) { mt = 0; --t; }
  };
  Sav s0, s1, s2;
  fetch(s0);
  fetch(s1);
  int n = 0;
  for (; n + 3 <= ntile; n += 3) {
    tile(s0, s2);
    tile(s1, s0);
    tile(s2, s1);
  }
  if (n < ntile) { tile(s0, s2); ++n; }
  if (n < ntile) { tile(s1, s0); ++n; }
  report_timeout(status, dstatus, budget);
}

// ------------------------------------------------------------------------------------------------------------------ launch
bool lstm_cluster_supported(int H) { return H > 128 && H <= 512 && H % 64 == 0; }

// "unwritten" marks: frames [t0, t1) of every sequence of h (16-byte chunks; rows x T batch-major or T x rows time-major)
__global__ void lstm_mark_kernel(uint4* h, int64_t rows, int T, int H8, int t0, int t1, int tmajor) {
  const int64_t per = (int64_t)(t1 - t0) * H8, n = rows * per;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / per, rem = i - row * per;
    const int64_t tt = rem / H8, c = rem - tt * H8;
    const int64_t o = tmajor ? ((t0 + tt) * rows + row) * H8 + c : (row * T + t0 + tt) * H8 + c;
    h[o] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
  }
}

// rows per workgroup = 16 * MT: the fewest dispatch rounds over the chip's CUs (one workgroup per CU: the weights fill its registers)
static int pick_mt(const LstmRec& d, int nc) {
  static int ncu = 0;
  if (!ncu) { hipDeviceProp_t p; int dev = 0; (void)hipGetDevice(&dev); (void)hipGetDeviceProperties(&p, dev); ncu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256; }
  if (const char* e = tune_str("LSTM_MT")) { const int v = atoi(e); if (v >= 1 && v <= kMaxMT) return v; }   // tuning / test override
  const int64_t nb16 = (d.B + 15) / 16;
  if (nb16 * nc * d.G <= ncu) return 1;
  // measured on FullSubNet's sub-band model (16448 rows, H = 384; ms per training step): MT 1: 187, 2: 156, 3: 158, 8: 183 - two
  // tiles overlap one hand-off with the other tile's work, more only lengthen a workgroup's frame; ties go to the larger of 1 / 2
  int best = 1; int64_t cost = -1;
  for (int mt = 2; mt >= 1; --mt) {
    const int64_t wgs = ((nb16 + mt - 1) / mt) * nc * d.G, c = ((wgs + ncu - 1) / ncu) * mt;
    if (cost < 0 || c < cost) { cost = c; best = mt; }
  }
  return best;
}

// Fallback status word for launches issued without a plan's own word (ArenaBases::status == nullptr): host-mapped, never read by the library.
static int* cluster_status_word() {
  static int* w = [] {
    int* q = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&q), sizeof(int), hipHostMallocMapped) != hipSuccess) return static_cast<int*>(nullptr);
    *q = 0;
    return q;
  }();
  return w;
}

template <int H>
static void launch_c(const LstmRec& d, const ArenaBases& ab, hipStream_t st, bool fwd) {
  int* status = ab.status ? ab.status : cluster_status_word();      // the issuing plan's word (api.hip plan_run)
  const int mt = pick_mt(d, H / 64);
  const dim3 grid(H / 64, (d.B + 16 * mt - 1) / (16 * mt), d.G);
  if (fwd) {
    if constexpr (H <= 384) { if (mt > 1) { hipLaunchKernelGGL((lstm_fwd_cluster_kernel<H, true>), grid, dim3(256), 0, st, d, ab, mt, status, ab.dstatus); return; } }
    hipLaunchKernelGGL((lstm_fwd_cluster_kernel<H, false>), grid, dim3(256), 0, st, d, ab, mt, status, ab.dstatus);
  } else {
    hipLaunchKernelGGL((lstm_bwd_cluster_kernel<H>), grid, dim3(256), 0, st, d, ab, mt, status, ab.dstatus);
  }
}

void launch_lstm_cluster(const LstmRec& d, const ArenaBases& ab, hipStream_t st, bool fwd) {
  // the exchanged arrays start out "unwritten": the frames of h this launch produces, all of dgates before the backward
  if (fwd) {
    const int t0 = d.t0, t1 = d.t1 > 0 ? d.t1 : d.T;
    const int64_t n = (int64_t)d.G * d.B * (t1 - t0) * (d.H / 8);
    for (int g = 0; g < d.G; ++g)
      hipLaunchKernelGGL(lstm_mark_kernel, dim3((unsigned)std::min<int64_t>((n / d.G + 255) / 256, 2048)), dim3(256), 0, st,
                         reinterpret_cast<uint4*>(rp(ab, d.h)) + (int64_t)g * d.B * d.T * (d.H / 8), (int64_t)d.B, d.T, d.H / 8, t0, t1, d.tmajor);
  } else {
    int64_t hi = 0;
    for (int g = 0; g < d.G; ++g) hi = d.gx_goff[g] > hi ? d.gx_goff[g] : hi;
    const int64_t n = hi + ((int64_t)d.B * d.T - 1) * d.gx_ld + 4 * d.H;
    (void)hipMemsetD16Async((hipDeviceptr_t)rp(ab, d.dgates), 0xffff, (size_t)n, st);
  }
  switch (d.H) {
    case 192: launch_c<192>(d, ab, st, fwd); break;
    case 256: launch_c<256>(d, ab, st, fwd); break;
    case 320: launch_c<320>(d, ab, st, fwd); break;
    case 384: launch_c<384>(d, ab, st, fwd); break;
    case 448: launch_c<448>(d, ab, st, fwd); break;
    default: launch_c<512>(d, ab, st, fwd); break;
  }
}

}  // namespace sefd
