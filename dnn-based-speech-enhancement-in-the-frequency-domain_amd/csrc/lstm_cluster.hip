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
// are next in the dispatch queue; every spin is bounded (a latched budget) so a lost block cannot hang the device.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "sefd_desc.h"
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

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- forward
template <int H>
__global__ __launch_bounds__(256) void lstm_fwd_cluster_kernel(const LstmRec d, const ArenaBases ab) {
  constexpr int KS = H / 32;
  __shared__ __attribute__((aligned(16))) uint16_t stage[4][16 * 16];
  const int T = d.T;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = blockIdx.x, b0 = blockIdx.y * 16, g = blockIdx.z;
  const float* whh = reinterpret_cast<const float*>(rp(ab, d.whh[g % d.nset]));
  const float* gx = reinterpret_cast<const float*>(rp(ab, d.gx)) + d.gx_goff[g];
  const int64_t GBT = (int64_t)d.B * T;
  uint16_t* hgrp = reinterpret_cast<uint16_t*>(rp(ab, d.h)) + (int64_t)g * GBT * H;       // this group's [B][T][H]
  float* gates = reinterpret_cast<float*>(rp(ab, d.gates));
  float* cs = reinterpret_cast<float*>(rp(ab, d.c));
  const int ubase = 64 * j + 16 * w;
  const int unit = ubase + (lane & 15);
  const int kq = lane >> 4;
  const __amdgpu_buffer_rsrc_t hres = __builtin_amdgcn_make_buffer_rsrc(hgrp, 0, (uint32_t)(GBT * H * 2), 0x00020000);

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

  bool rvalid[4];
  int64_t rowbt[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + 4 * kq + r;
    rvalid[r] = b < d.B;
    rowbt[r] = (int64_t)(rvalid[r] ? b : 0) * T;
  }
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  const float* gxp[4];
  int64_t so[4];                                 // (g*GBT + row*T + t) * H + unit
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    gxp[r] = gx + (rowbt[r] + tb) * d.gx_ld + gate_col(0, unit);
    so[r] = ((int64_t)g * GBT + rowbt[r] + tb) * H + unit;
    if (tb > 0) c[r] = cs[so[r] - H];
  }
  // A fragment source: sequence (lane & 15) of the block, inputs 32 ks + 8 kq .. +8 of frame t-1 (rows beyond the batch alias row 0)
  const int arow = b0 + (lane & 15) < d.B ? b0 + (lane & 15) : 0;
  uint32_t aoff = (uint32_t)((((int64_t)arow * T + (tb - 1)) * H + 8 * kq) * 2);
  // h store: the wave's 16 x 16 tile leaves as 32 chunks of 16 bytes (lanes 0..31: sequence lane >> 1, half lane & 1)
  const int srow = lane >> 1, shalf = lane & 1;
  const bool svalid = lane < 32 && b0 + srow < d.B;
  uint32_t soff = (uint32_t)((((int64_t)(b0 + srow) * T + tb) * H + ubase + 8 * shalf) * 2);
  uint16_t* stg = &stage[w][0];
  int budget = kSpinBudget;

  const int64_t gx_ld = d.gx_ld;
  auto load_gx = [&](int t, float4 (&dst)[4]) {
    const int64_t inc = t < T - 1 ? gx_ld : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) { dst[r] = *reinterpret_cast<const float4*>(gxp[r]); gxp[r] += inc; }
  };
  auto step = [&](int t, const float4 (&cur)[4], float4 (&pre)[4]) {
    f32x4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc[0][r] = cur[r].x; acc[1][r] = cur[r].y; acc[2][r] = cur[r].z; acc[3][r] = cur[r].w; }
    load_gx(t + 2, pre);
    if (t > 0) {
      u32x4 a[KS];
      gather<KS>(hres, aoff, 64u, a, budget);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[ks]), __builtin_bit_cast(bf16x8, wreg[q][ks]), acc[q], 0, 0, 0);
    }
    aoff += (uint32_t)(H * 2);
#pragma unroll
    for (int rp2 = 0; rp2 < 4; rp2 += 2) {
      const f32x2 ig = sigmoid2(f32x2{acc[0][rp2], acc[0][rp2 + 1]}), fg = sigmoid2(f32x2{acc[1][rp2], acc[1][rp2 + 1]});
      const f32x2 gg = tanh2(f32x2{acc[2][rp2], acc[2][rp2 + 1]}), og = sigmoid2(f32x2{acc[3][rp2], acc[3][rp2 + 1]});
      const f32x2 cn = fma2(fg, f32x2{c[rp2], c[rp2 + 1]}, ig * gg);
      const f32x2 hv = og * tanh2(cn);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int r = rp2 + k;
        c[r] = cn[k];
        uint16_t hb = f2bf(hv[k]);
        if (hb == 0xffffu) hb = 0x7fc0u;
        stg[(4 * kq + r) * 16 + (lane & 15)] = hb;
        if (rvalid[r]) {
          *reinterpret_cast<float4*>(gates + so[r] * 4) = make_float4(ig[k], fg[k], gg[k], og[k]);
          cs[so[r]] = cn[k];
        }
        so[r] += H;
      }
    }
    // wave-local transpose through LDS (DS operations of one wave execute in order; the fence only stops the compiler)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 32) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(stg + srow * 16 + 8 * shalf);
      if (svalid) __builtin_amdgcn_raw_buffer_store_b128(v, hres, soff, 0, kSc1);
    }
    soff += (uint32_t)(H * 2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  float4 b0v[4], b1v[4], b2v[4];
  load_gx(tb, b0v);
  load_gx(tb + 1, b1v);
  int t = tb;
  for (; t + 3 <= te; t += 3) {
    step(t, b0v, b2v);
    step(t + 1, b1v, b0v);
    step(t + 2, b2v, b1v);
  }
  if (t < te) { step(t, b0v, b2v); ++t; }
  if (t < te) { step(t, b1v, b0v); ++t; }
}

// --------------------------------------------------------------------------------------------------------------- backward
template <int H>
__global__ __launch_bounds__(256) void lstm_bwd_cluster_kernel(const LstmRec d, const ArenaBases ab) {
  constexpr int KS = 4 * H / 32;                 // k16x2 steps over the 4H gate columns
  constexpr int GK = KS % 16 == 0 ? 16 : 8;      // fragments gathered at a time (KS = H/8 is a multiple of 8)
  __shared__ __attribute__((aligned(16))) uint16_t stage[4][16 * 64];
  const int T = d.T;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = blockIdx.x, b0 = blockIdx.y * 16, g = blockIdx.z;
  const float* whh = reinterpret_cast<const float*>(rp(ab, d.whh[g % d.nset]));
  const float* gates = reinterpret_cast<const float*>(rp(ab, d.gates));
  const float* cs = reinterpret_cast<const float*>(rp(ab, d.c));
  const float* dh = reinterpret_cast<const float*>(rp(ab, d.dh));
  uint16_t* dgrp = reinterpret_cast<uint16_t*>(rp(ab, d.dgates)) + d.gx_goff[g];          // this group's rows, ld gx_ld
  const int ubase = 64 * j + 16 * w;
  const int unit = ubase + (lane & 15);
  const int kq = lane >> 4;
  const int64_t gx_ld = d.gx_ld;
  const int64_t GBT = (int64_t)d.B * T;
  const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(dgrp, 0, (uint32_t)(((GBT - 1) * gx_ld + 4 * H) * 2), 0x00020000);

  // B[k = gate column][n = unit] = W_hh[torch row of that column][unit]
  uint4 wreg[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = whh[(int64_t)gate_torch_row(32 * ks + 8 * kq + e, H) * H + unit];
    wreg[ks] = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
  }
  bool rvalid[4];
  int64_t fo[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + 4 * kq + r;
    rvalid[r] = b < d.B;
    fo[r] = ((int64_t)g * GBT + (int64_t)(rvalid[r] ? b : 0) * T + (T - 1)) * H + unit;
  }
  // dgates_t of this wave: 16 sequences x (16 units x 4 gates) = 16 x 128 bytes = 128 chunks, two per lane
  uint32_t soff[2];
  bool svalid[2];
  int srow[2], spiece[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int cidx = lane + 64 * i;
    srow[i] = cidx >> 3; spiece[i] = cidx & 7;
    svalid[i] = b0 + srow[i] < d.B;
    soff[i] = (uint32_t)((((int64_t)(b0 + srow[i]) * T + (T - 1)) * gx_ld + 4 * ubase + 8 * spiece[i]) * 2);
  }
  const int arow = b0 + (lane & 15) < d.B ? b0 + (lane & 15) : 0;
  uint32_t aoff = (uint32_t)((((int64_t)arow * T + (T - 1)) * gx_ld + 8 * kq) * 2);
  const uint32_t back = (uint32_t)(gx_ld * 2);
  uint16_t* stg = &stage[w][0];
  int budget = kSpinBudget;

  float dcarry[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 dhrec = {0.f, 0.f, 0.f, 0.f};
  struct Sav { float4 g[4]; float cp[4], dh[4]; };
  auto fetch = [&](int t_, Sav& s) {
    const int64_t bk = t_ > 0 ? H : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s.g[r] = *reinterpret_cast<const float4*>(gates + fo[r] * 4);
      s.cp[r] = cs[fo[r] - bk];
      s.dh[r] = dh[fo[r]];
      fo[r] -= bk;
    }
  };
  float pct[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) pct[r] = cs[fo[r]];
  auto step = [&](int t, const Sav& cur, Sav& pre) {
    fetch(t - 2, pre);
#pragma unroll
    for (int rp2 = 0; rp2 < 4; rp2 += 2) {
      const f32x2 ig = {cur.g[rp2].x, cur.g[rp2 + 1].x}, fg = {cur.g[rp2].y, cur.g[rp2 + 1].y};
      const f32x2 gg = {cur.g[rp2].z, cur.g[rp2 + 1].z}, og = {cur.g[rp2].w, cur.g[rp2 + 1].w};
      const f32x2 ct = {pct[rp2], pct[rp2 + 1]};
      const f32x2 cp = t > 0 ? f32x2{cur.cp[rp2], cur.cp[rp2 + 1]} : f32x2{0.f, 0.f};
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
        dcarry[r] = rvalid[r] ? dcn[k] : 0.f;
        pct[r] = cur.cp[r];
        *reinterpret_cast<uint2*>(stg + (4 * kq + r) * 64 + 4 * (lane & 15)) =
            make_uint2(clean2(pack_bf16x2(di[k], df[k])), clean2(pack_bf16x2(dg[k], dog[k])));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(stg + srow[i] * 64 + 8 * spiece[i]);
      if (svalid[i]) __builtin_amdgcn_raw_buffer_store_b128(v, gres, soff[i], 0, kSc1);
      soff[i] -= back;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    if (t > 0) {
#pragma unroll
      for (int k0 = 0; k0 < KS; k0 += GK) {
        u32x4 a[GK];
        gather<GK>(gres, aoff + 64u * k0, 64u, a, budget);
#pragma unroll
        for (int i = 0; i < GK; i += 2) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, wreg[k0 + i]), a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i + 1]), __builtin_bit_cast(bf16x8, wreg[k0 + i + 1]), a1, 0, 0, 0);
        }
      }
    }
    aoff -= back;
    dhrec = a0 + a1;
  };
  Sav s0, s1, s2;
  fetch(T - 1, s0);
  fetch(T - 2, s1);
  int t = T - 1;
  for (; t >= 2; t -= 3) {
    step(t, s0, s2);
    step(t - 1, s1, s0);
    step(t - 2, s2, s1);
  }
  if (t >= 0) { step(t, s0, s2); --t; }
  if (t >= 0) { step(t, s1, s0); --t; }
}

// ------------------------------------------------------------------------------------------------------------------ launch
bool lstm_cluster_supported(int H) { return H == 192 || H == 256 || H == 320 || H == 384 || H == 448 || H == 512; }

template <int H>
static void launch_c(const LstmRec& d, const ArenaBases& ab, hipStream_t st, bool fwd) {
  const dim3 grid(H / 64, (d.B + 15) / 16, d.G);
  if (fwd) hipLaunchKernelGGL((lstm_fwd_cluster_kernel<H>), grid, dim3(256), 0, st, d, ab);
  else hipLaunchKernelGGL((lstm_bwd_cluster_kernel<H>), grid, dim3(256), 0, st, d, ab);
}

// "unwritten" marks: frames [t0, t1) of every sequence of h [rows][T][H] (16-byte chunks)
__global__ void lstm_mark_kernel(uint4* h, int64_t rows, int T, int H8, int t0, int t1) {
  const int64_t per = (int64_t)(t1 - t0) * H8, n = rows * per;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / per, rem = i - row * per;
    h[(row * T + t0) * H8 + rem] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
  }
}

void launch_lstm_cluster(const LstmRec& d, const ArenaBases& ab, hipStream_t st, bool fwd) {
  // the exchanged arrays start out "unwritten": the frames of h this launch produces, all of dgates before the backward
  if (fwd) {
    const int t0 = d.t0, t1 = d.t1 > 0 ? d.t1 : d.T;
    const int64_t n = (int64_t)d.G * d.B * (t1 - t0) * (d.H / 8);
    hipLaunchKernelGGL(lstm_mark_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0, st,
                       reinterpret_cast<uint4*>(rp(ab, d.h)), (int64_t)d.G * d.B, d.T, d.H / 8, t0, t1);
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
