// bf16 variant of the persistent LSTM recurrence (see kernels.hip for the fp32 parity version and the design notes).
// v_mfma_f32_16x16x32_bf16: A[i=l&15][k=8*(l>>4)+e], B[k=8*(l>>4)+e][j=l&15], D[row=4*(l>>4)+r][col=l&15], fp32 accumulate.
// Arithmetic contract of bf16 mode: the recurrent operands (h_{t-1}, W_hh, and dgates_t in the backward) are rounded to
// bf16; gate pre-activations from the input GEMM, the cell state and all gate math stay fp32.
#include <hip/hip_runtime.h>
#include "sefd_desc.h"
#include "dev_common.h"

namespace sefd {

__device__ __forceinline__ uint32_t pack2(float a, float b) { return pack_bf16x2(a, b); }

// H is a compile-time constant (32 / 64 / 96 / 128): with run-time trip counts hipcc guards every MFMA with a branch and
// drains the software-prefetched loads before the matrix section, which serialises the 483-step loop.
template <int HMAX>
__global__ __launch_bounds__(HMAX * 4) void lstm_fwd_bf16_kernel(const LstmRec d, const ArenaBases ab) {
  extern __shared__ __attribute__((aligned(16))) uint16_t ldsh[];
  constexpr int H = HMAX;
  const int T = d.T;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = blockIdx.y, b0 = blockIdx.x * 16;
  const float* whh = reinterpret_cast<const float*>(rp(ab, d.whh[g % d.nset]));
  const float* gx = reinterpret_cast<const float*>(rp(ab, d.gx)) + d.gx_goff[g];
  uint16_t* hout = reinterpret_cast<uint16_t*>(rp(ab, d.h));
  float* gates = reinterpret_cast<float*>(rp(ab, d.gates));
  float* cs = reinterpret_cast<float*>(rp(ab, d.c));
  constexpr int hs = H + 8;                       // LDS row stride in bf16 elements (16-byte aligned rows, rotating 16-B slots)
  constexpr int KS = H / 32;
  const int unit = 16 * w + (lane & 15);
  const int kq = lane >> 4;

  uint4 wreg[4][HMAX / 32];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int ks = 0; ks < HMAX / 32; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ks < KS) {
        const float* p = whh + (int64_t)(q * H + unit) * H + 32 * ks + 8 * kq;
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v = make_uint4(pack2(a.x, a.y), pack2(a.z, a.w), pack2(b.x, b.y), pack2(b.z, b.w));
      }
      wreg[q][ks] = v;
    }
  for (int i = threadIdx.x; i < 2 * 16 * hs; i += blockDim.x) ldsh[i] = 0;
  __syncthreads();

  bool rvalid[4];
  int64_t rowbt[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + 4 * kq + r;
    rvalid[r] = b < d.B;
    rowbt[r] = (int64_t)(rvalid[r] ? b : 0) * T;
  }
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  // gate pre-activations are prefetched TWO steps ahead: one step (~1 us) does not cover an HBM round trip under load
  float gxv[4][4], gx1[4][4], gx2[4][4];
  auto load_gx = [&](int t, float (&dst)[4][4]) {
    const int tc = t < T ? t : T - 1;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[q][r] = gx[(rowbt[r] + tc) * d.gx_ld + q * H + unit];   // rows >= B alias row 0, never stored
  };
  load_gx(0, gxv);
  load_gx(1, gx1);
  const int64_t GBT = (int64_t)d.B * T;
  for (int t = 0; t < T; ++t) {
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][r] = gxv[q][r];
    load_gx(t + 2, gx2);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) { gxv[q][r] = gx1[q][r]; gx1[q][r] = gx2[q][r]; }
    const int hp = (t & 1) * 16 * hs;            // LDS offsets, not pointers: keeps the accesses in the LDS address space (ds_*, not flat_*)
    if (t > 0) {
#pragma unroll
      for (int ks = 0; ks < HMAX / 32; ++ks) {
        if (ks < KS) {
          const uint4 a = *reinterpret_cast<const uint4*>(&ldsh[hp + (lane & 15) * hs + 32 * ks + 8 * kq]);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, wreg[q][ks]), acc[q], 0, 0, 0);
        }
      }
    }
    const int hn = ((t + 1) & 1) * 16 * hs;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ig = sigmoidf_(acc[0][r]), fg = sigmoidf_(acc[1][r]), gg = tanhf_(acc[2][r]), og = sigmoidf_(acc[3][r]);
      c[r] = fg * c[r] + ig * gg;
      const uint16_t hb = f2bf(og * tanhf_(c[r]));
      ldsh[hn + (4 * kq + r) * hs + unit] = hb;
      if (rvalid[r]) {
        const int64_t row = (int64_t)g * GBT + rowbt[r] + t;
        hout[row * H + unit] = hb;
        gates[row * 4 * H + unit] = ig;
        gates[row * 4 * H + H + unit] = fg;
        gates[row * 4 * H + 2 * H + unit] = gg;
        gates[row * 4 * H + 3 * H + unit] = og;
        cs[row * H + unit] = c[r];
      }
    }
    lds_barrier();
  }
}

template <int HMAX>
__global__ __launch_bounds__(HMAX * 4) void lstm_bwd_bf16_kernel(const LstmRec d, const ArenaBases ab) {
  extern __shared__ __attribute__((aligned(16))) uint16_t ldsh[];
  constexpr int H = HMAX;
  const int T = d.T;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = blockIdx.y, b0 = blockIdx.x * 16;
  const float* whh = reinterpret_cast<const float*>(rp(ab, d.whh[g % d.nset]));
  const float* gates = reinterpret_cast<const float*>(rp(ab, d.gates));
  const float* cs = reinterpret_cast<const float*>(rp(ab, d.c));
  const float* dh = reinterpret_cast<const float*>(rp(ab, d.dh));
  char* dgo = rp(ab, d.dgates);
  constexpr int gs = 4 * H + 8;
  const int unit = 16 * w + (lane & 15);
  const int kq = lane >> 4;
  constexpr int KS = 4 * H / 32;

  uint4 wreg[HMAX / 8];                       // B[k = n][j = unit] = W_hh[n][unit], n = 32*ks + 8*kq + e
#pragma unroll
  for (int ks = 0; ks < HMAX / 8; ++ks) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (ks < KS) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = whh[(int64_t)(32 * ks + 8 * kq + e) * H + unit];
      v = make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
    }
    wreg[ks] = v;
  }
  bool rvalid[4];
  int64_t rowbt[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + 4 * kq + r;
    rvalid[r] = b < d.B;
    rowbt[r] = (int64_t)(rvalid[r] ? b : 0) * T;
  }
  const int64_t GBT = (int64_t)d.B * T;
  float dcarry[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 dhrec = {0.f, 0.f, 0.f, 0.f};
  // software prefetch: the saved activations of step t-1 are fetched while step t computes (all addresses are known)
  float pg[4][4], pct[4], pcp[4], pdh[4];       // gates i,f,g,o ; c_t ; c_{t-1} ; upstream dh   for the current step
  float ng[4][4], ncp[4], ndh[4];               // the same for step t-1
  float mg[4][4], mcp[4], mdh[4];               // ... and for step t-2 (two steps of look-ahead cover the HBM latency)
  auto fetch = [&](int t_, float (&g4)[4][4], float (&cp)[4], float (&dhv)[4]) {
    const int t = t_ > 0 ? t_ : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = (int64_t)g * GBT + rowbt[r] + t;
#pragma unroll
      for (int q = 0; q < 4; ++q) g4[r][q] = gates[row * 4 * H + q * H + unit];      // rows >= B alias row 0, results unused
      cp[r] = cs[(row - (t > 0 ? 1 : 0)) * H + unit];
      dhv[r] = dh[row * H + unit];
    }
  };
  fetch(T - 1, pg, pcp, pdh);
  fetch(T - 2, ng, ncp, ndh);
#pragma unroll
  for (int r = 0; r < 4; ++r) pct[r] = cs[((int64_t)g * GBT + rowbt[r] + T - 1) * H + unit];
  for (int t = T - 1; t >= 0; --t) {
    fetch(t - 2, mg, mcp, mdh);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float di = 0.f, df = 0.f, dg = 0.f, dog = 0.f;
      if (rvalid[r]) {
        const float ig = pg[r][0], fg = pg[r][1], gg = pg[r][2], og = pg[r][3];
        const float ct = pct[r];
        const float cp = t > 0 ? pcp[r] : 0.f;
        const float dht = pdh[r] + dhrec[r];
        const float tc = tanhf_(ct);
        dog = dht * tc * og * (1.f - og);
        const float dc = dht * og * (1.f - tc * tc) + dcarry[r];
        di = dc * gg * ig * (1.f - ig);
        df = dc * cp * fg * (1.f - fg);
        dg = dc * ig * (1.f - gg * gg);
        dcarry[r] = dc * fg;
      }
      uint16_t* lrow = ldsh + (4 * kq + r) * gs;
      lrow[unit] = f2bf(di); lrow[H + unit] = f2bf(df); lrow[2 * H + unit] = f2bf(dg); lrow[3 * H + unit] = f2bf(dog);
    }
    lds_barrier();
    // dgates_t leave through the LDS tile with 16-byte stores (2 per thread) instead of sixteen 2-byte stores per lane
    {
      constexpr int CPR = 4 * H / 8;                       // 16-byte chunks per row
      for (int ch = threadIdx.x; ch < 16 * CPR; ch += HMAX * 4) {
        const int row = ch / CPR, c8 = ch - row * CPR;
        const int b = b0 + row;
        if (b < d.B) {
          const uint4 v = *reinterpret_cast<const uint4*>(ldsh + row * gs + c8 * 8);
          uint16_t* dst = reinterpret_cast<uint16_t*>(dgo) + d.gx_goff[g] + ((int64_t)b * T + t) * d.gx_ld + c8 * 8;
          *reinterpret_cast<uint4*>(dst) = v;
        }
      }
    }
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    if (t > 0) {
#pragma unroll
      for (int ks = 0; ks < HMAX / 8; ks += 2) {
        if (ks < KS) {
          const uint4 x0 = *reinterpret_cast<const uint4*>(ldsh + (lane & 15) * gs + 32 * ks + 8 * kq);
          const uint4 x1 = *reinterpret_cast<const uint4*>(ldsh + (lane & 15) * gs + 32 * (ks + 1) + 8 * kq);
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x0), __builtin_bit_cast(bf16x8, wreg[ks]), a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x1), __builtin_bit_cast(bf16x8, wreg[ks + 1]), a1, 0, 0, 0);
        }
      }
    }
    dhrec = a0 + a1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pct[r] = pcp[r]; pcp[r] = ncp[r]; pdh[r] = ndh[r];
      ncp[r] = mcp[r]; ndh[r] = mdh[r];
#pragma unroll
      for (int q = 0; q < 4; ++q) { pg[r][q] = ng[r][q]; ng[r][q] = mg[r][q]; }
    }
    lds_barrier();
  }
}

template <int HMAX>
static void launch_t(const LstmRec& d, const ArenaBases& ab, hipStream_t st, bool fwd) {
  dim3 grid((d.B + 15) / 16, d.G);
  dim3 block(64 * (d.H / 16));
  if (fwd) hipLaunchKernelGGL((lstm_fwd_bf16_kernel<HMAX>), grid, block, 2 * 16 * (d.H + 8) * sizeof(uint16_t), st, d, ab);
  else hipLaunchKernelGGL((lstm_bwd_bf16_kernel<HMAX>), grid, block, 16 * (4 * d.H + 8) * sizeof(uint16_t), st, d, ab);
}

void launch_lstm_bf16(const LstmRec& d, const ArenaBases& ab, hipStream_t st, bool fwd) {
  switch (d.H) {          // planner guarantees H % 32 == 0, H <= 128 and gdt == bf16 in bf16 mode
    case 32: launch_t<32>(d, ab, st, fwd); break;
    case 64: launch_t<64>(d, ab, st, fwd); break;
    case 96: launch_t<96>(d, ab, st, fwd); break;
    default: launch_t<128>(d, ab, st, fwd); break;
  }
}

}  // namespace sefd
