// LSTM recurrence over THOUSANDS of independent sequences (FullSubNet's sub-band model: B * 257 rows, H = 384; reference
// tools_for_model.py:726-795 with nn.LSTM): a throughput problem, bound by the HBM traffic of the gate slabs (fp32 pre-activations in,
// i/f/g/o out: ~230 MB per frame at 16 448 rows), not by the matrix pipe (19 GFLOP per frame).  The cluster kernels (lstm_cluster.hip)
// pay an inter-workgroup hand-off per 16-row tile here; these kernels have NO inter-workgroup dependence:
//   * a workgroup (4 waves) owns 16 * MT sequences and all H hidden units for the whole sequence;
//   * forward: h_{t-1} of its rows lives in LDS (bf16, ping-pong); per block of 16 units a wave loads the 4 x H/32 B fragments of
//     W_hh straight from the packed bf16 copy in L2 into registers, multiplies all MT row tiles, adds the input GEMM's
//     pre-activations, does the cell update lane-locally and writes gates / c to memory and h_t to the other LDS buffer; one LDS
//     barrier per frame, then h_t leaves as whole 16-byte chunks;
//   * backward: the cells a lane owns in the accumulator layout of dh_{t-1} = dgates_t . W_hh are the cells whose gate gradients
//     it computes, so the recurrent gradient and the cell-state carry never leave its registers; dgates_t (16 MT x 4H, bf16) goes
//     to LDS (the GEMM's A operand) and to memory (the weight / input gradient GEMMs read it), W_hh^T fragments stream from L2.
// Same descriptor, buffers, gate-column order (unit-major) and arithmetic contract as the other LSTM kernels (LstmRec, impl == 1).
// Launch forms (round 5, profiles/r05_tuning_notes.md section 8): one workgroup per row block for the whole sequence (lstm_fwd_rows_kernel /
// lstm_bwd_rows_kernel), or - when a layer has more row blocks than the chip has CUs - ticket-drawn JOBS that keep every CU busy to the end:
// lstm_fwd_rows_pair_kernel runs two stacked layers as (layer, time chunk, row block) jobs behind agent-scope flags, lstm_bwd_rows_jobs_kernel one
// layer's backward as (time chunk, row block) jobs that hand the register-resident carry through memory.  Same arithmetic, bit-identical results.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include "sefd_desc.h"
#include "tuning.h"
#include "dev_common.h"

#pragma clang fp contract(off)

namespace sefd {

namespace {
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 exp2_2(f32x2 x) { return f32x2{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }
__device__ __forceinline__ f32x2 rcp_2(f32x2 x) { return f32x2{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
__device__ __forceinline__ f32x2 sigmoid2(f32x2 x) { return rcp_2(exp2_2(x * -1.4426950408889634f) + 1.f); }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 tanh2(f32x2 x) { return fma2(rcp_2(exp2_2(x * 2.8853900817779268f) + 1.f), f32x2{-2.f, -2.f}, f32x2{1.f, 1.f}); }
template <bool G16> struct GateRaw;
template <> struct GateRaw<true> {
  typedef uint2 type;
  static __device__ __forceinline__ float4 cvt(uint2 r) { return make_float4(bf2f(r.x & 0xffff), bf2f(r.x >> 16), bf2f(r.y & 0xffff), bf2f(r.y >> 16)); }
};
template <> struct GateRaw<false> {
  typedef float4 type;
  static __device__ __forceinline__ float4 cvt(float4 r) { return r; }
};
// four gate values of one cell from / to a slab of dtype fp32 or bf16 (element offset o, a multiple of 4)
template <bool G16>
__device__ __forceinline__ float4 ld_gate4(const char* base, int64_t o) {
  if constexpr (G16) {
    const uint2 r = *reinterpret_cast<const uint2*>(base + o * 2);
    return make_float4(bf2f(r.x & 0xffff), bf2f(r.x >> 16), bf2f(r.y & 0xffff), bf2f(r.y >> 16));
  } else {
    return *reinterpret_cast<const float4*>(base + o * 4);
  }
}
template <bool G16>
__device__ __forceinline__ void st_gate4(char* base, int64_t o, float a, float b, float c, float e) {
  if constexpr (G16) *reinterpret_cast<uint2*>(base + o * 2) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, e));
  else *reinterpret_cast<float4*>(base + o * 4) = make_float4(a, b, c, e);
}
// Reproducibility note (round 3; found with tests/test_gpu_model.py::test_two_stream_schedule_equals_program_order and by re-running the
// single LSTM_FWD op on identical inputs): the forward kernel used to be not bit-reproducible run to run.  Signature: the FORGET gate
// (accumulator q = 1, mt = 0, register .x) of local row 12 of a workgroup (lanes 48 .. 63 = the last quarter-wave) came out different in
// ~2.5e-4 of the cells of a frame, on the second wave of a SIMD; cell-state error up to 0.015 (|c| <= 0.85).  Not a waitcnt problem
// (-mllvm -amdgpu-waitcnt-forcezero: unchanged), not the saved-cell-state read (c kept in registers: unchanged).  Two changes:
//  * all accumulators pass through ONE asm statement behind 16 wait states before the epilogue reads them (hipcc of ROCm 7.2 scheduled the
//    first packed add on an MFMA result 8 wait states behind the v_mfma_f32_16x16x32_bf16 that wrote it): rate 2.5e-4 -> 1e-5;
//  * with the fused input projection the bias rides in the accumulators' initial value instead of being added to the MFMA results with
//    v_pk_add_f32 (op_sel broadcast of a bias register) in the epilogue: rate -> 0 (5 re-runs of both layers, 16 whole steps in both
//    schedules: identical bits).  Which of the packed add's operands was read early was not pinned down; the sequence is avoided.
template <int MT>
__device__ __forceinline__ void mfma_settle(f32x4 (&acc)[MT][4]) {
  static_assert(MT >= 1 && MT <= 5, "one asm statement takes every accumulator");
  // ONE statement with all accumulators as in/out operands: separate statements were scheduled apart (the nops behind the first MFMA only)
  if constexpr (MT == 1)
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]));
  else if constexpr (MT == 2)
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]),
                 "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]));
  else if constexpr (MT == 3)
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]),
                 "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]),
                 "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]));
  else {
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]),
                 "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]),
                 "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]));
    if constexpr (MT == 4)
      asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[3][2]), "+v"(acc[3][3]));
    else
      asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[3][2]), "+v"(acc[3][3]),
                   "+v"(acc[4][0]), "+v"(acc[4][1]), "+v"(acc[4][2]), "+v"(acc[4][3]));
  }
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------- forward
// NW = 8 waves per workgroup, two per SIMD: the loads of one wave (gate pre-activations from HBM, weight fragments from L2) wait under
// the other's MFMAs (measured on FullSubNet's sub-band layers, ms per launch: 4 waves 19.7 forward / 18.5 backward)
// XF: input features whose projection is fused into the recurrence: 0 (gx holds the hoisted GEMM's pre-activations), 32 (one extra k-step,
// A fragments of x_t in registers) or H (x_t = the layer below's h_t: a third LDS tile, H/32 more k-steps per unit block)
// The body takes the row block as a parameter: one launch per layer runs it on block blockIdx.x (lstm_fwd_rows_kernel); the two-layer launch
// (lstm_fwd_rows_pair_kernel below) runs the first layer's blocks and, behind a per-block flag, the second layer's on the same grid.
template <int H, int MT, int NW, bool G16, int XF>
__device__ __forceinline__ void lstm_fwd_rows_body(const LstmRec& d, const ArenaBases& ab, const int blk, uint16_t* hl /* LDS [2][RB][HS] (+ [RB][HS] for x_t when XH) */,
                                                   const int tb, const int te /* frames [tb, te); tb > 0 resumes from the stored h / c of frame tb - 1 */) {
  constexpr int KS = H / 32, NUB = H / 16, RB = 16 * MT, HS = H + 8, KC = (KS % 3 == 0 ? 3 : 4) * (NW == 4 ? 2 : 1), NTHR = NW * 64;
  constexpr bool XK = XF > 0, X32 = XF == 32, XH = XF == H && H != 32;
  static_assert(XF == 0 || X32 || XH, "fused input width");
  // CL: the cell state of the workgroup's rows stays in LDS between frames (fp32 [RB][CS]; a lane re-reads next frame exactly the cells it
  // wrote - same wave, same lane, no barrier) instead of being read back from the c slab in memory: only where the tiles leave room, i.e. not
  // next to the x_t tile of a stacked layer.  CS = H + 4: the four row groups of a wave instruction (rows 4 kq + r) fall into different banks.
  constexpr int CS = H + 4;
  constexpr bool CL = X32 && (size_t)2 * RB * HS * 2 + (size_t)RB * CS * 4 <= 160 * 1024;
  const int T = d.T;
  const int64_t rows = d.B;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row0 = (int64_t)blk * RB;
  const char* gx = rp(ab, d.gx);
  char* gates = rp(ab, d.gates);
  float* cs = reinterpret_cast<float*>(rp(ab, d.c));
  uint16_t* hout = reinterpret_cast<uint16_t*>(rp(ab, d.h));
  const uint16_t* wp = reinterpret_cast<const uint16_t*>(rp(ab, d.wpk_f));   // [4H][H] bf16, row = gate column 4 * unit + q
  const int kq = lane >> 4, ln = lane & 15;
  const int64_t gx_ld = d.gx_ld;
  const uint16_t* xin = XK ? reinterpret_cast<const uint16_t*>(rp(ab, d.xin)) : nullptr;      // [T][rows][32] bf16
  const uint16_t* wx = XK ? reinterpret_cast<const uint16_t*>(rp(ab, d.wpk_x)) : nullptr;
  const float* bias = XK ? reinterpret_cast<const float*>(rp(ab, d.bias)) : nullptr;
  uint16_t* hd = d.hd.arena >= 0 ? reinterpret_cast<uint16_t*>(rp(ab, d.hd)) : nullptr;
  const uint32_t seed0 = hd ? reinterpret_cast<const uint32_t*>(rp(ab, d.seed))[0] : 0u, seed1 = hd ? reinterpret_cast<const uint32_t*>(rp(ab, d.seed))[1] : 0u;
  uint16_t* xl = hl + 2 * RB * HS;
  float* cl = reinterpret_cast<float*>(hl + 2 * RB * HS);   // CL (never together with the x_t tile)
  if (tb == 0) {
    for (int i = tid; i < 2 * RB * HS; i += NTHR) hl[i] = 0;                // h_{-1} = 0
  } else {                                         // resume: h_{tb-1} (and, CL, c_{tb-1}) of the rows from the slabs; the other h tile is written whole in frame tb
    uint16_t* h0 = hl + (tb & 1) * RB * HS;
    for (int i = tid; i < RB * (H / 8); i += NTHR) {
      const int row = i / (H / 8), ch = i - row * (H / 8);
      const int64_t b = row0 + row;
      *reinterpret_cast<uint4*>(h0 + row * HS + 8 * ch) = *reinterpret_cast<const uint4*>(hout + ((int64_t)(tb - 1) * rows + (b < rows ? b : 0)) * H + 8 * ch);
    }
    if constexpr (CL) {
      for (int i = tid; i < RB * (H / 4); i += NTHR) {
        const int row = i / (H / 4), ch = i - row * (H / 4);
        const int64_t b = row0 + row;
        *reinterpret_cast<float4*>(cl + row * CS + 4 * ch) = *reinterpret_cast<const float4*>(cs + ((int64_t)(tb - 1) * rows + (b < rows ? b : 0)) * H + 4 * ch);
      }
    }
  }
  auto fill_x = [&](int tt) {                      // x_tt of the workgroup's rows -> LDS (16-byte chunks)
    if constexpr (XH) {
      for (int i = tid; i < RB * (H / 8); i += NTHR) {
        const int row = i / (H / 8), ch = i - row * (H / 8);
        const int64_t b = row0 + row;
        *reinterpret_cast<uint4*>(xl + row * HS + 8 * ch) = *reinterpret_cast<const uint4*>(xin + ((int64_t)tt * rows + (b < rows ? b : 0)) * H + 8 * ch);
      }
    }
  };
  fill_x(tb);
  __syncthreads();
  bool rvalid[MT][4];
  int64_t rrow[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t b = row0 + 16 * mt + 4 * kq + r;
      rvalid[mt][r] = b < rows;
      rrow[mt][r] = rvalid[mt][r] ? b : 0;
    }
  for (int t = tb; t < te; ++t) {
    const uint16_t* hc = hl + (t & 1) * RB * HS;
    uint16_t* hn = hl + ((t + 1) & 1) * RB * HS;
    uint4 xa[X32 ? MT : 1];                         // fused input projection: A fragments of x_t (one k-step of 32 features), all unit blocks
    if constexpr (X32) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int64_t b = row0 + 16 * mt + ln;
        xa[mt] = *reinterpret_cast<const uint4*>(xin + ((int64_t)t * rows + (b < rows ? b : 0)) * 32 + 8 * kq);
      }
    }
    for (int ub = w; ub < NUB; ub += NW) {
      const int unit = 16 * ub + ln;
      typename GateRaw<G16>::type gxr[XK ? 1 : MT][4];      // raw gate pre-activations (bf16: 8 bytes per cell), converted in the epilogue
      float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
      uint4 bx[4];
      float cpv[MT][4];
      if constexpr (XK) bias4 = *reinterpret_cast<const float4*>(bias + 4 * unit);
      if constexpr (X32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bx[q] = *reinterpret_cast<const uint4*>(wx + ((int64_t)(ub * 4 + q) * 64 + lane) * 8);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t rt = (int64_t)t * rows + rrow[mt][r];
          if constexpr (!XK) gxr[mt][r] = *reinterpret_cast<const typename GateRaw<G16>::type*>(gx + (rt * gx_ld + 4 * unit) * (G16 ? 2 : 4));
#if defined(SEFD_ROWS_DBG) && (SEFD_ROWS_DBG & 4)
          cpv[mt][r] = 0.25f;                                              // tuning: no c_{t-1} loads
#else
          if constexpr (CL) cpv[mt][r] = t > 0 ? cl[(16 * mt + 4 * kq + r) * CS + unit] : 0.f;
          else cpv[mt][r] = t > 0 ? cs[(rt - rows) * H + unit] : 0.f;
#endif
        }
      f32x4 acc[MT][4];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float b = !XK ? 0.f : q == 0 ? bias4.x : q == 1 ? bias4.y : q == 2 ? bias4.z : bias4.w;    // fused projection: the bias rides in the accumulator
          acc[mt][q] = f32x4{b, b, b, b};
        }
      if constexpr (!XK) {
        // hoisted input GEMM: its pre-activations are the accumulators' INITIAL value too - nothing is added onto fresh MFMA results in the
        // epilogue in any variant (the sequence that was not reproducible run to run, note above mfma_settle; the cause at ISA level is unconfirmed)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float4 g = GateRaw<G16>::cvt(gxr[mt][r]);
            acc[mt][0][r] = g.x; acc[mt][1][r] = g.y; acc[mt][2][r] = g.z; acc[mt][3][r] = g.w;
          }
      }
      // weight fragments in chunks of KC k-steps (fragment-major packing, sefd_desc.h rows_wf_index), A fragments from an LDS tile
      auto gemm_part = [&](const uint16_t* wbase, const uint16_t* atile) {
        uint4 bqA[KC][4];
        const uint16_t* wrow = wbase + ((int64_t)ub * KS * 4 * 64 + lane) * 8;
        auto loadc = [&](uint4 (&bq)[KC][4], int k0) {
#pragma unroll
          for (int ks = 0; ks < KC; ++ks)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#if defined(SEFD_ROWS_DBG) && (SEFD_ROWS_DBG & 1)
              if (ks > 0 || k0 > 0) { bq[ks][q] = bq[0][q]; continue; }     // tuning: one k-step's weights per unit block (wrong results)
#endif
#ifdef SEFD_ROWS_L1DBG
              int z = 0; asm volatile("" : "+s"(z));        // tuning: every weight load hits the same L1 lines (wrong results, latency probe)
              bq[ks][q] = *reinterpret_cast<const uint4*>(wbase + (((k0 + ks) * z * 4 + q) * 64 + lane) * 8);
#else
              bq[ks][q] = *reinterpret_cast<const uint4*>(wrow + ((k0 + ks) * 4 + q) * 512);
#endif
            }
        };
        auto mulc = [&](const uint4 (&bq)[KC][4], int k0) {
#pragma unroll
          for (int ks = 0; ks < KC; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const uint4 a = *reinterpret_cast<const uint4*>(atile + (16 * mt + ln) * HS + 32 * (k0 + ks) + 8 * kq);
#if defined(SEFD_ROWS_DBG) && (SEFD_ROWS_DBG & 8)
              asm volatile("" ::"v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w));   // tuning: no MFMAs (operands stay live)
#pragma unroll
              for (int q = 0; q < 4; ++q) asm volatile("" ::"v"(bq[ks][q].x), "v"(bq[ks][q].y), "v"(bq[ks][q].z), "v"(bq[ks][q].w));
#else
#pragma unroll
              for (int q = 0; q < 4; ++q)
                acc[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bq[ks][q]), acc[mt][q], 0, 0, 0);
#endif
            }
        };
        // measured (sub-band layers, ms per launch): one chunk of 3 k-steps at a time 15.6; chunks of 2 double buffered 17.5 (the second
        // buffer spills: 8 waves leave 256 registers each) - the two waves of a SIMD already overlap each other's load latency
#pragma unroll 1
        for (int k0 = 0; k0 < KS; k0 += KC) {
          loadc(bqA, k0);
          mulc(bqA, k0);
        }
      };
      if (t > 0) gemm_part(wp, hc);
      if constexpr (XH) gemm_part(wx, xl);
      if constexpr (X32) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xa[mt]), __builtin_bit_cast(bf16x8, bx[q]), acc[mt][q], 0, 0, 0);
      }
      mfma_settle(acc);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int rp2 = 0; rp2 < 4; rp2 += 2) {
#if defined(SEFD_ROWS_DBG) && (SEFD_ROWS_DBG & 16)
          const f32x2 ig = f32x2{acc[mt][0][rp2], acc[mt][0][rp2 + 1]} * 0.1f;     // tuning: no transcendentals
          const f32x2 fg = f32x2{acc[mt][1][rp2], acc[mt][1][rp2 + 1]} * 0.1f;
          const f32x2 gg = f32x2{acc[mt][2][rp2], acc[mt][2][rp2 + 1]} * 0.1f;
          const f32x2 og = f32x2{acc[mt][3][rp2], acc[mt][3][rp2 + 1]} * 0.1f;
          const f32x2 cn = fma2(fg, f32x2{cpv[mt][rp2], cpv[mt][rp2 + 1]}, ig * gg);
          const f32x2 hv = og * cn;
#else
          const f32x2 ig = sigmoid2(f32x2{acc[mt][0][rp2], acc[mt][0][rp2 + 1]});
          const f32x2 fg = sigmoid2(f32x2{acc[mt][1][rp2], acc[mt][1][rp2 + 1]});
          const f32x2 gg = tanh2(f32x2{acc[mt][2][rp2], acc[mt][2][rp2 + 1]});
          const f32x2 og = sigmoid2(f32x2{acc[mt][3][rp2], acc[mt][3][rp2 + 1]});
          const f32x2 cn = fma2(fg, f32x2{cpv[mt][rp2], cpv[mt][rp2 + 1]}, ig * gg);
          const f32x2 hv = og * tanh2(cn);
#endif
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int r = rp2 + k;
            hn[(16 * mt + 4 * kq + r) * HS + unit] = f2bf(hv[k]);
            if constexpr (CL) cl[(16 * mt + 4 * kq + r) * CS + unit] = cn[k];
#if defined(SEFD_ROWS_DBG) && (SEFD_ROWS_DBG & 2)
            asm volatile("" ::"v"(ig[k]), "v"(fg[k]), "v"(gg[k]), "v"(og[k]), "v"(cn[k]));     // tuning: no gate / cell-state stores
#else
            if (rvalid[mt][r]) {
              const int64_t rt = (int64_t)t * rows + rrow[mt][r];
              st_gate4<G16>(gates, rt * gx_ld + 4 * unit, ig[k], fg[k], gg[k], og[k]);
              cs[rt * H + unit] = cn[k];
            }
#endif
          }
        }
    }
    lds_barrier();
    // h_t leaves as 16-byte chunks: RB rows x H/8 chunks
    for (int i = tid; i < RB * (H / 8); i += NTHR) {
      const int row = i / (H / 8), ch = i - row * (H / 8);
      if (row0 + row < rows) {
        const uint4 v = *reinterpret_cast<const uint4*>(hn + row * HS + 8 * ch);
        const int64_t o = ((int64_t)t * rows + row0 + row) * H + 8 * ch;
        *reinterpret_cast<uint4*>(hout + o) = v;
        if (hd) {                                     // fused inter-layer dropout (same map as dropout_kernel)
          const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
          uint32_t ov[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            ov[e] = pack_bf16x2(bf2f(wv[e] & 0xffff) * drop_scale(seed0, seed1, d.drop_layer, d.keep, o + 2 * e),
                                bf2f(wv[e] >> 16) * drop_scale(seed0, seed1, d.drop_layer, d.keep, o + 2 * e + 1));
          *reinterpret_cast<uint4*>(hd + o) = make_uint4(ov[0], ov[1], ov[2], ov[3]);
        }
      }
    }
    if constexpr (XH) {                             // every wave is past the barrier: x_t is no longer read; x_{t+1} must be visible before the next frame
      if (t + 1 < te) fill_x(t + 1);
      lds_barrier();
    }
  }
}

template <int H, int MT, int NW, bool G16, int XF>
__global__ __launch_bounds__(NW * 64) void lstm_fwd_rows_kernel(const LstmRec d, const ArenaBases ab) {
  extern __shared__ __attribute__((aligned(16))) uint16_t hl[];
  lstm_fwd_rows_body<H, MT, NW, G16, XF>(d, ab, (int)blockIdx.x, hl, 0, d.T);
}

// ------------------------------------------------------------------------------------------------- two stacked layers, one launch
// FullSubNet's sub-band model at B = 64: 16 448 rows = 343 row blocks on 256 CUs (one block per CU: LDS) - a layer's launch is two
// dispatch rounds, the second with 87 blocks on 87 CUs, and the layer above cannot start before the last block has finished.  But block
// j of the layer above reads only what block j of the layer below wrote (its rows' h_t, after dropout), frame by frame.  One launch runs
// both layers as JOBS (layer, time chunk c of C, row block j): a workgroup draws a job number (an atomic ticket: jobs are taken in the
// order workgroups START, whatever the dispatch order is) and maps it to a job in wavefront order - L(0,.), then L(1,.) U(0,.), then
// L(2,.) U(1,.) ... U(C-1,.) - so that everything a job waits for carries a smaller number and has been taken by a running workgroup: the
// waits cannot deadlock.  L(c,j) resumes from the h / c that L(c-1,j) stored (flag); U(c,j) needs U(c-1,j) and L(c,j) (two flags).  A job
// publishes behind an agent-scope release of its stores (cdna_hip_programming.md guideline 16: drain, barrier, one lane releases and stores
// the flag; the consumer polls one word, one lane acquires, barrier, plain loads).  The waits are bounded all the same (a lost block must
// not hang the device): a workgroup whose budget ran out reports through the plan's status word like the cluster kernels do (guarded Adam
// skips, next run returns -5).  Time chunks make the jobs small against the launch: with whole-sequence jobs (C = 1) the 2 x 343 jobs of
// 4.75 / 6.5 ms pack into 17.8 ms on 256 CUs, with C = 4 into ~16 (ideal: 15.1; two launches: 22.5).
// sync[0] = ticket counter, sync[1 + (layer * C + c) * nblk + j] = flag of job (layer, c, j); zeroed by a memset node in front of every launch.
constexpr int kPairSpinBudget = 1 << 24;         // polls of ~1 us

template <int H, int MT, int NW, bool G16, int XF0>
__global__ __launch_bounds__(NW * 64) void lstm_fwd_rows_pair_kernel(const LstmRec d0, const LstmRec d1, const ArenaBases ab, unsigned* sync, const int nblk, const int C) {
  extern __shared__ __attribute__((aligned(16))) uint16_t hl[];
  // the job number travels through the first word of the dynamic LDS (a static word next to 160 KB of dynamic LDS would not fit the attribute)
  if (threadIdx.x == 0) reinterpret_cast<unsigned*>(hl)[0] = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int job = (int)reinterpret_cast<const unsigned*>(hl)[0];
  __syncthreads();                                                     // the body starts by writing the tiles
  const RowsJob rj = rows_pair_job(job, nblk, C, d0.T);               // wavefront order (sefd_desc.h)
  const int layer = rj.layer, c = rj.chunk, j = rj.block, tb = rj.tb, te = rj.te;
  unsigned* flags = sync + 1;
  if (threadIdx.x == 0) {
    int budget = kPairSpinBudget;
    auto wait = [&](int l2, int c2) {
      while (__hip_atomic_load(flags + (l2 * C + c2) * nblk + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && budget > 0) { --budget; __builtin_amdgcn_s_sleep(32); }
    };
    if (c > 0) wait(layer, c - 1);
    if (layer == 1) wait(0, c);
    if (budget <= 0) set_status(ab.status, ab.dstatus);
    if (c > 0 || layer == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (tb < te) {
    if (layer == 0) lstm_fwd_rows_body<H, MT, NW, G16, XF0>(d0, ab, j, hl, tb, te);
    else lstm_fwd_rows_body<H, MT, NW, G16, H>(d1, ab, j, hl, tb, te);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // every storing wave drains its own stores
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                 // write back this XCD's dirty lines (the job's h / hd / c among them)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(flags + (layer * C + c) * nblk + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// --------------------------------------------------------------------------------------------------------------- backward
// HV > 1: the gate columns are walked in HV parts through ONE LDS tile of 16 MT x 4H / HV (cell backward of the part's units, barrier,
// its share of the GEMM, barrier), so that 80 rows would fit (16 MT x 4H of bf16 is 245 KB at MT = 5) and the launch would need one
// dispatch round instead of two.  Tried (80 rows: 4 waves x 2 parts, 8 waves x 3 parts): both spill ~350 registers - the old and the new
// recurrent gradient and the cell-state carry are all live across the parts - so only HV = 1 with 48 rows is launched.
// UP: where the upstream gradient of h_t comes from: 0 the dh array, 1 dh x the inter-layer dropout mask, 2 the 2-output head (rank-2 update);
// 3, 4: as 0, 1 with a bf16 dh slab (LstmRec::dhdt).
// A template parameter, not a run-time flag: with a branch per cell the compiler stops batching the loads of a row tile (9.8 -> 13.3 ms).
// The body walks frames te-1 .. tb of row block blk.  te < T resumes: the recurrent gradient and the cell-state carry that frame te left (both
// live in registers across frames) come from `carry` (fp32 [blk][2][RB][H]), where the job of the frames above stored them; tb > 0 stores them.
template <int H, int MT, int NW, bool G16, int HV, int UP>
__device__ __forceinline__ void lstm_bwd_rows_body(const LstmRec& d, const ArenaBases& ab, const int blk, uint16_t* al /* LDS: dgates_t (one half of the gate
                                                   columns) of the workgroup's rows: [RB][AS] */, const int tb, const int te, float* carry) {
  constexpr int KS = 4 * H / 32, NT = H / 16 / NW, RB = 16 * MT, AS = 4 * H / HV + 8, KC = H >= 512 ? 4 : (HV > 1 ? 4 : 8);
  constexpr int NTH = NT / HV, KSH = KS / HV;       // unit tiles per wave and k-steps per half
  static_assert(NT % HV == 0, "unit tiles per half");
  const int T = d.T;
  const int64_t rows = d.B;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row0 = (int64_t)blk * RB;
  const char* gates = rp(ab, d.gates);
  const float* cs = reinterpret_cast<const float*>(rp(ab, d.c));
  constexpr bool DH16 = UP >= 3, DROP = UP == 1 || UP == 4;
  const float* dh = UP == 2 ? nullptr : reinterpret_cast<const float*>(rp(ab, d.dh));
  const uint16_t* dh16 = reinterpret_cast<const uint16_t*>(dh);
  uint16_t* dgo = reinterpret_cast<uint16_t*>(rp(ab, d.dgates));
  const uint16_t* wp = reinterpret_cast<const uint16_t*>(rp(ab, d.wpk_b));   // [H][4H] bf16: row = unit u', column = gate column (unit-major)
  const uint16_t* dyo = UP == 2 ? reinterpret_cast<const uint16_t*>(rp(ab, d.dyo)) : nullptr;
  const float* wo = UP == 2 ? reinterpret_cast<const float*>(rp(ab, d.wo)) : nullptr;
  const uint32_t seed0 = DROP ? reinterpret_cast<const uint32_t*>(rp(ab, d.seed))[0] : 0u, seed1 = DROP ? reinterpret_cast<const uint32_t*>(rp(ab, d.seed))[1] : 0u;
  const int kq = lane >> 4, ln = lane & 15;
  const int64_t gx_ld = d.gx_ld;
  // unit of (tile nt, this lane): half nt / NTH, inside the half the wave's NTH consecutive tiles
  auto unit_of = [&](int nt) { return (nt / NTH) * (H / HV) + (NTH * 16) * w + 16 * (nt % NTH) + ln; };
  f32x4 acc[MT][NT], accp[MT][NT], dcar[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dcar[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  float* cbase = carry ? carry + (int64_t)blk * 2 * RB * H : nullptr;
  if (te < T) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = (16 * mt + 4 * kq + r) * H + unit_of(nt);
          acc[mt][nt][r] = cbase[o];
          dcar[mt][nt][r] = cbase[RB * H + o];
        }
  }
  for (int t = te - 1; t >= tb; --t) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) { accp[mt][nt] = acc[mt][nt]; acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int hv = 0; hv < HV; ++hv) {
      // ---- cell backward of frame t for this lane's cells of half hv; dh_rec = the previous (later) frame's accumulators
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float4 gv[NTH][4];
        float ctv[NTH][4], cpv[NTH][4], dhv[NTH][4];
#pragma unroll
        for (int n2 = 0; n2 < NTH; ++n2)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t b = row0 + 16 * mt + 4 * kq + r;
            const int64_t rt = (int64_t)t * rows + (b < rows ? b : 0);
            const int unit = unit_of(hv * NTH + n2);
            gv[n2][r] = ld_gate4<G16>(gates, rt * gx_ld + 4 * unit);
            ctv[n2][r] = cs[rt * H + unit];
            cpv[n2][r] = t > 0 ? cs[(rt - rows) * H + unit] : 0.f;
            if constexpr (UP == 2) {                      // rank-2 upstream gradient from the 2-output head (bf16 pair per row)
              const uint32_t pr = reinterpret_cast<const uint32_t*>(dyo)[rt];
              dhv[n2][r] = bf2f(pr & 0xffff) * wo[unit] + bf2f(pr >> 16) * wo[H + unit];
            } else {
              const float up = DH16 ? bf2f(dh16[rt * H + unit]) : dh[rt * H + unit];
              if constexpr (DROP) dhv[n2][r] = up * drop_scale(seed0, seed1, d.drop_layer, d.keep, rt * H + unit);
              else dhv[n2][r] = up;
            }
          }
#pragma unroll
        for (int n2 = 0; n2 < NTH; ++n2) {
          const int nt = hv * NTH + n2;
          const int unit = unit_of(nt);
#pragma unroll
          for (int rp2 = 0; rp2 < 4; rp2 += 2) {
            const f32x2 ig = {gv[n2][rp2].x, gv[n2][rp2 + 1].x}, fg = {gv[n2][rp2].y, gv[n2][rp2 + 1].y};
            const f32x2 gg = {gv[n2][rp2].z, gv[n2][rp2 + 1].z}, og = {gv[n2][rp2].w, gv[n2][rp2 + 1].w};
            const f32x2 cp = {cpv[n2][rp2], cpv[n2][rp2 + 1]};
            const f32x2 dht = f32x2{dhv[n2][rp2], dhv[n2][rp2 + 1]} + f32x2{accp[mt][nt][rp2], accp[mt][nt][rp2 + 1]};
            const f32x2 tc = tanh2(f32x2{ctv[n2][rp2], ctv[n2][rp2 + 1]});
            const f32x2 dog = dht * tc * og * (1.f - og);
            const f32x2 dc = dht * og * (1.f - tc * tc) + f32x2{dcar[mt][nt][rp2], dcar[mt][nt][rp2 + 1]};
            const f32x2 di = dc * gg * ig * (1.f - ig);
            const f32x2 df = dc * cp * fg * (1.f - fg);
            const f32x2 dg = dc * ig * (1.f - gg * gg);
            const f32x2 dcn = dc * fg;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const int r = rp2 + k;
              const int64_t b = row0 + 16 * mt + 4 * kq + r;
              const bool v = b < rows;
              dcar[mt][nt][r] = v ? dcn[k] : 0.f;
              const uint2 pk = v ? make_uint2(pack_bf16x2(di[k], df[k]), pack_bf16x2(dg[k], dog[k])) : make_uint2(0u, 0u);
              *reinterpret_cast<uint2*>(al + (16 * mt + 4 * kq + r) * AS + 4 * (unit - hv * (H / HV))) = pk;
              if (v) *reinterpret_cast<uint2*>(dgo + ((int64_t)t * rows + b) * gx_ld + 4 * unit) = pk;
            }
          }
        }
      }
      lds_barrier();
      // ---- dh_{t-1}[rows, this wave's units] += dgates_t[:, half hv] . W_hh[half hv, :]
      if (t > 0) {
        uint4 bqA[KC][NT];
        auto loadc = [&](uint4 (&bq)[KC][NT], int k0) {
#pragma unroll
          for (int ks = 0; ks < KC; ++ks)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)                       // fragment-major packing (sefd_desc.h rows_wb_index)
            {
#ifdef SEFD_ROWS_L1DBG
              int z = 0; asm volatile("" : "+s"(z));
              bq[ks][nt] = *reinterpret_cast<const uint4*>(wp + (((int64_t)(nt * KS + (k0 + ks) * z)) * 64 + lane) * 8);
#else
              bq[ks][nt] = *reinterpret_cast<const uint4*>(wp + (((int64_t)(unit_of(nt) >> 4) * KS + k0 + ks) * 64 + lane) * 8);
#endif
            }
        };
        auto mulc = [&](const uint4 (&bq)[KC][NT], int k0) {
#pragma unroll
          for (int ks = 0; ks < KC; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const uint4 a = *reinterpret_cast<const uint4*>(al + (16 * mt + ln) * AS + 32 * (k0 + ks - hv * KSH) + 8 * kq);
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bq[ks][nt]), acc[mt][nt], 0, 0, 0);
            }
        };
#pragma unroll 1
        for (int k0 = hv * KSH; k0 < (hv + 1) * KSH; k0 += KC) {
          loadc(bqA, k0);
          mulc(bqA, k0);
        }
      }
      lds_barrier();
    }
  }
  if (tb > 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = (16 * mt + 4 * kq + r) * H + unit_of(nt);
          cbase[o] = acc[mt][nt][r];
          cbase[RB * H + o] = dcar[mt][nt][r];
        }
  }
}

template <int H, int MT, int NW, bool G16, int HV, int UP>
__global__ __launch_bounds__(NW * 64) void lstm_bwd_rows_kernel(const LstmRec d, const ArenaBases ab) {
  extern __shared__ __attribute__((aligned(16))) uint16_t al[];
  lstm_bwd_rows_body<H, MT, NW, G16, HV, UP>(d, ab, (int)blockIdx.x, al, 0, d.T, nullptr);
}

// One layer's backward recurrence as (time chunk c of C, row block j) jobs, drawn by ticket in chunk-major order (the lstm_fwd_rows_pair_kernel
// scheme with one layer): 343 blocks on 256 CUs are two dispatch rounds as one launch of whole-sequence workgroups, the second with 87 busy CUs;
// as C x 343 jobs every CU stays busy until the last chunk.  Job (c, j) covers frames [T - (c + 1) clen, T - c clen) and waits for job (c - 1, j)
// (smaller ticket: no deadlock; bounded wait, status word), which left the recurrent gradient and the cell-state carry in `carry`.
template <int H, int MT, int NW, bool G16, int HV, int UP>
__global__ __launch_bounds__(NW * 64) void lstm_bwd_rows_jobs_kernel(const LstmRec d, const ArenaBases ab, unsigned* sync, float* carry, const int nblk, const int C) {
  extern __shared__ __attribute__((aligned(16))) uint16_t al[];
  if (threadIdx.x == 0) reinterpret_cast<unsigned*>(al)[0] = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int job = (int)reinterpret_cast<const unsigned*>(al)[0];
  __syncthreads();
  const RowsJob rj = rows_bwd_job(job, nblk, C, d.T);                 // chunk-major order, last frames first (sefd_desc.h)
  const int c = rj.chunk, j = rj.block, tb = rj.tb, te = rj.te;
  unsigned* flags = sync + 1;
  if (c > 0) {
    if (threadIdx.x == 0) {
      int budget = kPairSpinBudget;
      while (__hip_atomic_load(flags + (c - 1) * nblk + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && budget > 0) { --budget; __builtin_amdgcn_s_sleep(32); }
      if (budget <= 0) set_status(ab.status, ab.dstatus);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  if (tb < te) lstm_bwd_rows_body<H, MT, NW, G16, HV, UP>(d, ab, j, al, tb, te, carry);
  if (c + 1 < C) {                                                     // a later job reads the carry (the dgates / dh stores need no hand-over)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(flags + c * nblk + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------ launch
bool lstm_rows_supported(int H) { return H == 256 || H == 384 || H == 512; }

// ticket + flag words of the job kernels
static unsigned* pair_sync(hipStream_t st, size_t words) {
  // one buffer per stream (launches on one stream are ordered); zeroed in front of every launch by the caller
  static std::mutex mu;
  static std::unordered_map<StreamKey, unsigned*, StreamKeyHash> map;        // per (device, stream): dev_common.h StreamKey
  constexpr size_t kWords = 1 << 18;
  if (words > kWords) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  const StreamKey key = stream_key(st);
  auto it = map.find(key);
  if (it != map.end()) return it->second;
  unsigned* p = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&p), kWords * sizeof(unsigned)) != hipSuccess) return nullptr;
  map.emplace(key, p);
  return p;
}

// carry of the chunked backward jobs (recurrent gradient + cell-state carry of every row block): one buffer per stream, grown on demand
static float* rows_carry(hipStream_t st, size_t floats) {
  static std::mutex mu;
  static std::unordered_map<StreamKey, std::pair<float*, size_t>, StreamKeyHash> map;
  std::lock_guard<std::mutex> lk(mu);
  auto& e = map[stream_key(st)];
  if (e.second >= floats) return e.first;
  if (e.first) { (void)hipStreamSynchronize(st); (void)hipFree(e.first); e = {nullptr, 0}; }
  float* p = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&p), floats * sizeof(float)) != hipSuccess) return nullptr;
  e = {p, floats};
  return p;
}

template <int H, int MT, bool G16, int NW = 8, int HV = 1>
static void launch_r2(const LstmRec& d, const ArenaBases& ab, hipStream_t st, bool fwd) {
  const unsigned grid = (unsigned)((d.B + 16 * MT - 1) / (16 * MT));
  if (fwd) {
    const size_t sh = (size_t)2 * 16 * MT * (H + 8) * 2;
    const size_t shc = sh + (size_t)16 * MT * (H + 4) * 4 <= 160 * 1024 ? sh + (size_t)16 * MT * (H + 4) * 4 : sh;   // + the cell-state tile (kernel: CL)
    if (d.xfeat == H && H != 32) {
      const size_t sh3 = (size_t)3 * 16 * MT * (H + 8) * 2;
      static bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_fwd_rows_kernel<H, MT, NW, G16, H>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }();
      (void)once;
      hipLaunchKernelGGL((lstm_fwd_rows_kernel<H, MT, NW, G16, H>), dim3(grid), dim3(NW * 64), sh3, st, d, ab);
    } else if (d.xfeat == 32) {
      static bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_fwd_rows_kernel<H, MT, NW, G16, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }();
      (void)once;
      hipLaunchKernelGGL((lstm_fwd_rows_kernel<H, MT, NW, G16, 32>), dim3(grid), dim3(NW * 64), shc, st, d, ab);
    } else {
      static bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_fwd_rows_kernel<H, MT, NW, G16, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }();
      (void)once;
      hipLaunchKernelGGL((lstm_fwd_rows_kernel<H, MT, NW, G16, 0>), dim3(grid), dim3(NW * 64), sh, st, d, ab);
    }
  } else {
    const size_t sh = (size_t)16 * MT * (4 * H / HV + 8) * 2;
    const int up = d.no == 2 ? 2 : (d.seed.arena >= 0 && d.keep < 1.f) ? 1 : 0;
    // more blocks than CUs: (time chunk, row block) jobs (SEFD_ROWS_BWD_CHUNKS, default 5, at least 16 frames each; 1 = whole-sequence workgroups)
    // (an explicit setting applies to any grid: the equivalence test runs it on a few blocks)
    const int cenv = tune_str("ROWS_BWD_CHUNKS") ? atoi(tune_str("ROWS_BWD_CHUNKS")) : (grid > 256 ? 5 : 1);
    const int C = std::max(1, std::min(std::min(cenv, 16), d.T / 16));
    auto go = [&](auto upc) {
      constexpr int UPC = decltype(upc)::value;
      if (C > 1) {
        const size_t words = 1 + (size_t)C * grid;
        unsigned* sync = pair_sync(st, words);
        float* carry = rows_carry(st, (size_t)grid * 2 * 16 * MT * H);
        if (sync && carry && hipMemsetAsync(sync, 0, words * sizeof(unsigned), st) == hipSuccess) {
          static bool once2 = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_rows_jobs_kernel<H, MT, NW, G16, HV, UPC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }();
          (void)once2;
          hipLaunchKernelGGL((lstm_bwd_rows_jobs_kernel<H, MT, NW, G16, HV, UPC>), dim3(C * grid), dim3(NW * 64), sh, st, d, ab, sync, carry, (int)grid, C);
          return;
        }
      }
      static bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_rows_kernel<H, MT, NW, G16, HV, UPC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }();
      (void)once;
      hipLaunchKernelGGL((lstm_bwd_rows_kernel<H, MT, NW, G16, HV, UPC>), dim3(grid), dim3(NW * 64), sh, st, d, ab);
    };
    const bool dh16 = up != 2 && d.dhdt == DT_BF16;
    if (up == 2) go(std::integral_constant<int, 2>{});
    else if (up == 1) { if (dh16) go(std::integral_constant<int, 4>{}); else go(std::integral_constant<int, 1>{}); }
    else { if (dh16) go(std::integral_constant<int, 3>{}); else go(std::integral_constant<int, 0>{}); }
  }
}
template <int H, int MT>
static void launch_r(const LstmRec& d, const ArenaBases& ab, hipStream_t st, bool fwd) {
  if (d.gxdt == DT_BF16) launch_r2<H, MT, true>(d, ab, st, fwd);
  else launch_r2<H, MT, false>(d, ab, st, fwd);
}

// ---- two stacked forward layers in one launch (lstm_fwd_rows_pair_kernel)
template <int H, int MT, bool G16>
static bool launch_pair2(const LstmRec& d0, const LstmRec& d1, const ArenaBases& ab, hipStream_t st) {
  constexpr int NW = 8;
  const unsigned nblk = (unsigned)((d0.B + 16 * MT - 1) / (16 * MT));
  // time chunks per layer: SEFD_ROWS_PAIR_CHUNKS, default 5, at least 16 frames each
  const int cenv = tune_str("ROWS_PAIR_CHUNKS") ? atoi(tune_str("ROWS_PAIR_CHUNKS")) : 5;
  const int C = std::max(1, std::min(std::min(cenv, 16), d0.T / 16));
  const size_t words = 1 + (size_t)2 * C * nblk;
  unsigned* sync = pair_sync(st, words);
  if (!sync) return false;
  const size_t sh2c = (size_t)2 * 16 * MT * (H + 8) * 2 + (size_t)16 * MT * (H + 4) * 4;      // lower layer: h tiles + cell-state tile (if it fits: CL)
  const size_t sh3 = std::max((size_t)3 * 16 * MT * (H + 8) * 2, sh2c <= 160 * 1024 ? sh2c : (size_t)0);
  static bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_fwd_rows_pair_kernel<H, MT, NW, G16, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }();
  (void)once;
  if (hipMemsetAsync(sync, 0, words * sizeof(unsigned), st) != hipSuccess) return false;
  hipLaunchKernelGGL((lstm_fwd_rows_pair_kernel<H, MT, NW, G16, 32>), dim3(2 * C * nblk), dim3(NW * 64), sh3, st, d0, d1, ab, sync, (int)nblk, C);
  return true;
}

// d0, d1: consecutive OP_LSTM_FWD descriptors of one stream (impl == 1).  True when both were issued as ONE launch; false: launch them one by one.
bool launch_lstm_rows_pair(const LstmRec& d0, const LstmRec& d1, const ArenaBases& ab, hipStream_t st) {
  // read per call: the equivalence test flips it between two steps of one model
  if ((tune_str("ROWS_PAIR") && atoi(tune_str("ROWS_PAIR")) == 0) || tune_str("ROWS_FWD")) return false;
  const Ptr& below = d0.hd.arena >= 0 ? d0.hd : d0.h;               // what the upper layer reads: h after the fused dropout, or h
  if (d0.impl != 1 || d1.impl != 1 || d0.H != d1.H || d0.B != d1.B || d0.T != d1.T || d0.gxdt != d1.gxdt || d0.hdt != DT_BF16 || d1.hdt != DT_BF16) return false;
  if (d0.xfeat != 32 || d1.xfeat != d1.H || d1.xin.arena != below.arena || d1.xin.off != below.off) return false;
  if (d0.t1 != 0 || d1.t1 != 0 || d0.t0 != 0 || d1.t0 != 0) return false;
  const bool g16 = d0.gxdt == DT_BF16;
  switch (d0.H) {
    case 256: return g16 ? launch_pair2<256, 3, true>(d0, d1, ab, st) : launch_pair2<256, 3, false>(d0, d1, ab, st);
    case 384: return g16 ? launch_pair2<384, 3, true>(d0, d1, ab, st) : launch_pair2<384, 3, false>(d0, d1, ab, st);
    case 512: return g16 ? launch_pair2<512, 2, true>(d0, d1, ab, st) : launch_pair2<512, 2, false>(d0, d1, ab, st);
    default: return false;
  }
}

void launch_lstm_rows(const LstmRec& d, const ArenaBases& ab, hipStream_t st, bool fwd) {
  // forward, H = 384 (FullSubNet's sub-band model), ms per training step.  Row-major packed weights: 48 rows x 8 waves 110.8, 48 x 4 111.9,
  // 80 rows x 4 waves 102.7.  Fragment-major weights: 88.6 / 88.7 / 89.9 - the geometry stopped mattering (HBM-bound); 48 x 8 is launched,
  // SEFD_ROWS_FWD=54 / 34 select the others.
  static const int fv = tune_str("ROWS_FWD") ? atoi(tune_str("ROWS_FWD")) : 0;
  if (fwd && d.H == 384 && fv == 54 && d.xfeat != 384) { if (d.gxdt == DT_BF16) launch_r2<384, 5, true, 4>(d, ab, st, true); else launch_r2<384, 5, false, 4>(d, ab, st, true); return; }
  switch (d.H) {
    case 256: launch_r<256, 3>(d, ab, st, fwd); break;
    case 384: launch_r<384, 3>(d, ab, st, fwd); break;
    default: launch_r<512, 2>(d, ab, st, fwd); break;      // 32 rows: dgates_t of 48 rows x 2048 columns does not fit 160 KB
  }
}

}  // namespace sefd
