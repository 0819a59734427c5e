// Device-side helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "sefd_desc.h"

namespace sefd {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

struct bf16_t { uint16_t v; };          // storage-only 16-bit type (sizeof == 2) used as the template tag of bf16 paths

// status / dstatus: the status word of the PLAN that issues the launch (api.hip; 0 = fine) as a host-mapped word (the host polls it without
// a copy) and as a device word (the guarded Adam reads it - a million lanes loading a host-mapped word over PCIe cost 1.7 ms per step).  A
// kernel that had to give up (the cluster LSTM's bounded hand-over waits) sets BOTH (report through set_status).
struct ArenaBases { char* p[A_COUNT]; int* status; int* dstatus; };
__device__ __forceinline__ void set_status(int* host_word, int* dev_word) {
  if (host_word) __hip_atomic_store(host_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (dev_word) __hip_atomic_store(dev_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__host__ __device__ __forceinline__ char* rp(const ArenaBases& ab, const Ptr& q) { return ab.p[q.arena] + q.off; }

// fp32 -> bf16, round-to-nearest-even, on the gfx950 conversion unit: ONE v_cvt_pk_bf16_f32 packs two values (the software
// sequence is 6 VALU ops per value and dominated the epilogues / the LSTM step loops).  No HIP builtin exists: inline asm.
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ uint16_t f2bf(float f) { return (uint16_t)pack_bf16x2(f, 0.f); }

// Epilogue transpose of the 32x32 MFMA accumulator layout.  A lane holds ONE column (lane & 31) and, per group q, the 4 consecutive rows
// 8q + 4 (lane >> 5) + 0..3: stored as they sit, a tile leaves as 2-byte pieces (round 4, SEFD_CG256_DBG=8: the staged epilogue of the wide
// kernel with 128 ds_write_b16 per lane was 28-34 % of its run time).  A 4 x 4 transpose inside each quad of lanes on PACKED bf16 pairs - two
// DPP row exchanges (lane ^ 1 with a byte permute, lane ^ 2 with selects), 11 VALU operations per 4 values - leaves lane (lane & 3) = p with
// row 8q + 4 (lane >> 5) + p and the 4 consecutive columns (lane & 28) .. + 3: one 8-byte store.  Must run with all 64 lanes active.
struct QuadT {
  uint32_t sel;      // v_perm selector of the lane ^ 1 exchange: even lanes take the low halves (self, partner), odd lanes the high halves (partner, self)
  bool hi2;          // lane & 2
  __device__ __forceinline__ explicit QuadT(int lane) : sel((lane & 1) ? 0x03020706u : 0x05040100u), hi2((lane & 2) != 0) {}
  __device__ __forceinline__ uint2 pack(uint32_t p01, uint32_t p23) const {                 // p01 = bf16 (row 0 | row 1 << 16), p23 = (row 2 | row 3 << 16) of this lane's column
    const uint32_t n01 = (uint32_t)__builtin_amdgcn_mov_dpp((int)p01, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
    const uint32_t n23 = (uint32_t)__builtin_amdgcn_mov_dpp((int)p23, 0xB1, 0xF, 0xF, true);
    const uint32_t q0 = __builtin_amdgcn_perm(n01, p01, sel);      // even lane: row 0 of columns (c, c + 1); odd lane: row 1 of columns (c - 1, c)
    const uint32_t q1 = __builtin_amdgcn_perm(n23, p23, sel);      // even lane: row 2; odd lane: row 3
    const uint32_t send = hi2 ? q0 : q1, keep = hi2 ? q1 : q0;
    const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xF, 0xF, true); // quad_perm [2, 3, 0, 1]
    return make_uint2(hi2 ? recv : keep, hi2 ? keep : recv);
  }
  __device__ __forceinline__ uint2 pack(float v0, float v1, float v2, float v3) const { return pack(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)); }
};
// 16-byte epilogue stores (round 5): after QuadT::pack lane l holds 4 consecutive columns of one row and lane l ^ 4 the NEXT 4 columns of the same
// row.  Stored as they are (8 bytes per lane) a tile leaves at the store-ISSUE rate of ~7 B/clk per CU (MI355X_MICROARCH.md, "attention epilogue
// store tail": 16 x dwordx2 per lane) - measured on the N = 128 layers the epilogue was HALF the launch (profiles/r05_tuning_notes.md section 7).
// OctW pairs the lanes: of the four row groups q the lane with bit 2 clear keeps q = 0, 2 and takes its partner's pieces of those rows, the lane with
// bit 2 set keeps q = 1, 3: two 16-byte stores per lane instead of four 8-byte ones.  xor4: DPP row shifts by 4 inside the 16-lane row (VALU only).
__device__ __forceinline__ uint32_t xor4(uint32_t x, bool hi4) {
  const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xF, 0xF, true);   // row_shl:4  lane i <- lane i + 4
  const uint32_t dn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);   // row_shr:4  lane i <- lane i - 4
  return hi4 ? dn : up;
}
struct OctW {
  bool hi4;          // lane & 4
  __device__ __forceinline__ explicit OctW(int lane) : hi4((lane & 4) != 0) {}
  // pk[q]: this lane's 8-byte piece of row group q.  Returns the two 16-byte chunks this lane stores: rows q = 0, 2 (bit 2 clear) or q = 1, 3 (set),
  // columns (lane & 24) .. + 7 of the 32-column group.
  __device__ __forceinline__ void widen(const uint2 (&pk)[4], uint4 (&out)[2]) const {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint2 keep = hi4 ? pk[2 * h + 1] : pk[2 * h], send = hi4 ? pk[2 * h] : pk[2 * h + 1];
      const uint2 recv = make_uint2(xor4(send.x, hi4), xor4(send.y, hi4));
      out[h] = hi4 ? make_uint4(recv.x, recv.y, keep.x, keep.y) : make_uint4(keep.x, keep.y, recv.x, recv.y);
    }
  }
};
__device__ __forceinline__ float bf2f(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

template <int DT> struct Elem;
__device__ __forceinline__ float ld_elem(const char* base, int dt, int64_t i) {
  return dt == DT_BF16 ? bf2f(reinterpret_cast<const uint16_t*>(base)[i]) : reinterpret_cast<const float*>(base)[i];
}
__device__ __forceinline__ void st_elem(char* base, int dt, int64_t i, float v) {
  if (dt == DT_BF16) reinterpret_cast<uint16_t*>(base)[i] = f2bf(v);
  else reinterpret_cast<float*>(base)[i] = v;
}

// 4 consecutive elements (i multiple of 4)
__device__ __forceinline__ float4 ld4(const char* base, int dt, int64_t i) {
  if (dt == DT_BF16) {
    const uint2 r = *reinterpret_cast<const uint2*>(base + i * 2);
    return make_float4(bf2f(r.x & 0xffff), bf2f(r.x >> 16), bf2f(r.y & 0xffff), bf2f(r.y >> 16));
  }
  return *reinterpret_cast<const float4*>(base + i * 4);
}
__device__ __forceinline__ void st4(char* base, int dt, int64_t i, float4 v) {
  if (dt == DT_BF16) {
    uint2 r;
    r.x = pack_bf16x2(v.x, v.y);
    r.y = pack_bf16x2(v.z, v.w);
    *reinterpret_cast<uint2*>(base + i * 2) = r;
  } else {
    *reinterpret_cast<float4*>(base + i * 4) = v;
  }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global store
// (s_waitcnt vmcnt(0)); in the per-time-step loops of the LSTM kernels that would expose the full HBM write latency
// of the saved activations on every one of the 483 steps.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// LDS-DMA issued as inline assembly.  With the builtin the compiler knows the instruction writes LDS and, unable to tell
// the ring-buffer stages apart, puts s_waitcnt vmcnt(0) in front of every later ds_read - which serialises "issue the
// next tile" with "read this tile" and removes all overlap.  As opaque asm the DMA is ordered by explicit counters only:
// each thread issues a fixed number of DMAs per stage and waits with wait_vm<N>() for all but the newest N.
// `lds_wave_base` (wave-uniform byte address) + lane * 16 is the destination.
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_wave_base) {
  // (s_nop 0: the one wait state the ISA asks for between an SALU write of M0 and the LDS-DMA that reads it - nothing pads inline asm,
  // MI355X_MICROARCH / cdna_hip_programming.md section 5.7; five rounds of bit-reproducibility tests never caught the hazard, the slot is free)
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(lds_wave_base)), "v"(gsrc)
               : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// Wait until the oldest in-flight stage has landed: `rem` newer stages (at most S-2) of NL DMAs each may stay in flight.
template <int NL, int S>
__device__ __forceinline__ void wait_stage(int rem) {
  if constexpr (S >= 4) { if (rem >= 2) { wait_vm<2 * NL>(); return; } }
  if constexpr (S >= 3) { if (rem >= 1) { wait_vm<NL>(); return; } }
  wait_vm<0>();
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Gate non-linearities on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1 ulp each): the recurrence is
// latency-bound per time step, libm-accurate expf/tanhf would triple the step time for ~1e-7 of accuracy.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }

// Inverted-dropout scale of element i (counter hash on (seed, layer, i); nn.LSTM(dropout=0.8) between FullSubNet's two layers,
// tools_for_model.py:746): 1/keep with probability keep, else 0.  Shared by dropout_kernel (fsn.hip) and the LSTM kernels that fuse it.
__device__ __forceinline__ uint32_t mix32(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x632BE5ABu) * 0xC2B2AE3Du;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float drop_scale(uint32_t seed0, uint32_t seed1, int layer, float keep, int64_t i) {
  if (keep >= 1.f) return 1.f;
  const uint32_t r = mix32(seed0 + (uint32_t)layer * 0x51ED27u, seed1 ^ (uint32_t)(i >> 32), (uint32_t)i);
  return ((r >> 8) * (1.f / 16777216.f)) < keep ? 1.f / keep : 0.f;
}

// kRunBnBwd epilogue (sefd_desc.h RunGemm::bnb_*): one stored gradient element dz against the layer's forward output yf
struct BnbCol { float mean, invstd, gamma, beta; };
__device__ __forceinline__ BnbCol bnb_col(const RunGemm& d, const ArenaBases& ab, int n) {
  BnbCol c{0.f, 0.f, 0.f, 0.f};
  if (n < d.N) {
    const float* mi = reinterpret_cast<const float*>(rp(ab, d.bnb_mi));
    c.mean = mi[n]; c.invstd = mi[d.N + n];
    c.gamma = reinterpret_cast<const float*>(rp(ab, d.bnb_gamma))[n];
    c.beta = reinterpret_cast<const float*>(rp(ab, d.bnb_beta))[n];
  }
  return c;
}
__device__ __forceinline__ void bnb_accum(const BnbCol& c, float slope, float dz, float yf, float& s0, float& s1, float& s2) {
  const float xh = (yf - c.mean) * c.invstd;
  const float bn = c.gamma * xh + c.beta;
  const float dbn = bn > 0.f ? dz : slope * dz;
  s0 += dbn;
  s1 += dbn * xh;
  s2 += bn > 0.f ? 0.f : bn * dz;
}

// Key of the per-stream scratch caches of the launchers (ticket words, carry buffers, finalize scratch): (device, stream) - the null / default stream
// handle is the same value on every device, so a process that runs plans on two GPUs through it must not get the first device's buffers on the second.
struct StreamKey {
  int dev; hipStream_t st;
  bool operator==(const StreamKey& o) const { return dev == o.dev && st == o.st; }
};
struct StreamKeyHash { size_t operator()(const StreamKey& k) const { return (size_t)(uintptr_t)k.st * 31u + (size_t)k.dev; } };
inline StreamKey stream_key(hipStream_t st) { int dev = 0; (void)hipGetDevice(&dev); return StreamKey{dev, st}; }

void launch_rungemm(const RunGemm& d, const ArenaBases& ab, hipStream_t st);
void launch_wgrad(const RunGemm& d, const ArenaBases& ab, hipStream_t st);
bool launch_cgemm256(const RunGemm& d, const ArenaBases& ab, hipStream_t st);
bool launch_enc0_fwd(const RunGemm& d, const ArenaBases& ab, hipStream_t st);        // first encoder layer on the fp32 spectrum (enc0.hip)
bool launch_enc0_wgrad(const RunGemm& d, const ArenaBases& ab, hipStream_t st);
bool launch_rundirect(const RunGemm& d, const ArenaBases& ab, hipStream_t st);       // thin layers, N <= 64 (thin.hip)
void launch_misc(const Op& op, const ArenaBases& ab, hipStream_t st);
void launch_stft_fft(const StftFft& d, const ArenaBases& ab, hipStream_t st);
void launch_istft_fft(const IstftFft& d, const ArenaBases& ab, hipStream_t st);
void launch_fsn(const Op& op, const ArenaBases& ab, hipStream_t st);
void launch_bn(const Op& op, const ArenaBases& ab, hipStream_t st);
void launch_cbn(const Op& op, const ArenaBases& ab, hipStream_t st);       // ComplexBatchNorm (cbn.hip)
void launch_lstm_bf16(const LstmRec& d, const ArenaBases& ab, hipStream_t st, bool fwd);
void launch_lstm_cluster(const LstmRec& d, const ArenaBases& ab, hipStream_t st, bool fwd);   // H > 128 (lstm_cluster.hip)
void launch_lstm_rows(const LstmRec& d, const ArenaBases& ab, hipStream_t st, bool fwd);      // impl == 1 (lstm_rows.hip)
bool launch_lstm_rows_pair(const LstmRec& d0, const LstmRec& d1, const ArenaBases& ab, hipStream_t st);   // two stacked forward layers, one launch

}  // namespace sefd
