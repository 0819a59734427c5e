// GEMM ladder: the wide-tile conv GEMMs of the DCCRN step on REAL planner descriptors (tools/probes/gemm_desc.bin, written on
// the CPU by tools/probes/dump_gemm_desc.py) and on plain row-major GEMMs of the same shapes, old kernel (cgemm256.hip) against the phase-staggered
// kernel (cgemm8p.hip) and its ablation arms, one process, interleaved rounds, random bf16 operands in [-1, 1).
//   build:  tools/probes/build_ladder.sh        run (MI355X):  tools/probes/gemm_ladder [rounds]
// Correctness: for every descriptor the two kernels must leave the SAME workspace (64-bit position-mixed checksum over the whole arena, both runs
// started from the same generated contents) - cgemm256 is the parity-green kernel of rounds 3-5.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "sefd_desc.h"
#include "dev_common.h"

namespace sefd {
extern int g_cgemm8p_var, g_cgemm256_dbg;
bool launch_cgemm8p(const RunGemm& d, const ArenaBases& ab, hipStream_t st);      // tools/probes/ladder/cgemm8p.hip
bool launch_cgemm128(const RunGemm& d, const ArenaBases& ab, hipStream_t st);     // tools/probes/ladder/cgemm128.hip
}
using namespace sefd;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t h32(uint64_t i, uint32_t seed) {
  uint32_t x = (uint32_t)i * 0x9E3779B1u ^ (uint32_t)(i >> 32) * 0x85EBCA77u ^ seed;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
// every 16-bit word of the buffer = a bf16 value uniform in [-1, 1)
__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float f = (float)(h32(i, seed) >> 8) * (2.f / 16777216.f) - 1.f;
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    p[i] = (uint16_t)(u >> 16);
  }
}
__global__ void fill_f32(float* p, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = (float)(h32(i, seed) >> 8) * (2.f / 16777216.f) - 1.f;
}
__global__ void checksum(const uint64_t* p, size_t n, unsigned long long* out) {
  unsigned long long s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long v = p[i];
    v ^= (unsigned long long)i * 0x9E3779B97F4A7C15ull;
    v *= 0xD6E8FEB86659FD93ull; v ^= v >> 32;
    s += v;
  }
  atomicAdd(out, s);
}

struct Rec { int phase, index; Op op; std::string name; };

// plain[n][ldw] <- K-tile-major tiled[(k >> 5) * Npad + n][32] (the layout kRunWTile32 names): the 128-row kernel reads the plain copy
__global__ void untile_w(const uint16_t* tiled, uint16_t* plain, int Npad, int ldw) {
  const size_t n = (size_t)Npad * ldw;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int nn = (int)(i / ldw), k = (int)(i % ldw);
    plain[i] = tiled[((size_t)(k >> 5) * Npad + nn) * 32 + (k & 31)];
  }
}

static ArenaBases g_ab;
static int64_t g_bytes[A_COUNT];
static unsigned long long* g_sum;

static void refill() {
  hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, (uint16_t*)g_ab.p[A_WS], (size_t)g_bytes[A_WS] / 2, 12345u);
}
static unsigned long long ws_sum() {
  CK(hipMemset(g_sum, 0, 8));
  hipLaunchKernelGGL(checksum, dim3(4096), dim3(256), 0, 0, (const uint64_t*)g_ab.p[A_WS], (size_t)g_bytes[A_WS] / 8, g_sum);
  unsigned long long h;
  CK(hipMemcpy(&h, g_sum, 8, hipMemcpyDeviceToHost));
  return h;
}

static int64_t g_spare;          // byte offset of the spare region behind the workspace proper (plain weight copies)
// arm: -1 = the old kernel (cgemm256; for widths that are an odd multiple of 128: the 128 x 128 kernel of rungemm.hip on a plain copy of the weights);
// >= 0 = cgemm8p with that VAR (odd multiples of 128: cgemm128, arm 0 only)
static void launch(const RunGemm& g, int arm) {
  const bool w128 = g.Npad % 256 != 0;
  if (arm < 0) {
    if (w128) {
      RunGemm o = g;
      o.flags &= ~kRunWTile32;
      o.w = Ptr{A_WS, 0, g_spare};
      static int64_t last = -1;                              // (refill() regenerates the same weights: one plain copy per weight buffer)
      if (last != g.w.off) {
        hipLaunchKernelGGL(untile_w, dim3(1024), dim3(256), 0, 0, (const uint16_t*)rp(g_ab, g.w), (uint16_t*)(g_ab.p[A_WS] + g_spare), g.Npad, g.ldw);
        last = g.w.off;
      }
      launch_rungemm(o, g_ab, 0);
    } else { g_cgemm256_dbg = arm == -1 ? 0 : -arm; if (!launch_cgemm256(g, g_ab, 0)) { fprintf(stderr, "cgemm256 refused\n"); exit(1); } }
  } else if (w128) {
    RunGemm n = g;
    n.flags |= kRunWTile32;                                  // the weight bytes are random: read as K-tile major here, the old arm reads their un-tiled copy
    if (!launch_cgemm128(n, g_ab, 0)) { fprintf(stderr, "cgemm128 refused\n"); exit(1); }
  }
  else { g_cgemm8p_var = arm; if (!launch_cgemm8p(g, g_ab, 0)) { fprintf(stderr, "cgemm8p refused\n"); exit(1); } }
}

static RunGemm plain(int M, int N, int K, int lda, int64_t offA, int64_t offW, int64_t offY) {
  RunGemm g;
  memset(&g, 0, sizeof(g));
  g.x[0] = Ptr{A_WS, 0, offA}; g.x[1] = Ptr{A_NONE, 0, 0};
  g.xdt = DT_BF16; g.ydt = DT_BF16;
  g.fstride[0] = lda; g.rowlen[0] = (int32_t)std::min<int64_t>((int64_t)(M - 1) * lda + K, 0x7fffffff); g.Tin[0] = 1; g.Tin[1] = 1;
  g.M = M; g.Tout = 1; g.Fo = M;
  g.nseg = 1; g.seg[0] = Seg{0, 0, 0, K, 0};
  g.w = Ptr{A_WS, 0, offW}; g.ldw = K; g.N = N; g.Npad = N;
  g.bias = Ptr{A_NONE, 0, 0};
  g.y = Ptr{A_WS, 0, offY}; g.y_fstride = N; g.y_tstride = M * N; g.y_bstride = (int64_t)M * N; g.y_off = 0;
  g.stats = Ptr{A_NONE, 0, 0};
  g.flags = kRunAligned | kRunYAligned | kRunWTile32;
  g.zero = Ptr{A_CONST, 0, 0};
  fastdiv_make((uint32_t)M, &g.div_tf_m, &g.div_tf_s);
  fastdiv_make((uint32_t)M, &g.div_fo_m, &g.div_fo_s);
  g.y2 = Ptr{A_NONE, 0, 0};
  g.bnb_y = g.bnb_mi = g.bnb_gamma = g.bnb_beta = g.bnb_slope = Ptr{A_NONE, 0, 0};
  return g;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 3;
  const char* path = argc > 2 ? argv[2] : "tools/probes/gemm_desc.bin";
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); return 1; }
  int64_t hdr[9];
  if (fread(hdr, 8, 9, f) != 9 || hdr[0] != 0x53454644 || hdr[1] != (int64_t)sizeof(Op)) { fprintf(stderr, "bad header (op size %lld vs %zu)\n", (long long)hdr[1], sizeof(Op)); return 1; }
  for (int a = 0; a < A_COUNT; ++a) g_bytes[a] = hdr[2 + a];
  std::vector<Rec> recs((size_t)hdr[8]);
  for (auto& r : recs) {
    if (fread(&r.phase, 4, 1, f) != 1 || fread(&r.index, 4, 1, f) != 1 || fread(&r.op, sizeof(Op), 1, f) != 1) { fprintf(stderr, "short file\n"); return 1; }
    char nm[96];
    snprintf(nm, sizeof nm, "p%d op%-3d tag%d%s", r.phase, r.index, r.op.tag, (r.op.g.flags & kRunBnBwd) ? " BNB" : "");
    r.name = nm;
  }
  fclose(f);
  g_bytes[A_WS] = std::max<int64_t>(g_bytes[A_WS], (int64_t)7 << 30);
  g_spare = (g_bytes[A_WS] + 4095) / 4096 * 4096;
  for (int a = 0; a < A_COUNT; ++a) {
    const size_t extra = a == A_WS ? (size_t)(g_spare - g_bytes[a]) + (64u << 20) : 4096;
    CK(hipMalloc((void**)&g_ab.p[a], (size_t)g_bytes[a] + extra));
    CK(hipMemset(g_ab.p[a], 0, (size_t)g_bytes[a] + extra));
  }
  g_ab.status = nullptr; g_ab.dstatus = nullptr;
  CK(hipMalloc((void**)&g_sum, 8));
  hipLaunchKernelGGL(fill_f32, dim3(1024), dim3(256), 0, 0, (float*)g_ab.p[A_PARAM], (size_t)g_bytes[A_PARAM] / 4, 777u);
  hipLaunchKernelGGL(fill_f32, dim3(64), dim3(256), 0, 0, (float*)g_ab.p[A_STATE], (size_t)g_bytes[A_STATE] / 4, 778u);
  CK(hipDeviceSynchronize());

  struct Case { std::string name; RunGemm g; bool plain; };
  std::vector<Case> cases;
  // plain GEMMs: A [M][lda] at 0, W (K-tile major) at 5 GB, Y at 6 GB of the workspace
  const int64_t oW = (int64_t)5 << 30, oY = (int64_t)6 << 30;
  cases.push_back({"plain 4096^3", plain(4096, 4096, 4096, 4096, 0, oW, oY), true});
  cases.push_back({"plain 8192^3", plain(8192, 8192, 8192, 8192, 0, oW, oY), true});
  cases.push_back({"plain M123904 N256 K3072", plain(123904, 256, 3072, 3072, 0, oW, oY), true});
  cases.push_back({"plain M123904 N256 K3072 lda320 (rows overlap: conv-like L2 reuse)", plain(123904, 256, 3072, 320, 0, oW, oY), true});
  cases.push_back({"plain M247296 N256 K1280", plain(247296, 256, 1280, 1280, 0, oW, oY), true});
  cases.push_back({"plain M247296 N256 K1280 lda128", plain(247296, 256, 1280, 128, 0, oW, oY), true});
  for (auto& r : recs) cases.push_back({r.name, r.op.g, false});

  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("# rounds %d; us per launch (median over rounds of 3-launch averages), TFLOP/s = 2 M N K / t\n", rounds);
  int bad = 0;
  for (auto& c : cases) {
    const RunGemm& g = c.g;
    const bool bnb = (g.flags & kRunBnBwd) != 0;
    int K = 0;
    for (int s = 0; s < g.nseg; ++s) K += g.seg[s].len;
    const double flops = 2.0 * g.M * g.N * K;
    // ---- parity: same workspace after the old and the new kernel
    refill(); launch(g, -1); const unsigned long long s_old = ws_sum();
    refill(); launch(g, 0); const unsigned long long s_new = ws_sum();
    refill(); const unsigned long long s_none = ws_sum();
    const bool ok = s_old == s_new && s_old != s_none;
    if (!ok) ++bad;
    std::vector<int> arms = {-1, 0};
    if (g.Npad % 256 == 0) { arms.push_back(-128); arms.push_back(-256); }      // old kernel with its DMA issue at 2 / 4 wave-dependent positions
    if (g.Npad % 256 != 0) { /* cgemm128 has one arm */ }
    else if (!bnb) { arms.push_back(1); arms.push_back(2); arms.push_back(3); arms.push_back(8); arms.push_back(16); arms.push_back(32); arms.push_back(64); if (c.plain) arms.push_back(4); }
    std::vector<std::vector<float>> t(arms.size());
    for (int r = 0; r < rounds; ++r)
      for (size_t a = 0; a < arms.size(); ++a) {
        launch(g, arms[a]);
        CK(hipEventRecord(e0));
        for (int k = 0; k < 3; ++k) launch(g, arms[a]);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        t[a].push_back(ms / 3.f * 1e3f);
      }
    printf("%-64s M%7d N%5d K%5d  parity %s |", c.name.c_str(), g.M, g.N, K, ok ? "same" : "DIFFERENT");
    for (size_t a = 0; a < arms.size(); ++a) {
      std::sort(t[a].begin(), t[a].end());
      const float us = t[a][t[a].size() / 2];
      char lab[16];
      if (arms[a] < -1) snprintf(lab, sizeof lab, "old/%d", -arms[a]); else if (arms[a] < 0) snprintf(lab, sizeof lab, "old"); else if (g.Npad % 256 != 0) snprintf(lab, sizeof lab, "c128"); else snprintf(lab, sizeof lab, "8p/%d", arms[a]);
      printf(" %s %7.1fus %6.0fTF |", lab, us, flops / (us * 1e-6) / 1e12);
    }
    printf("\n");
    fflush(stdout);
  }
  CK(hipDeviceSynchronize());
  printf("# parity failures: %d\n", bad);
  return bad ? 2 : 0;
}
