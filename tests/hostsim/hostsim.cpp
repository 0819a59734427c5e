// TEST INFRASTRUCTURE ONLY - never linked into the product library.
// CPU interpreter of the launch descriptors (csrc/sefd_desc.h).  It executes the very op list the HIP executor
// launches, on host copies of the arenas, with straightforward loops and double accumulation.  Two uses:
//   * CPU tests (-m "not gpu"): planner index arithmetic (descriptors + pack/unpack tables) checked against the oracle
//     without a GPU;
//   * GPU tests: per-op expected buffers, so that a wrong kernel is localised to the op that first deviates.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../../dnn-based-speech-enhancement-in-the-frequency-domain_amd/csrc/sefd_desc.h"

using namespace sefd;

namespace {
struct AB { char* p[A_COUNT]; };
inline char* rp(const AB& ab, const Ptr& q) { return ab.p[q.arena] + q.off; }
inline float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }
inline uint16_t f2bf(float f) {
  uint32_t u; std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline float ld(const char* b, int dt, int64_t i) { return dt == DT_BF16 ? bf2f(((const uint16_t*)b)[i]) : ((const float*)b)[i]; }
inline void st(char* b, int dt, int64_t i, float v) { if (dt == DT_BF16) ((uint16_t*)b)[i] = f2bf(v); else ((float*)b)[i] = v; }

// value of run element (seg, j) for row (b,u,fo); `ok` false -> 0
inline double a_elem(const RunGemm& d, const AB& ab, const Seg& sg, int b, int u, int fo, int j) {
  if (sg.src < 0) return j == 0 ? 1.0 : 0.0;
  const int s = sg.src;
  const int tt = u + sg.dt;
  if (tt < 0 || tt >= d.Tin[s]) return 0.0;
  const int r = sg.off + fo * d.fstride[s] + j;
  if (r < 0 || r >= d.rowlen[s] || j >= sg.len) return 0.0;
  return ld(rp(ab, d.x[s]), d.xdt, (int64_t)b * d.bstride[s] + (int64_t)tt * d.tstride[s] + d.base[s] + r);
}

void rungemm(const RunGemm& d, const AB& ab) {
  const char* w = rp(ab, d.w);
  const float* bias = d.bias.arena >= 0 ? (const float*)rp(ab, d.bias) : nullptr;
  char* y1 = rp(ab, d.y);
  const int TF = d.Tout * d.Fo;
  const int nblk = (d.M + kBM - 1) / kBM;
  std::vector<double> s1, s2, s3;
  const bool bnb = (d.flags & kRunBnBwd) != 0;
  if (d.stats.arena >= 0) { s1.assign((size_t)nblk * d.Npad, 0.0); s2.assign((size_t)nblk * d.Npad, 0.0); s3.assign((size_t)nblk * d.Npad, 0.0); }
  const float* bmi = bnb ? (const float*)rp(ab, d.bnb_mi) : nullptr;
  const float* bgamma = bnb ? (const float*)rp(ab, d.bnb_gamma) : nullptr;
  const float* bbeta = bnb ? (const float*)rp(ab, d.bnb_beta) : nullptr;
  const float bslope = bnb ? *(const float*)rp(ab, d.bnb_slope) : 0.f;
  // dense copy of the weights (w_index_g is an index FUNCTION: one call per MAC was most of this loop)
  std::vector<double> wd((size_t)d.N * d.ldw);
  for (int nn = 0; nn < d.N; ++nn)
    for (int k = 0; k < d.ldw; ++k) wd[(size_t)nn * d.ldw + k] = ld(w, d.xdt, w_index_g(d, nn, k));
  // 128-row blocks are independent (the statistics are per block, summed in row order inside a block): one block per thread,
  // same summation order as the serial loop.  (kRunAccum reads y at the element it writes: rows never alias.)
#pragma omp parallel for schedule(dynamic, 1)
  for (int blk = 0; blk < nblk; ++blk) {
  std::vector<double> arow(d.ldw);
  const int mend = std::min(d.M, (blk + 1) * kBM);
  for (int m = blk * kBM; m < mend; ++m) {
    const int b = m / TF, rem = m % TF, u = rem / d.Fo, fo = rem % d.Fo;
    std::fill(arow.begin(), arow.end(), 0.0);
    for (int s = 0; s < d.nseg; ++s)
      for (int j = 0; j < d.seg[s].len; ++j) arow[d.seg[s].koff + j] = a_elem(d, ab, d.seg[s], b, u, fo, j);
    const int64_t o = (int64_t)b * d.y_bstride + (int64_t)u * d.y_tstride + (int64_t)fo * d.y_fstride + d.y_off;
    for (int nn = 0; nn < d.N; ++nn) {
      double acc = 0.0;
      const double* wr = &wd[(size_t)nn * d.ldw];
      for (int k = 0; k < d.ldw; ++k) acc += arow[k] * wr[k];
      float v = (float)acc + (bias ? bias[nn] : 0.f);
      const bool second = d.n2 > 0 && nn >= d.n2;             // two destinations: columns >= n2 go to y2 at column n - n2
      char* y = second ? rp(ab, d.y2) : y1;
      const int n = second ? nn - d.n2 : nn;
      if (d.flags & kRunAccum) v += ((const float*)y)[o + n];
      if (d.flags & kRunRelu) v = v > 0.f ? v : 0.f;
      st(y, d.ydt, o + n, v);
      if (d.stats.arena >= 0 && bnb) {             // BatchNorm-backward sums of the STORED gradient against the layer's forward output
        const float dz = ld(y, d.ydt, o + n);
        const int64_t ob = (int64_t)b * d.bnb_bstride + (int64_t)u * d.bnb_tstride + (int64_t)fo * d.bnb_fstride + d.bnb_off;
        const float xh = (ld(rp(ab, d.bnb_y), d.ydt, ob + n) - bmi[n]) * bmi[d.N + n];
        const float bn = bgamma[n] * xh + bbeta[n];
        const double dbn = bn > 0.f ? dz : bslope * dz;
        const size_t q = (size_t)(m / kBM) * d.Npad + n;
        s1[q] += dbn; s2[q] += dbn * xh; if (!(bn > 0.f)) s3[q] += (double)bn * dz;
      } else
      if (d.stats.arena >= 0) { s1[(size_t)(m / kBM) * d.Npad + n] += v; s2[(size_t)(m / kBM) * d.Npad + n] += (double)v * v; }
    }
  }
  }
  if (d.stats.arena >= 0) {
    float* part = (float*)rp(ab, d.stats);
    const int nst = bnb ? 3 : 2;
    for (int blk = 0; blk < nblk; ++blk)
      for (int n = 0; n < d.Npad; ++n) {
        part[((int64_t)blk * nst + 0) * d.Npad + n] = (float)s1[(size_t)blk * d.Npad + n];
        part[((int64_t)blk * nst + 1) * d.Npad + n] = (float)s2[(size_t)blk * d.Npad + n];
        if (bnb) part[((int64_t)blk * nst + 2) * d.Npad + n] = (float)s3[(size_t)blk * d.Npad + n];
      }
  }
}

inline void cell_off(const LstmCell& d, int64_t r, int64_t* o) {
  if (d.G > 0) {
    const int g = (int)(r / d.Bg);
    const int64_t b = r - (int64_t)g * d.Bg;
    for (int k = 0; k < 5; ++k) o[k] = d.go[k][g] + b * d.rs[k];
  } else {
    o[0] = r * 4 * d.H; o[1] = r * d.H; o[2] = r * d.H; o[3] = r * d.H; o[4] = r * 4 * d.H;
  }
}
inline int cell_col(const LstmCell& d, int q, int j) { return d.unit_major ? gate_col(q, j) : q * d.H + j; }

void pack(const Pack& d, const AB& ab) {
  const int32_t* tab = (const int32_t*)rp(ab, d.tab);
  const float* src = (const float*)rp(ab, d.src);
  for (int64_t i = 0; i < d.n; ++i) {
    float v = 0.f;
    for (int e = 0; e < d.width; ++e) {
      const int32_t t = tab[i * d.width + e];
      if (t > 0) v += src[t - 1]; else if (t < 0) v -= src[-t - 1];
    }
    st(rp(ab, d.dst), d.ddt, i, v);
  }
}

void wgrad(const RunGemm& d, const AB& ab) {
  const char* dy = rp(ab, d.y);
  float* part = (float*)rp(ab, d.w);
  const int TF = d.Tout * d.Fo;
  const int64_t sz = (int64_t)d.Npad * d.ldw;
  // The LDS-DMA bf16 kernel walks the rows per batch item (steps of kWgRows rows, the last step of an item padded); the
  // other kernels walk the flat row index.  The per-split partials are compared, so the partition is mirrored here.
  const bool per_item = d.xdt == DT_BF16 && (d.flags & kRunAligned);
  const int nb = d.M / TF;
  const int spb = (TF + kWgRows - 1) / kWgRows;
  const int nsteps = per_item ? nb * spb : (d.M + kWgRows - 1) / kWgRows;
  const int per = (nsteps + d.nsplit - 1) / d.nsplit;
  std::vector<double> acc(sz);
  for (int sp = 0; sp < d.nsplit; ++sp) {
    std::fill(acc.begin(), acc.end(), 0.0);
    const int st0 = sp * per, st1 = std::min(nsteps, (sp + 1) * per);
    // every thread owns a range of output columns n and walks ALL rows of the split in order (the operand row is rebuilt per
    // thread): per accumulator element the same sequence of additions as the serial loop
#pragma omp parallel
    {
    int nth = 1, tid = 0;
#ifdef _OPENMP
    nth = omp_get_num_threads(); tid = omp_get_thread_num();
#endif
    const int n0 = (int)((int64_t)d.N * tid / nth), n1 = (int)((int64_t)d.N * (tid + 1) / nth);
    std::vector<double> arow(d.ldw);
    if (n1 > n0)
    for (int stp = st0; stp < st1; ++stp)
    for (int r = 0; r < kWgRows; ++r) {
      int m;
      if (per_item) {
        const int b = stp / spb, q = (stp - b * spb) * kWgRows + r;
        if (q >= TF) continue;
        m = b * TF + q;
      } else {
        m = stp * kWgRows + r;
        if (m >= d.M) continue;
      }
      const int b = m / TF, rem = m % TF, u = rem / d.Fo, fo = rem % d.Fo;
      std::fill(arow.begin(), arow.end(), 0.0);
      for (int s = 0; s < d.nseg; ++s)
        for (int j = 0; j < d.seg[s].len; ++j) arow[d.seg[s].koff + j] = a_elem(d, ab, d.seg[s], b, u, fo, j);
      const int64_t o = (int64_t)b * d.y_bstride + (int64_t)u * d.y_tstride + (int64_t)fo * d.y_fstride + d.y_off;
      for (int n = n0; n < n1; ++n) {
        double g = ld(dy, d.ydt, o + n);
        if (d.flags & kRunDyFromBn) {                // `y` is dz0: through the layer's BatchNorm + PReLU backward on the way in (sefd_desc.h)
          float dz = (float)g;
          if (d.bnb_dz1.arena >= 0) dz += ld(rp(ab, d.bnb_dz1), d.ydt, o + n);
          const float* mi = (const float*)rp(ab, d.bnb_mi);
          const float* tot = (const float*)rp(ab, d.bnb_totals);
          const float ga = ((const float*)rp(ab, d.bnb_gamma))[n], be = ((const float*)rp(ab, d.bnb_beta))[n], sl = *(const float*)rp(ab, d.bnb_slope);
          const float xh = (ld(rp(ab, d.bnb_y), d.ydt, o + n) - mi[n]) * mi[d.N + n];
          const float bn = ga * xh + be;
          const float dbn = bn > 0.f ? dz : sl * dz;
          g = (double)(ga * mi[d.N + n] * (dbn - tot[n] * d.bnb_inv_count - xh * (tot[d.N + n] * d.bnb_inv_count)));
        }
        if (g == 0.0) continue;
        double* a = &acc[(size_t)n * d.ldw];
        for (int k = 0; k < d.ldw; ++k) a[k] += g * arow[k];
      }
    }
    }
    for (int64_t i = 0; i < sz; ++i) part[sp * sz + i] = (float)acc[i];
  }
}


inline void load_dz(const BnBwdReduce& d, const AB& ab, int64_t r, int c, double* g) {
  const int64_t b = r / d.rpb, q = r - b * d.rpb;
  *g = 0.0;
  if (q >= d.skip) *g = ld(rp(ab, d.dz0), d.dt, (b * (d.rpb - d.skip) + q - d.skip) * d.C + c);
  if (d.dz1.arena >= 0) *g += ld(rp(ab, d.dz1), d.dt, r * d.C + c);
}


inline double load_dz_c(const CbnBwd& d, const AB& ab, int64_t r, int c) {
  const int64_t b = r / d.rpb, q = r - b * d.rpb;
  double g = 0.0;
  if (q >= d.skip) g = ld(rp(ab, d.dz0), d.dt, (b * (d.rpb - d.skip) + q - d.skip) * d.C + c);
  if (d.dz1.arena >= 0) g += ld(rp(ab, d.dz1), d.dt, r * d.C + c);
  return g;
}
// V^-1/2 of a symmetric positive definite 2 x 2 matrix (tools_for_model.py:567-576)
inline void inv_sqrt_2x2(double vrr, double vri, double vii, double& urr, double& uri, double& uii, double& s, double& t, double& rst) {
  const double tau = vrr + vii, delta = vrr * vii - vri * vri;
  s = std::sqrt(delta); t = std::sqrt(tau + 2 * s); rst = 1.0 / (s * t);
  urr = (s + vii) * rst; uii = (s + vrr) * rst; uri = -vri * rst;
}

inline float sgm(double x) { return (float)(1.0 / (1.0 + std::exp(-x))); }
inline uint32_t mix32(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x632BE5ABu) * 0xC2B2AE3Du;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
inline int reflect_idx(int f, int F) { return f < 0 ? -f : (f >= F ? 2 * (F - 1) - f : f); }
inline float sb_raw(const Fsn& d, const float* mt, const float* fbo, int t, int b, int f, int k) {
  const int n = (d.NB - 1) / 2;
  if (k < d.NB) return mt[((int64_t)t * d.B + b) * d.F + reflect_idx(f - n + k, d.F)];
  return fbo[((int64_t)t * d.B + b) * d.FP + f];
}
void means(const Fsn& d, const AB& ab, double count) {
  const float* sums = (const float*)rp(ab, d.sums);
  float* mu = (float*)rp(ab, d.aux2);
  for (int b = 0; b < d.B; ++b) { double s = 0; for (int f = 0; f < d.F; ++f) s += sums[b * d.F + f]; mu[b] = (float)(s / count); }
}

constexpr float kNormEps = 1.1920928955078125e-07f;
// one value through cfg.norm_type (struct Fsn): mu = per-utterance means (mode 0), stat = the statistics of FSN_NORMSTAT, r = frame*rows + row
inline float norm_apply(const Fsn& d, const AB& ab, float x, const float* mu, int b, int64_t r) {
  if (d.mode == 0) return x / (mu[b] + 1e-5f);
  const float* stt = (const float*)rp(ab, d.stat);
  if (d.mode == 2) return (x - stt[b]) / (stt[d.B + b] + 1e-5f);
  if (d.mode == 1) return x / (stt[r * 2] + kNormEps);
  return (x - stt[r * 2]) / stt[r * 2 + 1];
}

bool run_fsn(const Op& op, const AB& ab) {
  switch (op.kind) {
    case OP_CELL_FWD: {
      const LstmCell& d = op.cell;
      float* g = (float*)rp(ab, d.gates);
      if (d.kind == 1) {                                  // GRU (sefd_desc.h LstmCell kind 1)
        const float* gh = (const float*)rp(ab, d.gh);
        const int H = d.H;
        for (int64_t i = 0; i < d.rows * H; ++i) {
          const int64_t r = i / H; const int j = (int)(i % H);
          float* gr = g + r * 4 * H;
          const float* hr = gh + r * 3 * H;
          const double rg = sgm(gr[j] + hr[j]), zg = sgm(gr[H + j] + hr[H + j]), hn = hr[2 * H + j];
          const double ng = std::tanh((double)gr[2 * H + j] + rg * hn);
          const double hprev = d.first ? 0.0 : ld(rp(ab, d.c_prev), d.hdt, r * H + j);
          gr[j] = (float)rg; gr[H + j] = (float)zg; gr[2 * H + j] = (float)ng; gr[3 * H + j] = (float)hn;
          st(rp(ab, d.h), d.hdt, r * H + j, (float)((1 - zg) * ng + zg * hprev));
        }
        return true;
      }
      const float* cp = d.first ? nullptr : (const float*)rp(ab, d.c_prev);
      float* c = (float*)rp(ab, d.c);
      for (int64_t i = 0; i < d.rows * d.H; ++i) {
        const int64_t r = i / d.H; const int j = (int)(i % d.H);
        int64_t o[5];
        cell_off(d, r, o);
        float* gr = g + o[0];
        const int ci = cell_col(d, 0, j), cf = cell_col(d, 1, j), cg = cell_col(d, 2, j), co = cell_col(d, 3, j);
        const double ig = sgm(gr[ci]), fg = sgm(gr[cf]), gg = std::tanh((double)gr[cg]), og = sgm(gr[co]);
        const double cn = fg * (cp ? cp[o[1] + j] : 0.0) + ig * gg;
        gr[ci] = (float)ig; gr[cf] = (float)fg; gr[cg] = (float)gg; gr[co] = (float)og;
        c[o[1] + j] = (float)cn;
        st(rp(ab, d.h), d.hdt, o[2] + j, (float)(og * std::tanh(cn)));
      }
      return true;
    }
    case OP_CELL_BWD: {
      const LstmCell& d = op.cell;
      const float* g = (const float*)rp(ab, d.gates);
      if (d.kind == 1) {
        const float* dh = (const float*)rp(ab, d.dh);
        float* dhp = d.dc.arena >= 0 ? (float*)rp(ab, d.dc) : nullptr;
        const int H = d.H;
        for (int64_t i = 0; i < d.rows * H; ++i) {
          const int64_t r = i / H; const int j = (int)(i % H);
          const float* gr = g + r * 4 * H;
          const double rg = gr[j], zg = gr[H + j], ng = gr[2 * H + j], hn = gr[3 * H + j];
          const double hprev = d.c_prev.arena >= 0 ? ld(rp(ab, d.c_prev), d.hdt, r * H + j) : 0.0;
          const double dht = dh[r * H + j];
          const double dn = dht * (1 - zg) * (1 - ng * ng), dz = dht * (hprev - ng) * zg * (1 - zg), dr = dn * hn * rg * (1 - rg);
          st(rp(ab, d.dgates), d.gdt, r * 3 * H + j, (float)dr); st(rp(ab, d.dgates), d.gdt, r * 3 * H + H + j, (float)dz);
          st(rp(ab, d.dgates), d.gdt, r * 3 * H + 2 * H + j, (float)dn);
          st(rp(ab, d.gh), d.gdt, r * 3 * H + j, (float)dr); st(rp(ab, d.gh), d.gdt, r * 3 * H + H + j, (float)dz);
          st(rp(ab, d.gh), d.gdt, r * 3 * H + 2 * H + j, (float)(dn * rg));
          if (dhp) dhp[r * H + j] = (float)(dhp[r * H + j] + dht * zg);
        }
        return true;
      }
      const float* cp = d.c_prev.arena >= 0 ? (const float*)rp(ab, d.c_prev) : nullptr;
      const float* c = (const float*)rp(ab, d.c);
      const float* dh = (const float*)rp(ab, d.dh);
      float* dc = (float*)rp(ab, d.dc);
      for (int64_t i = 0; i < d.rows * d.H; ++i) {
        const int64_t r = i / d.H; const int j = (int)(i % d.H);
        int64_t o[5];
        cell_off(d, r, o);
        const float* gr = g + o[0];
        const int ci = cell_col(d, 0, j), cf = cell_col(d, 1, j), cg = cell_col(d, 2, j), co = cell_col(d, 3, j);
        const double ig = gr[ci], fg = gr[cf], gg = gr[cg], og = gr[co];
        const double tc = std::tanh((double)c[o[1] + j]), dht = dh[o[3] + j];
        const double dcv = dht * og * (1 - tc * tc) + (d.first ? 0.0 : dc[i]);
        st(rp(ab, d.dgates), d.gdt, o[4] + ci, (float)(dcv * gg * ig * (1 - ig)));
        st(rp(ab, d.dgates), d.gdt, o[4] + cf, (float)(dcv * (cp ? cp[o[1] + j] : 0.0) * fg * (1 - fg)));
        st(rp(ab, d.dgates), d.gdt, o[4] + cg, (float)(dcv * ig * (1 - gg * gg)));
        st(rp(ab, d.dgates), d.gdt, o[4] + co, (float)(dht * tc * og * (1 - og)));
        dc[i] = (float)(dcv * fg);
      }
      return true;
    }
    case OP_DROPOUT_FWD:
    case OP_DROPOUT_BWD: {
      const Dropout& d = op.drop;
      const uint32_t* seed = (const uint32_t*)rp(ab, d.seed);
      for (int64_t i = 0; i < d.n; ++i) {
        float sc = 1.f;
        if (d.keep < 1.f) {
          const uint32_t r = mix32(seed[0] + (uint32_t)d.layer * 0x51ED27u, seed[1] ^ (uint32_t)(i >> 32), (uint32_t)i);
          sc = ((r >> 8) * (1.f / 16777216.f)) < d.keep ? 1.f / d.keep : 0.f;
        }
        st(rp(ab, d.y), d.dt, i, ld(rp(ab, d.x), d.dt, i) * sc);
      }
      return true;
    }
    case OP_FSN_IN: {
      const Fsn& d = op.fsn;
      const float* in = (const float*)rp(ab, d.in);
      float* mt = (float*)rp(ab, d.out);
      float* sums = (float*)rp(ab, d.sums);
      for (int b = 0; b < d.B; ++b)
        for (int f = 0; f < d.F; ++f) {
          double s = 0;
          for (int t = 0; t < d.TP; ++t) {
            const float v = t < d.T ? in[((int64_t)b * d.F + f) * d.T + t] : 0.f;
            mt[((int64_t)t * d.B + b) * d.F + f] = v;
            s += v;
          }
          sums[b * d.F + f] = (float)s;
        }
      means(d, ab, (double)d.F * d.TP);
      return true;
    }
    case OP_FSN_SCALE: {
      const Fsn& d = op.fsn;
      const float* mt = (const float*)rp(ab, d.in);
      const float* mu = (const float*)rp(ab, d.sums);
      for (int64_t i = 0; i < (int64_t)d.TP * d.B * d.FP; ++i) {
        const int f = (int)(i % d.FP); const int64_t tb = i / d.FP; const int b = (int)(tb % d.B);
        float v = 0.f;
        if (f < d.F) v = norm_apply(d, ab, mt[tb * d.F + f], mu, b, tb);
        st(rp(ab, d.out), d.dt, i, v);
      }
      return true;
    }
    case OP_FSN_SBSUM:
    case OP_FSN_SBBWD_SUM: {
      const Fsn& d = op.fsn;
      const int W = d.NB + 1;
      float* sums = (float*)rp(ab, d.sums);
      for (int b = 0; b < d.B; ++b)
        for (int f = 0; f < d.F; ++f) {
          double s = 0;
          for (int t = 0; t < d.TP; ++t)
            for (int k = 0; k < W; ++k) {
              if (op.kind == OP_FSN_SBSUM) s += sb_raw(d, (const float*)rp(ab, d.in), (const float*)rp(ab, d.aux), t, b, f, k);
              else {
                const int64_t o = (((int64_t)t * d.B + b) * d.F + f) * W + k;
                s += (double)((const float*)rp(ab, d.in))[o] * ld(rp(ab, d.aux), d.dt, o);
              }
            }
          sums[b * d.F + f] = (float)s;
        }
      means(d, ab, (double)d.F * d.TP * W);
      return true;
    }
    case OP_FSN_SBBUILD: {
      const Fsn& d = op.fsn;
      const int W = d.NB + 1;
      const float* mu = (const float*)rp(ab, d.sums);
      for (int64_t i = 0; i < (int64_t)d.TP * d.B * d.F * W; ++i) {
        const int k = (int)(i % W); const int64_t r = i / W; const int f = (int)(r % d.F); const int64_t tb = r / d.F;
        const int b = (int)(tb % d.B), t = (int)(tb / d.B);
        st(rp(ab, d.out), d.dt, i, norm_apply(d, ab, sb_raw(d, (const float*)rp(ab, d.in), (const float*)rp(ab, d.aux), t, b, f, k), mu, b, r));
      }
      return true;
    }
    case OP_FSN_NORMSTAT: {
      const Fsn& d = op.fsn;
      const float* mt = (const float*)rp(ab, d.in);
      const float* fbo = d.src ? (const float*)rp(ab, d.aux) : nullptr;
      float* stt = (float*)rp(ab, d.stat);
      const int W = d.src ? d.NB + 1 : d.F, nf = d.src ? d.F : 1;
      auto val = [&](int t, int b, int f, int j) { return d.src ? sb_raw(d, mt, fbo, t, b, f, j) : mt[((int64_t)t * d.B + b) * d.F + j]; };
      for (int b = 0; b < d.B; ++b) {
        if (d.mode == 2) {
          double s_ = 0, q = 0;
          const double n = (double)d.TP * nf * W;
          for (int t = 0; t < d.TP; ++t) for (int f = 0; f < nf; ++f) for (int j = 0; j < W; ++j) { const double x = val(t, b, f, j); s_ += x; q += x * x; }
          const double mu = s_ / n, var = (q - n * mu * mu) / (n - 1);
          stt[b] = (float)mu; stt[d.B + b] = (float)std::sqrt(var > 0 ? var : 0.0);
          continue;
        }
        for (int f = 0; f < nf; ++f) {
          const int64_t rows = (int64_t)d.B * nf, row = (int64_t)b * nf + f;
          double cs = 0, cq = 0;
          for (int t = 0; t < d.TP; ++t) {
            for (int j = 0; j < W; ++j) { const double x = val(t, b, f, j); cs += x; cq += x * x; }
            const double n = (double)W * (t + 1), m = cs / n;
            stt[((int64_t)t * rows + row) * 2] = (float)m;
            stt[((int64_t)t * rows + row) * 2 + 1] = d.mode == 3 ? (float)std::sqrt((cq - 2 * m * cs) / n + m * m + (double)kNormEps) : 0.f;
          }
        }
      }
      return true;
    }
    case OP_FSN_NORMBWD: {
      const Fsn& d = op.fsn;
      const float* dsb = (const float*)rp(ab, d.in);
      const float* fbo = (const float*)rp(ab, d.aux);
      const float* stt = (const float*)rp(ab, d.stat);
      float* out = (float*)rp(ab, d.out);
      const int W = d.NB + 1;
      const int64_t rows = (int64_t)d.B * d.F;
      for (int b = 0; b < d.B; ++b) {
        if (d.mode == 2) {
          double S = 0, G = 0;
          float* part = (float*)rp(ab, d.sums);                 // the kernel's per-(b, f) partial sums are part of the op's output
          for (int f = 0; f < d.F; ++f) {
            double s1 = 0, g1 = 0;
            for (int t = 0; t < d.TP; ++t) for (int k = 0; k < W; ++k) {
              const int64_t o = (((int64_t)t * d.B + b) * d.F + f) * W + k;
              g1 += dsb[o]; s1 += (double)dsb[o] * ld(rp(ab, d.aux2), d.dt, o);
            }
            part[b * d.F + f] = (float)s1; part[(d.B + b) * d.F + f] = (float)g1;
            S += s1; G += g1;
          }
          const double N = (double)d.F * W * d.TP, sdv = stt[d.B + b], sden = sdv + 1e-5;
          for (int t = 0; t < d.TP; ++t) for (int f = 0; f < d.F; ++f) {
            const int64_t o = (((int64_t)t * d.B + b) * d.F + f) * W + d.NB;
            out[((int64_t)t * d.B + b) * d.F + f] = (float)((dsb[o] - G / N) / sden - (double)ld(rp(ab, d.aux2), d.dt, o) * S / ((N - 1) * sdv));
          }
          continue;
        }
        for (int f = 0; f < d.F; ++f) {
          const int64_t row = (int64_t)b * d.F + f;
          double accA = 0, accB = 0, accBM = 0;
          for (int t = d.TP - 1; t >= 0; --t) {
            const int64_t o = ((int64_t)t * rows + row) * W;
            double G = 0, S = 0;
            for (int k = 0; k < W; ++k) { G += dsb[o + k]; S += (double)dsb[o + k] * ld(rp(ab, d.aux2), d.dt, o + k); }
            const double m = stt[((int64_t)t * rows + row) * 2], sd = stt[((int64_t)t * rows + row) * 2 + 1], n = (double)W * (t + 1), gk = dsb[o + d.NB];
            double v;
            if (d.mode == 1) { const double den = m + (double)kNormEps; accA += S / (den * n); v = gk / den - accA; }
            else {
              const double bb = S / (n * sd * sd);
              accA += G / (n * sd); accB += bb; accBM += bb * m;
              v = gk / sd - accA - (double)fbo[((int64_t)t * d.B + b) * d.FP + f] * accB + accBM;
            }
            out[((int64_t)t * d.B + b) * d.F + f] = (float)v;
          }
        }
      }
      return true;
    }
    case OP_FSN_OUT: {
      const Fsn& d = op.fsn;
      const float* sbo = (const float*)rp(ab, d.in);
      float* crm = (float*)rp(ab, d.out);
      for (int64_t i = 0; i < (int64_t)d.B * d.F * d.T * 2; ++i) {
        const int cch = (int)(i & 1); const int64_t q = i >> 1; const int t = (int)(q % d.T); const int64_t bf = q / d.T;
        crm[i] = sbo[(((int64_t)(t + d.LA)) * d.B * d.F + bf) * 2 + cch];
      }
      return true;
    }
    case OP_FSN_OUT_BWD: {
      const Fsn& d = op.fsn;
      const float* g = (const float*)rp(ab, d.in);
      for (int64_t i = 0; i < (int64_t)d.TP * d.B * d.F * 2; ++i) {
        const int cch = (int)(i & 1); const int64_t q = i >> 1; const int64_t bf = q % ((int64_t)d.B * d.F); const int t = (int)(q / ((int64_t)d.B * d.F));
        st(rp(ab, d.out), d.dt, i, t >= d.LA ? g[(bf * d.T + (t - d.LA)) * 2 + cch] : 0.f);
      }
      return true;
    }
    case OP_STFT_FFT: {
      const StftFft& d = op.fft;
      const float* src = (const float*)rp(ab, d.src);
      const float* win = (const float*)rp(ab, d.win);
      float* spec = (float*)rp(ab, d.spec);
      std::vector<double> v(512);
      for (int64_t fr = 0; fr < (int64_t)d.B * d.T; ++fr) {
        const int64_t b = fr / d.T; const int t = (int)(fr % d.T);
        for (int n = 0; n < 512; ++n) { const int p = t * d.hop - d.off + n; v[n] = (p >= 0 && p < d.L) ? (double)src[b * d.L + p] * win[n] : 0.0; }
        float* o = spec + fr * 516;
        o[0] = o[1] = 0.f;
        double ge = 0, go = 0;
        for (int n = 0; n < 512; ++n) (n & 1 ? go : ge) += v[n];
        const float* cr = d.corr.arena >= 0 ? (const float*)rp(ab, d.corr) : nullptr;
        for (int k = 0; k <= 256; ++k) {
          double re = 0, im = 0;
          for (int n = 0; n < 512; ++n) { const double a = 2.0 * 3.14159265358979323846 * (double)((k * n) % 512) / 512.0; re += v[n] * std::cos(a); im -= v[n] * std::sin(a); }
          if (cr) { re = d.scale * (re - cr[k] * ge - cr[2 * 257 + k] * go); im = d.scale * (im - cr[257 + k] * ge - cr[3 * 257 + k] * go); }
          o[2 * (k + 1)] = (float)re; o[2 * (k + 1) + 1] = (float)im;
        }
        if (d.lp.arena >= 0) {                    // channel-padded copy [frames][258][8] in the activation dtype
          char* lp = rp(ab, d.lp);
          for (int slot = 0; slot < 258; ++slot)
            for (int ch = 0; ch < 8; ++ch) st(lp, d.lp_dt, (fr * 258 + slot) * 8 + ch, ch < 2 ? o[2 * slot + ch] : 0.f);
        }
      }
      return true;
    }
    case OP_ISTFT_FFT: {
      const IstftFft& d = op.ifft;
      const float* est = (const float*)rp(ab, d.est);
      const float* win = (const float*)rp(ab, d.win);
      const float* cr = (const float*)rp(ab, d.corr);
      float* frames = (float*)rp(ab, d.frames);
      for (int64_t fr = 0; fr < d.nframes; ++fr) {
        const float* in = est + fr * 516 + 2;
        double ce = 0, co = 0;
        for (int k = 0; k <= 256; ++k) {
          ce += in[2 * k] * (double)cr[k] + in[2 * k + 1] * (double)cr[257 + k];
          co += in[2 * k] * (double)cr[2 * 257 + k] + in[2 * k + 1] * (double)cr[3 * 257 + k];
        }
        for (int j = 0; j < d.W; ++j) {
          double sre = 0;
          for (int k = 0; k <= 256; ++k) {
            const double a = 2.0 * 3.14159265358979323846 * (double)((k * j) % 512) / 512.0;
            sre += in[2 * k] * std::cos(a) - in[2 * k + 1] * std::sin(a);
          }
          frames[fr * d.W + j] = (float)((sre - ((j & 1) ? co : ce)) * win[j] / 256.0);
        }
      }
      return true;
    }
    case OP_REFLECTPAD: {
      const ReflectPad& d = op.rpad;
      const float* src = (const float*)rp(ab, d.src);
      float* dst = (float*)rp(ab, d.dst);
      const int Lp = d.L + 2 * d.pad;
      for (int b = 0; b < d.B; ++b)
        for (int i = 0; i < Lp; ++i) { int j = i - d.pad; j = j < 0 ? -j : (j >= d.L ? 2 * (d.L - 1) - j : j); dst[(int64_t)b * Lp + i] = src[(int64_t)b * d.L + j]; }
      return true;
    }
    case OP_FSN_SBBWD_APPLY: {
      const Fsn& d = op.fsn;
      const float* dsb = (const float*)rp(ab, d.in);
      const float* fbo = (const float*)rp(ab, d.aux);
      const float* mu = (const float*)rp(ab, d.aux2);
      const float* Sm = (const float*)rp(ab, d.sums);
      const int W = d.NB + 1;
      for (int64_t i = 0; i < (int64_t)d.TP * d.B * d.FP; ++i) {
        const int f = (int)(i % d.FP); const int64_t tb = i / d.FP; const int b = (int)(tb % d.B);
        float v = 0.f;
        if (f < d.F) {
          if (d.mode == 0) { const float den = mu[b] + 1e-5f; v = dsb[(tb * d.F + f) * W + d.NB] / den - Sm[b] / den; }
          else v = dsb[tb * d.F + f];
          const float y = fbo[i];
          if (d.act == 1) v = y > 0.f ? v : 0.f; else if (d.act == 2) v *= (1.f - y * y); else if (d.act == 3) v = (y > 0.f && y < 6.f) ? v : 0.f;
        }
        st(rp(ab, d.out), d.dt, i, v);
      }
      return true;
    }
    default: return false;
  }
}

void run_op(const Op& op, const AB& ab) {
  if (run_fsn(op, ab)) return;
  switch (op.kind) {
    case OP_RUNGEMM: rungemm(op.g, ab); break;
    case OP_WGRAD: wgrad(op.g, ab); break;
    case OP_PACK: pack(op.pack, ab); break;
    case OP_PACKMULTI: {
      const Pack* e = (const Pack*)rp(ab, op.packm.entries);
      for (int i = 0; i < op.packm.count; ++i) pack(e[i], ab);
      break;
    }
    case OP_SPLITSUM: {
      const Unpack& d = op.unpack;
      float* base = (float*)rp(ab, d.part);
      const int64_t* tab = d.nseg > 0 ? (const int64_t*)rp(ab, d.start) : nullptr;
      for (int sgi = 0; sgi < (d.nseg > 0 ? d.nseg : 1); ++sgi) {
        float* part = tab ? base + tab[3 * sgi] : base;
        const int64_t n = tab ? tab[3 * sgi + 1] : d.n, stride = tab ? n : d.sstride, ns = tab ? tab[3 * sgi + 2] : d.nsplit;
        for (int64_t i = 0; i < n; ++i) {
          double s = 0;
          for (int64_t k = 0; k < ns; ++k) s += part[k * stride + i];
          part[i] = (float)s;
        }
      }
      break;
    }
    case OP_UNPACK: {
      const Unpack& d = op.unpack;
      const int32_t* start = (const int32_t*)rp(ab, d.start);
      const int32_t* ent = (const int32_t*)rp(ab, d.ent);
      const float* part = (const float*)rp(ab, d.part);
      float* dst = (float*)rp(ab, d.dst);
      for (int64_t j = 0; j < d.n; ++j) {
        double v = 0;
        for (int e = start[j]; e < start[j + 1]; ++e) { const int32_t t = ent[e]; if (t > 0) v += part[t - 1]; else if (t < 0) v -= part[-t - 1]; }
        if (start[j + 1] > start[j]) dst[j] = (float)v;
      }
      break;
    }
    case OP_BN_FINALIZE: {
      const BnFinalize& d = op.bnf;
      float* mi = (float*)rp(ab, d.mean_invstd);
      for (int c = 0; c < d.C; ++c) {
        if (d.nblk < 0) {
          mi[c] = ((const float*)rp(ab, d.running_mean))[c];
          mi[d.C + c] = 1.f / std::sqrt(((const float*)rp(ab, d.running_var))[c] + d.eps);
          continue;
        }
        const float* part = (const float*)rp(ab, d.part);
        double s1 = 0, s2 = 0;
        if (d.mode == 2) {
          const double* tot = (const double*)rp(ab, d.totals);
          s1 = tot[c]; s2 = tot[d.C + c];
        } else {
          for (int b = 0; b < d.nblk; ++b)
            for (int u = 0; u < (d.nsub > 1 ? d.nsub : 1); ++u) {
              s1 += part[((int64_t)b * 2) * d.Cpad + u * d.substride + c];
              s2 += part[((int64_t)b * 2 + 1) * d.Cpad + u * d.substride + c];
            }
          if (d.mode == 1) {
            double* tot = (double*)rp(ab, d.totals);
            tot[c] = s1; tot[d.C + c] = s2;
            continue;
          }
        }
        const double mean = s1 / d.count;
        double var = s2 / d.count - mean * mean;
        if (var < 0) var = 0;
        mi[c] = (float)mean;
        mi[d.C + c] = (float)(1.0 / std::sqrt(var + (double)d.eps));
        if (d.running_mean.arena >= 0) {
          float* rm = (float*)rp(ab, d.running_mean);
          float* rv = (float*)rp(ab, d.running_var);
          const double unb = var * (d.count / (d.count > 1 ? d.count - 1 : 1));
          rm[c] = (float)((1.0 - d.momentum) * rm[c] + d.momentum * mean);
          rv[c] = (float)((1.0 - d.momentum) * rv[c] + d.momentum * unb);
        }
      }
      break;
    }
    case OP_BN_APPLY: {
      const BnApply& d = op.bna;
      const float* mi = (const float*)rp(ab, d.mean_invstd);
      const float* gamma = (const float*)rp(ab, d.gamma);
      const float* beta = (const float*)rp(ab, d.beta);
      const float a = *(const float*)rp(ab, d.slope);
      for (int64_t i = 0; i < d.R * d.C; ++i) {
        const int c = (int)(i % d.C);
        const float bn = gamma[c] * ((ld(rp(ab, d.y), d.dt, i) - mi[c]) * mi[d.C + c]) + beta[c];
        st(rp(ab, d.z), d.dt, i, bn > 0.f ? bn : a * bn);
      }
      break;
    }
    case OP_BN_BWD_REDUCE: {
      const BnBwdReduce& d = op.bnr;
      const float* mi = (const float*)rp(ab, d.mean_invstd);
      const float* gamma = (const float*)rp(ab, d.gamma);
      const float* beta = (const float*)rp(ab, d.beta);
      const float a = *(const float*)rp(ab, d.slope);
      float* part = (float*)rp(ab, d.part);
      for (int blk = 0; blk < d.nblk; ++blk) {
        std::vector<double> s0(d.C, 0.0), s1(d.C, 0.0);
        double sa = 0;
        const int64_t r0 = (int64_t)blk * d.rows_per_blk, r1 = std::min<int64_t>(d.R, r0 + d.rows_per_blk);
        for (int64_t r = r0; r < r1; ++r)
          for (int c = 0; c < d.C; ++c) {
            double g;
            load_dz(d, ab, r, c, &g);
            const float xh = (ld(rp(ab, d.y), d.dt, r * d.C + c) - mi[c]) * mi[d.C + c];
            const float bn = gamma[c] * xh + beta[c];
            const double dbn = bn > 0.f ? g : a * g;
            if (!(bn > 0.f)) sa += bn * g;
            s0[c] += dbn; s1[c] += dbn * xh;
          }
        for (int c = 0; c < d.C; ++c) { part[(int64_t)blk * 3 * d.C + c] = (float)s0[c]; part[(int64_t)blk * 3 * d.C + d.C + c] = (float)s1[c]; }
        part[(int64_t)blk * 3 * d.C + 2 * d.C] = (float)sa;
        for (int c = 1; c < d.C; ++c) part[(int64_t)blk * 3 * d.C + 2 * d.C + c] = 0.f;
      }
      break;
    }
    case OP_BN_BWD_FINALIZE: {
      const BnBwdApply& d = op.bnb;
      const int C = d.r.C;
      const float* part = (const float*)rp(ab, d.r.part);
      float* tot = (float*)rp(ab, d.totals);
      const int ldp = d.r.ldp > 0 ? d.r.ldp : C;
      double sa = 0;
      for (int c = 0; c < C; ++c) {
        double s0 = 0, s1 = 0;
        for (int b = 0; b < d.r.nblk; ++b) { s0 += part[((int64_t)b * 3 + 0) * ldp + c]; s1 += part[((int64_t)b * 3 + 1) * ldp + c]; sa += part[((int64_t)b * 3 + 2) * ldp + c]; }
        tot[c] = (float)s0; tot[C + c] = (float)s1;
        ((float*)rp(ab, d.dbeta))[c] = (float)s0;
        ((float*)rp(ab, d.dgamma))[c] = (float)s1;
      }
      ((float*)rp(ab, d.dslope))[0] = (float)sa;
      break;
    }
    case OP_BN_BWD_APPLY: {
      const BnBwdApply& d = op.bnb;
      const BnBwdReduce& r = d.r;
      const int C = r.C;
      const float* mi = (const float*)rp(ab, r.mean_invstd);
      const float* gamma = (const float*)rp(ab, r.gamma);
      const float* beta = (const float*)rp(ab, r.beta);
      const float* tot = (const float*)rp(ab, d.totals);
      const float a = *(const float*)rp(ab, r.slope);
      for (int64_t row = 0; row < r.R; ++row)
        for (int c = 0; c < C; ++c) {
          double g;
          load_dz(r, ab, row, c, &g);
          const float xh = (ld(rp(ab, r.y), r.dt, row * C + c) - mi[c]) * mi[C + c];
          const float bn = gamma[c] * xh + beta[c];
          const double dbn = bn > 0.f ? g : a * g;
          st(rp(ab, d.dy), r.dt, row * C + c, (float)(gamma[c] * mi[C + c] * (dbn - tot[c] / d.count - xh * tot[C + c] / d.count)));
        }
      break;
    }
    // ---- ComplexBatchNorm (csrc/cbn.hip): same per-block partial sums (fp32 rows), fp64 finalize
    case OP_CBN_STATS: {
      const CbnFwd& d = op.cbf;
      const int h = d.C / 2;
      float* part = (float*)rp(ab, d.part);
      for (int blk = 0; blk < d.nblk; ++blk) {
        std::vector<double> s(5 * h, 0.0);
        const int64_t r0 = (int64_t)blk * d.rows_per_blk, r1 = std::min<int64_t>(d.R, r0 + d.rows_per_blk);
        for (int64_t r = r0; r < r1; ++r)
          for (int k = 0; k < h; ++k) {
            const double xr = ld(rp(ab, d.y), d.dt, r * d.C + k), xi = ld(rp(ab, d.y), d.dt, r * d.C + h + k);
            s[k] += xr; s[h + k] += xi; s[2 * h + k] += xr * xr; s[3 * h + k] += xi * xi; s[4 * h + k] += xr * xi;
          }
        for (int i = 0; i < 5 * h; ++i) part[(int64_t)blk * 5 * h + i] = (float)s[i];
      }
      break;
    }
    case OP_CBN_FINALIZE: {
      const CbnFwd& d = op.cbf;
      const int h = d.C / 2;
      float* coef = (float*)rp(ab, d.coef);
      for (int k = 0; k < h; ++k) {
        double mr, mi, vrr, vri, vii;
        if (d.training) {
          const float* part = (const float*)rp(ab, d.part);
          double t[5] = {0, 0, 0, 0, 0};
          for (int b = 0; b < d.nblk; ++b)
            for (int j = 0; j < 5; ++j) t[j] += part[(int64_t)b * 5 * h + j * h + k];
          mr = t[0] / d.count; mi = t[1] / d.count;
          vrr = std::max(0.0, t[2] / d.count - mr * mr); vii = std::max(0.0, t[3] / d.count - mi * mi); vri = t[4] / d.count - mr * mi;
          if (d.RM[0].arena >= 0) {
            float* rm[2] = {(float*)rp(ab, d.RM[0]), (float*)rp(ab, d.RM[1])};
            float* rv[3] = {(float*)rp(ab, d.RV[0]), (float*)rp(ab, d.RV[1]), (float*)rp(ab, d.RV[2])};
            rm[0][k] += d.momentum * ((float)mr - rm[0][k]); rm[1][k] += d.momentum * ((float)mi - rm[1][k]);
            rv[0][k] += d.momentum * ((float)vrr - rv[0][k]); rv[1][k] += d.momentum * ((float)vri - rv[1][k]); rv[2][k] += d.momentum * ((float)vii - rv[2][k]);
          }
        } else {
          mr = ((const float*)rp(ab, d.RM[0]))[k]; mi = ((const float*)rp(ab, d.RM[1]))[k];
          vrr = ((const float*)rp(ab, d.RV[0]))[k]; vri = ((const float*)rp(ab, d.RV[1]))[k]; vii = ((const float*)rp(ab, d.RV[2]))[k];
        }
        vrr += d.eps; vii += d.eps;
        double urr, uri, uii, s_, t_, rst;
        inv_sqrt_2x2(vrr, vri, vii, urr, uri, uii, s_, t_, rst);
        const double wrr = ((const float*)rp(ab, d.W[0]))[k], wri = ((const float*)rp(ab, d.W[1]))[k], wii = ((const float*)rp(ab, d.W[2]))[k];
        const double zrr = wrr * urr + wri * uri, zri = wrr * uri + wri * uii, zir = wri * urr + wii * uri, zii = wri * uri + wii * uii;
        const double br = ((const float*)rp(ab, d.Bv[0]))[k], bi = ((const float*)rp(ab, d.Bv[1]))[k];
        const double v[14] = {zrr, zri, zir, zii, br - zrr * mr - zri * mi, bi - zir * mr - zii * mi, mr, mi, urr, uri, uii, vrr, vri, vii};
        for (int j = 0; j < 14; ++j) coef[j * h + k] = (float)v[j];
      }
      break;
    }
    case OP_CBN_APPLY: {
      const CbnFwd& d = op.cbf;
      const int h = d.C / 2;
      const float* cf = (const float*)rp(ab, d.coef);
      const float a = *(const float*)rp(ab, d.slope);
      for (int64_t r = 0; r < d.R; ++r)
        for (int k = 0; k < h; ++k) {
          const float xr = ld(rp(ab, d.y), d.dt, r * d.C + k), xi = ld(rp(ab, d.y), d.dt, r * d.C + h + k);
          const float yr = cf[k] * xr + cf[h + k] * xi + cf[4 * h + k], yi = cf[2 * h + k] * xr + cf[3 * h + k] * xi + cf[5 * h + k];
          st(rp(ab, d.z), d.dt, r * d.C + k, yr > 0.f ? yr : a * yr);
          st(rp(ab, d.z), d.dt, r * d.C + h + k, yi > 0.f ? yi : a * yi);
        }
      break;
    }
    case OP_CBN_BWD_REDUCE: {
      const CbnBwd& d = op.cbb;
      const int h = d.C / 2;
      const float* cf = (const float*)rp(ab, d.coef);
      const float a = *(const float*)rp(ab, d.slope);
      float* part = (float*)rp(ab, d.part);
      for (int blk = 0; blk < d.nblk; ++blk) {
        std::vector<double> s(6 * h, 0.0);
        double sa = 0;
        const int64_t r0 = (int64_t)blk * d.rows_per_blk, r1 = std::min<int64_t>(d.R, r0 + d.rows_per_blk);
        for (int64_t r = r0; r < r1; ++r)
          for (int k = 0; k < h; ++k) {
            const float xr = ld(rp(ab, d.y), d.dt, r * d.C + k), xi = ld(rp(ab, d.y), d.dt, r * d.C + h + k);
            const double gr = load_dz_c(d, ab, r, k), gi = load_dz_c(d, ab, r, h + k);
            const float yr = cf[k] * xr + cf[h + k] * xi + cf[4 * h + k], yi = cf[2 * h + k] * xr + cf[3 * h + k] * xi + cf[5 * h + k];
            const double dr = yr > 0.f ? gr : a * gr, di = yi > 0.f ? gi : a * gi;
            if (!(yr > 0.f)) sa += yr * gr;
            if (!(yi > 0.f)) sa += yi * gi;
            const double cr = xr - cf[6 * h + k], ci = xi - cf[7 * h + k];
            s[k] += dr; s[h + k] += di; s[2 * h + k] += dr * cr; s[3 * h + k] += dr * ci; s[4 * h + k] += di * cr; s[5 * h + k] += di * ci;
          }
        for (int i = 0; i < 6 * h; ++i) part[(int64_t)blk * 7 * h + i] = (float)s[i];
        part[(int64_t)blk * 7 * h + 6 * h] = (float)sa;
        for (int k = 1; k < h; ++k) part[(int64_t)blk * 7 * h + 6 * h + k] = 0.f;
      }
      break;
    }
    case OP_CBN_BWD_FINALIZE: {
      const CbnBwd& d = op.cbb;
      const int h = d.C / 2;
      const float* cf = (const float*)rp(ab, d.coef);
      const float* part = (const float*)rp(ab, d.part);
      float* cb = (float*)rp(ab, d.coefb);
      double sa = 0;
      for (int b = 0; b < d.nblk; ++b) sa += part[(int64_t)b * 7 * h + 6 * h];
      ((float*)rp(ab, d.dslope))[0] = (float)sa;
      const double N = d.count;
      for (int k = 0; k < h; ++k) {
        double t[6] = {0, 0, 0, 0, 0, 0};
        for (int b = 0; b < d.nblk; ++b)
          for (int j = 0; j < 6; ++j) t[j] += part[(int64_t)b * 7 * h + j * h + k];
        const double dbr = t[0], dbi = t[1], dzrr = t[2], dzri = t[3], dzir = t[4], dzii = t[5];
        const double urr = cf[8 * h + k], uri = cf[9 * h + k], uii = cf[10 * h + k];
        const double wrr = ((const float*)rp(ab, d.W[0]))[k], wri = ((const float*)rp(ab, d.W[1]))[k], wii = ((const float*)rp(ab, d.W[2]))[k];
        ((float*)rp(ab, d.dB[0]))[k] = (float)dbr; ((float*)rp(ab, d.dB[1]))[k] = (float)dbi;
        ((float*)rp(ab, d.dW[0]))[k] = (float)(dzrr * urr + dzri * uri);
        ((float*)rp(ab, d.dW[1]))[k] = (float)(dzrr * uri + dzri * uii + dzir * urr + dzii * uri);
        ((float*)rp(ab, d.dW[2]))[k] = (float)(dzir * uri + dzii * uii);
        const double durr = wrr * dzrr + wri * dzir, duii = wri * dzri + wii * dzii, duri = (wrr * dzri + wri * dzii) + (wri * dzrr + wii * dzir);
        const double vrr = cf[11 * h + k], vri = cf[12 * h + k], vii = cf[13 * h + k];
        double u0, u1, u2, s_, t_, rst;
        inv_sqrt_2x2(vrr, vri, vii, u0, u1, u2, s_, t_, rst);
        double dvrr = 0, dvri = 0, dvii = 0;
        const double d_rst = durr * (s_ + vii) + duii * (s_ + vrr) + duri * (-vri);
        double d_s = (durr + duii) * rst;
        dvii += durr * rst; dvrr += duii * rst; dvri += -duri * rst;
        const double d_st = -d_rst * rst * rst;
        d_s += d_st * t_;
        const double d_t = d_st * s_;
        const double d_u = d_t / (2 * t_);
        d_s += 2 * d_u;
        const double d_delta = d_s / (2 * s_);
        dvrr += d_delta * vii + d_u; dvii += d_delta * vrr + d_u; dvri += -2 * vri * d_delta;
        const double v[9] = {cf[k], cf[h + k], cf[2 * h + k], cf[3 * h + k], dbr / N, dbi / N, 2 * dvrr / N, dvri / N, 2 * dvii / N};
        for (int j = 0; j < 9; ++j) cb[j * h + k] = (float)v[j];
      }
      break;
    }
    case OP_CBN_BWD_APPLY: {
      const CbnBwd& d = op.cbb;
      const int h = d.C / 2;
      const float* cf = (const float*)rp(ab, d.coef);
      const float* cb = (const float*)rp(ab, d.coefb);
      const float a = *(const float*)rp(ab, d.slope);
      for (int64_t r = 0; r < d.R; ++r)
        for (int k = 0; k < h; ++k) {
          const float xr = ld(rp(ab, d.y), d.dt, r * d.C + k), xi = ld(rp(ab, d.y), d.dt, r * d.C + h + k);
          const float gr = (float)load_dz_c(d, ab, r, k), gi = (float)load_dz_c(d, ab, r, h + k);
          const float yr = cf[k] * xr + cf[h + k] * xi + cf[4 * h + k], yi = cf[2 * h + k] * xr + cf[3 * h + k] * xi + cf[5 * h + k];
          const float dr = (yr > 0.f ? gr : a * gr) - cb[4 * h + k], di = (yi > 0.f ? gi : a * gi) - cb[5 * h + k];
          const float cr = xr - cf[6 * h + k], ci = xi - cf[7 * h + k];
          st(rp(ab, d.dy), d.dt, r * d.C + k, cf[k] * dr + cf[2 * h + k] * di + cb[6 * h + k] * cr + cb[7 * h + k] * ci);
          st(rp(ab, d.dy), d.dt, r * d.C + h + k, cf[h + k] * dr + cf[3 * h + k] * di + cb[7 * h + k] * cr + cb[8 * h + k] * ci);
        }
      break;
    }
    case OP_LSTM_FWD: {
      const LstmRec& d = op.lstm;
      const int H = d.H, T = d.T;
      const int sdt = d.impl == 1 ? d.gxdt : DT_F32;        // dtype of the gx / gates slabs (row-block kernels: optionally bf16)
      float* cs = (float*)rp(ab, d.c);
      for (int g = 0; g < d.G; ++g) {
        const float* whh_ = (const float*)rp(ab, d.whh[g % d.nset]);
        std::vector<float> whh(whh_, whh_ + (size_t)4 * H * H);
        if (d.hdt == DT_BF16) for (auto& x : whh) x = bf2f(f2bf(x));     // bf16 mode: recurrent operands are bf16
        for (int b = 0; b < d.B; ++b) {
          std::vector<double> h(H, 0.0), c(H, 0.0), hn(H);
          const int tb = d.t0, te = d.t1 > 0 ? d.t1 : T;
          if (tb > 0) {                                   // resume from the saved state of frame tb - 1
            const int64_t rp_ = (int64_t)g * d.B * T + (d.tmajor ? (int64_t)(tb - 1) * d.B + b : (int64_t)b * T + tb - 1);
            for (int j = 0; j < H; ++j) { h[j] = ld(rp(ab, d.h), d.hdt, rp_ * H + j); c[j] = cs[rp_ * H + j]; }
          }
          for (int t = tb; t < te; ++t) {
            const int64_t rt = d.tmajor ? (int64_t)t * d.B + b : (int64_t)b * T + t;      // (sequence, frame) -> row of the buffers
            const int64_t gxo = d.gx_goff[g] + rt * d.gx_ld;
            const int64_t row = (int64_t)g * d.B * T + rt;
            for (int j = 0; j < H; ++j) {
              double pre[4];
              for (int q = 0; q < 4; ++q) {
                double s;
                if (d.impl == 1 && d.xfeat > 0) {            // fused input projection: bias + x_t . W_ih^T from the fragment-ordered packed copy
                  const int c = gate_col(q, j), xf = d.xfeat;
                  s = ((const float*)rp(ab, d.bias))[c];
                  for (int k = 0; k < xf; ++k) s += (double)ld(rp(ab, d.xin), DT_BF16, rt * xf + k) * ld(rp(ab, d.wpk_x), DT_BF16, rows_wf_index(xf, c, k));
                } else {
                  s = ld(rp(ab, d.gx), sdt, gxo + gate_col(q, j));
                }
                const float* wr = whh.data() + (int64_t)(q * H + j) * H;
                for (int k = 0; k < H; ++k) s += h[k] * wr[k];
                pre[q] = s;
              }
              const double ig = 1 / (1 + std::exp(-pre[0])), fg = 1 / (1 + std::exp(-pre[1])), gg = std::tanh(pre[2]), og = 1 / (1 + std::exp(-pre[3]));
              c[j] = (double)(float)(fg * c[j] + ig * gg);      // the kernels carry the cell state in fp32 (and resume chunks from the stored value)
              hn[j] = og * std::tanh(c[j]);
              const int64_t gq = (row * H + j) * 4;
              st(rp(ab, d.gates), sdt, gq, (float)ig); st(rp(ab, d.gates), sdt, gq + 1, (float)fg);
              st(rp(ab, d.gates), sdt, gq + 2, (float)gg); st(rp(ab, d.gates), sdt, gq + 3, (float)og);
              cs[row * H + j] = (float)c[j];
            }
            for (int j = 0; j < H; ++j) {
              st(rp(ab, d.h), d.hdt, row * H + j, (float)hn[j]);
              h[j] = d.hdt == DT_BF16 ? bf2f(f2bf((float)hn[j])) : hn[j];
              if (d.impl == 1 && d.hd.arena >= 0) {           // fused inter-layer dropout: the same map as OP_DROPOUT_FWD on the stored h
                const uint32_t* seed = (const uint32_t*)rp(ab, d.seed);
                const int64_t i = row * H + j;
                float sc = 1.f;
                if (d.keep < 1.f) {
                  const uint32_t r = mix32(seed[0] + (uint32_t)d.drop_layer * 0x51ED27u, seed[1] ^ (uint32_t)(i >> 32), (uint32_t)i);
                  sc = ((r >> 8) * (1.f / 16777216.f)) < d.keep ? 1.f / d.keep : 0.f;
                }
                st(rp(ab, d.hd), d.hdt, i, ld(rp(ab, d.h), d.hdt, i) * sc);
              }
            }
          }
        }
      }
      break;
    }
    case OP_LSTM_BWD: {
      const LstmRec& d = op.lstm;
      const int H = d.H, T = d.T;
      const int sdt = d.impl == 1 ? d.gxdt : DT_F32;
      const float* cs = (const float*)rp(ab, d.c);
      const float* dh = (const float*)rp(ab, d.dh);
      for (int g = 0; g < d.G; ++g) {
        const float* whh_ = (const float*)rp(ab, d.whh[g % d.nset]);
        std::vector<float> whh(whh_, whh_ + (size_t)4 * H * H);
        if (d.hdt == DT_BF16) for (auto& x : whh) x = bf2f(f2bf(x));
        for (int b = 0; b < d.B; ++b) {
          std::vector<double> dhrec(H, 0.0), dc(H, 0.0), dg(4 * H);
          for (int t = T - 1; t >= 0; --t) {
            const int64_t rt = d.tmajor ? (int64_t)t * d.B + b : (int64_t)b * T + t;
            const int64_t row = (int64_t)g * d.B * T + rt, rowp = row - (d.tmajor ? d.B : 1);
            for (int j = 0; j < H; ++j) {
              const int64_t gq = (row * H + j) * 4;
              const double ig = ld(rp(ab, d.gates), sdt, gq), fg = ld(rp(ab, d.gates), sdt, gq + 1), gg = ld(rp(ab, d.gates), sdt, gq + 2), og = ld(rp(ab, d.gates), sdt, gq + 3);
              const double ct = cs[row * H + j], cp = t > 0 ? cs[rowp * H + j] : 0.0;
              double dup;
              if (d.impl == 1 && d.no == 2) {                 // rank-2 upstream gradient from the 2-output head
                const float* wo = (const float*)rp(ab, d.wo);
                dup = (float)(ld(rp(ab, d.dyo), d.gdt, row * 2) * wo[j] + ld(rp(ab, d.dyo), d.gdt, row * 2 + 1) * wo[H + j]);
              } else {
                dup = (d.impl == 1 && d.dhdt == DT_BF16) ? (double)ld(rp(ab, d.dh), DT_BF16, row * H + j) : (double)dh[row * H + j];
              }
              if (d.impl == 1 && d.seed.arena >= 0 && d.keep < 1.f) {   // fused inter-layer dropout backward: dh is the gradient of the dropped h
                const uint32_t* seed = (const uint32_t*)rp(ab, d.seed);
                const int64_t i = row * H + j;
                const uint32_t r = mix32(seed[0] + (uint32_t)d.drop_layer * 0x51ED27u, seed[1] ^ (uint32_t)(i >> 32), (uint32_t)i);
                dup = (float)dup * (((r >> 8) * (1.f / 16777216.f)) < d.keep ? 1.f / d.keep : 0.f);
              }
              const double dht = dup + dhrec[j];
              const double tc = std::tanh(ct);
              const double dcv = dht * og * (1 - tc * tc) + dc[j];
              dg[j] = dcv * gg * ig * (1 - ig);
              dg[H + j] = dcv * cp * fg * (1 - fg);
              dg[2 * H + j] = dcv * ig * (1 - gg * gg);
              dg[3 * H + j] = dht * tc * og * (1 - og);
              dc[j] = dcv * fg;
            }
            const int64_t o = d.gx_goff[g] + rt * d.gx_ld;
            for (int k = 0; k < 4 * H; ++k) st(rp(ab, d.dgates), d.gdt, o + gate_col(k / H, k % H), (float)dg[k]);
            for (int j = 0; j < H; ++j) {
              double s = 0;
              if (t > 0) for (int k = 0; k < 4 * H; ++k) s += (d.hdt == DT_BF16 ? (double)bf2f(f2bf((float)dg[k])) : dg[k]) * whh[(int64_t)k * H + j];
              dhrec[j] = s;
            }
          }
        }
      }
      break;
    }
    case OP_COMBINE_FWD: {
      const Combine& d = op.comb;
      const int64_t gsz = d.rows * d.H;
      for (int64_t i = 0; i < gsz; ++i) {
        const int64_t row = i / d.H; const int j = (int)(i % d.H);
        if (d.t1 > 0) { const int t = (int)(row % d.T); if (t < d.t0 || t >= d.t1) continue; }
        const char* h = rp(ab, d.h);
        st(rp(ab, d.out), d.dt, row * 2 * d.H + j, ld(h, d.dt, i) - ld(h, d.dt, 3 * gsz + i));
        st(rp(ab, d.out), d.dt, row * 2 * d.H + d.H + j, ld(h, d.dt, 2 * gsz + i) + ld(h, d.dt, gsz + i));
      }
      break;
    }
    case OP_COMBINE_BWD: {
      const Combine& d = op.comb;
      const int64_t gsz = d.rows * d.H;
      float* dh = (float*)rp(ab, d.h);
      const float* dout = (const float*)rp(ab, d.out);
      for (int64_t i = 0; i < gsz; ++i) {
        const int64_t row = i / d.H; const int j = (int)(i % d.H);
        const float dr = dout[row * 2 * d.H + j], di = dout[row * 2 * d.H + d.H + j];
        dh[i] = dr; dh[gsz + i] = di; dh[2 * gsz + i] = di; dh[3 * gsz + i] = -dr;
      }
      break;
    }
    case OP_MASK_FWD: {
      const Mask& d = op.mask;
      const float* spec = (const float*)rp(ab, d.spec);
      float* est = (float*)rp(ab, d.est);
      const int NS = d.NF + 1;
      for (int64_t f = 0; f < d.frames; ++f)
        for (int slot = 0; slot < NS; ++slot) {
          const int64_t i = f * NS + slot;
          double er = 0, ei = 0, emag = 0;
          if (slot >= 2) {
            const int64_t b = f / d.T, t = f % d.T;
            const int64_t mo = b * d.mask_bstride + t * d.mask_fstride + d.mask_base + (slot - 2) * d.mch;
            const double mr = ld(rp(ab, d.mask), d.mdt, mo), mi = d.mch >= 2 ? ld(rp(ab, d.mask), d.mdt, mo + 1) : 0.0;
            const double sr = spec[i * 2], si = spec[i * 2 + 1];
            if (d.mode == 3) {
              emag = std::tanh(mr) * std::sqrt(sr * sr + si * si);
              const double ph = std::atan2(si, sr);
              er = emag * std::cos(ph); ei = emag * std::sin(ph);
            } else if (d.mode == 5) {
              emag = mr;
              const double ph = std::atan2(si, sr);
              er = emag * std::cos(ph); ei = emag * std::sin(ph);
            } else if (d.mode == 0) {
              const double mag = std::sqrt(sr * sr + si * si + 1e-8), ph = std::atan2(si, sr), mm = std::sqrt(mr * mr + mi * mi);
              const double mph = std::atan2(mi / (mm + 1e-8), mr / (mm + 1e-8));
              const double em = std::tanh(mm) * mag;
              er = em * std::cos(ph + mph); ei = em * std::sin(ph + mph);
            } else if (d.mode == 1) { er = sr * mr - si * mi; ei = sr * mi + si * mr; }
            else if (d.mode == 4) { er = mr; ei = mi; }
            else { er = sr * mr; ei = si * mi; }
          }
          est[i * 2] = (float)er; est[i * 2 + 1] = (float)ei;
          if ((d.mode == 3 || d.mode == 5) && slot >= 1) ((float*)rp(ab, d.estm))[f * d.NF + slot - 1] = (float)emag;
        }
      break;
    }
    case OP_MASK_BWD: {
      const Mask& d = op.mask;
      const float* spec = (const float*)rp(ab, d.spec);
      const float* dest = (const float*)rp(ab, d.dest);
      const int NB = d.NF - 1, NS = d.NF + 1;
      const int lead = (int)(d.mask_base / d.mask_fstride), TT = d.T + lead;
      const int64_t B = d.frames / d.T;
      // column sums of dmask: the kernel's workgroup w (256 threads, grid stride) owns the elements i with (i / 256) % rows == w and leaves
      // its share in row w - mirrored here so that the partial-sum buffer compares element by element
      const int csr = d.colsum_rows > 0 ? d.colsum_rows : 1;
      std::vector<double> csum(2 * (size_t)csr, 0.0);
      for (int64_t b = 0; b < B; ++b)
        for (int u = 0; u < TT; ++u)
          for (int k = 0; k < NB; ++k) {
            const int64_t mo = b * d.mask_bstride + (int64_t)u * d.mask_fstride + k * d.mch;
            double gr = 0, gi = 0;
            if (u >= lead) {
              const int64_t f = b * d.T + (u - lead);
              const int64_t s_ = (f * NS + k + 2) * 2;
              const double sr = spec[s_], si = spec[s_ + 1], der = dest[s_], dei = dest[s_ + 1];
              const double mr = ld(rp(ab, d.mask), d.mdt, mo), mi = d.mch >= 2 ? ld(rp(ab, d.mask), d.mdt, mo + 1) : 0.0;
              if (d.mode == 3 || d.mode == 5) {
                const double tm = std::tanh(mr), ph = std::atan2(si, sr);
                double d_em = der * std::cos(ph) + dei * std::sin(ph);
                if (d.destm.arena >= 0) d_em += ((const float*)rp(ab, d.destm))[f * d.NF + k + 1];
                gr = d.mode == 3 ? d_em * std::sqrt(sr * sr + si * si) * (1 - tm * tm) : d_em;
              } else if (d.mode == 0) {
                const double mag = std::sqrt(sr * sr + si * si + 1e-8), ph = std::atan2(si, sr), mm = std::sqrt(mr * mr + mi * mi);
                const double den = mm + 1e-8, rpv = mr / den, ipv = mi / den, mph = std::atan2(ipv, rpv), tm = std::tanh(mm), em = tm * mag;
                const double sn = std::sin(ph + mph), cs = std::cos(ph + mph);
                const double d_em = der * cs + dei * sn, d_ph = em * (-der * sn + dei * cs);
                double d_mm = d_em * mag * (1 - tm * tm);
                const double q = rpv * rpv + ipv * ipv;
                double d_rp = 0, d_ip = 0;
                if (q > 0) { d_ip = d_ph * rpv / q; d_rp = -d_ph * ipv / q; }
                gr = d_rp / den; gi = d_ip / den;
                d_mm += -(d_rp * mr + d_ip * mi) / (den * den);
                if (mm > 0) { gr += d_mm * mr / mm; gi += d_mm * mi / mm; }
              } else if (d.mode == 1) { gr = der * sr + dei * si; gi = -der * si + dei * sr; }
              else if (d.mode == 4) { gr = der; gi = dei; }
              else { gr = der * sr; gi = dei * si; }
            }
            st(rp(ab, d.dmask), d.mdt, mo, (float)gr);
            if (d.mch >= 2) st(rp(ab, d.dmask), d.mdt, mo + 1, (float)gi);
            const int64_t wg = ((((b * TT + u) * NB + k) / 256) % csr);
            csum[2 * wg] += ld(rp(ab, d.dmask), d.mdt, mo);
            if (d.mch >= 2) csum[2 * wg + 1] += ld(rp(ab, d.dmask), d.mdt, mo + 1);
          }
      if (d.colsum_rows > 0) {
        float* cs = (float*)rp(ab, d.colsum);
        std::fill(cs, cs + (int64_t)d.colsum_rows * 8, 0.f);
        for (int w = 0; w < d.colsum_rows; ++w) { cs[w * 8] = (float)csum[2 * w]; cs[w * 8 + 1] = (float)csum[2 * w + 1]; }
      }
      break;
    }
    case OP_OLA_FWD: {
      const Ola& d = op.ola;
      const float* fr = (const float*)rp(ab, d.frames);
      const float* coff = (const float*)rp(ab, d.coff);
      float* wav = (float*)rp(ab, d.wav);
      for (int b = 0; b < d.B; ++b)
        for (int n = 0; n < d.L; ++n) {
          const int p = n + d.trim;
          double s = 0;
          for (int t = 0; t < d.T; ++t) { const int j = p - t * d.hop; if (j >= 0 && j < d.win) s += fr[((int64_t)b * d.T + t) * d.win + j]; }
          float v = (float)s / (coff[p] + 1e-8f);
          wav[(int64_t)b * d.L + n] = d.noclamp ? v : std::fmin(1.f, std::fmax(-1.f, v));
        }
      break;
    }
    case OP_OLA_BWD: {
      const Ola& d = op.ola;
      const float* coff = (const float*)rp(ab, d.coff);
      const float* wav = (const float*)rp(ab, d.wav);
      const float* dwav = (const float*)rp(ab, d.dwav);
      float* dpad = (float*)rp(ab, d.dpad);
      const int Lp = (d.T - 1) * d.hop + d.win;
      for (int b = 0; b < d.B; ++b)
        for (int p = 0; p < Lp; ++p) {
          float v = 0.f;
          const int s = p - d.trim;
          if (s >= 0 && s < d.L) { const float w = wav[(int64_t)b * d.L + s]; if (w > -1.f && w < 1.f) v = dwav[(int64_t)b * d.L + s] / (coff[p] + 1e-8f); }
          dpad[(int64_t)b * Lp + p] = v;
        }
      break;
    }
    case OP_SPECOUT_FWD:
    case OP_SPECOUT_BWD: {
      const SpecOut& d = op.so;
      float* est = (float*)rp(ab, d.est);
      float* orr = (float*)rp(ab, d.out_real);
      float* oi = d.mode == 0 ? (float*)rp(ab, d.out_imag) : nullptr;
      const int NS = d.NF + 1;
      for (int b = 0; b < d.B; ++b)
        for (int t = 0; t < d.T; ++t)
          for (int k = 0; k < d.NF; ++k) {
            const int64_t e = (((int64_t)b * d.T + t) * NS + k + 1) * 2, o = ((int64_t)b * d.NF + k) * d.T + t;
            if (op.kind == OP_SPECOUT_BWD && d.mode == 2) { est[((int64_t)b * d.T + t) * d.NF + k] = orr[o]; continue; }
            if (op.kind == OP_SPECOUT_BWD && d.mode == 3) { est[e] = orr[2 * o]; est[e + 1] = orr[2 * o + 1]; continue; }
            if (op.kind == OP_SPECOUT_FWD && d.mode == 3) { orr[2 * o] = est[e]; orr[2 * o + 1] = est[e + 1]; continue; }
            if (op.kind == OP_SPECOUT_FWD && d.mode == 2) { orr[o] = est[((int64_t)b * d.T + t) * d.NF + k]; continue; }
            if (op.kind == OP_SPECOUT_FWD && d.mode == 1) { orr[o] = std::sqrt(est[e] * est[e] + est[e + 1] * est[e + 1]); continue; }
            if (op.kind == OP_SPECOUT_FWD) { orr[o] = est[e]; oi[o] = est[e + 1]; }
            else if (d.accumulate) { est[e] += orr[o]; est[e + 1] += oi[o]; }
            else { est[e] = orr[o]; est[e + 1] = oi[o]; }
          }
      break;
    }
    case OP_SPECPAD: {
      const Mags& d = op.mags;
      const float* spec = (const float*)rp(ab, d.spec);
      for (int64_t i = 0; i < d.frames * d.NF * d.MS; ++i) { const int ch = (int)(i % d.MS); st(rp(ab, d.mags), d.dt, i, ch < 2 ? spec[(i / d.MS) * 2 + ch] : 0.f); }
      break;
    }
    case OP_MAGS: {
      const Mags& d = op.mags;
      const float* spec = (const float*)rp(ab, d.spec);
      const int NS = d.NF + 1;
      for (int64_t f = 0; f < d.frames; ++f)
        for (int j = 0; j < d.MS; ++j) {
          const int k = j - d.MO;
          float v = 0.f;
          if (k >= 0 && k < d.NF) { const float sr = spec[(f * NS + k + 1) * 2], si = spec[(f * NS + k + 1) * 2 + 1]; v = std::sqrt(sr * sr + si * si); }
          st(rp(ab, d.mags), d.dt, f * d.MS + j, v);
        }
      break;
    }
    case OP_MEMSET: std::memset(rp(ab, op.ms.dst), 0, op.ms.bytes); break;
    default: std::fprintf(stderr, "hostsim: unknown op %d\n", op.kind);
  }
}
}  // namespace

extern "C" int hostsim_run(const void* ops_, int first, int last, void* const* arenas) {
  const Op* ops = (const Op*)ops_;
  AB ab;
  for (int a = 0; a < A_COUNT; ++a) ab.p[a] = (char*)arenas[a];
  for (int i = first; i < last; ++i) run_op(ops[i], ab);
  return 0;
}
extern "C" int hostsim_op_size() { return (int)sizeof(Op); }
// job order of the ticket-drawn recurrence launches (sefd_desc.h rows_pair_job / rows_bwd_job): out = {layer, chunk, block, tb, te}
extern "C" void hostsim_rows_job(int bwd, int job, int nblk, int C, int T, int* out) {
  const RowsJob r = bwd ? rows_bwd_job(job, nblk, C, T) : rows_pair_job(job, nblk, C, T);
  out[0] = r.layer; out[1] = r.chunk; out[2] = r.block; out[3] = r.tb; out[4] = r.te;
}
