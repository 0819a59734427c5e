// Planner: DCCRN (models.py:15-284 of the reference) -> op list over channels-last buffers.
//
// Layout decisions (MI355X-first, not the reference's NCHW):
//   * every activation is channels-last  [B][T(+1)][F][C]  so that each A-row of the implicit GEMM is a few contiguous
//     runs (all 5 frequency taps x C channels of one frame are ONE run) -> 16-byte coalesced loads, no im2col;
//   * a complex conv is one real GEMM with the block weight [[Wr,-Wi],[Wi,Wr]] (same MACs as the reference's 4 convs);
//   * the transposed conv is two dense sub-pixel GEMMs (even / odd output rows), never a scatter;
//   * complex_cat / chunk / permute / reshape glue of the reference (23 % of its CPU step) is index arithmetic in the
//     run descriptors: the skip connection is a second source pointer, the LSTM feature order c*D+d is a weight permutation;
//   * decoder buffers keep the extra frame that `out[..., 1:]` drops, because BatchNorm statistics include it.
#include "plan.h"
#include "tuning.h"

#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstring>
#include <functional>

namespace sefd {
namespace {

constexpr double kPi = 3.14159265358979323846;
inline int64_t rup(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// Runs of a RUNGEMM / WGRAD are whole, 16-byte aligned chunks (then the kernels use the LDS-DMA loaders).  Arena buffers are
// 256-byte aligned, so only element offsets matter.  WGRAD: the upstream-gradient operand must be chunk aligned as well.
static bool runs_aligned(const RunGemm& g, bool is_wgrad) {
  const int vec = 16 / esize(g.xdt);
  bool ok = true;
  for (int s = 0; s < g.nseg && ok; ++s) {
    const Seg& sg = g.seg[s];
    if (sg.src < 0) { ok = is_wgrad; continue; }          // the ones run exists only in WGRAD
    const int q = sg.src;
    ok = sg.off % vec == 0 && sg.len % vec == 0 && g.fstride[q] % vec == 0 && g.base[q] % vec == 0 && g.rowlen[q] % vec == 0 &&
         g.tstride[q] % vec == 0 && g.bstride[q] % vec == 0 && (g.x[q].off % 16) == 0;
  }
  if (is_wgrad)
    ok = ok && g.xdt == DT_BF16 && g.ydt == DT_BF16 && g.N % 8 == 0 && g.y_off % 8 == 0 && g.y_fstride % 8 == 0 && g.y_tstride % 8 == 0 &&
         g.y_bstride % 8 == 0 && (g.y.off % 16) == 0;
  return ok;
}

struct Builder {
  Plan* P;
  ModelConfig c;
  int64_t ws_off = 0;
  int64_t io_off = 0;
  std::map<std::string, int> pidx, sidx;

  // gradient partial region (allocated at the end) and the inverse (unpack) table
  int64_t gp_off = 0;                                     // floats
  struct Fix { int op; int64_t rel; int which; };         // which: 0 -> op.g.w, 1 -> op.unpack.part, 2 -> op.mask.colsum
  std::vector<Fix> fixes;
  std::vector<std::vector<int32_t>> inv;                  // per trainable element: signed (gp-relative position + 1)
  std::vector<char> zero_grad;                            // elements without any contribution that UNPACK still writes (an exact 0)

  Ptr mk(int arena, int64_t off) { Ptr p; p.arena = arena; p.pad_ = 0; p.off = off; return p; }
  Ptr none() { return mk(A_NONE, 0); }

  Ptr ws(const std::string& name, int64_t elems, int dt) {
    const int64_t bytes = rup(elems * esize(dt), 256);
    Ptr p = mk(A_WS, ws_off);
    P->bufs[name] = BufInfo{ws_off, elems * esize(dt), dt};
    ws_off += bytes;
    return p;
  }
  Ptr io(const std::string& name, int64_t elems) {
    Ptr p = mk(A_IO, io_off);
    P->bufs["io." + name] = BufInfo{io_off, elems * 4, DT_F32};
    io_off += rup(elems * 4, 256);
    return p;
  }
  Ptr cst(const void* data, int64_t bytes) {
    const int64_t off = rup((int64_t)P->consts.size(), 256);
    P->consts.resize(off + bytes);
    std::memcpy(P->consts.data() + off, data, bytes);
    return mk(A_CONST, off);
  }
  void add_param(const std::string& name, std::vector<int64_t> shape, bool trainable) {
    ParamInfo pi;
    pi.name = name;
    pi.shape = shape;
    pi.numel = 1;
    for (auto s : shape) pi.numel *= s;
    auto& vec = trainable ? P->params : P->state;
    pi.arena = trainable ? A_PARAM : A_STATE;
    pi.off = vec.empty() ? 0 : vec.back().off + vec.back().numel;
    (trainable ? pidx : sidx)[name] = (int)vec.size();
    vec.push_back(pi);
  }
  const ParamInfo& par(const std::string& n) const {
    auto it = pidx.find(n);
    if (it == pidx.end()) { P->error = "missing param " + n; static ParamInfo z; return z; }
    return P->params[it->second];
  }
  Ptr pptr(const std::string& n, int arena = A_PARAM) { return mk(arena, par(n).off * 4); }
  Ptr sptr(const std::string& n) { return mk(A_STATE, P->state[sidx.at(n)].off * 4); }

  int wg_rounds = 1;                                       // see wgrad(): row splits sized for this many dispatch rounds
  int cur_lane = 0;
  int cur_hold = 0;                                        // lane-1 ops pushed while set wait for the NEXT recurrence launch (kOpHold)
  Op& push(std::vector<Op>& v, int kind, int tag) {
    Op op;
    std::memset(&op, 0, sizeof(op));
    op.kind = kind;
    op.tag = tag;
    op.lane = cur_lane;
    if (cur_lane == 1 && cur_hold) op.join = kOpHold;
    v.push_back(op);
    return v.back();
  }

  static RunGemm gemm0() {
    RunGemm g;
    std::memset(&g, 0, sizeof(g));
    g.x[0].arena = g.x[1].arena = g.w.arena = g.bias.arena = g.y.arena = g.stats.arena = g.y2.arena = g.bnb_dz1.arena = g.bnb_totals.arena = A_NONE;
    g.nsplit = 1;
    return g;
  }
  // lay out run segments: assigns koff (padded to the K-tile of the operand dtype) and ldw
  static void layout_segs(RunGemm& g) {
    const int bk = bk_of(g.xdt);
    int k = 0;
    for (int s = 0; s < g.nseg; ++s) { g.seg[s].koff = k; k += (int)rup(g.seg[s].len, bk); }
    g.ldw = k;
    g.Npad = (int)rup(g.N, bn_of(g.N));
  }

  using Coef = std::function<int32_t(int n, int seg, int j)>;   // signed 1-based flat param element, 0 = structural zero

  // PACK op for the weights of `g` (fills g.w), optional bias table (width 2) -> g.bias
  void pack_weights(std::vector<Op>& ops, RunGemm& g, const Coef& coef, const std::string& name, int tag,
                    const std::function<void(int n, int32_t out[2])>* bias = nullptr) {
    std::vector<int32_t> tab((size_t)g.Npad * g.ldw, 0);
    for (int n = 0; n < g.N; ++n)
      for (int s = 0; s < g.nseg; ++s)
        if (g.seg[s].src >= 0)
          for (int j = 0; j < g.seg[s].len; ++j) tab[(size_t)n * g.ldw + g.seg[s].koff + j] = coef(n, s, j);
    g.w = ws("w." + name, (int64_t)tab.size(), g.xdt);
    Op& op = push(ops, OP_PACK, tag);
    op.pack.tab = cst(tab.data(), (int64_t)tab.size() * 4);
    op.pack.src = mk(A_PARAM, 0);
    op.pack.dst = g.w;
    op.pack.n = (int64_t)tab.size();
    op.pack.ddt = g.xdt;
    op.pack.width = 1;
    if (bias) {
      std::vector<int32_t> bt((size_t)g.N * 2, 0);
      for (int n = 0; n < g.N; ++n) (*bias)(n, &bt[(size_t)n * 2]);
      g.bias = ws("b." + name, g.N, DT_F32);
      Op& ob = push(ops, OP_PACK, tag);
      ob.pack.tab = cst(bt.data(), (int64_t)bt.size() * 4);
      ob.pack.src = mk(A_PARAM, 0);
      ob.pack.dst = g.bias;
      ob.pack.n = g.N;
      ob.pack.ddt = DT_F32;
      ob.pack.width = 2;
    }
  }

  // Fused FFT STFT (stft_fft.hip) for fft_len == 512: frame t reads src[t*hop - off + j] * win[j], j < W.  Returns false
  // (caller plans the framing GEMM instead) for other transform sizes.
  bool stft_fft(std::vector<Op>& ops, int tag, Ptr src, Ptr spec, int B, int L, int T, int hop, int off, int NFFT,
                const std::vector<double>& win) {
    if (NFFT != 512 || (int)win.size() > 512 || tune_str("STFT_GEMM")) return false;
    if (fft_tw.arena < 0) {
      std::vector<float> tw(1024);
      for (int k = 0; k < 512; ++k) { tw[2 * k] = (float)std::cos(2.0 * kPi * k / 512.0); tw[2 * k + 1] = (float)std::sin(2.0 * kPi * k / 512.0); }
      fft_tw = cst(tw.data(), 4096);
    }
    std::vector<float> wf(512, 0.f);
    for (size_t j = 0; j < win.size(); ++j) wf[j] = (float)win[j];
    Op& op = push(ops, OP_STFT_FFT, tag);
    op.fft.src = src; op.fft.spec = spec; op.fft.tw = fft_tw; op.fft.win = cst(wf.data(), 2048);
    op.fft.B = B; op.fft.L = L; op.fft.T = T; op.fft.hop = hop; op.fft.off = off; op.fft.lp_dt = 0;
    op.fft.corr = none(); op.fft.scale = 1.f; op.fft.lp = none(); op.fft.pair = fft_pair();
    return true;
  }
  // two frames per transform in the bf16 plans only (stft_fft.hip); SEFD_STFT_PAIR=0 / 1 forces one form (A/B runs)
  int fft_pair() const { return tune_str("STFT_PAIR") ? atoi(tune_str("STFT_PAIR")) != 0 : c.act_dtype == DT_BF16; }
  Ptr fft_tw = Ptr{-1, 0, 0};
  Ptr fft_corr = Ptr{-1, 0, 0};
  // rank-2 correction of the closed-form pinv synthesis basis: cE/cO[part][k] = sum over even/odd j < W of the
  // un-windowed analysis basis, divided by (NFFT/2 + number of such j)
  Ptr istft_corr(int W) {
    if (fft_corr.arena >= 0) return fft_corr;
    std::vector<float> c(4 * 257, 0.f);
    const double ne = (W + 1) / 2, no = W / 2;
    for (int part = 0; part < 2; ++part)
      for (int k = 0; k <= 256; ++k) {
        double se = 0, so = 0;
        for (int m = 0; m < W; ++m) {
          const double ang = 2.0 * kPi * (double)(((int64_t)k * m) % 512) / 512.0;
          (m % 2 == 0 ? se : so) += part == 0 ? std::cos(ang) : -std::sin(ang);
        }
        c[(0 * 2 + part) * 257 + k] = (float)(se / (256.0 + ne));
        c[(1 * 2 + part) * 257 + k] = (float)(so / (256.0 + no));
      }
    fft_corr = cst(c.data(), (int64_t)c.size() * 4);
    return fft_corr;
  }
  Ptr win512(const std::vector<double>& win) {
    std::vector<float> wf(512, 0.f);
    for (size_t j = 0; j < win.size(); ++j) wf[j] = (float)win[j];
    return cst(wf.data(), 2048);
  }
  // iSTFT synthesis est -> frames as an inverse FFT (istft_fft_kernel); false: plan the synthesis GEMM instead
  bool istft_fft(std::vector<Op>& ops, int tag, Ptr est, Ptr frames, int64_t nframes, int NFFT, const std::vector<double>& win) {
    if (NFFT != 512 || (int)win.size() > 512 || tune_str("STFT_GEMM")) return false;
    if (fft_tw.arena < 0) return false;                      // the STFT helper creates the twiddle table first
    Op& op = push(ops, OP_ISTFT_FFT, tag);
    op.ifft.est = est; op.ifft.frames = frames; op.ifft.tw = fft_tw; op.ifft.win = win512(win); op.ifft.corr = istft_corr((int)win.size());
    op.ifft.nframes = nframes; op.ifft.W = (int)win.size();
    return true;
  }
  // its backward: d est = Kinv . (frames of the padded waveform gradient) = the analysis transform with the same correction
  bool istft_bwd_fft(std::vector<Op>& ops, int tag, Ptr dpad, Ptr dest, int B, int Lp, int T, int hop, int NFFT, const std::vector<double>& win) {
    if (NFFT != 512 || (int)win.size() > 512 || tune_str("STFT_GEMM") || fft_tw.arena < 0) return false;
    Op& op = push(ops, OP_STFT_FFT, tag);
    op.fft.src = dpad; op.fft.spec = dest; op.fft.tw = fft_tw; op.fft.win = win512(win);
    op.fft.B = B; op.fft.L = Lp; op.fft.T = T; op.fft.hop = hop; op.fft.off = 0; op.fft.lp_dt = 0;
    op.fft.corr = istft_corr((int)win.size()); op.fft.scale = 1.f / 256.f; op.fft.lp = none(); op.fft.pair = fft_pair();
    return true;
  }

  // WGRAD for the layer whose forward descriptor is `f` (same A runs + a ones run) against upstream gradient `dy`.
  void wgrad(std::vector<Op>& ops, const RunGemm& f, Ptr dy, const Coef& coef, int tag,
             const std::function<void(int n, int32_t out[2])>* bias) {
    RunGemm g = f;            // operands keep the forward dtype: fp32 -> 32x32x2 fp32 MFMA, bf16 -> transposing 16x16x32 bf16 MFMA
    if (bias && g.nseg < kMaxSeg) {
      Seg& o = g.seg[g.nseg++];
      o.src = -1; o.dt = 0; o.off = 0; o.len = 1; o.koff = 0;
    }
    layout_segs(g);
    g.y = dy;
    g.bias = none();
    g.stats = none();
    // Row splits: every workgroup of a WGRAD launch does the same amount of work, so the grid is sized to fill the
    // co-resident slots of the 256 CUs in ONE wave of workgroups and never spill a few stragglers into a second one
    // (sized for the 4-stage ring: 64 KiB (128-wide n tile) or 48 KiB (64-wide) of LDS, 2 or 3 workgroups per CU; the shipped
    //  3-stage ring needs 48 / 36 KiB, so the same grids still fit in one wave with room for the other stream's kernels).
    const bool narrow = runs_aligned(g, true);       // thin layers: 32 / 16 wide n tiles (aligned bf16 kernel only)
    int tn = narrow ? wgrad_tn(g.xdt, g.N, g.Npad) : (g.xdt == DT_BF16 && g.Npad >= 128) ? 128 : kWgTN;
    // the layers that carry the FLOPs: 256 x 256 tile of the 8-wave kernel.  SEFD_WG256=0 keeps the 128 x 128 tile; SEFD_WG256_MINM
    // lowers the row threshold (tests run the wide kernel on small cases)
    const bool wide_on = !(tune_str("WG256") && atoi(tune_str("WG256")) == 0);
    const int64_t wide_minm = tune_str("WG256_MINM") ? atoll(tune_str("WG256_MINM")) : 32768;
    int tk = kWgTK;
    if (narrow && wide_on && g.Npad % 256 == 0 && g.ldw >= 384 && g.M >= wide_minm) { tn = 256; tk = 256; g.flags |= kRunWgWide; }
    else if (narrow && wide_on && g.Npad == 128 && g.ldw >= 1024 && g.M >= wide_minm) { tn = 128; tk = 512; g.flags |= kRunWgWide; }
    // a bias ones run that would open a k tile of its own (the data columns fill whole 256-wide tiles): the kernel forms the bias with a constant ones
    // operand in the workgroups of k tile 0 instead (kRunOnesMfma, rungemm.hip); ONES_MFMA=0 keeps the run a DMA'd column
    if ((g.flags & kRunWgWide) && tn == 256 && bias && g.nseg >= 2 && g.seg[g.nseg - 1].src < 0 && g.seg[g.nseg - 1].koff == g.ldw - 64 &&
        (g.ldw - 64) % 256 == 0 && !(tune_str("ONES_MFMA") && atoi(tune_str("ONES_MFMA")) == 0))
      g.flags |= kRunOnesMfma;
    const int64_t ldk = (g.flags & kRunOnesMfma) ? g.ldw - 64 : g.ldw;    // columns the k tiles cover
    // wg_rounds > 1 (FullSubNet): that many dispatch rounds of shorter workgroups - the launch shares the chip with a recurrence whose
    // second round leaves 2/3 of the CUs idle, and a workgroup that needs the whole kernel's duration on its CU cannot use such a hole
    // SEFD_WG_ROUNDS / SEFD_WGW_ROUNDS (tuning): rounds of every weight-gradient GEMM / of the wide-tile ones when the model did not set its own
    const int env_rounds = (g.flags & kRunWgWide) && tune_str("WGW_ROUNDS") ? atoi(tune_str("WGW_ROUNDS")) : tune_str("WG_ROUNDS") ? atoi(tune_str("WG_ROUNDS")) : 1;
    // Wide-tile launches of SHORT workgroups (at most 10 tiles, fewer than 8192 rows per workgroup at 256 slots) fill 224 CUs, not 256: beside them the
    // main stream's 160 KB-LDS GEMMs need whole CUs, and 220 instead of 250 workgroups leave every XCD four - DCCRN default 10.60 -> 10.50 ms per step
    // (slots 160 / 192 / 208 / 216 / 224 / 232 / 240 / 248: 10.59 / 10.55 / 10.53 / 10.50 / 10.50 / 10.61 / 10.61 / 10.59, profiles/r05_tuning_notes.md);
    // DCCRN-large's launches (20 / 40 tiles, or 5 tiles of 19 000-row workgroups) LOSE 0.3-0.7 ms that way and keep 256.  SEFD_WGW_SLOTS overrides.
    const int tiles_w = std::max(1, (int)(rup(std::min(g.N, g.Npad), tn) / tn * rup(ldk, tk) / tk));
    const bool short_wg = wg_rounds <= 1 && tiles_w <= 10 && (int64_t)g.M * tiles_w < (int64_t)8192 * 256;
    const int wide_slots = tune_str("WGW_SLOTS") ? atoi(tune_str("WGW_SLOTS")) : (short_wg ? 224 : 256);
    const int nscale = tune_str("WGN_SCALE") ? atoi(tune_str("WGN_SCALE")) : 100;      // tuning: percent of the slots of the narrow-tile launches
    // the first encoder layer's kernel on the spectrum (enc0.hip; 21 KB of LDS and <= 124 registers: up to 4 workgroups per CU).  Row splits
    // 512 / 768 / 1024 / 2048: 10.267 / 10.282 / 10.285 / 10.315 ms per step (three alternating runs each): the launch is not grid-bound, fewer partials fold faster
    const int enc0_slots = tune_str("ENC0_WG_SLOTS") ? atoi(tune_str("ENC0_WG_SLOTS")) : 512;
    const int slots = ((g.flags & kRunEnc0) ? enc0_slots : g.xdt == DT_BF16 ? ((g.flags & kRunWgWide) ? wide_slots : (tn == 128 ? 512 : tn == 64 ? 768 : 1024) * nscale / 100) : 768) * std::max(1, wg_rounds > 1 ? wg_rounds : env_rounds);
    const int tiles = (int)(rup(std::min(g.N, g.Npad), tn) / tn * rup(ldk, tk) / tk);   // tiles that hold real rows
    const int steps = (int)((g.M + kWgRows - 1) / kWgRows);
    int ns = std::max(1, slots / tiles);
    ns = std::max(1, std::min(ns, std::max(1, steps / 4)));
    // N <= 4 outputs over a contiguous array (FullSubNet's sub-band head): the streaming kernel, one workgroup per row split (WGRANK=0: the tiled kernels;
    // WGRANK_MINM: fewest rows, tests lower it)
    if (wgrad_rank_form(g) && g.M >= (tune_str("WGRANK_MINM") ? atoll(tune_str("WGRANK_MINM")) : 65536) && !(tune_str("WGRANK") && atoi(tune_str("WGRANK")) == 0)) {
      g.flags |= kRunRank;
      ns = std::max(1, std::min(1024, steps / 4));
    }
    g.nsplit = ns;
    const int64_t sz = (int64_t)g.Npad * g.ldw;
    const int64_t rel = gp_off;
    gp_off += sz * ns;
    Op& op = push(ops, OP_WGRAD, tag);
    op.g = g;
    fixes.push_back(Fix{(int)ops.size() - 1, rel, 0});
    if (ns > 1) split_sum(ops, rel, sz, ns, tag);
    // inverse table
    for (int n = 0; n < g.N; ++n) {
      for (int s = 0; s < g.nseg; ++s) {
        if (g.seg[s].src >= 0) {
          for (int j = 0; j < g.seg[s].len; ++j) {
            const int32_t t = coef(n, s, j);
            if (t == 0) continue;
            const int64_t pos = rel + (int64_t)n * g.ldw + g.seg[s].koff + j + 1;
            assert(pos < (1LL << 31));
            inv[std::abs(t) - 1].push_back((int32_t)(t > 0 ? pos : -pos));
          }
        } else if (bias) {
          int32_t bt[2] = {0, 0};
          (*bias)(n, bt);
          const int64_t pos = rel + (int64_t)n * g.ldw + g.seg[s].koff + 1;
          for (int e = 0; e < 2; ++e)
            if (bt[e] != 0) inv[std::abs(bt[e]) - 1].push_back((int32_t)(bt[e] > 0 ? pos : -pos));
        }
      }
    }
  }

  // Split sums: nothing reads a weight gradient's partial sums before the UNPACK that gathers them, so the folds of ALL weight gradients
  // planned since the last UNPACK wait in `pending_sums` and become ONE table-driven SPLITSUM launch in front of it (43 launches of
  // 6-20 us on the weight-gradient lane before: 0.37 ms per step).  SEFD_SPLITSUM_MULTI=0 plans one SPLITSUM behind every WGRAD again.
  struct SumSeg { int64_t rel, n, ns; };
  std::vector<SumSeg> pending_sums;
  void split_sum(std::vector<Op>& ops, int64_t rel, int64_t n, int64_t ns, int tag) {
    const bool multi = !(tune_str("SPLITSUM_MULTI") && atoi(tune_str("SPLITSUM_MULTI")) == 0);
    if (multi) { pending_sums.push_back(SumSeg{rel, n, ns}); return; }
    Op& os = push(ops, OP_SPLITSUM, tag);
    os.unpack.n = n;
    os.unpack.sstride = n;
    os.unpack.nsplit = (int32_t)ns;
    os.unpack.start = os.unpack.ent = os.unpack.dst = none();
    fixes.push_back(Fix{(int)ops.size() - 1, rel, 1});
  }
  // side = true: the launch rides the weight-gradient lane behind the WGRADs it folds (a pure HBM stream next to the other lane's GEMMs);
  // false: main stream, which first waits for the weight-gradient lane (the fold in front of an UNPACK)
  void flush_sums(std::vector<Op>& ops, int tag, bool side = false, bool join = true) {
    if (pending_sums.empty()) return;
    std::vector<int64_t> tab;
    int64_t nmax = 0;
    for (const SumSeg& sg : pending_sums) { tab.push_back(sg.rel); tab.push_back(sg.n); tab.push_back(sg.ns); nmax = std::max(nmax, sg.n); }
    const int save = cur_lane;
    cur_lane = side ? 1 : 0;
    Op& os = push(ops, OP_SPLITSUM, tag);
    cur_lane = save;
    os.join = (side || !join) ? 0 : 1;                     // the partial sums come from the weight-gradient lane (join = false: from the main stream)
    os.unpack.start = cst(tab.data(), (int64_t)tab.size() * 8);     // int64 [nseg][3]: offset from the partial-sum base, elements, splits
    os.unpack.ent = os.unpack.dst = none();
    os.unpack.n = nmax;
    os.unpack.sstride = 0;
    os.unpack.nsplit = 0;
    os.unpack.nseg = (int32_t)pending_sums.size();
    fixes.push_back(Fix{(int)ops.size() - 1, 0, 1});
    pending_sums.clear();
  }

  int64_t unpack_lo = 0;                                   // flat gradient elements below this are already unpacked (FullSubNet's first bucket)
  int64_t unpack_hi = -1;                                  // (set by unpack_range: elements [unpack_hi, end) are already done)
  // UNPACK of the flat gradient elements [lo, hi): every weight gradient GEMM that contributes to them must have been planned.
  // The partial-sum base is not known yet (finish_unpack allocates it): recorded as a fix-up.
  void unpack_range(std::vector<Op>& ops, int64_t lo, int64_t hi, int tag, bool nojoin = false) {
    flush_sums(ops, tag, false, !nojoin);
    const int64_t n = hi - lo;
    std::vector<int32_t> start(n + 1, 0), ent;
    for (int64_t j = 0; j < n; ++j) {
      start[j] = (int32_t)ent.size();
      for (auto e : inv[lo + j]) ent.push_back(e);
      if (inv[lo + j].empty() && (size_t)(lo + j) < zero_grad.size() && zero_grad[lo + j]) ent.push_back(0);   // entry 0 adds nothing: writes 0
    }
    start[n] = (int32_t)ent.size();
    if (ent.empty()) ent.push_back(0);
    Op& op = push(ops, OP_UNPACK, tag);
    op.unpack.start = cst(start.data(), (int64_t)start.size() * 4);
    op.unpack.ent = cst(ent.data(), (int64_t)ent.size() * 4);
    op.unpack.part = none();
    op.unpack.dst = mk(A_GRAD, lo * 4);
    op.unpack.n = n;
    op.unpack.sstride = 0;
    op.unpack.nsplit = 1;
    if (nojoin) op.join = kOpNoJoin;
    fixes.push_back(Fix{(int)ops.size() - 1, 0, 1});
  }
  void finish_unpack(std::vector<Op>& ops) {
    const int64_t n = unpack_hi >= 0 ? unpack_hi : (int64_t)inv.size();
    unpack_range(ops, unpack_lo, n, 999);
    Ptr base = ws("gradpart", std::max<int64_t>(gp_off, 1), DT_F32);
    for (auto& f : fixes) {
      Ptr p = mk(A_WS, base.off + f.rel * 4);
      if (f.which == 0) ops[f.op].g.w = p; else if (f.which == 1) ops[f.op].unpack.part = p; else ops[f.op].mask.colsum = p;
    }
  }
};

// ConvSTFT / ConviSTFT window sample j (tools_for_model.py:17-20)
static double window_value(const ModelConfig& cfg, int j, int W) {
  const double kPi_ = 3.14159265358979323846;
  if (cfg.window == 1) return 1.0;
  if (cfg.window == 2 && cfg.window_values) return cfg.window_values[j];
  return 0.5 - 0.5 * std::cos(2.0 * kPi_ * j / W);
}

int32_t pe(const ParamInfo& p, int64_t idx, int sign = 1) { return (int32_t)(sign * (p.off + idx + 1)); }


// Post-pass over a finished plan: give every RUNGEMM the zero page and mark the ones whose runs are whole, 16-byte aligned
// chunks (then the kernel uses the LDS-DMA loader).  Arena buffers are 256-byte aligned, so only element offsets matter.
void finalize_rungemms(Builder& b, Plan* P) {
  // page 0: 256 zero bytes ; page 1: bf16 (1, 0, 0, ...) for the bias "ones" run of the LDS-DMA WGRAD
  char pages[512] = {0};
  pages[256] = (char)0x80; pages[257] = (char)0x3f;
  Ptr z = b.cst(pages, sizeof(pages));
  for (auto* ops : {&P->fwd, &P->bwd})
    for (Op& op : *ops) {
      if (op.kind != OP_RUNGEMM && op.kind != OP_WGRAD) continue;
      RunGemm& g = op.g;
      g.zero = z;
      fastdiv_make((uint32_t)(g.Tout * g.Fo), &g.div_tf_m, &g.div_tf_s);
      fastdiv_make((uint32_t)g.Fo, &g.div_fo_m, &g.div_fo_s);
      const bool ok = runs_aligned(g, op.kind == OP_WGRAD);
      g.flags = (g.flags & ~kRunAligned) | (ok ? kRunAligned : 0);
      if (op.kind == OP_RUNGEMM) {
        const bool ya = g.ydt == DT_BF16 && g.N % 8 == 0 && g.y_off % 8 == 0 && g.y_fstride % 8 == 0 && g.y_tstride % 8 == 0 &&
                        g.y_bstride % 8 == 0 && (g.y.off % 16) == 0;
        g.flags = (g.flags & ~kRunYAligned) | (ya ? kRunYAligned : 0);
      }
    }
  // Wide-tile kernel (cgemm256.hip) for the bf16 layers that carry the FLOPs: N a multiple of 256, LDS-DMA-able runs, enough rows.
  // Its weights are packed K-tile major (kRunWTile32): a property of the packed BUFFER, so it is chosen only when every GEMM
  // that reads the buffer qualifies, and the PACK table of the matrix is permuted here, once.  SEFD_CG256=0: 128 x 128 kernel
  // everywhere (A/B runs).
  {
    const bool wide = !(tune_str("CG256") && atoi(tune_str("CG256")) == 0);
    const int wide_minm = tune_str("CG256_MINM") ? atoi(tune_str("CG256_MINM")) : 4096;
    // ... and enough 256 x 256 tiles to occupy the chip: the projection's input gradient (M = B*T = 15 456, N = 256: 61 tiles on 256 CUs) ran
    // 65 us on the wide kernel; as 242 workgroups of the 128-row kernel it fills the chip: 33 us.  Only up to K = 1024: DCCRN-large's few-tile
    // GEMMs have K = 2048 and lost 0.75 ms per step on the 128-row kernel.  (Tests that lower MINM run small cases on purpose.)
    const int wide_mintiles = tune_str("CG256_MINTILES") ? atoi(tune_str("CG256_MINTILES")) : (tune_str("CG256_MINM") ? 0 : 100);
    std::map<int64_t, bool> elig;                            // weight buffer offset -> every reader (either phase) qualifies
    std::vector<Op*> all;
    for (auto* ops : {&P->fwd, &P->bwd})
      for (Op& op : *ops) all.push_back(&op);
    for (Op* op : all) {
      if (op->kind != OP_RUNGEMM || op->g.w.arena != A_WS) continue;
      const RunGemm& g = op->g;
      const bool e = wide && (g.flags & kRunAligned) && g.xdt == DT_BF16 && g.Npad % 256 == 0 && g.ldw % 64 == 0 && g.M >= wide_minm && g.n2 == 0 &&
                     ((((g.M + 255) / 256) * (int64_t)(g.Npad / 256) >= wide_mintiles && (g.ldw >= 256 || tune_str("CG256_MINM"))) || g.ldw > 1024);
      // (... and at least four 64-deep K tiles: the layer-1 LSTM input GEMMs of the chunked forward - M 7 744, N 1024, K 128, 124 tiles - are all prologue and
      //  epilogue on the persistent 256 x 256 tile: 32 us each against 18 us on the 128-row kernel, round 6; the tests that lower MINM run small cases on purpose)
      auto it = elig.find(g.w.off);
      if (it == elig.end()) elig[g.w.off] = e; else it->second = it->second && e;
    }
    for (auto& kv : elig) {
      if (!kv.second) continue;
      const RunGemm* g = nullptr;
      for (Op* op : all) if (op->kind == OP_RUNGEMM && op->g.w.arena == A_WS && op->g.w.off == kv.first) { g = &op->g; break; }
      bool done = false;
      for (Op* po : all) {
        if (po->kind != OP_PACK || po->pack.width != 1 || po->pack.dst.arena != A_WS || po->pack.dst.off != kv.first) continue;
        if (!g || po->pack.n != (int64_t)g->Npad * g->ldw) break;
        int32_t* tab = reinterpret_cast<int32_t*>(P->consts.data() + po->pack.tab.off);
        std::vector<int32_t> old(tab, tab + po->pack.n);
        for (int n = 0; n < g->Npad; ++n)
          for (int k = 0; k < g->ldw; ++k) tab[w_index(kRunWTile32, g->ldw, g->Npad, n, k)] = old[(size_t)n * g->ldw + k];
        done = true;
        break;
      }
      if (done)
        for (Op* op : all) if (op->kind == OP_RUNGEMM && op->g.w.arena == A_WS && op->g.w.off == kv.first) op->g.flags |= kRunWTile32;
    }
  }
  if (tune_str("DUMP_GEMMS")) {                          // planner debugging: every GEMM descriptor of the plan on stderr
    int ph = 0;
    for (auto* ops : {&P->fwd, &P->bwd}) {
      int i = 0;
      for (Op& op : *ops) {
        if (op.kind == OP_RUNGEMM || op.kind == OP_WGRAD) {
          const RunGemm& g = op.g;
          fprintf(stderr, "%s ph%d op%d tag%d M=%d Tout=%d Fo=%d N=%d Npad=%d ldw=%d flags=%d xdt=%d ydt=%d n2=%d nsplit=%d\n", op.kind == OP_RUNGEMM ? "RUNGEMM" : "WGRAD", ph, i,
                  op.tag, g.M, g.Tout, g.Fo, g.N, g.Npad, g.ldw, g.flags, g.xdt, g.ydt, g.n2, g.nsplit);
          for (int s = 0; s < 2; ++s)
            if (g.x[s].arena >= 0) fprintf(stderr, "    src%d bstride=%lld tstride=%d base=%d rowlen=%d fstride=%d Tin=%d\n", s, (long long)g.bstride[s], g.tstride[s], g.base[s], g.rowlen[s], g.fstride[s], g.Tin[s]);
          for (int s = 0; s < g.nseg; ++s) fprintf(stderr, "    seg%d src=%d dt=%d off=%d len=%d koff=%d\n", s, g.seg[s].src, g.seg[s].dt, g.seg[s].off, g.seg[s].len, g.seg[s].koff);
        }
        ++i;
      }
      ++ph;
    }
  }
  // weight repacking: one launch per phase instead of one per matrix (71 launches of ~5 us in a DCCRN step)
  for (auto* ops : {&P->fwd, &P->bwd}) {
    std::vector<Pack> packs;
    std::vector<Op> rest;
    // The first encoder layer on the spectrum (kRunEnc0) keeps its own two tiny PACK launches (2 048 + 32 elements) on the main stream: as part of the
    // phase's PACKMULTI (48 us on the second stream, joined by the first GEMM) it held the first layer back until 70 us into the step although the STFT
    // in front of it ends at 17 us (profiles/r06_timeline.txt); the join moves to the second layer's GEMM
    int64_t own_w = -1, own_b = -1;
    for (Op& op : *ops)
      if (op.kind == OP_RUNGEMM && (op.g.flags & kRunEnc0)) { own_w = op.g.w.off; own_b = op.g.bias.arena >= 0 ? op.g.bias.off : -1; break; }
    for (Op& op : *ops) {
      const bool own = op.kind == OP_PACK && op.pack.dst.arena == A_WS && (op.pack.dst.off == own_w || op.pack.dst.off == own_b);
      if (op.kind == OP_PACK && !own) packs.push_back(op.pack); else rest.push_back(op);
    }
    if (packs.size() < 2) continue;
    Op m;
    std::memset(&m, 0, sizeof(m));
    m.kind = OP_PACKMULTI;
    m.tag = 1;
    m.packm.entries = b.cst(packs.data(), (int64_t)packs.size() * sizeof(Pack));
    m.packm.count = (int32_t)packs.size();
    rest.insert(rest.begin(), m);
    ops->swap(rest);
  }
  // Training plans: both phases' packs ride the second stream during the forward phase (they read only parameters; the backward's 57 us
  // pass sat in the serial loss section, the forward's in front of the STFT).  The forward packs are issued first and joined by the first
  // op that reads a packed matrix; the backward packs are issued right after that op (one launch of both slowed the STFT / spectrum
  // kernels next to it and delayed the first GEMM by 55 us).  SEFD_PACK_EARLY=0 keeps one launch per phase at its head.
  if (!(tune_str("PACK_EARLY") && atoi(tune_str("PACK_EARLY")) == 0) && !P->fwd.empty() && !P->bwd.empty() &&
      P->fwd[0].kind == OP_PACKMULTI && P->bwd[0].kind == OP_PACKMULTI) {
    P->fwd[0].lane = 2;
    size_t first = 0;
    for (size_t i = 1; i < P->fwd.size(); ++i) {
      Op& op = P->fwd[i];
      if (op.lane != 0) continue;
      if (op.kind == OP_RUNGEMM && (op.g.flags & kRunEnc0)) continue;                     // reads its own PACK launches, not the PACKMULTI's output
      if (op.kind == OP_RUNGEMM || op.kind == OP_LSTM_FWD || op.kind == OP_WGRAD) { op.join = 1; first = i; break; }
    }
    if (first > 0) {
      Op m = P->bwd[0];
      m.lane = 2;
      P->bwd.erase(P->bwd.begin());
      P->fwd.insert(P->fwd.begin() + first + 1, m);
    } else {
      P->fwd[0].lane = 0;
    }
  }
  // SyncBN (cfg.bn_world > 1): every training-mode BN_FINALIZE becomes "publish this rank's sums" + "statistics from the
  // all-reduced sums" with a sync point in between; every BN_BWD_FINALIZE is followed by a sync point on its totals.
  // Counts become global.  The caller (models.py / hostsim tests) runs the op ranges between sync points and all-reduces.
  const int world = P->cfg.bn_world;
  if (world > 1 && P->cfg.training) {
    int k = 0;
    for (int phase = 0; phase < 2; ++phase) {
      std::vector<Op>& ops = phase == 0 ? P->fwd : P->bwd;
      std::vector<Op> out;
      for (Op op : ops) {
        if (op.kind == OP_BN_FINALIZE && op.bnf.nblk >= 0) {
          op.bnf.totals = b.ws("syncbn.tot" + std::to_string(k++), (int64_t)2 * op.bnf.C * 2, DT_F32);   // 2*C doubles
          op.bnf.mode = 1;
          out.push_back(op);
          P->syncs.push_back(SyncPoint{phase, (int32_t)out.size() - 1, op.bnf.totals, 2 * (int64_t)op.bnf.C, 1});
          op.bnf.mode = 2;
          op.bnf.count *= world;
          out.push_back(op);
        } else if (op.kind == OP_BN_BWD_FINALIZE) {
          out.push_back(op);
          P->syncs.push_back(SyncPoint{phase, (int32_t)out.size() - 1, op.bnb.totals, 2 * (int64_t)op.bnb.r.C, 0});
        } else if (op.kind == OP_BN_BWD_APPLY) {
          op.bnb.count *= world;
          out.push_back(op);
        } else {
          out.push_back(op);
        }
      }
      ops.swap(out);
    }
  }
}

}  // namespace

// =================================================================================================================
Plan* build_dccrn_plan(const ModelConfig& cfg) {
  Plan* P = new Plan();
  P->cfg = cfg;
  Builder b;
  b.P = P;
  b.c = cfg;
  const int n = cfg.n_layers;
  const int B = cfg.B, L = cfg.L, W = cfg.win_len, hop = cfg.hop, NFFT = cfg.fft_len;
  const int trim = W - hop;
  const int T = (L + 2 * trim - W) / hop + 1;
  const int NF = NFFT / 2 + 1, NS = NF + 1, SW = NS * 2;
  const int Lp = (T - 1) * hop + W;
  const int adt = cfg.act_dtype;
  const int KS = cfg.kernel_size;
  P->T = T;
  P->NF = NF;
  if (cfg.model != 0 || KS != 5 || n < 1 || n > 7) { P->error = "unsupported configuration"; return P; }
  const bool cx = cfg.lstm_complex != 0;     // cfg.lstm: 'complex' (NavieComplexLSTM stack) or 'real' (nn.LSTM(2 layers) + tranform, models.py:96-105)
  std::vector<int> ch(n + 1), Fe(n + 1);
  ch[0] = 2;
  for (int i = 0; i < n; ++i) ch[i + 1] = cfg.kernel_num[i];
  Fe[0] = NF - 1;
  for (int i = 0; i < n; ++i) Fe[i + 1] = Fe[i] / 2;
  const int D = Fe[n];                       // hidden_dim (models.py:81)
  const int H = cfg.lstm_complex ? cfg.rnn_units / 2 : cfg.rnn_units;   // per-part hidden size of the complex LSTM / hidden size of the real one
  const int NL = cfg.rnn_layers;
  const int Cl = ch[n];                      // channels entering the LSTM
  for (int i = 1; i <= n; ++i)
    if (ch[i] % 8 != 0 && !(i == 0)) { P->error = "channel counts must be multiples of 8"; return P; }
  // H <= 128: persistent recurrence kernels (W_hh resident in the VGPRs of one CU).  Larger H (DCCRN-large: rnn_units 512):
  // bf16 mode runs the cluster kernels of lstm_cluster.hip (W_hh spread over H/64 CUs, h handed over in memory every step);
  // fp32 mode and odd sizes fall back to one GEMM + one cell launch per time step on the same buffers.
  const bool cluster_ok = adt == DT_BF16 && H > 128 && H <= 512 && H % 64 == 0;
  const bool stepped = (H > 128 && !cluster_ok) || tune_str("LSTM_STEPPED") != nullptr;
  // all weight gradients ride the second stream (after the fork they run next to the encoder's dgrad / BatchNorm chain and
  // fill the tails of its kernels: 14.42 -> 14.30 ms/step); SEFD_LANE_ALL=0 keeps only the decoder's there
  const bool lane_all = !(tune_str("LANE_ALL") != nullptr && atoi(tune_str("LANE_ALL")) == 0);
  if (H % 16 != 0 || (adt == DT_BF16 && H % 32 != 0)) { P->error = "rnn_units/2 must be a multiple of 16 (32 for bf16)"; return P; }
  if (adt == DT_BF16 && H % 32 != 0) { P->error = "bf16: rnn_units/2 must be a multiple of 32"; return P; }
  if (Fe[n] < 1 || (Fe[0] % (1 << n)) != 0) { P->error = "fft_len/2 must be divisible by 2^n_layers"; return P; }

  // ------------------------------------------------------------------ parameters (reference registration order)
  const bool cbn = cfg.use_cbn != 0;
  if (cbn && cfg.bn_world > 1) { P->error = "ComplexBatchNorm has no SyncBN plan"; return P; }
  // the normalisation + PReLU behind a conv: nn.BatchNorm2d(C) or ComplexBatchNorm(C) (tools_for_model.py:441-467: 5 parameters and 5 buffers of C / 2)
  auto add_norm = [&](const std::string& p, int C) {
    if (cbn) {
      for (const char* w : {"Wrr", "Wri", "Wii", "Br", "Bi"}) b.add_param(p + ".1." + w, {C / 2}, true);
      for (const char* w : {"RMr", "RMi", "RVrr", "RVri", "RVii"}) b.add_param(p + ".1." + w, {C / 2}, false);
    } else {
      b.add_param(p + ".1.weight", {C}, true);
      b.add_param(p + ".1.bias", {C}, true);
      b.add_param(p + ".1.running_mean", {C}, false);
      b.add_param(p + ".1.running_var", {C}, false);
    }
    b.add_param(p + ".2.weight", {1}, true);
  };
  // forward of that layer: y [Rr][C] -> z
  auto cbn_fwd = [&](int tag, const std::string& pp, const std::string& nm, Ptr y, Ptr z, int C, int64_t Rr) -> Ptr {
    if ((C / 2) % 4 != 0) { P->error = "ComplexBatchNorm: channel pairs per layer must be a multiple of 4"; return b.none(); }
    if (C / 2 > 1024) { P->error = "ComplexBatchNorm: at most 1024 channel pairs per layer (cbn.hip reduces a row of pairs in one workgroup)"; return b.none(); }
    CbnFwd c;
    std::memset(&c, 0, sizeof(c));
    const int h = C / 2;
    const int64_t rpbk = std::max<int64_t>(64, (Rr + 2047) / 2048);
    c.y = y; c.z = z; c.R = Rr; c.C = C; c.dt = adt; c.nblk = (int)((Rr + rpbk - 1) / rpbk); c.rows_per_blk = (int)rpbk;
    c.training = cfg.training; c.count = (double)Rr; c.eps = 1e-5f; c.momentum = 0.1f;
    c.part = cfg.training ? b.ws(nm + ".cstat", (int64_t)c.nblk * 5 * h, DT_F32) : b.none();
    c.coef = b.ws(nm + ".ccoef", 14 * h, DT_F32);
    const char* wn[3] = {"Wrr", "Wri", "Wii"};
    const char* rvn[3] = {"RVrr", "RVri", "RVii"};
    for (int q = 0; q < 3; ++q) { c.W[q] = b.pptr(pp + ".1." + wn[q]); c.RV[q] = b.sptr(pp + ".1." + rvn[q]); }
    c.Bv[0] = b.pptr(pp + ".1.Br"); c.Bv[1] = b.pptr(pp + ".1.Bi");
    c.RM[0] = b.sptr(pp + ".1.RMr"); c.RM[1] = b.sptr(pp + ".1.RMi");
    c.slope = b.pptr(pp + ".2.weight");
    if (cfg.training) b.push(P->fwd, OP_CBN_STATS, tag).cbf = c;
    b.push(P->fwd, OP_CBN_FINALIZE, tag).cbf = c;
    b.push(P->fwd, OP_CBN_APPLY, tag).cbf = c;
    return c.coef;
  };
  for (int i = 0; i < n; ++i) {
    const std::string p = "encoder." + std::to_string(i);
    for (const char* part : {"real_conv", "imag_conv"}) {
      b.add_param(p + ".0." + part + ".weight", {ch[i + 1] / 2, ch[i] / 2, KS, 2}, true);
      b.add_param(p + ".0." + part + ".bias", {ch[i + 1] / 2}, true);
    }
    add_norm(p, ch[i + 1]);
  }
  for (int d = 0; d < n; ++d) {
    const int idx = n - d;
    const int cin = ch[idx] * (cfg.skip ? 2 : 1), cout = ch[idx - 1];
    const std::string p = "decoder." + std::to_string(d);
    for (const char* part : {"real_conv", "imag_conv"}) {
      b.add_param(p + ".0." + part + ".weight", {cin / 2, cout / 2, KS, 2}, true);
      b.add_param(p + ".0." + part + ".bias", {cout / 2}, true);
    }
    if (idx != 1) add_norm(p, cout);
  }
  const int hid = D * Cl;                    // LSTM feature size real+imag
  if (!cx) {                                  // nn.LSTM(hid, rnn_units, num_layers=2) then nn.Linear(rnn_units, hid)
    for (int l = 0; l < 2; ++l) {
      const std::string sl = std::to_string(l);
      b.add_param("enhance.weight_ih_l" + sl, {4 * H, l == 0 ? hid : H}, true);
      b.add_param("enhance.weight_hh_l" + sl, {4 * H, H}, true);
      b.add_param("enhance.bias_ih_l" + sl, {4 * H}, true);
      b.add_param("enhance.bias_hh_l" + sl, {4 * H}, true);
    }
    b.add_param("tranform.weight", {hid, H}, true);
    b.add_param("tranform.bias", {hid}, true);
  }
  for (int l = 0; l < (cx ? NL : 0); ++l) {
    const int I = (l == 0 ? hid : cfg.rnn_units) / 2;
    const std::string p = "enhance." + std::to_string(l);
    for (const char* part : {"real_lstm", "imag_lstm"}) {
      b.add_param(p + "." + part + ".weight_ih_l0", {4 * H, I}, true);
      b.add_param(p + "." + part + ".weight_hh_l0", {4 * H, H}, true);
      b.add_param(p + "." + part + ".bias_ih_l0", {4 * H}, true);
      b.add_param(p + "." + part + ".bias_hh_l0", {4 * H}, true);
    }
    if (l == NL - 1)
      for (const char* part : {"r_trans", "i_trans"}) {
        b.add_param(p + "." + part + ".weight", {hid / 2, H}, true);
        b.add_param(p + "." + part + ".bias", {hid / 2}, true);
      }
  }
  const int64_t nparam = P->params.back().off + P->params.back().numel;
  const int64_t nstate = P->state.empty() ? 0 : P->state.back().off + P->state.back().numel;
  b.inv.resize(nparam);

  // ------------------------------------------------------------------ I/O block
  Ptr io_wav = b.io("wav", (int64_t)B * L);
  Ptr io_out = b.io("out_wav", (int64_t)B * L);
  Ptr io_or = b.io("out_real", (int64_t)B * NF * T);
  Ptr io_oi = b.io("out_imag", (int64_t)B * NF * T);
  Ptr io_gw = b.io("grad_wav", (int64_t)B * L);
  Ptr io_gr = b.io("grad_real", (int64_t)B * NF * T);
  Ptr io_gi = b.io("grad_imag", (int64_t)B * NF * T);

  // ------------------------------------------------------------------ constants: STFT bases, OLA normaliser
  // analysis basis (tools_for_model.py:16-33): K[part*NF+k][j] = w[j]*{cos,-sin}(2 pi k j / NFFT); periodic Hann
  std::vector<double> win(W);
  for (int j = 0; j < W; ++j) win[j] = window_value(cfg, j, W);    // win_type None: np.ones (tools_for_model.py:17-18)
  auto Kun = [&](int part, int k, int j) {
    const double ang = 2.0 * kPi * (double)(((int64_t)k * j) % NFFT) / NFFT;
    return part == 0 ? std::cos(ang) : -std::sin(ang);
  };
  // synthesis basis = pinv(K_unwindowed)^T * w, closed form: K^T K = (NFFT/2) I + E, E[n][m] = [n-m even]
  //   pinv(K)[j][r] = (K[r][j] - sum_{m == j mod 2} K[r][m] / (NFFT/2 + |{m == j mod 2}|)) / (NFFT/2)
  std::vector<double> Kinv((size_t)2 * NF * W);
  {
    const double ne = (W + 1) / 2, no = W / 2;
    for (int part = 0; part < 2; ++part)
      for (int k = 0; k < NF; ++k) {
        double se = 0, so = 0;
        for (int m = 0; m < W; ++m) (m % 2 == 0 ? se : so) += Kun(part, k, m);
        for (int j = 0; j < W; ++j) {
          const double corr = (j % 2 == 0) ? se / (NFFT / 2.0 + ne) : so / (NFFT / 2.0 + no);
          Kinv[((size_t)part * NF + k) * W + j] = (Kun(part, k, j) - corr) / (NFFT / 2.0) * win[j];
        }
      }
  }
  std::vector<float> coff(Lp, 0.f);
  {
    std::vector<float> w2(W);
    for (int j = 0; j < W; ++j) { const float wf = (float)win[j]; w2[j] = wf * wf; }
    for (int t = 0; t < T; ++t)
      for (int j = 0; j < W; ++j) coff[t * hop + j] += w2[j];
  }
  Ptr c_coff = b.cst(coff.data(), (int64_t)coff.size() * 4);

  auto const_weights = [&](RunGemm& g, const std::function<double(int n, int j)>& val) {
    std::vector<float> wt((size_t)g.Npad * g.ldw, 0.f);
    for (int nn = 0; nn < g.N; ++nn)
      for (int j = 0; j < g.seg[0].len; ++j) wt[(size_t)nn * g.ldw + j] = (float)val(nn, j);
    g.w = b.cst(wt.data(), (int64_t)wt.size() * 4);
  };

  std::vector<Op>& F = P->fwd;
  std::vector<Op>& R = P->bwd;

  // ------------------------------------------------------------------ STFT (ConvSTFT.forward, tools_for_model.py:54-61)
  Ptr spec = b.ws("spec", (int64_t)B * T * SW, DT_F32);
  Ptr spec_lp = spec;
  const bool spec_fft = b.stft_fft(F, 1, io_wav, spec, B, L, T, hop, trim, NFFT, win);
  if (!spec_fft) {
    RunGemm g = Builder::gemm0();
    g.x[0] = io_wav; g.xdt = DT_F32; g.ydt = DT_F32;
    g.bstride[0] = L; g.tstride[0] = 0; g.base[0] = 0; g.rowlen[0] = L; g.fstride[0] = hop; g.Tin[0] = 1;
    g.M = B * T; g.Tout = 1; g.Fo = T;
    g.nseg = 1; g.seg[0] = Seg{0, 0, -trim, W, 0};
    g.N = SW;
    Builder::layout_segs(g);
    const_weights(g, [&](int nn, int j) { return nn < 2 ? 0.0 : Kun(nn & 1, nn / 2 - 1, j) * win[j]; });
    g.y = spec; g.y_bstride = (int64_t)T * SW; g.y_tstride = 0; g.y_fstride = SW; g.y_off = 0;
    b.push(F, OP_RUNGEMM, 1).g = g;
  }
  // encoder input: spectrogram with the 2 channels padded to CP (aligned 16-byte runs for the thin first layer), act dtype
  const int CP = 8;
  // bf16 plans (round 6): the first layer reads the fp32 spectrum itself - no padded copy (64 MB written and read per step at B = 32), K = 20 instead of
  // 128 mostly-zero columns; kernels: enc0.hip.  ENC0_DIRECT=0: the padded copy and the generic kernels (A/B runs)
  const bool enc0_direct = adt == DT_BF16 && spec_fft && NS == 258 && KS == 5 && Fe[1] == 128 && (ch[1] == 16 || ch[1] == 32 || ch[1] == 64) &&
                           !(tune_str("ENC0_DIRECT") && atoi(tune_str("ENC0_DIRECT")) == 0);
  if (!enc0_direct) {
    spec_lp = b.ws("xin", (int64_t)B * T * NS * CP, adt);
    const bool fuse_pad = !(tune_str("SPECPAD_FUSE") && atoi(tune_str("SPECPAD_FUSE")) == 0);
    if (spec_fft && fuse_pad && NS == 258) {      // the FFT kernel writes the padded copy beside the spectrogram (no SPECPAD pass: 48 us at B = 32)
      F.back().fft.lp = spec_lp; F.back().fft.lp_dt = adt;
    } else {
      Op& op = b.push(F, OP_SPECPAD, 1);
      op.mags.spec = spec; op.mags.mags = spec_lp; op.mags.frames = (int64_t)B * T; op.mags.NF = NS; op.mags.MS = CP; op.mags.MO = 0; op.mags.dt = adt;
    }
  }

  // ------------------------------------------------------------------ encoder
  struct Layer { RunGemm f[2]; Builder::Coef coef[2]; std::function<void(int, int32_t*)> bias; bool has_bias_fn; Ptr y, z, mi; int C, Fq; int64_t R; };
  std::vector<Layer> enc(n), dec(n);
  std::vector<Ptr> encz(n), ency(n), enc_mi(n);
  Ptr prev = enc0_direct ? spec : spec_lp;
  for (int i = 0; i < n; ++i) {
    const int Ci = ch[i], Co = ch[i + 1], Fi = Fe[i], Fo = Fe[i + 1];
    const bool direct0 = i == 0 && enc0_direct;
    const int Cib = i == 0 ? (direct0 ? 2 : CP) : Ci;             // channels of the input BUFFER (first layer: padded, or the spectrum's (re, im) pairs)
    const std::string nm = "enc" + std::to_string(i);
    const std::string pp = "encoder." + std::to_string(i);
    const ParamInfo &Wr = b.par(pp + ".0.real_conv.weight"), &Wi = b.par(pp + ".0.imag_conv.weight");
    const ParamInfo &br = b.par(pp + ".0.real_conv.bias"), &bi = b.par(pp + ".0.imag_conv.bias");
    RunGemm g = Builder::gemm0();
    g.x[0] = prev;
    g.xdt = direct0 ? DT_F32 : adt;
    g.ydt = adt;
    if (direct0) g.flags |= kRunEnc0;
    if (i == 0) { g.bstride[0] = (int64_t)T * NS * Cib; g.tstride[0] = NS * Cib; g.base[0] = 2 * Cib; }
    else { g.bstride[0] = (int64_t)T * Fi * Ci; g.tstride[0] = Fi * Ci; g.base[0] = 0; }
    g.rowlen[0] = Fi * Cib; g.fstride[0] = 2 * Cib; g.Tin[0] = T;
    g.M = B * T * Fo; g.Tout = T; g.Fo = Fo;
    g.nseg = 2;
    g.seg[0] = Seg{0, -1, -2 * Cib, KS * Cib, 0};   // kw = 0 : frame t-1
    g.seg[1] = Seg{0, 0, -2 * Cib, KS * Cib, 0};    // kw = 1 : frame t
    g.N = Co;
    Builder::layout_segs(g);
    const int Ci2 = Ci / 2, Co2 = Co / 2;
    Builder::Coef coef = [=](int nn, int s, int j) -> int32_t {
      const int kw = s, kh = j / Cib, ci = j % Cib;
      if (ci >= Ci) return 0;                       // pad channel
      const bool oi = nn >= Co2, ii = ci >= Ci2;
      const int co2 = oi ? nn - Co2 : nn, ci2 = ii ? ci - Ci2 : ci;
      const int64_t idx = (((int64_t)co2 * Ci2 + ci2) * KS + kh) * 2 + kw;
      if (!oi) return ii ? pe(Wi, idx, -1) : pe(Wr, idx, 1);
      return ii ? pe(Wr, idx, 1) : pe(Wi, idx, 1);
    };
    std::function<void(int, int32_t*)> bias = [=](int nn, int32_t* o) {
      if (nn < Co2) { o[0] = pe(br, nn, 1); o[1] = pe(bi, nn, -1); }
      else { o[0] = pe(br, nn - Co2, 1); o[1] = pe(bi, nn - Co2, 1); }
    };
    b.pack_weights(F, g, coef, nm, 100 + i, &bias);
    const int64_t Rr = (int64_t)B * T * Fo;
    ency[i] = b.ws(nm + ".y", Rr * Co, adt);
    encz[i] = b.ws(nm + ".z", Rr * Co, adt);
    enc_mi[i] = b.ws(nm + ".mi", 2 * Co, DT_F32);
    const int nblk = (int)((g.M + kBM - 1) / kBM);
    Ptr part = b.ws(nm + ".stat", (int64_t)nblk * 2 * g.Npad, DT_F32);
    g.y = ency[i]; g.y_bstride = (int64_t)T * Fo * Co; g.y_tstride = Fo * Co; g.y_fstride = Co; g.y_off = 0;
    g.stats = cfg.training && !cbn ? part : b.none();
    b.push(F, OP_RUNGEMM, 100 + i).g = g;
    if (cbn) {
      enc_mi[i] = cbn_fwd(100 + i, pp, nm, ency[i], encz[i], Co, Rr);       // the layer's coefficient table takes the place of (mean, invstd)
      if (!P->error.empty()) return P;
    } else {
      Op& op = b.push(F, OP_BN_FINALIZE, 100 + i);
      op.bnf.part = part; op.bnf.mean_invstd = enc_mi[i];
      op.bnf.running_mean = b.sptr(pp + ".1.running_mean"); op.bnf.running_var = b.sptr(pp + ".1.running_var");
      op.bnf.nblk = cfg.training ? nblk : -1; op.bnf.C = Co; op.bnf.Cpad = g.Npad; op.bnf.count = (double)Rr;
      op.bnf.eps = 1e-5f; op.bnf.momentum = 0.1f;
    }
    if (!cbn) {
      Op& op = b.push(F, OP_BN_APPLY, 100 + i);
      op.bna.y = ency[i]; op.bna.z = encz[i]; op.bna.mean_invstd = enc_mi[i];
      op.bna.gamma = b.pptr(pp + ".1.weight"); op.bna.beta = b.pptr(pp + ".1.bias"); op.bna.slope = b.pptr(pp + ".2.weight");
      op.bna.R = Rr; op.bna.C = Co; op.bna.dt = adt;
    }
    enc[i].f[0] = g; enc[i].coef[0] = coef; enc[i].bias = bias; enc[i].C = Co; enc[i].Fq = Fo; enc[i].R = Rr;
    prev = encz[i];
  }

  // ------------------------------------------------------------------ complex LSTM stack (tools_for_model.py:141-181)
  const int64_t BT = (int64_t)B * T;
  struct Lstm { RunGemm gx[2]; Builder::Coef cgx[2]; std::function<void(int, int32_t*)> bgx; Ptr gxb, h, gates, cst, hc; RunGemm hh[4]; Builder::Coef chh[4]; };
  std::vector<Lstm> ls(NL);
  Ptr lin = encz[n - 1];
  // ---- cfg.lstm == 'real': two stacked real LSTM layers over all D*Cl features (feature order c*D + d, models.py:214-218)
  struct RealL { RunGemm gx, hh; Builder::Coef cgx; std::function<void(int, int32_t*)> bgx; Ptr gxb, h, gates, cst; const ParamInfo* Whh; };
  RealL rl[2];
  auto real_cell = [&](LstmCell& cl, int l, int t, bool fwd, Ptr dh, Ptr dcb, Ptr dgates) {
    cl.gates = b.mk(A_WS, rl[l].gxb.off + (int64_t)t * 4 * H * 4);
    cl.c = b.mk(A_WS, rl[l].cst.off + (int64_t)t * H * 4);
    cl.c_prev = t > 0 ? b.mk(A_WS, rl[l].cst.off + (int64_t)(t - 1) * H * 4) : b.none();
    cl.h = fwd ? b.mk(A_WS, rl[l].h.off + (int64_t)t * H * esize(adt)) : b.none();
    cl.dh = fwd ? b.none() : b.mk(A_WS, dh.off + (int64_t)t * H * 4);
    cl.dc = fwd ? b.none() : dcb;
    cl.dgates = fwd ? b.none() : b.mk(A_WS, dgates.off + (int64_t)t * 4 * H * esize(adt));
    cl.rows = B; cl.H = H; cl.hdt = adt; cl.gdt = adt; cl.first = fwd ? t == 0 : t == T - 1;
    cl.G = 1; cl.Bg = B; cl.unit_major = 1;
    cl.rs[0] = (int64_t)T * 4 * H; cl.rs[1] = cl.rs[2] = cl.rs[3] = (int64_t)T * H; cl.rs[4] = (int64_t)T * 4 * H;
  };
  if (!cx) {
    for (int l = 0; l < 2; ++l) {
      const std::string nm = "lstm" + std::to_string(l), sl = std::to_string(l);
      const ParamInfo &Wih = b.par("enhance.weight_ih_l" + sl), &Whh = b.par("enhance.weight_hh_l" + sl);
      const ParamInfo &bih = b.par("enhance.bias_ih_l" + sl), &bhh = b.par("enhance.bias_hh_l" + sl);
      RealL& Lr = rl[l];
      Lr.Whh = &Whh;
      Lr.gxb = b.ws(nm + ".gx", BT * 4 * H, DT_F32);
      Lr.h = b.ws(nm + ".h", BT * H, adt);
      Lr.gates = b.ws(nm + ".gates", BT * 4 * H, DT_F32);
      Lr.cst = b.ws(nm + ".c", BT * H, DT_F32);
      RunGemm g = Builder::gemm0();
      g.x[0] = lin; g.xdt = adt; g.ydt = DT_F32;
      const int rowlen = l == 0 ? D * Cl : H;
      g.bstride[0] = (int64_t)T * rowlen; g.tstride[0] = rowlen; g.rowlen[0] = rowlen; g.Tin[0] = T;
      g.M = (int)BT; g.Tout = T; g.Fo = 1;
      if (l == 0) { g.nseg = D; for (int dd = 0; dd < D; ++dd) g.seg[dd] = Seg{0, 0, dd * Cl, Cl, 0}; }
      else { g.nseg = 1; g.seg[0] = Seg{0, 0, 0, H, 0}; }
      g.N = 4 * H;
      Builder::layout_segs(g);
      const int I = l == 0 ? hid : H;
      const ParamInfo* Wp = &Wih;
      Lr.cgx = [=](int nn, int sg, int j) -> int32_t { return pe(*Wp, (int64_t)gate_torch_row(nn, H) * I + (l == 0 ? j * D + sg : j), 1); };
      const ParamInfo *bi = &bih, *bh = &bhh;
      Lr.bgx = [=](int nn, int32_t* o) { o[0] = pe(*bi, gate_torch_row(nn, H), 1); o[1] = pe(*bh, gate_torch_row(nn, H), 1); };
      b.pack_weights(F, g, Lr.cgx, nm + ".ih", 200 + l, &Lr.bgx);
      g.y = Lr.gxb; g.y_bstride = (int64_t)T * 4 * H; g.y_tstride = 4 * H;
      b.push(F, OP_RUNGEMM, 200 + l).g = g;
      Lr.gx = g;
      if (!stepped) {
        LstmRec& r = b.push(F, OP_LSTM_FWD, 200 + l).lstm;
        r.gx = Lr.gxb; r.whh[0] = r.whh[1] = b.pptr("enhance.weight_hh_l" + sl);
        r.h = Lr.h; r.gates = Lr.gates; r.c = Lr.cst; r.dh = r.dgates = b.none();
        r.gx_ld = 4 * H; r.G = 1; r.nset = 1; r.B = B; r.T = T; r.H = H; r.hdt = adt; r.gdt = DT_F32;
      } else {
        RunGemm hg = Builder::gemm0();
        hg.x[0] = Lr.h; hg.xdt = adt; hg.ydt = DT_F32;
        hg.fstride[0] = T * H; hg.rowlen[0] = (int)(BT * H); hg.Tin[0] = 1;
        hg.M = B; hg.Tout = 1; hg.Fo = B;
        hg.nseg = 1; hg.seg[0] = Seg{0, 0, 0, H, 0};
        hg.N = 4 * H;
        Builder::layout_segs(hg);
        const ParamInfo* Wh = &Whh;
        Builder::Coef chh = [=](int nn, int sg, int j) -> int32_t { return pe(*Wh, (int64_t)gate_torch_row(nn, H) * H + j, 1); };
        b.pack_weights(F, hg, chh, nm + ".hh", 200 + l);
        hg.y = Lr.gxb; hg.y_fstride = T * 4 * H; hg.flags = kRunAccum;
        Lr.hh = hg;
        for (int t = 0; t < T; ++t) {
          if (t > 0) {
            RunGemm q = hg;
            q.base[0] = (t - 1) * H;
            q.y_off = t * 4 * H;
            b.push(F, OP_RUNGEMM, 200 + l).g = q;
          }
          real_cell(b.push(F, OP_CELL_FWD, 200 + l).cell, l, t, true, b.none(), b.none(), b.none());
        }
      }
      lin = Lr.h;
    }
  }
  // Two complex layers, persistent bf16 kernels: the sequence is cut into chunks of frames and layer 1 (combine + input
  // GEMM + recurrence of a chunk, second HIP stream) runs while layer 0 already works on the next chunk - the two 483-step
  // recurrences (8 workgroups each, latency-bound) overlap instead of running back to back.  SEFD_LSTM_CHUNKS=1 disables.
  // Measured (B = 32, T = 483): 1 chunk 14.08 ms/step, 2-6 chunks 13.84-13.94, 8: 13.94, 16: 14.50 -> 4.
  int nchunk = tune_str("LSTM_CHUNKS") ? atoi(tune_str("LSTM_CHUNKS")) : 4;
  if (!(cx && !stepped && adt == DT_BF16 && NL == 2) || nchunk < 2 || T < 8 * nchunk) nchunk = 1;
  const bool pipe = nchunk > 1;
  LstmRec pipe_rec[2];
  RunGemm pipe_gx1[2];
  for (int l = 0; l < (cx ? NL : 0); ++l) {
    const std::string nm = "lstm" + std::to_string(l);
    const std::string pp = "enhance." + std::to_string(l);
    const ParamInfo* Wih[2] = {&b.par(pp + ".real_lstm.weight_ih_l0"), &b.par(pp + ".imag_lstm.weight_ih_l0")};
    const ParamInfo* bih[2] = {&b.par(pp + ".real_lstm.bias_ih_l0"), &b.par(pp + ".imag_lstm.bias_ih_l0")};
    const ParamInfo* bhh[2] = {&b.par(pp + ".real_lstm.bias_hh_l0"), &b.par(pp + ".imag_lstm.bias_hh_l0")};
    const int I = (l == 0 ? hid : 2 * H) / 2;      // features per part
    const int rowlen = l == 0 ? D * Cl : 2 * H;
    ls[l].gxb = b.ws(nm + ".gx", 2 * BT * 8 * H, DT_F32);
    ls[l].h = b.ws(nm + ".h", 4 * BT * H, adt);
    ls[l].gates = b.ws(nm + ".gates", 4 * BT * 4 * H, DT_F32);
    ls[l].cst = b.ws(nm + ".c", 4 * BT * H, DT_F32);
    ls[l].hc = b.ws(nm + ".hc", BT * 2 * H, adt);
    std::function<void(int, int32_t*)> bias = [=](int nn, int32_t* o) {
      const int set = nn / (4 * H), gq = gate_torch_row(nn % (4 * H), H);
      o[0] = pe(*bih[set], gq, 1); o[1] = pe(*bhh[set], gq, 1);
    };
    ls[l].bgx = bias;
    const bool gx_merge = !(tune_str("GX_MERGE") && atoi(tune_str("GX_MERGE")) == 0) && BT * 8 * H < (1LL << 31);
    for (int p = 0; p < 2; ++p) {
      RunGemm g = Builder::gemm0();
      g.x[0] = lin; g.xdt = adt; g.ydt = DT_F32;
      g.bstride[0] = (int64_t)T * rowlen; g.tstride[0] = rowlen; g.base[0] = 0; g.rowlen[0] = rowlen; g.fstride[0] = 0; g.Tin[0] = T;
      g.M = (int)BT; g.Tout = T; g.Fo = 1;
      if (l == 0) {
        g.nseg = D;
        for (int dd = 0; dd < D; ++dd) g.seg[dd] = Seg{0, 0, dd * Cl + p * (Cl / 2), Cl / 2, 0};
      } else {
        g.nseg = 1;
        g.seg[0] = Seg{0, 0, p * H, H, 0};
      }
      g.N = 8 * H;
      Builder::layout_segs(g);
      Builder::Coef coef = [=](int nn, int s, int j) -> int32_t {
        const int set = nn / (4 * H), gq = gate_torch_row(nn % (4 * H), H);
        const int feat = (l == 0) ? j * D + s : j;      // reference feature order c*D + d (models.py:203-206)
        return pe(*Wih[set], (int64_t)gq * I + feat, 1);
      };
      b.pack_weights(F, g, coef, nm + ".ih" + std::to_string(p), 200 + l, p == 0 ? &bias : nullptr);
      if (p == 1) g.bias = ls[l].gx[0].bias;
      g.y = b.mk(A_WS, ls[l].gxb.off + (int64_t)p * BT * 8 * H * 4);
      g.y_bstride = (int64_t)T * 8 * H; g.y_tstride = 8 * H; g.y_fstride = 0; g.y_off = 0;
      ls[l].gx[p] = g; ls[l].cgx[p] = coef;
      if (gx_merge) continue;
      if (pipe && l == 1) pipe_gx1[p] = g; else b.push(F, OP_RUNGEMM, 200 + l).g = g;
    }
    if (gx_merge) {
      // both parts in ONE launch: the two GEMMs share their weights (W_ih of the real and the imag LSTM side by side) and differ only in the
      // input columns (part p) and the output slab - the part becomes the row index f of the run descriptor (rows (b, t, p))
      RunGemm g = ls[l].gx[0];
      g.Fo = 2; g.M = (int)(2 * BT);
      g.fstride[0] = l == 0 ? Cl / 2 : H;
      g.y_fstride = (int)(BT * 8 * H);
      if (pipe && l == 1) pipe_gx1[0] = g; else b.push(F, OP_RUNGEMM, 200 + l).g = g;
    }
    if (!stepped) {
      LstmRec r;
      std::memset(&r, 0, sizeof(r));
      r.gx = ls[l].gxb;
      r.whh[0] = b.pptr(pp + ".real_lstm.weight_hh_l0"); r.whh[1] = b.pptr(pp + ".imag_lstm.weight_hh_l0");
      r.h = ls[l].h; r.gates = ls[l].gates; r.c = ls[l].cst;
      r.dh = r.dgates = b.none();
      for (int g4 = 0; g4 < 4; ++g4) r.gx_goff[g4] = (int64_t)(g4 / 2) * BT * 8 * H + (int64_t)(g4 % 2) * 4 * H;
      r.gx_ld = 8 * H; r.G = 4; r.nset = 2; r.B = B; r.T = T; r.H = H; r.hdt = adt; r.gdt = DT_F32;
      if (pipe) pipe_rec[l] = r; else b.push(F, OP_LSTM_FWD, 200 + l).lstm = r;
    } else {
      // per time step: gx[t] += h[t-1] . W_hh^T (one GEMM per parameter set over the 2B rows (part, b)), then one cell launch
      // over the 4 groups; gx is overwritten in place by the gates i,f,g,o, which is what the backward cells read
      const ParamInfo* Whh[2] = {&b.par(pp + ".real_lstm.weight_hh_l0"), &b.par(pp + ".imag_lstm.weight_hh_l0")};
      for (int set = 0; set < 2; ++set) {
        RunGemm g = Builder::gemm0();
        g.x[0] = ls[l].h; g.xdt = adt; g.ydt = DT_F32;
        g.bstride[0] = 0; g.tstride[0] = 2 * BT * H; g.fstride[0] = T * H; g.rowlen[0] = (int)(BT * H); g.Tin[0] = 2;
        g.M = 2 * B; g.Tout = 2; g.Fo = B;
        g.nseg = 1; g.seg[0] = Seg{0, 0, 0, H, 0};
        g.N = 4 * H;
        Builder::layout_segs(g);
        const ParamInfo* Wp = Whh[set];
        Builder::Coef chh = [=](int nn, int sg, int j) -> int32_t { return pe(*Wp, (int64_t)gate_torch_row(nn, H) * H + j, 1); };
        b.pack_weights(F, g, chh, nm + ".hh" + std::to_string(set), 200 + l);
        g.y = ls[l].gxb; g.y_bstride = 0; g.y_tstride = (int)(BT * 8 * H); g.y_fstride = T * 8 * H; g.flags = kRunAccum;
        ls[l].hh[set] = g;
      }
      for (int t = 0; t < T; ++t) {
        if (t > 0)
          for (int set = 0; set < 2; ++set) {
            RunGemm g = ls[l].hh[set];
            g.base[0] = (int64_t)set * BT * H + (int64_t)(t - 1) * H;
            g.y_off = t * 8 * H + set * 4 * H;
            b.push(F, OP_RUNGEMM, 200 + l).g = g;
          }
        LstmCell& cl = b.push(F, OP_CELL_FWD, 200 + l).cell;
        cl.gates = b.mk(A_WS, ls[l].gxb.off + (int64_t)t * 8 * H * 4);
        cl.c = b.mk(A_WS, ls[l].cst.off + (int64_t)t * H * 4);
        cl.c_prev = t > 0 ? b.mk(A_WS, ls[l].cst.off + (int64_t)(t - 1) * H * 4) : b.none();
        cl.h = b.mk(A_WS, ls[l].h.off + (int64_t)t * H * esize(adt));
        cl.dh = cl.dc = cl.dgates = b.none();
        cl.rows = 4 * B; cl.H = H; cl.hdt = adt; cl.gdt = adt; cl.first = t == 0;
        cl.G = 4; cl.Bg = B; cl.unit_major = 1;
        cl.rs[0] = (int64_t)T * 8 * H; cl.rs[1] = cl.rs[2] = cl.rs[3] = (int64_t)T * H; cl.rs[4] = (int64_t)T * 8 * H;
        for (int g4 = 0; g4 < 4; ++g4) {
          cl.go[0][g4] = cl.go[4][g4] = (int64_t)(g4 / 2) * BT * 8 * H + (int64_t)(g4 % 2) * 4 * H;
          cl.go[1][g4] = cl.go[2][g4] = cl.go[3][g4] = (int64_t)g4 * BT * H;
        }
      }
    }
    if (!pipe) {
      Op& op = b.push(F, OP_COMBINE_FWD, 200 + l);
      op.comb.h = ls[l].h; op.comb.out = ls[l].hc; op.comb.rows = BT; op.comb.H = H; op.comb.dt = adt; op.comb.T = T;
    }
    lin = ls[l].hc;
  }
  if (pipe) {
    for (int c = 0; c < nchunk; ++c) {
      const int t0 = (int)((int64_t)T * c / nchunk), t1 = (int)((int64_t)T * (c + 1) / nchunk), Tc = t1 - t0;
      {
        LstmRec r = pipe_rec[0];
        r.t0 = t0; r.t1 = t1;
        b.push(F, OP_LSTM_FWD, 200).lstm = r;
      }
      // lane 3 (third stream): the input GEMM of layer 1 for this chunk reads layer 0's chunk only, so it runs BESIDE layer 1's recurrence
      // over the previous chunk instead of queueing behind it on the second stream (round 4 timeline: 657 -> ~520 us for the LSTM block)
      static const bool lane3 = !(tune_str("LSTM_LANE3") && atoi(tune_str("LSTM_LANE3")) == 0);
      b.cur_lane = lane3 ? 3 : 2;
      {
        Op& op = b.push(F, OP_COMBINE_FWD, 200);
        op.comb.h = ls[0].h; op.comb.out = ls[0].hc; op.comb.rows = BT; op.comb.H = H; op.comb.dt = adt;
        op.comb.T = T; op.comb.t0 = t0; op.comb.t1 = t1;
      }
      const bool gxm = pipe_gx1[0].Fo == 2;               // both parts in one launch (gx_merge)
      for (int p = 0; p < (gxm ? 1 : 2); ++p) {           // input GEMM of layer 1 for the frames of this chunk
        RunGemm g = pipe_gx1[p];
        g.M = B * Tc * (gxm ? 2 : 1); g.Tout = Tc; g.Tin[0] = Tc;
        g.base[0] += t0 * g.tstride[0];
        g.y_off += t0 * g.y_tstride;
        b.push(F, OP_RUNGEMM, 201).g = g;
      }
      b.cur_lane = 2;
      {
        LstmRec r = pipe_rec[1];
        r.t0 = t0; r.t1 = t1;
        b.push(F, OP_LSTM_FWD, 201).lstm = r;
      }
      b.cur_lane = 0;
    }
    Op& op = b.push(F, OP_COMBINE_FWD, 201);
    op.join = 1;                                           // the main stream needs layer 1's last chunk
    op.comb.h = ls[1].h; op.comb.out = ls[1].hc; op.comb.rows = BT; op.comb.H = H; op.comb.dt = adt; op.comb.T = T;
  }
  // projection r_trans / i_trans (tools_for_model.py:173-175) writing the decoder input [B][T][D][Cl] directly
  Ptr decin = b.ws("decin", BT * D * Cl, adt);
  RunGemm proj = Builder::gemm0();
  Builder::Coef cproj;
  std::function<void(int, int32_t*)> bproj;
  if (!cx) {                                   // tranform: Linear(rnn_units -> D*Cl), output feature c*D + d -> decoder input [B][T][D][Cl]
    const ParamInfo &Wt = b.par("tranform.weight"), &bt = b.par("tranform.bias");
    RunGemm& g = proj;
    g.x[0] = lin; g.xdt = adt; g.ydt = adt;
    g.bstride[0] = (int64_t)T * H; g.tstride[0] = H; g.rowlen[0] = H; g.Tin[0] = T;
    g.M = (int)BT; g.Tout = T; g.Fo = 1;
    g.nseg = 1; g.seg[0] = Seg{0, 0, 0, H, 0};
    g.N = D * Cl;
    Builder::layout_segs(g);
    const ParamInfo *Wp = &Wt, *bp = &bt;
    cproj = [=](int nn, int s_, int j) -> int32_t { const int dd = nn / Cl, cc = nn % Cl; return pe(*Wp, (int64_t)(cc * D + dd) * H + j, 1); };
    bproj = [=](int nn, int32_t* o) { const int dd = nn / Cl, cc = nn % Cl; o[0] = pe(*bp, cc * D + dd, 1); o[1] = 0; };
    b.pack_weights(F, g, cproj, "proj", 300, &bproj);
    g.y = decin; g.y_bstride = (int64_t)T * D * Cl; g.y_tstride = D * Cl;
    b.push(F, OP_RUNGEMM, 300).g = g;
  } else {
    const std::string pp = "enhance." + std::to_string(NL - 1);
    const ParamInfo* Wt[2] = {&b.par(pp + ".r_trans.weight"), &b.par(pp + ".i_trans.weight")};
    const ParamInfo* bt[2] = {&b.par(pp + ".r_trans.bias"), &b.par(pp + ".i_trans.bias")};
    RunGemm& g = proj;
    g.x[0] = lin; g.xdt = adt; g.ydt = adt;
    g.bstride[0] = (int64_t)T * 2 * H; g.tstride[0] = 2 * H; g.rowlen[0] = 2 * H; g.Tin[0] = T;
    g.M = (int)BT; g.Tout = T; g.Fo = 1;
    g.nseg = 1; g.seg[0] = Seg{0, 0, 0, 2 * H, 0};
    g.N = D * Cl;
    Builder::layout_segs(g);
    const int Ch = Cl / 2;
    cproj = [=](int nn, int s, int j) -> int32_t {
      const int dd = nn / Cl, rem = nn % Cl, p = rem / Ch, cc = rem % Ch;
      if ((j >= H) != (p == 1)) return 0;
      return pe(*Wt[p], (int64_t)(cc * D + dd) * H + (j - p * H), 1);
    };
    bproj = [=](int nn, int32_t* o) {
      const int dd = nn / Cl, rem = nn % Cl, p = rem / Ch, cc = rem % Ch;
      o[0] = pe(*bt[p], cc * D + dd, 1); o[1] = 0;
    };
    b.pack_weights(F, g, cproj, "proj", 300, &bproj);
    g.y = decin; g.y_bstride = (int64_t)T * D * Cl; g.y_tstride = D * Cl; g.y_fstride = 0; g.y_off = 0;
    b.push(F, OP_RUNGEMM, 300).g = g;
  }

  // ------------------------------------------------------------------ decoder (models.py:222-226; sub-pixel phases)
  std::vector<Ptr> decy(n), decz(n), dec_mi(n);
  struct DecSrc { Ptr p; int64_t bstride; int tstride, base, C; };
  DecSrc dprev{decin, (int64_t)T * D * Cl, D * Cl, 0, Cl};
  std::vector<std::array<DecSrc, 2>> dec_src(n);
  for (int d = 0; d < n; ++d) {
    const int idx = n - d;
    const int C0 = ch[idx], C1 = cfg.skip ? ch[idx] : 0, Co = ch[idx - 1];
    const int Fi = Fe[idx], Fo = 2 * Fi;
    const bool last = (idx == 1);
    const int Cob = last ? std::max(Co, CP) : Co;   // channels of the output BUFFER (mask layer: 2 -> 8, pad stays 0)
    const std::string nm = "dec" + std::to_string(d);
    const std::string pp = "decoder." + std::to_string(d);
    const ParamInfo &Wr = b.par(pp + ".0.real_conv.weight"), &Wi = b.par(pp + ".0.imag_conv.weight");
    const ParamInfo &br = b.par(pp + ".0.real_conv.bias"), &bi = b.par(pp + ".0.imag_conv.bias");
    const int Co2 = Co / 2, Cin2 = (C0 + C1) / 2;
    const int64_t Rr = (int64_t)B * (T + 1) * Fo;
    decy[d] = b.ws(nm + ".y", Rr * Cob, adt);
    if (!last) { decz[d] = b.ws(nm + ".z", Rr * Co, adt); dec_mi[d] = b.ws(nm + ".mi", 2 * Co, DT_F32); }
    // reference input-channel index (within the real or imag half) of channel c of source s (complex_cat order)
    auto refc = [=](int s, int cc, bool& imag) {
      const int Cs = s == 0 ? C0 : C1;
      imag = cc >= Cs / 2;
      const int q = imag ? cc - Cs / 2 : cc;
      return s == 0 ? q : C0 / 2 + q;
    };
    std::function<int32_t(int, int, int, int, int)> wcoef = [=](int nn, int s, int cc, int kh, int kw) -> int32_t {
      if (nn >= Co) return 0;                        // pad output channel
      bool ii;
      const int rc = refc(s, cc, ii);
      const bool oi = nn >= Co2;
      const int co2 = oi ? nn - Co2 : nn;
      const int64_t ix = (((int64_t)rc * Co2 + co2) * KS + kh) * 2 + kw;
      if (!oi) return ii ? pe(Wi, ix, -1) : pe(Wr, ix, 1);
      return ii ? pe(Wr, ix, 1) : pe(Wi, ix, 1);
    };
    std::function<void(int, int32_t*)> bias = [=](int nn, int32_t* o) {
      if (nn >= Co) { o[0] = o[1] = 0; }
      else if (nn < Co2) { o[0] = pe(br, nn, 1); o[1] = pe(bi, nn, -1); }
      else { o[0] = pe(br, nn - Co2, 1); o[1] = pe(bi, nn - Co2, 1); }
    };
    (void)Cin2;
    const int nblk1 = (int)(((int64_t)B * (T + 1) * Fi + kBM - 1) / kBM);
    Ptr part = b.none();
    int npad_stat = (int)rup(Co, bn_of(Co));
    if (!last) part = b.ws(nm + ".stat", (int64_t)2 * nblk1 * 2 * npad_stat, DT_F32);
    // Thin layers (Cob <= SEFD_PHASE_MERGE_MAXN, default 32: dec4 and the mask layer): ONE GEMM for both sub-pixel phases - the even
    // phase's runs (input bins f-1, f, f+1, two frames), 2 * Cob output columns [phase][channel] (= bins 2f and 2f+1 of the output row:
    // contiguous in the channels-last buffer), zero weights where the odd phase has no tap.  These layers are bound by streaming the
    // tap-expanded activation operand through L2 -> LDS, not by MFMAs: 20 % more MACs, the operand streamed once instead of twice.
    // The backward reads only the per-phase coefficient functions.
    const int merge_maxn = tune_str("PHASE_MERGE_MAXN") ? atoi(tune_str("PHASE_MERGE_MAXN")) : 64;
    const bool merge = Cob <= merge_maxn && !(tune_str("WG_SWAP") && atoi(tune_str("WG_SWAP")) == 0);
    for (int par = 0; par < 2; ++par) {
      RunGemm g = Builder::gemm0();
      g.xdt = adt; g.ydt = adt;
      const int nsrc = cfg.skip ? 2 : 1;
      DecSrc src[2] = {dprev, DecSrc{encz[idx - 1], (int64_t)T * Fi * C1, Fi * C1, 0, C1}};
      dec_src[d] = {src[0], src[1]};
      g.nseg = 0;
      const int ntap = par == 0 ? 3 : 2;
      for (int s = 0; s < nsrc; ++s) {
        g.x[s] = src[s].p; g.bstride[s] = src[s].bstride; g.tstride[s] = src[s].tstride; g.base[s] = src[s].base;
        g.rowlen[s] = Fi * src[s].C; g.fstride[s] = src[s].C; g.Tin[s] = T;
        for (int kw = 0; kw < 2; ++kw) g.seg[g.nseg++] = Seg{s, -kw, par == 0 ? -src[s].C : 0, ntap * src[s].C, 0};
      }
      g.M = B * (T + 1) * Fi; g.Tout = T + 1; g.Fo = Fi;
      g.N = Cob;
      Builder::layout_segs(g);
      const int c0 = C0, c1 = C1;
      Builder::Coef coef = [=](int nn, int sg, int j) -> int32_t {
        const int s = sg / 2, kw = sg % 2;
        const int Cs = s == 0 ? c0 : c1;
        const int jj = j / Cs, cc = j % Cs;
        const int kh = par == 0 ? 4 - 2 * jj : 3 - 2 * jj;
        return wcoef(nn, s, cc, kh, kw);
      };
      if (merge) { dec[d].f[par] = g; dec[d].coef[par] = coef; continue; }     // descriptor only (no packed weights): see below
      b.pack_weights(F, g, coef, nm + ".p" + std::to_string(par), 400 + d, par == 0 ? &bias : nullptr);
      if (par == 1) g.bias = dec[d].f[0].bias;
      g.y = decy[d]; g.y_bstride = (int64_t)(T + 1) * Fo * Cob; g.y_tstride = Fo * Cob; g.y_fstride = 2 * Cob; g.y_off = par * Cob;
      if (!last && cfg.training && !cbn) g.stats = b.mk(A_WS, part.off + (int64_t)par * nblk1 * 2 * g.Npad * 4);
      b.push(F, OP_RUNGEMM, 400 + d).g = g;
      dec[d].f[par] = g; dec[d].coef[par] = coef;
    }
    int fin_nblk = 2 * nblk1, fin_cpad = npad_stat, fin_nsub = 0;
    if (merge) {
      RunGemm g = dec[d].f[0];                  // the even phase's runs
      g.N = 2 * Cob;
      Builder::layout_segs(g);
      const Builder::Coef f0 = dec[d].coef[0], f1 = dec[d].coef[1];
      const int c0 = C0, c1 = C1, cob = Cob;
      Builder::Coef coef = [=](int nn, int sg, int j) -> int32_t {
        if (nn >= 2 * cob) return 0;
        if (nn < cob) return f0(nn, sg, j);
        const int Cs = sg / 2 == 0 ? c0 : c1;
        return j < Cs ? 0 : f1(nn - cob, sg, j - Cs);          // the odd phase's taps are bins f, f+1: one bin into the even phase's run
      };
      std::function<void(int, int32_t*)> bias2 = [=](int nn, int32_t* o) { if (nn >= 2 * cob) { o[0] = o[1] = 0; } else bias(nn % cob, o); };
      b.pack_weights(F, g, coef, nm + ".pm", 400 + d, &bias2);
      g.y = decy[d]; g.y_bstride = (int64_t)(T + 1) * Fo * Cob; g.y_tstride = Fo * Cob; g.y_fstride = 2 * Cob; g.y_off = 0;
      if (!last && cfg.training && !cbn) {
        if ((int64_t)nblk1 * 2 * g.Npad > (int64_t)2 * nblk1 * 2 * npad_stat) { P->error = "merged sub-pixel GEMM: statistics pitch"; return P; }
        g.stats = part;
        fin_nblk = nblk1; fin_cpad = g.Npad; fin_nsub = 2;
      }
      b.push(F, OP_RUNGEMM, 400 + d).g = g;
    }
    dec[d].bias = bias; dec[d].C = Co; dec[d].Fq = Fo; dec[d].R = Rr;
    if (!last && cbn) {
      dec_mi[d] = cbn_fwd(400 + d, pp, nm, decy[d], decz[d], Co, Rr);
      if (!P->error.empty()) return P;
      dprev = DecSrc{decz[d], (int64_t)(T + 1) * Fo * Co, Fo * Co, Fo * Co, Co};
    } else if (!last) {
      Op& op = b.push(F, OP_BN_FINALIZE, 400 + d);
      op.bnf.part = part; op.bnf.mean_invstd = dec_mi[d];
      op.bnf.running_mean = b.sptr(pp + ".1.running_mean"); op.bnf.running_var = b.sptr(pp + ".1.running_var");
      op.bnf.nblk = cfg.training ? fin_nblk : -1; op.bnf.C = Co; op.bnf.Cpad = fin_cpad; op.bnf.count = (double)Rr;
      op.bnf.nsub = fin_nsub; op.bnf.substride = Cob;
      op.bnf.eps = 1e-5f; op.bnf.momentum = 0.1f;
      Op& oa = b.push(F, OP_BN_APPLY, 400 + d);
      oa.bna.y = decy[d]; oa.bna.z = decz[d]; oa.bna.mean_invstd = dec_mi[d];
      oa.bna.gamma = b.pptr(pp + ".1.weight"); oa.bna.beta = b.pptr(pp + ".1.bias"); oa.bna.slope = b.pptr(pp + ".2.weight");
      oa.bna.R = Rr; oa.bna.C = Co; oa.bna.dt = adt;
      dprev = DecSrc{decz[d], (int64_t)(T + 1) * Fo * Co, Fo * Co, Fo * Co, Co};   // frames 1..T of the T+1 buffer
    }
  }

  // ------------------------------------------------------------------ mask, iSTFT, outputs (models.py:253-282)
  Ptr est = b.ws("est", BT * SW, DT_F32);
  Ptr frames = b.ws("frames", BT * W, DT_F32);
  Mask mk;
  std::memset(&mk, 0, sizeof(mk));
  {
    const int Fo = Fe[0], Co = std::max(2, CP);
    mk.spec = spec; mk.mask = decy[n - 1]; mk.est = est; mk.dest = mk.dmask = b.none();
    mk.frames = BT; mk.NF = NF; mk.mode = cfg.mask_mode; mk.mdt = adt; mk.mch = Co; mk.estm = mk.destm = b.none();
    mk.mask_fstride = (int64_t)Fo * Co; mk.mask_bstride = (int64_t)(T + 1) * Fo * Co; mk.mask_base = (int64_t)Fo * Co; mk.T = T;
    b.push(F, OP_MASK_FWD, 500).mask = mk;
  }
  RunGemm gi = Builder::gemm0();
  if (!b.istft_fft(F, 501, est, frames, BT, NFFT, win)) {
    RunGemm& g = gi;
    g.x[0] = est; g.xdt = DT_F32; g.ydt = DT_F32;
    g.bstride[0] = (int64_t)T * SW; g.tstride[0] = SW; g.rowlen[0] = SW; g.Tin[0] = T;
    g.M = (int)BT; g.Tout = T; g.Fo = 1;
    g.nseg = 1; g.seg[0] = Seg{0, 0, 0, SW, 0};
    g.N = W;
    Builder::layout_segs(g);
    const_weights(g, [&](int nn, int j) { return j < 2 ? 0.0 : Kinv[((size_t)(j & 1) * NF + (j / 2 - 1)) * W + nn]; });
    g.y = frames; g.y_bstride = (int64_t)T * W; g.y_tstride = W;
    b.push(F, OP_RUNGEMM, 501).g = g;
  }
  Ola ola;
  std::memset(&ola, 0, sizeof(ola));
  ola.frames = frames; ola.wav = io_out; ola.coff = c_coff; ola.dwav = ola.dpad = b.none();
  ola.B = B; ola.T = T; ola.L = L; ola.win = W; ola.hop = hop; ola.trim = trim;
  b.push(F, OP_OLA_FWD, 502).ola = ola;
  SpecOut so;
  std::memset(&so, 0, sizeof(so));
  so.est = est; so.out_real = io_or; so.out_imag = io_oi; so.B = B; so.T = T; so.NF = NF; so.accumulate = 0;
  b.push(F, OP_SPECOUT_FWD, 503).so = so;

  // =================================================================================================== backward
  if (cfg.training) {
    Ptr dpad = b.ws("dpad", (int64_t)B * Lp, DT_F32);
    Ptr dest = b.ws("dest", BT * SW, DT_F32);
    {
      Ola o = ola;
      o.dwav = io_gw; o.dpad = dpad;
      b.push(R, OP_OLA_BWD, 502).ola = o;
      if (!b.istft_bwd_fft(R, 501, dpad, dest, B, Lp, T, hop, NFFT, win)) {
        RunGemm g = Builder::gemm0();
        g.x[0] = dpad; g.xdt = DT_F32; g.ydt = DT_F32;
        g.bstride[0] = Lp; g.tstride[0] = 0; g.rowlen[0] = Lp; g.fstride[0] = hop; g.Tin[0] = 1;
        g.M = (int)BT; g.Tout = 1; g.Fo = T;
        g.nseg = 1; g.seg[0] = Seg{0, 0, 0, W, 0};
        g.N = SW;
        Builder::layout_segs(g);
        const_weights(g, [&](int nn, int j) { return nn < 2 ? 0.0 : Kinv[((size_t)(nn & 1) * NF + (nn / 2 - 1)) * W + j]; });
        g.y = dest; g.y_bstride = (int64_t)T * SW; g.y_fstride = SW;
        b.push(R, OP_RUNGEMM, 501).g = g;
      }
      SpecOut s2 = so;
      s2.est = dest; s2.out_real = io_gr; s2.out_imag = io_gi; s2.accumulate = 1;
      b.push(R, OP_SPECOUT_BWD, 503).so = s2;
    }
    // gradient buffers
    std::vector<Ptr> d_decy(n), d_decz(n), d_skip(n), d_encz(n), d_ency(n);
    for (int d = 0; d < n; ++d) {
      const int idx = n - d;
      const int Co = idx == 1 ? std::max(ch[idx - 1], CP) : ch[idx - 1], Fo = 2 * Fe[idx];
      d_decy[d] = b.ws("dec" + std::to_string(d) + ".dy", (int64_t)B * (T + 1) * Fo * Co, adt);
      if (idx != 1) d_decz[d] = b.ws("dec" + std::to_string(d) + ".dz", (int64_t)B * T * Fo * Co, adt);
    }
    for (int i = 0; i < n; ++i) {
      const int64_t e = (int64_t)B * T * Fe[i + 1] * ch[i + 1];
      d_ency[i] = b.ws("enc" + std::to_string(i) + ".dy", e, adt);
      d_encz[i] = b.ws("enc" + std::to_string(i) + ".dz", e, adt);
      if (cfg.skip) d_skip[i] = b.ws("enc" + std::to_string(i) + ".dskip", e, adt);
    }
    constexpr int kCsRows = 2048;                // workgroups of MASK_BWD when it also leaves the mask layer's bias-gradient shares
    const bool mask_colsum = !(tune_str("MASK_COLSUM") && atoi(tune_str("MASK_COLSUM")) == 0) && CP >= 2 && CP <= 8 &&
                             !(tune_str("WG_SWAP") && atoi(tune_str("WG_SWAP")) == 0);
    int mask_colsum_op = -1;
    Ptr d_decin = b.ws("decin.d", BT * D * Cl, adt);
    {
      Mask m2 = mk;
      m2.dest = dest; m2.dmask = d_decy[n - 1];
      if (mask_colsum) { m2.colsum_rows = kCsRows; mask_colsum_op = (int)R.size(); }
      b.push(R, OP_MASK_BWD, 500).mask = m2;
    }
    // BatchNorm backward reductions in the epilogues of the GEMMs that PRODUCE the upstream gradient (kRunBnBwd): every dgrad GEMM
    // that writes (a component of) dz of a BatchNorm layer gets the layer's forward output and parameters and a range of partial rows;
    // BN_BWD_FINALIZE then adds all of them.  The separate reduce pass (two or three tensor reads per layer) is gone.  Not fused: the
    // last encoder layer (its dz arrives in channel slices from the LSTM input-gradient GEMMs).
    // Which layers: measured on the default model (profiles/r03_tuning_notes.md) the extra epilogue read costs the wide-tile kernel
    // (cgemm256, N % 256 == 0, compute-bound) 10-17 us per launch against 38-91 us for the pass it replaces, but it costs the thin
    // GEMMs (N <= 128: latency-bound tiles that stream at ~2 TB/s) 45-85 us per launch - more than the pass, which streams at 4-5 TB/s.
    // So by default only the layers whose producers all run on the wide-tile kernel are fused (bf16, C % 256 == 0).
    // SEFD_BN_FUSE=0: none; SEFD_BN_FUSE=2: every layer (the per-op tests run the epilogue of all three GEMM kernels that way).
    const int bn_fuse_mode = tune_str("BN_FUSE") ? atoi(tune_str("BN_FUSE")) : 1;
    const bool bn_fuse = bn_fuse_mode != 0 && !cbn;
    auto bn_fuse_layer = [&](int C, int64_t Rr) { return bn_fuse_mode == 2 || (adt == DT_BF16 && C % 256 == 0 && Rr >= 8192); };
    struct BnbAcc { Ptr part; int rows = 0, cap = 0, ldp = 0; bool on = false; Ptr y, mi; std::string pp; };
    std::vector<BnbAcc> bnb_dec(n), bnb_enc(n);
    auto bnb_init = [&](BnbAcc& a, const std::string& nm, Ptr y, Ptr mi, const std::string& pp, int C, int64_t Rr) {
      a.on = true; a.y = y; a.mi = mi; a.pp = pp;
      a.ldp = (int)rup(C, bn_of(C));
      a.cap = (int)(2 * ((Rr + kBM - 1) / kBM) + 16);
      a.part = b.ws(nm + ".bnpart", (int64_t)a.cap * 3 * a.ldp, DT_F32);
    };
    if (bn_fuse) {
      for (int d = 0; d + 1 < n; ++d) if (bn_fuse_layer(dec[d].C, dec[d].R)) bnb_init(bnb_dec[d], "dec" + std::to_string(d), decy[d], dec_mi[d], "decoder." + std::to_string(d), dec[d].C, dec[d].R);
      for (int i = 0; i + 1 < n; ++i) if (bn_fuse_layer(enc[i].C, enc[i].R)) bnb_init(bnb_enc[i], "enc" + std::to_string(i), ency[i], enc_mi[i], "encoder." + std::to_string(i), enc[i].C, enc[i].R);
    }
    // the GEMM `g` writes dz rows (b, u, fo) of that layer; (bs, ts, fs, off) address the same rows of the layer's forward output y
    auto bnb_attach = [&](RunGemm& g, BnbAcc& a, int64_t bs, int ts, int fs, int off) {
      if (!a.on) return;
      const int rows = (g.M + kBM - 1) / kBM;
      if (a.rows + rows > a.cap || g.Npad != a.ldp) { P->error = "BatchNorm backward partial rows: capacity / pitch"; return; }
      g.flags |= kRunBnBwd;
      g.bnb_y = a.y; g.bnb_mi = a.mi;
      g.bnb_gamma = b.pptr(a.pp + ".1.weight"); g.bnb_beta = b.pptr(a.pp + ".1.bias"); g.bnb_slope = b.pptr(a.pp + ".2.weight");
      g.bnb_bstride = bs; g.bnb_tstride = ts; g.bnb_fstride = fs; g.bnb_off = off;
      g.stats = b.mk(A_WS, a.part.off + (int64_t)a.rows * 3 * a.ldp * 4);
      a.rows += rows;
    };
    BnBwdApply last_bnb;                         // the BnBwdApply of the most recent bn_bwd (the fused first-layer weight gradient reads its totals)
    std::memset(&last_bnb, 0, sizeof(last_bnb));
    auto bn_bwd = [&](int tag, Ptr y, Ptr dz0, Ptr dz1, Ptr mi, const std::string& pp, int C, int64_t Rr, int64_t rpb, int skip, Ptr dy,
                      const std::string& nm, const BnbAcc* fused, bool no_apply = false) {
      int64_t rpbk = std::max<int64_t>(64, (Rr + 2047) / 2048);
      const int nblk = (int)((Rr + rpbk - 1) / rpbk);
      if (cbn) {                                  // `mi` is the layer's coefficient table (cbn_fwd)
        const int h = C / 2;
        CbnBwd c;
        std::memset(&c, 0, sizeof(c));
        c.y = y; c.dz0 = dz0; c.dz1 = dz1; c.dy = dy; c.coef = mi;
        c.coefb = b.ws(nm + ".ccoefb", 9 * h, DT_F32);
        c.part = b.ws(nm + ".cbnpart", (int64_t)nblk * 7 * h, DT_F32);
        const char* wn[3] = {"Wrr", "Wri", "Wii"};
        for (int q = 0; q < 3; ++q) { c.W[q] = b.pptr(pp + ".1." + wn[q]); c.dW[q] = b.pptr(pp + ".1." + wn[q], A_GRAD); }
        c.dB[0] = b.pptr(pp + ".1.Br", A_GRAD); c.dB[1] = b.pptr(pp + ".1.Bi", A_GRAD);
        c.slope = b.pptr(pp + ".2.weight"); c.dslope = b.pptr(pp + ".2.weight", A_GRAD);
        c.R = Rr; c.rpb = rpb; c.C = C; c.dt = adt; c.nblk = nblk; c.rows_per_blk = (int)rpbk; c.skip = skip; c.count = (double)Rr;
        b.push(R, OP_CBN_BWD_REDUCE, tag).cbb = c;
        b.push(R, OP_CBN_BWD_FINALIZE, tag).cbb = c;
        b.push(R, OP_CBN_BWD_APPLY, tag).cbb = c;
        return;
      }
      BnBwdReduce r;
      std::memset(&r, 0, sizeof(r));
      r.y = y; r.dz0 = dz0; r.dz1 = dz1; r.mean_invstd = mi;
      r.gamma = b.pptr(pp + ".1.weight"); r.beta = b.pptr(pp + ".1.bias"); r.slope = b.pptr(pp + ".2.weight");
      r.R = Rr; r.C = C; r.dt = adt; r.nblk = nblk; r.rows_per_blk = (int)rpbk; r.rpb = rpb; r.skip = skip;
      if (fused && fused->on) {                   // the producers' epilogues wrote the partial rows
        r.part = fused->part; r.nblk = fused->rows; r.ldp = fused->ldp;
      } else {
        r.part = b.ws(nm + ".bnpart", (int64_t)nblk * 3 * C, DT_F32);
        b.push(R, OP_BN_BWD_REDUCE, tag).bnr = r;
      }
      BnBwdApply a;
      std::memset(&a, 0, sizeof(a));
      a.r = r; a.totals = b.ws(nm + ".bntot", 3 * C, DT_F32); a.dy = dy;
      a.dgamma = b.pptr(pp + ".1.weight", A_GRAD); a.dbeta = b.pptr(pp + ".1.bias", A_GRAD); a.dslope = b.pptr(pp + ".2.weight", A_GRAD);
      a.count = (double)Rr;
      b.push(R, OP_BN_BWD_FINALIZE, tag).bnb = a;
      if (!no_apply) b.push(R, OP_BN_BWD_APPLY, tag).bnb = a;
      last_bnb = a;
    };

    // ---- decoder backward
    for (int d = n - 1; d >= 0; --d) {
      const int idx = n - d;
      const int C0 = ch[idx], C1 = cfg.skip ? ch[idx] : 0;
      const int Fi = Fe[idx], Fo = 2 * Fi;
      const bool last = (idx == 1);
      const int Co = last ? std::max(ch[idx - 1], CP) : ch[idx - 1];     // buffer channels (pad rows of the mask layer carry zero weights)
      const std::string nm = "dec" + std::to_string(d);
      const std::string pp = "decoder." + std::to_string(d);
      if (!last)
        bn_bwd(400 + d, decy[d], d_decz[d], b.none(), dec_mi[d], pp, Co, dec[d].R, (int64_t)(T + 1) * Fo, Fo, d_decy[d], nm, &bnb_dec[d]);
      // Weight gradients.  Forward form (SEFD_WG_SWAP=0): one WGRAD per sub-pixel phase, A = the forward runs (3 or 2 taps x C channels of
      // both sources, two frames: every input element is streamed through LDS ~5 times per phase pair), dense operand = dy.
      // Swapped form (default): the SAME tensor, contracted over INPUT pixels - dense operand = the source activation x_s (each element
      // read once), A = the runs of the input-gradient GEMM over dy (5 taps x Co channels, two frames): the tap expansion moves to the
      // operand with the FEWER channels (Co <= C_in / 2 in every decoder layer).  Mask layer: 3.0 GB -> 1.3 GB through LDS-DMA.
      // The bias gradient needs its own pass over dy then (ones run only) - planned for the mask layer; a conv bias in front of
      // BatchNorm has an identically zero gradient (the sum over all rows of the BatchNorm input gradient vanishes), which the reference
      // computes as rounding noise and this plan leaves at exactly 0.
      const bool wg_swap = !(tune_str("WG_SWAP") && atoi(tune_str("WG_SWAP")) == 0);
      b.cur_lane = 1;                           // weight gradients of the decoder: nothing downstream needs them before UNPACK
      if (!wg_swap) for (int par = 0; par < 2; ++par) b.wgrad(R, dec[d].f[par], d_decy[d], dec[d].coef[par], 400 + d, &dec[d].bias);
      else if (!last) {                          // conv biases in front of BatchNorm: UNPACK writes their exact zero
        b.zero_grad.resize(nparam, 0);
        for (const char* part : {".0.real_conv.bias", ".0.imag_conv.bias"}) {
          const ParamInfo& pb = b.par(pp + part);
          for (int64_t e = 0; e < pb.numel; ++e) b.zero_grad[pb.off + e] = 1;
        }
      } else if (mask_colsum_op >= 0) {
        // the mask layer's bias gradient = column sums of dmask: MASK_BWD's workgroups leave their shares in the partial-sum buffer
        // ([kCsRows][8] fp32), a SPLITSUM folds them to 128 rows and UNPACK adds those (a 110 us bias-only WGRAD pass over dmask before)
        const int64_t rel = b.gp_off;
        b.gp_off += (int64_t)kCsRows * 8;
        b.fixes.push_back(Builder::Fix{mask_colsum_op, rel, 2});
        b.split_sum(R, rel, 128 * 8, kCsRows / 128, 400 + d);
        for (int nn = 0; nn < Co; ++nn) {
          int32_t bt[2] = {0, 0};
          dec[d].bias(nn, bt);
          for (int r = 0; r < 128; ++r) {
            const int64_t pos = rel + (int64_t)r * 8 + nn + 1;
            for (int e = 0; e < 2; ++e)
              if (bt[e] != 0) b.inv[std::abs(bt[e]) - 1].push_back((int32_t)(bt[e] > 0 ? pos : -pos));
          }
        }
      } else {
        RunGemm fb = Builder::gemm0();           // all output rows (both phases), no activation run: wgrad() appends the ones run
        fb.xdt = adt; fb.ydt = adt;
        fb.M = B * (T + 1) * Fo; fb.Tout = T + 1; fb.Fo = Fo;
        fb.nseg = 0; fb.N = Co;
        fb.y_bstride = (int64_t)(T + 1) * Fo * Co; fb.y_tstride = Fo * Co; fb.y_fstride = Co; fb.y_off = 0;
        Builder::Coef none_coef = [](int, int, int) -> int32_t { return 0; };
        b.wgrad(R, fb, d_decy[d], none_coef, 400 + d, &dec[d].bias);
      }
      b.cur_lane = 0;
      // input gradients: conv-form over dy [B][T+1][Fo][Co]; dx[ci,f,t] = sum W[ci,co,kh,kw] dy[co, 2f+kh-2, t+kw]
      const int nsrc = cfg.skip ? 2 : 1;
      // Thin layers: ONE GEMM over dy for the input gradients of both sources (previous layer's output | skip connection): the same runs
      // of dy, C0 + C1 output columns, the second half stored to the second destination (RunGemm::y2 / n2).  The A operand - what bounds
      // these layers - is streamed once instead of twice.  Not when a destination's BatchNorm sums ride in the epilogue (one layer per GEMM).
      const int dg_maxn = tune_str("DGRAD_MERGE_MAXN") ? atoi(tune_str("DGRAD_MERGE_MAXN")) : 128;
      const bool dg_merge = nsrc == 2 && C0 == C1 && C0 % 8 == 0 && C0 + C1 <= dg_maxn && !(d > 0 && bnb_dec[d - 1].on) && !bnb_enc[idx - 1].on;
      RunGemm dg_g[2];
      Builder::Coef dg_coef[2];
      for (int s = 0; s < nsrc; ++s) {
        const int Cs = s == 0 ? C0 : C1;
        RunGemm g = Builder::gemm0();
        g.x[0] = d_decy[d]; g.xdt = adt; g.ydt = adt;
        g.bstride[0] = (int64_t)(T + 1) * Fo * Co; g.tstride[0] = Fo * Co; g.base[0] = 0; g.rowlen[0] = Fo * Co; g.fstride[0] = 2 * Co; g.Tin[0] = T + 1;
        g.M = B * T * Fi; g.Tout = T; g.Fo = Fi;
        g.nseg = 2;
        g.seg[0] = Seg{0, 0, -2 * Co, KS * Co, 0};    // kw = 0 : buffer frame u = t
        g.seg[1] = Seg{0, 1, -2 * Co, KS * Co, 0};    // kw = 1 : buffer frame u = t + 1
        g.N = Cs;
        Builder::layout_segs(g);
        // d y[co] / d x[(s,cc)] is the forward coefficient of phase (kh odd) at tap jj: look it up in the forward tables
        const Builder::Coef f0 = dec[d].coef[0], f1 = dec[d].coef[1];
        const int Csx = Cs;
        Builder::Coef coef = [=](int nn, int sg, int j) -> int32_t {
          const int kw = sg, kh = j / Co, co = j % Co;
          const int par = kh & 1;
          const int jj = par == 0 ? (4 - kh) / 2 : (3 - kh) / 2;
          return (par == 0 ? f0 : f1)(co, s * 2 + kw, jj * Csx + nn);
        };
        if (!dg_merge) b.pack_weights(R, g, coef, nm + ".dg" + std::to_string(s), 400 + d);
        if (s == 0) {
          if (d > 0) { g.y = d_decz[d - 1]; }
          else g.y = d_decin;
        } else {
          g.y = d_skip[idx - 1];
        }
        g.y_bstride = (int64_t)T * Fi * Cs; g.y_tstride = Fi * Cs; g.y_fstride = Cs; g.y_off = 0;
        // dz of the previous decoder layer (its y keeps the frame that `[..., 1:]` drops: rows start one frame in) / of encoder layer idx-1
        if (s == 0 && d > 0) bnb_attach(g, bnb_dec[d - 1], (int64_t)(T + 1) * Fi * Cs, Fi * Cs, Cs, Fi * Cs);
        else if (s == 1) bnb_attach(g, bnb_enc[idx - 1], (int64_t)T * Fi * Cs, Fi * Cs, Cs, 0);
        if (!dg_merge) b.push(R, OP_RUNGEMM, 400 + d).g = g;
        dg_g[s] = g; dg_coef[s] = coef;
        if (wg_swap) {                           // weight gradient, swapped form: the runs of this GEMM against the source activation
          RunGemm fw = g;
          fw.flags = 0; fw.stats = b.none(); fw.bias = b.none(); fw.ydt = adt;
          const DecSrc& xs = dec_src[d][s];
          fw.y_bstride = xs.bstride; fw.y_tstride = xs.tstride; fw.y_fstride = xs.C; fw.y_off = xs.base;
          b.cur_lane = 1;
          b.wgrad(R, fw, xs.p, coef, 400 + d, nullptr);
          b.cur_lane = 0;
        }
      }
      if (dg_merge) {
        RunGemm g = dg_g[0];
        g.N = C0 + C1;
        Builder::layout_segs(g);
        const Builder::Coef f0 = dg_coef[0], f1 = dg_coef[1];
        const int c0 = C0;
        Builder::Coef coef = [=](int nn, int sg, int j) -> int32_t { return nn < c0 ? f0(nn, sg, j) : f1(nn - c0, sg, j); };
        b.pack_weights(R, g, coef, nm + ".dgm", 400 + d);
        g.y2 = dg_g[1].y; g.n2 = C0;
        b.push(R, OP_RUNGEMM, 400 + d).g = g;
      }
    }
    // ---- cfg.lstm == 'real': tranform, then the two LSTM layers last to first, then the gradient into the encoder output
    if (!cx) {
      Ptr dh[2] = {b.ws("lstm0.dh", BT * H, DT_F32), b.ws("lstm1.dh", BT * H, DT_F32)};
      b.cur_lane = 1;
      b.wgrad(R, proj, d_decin, cproj, 300, &bproj);
      b.cur_lane = 0;
      {
        RunGemm g = Builder::gemm0();
        g.x[0] = d_decin; g.xdt = adt; g.ydt = DT_F32;
        g.bstride[0] = (int64_t)T * D * Cl; g.tstride[0] = D * Cl; g.rowlen[0] = D * Cl; g.Tin[0] = T;
        g.M = (int)BT; g.Tout = T; g.Fo = 1;
        g.nseg = 1; g.seg[0] = Seg{0, 0, 0, D * Cl, 0};
        g.N = H;
        Builder::layout_segs(g);
        Builder::Coef coef = [=](int nn, int sg, int j) -> int32_t { return cproj(j, 0, nn); };
        b.pack_weights(R, g, coef, "proj.dg", 300);
        g.y = dh[1]; g.y_bstride = (int64_t)T * H; g.y_tstride = H;
        b.push(R, OP_RUNGEMM, 300).g = g;
      }
      for (int l = 1; l >= 0; --l) {
        const std::string nm = "lstm" + std::to_string(l), sl = std::to_string(l);
        RealL& Lr = rl[l];
        Ptr dgates = b.ws(nm + ".dgates", BT * 4 * H, adt);
        if (!stepped) {
          LstmRec& r = b.push(R, OP_LSTM_BWD, 200 + l).lstm;
          r.gx = Lr.gxb; r.whh[0] = r.whh[1] = b.pptr("enhance.weight_hh_l" + sl);
          r.h = Lr.h; r.gates = Lr.gates; r.c = Lr.cst; r.dh = dh[l]; r.dgates = dgates;
          r.gx_ld = 4 * H; r.G = 1; r.nset = 1; r.B = B; r.T = T; r.H = H; r.hdt = adt; r.gdt = adt;
        } else {
          Ptr dcb = b.ws(nm + ".dc", (int64_t)B * H, DT_F32);
          RunGemm rb = Builder::gemm0();
          rb.x[0] = dgates; rb.xdt = adt; rb.ydt = DT_F32;
          rb.fstride[0] = T * 4 * H; rb.rowlen[0] = (int)(BT * 4 * H); rb.Tin[0] = 1;
          rb.M = B; rb.Tout = 1; rb.Fo = B;
          rb.nseg = 1; rb.seg[0] = Seg{0, 0, 0, 4 * H, 0};
          rb.N = H;
          Builder::layout_segs(rb);
          const ParamInfo* Wh = Lr.Whh;
          Builder::Coef cT = [=](int nn, int sg, int j) -> int32_t { return pe(*Wh, (int64_t)gate_torch_row(j, H) * H + nn, 1); };
          b.pack_weights(R, rb, cT, nm + ".hhT", 200 + l);
          rb.y = dh[l]; rb.y_fstride = T * H; rb.flags = kRunAccum;
          for (int t = T - 1; t >= 0; --t) {
            real_cell(b.push(R, OP_CELL_BWD, 200 + l).cell, l, t, false, dh[l], dcb, dgates);
            if (t > 0) {
              RunGemm q = rb;
              q.base[0] = t * 4 * H;
              q.y_off = (t - 1) * H;
              b.push(R, OP_RUNGEMM, 200 + l).g = q;
            }
          }
        }
        RunGemm fw = Lr.gx;
        fw.ydt = adt;
        b.wgrad(R, fw, dgates, Lr.cgx, 200 + l, &Lr.bgx);
        {                                                // W_hh: dW[n][k] = sum_t dgates[t][n] * h[t-1][k]
          RunGemm f = Builder::gemm0();
          f.x[0] = Lr.h; f.xdt = adt; f.ydt = adt;
          f.bstride[0] = (int64_t)T * H; f.tstride[0] = H; f.rowlen[0] = H; f.Tin[0] = T;
          f.M = (int)BT; f.Tout = T; f.Fo = 1;
          f.nseg = 1; f.seg[0] = Seg{0, -1, 0, H, 0};
          f.N = 4 * H;
          Builder::layout_segs(f);
          f.y_bstride = (int64_t)T * 4 * H; f.y_tstride = 4 * H;
          const ParamInfo* Wh = Lr.Whh;
          Builder::Coef chh = [=](int nn, int sg, int j) -> int32_t { return pe(*Wh, (int64_t)gate_torch_row(nn, H) * H + j, 1); };
          b.wgrad(R, f, dgates, chh, 200 + l, nullptr);
        }
        const int nout = l == 0 ? D : 1;                 // input gradient: layer 1 -> dh of layer 0 ; layer 0 -> encoder output, one slice per d
        for (int q = 0; q < nout; ++q) {
          RunGemm g = Builder::gemm0();
          g.x[0] = dgates; g.xdt = adt;
          g.bstride[0] = (int64_t)T * 4 * H; g.tstride[0] = 4 * H; g.rowlen[0] = 4 * H; g.Tin[0] = T;
          g.M = (int)BT; g.Tout = T; g.Fo = 1;
          g.nseg = 1; g.seg[0] = Seg{0, 0, 0, 4 * H, 0};
          g.N = l == 0 ? Cl : H;
          Builder::layout_segs(g);
          const Builder::Coef cf = Lr.cgx;
          Builder::Coef coef = [=](int nn, int sg, int j) -> int32_t { return l == 0 ? cf(j, q, nn) : cf(j, 0, nn); };
          b.pack_weights(R, g, coef, nm + ".dx" + std::to_string(q), 200 + l);
          if (l == 0) { g.ydt = adt; g.y = d_encz[n - 1]; g.y_bstride = (int64_t)T * D * Cl; g.y_tstride = D * Cl; g.y_off = q * Cl; }
          else { g.ydt = DT_F32; g.y = dh[0]; g.y_bstride = (int64_t)T * H; g.y_tstride = H; }
          b.push(R, OP_RUNGEMM, 200 + l).g = g;
        }
      }
    }
    // ---- projection backward
    Ptr dhc_next = b.ws("dhc" + std::to_string(NL - 1), BT * 2 * H, DT_F32);
    if (cx) {
      b.cur_lane = 1;
      b.wgrad(R, proj, d_decin, cproj, 300, &bproj);
      b.cur_lane = 0;
      RunGemm g = Builder::gemm0();
      g.x[0] = d_decin; g.xdt = adt; g.ydt = DT_F32;
      g.bstride[0] = (int64_t)T * D * Cl; g.tstride[0] = D * Cl; g.rowlen[0] = D * Cl; g.Tin[0] = T;
      g.M = (int)BT; g.Tout = T; g.Fo = 1;
      g.nseg = 1; g.seg[0] = Seg{0, 0, 0, D * Cl, 0};
      g.N = 2 * H;
      Builder::layout_segs(g);
      Builder::Coef coef = [=](int nn, int sg, int j) -> int32_t { return cproj(j, 0, nn); };
      b.pack_weights(R, g, coef, "proj.dg", 300);
      g.y = dhc_next; g.y_bstride = (int64_t)T * 2 * H; g.y_tstride = 2 * H;
      b.push(R, OP_RUNGEMM, 300).g = g;
    }
    // ---- LSTM backward
    for (int l = cx ? NL - 1 : -1; l >= 0; --l) {
      const std::string nm = "lstm" + std::to_string(l);
      const std::string pp = "enhance." + std::to_string(l);
      const ParamInfo* Whh[2] = {&b.par(pp + ".real_lstm.weight_hh_l0"), &b.par(pp + ".imag_lstm.weight_hh_l0")};
      Ptr dh = b.ws(nm + ".dh", 4 * BT * H, DT_F32);
      Ptr dgates = b.ws(nm + ".dgates", 2 * BT * 8 * H, adt);
      const int64_t dg_half = BT * 8 * H * esize(adt);
      {
        Op& op = b.push(R, OP_COMBINE_BWD, 200 + l);
        op.comb.h = dh; op.comb.out = dhc_next; op.comb.rows = BT; op.comb.H = H; op.comb.dt = DT_F32; op.comb.T = T;
      }
      if (!stepped) {
        Op& op = b.push(R, OP_LSTM_BWD, 200 + l);
        LstmRec& r = op.lstm;
        r.gx = ls[l].gxb;
        r.whh[0] = b.pptr(pp + ".real_lstm.weight_hh_l0"); r.whh[1] = b.pptr(pp + ".imag_lstm.weight_hh_l0");
        r.h = ls[l].h; r.gates = ls[l].gates; r.c = ls[l].cst; r.dh = dh; r.dgates = dgates;
        for (int g4 = 0; g4 < 4; ++g4) r.gx_goff[g4] = (int64_t)(g4 / 2) * BT * 8 * H + (int64_t)(g4 % 2) * 4 * H;
        r.gx_ld = 8 * H; r.G = 4; r.nset = 2; r.B = B; r.T = T; r.H = H; r.hdt = adt; r.gdt = adt;
      } else {
        // per time step, last to first: cell backward (dgates[t], carry dc), then dh[t-1] += dgates[t] . W_hh per parameter set
        Ptr dcb = b.ws(nm + ".dc", (int64_t)4 * B * H, DT_F32);
        RunGemm rb[2];
        for (int set = 0; set < 2; ++set) {
          RunGemm g = Builder::gemm0();
          g.x[0] = dgates; g.xdt = adt; g.ydt = DT_F32;
          g.bstride[0] = 0; g.tstride[0] = (int)(BT * 8 * H); g.fstride[0] = T * 8 * H; g.rowlen[0] = (int)(BT * 8 * H); g.Tin[0] = 2;
          g.M = 2 * B; g.Tout = 2; g.Fo = B;
          g.nseg = 1; g.seg[0] = Seg{0, 0, 0, 4 * H, 0};
          g.N = H;
          Builder::layout_segs(g);
          const ParamInfo* Wp = Whh[set];
          Builder::Coef cT = [=](int nn, int sg, int j) -> int32_t { return pe(*Wp, (int64_t)gate_torch_row(j, H) * H + nn, 1); };
          b.pack_weights(R, g, cT, nm + ".hhT" + std::to_string(set), 200 + l);
          g.y = dh; g.y_bstride = 0; g.y_tstride = (int)(2 * BT * H); g.y_fstride = T * H; g.flags = kRunAccum;
          rb[set] = g;
        }
        for (int t = T - 1; t >= 0; --t) {
          LstmCell& cl = b.push(R, OP_CELL_BWD, 200 + l).cell;
          cl.gates = b.mk(A_WS, ls[l].gxb.off + (int64_t)t * 8 * H * 4);
          cl.c = b.mk(A_WS, ls[l].cst.off + (int64_t)t * H * 4);
          cl.c_prev = t > 0 ? b.mk(A_WS, ls[l].cst.off + (int64_t)(t - 1) * H * 4) : b.none();
          cl.h = b.none();
          cl.dh = b.mk(A_WS, dh.off + (int64_t)t * H * 4);
          cl.dc = dcb;
          cl.dgates = b.mk(A_WS, dgates.off + (int64_t)t * 8 * H * esize(adt));
          cl.rows = 4 * B; cl.H = H; cl.hdt = adt; cl.gdt = adt; cl.first = t == T - 1;
          cl.G = 4; cl.Bg = B; cl.unit_major = 1;
          cl.rs[0] = (int64_t)T * 8 * H; cl.rs[1] = cl.rs[2] = cl.rs[3] = (int64_t)T * H; cl.rs[4] = (int64_t)T * 8 * H;
          for (int g4 = 0; g4 < 4; ++g4) {
            cl.go[0][g4] = cl.go[4][g4] = (int64_t)(g4 / 2) * BT * 8 * H + (int64_t)(g4 % 2) * 4 * H;
            cl.go[1][g4] = cl.go[2][g4] = cl.go[3][g4] = (int64_t)g4 * BT * H;
          }
          if (t > 0)
            for (int set = 0; set < 2; ++set) {
              RunGemm g = rb[set];
              g.base[0] = (int64_t)t * 8 * H + (int64_t)set * 4 * H;
              g.y_off = (int)((int64_t)set * BT * H + (int64_t)(t - 1) * H);
              b.push(R, OP_RUNGEMM, 200 + l).g = g;
            }
        }
      }
      for (int p = 0; p < 2; ++p) {
        Ptr dyp = b.mk(A_WS, dgates.off + (int64_t)p * dg_half);
        RunGemm fw = ls[l].gx[p];
        fw.ydt = adt;                       // WGRAD reads dy = dgates (act dtype), not the fp32 gx the forward wrote
        b.cur_lane = lane_all ? 1 : 0;
        b.wgrad(R, fw, dyp, ls[l].cgx[p], 200 + l, &ls[l].bgx);
        b.cur_lane = 0;
      }
      for (int g4 = 0; g4 < 4; ++g4) {       // W_hh: dW[n][k] = sum_t dgates[g][t][n] * h[g][t-1][k]
        const int p = g4 / 2, set = g4 % 2;
        RunGemm f = Builder::gemm0();
        f.x[0] = b.mk(A_WS, ls[l].h.off + (int64_t)g4 * BT * H * esize(adt)); f.xdt = adt; f.ydt = adt;
        f.bstride[0] = (int64_t)T * H; f.tstride[0] = H; f.rowlen[0] = H; f.Tin[0] = T;
        f.M = (int)BT; f.Tout = T; f.Fo = 1;
        f.nseg = 1; f.seg[0] = Seg{0, -1, 0, H, 0};
        f.N = 4 * H;
        Builder::layout_segs(f);
        f.y_bstride = (int64_t)T * 8 * H; f.y_tstride = 8 * H; f.y_off = set * 4 * H;
        const ParamInfo* Wp = Whh[set];
        Builder::Coef coef = [=](int nn, int sg, int j) -> int32_t { return pe(*Wp, (int64_t)gate_torch_row(nn, H) * H + j, 1); };
        Ptr dyp = b.mk(A_WS, dgates.off + (int64_t)p * dg_half);
        b.cur_lane = lane_all ? 1 : 0;
        b.wgrad(R, f, dyp, coef, 200 + l, nullptr);
        b.cur_lane = 0;
      }
      // input gradient of the layer
      Ptr dx_full;
      if (l > 0) dx_full = b.ws("dhc" + std::to_string(l - 1), BT * 2 * H, DT_F32);
      // Layer 0 in bf16: ONE GEMM over both gate halves (two sources, K = 2 x 8H) with block weights - the half of the K range that
      // does not feed an output column is zero - writing the whole [D][Cl] row of d_encz contiguously, instead of 2 x D launches of
      // N = Cl / 2 (M = B*T rows only: 8 x 24 us of latency-bound tiles vs one wide-tile launch; twice the MACs, 65 GFLOP).
      const bool dx_merge = l == 0 && adt == DT_BF16 && (D * Cl) % 256 == 0 && (8 * H) % 64 == 0 &&
                            !(tune_str("DX_MERGE") && atoi(tune_str("DX_MERGE")) == 0);
      if (dx_merge) {
        RunGemm g = Builder::gemm0();
        g.xdt = adt; g.ydt = adt;
        for (int p = 0; p < 2; ++p) {
          g.x[p] = b.mk(A_WS, dgates.off + (int64_t)p * dg_half);
          g.bstride[p] = (int64_t)T * 8 * H; g.tstride[p] = 8 * H; g.rowlen[p] = 8 * H; g.Tin[p] = T;
        }
        g.M = (int)BT; g.Tout = T; g.Fo = 1;
        g.nseg = 2; g.seg[0] = Seg{0, 0, 0, 8 * H, 0}; g.seg[1] = Seg{1, 0, 0, 8 * H, 0};
        g.N = D * Cl;
        Builder::layout_segs(g);
        const Builder::Coef cf0 = ls[l].cgx[0], cf1 = ls[l].cgx[1];
        const int Ch = Cl / 2;
        Builder::Coef coef = [=](int nn, int sg, int j) -> int32_t {
          const int q = nn / Cl, rem = nn % Cl, p = rem / Ch, c = rem % Ch;
          if (sg != p) return 0;
          return (p == 0 ? cf0 : cf1)(j, q, c);
        };
        b.pack_weights(R, g, coef, nm + ".dxm", 200 + l);
        g.y = d_encz[n - 1]; g.y_bstride = (int64_t)T * D * Cl; g.y_tstride = D * Cl; g.y_off = 0;
        b.push(R, OP_RUNGEMM, 200 + l).g = g;
      }
      for (int p = 0; p < (dx_merge ? 0 : 2); ++p) {
        const int nout = l == 0 ? D : 1;
        for (int q = 0; q < nout; ++q) {
          RunGemm g = Builder::gemm0();
          g.x[0] = b.mk(A_WS, dgates.off + (int64_t)p * dg_half); g.xdt = adt;
          g.bstride[0] = (int64_t)T * 8 * H; g.tstride[0] = 8 * H; g.rowlen[0] = 8 * H; g.Tin[0] = T;
          g.M = (int)BT; g.Tout = T; g.Fo = 1;
          g.nseg = 1; g.seg[0] = Seg{0, 0, 0, 8 * H, 0};
          g.N = l == 0 ? Cl / 2 : H;
          Builder::layout_segs(g);
          const Builder::Coef cf = ls[l].cgx[p];
          Builder::Coef coef = [=](int nn, int sg, int j) -> int32_t { return l == 0 ? cf(j, q, nn) : cf(j, 0, nn); };
          b.pack_weights(R, g, coef, nm + ".dx" + std::to_string(p) + "_" + std::to_string(q), 200 + l);
          if (l == 0) {
            g.ydt = adt; g.y = d_encz[n - 1];
            g.y_bstride = (int64_t)T * D * Cl; g.y_tstride = D * Cl; g.y_off = q * Cl + p * (Cl / 2);
          } else {
            g.ydt = DT_F32; g.y = dx_full;
            g.y_bstride = (int64_t)T * 2 * H; g.y_tstride = 2 * H; g.y_off = p * H;
          }
          b.push(R, OP_RUNGEMM, 200 + l).g = g;
        }
      }
      if (l > 0) dhc_next = dx_full;
    }
    // ---- data-parallel overlap: the gradients of decoder + LSTM (flat range [decoder.0 ..., end)) are complete here - every
    // weight gradient GEMM and BatchNorm parameter gradient that writes them has been planned above.  Their UNPACK goes here, so
    // a caller can start their all-reduce while the encoder backward still runs (sefd_plan_grad_bucket / sefd_plan_run_cb).
    // The folds of the decoder + LSTM weight gradients (3/4 of the 1.2 GB of partial sums of a step) go here, on the weight-gradient lane:
    // a bandwidth-bound pass beside the encoder's input-gradient GEMMs instead of in front of the final UNPACK on the main stream.
    if (!(tune_str("SPLITSUM_MID") && atoi(tune_str("SPLITSUM_MID")) == 0)) b.flush_sums(R, 997, true);
    // Without an exchange (one bucket) the same early UNPACK rides the weight-gradient lane (tag 997): the gather of 83 % of the parameters
    // leaves the tail of the main stream (79 us for all of them in front of Adam before); SEFD_UNPACK_MID=0 keeps the single UNPACK.
    const bool unpack_mid = cfg.grad_buckets < 2 && !(tune_str("UNPACK_MID") && atoi(tune_str("UNPACK_MID")) == 0);
    if (cfg.grad_buckets >= 2 || unpack_mid) {
      const int64_t lo = b.par("decoder.0.0.real_conv.weight").off;
      b.flush_sums(R, 997, unpack_mid);                      // (nothing pending unless SEFD_SPLITSUM_MID=0)
      if (unpack_mid) b.cur_lane = 1;
      b.unpack_range(R, lo, nparam, unpack_mid ? 997 : 998);
      b.cur_lane = 0;
      b.unpack_hi = lo;
      if (!unpack_mid) P->bucket_elem = lo;                  // (the op index is looked up after the op list is final)
    }
    // ---- encoder backward
    for (int i = n - 1; i >= 0; --i) {
      const int Ci = ch[i], Co = ch[i + 1], Fi = Fe[i], Fo = Fe[i + 1];
      const std::string nm = "enc" + std::to_string(i);
      const std::string pp = "encoder." + std::to_string(i);
      // First layer on the spectrum (enc0.hip): it has no input gradient, so its BatchNorm input gradient dy is read by the weight gradient alone -
      // BN_BWD_APPLY is not planned, the weight-gradient kernel takes dz through the BatchNorm + PReLU backward as it loads it (kRunDyFromBn) and runs on
      // the MAIN stream right behind BN_BWD_FINALIZE: apply (117 us) -> fold -> weight gradient (52 us) was the serial tail of the step.  ENC0_BNFUSE=0: off
      const bool dy_fused = i == 0 && (enc[0].f[0].flags & kRunEnc0) && !cbn && !(tune_str("ENC0_BNFUSE") && atoi(tune_str("ENC0_BNFUSE")) == 0);
      bn_bwd(100 + i, ency[i], d_encz[i], cfg.skip ? d_skip[i] : b.none(), enc_mi[i], pp, Co, enc[i].R, (int64_t)T * Fo, 0, d_ency[i], nm, &bnb_enc[i], dy_fused);
      // the folds of enc5 .. enc1 go in front of the LAST weight gradient on its lane (its input is the last thing the dgrad chain produces,
      // the lane usually waits for it): the fold in front of the final UNPACK then covers one thin layer
      if (i == 0 && lane_all && n > 1 && !(tune_str("SPLITSUM_MID") && atoi(tune_str("SPLITSUM_MID")) == 0)) b.flush_sums(R, 996, true);
      b.cur_lane = (lane_all && !dy_fused) ? 1 : 0;             // encoder weight gradients next to the dgrad chain
      // Every encoder conv bias sits in front of a training-mode BatchNorm: its gradient is identically zero (the sum over all rows of the
      // BatchNorm input gradient vanishes; the reference computes rounding noise there).  No bias "ones" run in these GEMMs - it cost a
      // whole 64-column K segment (enc0: 192 -> 128 columns, half the K tiles; enc3: 6 -> 5 wide tiles) - UNPACK writes the exact zero.
      const bool enc_bias_zero = !(tune_str("ENC_BIAS_ZERO") && atoi(tune_str("ENC_BIAS_ZERO")) == 0);
      if (enc_bias_zero) {
        b.zero_grad.resize(nparam, 0);
        for (const char* part : {".0.real_conv.bias", ".0.imag_conv.bias"}) {
          const ParamInfo& pb = b.par(pp + part);
          for (int64_t e = 0; e < pb.numel; ++e) b.zero_grad[pb.off + e] = 1;
        }
      }
      b.wgrad(R, enc[i].f[0], dy_fused ? d_encz[i] : d_ency[i], enc[i].coef[0], 100 + i, enc_bias_zero ? nullptr : &enc[i].bias);
      if (dy_fused) {
        for (size_t q = R.size(); q-- > 0;)
          if (R[q].kind == OP_WGRAD && R[q].tag == 100 + i) {
            RunGemm& g = R[q].g;
            g.flags |= kRunDyFromBn;
            g.bnb_dz1 = cfg.skip ? d_skip[i] : b.none();
            g.bnb_y = ency[i]; g.bnb_mi = enc_mi[i];
            g.bnb_gamma = b.pptr(pp + ".1.weight"); g.bnb_beta = b.pptr(pp + ".1.bias"); g.bnb_slope = b.pptr(pp + ".2.weight");
            g.bnb_bstride = g.y_bstride; g.bnb_tstride = g.y_tstride; g.bnb_fstride = g.y_fstride; g.bnb_off = g.y_off;
            g.bnb_totals = last_bnb.totals; g.bnb_inv_count = (float)(1.0 / last_bnb.count);
            break;
          }
      }
      b.cur_lane = 0;
      if (i == 0) continue;
      // dx[ci,f,t] = sum W[co,ci,kh,kw] dy[co,(f+2-kh)/2, t+1-kw]  -> two sub-pixel phases over dy [B][T][Fo][Co]
      // thin layers: both phases in one GEMM over the even phase's runs (see the decoder forward), unless this layer's BatchNorm sums
      // ride in the epilogue (their partial rows have one column per channel)
      const int merge_maxn = tune_str("PHASE_MERGE_MAXN") ? atoi(tune_str("PHASE_MERGE_MAXN")) : 64;
      if (Ci <= merge_maxn && !bnb_enc[i - 1].on) {
        RunGemm g = Builder::gemm0();
        g.x[0] = d_ency[i]; g.xdt = adt; g.ydt = adt;
        g.bstride[0] = (int64_t)T * Fo * Co; g.tstride[0] = Fo * Co; g.rowlen[0] = Fo * Co; g.fstride[0] = Co; g.Tin[0] = T;
        g.M = B * T * Fo; g.Tout = T; g.Fo = Fo;
        g.nseg = 2;
        g.seg[0] = Seg{0, 1, -Co, 3 * Co, 0};
        g.seg[1] = Seg{0, 0, -Co, 3 * Co, 0};
        g.N = 2 * Ci;
        Builder::layout_segs(g);
        const Builder::Coef cf = enc[i].coef[0];
        Builder::Coef coef = [=](int nn, int sg, int j) -> int32_t {
          const int kw = sg, jj = j / Co, co = j % Co, par = nn / Ci;
          if (par > 1 || (par == 1 && jj == 0)) return 0;
          const int kh = par == 0 ? 4 - 2 * jj : 5 - 2 * jj;
          return cf(co, kw, kh * Ci + nn % Ci);
        };
        b.pack_weights(R, g, coef, nm + ".dgm", 100 + i);
        g.y = d_encz[i - 1]; g.y_bstride = (int64_t)T * Fi * Ci; g.y_tstride = Fi * Ci; g.y_fstride = 2 * Ci; g.y_off = 0;
        b.push(R, OP_RUNGEMM, 100 + i).g = g;
        continue;
      }
      for (int par = 0; par < 2; ++par) {
        RunGemm g = Builder::gemm0();
        g.x[0] = d_ency[i]; g.xdt = adt; g.ydt = adt;
        g.bstride[0] = (int64_t)T * Fo * Co; g.tstride[0] = Fo * Co; g.rowlen[0] = Fo * Co; g.fstride[0] = Co; g.Tin[0] = T;
        g.M = B * T * Fo; g.Tout = T; g.Fo = Fo;           // Fi/2 == Fo output rows per phase
        const int ntap = par == 0 ? 3 : 2;
        g.nseg = 2;
        g.seg[0] = Seg{0, 1, par == 0 ? -Co : 0, ntap * Co, 0};   // kw = 0 : frame t+1
        g.seg[1] = Seg{0, 0, par == 0 ? -Co : 0, ntap * Co, 0};   // kw = 1 : frame t
        g.N = Ci;
        Builder::layout_segs(g);
        const Builder::Coef cf = enc[i].coef[0];
        Builder::Coef coef = [=](int nn, int sg, int j) -> int32_t {
          const int kw = sg, jj = j / Co, co = j % Co;
          const int kh = par == 0 ? 4 - 2 * jj : 3 - 2 * jj;
          return cf(co, kw, kh * Ci + nn);
        };
        b.pack_weights(R, g, coef, nm + ".dg" + std::to_string(par), 100 + i);
        g.y = d_encz[i - 1]; g.y_bstride = (int64_t)T * Fi * Ci; g.y_tstride = Fi * Ci; g.y_fstride = 2 * Ci; g.y_off = par * Ci;
        bnb_attach(g, bnb_enc[i - 1], (int64_t)T * Fi * Ci, Fi * Ci, 2 * Ci, par * Ci);      // rows of encoder layer i-1's output, this phase's bins
        b.push(R, OP_RUNGEMM, 100 + i).g = g;
      }
    }
    b.finish_unpack(R);
  }

  finalize_rungemms(b, P);
  for (size_t k = 0; k < P->bwd.size(); ++k)
    if (P->bwd[k].kind == OP_UNPACK && P->bwd[k].tag == 998) P->bucket_op = (int32_t)k;
  P->arena_bytes[A_WS] = b.ws_off;
  P->arena_bytes[A_PARAM] = nparam * 4;
  P->arena_bytes[A_GRAD] = nparam * 4;
  P->arena_bytes[A_STATE] = std::max<int64_t>(nstate, 1) * 4;
  P->arena_bytes[A_CONST] = (int64_t)P->consts.size();
  P->arena_bytes[A_IO] = b.io_off;
  return P;
}

// =================================================================================================================
// CRN (reference models.py:329-565): the real twin of DCCRN on magnitudes.  Same kernels, different planner:
// real convs with half the channels (C_in = 1 magnitudes), plain [prev, skip] concat, ONE-layer nn.LSTM + `tranform`
// Linear, mask = tanh(out) * |spec| re-attached to the noisy phase.  I/O: wav, tgt -> out_wav, out_real (= est_mags),
// out_imag (= target_mags).  Gradients flow from out_wav only (est_mags / target_mags feed nothing the reference can
// reach: CRN + perceptual crashes in the reference).
Plan* build_crn_plan(const ModelConfig& cfg) {
  Plan* P = new Plan();
  P->cfg = cfg;
  Builder b;
  b.P = P;
  b.c = cfg;
  const int n = cfg.n_layers;
  const int B = cfg.B, L = cfg.L, W = cfg.win_len, hop = cfg.hop, NFFT = cfg.fft_len;
  const int trim = W - hop;
  const int T = (L + 2 * trim - W) / hop + 1;
  const int NF = NFFT / 2 + 1, NS = NF + 1, SW = NS * 2;
  const int Lp = (T - 1) * hop + W;
  const int adt = cfg.act_dtype;
  const int KS = cfg.kernel_size;
  const int MS = NF + 7, MO = 7;            // magnitude rows: bin k at element k + 7 -> bin 1 is 16-byte aligned in fp32 and bf16
  P->T = T;
  P->NF = NF;
  // models.py:506-532: 'Direct(None make)' (mask_mode 4) = spectral mapping, every other cfg.masking_mode = the tanh magnitude mask
  if (KS != 5 || n < 1 || n > 7) { P->error = "CRN: unsupported configuration"; return P; }
  const bool direct = cfg.mask_mode == 4;
  std::vector<int> ch(n + 1), Fe(n + 1);
  ch[0] = 1;
  for (int i = 0; i < n; ++i) ch[i + 1] = cfg.kernel_num[i] / 2;
  Fe[0] = NF - 1;
  for (int i = 0; i < n; ++i) Fe[i + 1] = Fe[i] / 2;
  const int D = Fe[n], Cl = ch[n];
  const int H = cfg.rnn_units / 2;
  const int hid = D * Cl;                    // must equal cfg.rnn_input_size
  for (int i = 1; i <= n; ++i)
    if (ch[i] % 8 != 0) { P->error = "channel counts must be multiples of 8"; return P; }
  if (H % 16 != 0 || H > 128 || (adt == DT_BF16 && H % 32 != 0)) { P->error = "rnn_units/2 must be a multiple of 16 (32 for bf16) and <= 128"; return P; }
  if (Fe[n] < 1 || (Fe[0] % (1 << n)) != 0) { P->error = "fft_len/2 must be divisible by 2^n_layers"; return P; }

  for (int i = 0; i < n; ++i) {
    const std::string p = "encoder." + std::to_string(i);
    b.add_param(p + ".0.conv.weight", {ch[i + 1], ch[i], KS, 2}, true);
    b.add_param(p + ".0.conv.bias", {ch[i + 1]}, true);
    b.add_param(p + ".1.weight", {ch[i + 1]}, true);
    b.add_param(p + ".1.bias", {ch[i + 1]}, true);
    b.add_param(p + ".1.running_mean", {ch[i + 1]}, false);
    b.add_param(p + ".1.running_var", {ch[i + 1]}, false);
    b.add_param(p + ".2.weight", {1}, true);
  }
  for (int d = 0; d < n; ++d) {
    const int idx = n - d;
    const int cin = ch[idx] * (cfg.skip ? 2 : 1), cout = ch[idx - 1];
    const std::string p = "decoder." + std::to_string(d);
    b.add_param(p + ".0.conv.weight", {cin, cout, KS, 2}, true);
    b.add_param(p + ".0.conv.bias", {cout}, true);
    if (idx != 1) {
      b.add_param(p + ".1.weight", {cout}, true);
      b.add_param(p + ".1.bias", {cout}, true);
      b.add_param(p + ".1.running_mean", {cout}, false);
      b.add_param(p + ".1.running_var", {cout}, false);
      b.add_param(p + ".2.weight", {1}, true);
    }
  }
  b.add_param("enhance.weight_ih_l0", {4 * H, hid}, true);
  b.add_param("enhance.weight_hh_l0", {4 * H, H}, true);
  b.add_param("enhance.bias_ih_l0", {4 * H}, true);
  b.add_param("enhance.bias_hh_l0", {4 * H}, true);
  b.add_param("tranform.weight", {hid, H}, true);
  b.add_param("tranform.bias", {hid}, true);
  const int64_t nparam = P->params.back().off + P->params.back().numel;
  const int64_t nstate = P->state.empty() ? 0 : P->state.back().off + P->state.back().numel;
  b.inv.resize(nparam);

  Ptr io_wav = b.io("wav", (int64_t)B * L);
  Ptr io_out = b.io("out_wav", (int64_t)B * L);
  Ptr io_or = b.io("out_real", (int64_t)B * NF * T);      // est_mags
  Ptr io_oi = b.io("out_imag", (int64_t)B * NF * T);      // target_mags
  Ptr io_gw = b.io("grad_wav", (int64_t)B * L);
  Ptr io_gr = b.io("grad_real", (int64_t)B * NF * T);     // gradient w.r.t. est_mags (crn_direct_train's loss lives there)
  b.io("grad_imag", (int64_t)B * NF * T);
  Ptr io_tgt = b.io("tgt", (int64_t)B * L);

  std::vector<double> win(W);
  for (int j = 0; j < W; ++j) win[j] = window_value(cfg, j, W);    // win_type None: np.ones (tools_for_model.py:17-18)
  auto Kun = [&](int part, int k, int j) {
    const double ang = 2.0 * kPi * (double)(((int64_t)k * j) % NFFT) / NFFT;
    return part == 0 ? std::cos(ang) : -std::sin(ang);
  };
  std::vector<double> Kinv((size_t)2 * NF * W);
  {
    const double ne = (W + 1) / 2, no = W / 2;
    for (int part = 0; part < 2; ++part)
      for (int k = 0; k < NF; ++k) {
        double se = 0, so = 0;
        for (int m = 0; m < W; ++m) (m % 2 == 0 ? se : so) += Kun(part, k, m);
        for (int j = 0; j < W; ++j) {
          const double corr = (j % 2 == 0) ? se / (NFFT / 2.0 + ne) : so / (NFFT / 2.0 + no);
          Kinv[((size_t)part * NF + k) * W + j] = (Kun(part, k, j) - corr) / (NFFT / 2.0) * win[j];
        }
      }
  }
  std::vector<float> coff(Lp, 0.f);
  {
    std::vector<float> w2(W);
    for (int j = 0; j < W; ++j) { const float wf = (float)win[j]; w2[j] = wf * wf; }
    for (int t = 0; t < T; ++t)
      for (int j = 0; j < W; ++j) coff[t * hop + j] += w2[j];
  }
  Ptr c_coff = b.cst(coff.data(), (int64_t)coff.size() * 4);
  auto const_weights = [&](RunGemm& g, const std::function<double(int n, int j)>& val) {
    std::vector<float> wt((size_t)g.Npad * g.ldw, 0.f);
    for (int nn = 0; nn < g.N; ++nn)
      for (int j = 0; j < g.seg[0].len; ++j) wt[(size_t)nn * g.ldw + j] = (float)val(nn, j);
    g.w = b.cst(wt.data(), (int64_t)wt.size() * 4);
  };
  std::vector<Op>& F = P->fwd;
  std::vector<Op>& R = P->bwd;
  const int64_t BT = (int64_t)B * T;

  // ---- STFT of the noisy input and of the target (CRN.forward always does both, models.py:468, 505)
  Ptr spec = b.ws("spec", BT * SW, DT_F32);
  Ptr spec_t = b.ws("spec_t", BT * SW, DT_F32);
  if (b.stft_fft(F, 1, io_wav, spec, B, L, T, hop, trim, NFFT, win)) {
    b.stft_fft(F, 2, io_tgt, spec_t, B, L, T, hop, trim, NFFT, win);
  } else {
    RunGemm g = Builder::gemm0();
    g.x[0] = io_wav; g.xdt = DT_F32; g.ydt = DT_F32;
    g.bstride[0] = L; g.rowlen[0] = L; g.fstride[0] = hop; g.Tin[0] = 1;
    g.M = (int)BT; g.Tout = 1; g.Fo = T;
    g.nseg = 1; g.seg[0] = Seg{0, 0, -trim, W, 0};
    g.N = SW;
    Builder::layout_segs(g);
    const_weights(g, [&](int nn, int j) { return nn < 2 ? 0.0 : Kun(nn & 1, nn / 2 - 1, j) * win[j]; });
    g.y = spec; g.y_bstride = (int64_t)T * SW; g.y_fstride = SW;
    b.push(F, OP_RUNGEMM, 1).g = g;
    g.x[0] = io_tgt; g.y = spec_t;
    b.push(F, OP_RUNGEMM, 2).g = g;
  }
  Ptr mags = b.ws("mags", BT * MS, adt);
  {
    Op& op = b.push(F, OP_MAGS, 3);
    op.mags.spec = spec; op.mags.mags = mags; op.mags.frames = BT; op.mags.NF = NF; op.mags.MS = MS; op.mags.MO = MO; op.mags.dt = adt;
  }

  // ---- encoder (RealConv2d, tools_for_model.py:341-386)
  struct Layer { RunGemm f[2]; Builder::Coef coef[2]; std::function<void(int, int32_t*)> bias; int C, Fq; int64_t R; };
  std::vector<Layer> enc(n), dec(n);
  std::vector<Ptr> encz(n), ency(n), enc_mi(n);
  Ptr prev = mags;
  for (int i = 0; i < n; ++i) {
    const int Ci = ch[i], Co = ch[i + 1], Fi = Fe[i], Fo = Fe[i + 1];
    const std::string nm = "enc" + std::to_string(i);
    const std::string pp = "encoder." + std::to_string(i);
    const ParamInfo &Wc = b.par(pp + ".0.conv.weight"), &bc = b.par(pp + ".0.conv.bias");
    RunGemm g = Builder::gemm0();
    g.x[0] = prev; g.xdt = adt; g.ydt = adt;
    if (i == 0) { g.bstride[0] = (int64_t)T * MS; g.tstride[0] = MS; g.base[0] = MO + 1; }
    else { g.bstride[0] = (int64_t)T * Fi * Ci; g.tstride[0] = Fi * Ci; g.base[0] = 0; }
    g.rowlen[0] = Fi * Ci; g.fstride[0] = 2 * Ci; g.Tin[0] = T;
    g.M = B * T * Fo; g.Tout = T; g.Fo = Fo;
    g.nseg = 2;
    g.seg[0] = Seg{0, -1, -2 * Ci, KS * Ci, 0};
    g.seg[1] = Seg{0, 0, -2 * Ci, KS * Ci, 0};
    g.N = Co;
    Builder::layout_segs(g);
    Builder::Coef coef = [=](int nn, int s, int j) -> int32_t {
      const int kw = s, kh = j / Ci, ci = j % Ci;
      return pe(Wc, (((int64_t)nn * Ci + ci) * KS + kh) * 2 + kw, 1);
    };
    std::function<void(int, int32_t*)> bias = [=](int nn, int32_t* o) { o[0] = pe(bc, nn, 1); o[1] = 0; };
    b.pack_weights(F, g, coef, nm, 100 + i, &bias);
    const int64_t Rr = (int64_t)B * T * Fo;
    ency[i] = b.ws(nm + ".y", Rr * Co, adt);
    encz[i] = b.ws(nm + ".z", Rr * Co, adt);
    enc_mi[i] = b.ws(nm + ".mi", 2 * Co, DT_F32);
    const int nblk = (int)((g.M + kBM - 1) / kBM);
    Ptr part = b.ws(nm + ".stat", (int64_t)nblk * 2 * g.Npad, DT_F32);
    g.y = ency[i]; g.y_bstride = (int64_t)T * Fo * Co; g.y_tstride = Fo * Co; g.y_fstride = Co; g.y_off = 0;
    g.stats = cfg.training ? part : b.none();
    b.push(F, OP_RUNGEMM, 100 + i).g = g;
    {
      Op& op = b.push(F, OP_BN_FINALIZE, 100 + i);
      op.bnf.part = part; op.bnf.mean_invstd = enc_mi[i];
      op.bnf.running_mean = b.sptr(pp + ".1.running_mean"); op.bnf.running_var = b.sptr(pp + ".1.running_var");
      op.bnf.nblk = cfg.training ? nblk : -1; op.bnf.C = Co; op.bnf.Cpad = g.Npad; op.bnf.count = (double)Rr;
      op.bnf.eps = 1e-5f; op.bnf.momentum = 0.1f;
      Op& oa = b.push(F, OP_BN_APPLY, 100 + i);
      oa.bna.y = ency[i]; oa.bna.z = encz[i]; oa.bna.mean_invstd = enc_mi[i];
      oa.bna.gamma = b.pptr(pp + ".1.weight"); oa.bna.beta = b.pptr(pp + ".1.bias"); oa.bna.slope = b.pptr(pp + ".2.weight");
      oa.bna.R = Rr; oa.bna.C = Co; oa.bna.dt = adt;
    }
    enc[i].f[0] = g; enc[i].coef[0] = coef; enc[i].bias = bias; enc[i].C = Co; enc[i].Fq = Fo; enc[i].R = Rr;
    prev = encz[i];
  }

  // ---- single-layer LSTM + Linear (models.py:391-398, 483-486); feature order c*D + d
  const ParamInfo &Wih = b.par("enhance.weight_ih_l0"), &Whh = b.par("enhance.weight_hh_l0");
  const ParamInfo &bih = b.par("enhance.bias_ih_l0"), &bhh = b.par("enhance.bias_hh_l0");
  Ptr gxb = b.ws("lstm.gx", BT * 4 * H, DT_F32);
  Ptr hbuf = b.ws("lstm.h", BT * H, adt);
  Ptr gatesb = b.ws("lstm.gates", BT * 4 * H, DT_F32);
  Ptr cbuf = b.ws("lstm.c", BT * H, DT_F32);
  RunGemm ggx = Builder::gemm0();
  Builder::Coef cgx;
  std::function<void(int, int32_t*)> bgx = [=](int nn, int32_t* o) { o[0] = pe(bih, gate_torch_row(nn, H), 1); o[1] = pe(bhh, gate_torch_row(nn, H), 1); };
  {
    RunGemm& g = ggx;
    g.x[0] = encz[n - 1]; g.xdt = adt; g.ydt = DT_F32;
    g.bstride[0] = (int64_t)T * D * Cl; g.tstride[0] = D * Cl; g.rowlen[0] = D * Cl; g.Tin[0] = T;
    g.M = (int)BT; g.Tout = T; g.Fo = 1;
    g.nseg = D;
    for (int dd = 0; dd < D; ++dd) g.seg[dd] = Seg{0, 0, dd * Cl, Cl, 0};
    g.N = 4 * H;
    Builder::layout_segs(g);
    cgx = [=](int nn, int s, int j) -> int32_t { return pe(Wih, (int64_t)gate_torch_row(nn, H) * hid + (j * D + s), 1); };
    b.pack_weights(F, g, cgx, "lstm.ih", 200, &bgx);
    g.y = gxb; g.y_bstride = (int64_t)T * 4 * H; g.y_tstride = 4 * H;
    b.push(F, OP_RUNGEMM, 200).g = g;
  }
  auto lstm_desc = [&](LstmRec& r) {
    std::memset(&r, 0, sizeof(r));
    r.gx = gxb; r.whh[0] = r.whh[1] = b.pptr("enhance.weight_hh_l0");
    r.h = hbuf; r.gates = gatesb; r.c = cbuf; r.dh = r.dgates = b.none();
    r.gx_ld = 4 * H; r.G = 1; r.nset = 1; r.B = B; r.T = T; r.H = H; r.hdt = adt; r.gdt = adt;
  };
  lstm_desc(b.push(F, OP_LSTM_FWD, 200).lstm);
  Ptr decin = b.ws("decin", BT * D * Cl, adt);
  RunGemm proj = Builder::gemm0();
  Builder::Coef cproj;
  std::function<void(int, int32_t*)> bproj;
  {
    const ParamInfo &Wt = b.par("tranform.weight"), &bt = b.par("tranform.bias");
    RunGemm& g = proj;
    g.x[0] = hbuf; g.xdt = adt; g.ydt = adt;
    g.bstride[0] = (int64_t)T * H; g.tstride[0] = H; g.rowlen[0] = H; g.Tin[0] = T;
    g.M = (int)BT; g.Tout = T; g.Fo = 1;
    g.nseg = 1; g.seg[0] = Seg{0, 0, 0, H, 0};
    g.N = D * Cl;
    Builder::layout_segs(g);
    cproj = [=](int nn, int s, int j) -> int32_t { const int dd = nn / Cl, cc = nn % Cl; return pe(Wt, (int64_t)(cc * D + dd) * H + j, 1); };
    bproj = [=](int nn, int32_t* o) { const int dd = nn / Cl, cc = nn % Cl; o[0] = pe(bt, cc * D + dd, 1); o[1] = 0; };
    b.pack_weights(F, g, cproj, "proj", 300, &bproj);
    g.y = decin; g.y_bstride = (int64_t)T * D * Cl; g.y_tstride = D * Cl;
    b.push(F, OP_RUNGEMM, 300).g = g;
  }

  // ---- decoder (RealConvTranspose2d, tools_for_model.py:389-425; torch.cat([out, enc], 1) skips)
  std::vector<Ptr> decy(n), decz(n), dec_mi(n);
  struct DecSrc { Ptr p; int64_t bstride; int tstride, base, C; };
  DecSrc dprev{decin, (int64_t)T * D * Cl, D * Cl, 0, Cl};
  for (int d = 0; d < n; ++d) {
    const int idx = n - d;
    const int C0 = ch[idx], C1 = cfg.skip ? ch[idx] : 0, Co = ch[idx - 1];
    const int Fi = Fe[idx], Fo = 2 * Fi;
    const bool last = (idx == 1);
    const std::string nm = "dec" + std::to_string(d);
    const std::string pp = "decoder." + std::to_string(d);
    const ParamInfo &Wc = b.par(pp + ".0.conv.weight"), &bc = b.par(pp + ".0.conv.bias");
    const int64_t Rr = (int64_t)B * (T + 1) * Fo;
    decy[d] = b.ws(nm + ".y", Rr * Co, adt);
    if (!last) { decz[d] = b.ws(nm + ".z", Rr * Co, adt); dec_mi[d] = b.ws(nm + ".mi", 2 * Co, DT_F32); }
    std::function<int32_t(int, int, int, int, int)> wcoef = [=](int nn, int s, int cc, int kh, int kw) -> int32_t {
      const int rc = s == 0 ? cc : C0 + cc;
      return pe(Wc, (((int64_t)rc * Co + nn) * KS + kh) * 2 + kw, 1);
    };
    std::function<void(int, int32_t*)> bias = [=](int nn, int32_t* o) { o[0] = pe(bc, nn, 1); o[1] = 0; };
    const int nblk1 = (int)(((int64_t)B * (T + 1) * Fi + kBM - 1) / kBM);
    Ptr part = b.none();
    const int npad_stat = (int)rup(Co, bn_of(Co));
    if (!last) part = b.ws(nm + ".stat", (int64_t)2 * nblk1 * 2 * npad_stat, DT_F32);
    for (int par = 0; par < 2; ++par) {
      RunGemm g = Builder::gemm0();
      g.xdt = adt; g.ydt = adt;
      const int nsrc = cfg.skip ? 2 : 1;
      DecSrc src[2] = {dprev, DecSrc{encz[idx - 1], (int64_t)T * Fi * C1, Fi * C1, 0, C1}};
      g.nseg = 0;
      const int ntap = par == 0 ? 3 : 2;
      for (int s = 0; s < nsrc; ++s) {
        g.x[s] = src[s].p; g.bstride[s] = src[s].bstride; g.tstride[s] = src[s].tstride; g.base[s] = src[s].base;
        g.rowlen[s] = Fi * src[s].C; g.fstride[s] = src[s].C; g.Tin[s] = T;
        for (int kw = 0; kw < 2; ++kw) g.seg[g.nseg++] = Seg{s, -kw, par == 0 ? -src[s].C : 0, ntap * src[s].C, 0};
      }
      g.M = B * (T + 1) * Fi; g.Tout = T + 1; g.Fo = Fi;
      g.N = Co;
      Builder::layout_segs(g);
      const int c0 = C0, c1 = C1;
      Builder::Coef coef = [=](int nn, int sg, int j) -> int32_t {
        const int s = sg / 2, kw = sg % 2;
        const int Cs = s == 0 ? c0 : c1;
        const int jj = j / Cs, cc = j % Cs;
        return wcoef(nn, s, cc, par == 0 ? 4 - 2 * jj : 3 - 2 * jj, kw);
      };
      b.pack_weights(F, g, coef, nm + ".p" + std::to_string(par), 400 + d, par == 0 ? &bias : nullptr);
      if (par == 1) g.bias = dec[d].f[0].bias;
      g.y = decy[d]; g.y_bstride = (int64_t)(T + 1) * Fo * Co; g.y_tstride = Fo * Co; g.y_fstride = 2 * Co; g.y_off = par * Co;
      if (!last && cfg.training) g.stats = b.mk(A_WS, part.off + (int64_t)par * nblk1 * 2 * g.Npad * 4);
      b.push(F, OP_RUNGEMM, 400 + d).g = g;
      dec[d].f[par] = g; dec[d].coef[par] = coef;
    }
    dec[d].bias = bias; dec[d].C = Co; dec[d].Fq = Fo; dec[d].R = Rr;
    if (!last) {
      Op& op = b.push(F, OP_BN_FINALIZE, 400 + d);
      op.bnf.part = part; op.bnf.mean_invstd = dec_mi[d];
      op.bnf.running_mean = b.sptr(pp + ".1.running_mean"); op.bnf.running_var = b.sptr(pp + ".1.running_var");
      op.bnf.nblk = cfg.training ? 2 * nblk1 : -1; op.bnf.C = Co; op.bnf.Cpad = npad_stat; op.bnf.count = (double)Rr;
      op.bnf.eps = 1e-5f; op.bnf.momentum = 0.1f;
      Op& oa = b.push(F, OP_BN_APPLY, 400 + d);
      oa.bna.y = decy[d]; oa.bna.z = decz[d]; oa.bna.mean_invstd = dec_mi[d];
      oa.bna.gamma = b.pptr(pp + ".1.weight"); oa.bna.beta = b.pptr(pp + ".1.bias"); oa.bna.slope = b.pptr(pp + ".2.weight");
      oa.bna.R = Rr; oa.bna.C = Co; oa.bna.dt = adt;
      dprev = DecSrc{decz[d], (int64_t)(T + 1) * Fo * Co, Fo * Co, Fo * Co, Co};
    }
  }

  // ---- mask, iSTFT, outputs (models.py:519-532)
  Ptr est = b.ws("est", BT * SW, DT_F32);
  Ptr estm = b.ws("estm", BT * NF, DT_F32);
  Ptr frames = b.ws("frames", BT * W, DT_F32);
  Mask mk;
  std::memset(&mk, 0, sizeof(mk));
  {
    const int Fo = Fe[0];
    mk.spec = spec; mk.mask = decy[n - 1]; mk.est = est; mk.estm = estm; mk.dest = mk.dmask = mk.destm = b.none();
    mk.frames = BT; mk.NF = NF; mk.mode = direct ? 5 : 3; mk.mdt = adt; mk.mch = 1;
    mk.mask_fstride = Fo; mk.mask_bstride = (int64_t)(T + 1) * Fo; mk.mask_base = Fo; mk.T = T;
    b.push(F, OP_MASK_FWD, 500).mask = mk;
  }
  if (!b.istft_fft(F, 501, est, frames, BT, NFFT, win)) {
    RunGemm g = Builder::gemm0();
    g.x[0] = est; g.xdt = DT_F32; g.ydt = DT_F32;
    g.bstride[0] = (int64_t)T * SW; g.tstride[0] = SW; g.rowlen[0] = SW; g.Tin[0] = T;
    g.M = (int)BT; g.Tout = T; g.Fo = 1;
    g.nseg = 1; g.seg[0] = Seg{0, 0, 0, SW, 0};
    g.N = W;
    Builder::layout_segs(g);
    const_weights(g, [&](int nn, int j) { return j < 2 ? 0.0 : Kinv[((size_t)(j & 1) * NF + (j / 2 - 1)) * W + nn]; });
    g.y = frames; g.y_bstride = (int64_t)T * W; g.y_tstride = W;
    b.push(F, OP_RUNGEMM, 501).g = g;
  }
  Ola ola;
  std::memset(&ola, 0, sizeof(ola));
  ola.frames = frames; ola.wav = io_out; ola.coff = c_coff; ola.dwav = ola.dpad = b.none();
  ola.B = B; ola.T = T; ola.L = L; ola.win = W; ola.hop = hop; ola.trim = trim;
  b.push(F, OP_OLA_FWD, 502).ola = ola;
  {
    SpecOut so;
    std::memset(&so, 0, sizeof(so));
    so.est = estm; so.out_real = io_or; so.out_imag = b.none(); so.B = B; so.T = T; so.NF = NF; so.mode = 2;
    b.push(F, OP_SPECOUT_FWD, 503).so = so;
    so.est = spec_t; so.out_real = io_oi; so.mode = 1;
    b.push(F, OP_SPECOUT_FWD, 504).so = so;
  }

  // =================================================================================================== backward
  if (cfg.training) {
    Ptr dpad = b.ws("dpad", (int64_t)B * Lp, DT_F32);
    Ptr dest = b.ws("dest", BT * SW, DT_F32);
    {
      Ola o = ola;
      o.dwav = io_gw; o.dpad = dpad;
      b.push(R, OP_OLA_BWD, 502).ola = o;
      if (!b.istft_bwd_fft(R, 501, dpad, dest, B, Lp, T, hop, NFFT, win)) {
        RunGemm g = Builder::gemm0();
        g.x[0] = dpad; g.xdt = DT_F32; g.ydt = DT_F32;
        g.bstride[0] = Lp; g.rowlen[0] = Lp; g.fstride[0] = hop; g.Tin[0] = 1;
        g.M = (int)BT; g.Tout = 1; g.Fo = T;
        g.nseg = 1; g.seg[0] = Seg{0, 0, 0, W, 0};
        g.N = SW;
        Builder::layout_segs(g);
        const_weights(g, [&](int nn, int j) { return nn < 2 ? 0.0 : Kinv[((size_t)(nn & 1) * NF + (nn / 2 - 1)) * W + j]; });
        g.y = dest; g.y_bstride = (int64_t)T * SW; g.y_fstride = SW;
        b.push(R, OP_RUNGEMM, 501).g = g;
      }
    }
    std::vector<Ptr> d_decy(n), d_decz(n), d_skip(n), d_encz(n), d_ency(n);
    for (int d = 0; d < n; ++d) {
      const int idx = n - d;
      const int Co = ch[idx - 1], Fo = 2 * Fe[idx];
      d_decy[d] = b.ws("dec" + std::to_string(d) + ".dy", (int64_t)B * (T + 1) * Fo * Co, adt);
      if (idx != 1) d_decz[d] = b.ws("dec" + std::to_string(d) + ".dz", (int64_t)B * T * Fo * Co, adt);
    }
    for (int i = 0; i < n; ++i) {
      const int64_t e = (int64_t)B * T * Fe[i + 1] * ch[i + 1];
      d_ency[i] = b.ws("enc" + std::to_string(i) + ".dy", e, adt);
      d_encz[i] = b.ws("enc" + std::to_string(i) + ".dz", e, adt);
      if (cfg.skip) d_skip[i] = b.ws("enc" + std::to_string(i) + ".dskip", e, adt);
    }
    Ptr d_decin = b.ws("decin.d", BT * D * Cl, adt);
    {
      Ptr d_estm = b.ws("destm", BT * NF, DT_F32);       // io.grad_real [B][NF][T] -> [B*T][NF]
      SpecOut s2;
      std::memset(&s2, 0, sizeof(s2));
      s2.est = d_estm; s2.out_real = io_gr; s2.out_imag = b.none(); s2.B = B; s2.T = T; s2.NF = NF; s2.mode = 2;
      b.push(R, OP_SPECOUT_BWD, 503).so = s2;
      Mask m2 = mk;
      m2.dest = dest; m2.dmask = d_decy[n - 1]; m2.destm = d_estm;
      b.push(R, OP_MASK_BWD, 500).mask = m2;
    }
    auto bn_bwd = [&](int tag, Ptr y, Ptr dz0, Ptr dz1, Ptr mi, const std::string& pp, int C, int64_t Rr, int64_t rpb, int skip, Ptr dy,
                      const std::string& nm) {
      int64_t rpbk = std::max<int64_t>(64, (Rr + 2047) / 2048);
      const int nblk = (int)((Rr + rpbk - 1) / rpbk);
      BnBwdReduce r;
      std::memset(&r, 0, sizeof(r));
      r.y = y; r.dz0 = dz0; r.dz1 = dz1; r.mean_invstd = mi;
      r.gamma = b.pptr(pp + ".1.weight"); r.beta = b.pptr(pp + ".1.bias"); r.slope = b.pptr(pp + ".2.weight");
      r.part = b.ws(nm + ".bnpart", (int64_t)nblk * 3 * C, DT_F32);
      r.R = Rr; r.C = C; r.dt = adt; r.nblk = nblk; r.rows_per_blk = (int)rpbk; r.rpb = rpb; r.skip = skip;
      b.push(R, OP_BN_BWD_REDUCE, tag).bnr = r;
      BnBwdApply a;
      std::memset(&a, 0, sizeof(a));
      a.r = r; a.totals = b.ws(nm + ".bntot", 3 * C, DT_F32); a.dy = dy;
      a.dgamma = b.pptr(pp + ".1.weight", A_GRAD); a.dbeta = b.pptr(pp + ".1.bias", A_GRAD); a.dslope = b.pptr(pp + ".2.weight", A_GRAD);
      a.count = (double)Rr;
      b.push(R, OP_BN_BWD_FINALIZE, tag).bnb = a;
      b.push(R, OP_BN_BWD_APPLY, tag).bnb = a;
    };
    for (int d = n - 1; d >= 0; --d) {
      const int idx = n - d;
      const int C0 = ch[idx], C1 = cfg.skip ? ch[idx] : 0, Co = ch[idx - 1];
      const int Fi = Fe[idx], Fo = 2 * Fi;
      const bool last = (idx == 1);
      const std::string nm = "dec" + std::to_string(d);
      const std::string pp = "decoder." + std::to_string(d);
      if (!last)
        bn_bwd(400 + d, decy[d], d_decz[d], b.none(), dec_mi[d], pp, Co, dec[d].R, (int64_t)(T + 1) * Fo, Fo, d_decy[d], nm);
      b.cur_lane = 1;                           // weight gradients of the decoder: nothing downstream needs them before UNPACK
      for (int par = 0; par < 2; ++par) b.wgrad(R, dec[d].f[par], d_decy[d], dec[d].coef[par], 400 + d, &dec[d].bias);
      b.cur_lane = 0;
      const int nsrc = cfg.skip ? 2 : 1;
      for (int s = 0; s < nsrc; ++s) {
        const int Cs = s == 0 ? C0 : C1;
        RunGemm g = Builder::gemm0();
        g.x[0] = d_decy[d]; g.xdt = adt; g.ydt = adt;
        g.bstride[0] = (int64_t)(T + 1) * Fo * Co; g.tstride[0] = Fo * Co; g.rowlen[0] = Fo * Co; g.fstride[0] = 2 * Co; g.Tin[0] = T + 1;
        g.M = B * T * Fi; g.Tout = T; g.Fo = Fi;
        g.nseg = 2;
        g.seg[0] = Seg{0, 0, -2 * Co, KS * Co, 0};
        g.seg[1] = Seg{0, 1, -2 * Co, KS * Co, 0};
        g.N = Cs;
        Builder::layout_segs(g);
        const Builder::Coef f0 = dec[d].coef[0], f1 = dec[d].coef[1];
        Builder::Coef coef = [=](int nn, int sg, int j) -> int32_t {
          const int kw = sg, kh = j / Co, co = j % Co;
          const int par = kh & 1;
          const int jj = par == 0 ? (4 - kh) / 2 : (3 - kh) / 2;
          return (par == 0 ? f0 : f1)(co, s * 2 + kw, jj * Cs + nn);
        };
        b.pack_weights(R, g, coef, nm + ".dg" + std::to_string(s), 400 + d);
        g.y = s == 0 ? (d > 0 ? d_decz[d - 1] : d_decin) : d_skip[idx - 1];
        g.y_bstride = (int64_t)T * Fi * Cs; g.y_tstride = Fi * Cs; g.y_fstride = Cs; g.y_off = 0;
        b.push(R, OP_RUNGEMM, 400 + d).g = g;
      }
    }
    // projection + LSTM backward
    Ptr dh = b.ws("lstm.dh", BT * H, DT_F32);
    {
      b.cur_lane = 1;
      b.wgrad(R, proj, d_decin, cproj, 300, &bproj);
      b.cur_lane = 0;
      RunGemm g = Builder::gemm0();
      g.x[0] = d_decin; g.xdt = adt; g.ydt = DT_F32;
      g.bstride[0] = (int64_t)T * D * Cl; g.tstride[0] = D * Cl; g.rowlen[0] = D * Cl; g.Tin[0] = T;
      g.M = (int)BT; g.Tout = T; g.Fo = 1;
      g.nseg = 1; g.seg[0] = Seg{0, 0, 0, D * Cl, 0};
      g.N = H;
      Builder::layout_segs(g);
      Builder::Coef coef = [=](int nn, int sg, int j) -> int32_t { return cproj(j, 0, nn); };
      b.pack_weights(R, g, coef, "proj.dg", 300);
      g.y = dh; g.y_bstride = (int64_t)T * H; g.y_tstride = H;
      b.push(R, OP_RUNGEMM, 300).g = g;
    }
    Ptr dgates = b.ws("lstm.dgates", BT * 4 * H, adt);
    {
      LstmRec& r = b.push(R, OP_LSTM_BWD, 200).lstm;
      lstm_desc(r);
      r.dh = dh; r.dgates = dgates;
    }
    {
      RunGemm fw = ggx;
      fw.ydt = adt;
      b.wgrad(R, fw, dgates, cgx, 200, &bgx);
      RunGemm f = Builder::gemm0();
      f.x[0] = hbuf; f.xdt = adt; f.ydt = adt;
      f.bstride[0] = (int64_t)T * H; f.tstride[0] = H; f.rowlen[0] = H; f.Tin[0] = T;
      f.M = (int)BT; f.Tout = T; f.Fo = 1;
      f.nseg = 1; f.seg[0] = Seg{0, -1, 0, H, 0};
      f.N = 4 * H;
      Builder::layout_segs(f);
      f.y_bstride = (int64_t)T * 4 * H; f.y_tstride = 4 * H;
      Builder::Coef chh = [=](int nn, int sg, int j) -> int32_t { return pe(Whh, (int64_t)gate_torch_row(nn, H) * H + j, 1); };
      b.wgrad(R, f, dgates, chh, 200, nullptr);
      for (int q = 0; q < D; ++q) {            // dX into the encoder-output gradient, one slice per frequency row d
        RunGemm g = Builder::gemm0();
        g.x[0] = dgates; g.xdt = adt; g.ydt = adt;
        g.bstride[0] = (int64_t)T * 4 * H; g.tstride[0] = 4 * H; g.rowlen[0] = 4 * H; g.Tin[0] = T;
        g.M = (int)BT; g.Tout = T; g.Fo = 1;
        g.nseg = 1; g.seg[0] = Seg{0, 0, 0, 4 * H, 0};
        g.N = Cl;
        Builder::layout_segs(g);
        const Builder::Coef cf = cgx;
        Builder::Coef coef = [=](int nn, int sg, int j) -> int32_t { return cf(j, q, nn); };
        b.pack_weights(R, g, coef, "lstm.dx" + std::to_string(q), 200);
        g.y = d_encz[n - 1]; g.y_bstride = (int64_t)T * D * Cl; g.y_tstride = D * Cl; g.y_off = q * Cl;
        b.push(R, OP_RUNGEMM, 200).g = g;
      }
    }
    for (int i = n - 1; i >= 0; --i) {
      const int Ci = ch[i], Co = ch[i + 1], Fi = Fe[i], Fo = Fe[i + 1];
      const std::string nm = "enc" + std::to_string(i);
      const std::string pp = "encoder." + std::to_string(i);
      bn_bwd(100 + i, ency[i], d_encz[i], cfg.skip ? d_skip[i] : b.none(), enc_mi[i], pp, Co, enc[i].R, (int64_t)T * Fo, 0, d_ency[i], nm);
      b.wgrad(R, enc[i].f[0], d_ency[i], enc[i].coef[0], 100 + i, &enc[i].bias);
      if (i == 0) continue;
      for (int par = 0; par < 2; ++par) {
        RunGemm g = Builder::gemm0();
        g.x[0] = d_ency[i]; g.xdt = adt; g.ydt = adt;
        g.bstride[0] = (int64_t)T * Fo * Co; g.tstride[0] = Fo * Co; g.rowlen[0] = Fo * Co; g.fstride[0] = Co; g.Tin[0] = T;
        g.M = B * T * Fo; g.Tout = T; g.Fo = Fo;
        const int ntap = par == 0 ? 3 : 2;
        g.nseg = 2;
        g.seg[0] = Seg{0, 1, par == 0 ? -Co : 0, ntap * Co, 0};
        g.seg[1] = Seg{0, 0, par == 0 ? -Co : 0, ntap * Co, 0};
        g.N = Ci;
        Builder::layout_segs(g);
        const Builder::Coef cf = enc[i].coef[0];
        Builder::Coef coef = [=](int nn, int sg, int j) -> int32_t {
          const int kw = sg, jj = j / Co, co = j % Co;
          return cf(co, kw, (par == 0 ? 4 - 2 * jj : 3 - 2 * jj) * Ci + nn);
        };
        b.pack_weights(R, g, coef, nm + ".dg" + std::to_string(par), 100 + i);
        g.y = d_encz[i - 1]; g.y_bstride = (int64_t)T * Fi * Ci; g.y_tstride = Fi * Ci; g.y_fstride = 2 * Ci; g.y_off = par * Ci;
        b.push(R, OP_RUNGEMM, 100 + i).g = g;
      }
    }
    b.finish_unpack(R);
  }
  finalize_rungemms(b, P);
  P->arena_bytes[A_WS] = b.ws_off;
  P->arena_bytes[A_PARAM] = nparam * 4;
  P->arena_bytes[A_GRAD] = nparam * 4;
  P->arena_bytes[A_STATE] = std::max<int64_t>(nstate, 1) * 4;
  P->arena_bytes[A_CONST] = (int64_t)P->consts.size();
  P->arena_bytes[A_IO] = b.io_off;
  return P;
}

// =================================================================================================================
// Front end only (model 2): ConvSTFT 'complex' of io.wav in the reference layout -> io.out_real / io.out_imag [B][NF][T].
// Used by DCCRN.loss for the clean spectrum of the LMS loss (models.py:306-309).
Plan* build_frontend_plan(const ModelConfig& cfg) {
  Plan* P = new Plan();
  P->cfg = cfg;
  Builder b;
  b.P = P;
  b.c = cfg;
  const int B = cfg.B, L = cfg.L, W = cfg.win_len, hop = cfg.hop, NFFT = cfg.fft_len;
  const int trim = W - hop;
  const int T = (L + 2 * trim - W) / hop + 1;
  const int NF = NFFT / 2 + 1, NS = NF + 1, SW = NS * 2;
  P->T = T;
  P->NF = NF;
  Ptr io_wav = b.io("wav", (int64_t)B * L);
  Ptr io_or = b.io("out_real", (int64_t)B * NF * T);
  Ptr io_oi = b.io("out_imag", (int64_t)B * NF * T);
  std::vector<double> win(W);
  for (int j = 0; j < W; ++j) win[j] = window_value(cfg, j, W);    // win_type None: np.ones (tools_for_model.py:17-18)
  Ptr spec = b.ws("spec", (int64_t)B * T * SW, DT_F32);
  if (!b.stft_fft(P->fwd, 1, io_wav, spec, B, L, T, hop, trim, NFFT, win)) {
  RunGemm g = Builder::gemm0();
  g.x[0] = io_wav; g.xdt = DT_F32; g.ydt = DT_F32;
  g.bstride[0] = L; g.rowlen[0] = L; g.fstride[0] = hop; g.Tin[0] = 1;
  g.M = B * T; g.Tout = 1; g.Fo = T;
  g.nseg = 1; g.seg[0] = Seg{0, 0, -trim, W, 0};
  g.N = SW;
  Builder::layout_segs(g);
  {
    std::vector<float> wt((size_t)g.Npad * g.ldw, 0.f);
    for (int nn = 2; nn < g.N; ++nn)
      for (int j = 0; j < W; ++j) {
        const double ang = 2.0 * kPi * (double)(((int64_t)(nn / 2 - 1) * j) % NFFT) / NFFT;
        wt[(size_t)nn * g.ldw + j] = (float)(((nn & 1) == 0 ? std::cos(ang) : -std::sin(ang)) * win[j]);
      }
    g.w = b.cst(wt.data(), (int64_t)wt.size() * 4);
  }
  g.y = spec; g.y_bstride = (int64_t)T * SW; g.y_fstride = SW;
  b.push(P->fwd, OP_RUNGEMM, 1).g = g;
  }
  SpecOut so;
  std::memset(&so, 0, sizeof(so));
  so.est = spec; so.out_real = io_or; so.out_imag = io_oi; so.B = B; so.T = T; so.NF = NF;
  b.push(P->fwd, OP_SPECOUT_FWD, 2).so = so;
  finalize_rungemms(b, P);
  P->arena_bytes[A_WS] = b.ws_off;
  P->arena_bytes[A_PARAM] = 4; P->arena_bytes[A_GRAD] = 4; P->arena_bytes[A_STATE] = 4;
  P->arena_bytes[A_CONST] = (int64_t)P->consts.size();
  P->arena_bytes[A_IO] = b.io_off;
  return P;
}

// =================================================================================================================
// FullSubNet (reference models.py:568-682; SequenceModel tools_for_model.py:726-795).  model == 3.
// Config fields reused: kernel_num = {sb_num_neighbors, fb_num_neighbors(=0), look_ahead, fb_hidden, sb_hidden,
//                                     fb activation (0 none, 1 ReLU, 2 Tanh, 3 ReLU6), sb activation, dropout keep in 1/1000};
// T = frames of the input magnitude (passed in cfg.L as T, cfg.fft_len/2+1 = F).  I/O: io.mag [B][F][T] -> io.crm [B][F][T][2];
// backward: io.grad_crm -> A_GRAD.
// Every LSTM layer = one hoisted input GEMM over all T' steps + per step {recurrent GEMM accumulating onto the gate
// slab, cell kernel}; with B*257 = 8224 rows (B = 32) each step GEMM is a full-chip 8224 x 1536 x 384 problem.
Plan* build_fsn_plan(const ModelConfig& cfg) {
  Plan* P = new Plan();
  P->cfg = cfg;
  Builder b;
  b.P = P;
  b.c = cfg;
  const int B = cfg.B, T = cfg.L, F = cfg.fft_len / 2 + 1;
  const int nsb = cfg.kernel_num[0], nfb = cfg.kernel_num[1], LA = cfg.kernel_num[2];
  const int Hf = cfg.kernel_num[3], Hs = cfg.kernel_num[4], actf = cfg.kernel_num[5], acts = cfg.kernel_num[6];
  const float keep = cfg.training ? cfg.kernel_num[7] / 1000.f : 1.f;
  const bool gru = cfg.kernel_num[8] == 1;        // cfg.sequence_model: nn.GRU instead of nn.LSTM (tools_for_model.py:739-756)
  const int nmode = cfg.kernel_num[9];            // cfg.norm_type (sefd_desc.h struct Fsn): 0 offline_laplace ... 3 cumulative_layer_norm
  if (nmode < 0 || nmode > 3) { P->error = "FullSubNet: unknown norm_type"; return P; }
  const int NG = gru ? 3 : 4;                     // gate blocks of the recurrent weights
  const int adt = cfg.act_dtype;
  const int TP = T + LA, NB = 2 * nsb + 1, W = NB + 1;
  const int FP = (int)rup(F, 8);
  P->T = T;
  P->NF = F;
  if (nfb != 0 || acts != 0) { P->error = "FullSubNet: fb_num_neighbors must be 0 and the sub-band output activation None"; return P; }
  if (Hf % 8 || Hs % 8 || W % 8) { P->error = "FullSubNet: hidden sizes and sub-band width must be multiples of 8"; return P; }
  struct Net { std::string name; int I, H, O; };
  Net nets[2] = {{"fb_model", F, Hf, F}, {"sb_model", W, Hs, 2}};
  for (auto& nt : nets) {
    for (int l = 0; l < 2; ++l) {
      const std::string p = nt.name + ".sequence_model.";
      b.add_param(p + "weight_ih_l" + std::to_string(l), {NG * nt.H, l == 0 ? nt.I : nt.H}, true);
      b.add_param(p + "weight_hh_l" + std::to_string(l), {NG * nt.H, nt.H}, true);
      b.add_param(p + "bias_ih_l" + std::to_string(l), {NG * nt.H}, true);
      b.add_param(p + "bias_hh_l" + std::to_string(l), {NG * nt.H}, true);
    }
    b.add_param(nt.name + ".fc_output_layer.weight", {nt.O, nt.H}, true);
    b.add_param(nt.name + ".fc_output_layer.bias", {nt.O}, true);
  }
  const int64_t nparam = P->params.back().off + P->params.back().numel;
  b.inv.resize(nparam);
  Ptr io_mag = b.io("mag", (int64_t)B * F * T);
  Ptr io_crm = b.io("crm", (int64_t)B * F * T * 2);
  Ptr io_gcrm = b.io("grad_crm", (int64_t)B * F * T * 2);
  Ptr io_seed = b.io("seed", 2);
  std::vector<Op>& Fw = P->fwd;
  std::vector<Op>& R = P->bwd;

  auto fsn0 = [&]() { Fsn f; std::memset(&f, 0, sizeof(f)); f.in = f.out = f.aux = f.aux2 = f.sums = f.stat = b.none();
                      f.B = B; f.F = F; f.T = T; f.TP = TP; f.FP = FP; f.NB = NB; f.LA = LA; f.dt = adt; f.act = actf; return f; };
  // time-major GEMM over all steps: rows (t, r), source [TP][rows][feat]
  auto seq_gemm = [&](Ptr x, int xdt, int64_t rows, int feat, int off, int len, int N, int ydt) {
    RunGemm g = Builder::gemm0();
    g.x[0] = x; g.xdt = xdt; g.ydt = ydt;
    g.bstride[0] = 0; g.tstride[0] = (int)(rows * feat); g.base[0] = 0; g.rowlen[0] = (int)(rows * feat); g.fstride[0] = feat; g.Tin[0] = TP;
    g.M = (int)(TP * rows); g.Tout = TP; g.Fo = (int)rows;
    g.nseg = 1; g.seg[0] = Seg{0, 0, off, len, 0};
    g.N = N;
    Builder::layout_segs(g);
    return g;
  };
  auto set_y = [&](RunGemm& g, Ptr y, int64_t rows, int ld, int yoff) {
    g.y = y; g.y_bstride = 0; g.y_tstride = (int)(rows * ld); g.y_fstride = ld; g.y_off = yoff;
  };
  // one time step: rows (r), source slab [rows][feat] at step t
  auto step_gemm = [&](Ptr x, int xdt, int64_t rows, int feat, int t, int N, Ptr y, int ld, int64_t yslab_elems, int ydt, int flags) {
    RunGemm g = Builder::gemm0();
    g.x[0] = b.mk(A_WS, x.off + (int64_t)t * rows * feat * esize(xdt)); g.xdt = xdt; g.ydt = ydt;
    g.tstride[0] = 0; g.rowlen[0] = (int)(rows * feat); g.fstride[0] = feat; g.Tin[0] = 1;
    g.M = (int)rows; g.Tout = 1; g.Fo = (int)rows;
    g.nseg = 1; g.seg[0] = Seg{0, 0, 0, feat, 0};
    g.N = N;
    Builder::layout_segs(g);
    g.y = b.mk(A_WS, y.off + yslab_elems * esize(ydt)); g.y_fstride = ld; g.flags = flags;
    return g;
  };

  // ---- input: transpose, laplace norm (models.py:640-645)
  Ptr mag_t = b.ws("mag_t", (int64_t)TP * B * F, DT_F32);
  Ptr sum_fb = b.ws("sum_fb", (int64_t)B * F, DT_F32);
  Ptr mu_fb = b.ws("mu_fb", B, DT_F32);
  Ptr fb_in = b.ws("fb_in", (int64_t)TP * B * FP, adt);
  { Fsn f = fsn0(); f.in = io_mag; f.out = mag_t; f.sums = sum_fb; f.aux2 = mu_fb; b.push(Fw, OP_FSN_IN, 1).fsn = f; }
  Ptr st_fb = b.none(), st_sb = b.none();
  if (nmode == 2) { st_fb = b.ws("stat_fb", 2 * B, DT_F32); st_sb = b.ws("stat_sb", 2 * B, DT_F32); }
  else if (nmode) { st_fb = b.ws("stat_fb", (int64_t)TP * B * 2, DT_F32); st_sb = b.ws("stat_sb", (int64_t)TP * B * F * 2, DT_F32); }
  if (nmode) { Fsn f = fsn0(); f.in = mag_t; f.stat = st_fb; f.mode = nmode; f.src = 0; b.push(Fw, OP_FSN_NORMSTAT, 2).fsn = f; }
  { Fsn f = fsn0(); f.in = mag_t; f.out = fb_in; f.sums = mu_fb; f.mode = nmode; f.stat = st_fb; b.push(Fw, OP_FSN_SCALE, 2).fsn = f; }

  struct LayerRt { RunGemm gx; Builder::Coef cgx; std::function<void(int, int32_t*)> bgx; Ptr gates, c, h, hd, x; int xfeat, xlen, H; int64_t rows;
                   const ParamInfo* Whh; RunGemm rec; std::string nm; int lid; bool cluster, rowsk, xfuse, dropfused = false, dropbwd = false, headfuse = false; Ptr dyo, wo; int sdt, xf; int dhdt = DT_F32; Ptr hd_fused; Ptr gh, hzero; std::function<void(int, int32_t*)> bhh; };
  std::vector<LayerRt> layers;
  auto lstm_forward = [&](const std::string& netname, int l, int lid, Ptr x, int xfeat, int xlen, int64_t rows, int H, int tag) -> Ptr {
    LayerRt L;
    L.nm = netname + ".l" + std::to_string(l); L.lid = lid; L.x = x; L.xfeat = xfeat; L.xlen = xlen; L.rows = rows; L.H = H;
    const std::string pp = netname + ".sequence_model.";
    const ParamInfo &Wih = b.par(pp + "weight_ih_l" + std::to_string(l)), &Whh = b.par(pp + "weight_hh_l" + std::to_string(l));
    const ParamInfo &bih = b.par(pp + "bias_ih_l" + std::to_string(l)), &bhh = b.par(pp + "bias_hh_l" + std::to_string(l));
    L.Whh = &Whh;
    const int I = (int)Wih.shape[1];
    // thousands of rows (the sub-band model) in bf16: row-block kernels (lstm_rows.hip) on a packed bf16 copy of W_hh; their gate
    // slabs are bf16 too - at B * 257 rows those layers are bound by the HBM traffic of exactly these slabs (SEFD_LSTM_SLAB32=1: fp32)
    const int64_t rows_min = tune_str("LSTM_ROWS_MIN") ? atoll(tune_str("LSTM_ROWS_MIN")) : 1024;
    L.cluster = !gru && adt == DT_BF16 && H > 128 && H <= 512 && H % 64 == 0 && tune_str("LSTM_STEPPED") == nullptr;
    L.rowsk = L.cluster && rows >= rows_min && (H == 256 || H == 384 || H == 512);
    L.sdt = (L.rowsk && tune_str("LSTM_SLAB32") == nullptr) ? DT_BF16 : DT_F32;
    L.gates = b.ws(L.nm + ".gates", (int64_t)TP * rows * 4 * H, L.sdt);
    L.c = b.ws(L.nm + ".c", (int64_t)TP * rows * H, DT_F32);
    L.h = b.ws(L.nm + ".h", (int64_t)TP * rows * H, adt);
    // bf16 mode, 128 < H <= 512: the whole recurrence is ONE launch of the cluster kernels (lstm_cluster.hip) on the time-major
    // slabs, gate columns unit-major (sefd_desc.h gate_col); otherwise one GEMM + one cell launch per frame, gate-major columns
    const bool um = L.cluster;
    RunGemm g = seq_gemm(x, adt, rows, xfeat, 0, xlen, NG * H, L.sdt);
    L.cgx = [=](int nn, int s, int j) -> int32_t { return j < I ? pe(Wih, (int64_t)(um ? gate_torch_row(nn, H) : nn) * I + j, 1) : 0; };
    if (gru) L.bgx = [=](int nn, int32_t* o) { o[0] = pe(bih, nn, 1); o[1] = 0; };     // b_hh rides the recurrent GEMM: n = tanh(.. + r * (W_hn h + b_hn))
    else L.bgx = [=](int nn, int32_t* o) { const int q = um ? gate_torch_row(nn, H) : nn; o[0] = pe(bih, q, 1); o[1] = pe(bhh, q, 1); };
    b.pack_weights(Fw, g, L.cgx, L.nm + ".ih", tag, &L.bgx);
    set_y(g, L.gates, rows, 4 * H, 0);
    // row-block kernels, 32 input features (the sub-band model's first layer): the input projection is fused into the recurrence (one more
    // k-step per frame) instead of writing and re-reading a [T x rows x 4H] pre-activation slab (8 GB at B = 64); SEFD_LSTM_XFUSE=0 keeps the GEMM
    // ... and the layers above it (input = the layer below's h, H features): H/32 more k-steps per frame instead of an 8 GB slab + a GEMM
    const bool x32 = xlen == 32 && xfeat == 32 && g.ldw == 64, xh = xlen == H && xfeat == H && g.ldw == H;
    L.xfuse = L.rowsk && (x32 || xh) && g.Npad == 4 * H && !(tune_str("LSTM_XFUSE") && atoi(tune_str("LSTM_XFUSE")) == 0);
    if (L.xfuse) {
      // the packed W_ih re-ordered to MFMA B-fragment order ([4H][64] with K = 32 zero padded: in its first 4H x 32 slots)
      int32_t* tab = nullptr;
      for (auto it = Fw.rbegin(); it != Fw.rend(); ++it)
        if (it->kind == OP_PACK && it->pack.dst.arena == g.w.arena && it->pack.dst.off == g.w.off) { tab = reinterpret_cast<int32_t*>(P->consts.data() + it->pack.tab.off); break; }
      if (!tab) { P->error = "FullSubNet: packed W_ih not found"; return b.none(); }
      const int ldw = g.ldw, kf = x32 ? 32 : H;
      std::vector<int32_t> old(tab, tab + (size_t)4 * H * ldw);
      std::fill(tab, tab + (size_t)4 * H * ldw, 0);
      for (int c = 0; c < 4 * H; ++c)
        for (int k = 0; k < kf; ++k) tab[rows_wf_index(kf, c, k)] = old[(size_t)c * ldw + k];
      L.xf = kf;
    } else {
      b.push(Fw, OP_RUNGEMM, tag).g = g;
    }
    L.gx = g;
    if (L.cluster) {
      LstmRec r;
      std::memset(&r, 0, sizeof(r));
      r.gx = L.gates; r.gates = L.gates;                   // pre-activations are overwritten in place by i, f, g, o
      r.whh[0] = r.whh[1] = b.pptr(pp + "weight_hh_l" + std::to_string(l));
      r.h = L.h; r.c = L.c; r.dh = r.dgates = b.none();
      r.gx_ld = 4 * H; r.G = 1; r.nset = 1; r.B = (int)rows; r.T = TP; r.H = H; r.hdt = adt; r.gdt = DT_F32; r.tmajor = 1;
      // thousands of rows (the sub-band model): row-block kernels (lstm_rows.hip) on a packed bf16 copy of W_hh, rows = gate columns
      L.rec = Builder::gemm0();
      if (L.rowsk) {
        RunGemm pk = step_gemm(L.h, adt, rows, H, 0, 4 * H, L.gates, 4 * H, 0, DT_F32, 0);
        Builder::Coef chh = [=](int nn, int s, int j) -> int32_t { return pe(Whh, (int64_t)gate_torch_row(nn, H) * H + j, 1); };
        b.pack_weights(Fw, pk, chh, L.nm + ".hhpk", tag);
        if (pk.ldw != H || pk.Npad != 4 * H) { P->error = "FullSubNet: packed W_hh layout"; return b.none(); }
        {   // re-order the gather table into MFMA B-fragment order: one wave-load of the kernel = 1 KB contiguous (lstm_rows.hip)
          int32_t* tab = reinterpret_cast<int32_t*>(P->consts.data() + Fw.back().pack.tab.off);
          std::vector<int32_t> old(tab, tab + (size_t)4 * H * H);
          for (int c = 0; c < 4 * H; ++c)
            for (int k = 0; k < H; ++k) tab[rows_wf_index(H, c, k)] = old[(size_t)c * H + k];
        }
        r.impl = 1; r.wpk_f = pk.w; r.wpk_b = b.none(); r.gxdt = L.sdt;
        r.xin = r.wpk_x = r.bias = b.none();
        if (L.xfuse) { r.xin = x; r.wpk_x = g.w; r.bias = g.bias; r.xfeat = L.xf; }
        r.hd = r.seed = b.none();
        if (l == 0 && keep < 1.f && !(tune_str("LSTM_DROPFUSE") && atoi(tune_str("LSTM_DROPFUSE")) == 0)) {   // dropout applied while h_t is stored
          L.hd_fused = b.ws(L.nm + ".hd", (int64_t)TP * rows * H, adt);
          r.hd = L.hd_fused; r.seed = io_seed; r.keep = keep; r.drop_layer = lid;
          L.dropfused = true;
        }
      }
      b.push(Fw, OP_LSTM_FWD, tag).lstm = r;
    } else if (gru) {
      // per frame: gh = h_{t-1} . W_hh^T + b_hh into one reused [rows][3H] buffer (t = 0 reads a zero slab), then the GRU cell
      L.gh = b.ws(L.nm + ".gh", rows * 3 * H, DT_F32);
      L.hzero = b.ws(L.nm + ".h0", rows * H, adt);
      { Op& m = b.push(Fw, OP_MEMSET, tag); m.ms.dst = L.hzero; m.ms.bytes = rows * H * esize(adt); }
      RunGemm rec0 = step_gemm(L.hzero, adt, rows, H, 0, 3 * H, L.gh, 3 * H, 0, DT_F32, 0);
      Builder::Coef chh = [=](int nn, int s, int j) -> int32_t { return pe(Whh, (int64_t)nn * H + j, 1); };
      L.bhh = [=](int nn, int32_t* o) { o[0] = pe(bhh, nn, 1); o[1] = 0; };
      b.pack_weights(Fw, rec0, chh, L.nm + ".hh", tag, &L.bhh);
      L.rec = rec0;
      for (int t = 0; t < TP; ++t) {
        RunGemm r = rec0;
        if (t > 0) r.x[0] = b.mk(A_WS, L.h.off + (int64_t)(t - 1) * rows * H * esize(adt));
        b.push(Fw, OP_RUNGEMM, tag).g = r;
        LstmCell& cl = b.push(Fw, OP_CELL_FWD, tag).cell;
        std::memset(&cl, 0, sizeof(cl));
        cl.gates = b.mk(A_WS, L.gates.off + (int64_t)t * rows * 4 * H * 4);
        cl.gh = L.gh;
        cl.c = b.none();
        cl.c_prev = t > 0 ? b.mk(A_WS, L.h.off + (int64_t)(t - 1) * rows * H * esize(adt)) : b.none();
        cl.h = b.mk(A_WS, L.h.off + (int64_t)t * rows * H * esize(adt));
        cl.dh = cl.dc = cl.dgates = b.none();
        cl.rows = rows; cl.H = H; cl.hdt = adt; cl.gdt = adt; cl.first = t == 0; cl.kind = 1;
      }
    } else {
    // recurrent weights, packed once per step list
    RunGemm rec0 = step_gemm(L.h, adt, rows, H, 0, 4 * H, L.gates, 4 * H, 0, DT_F32, kRunAccum);
    Builder::Coef chh = [=](int nn, int s, int j) -> int32_t { return pe(Whh, (int64_t)nn * H + j, 1); };
    b.pack_weights(Fw, rec0, chh, L.nm + ".hh", tag);
    L.rec = rec0;
    for (int t = 0; t < TP; ++t) {
      if (t > 0) {
        RunGemm r = step_gemm(L.h, adt, rows, H, t - 1, 4 * H, L.gates, 4 * H, (int64_t)t * rows * 4 * H, DT_F32, kRunAccum);
        r.w = rec0.w;
        b.push(Fw, OP_RUNGEMM, tag).g = r;
      }
      Op& op = b.push(Fw, OP_CELL_FWD, tag);
      LstmCell& cl = op.cell;
      cl.kind = 0; cl.gh = b.none();
      cl.gates = b.mk(A_WS, L.gates.off + (int64_t)t * rows * 4 * H * 4);
      cl.c = b.mk(A_WS, L.c.off + (int64_t)t * rows * H * 4);
      cl.c_prev = t > 0 ? b.mk(A_WS, L.c.off + (int64_t)(t - 1) * rows * H * 4) : b.none();
      cl.h = b.mk(A_WS, L.h.off + (int64_t)t * rows * H * esize(adt));
      cl.dh = cl.dc = cl.dgates = b.none();
      cl.rows = rows; cl.H = H; cl.hdt = adt; cl.gdt = adt; cl.first = t == 0;
    }
    }
    L.hd = L.h;
    if (l == 0) {               // inter-layer dropout (nn.LSTM(dropout=0.8)): only after the first of the two layers
      L.hd = L.dropfused ? L.hd_fused : keep < 1.f ? b.ws(L.nm + ".hd", (int64_t)TP * rows * H, adt) : L.h;
      if (keep < 1.f && !L.dropfused) {
        Op& op = b.push(Fw, OP_DROPOUT_FWD, tag);
        op.drop.x = L.h; op.drop.y = L.hd; op.drop.seed = io_seed; op.drop.n = (int64_t)TP * rows * H; op.drop.keep = keep; op.drop.dt = adt; op.drop.layer = lid;
      }
    }
    layers.push_back(L);
    return L.hd;
  };
  struct FcRt { RunGemm g; Builder::Coef coef; std::function<void(int, int32_t*)> bias; };
  auto fc_forward = [&](const std::string& netname, Ptr x, int64_t rows, int H, int O, Ptr y, int ld, int flags, int tag) -> FcRt {
    const ParamInfo &Wf = b.par(netname + ".fc_output_layer.weight"), &bf = b.par(netname + ".fc_output_layer.bias");
    FcRt fc;
    fc.g = seq_gemm(x, adt, rows, H, 0, H, O, DT_F32);
    fc.coef = [=](int nn, int s, int j) -> int32_t { return pe(Wf, (int64_t)nn * H + j, 1); };
    fc.bias = [=](int nn, int32_t* o) { o[0] = pe(bf, nn, 1); o[1] = 0; };
    b.pack_weights(Fw, fc.g, fc.coef, netname + ".fc", tag, &fc.bias);
    set_y(fc.g, y, rows, ld, 0);
    fc.g.flags = flags;
    b.push(Fw, OP_RUNGEMM, tag).g = fc.g;
    return fc;
  };

  // ---- full-band model
  Ptr h0 = lstm_forward("fb_model", 0, 0, fb_in, FP, FP, B, Hf, 100);
  Ptr h1 = lstm_forward("fb_model", 1, 1, h0, Hf, Hf, B, Hf, 101);
  Ptr fbo = b.ws("fbo", (int64_t)TP * B * FP, DT_F32);
  { Op& m = b.push(Fw, OP_MEMSET, 102); m.ms.dst = fbo; m.ms.bytes = (int64_t)TP * B * FP * 4; }   // pad columns F..FP-1 stay 0
  FcRt fcf = fc_forward("fb_model", h1, B, Hf, F, fbo, FP, actf == 1 ? kRunRelu : 0, 102);
  if (actf > 1) { P->error = "FullSubNet: only ReLU / None full-band activations are on the HIP path"; return P; }

  // ---- sub-band input (models.py:647-665)
  const int64_t rs = (int64_t)B * F;
  Ptr sum_sb = b.ws("sum_sb", (int64_t)B * F, DT_F32);
  Ptr mu_sb = b.ws("mu_sb", B, DT_F32);
  Ptr sb_in = b.ws("sb_in", (int64_t)TP * rs * W, adt);
  if (nmode == 0) { Fsn f = fsn0(); f.in = mag_t; f.aux = fbo; f.sums = sum_sb; f.aux2 = mu_sb; b.push(Fw, OP_FSN_SBSUM, 200).fsn = f; }
  else { Fsn f = fsn0(); f.in = mag_t; f.aux = fbo; f.stat = st_sb; f.mode = nmode; f.src = 1; b.push(Fw, OP_FSN_NORMSTAT, 200).fsn = f; }
  { Fsn f = fsn0(); f.in = mag_t; f.aux = fbo; f.sums = mu_sb; f.out = sb_in; f.mode = nmode; f.stat = st_sb; b.push(Fw, OP_FSN_SBBUILD, 201).fsn = f; }
  Ptr h2 = lstm_forward("sb_model", 0, 2, sb_in, W, W, rs, Hs, 202);
  Ptr h3 = lstm_forward("sb_model", 1, 3, h2, Hs, Hs, rs, Hs, 203);
  Ptr sbo = b.ws("sbo", (int64_t)TP * rs * 2, DT_F32);
  FcRt fcs = fc_forward("sb_model", h3, rs, Hs, 2, sbo, 2, 0, 204);
  { Fsn f = fsn0(); f.in = sbo; f.out = io_crm; b.push(Fw, OP_FSN_OUT, 205).fsn = f; }

  // =================================================================================================== backward
  if (cfg.training) {
    // lane of the weight-gradient GEMMs: 1 = second stream (api.hip: issued behind the first recurrence kernel of the phase, joined in front of
    // the UNPACK); only when the recurrences are single launches (the per-frame GRU / fp32 formulation has no OP_LSTM_BWD to fork at)
    int wg_lane = (!gru && adt == DT_BF16 && !(tune_str("FSN_LANES") && atoi(tune_str("FSN_LANES")) == 0)) ? 1 : 0;
    // data parallel (cfg.grad_buckets >= 2): the sub-band model's weight gradients keep the second lane busy for ~12 ms after the main stream
    // is through (profiles/r03_tuning_notes.md section 8) - the full-band model's gradients (the FRONT of the flat arena, 2/3 of it) are
    // therefore produced ON the main stream, folded and unpacked there without waiting for the lane, and their all-reduce (started by the
    // caller at that op: sefd_plan_grad_bucket_range) runs under the sub-band weight gradients; the sub-band range follows at the end
    const bool fsn_buckets = cfg.grad_buckets >= 2 && wg_lane == 1;
    // wg_hold: the weight gradients of the layer wait for the NEXT recurrence launch instead of starting beside the input-gradient GEMM in
    // between (two MFMA-bound GEMMs side by side ran 10 % slower than one after the other; beside the HBM-bound recurrence they fill its idle CUs)
    int wg_hold = 0;
    auto lstm_backward = [&](LayerRt& L, Ptr dh, bool need_dx, Ptr dx, int dx_ld, int dx_off, int dx_N, int dx_dt, int tag) {
      const int H = L.H;
      const int64_t rows = L.rows;
      Ptr dgates = b.ws(L.nm + ".dgates", (int64_t)TP * rows * NG * H, adt);
      const ParamInfo* Whh = L.Whh;
      const bool um = L.cluster;
      Ptr dgh = dgates;                                   // gradient of the recurrent pre-activations: the same slab for the LSTM
      if (gru) {
        dgh = b.ws(L.nm + ".dgh", (int64_t)TP * rows * 3 * H, adt);
        RunGemm rb0 = step_gemm(dgh, adt, rows, 3 * H, 0, H, dh, H, 0, DT_F32, kRunAccum);
        Builder::Coef cT = [=](int nn, int s, int j) -> int32_t { return pe(*Whh, (int64_t)j * H + nn, 1); };
        b.pack_weights(R, rb0, cT, L.nm + ".hhT", tag);
        for (int t = TP - 1; t >= 0; --t) {
          LstmCell& cl = b.push(R, OP_CELL_BWD, tag).cell;
          std::memset(&cl, 0, sizeof(cl));
          cl.gates = b.mk(A_WS, L.gates.off + (int64_t)t * rows * 4 * H * 4);
          cl.c = cl.h = b.none();
          cl.c_prev = t > 0 ? b.mk(A_WS, L.h.off + (int64_t)(t - 1) * rows * H * esize(adt)) : b.none();
          cl.dh = b.mk(A_WS, dh.off + (int64_t)t * rows * H * 4);
          cl.dc = t > 0 ? b.mk(A_WS, dh.off + (int64_t)(t - 1) * rows * H * 4) : b.none();
          cl.dgates = b.mk(A_WS, dgates.off + (int64_t)t * rows * 3 * H * esize(adt));
          cl.gh = b.mk(A_WS, dgh.off + (int64_t)t * rows * 3 * H * esize(adt));
          cl.rows = rows; cl.H = H; cl.hdt = adt; cl.gdt = adt; cl.first = t == TP - 1; cl.kind = 1;
          if (t > 0) {
            RunGemm r = step_gemm(dgh, adt, rows, 3 * H, t, H, dh, H, (int64_t)(t - 1) * rows * H, DT_F32, kRunAccum);
            r.w = rb0.w;
            b.push(R, OP_RUNGEMM, tag).g = r;
          }
        }
      } else if (L.cluster) {
        LstmRec r;
        std::memset(&r, 0, sizeof(r));
        r.gx = L.gates; r.gates = L.gates; r.h = L.h; r.c = L.c; r.dh = dh; r.dgates = dgates;
        r.whh[0] = r.whh[1] = b.mk(A_PARAM, Whh->off * 4);
        r.gx_ld = 4 * H; r.G = 1; r.nset = 1; r.B = (int)rows; r.T = TP; r.H = H; r.hdt = adt; r.gdt = adt; r.tmajor = 1;
        if (L.rowsk) {                                      // W_hh^T packed: row = hidden unit, column = gate column (unit-major)
          RunGemm pk = step_gemm(dgates, adt, rows, 4 * H, 0, H, dh, H, 0, DT_F32, 0);
          Builder::Coef cT = [=](int nn, int s, int j) -> int32_t { return pe(*Whh, (int64_t)gate_torch_row(j, H) * H + nn, 1); };
          b.pack_weights(R, pk, cT, L.nm + ".hhTpk", tag);
          if (pk.ldw != 4 * H || pk.Npad != H) { P->error = "FullSubNet: packed W_hh^T layout"; return; }
          {
            int32_t* tab = reinterpret_cast<int32_t*>(P->consts.data() + R.back().pack.tab.off);
            std::vector<int32_t> old(tab, tab + (size_t)4 * H * H);
            for (int n = 0; n < H; ++n)
              for (int k = 0; k < 4 * H; ++k) tab[rows_wb_index(H, n, k)] = old[(size_t)n * 4 * H + k];
          }
          r.impl = 1; r.wpk_b = pk.w; r.wpk_f = b.none(); r.gxdt = L.sdt;
          r.xin = r.wpk_x = r.bias = r.hd = r.seed = b.none();
          if (L.dropbwd) { r.seed = io_seed; r.keep = keep; r.drop_layer = L.lid; }
          r.dhdt = L.dhdt;
          r.dyo = r.wo = b.none();
          if (L.headfuse) { r.dyo = L.dyo; r.wo = L.wo; r.no = 2; }
        }
        b.push(R, OP_LSTM_BWD, tag).lstm = r;
      } else {
      Ptr dc = b.ws(L.nm + ".dc", rows * H, DT_F32);
      // dh_{t-1} += dgates_t . W_hh : packed transposed recurrent weights
      RunGemm rb0 = step_gemm(dgates, adt, rows, 4 * H, 0, H, dh, H, 0, DT_F32, kRunAccum);
      Builder::Coef cT = [=](int nn, int s, int j) -> int32_t { return pe(*Whh, (int64_t)j * H + nn, 1); };
      b.pack_weights(R, rb0, cT, L.nm + ".hhT", tag);
      for (int t = TP - 1; t >= 0; --t) {
        Op& op = b.push(R, OP_CELL_BWD, tag);
        LstmCell& cl = op.cell;
        cl.gates = b.mk(A_WS, L.gates.off + (int64_t)t * rows * 4 * H * 4);
        cl.c = b.mk(A_WS, L.c.off + (int64_t)t * rows * H * 4);
        cl.c_prev = t > 0 ? b.mk(A_WS, L.c.off + (int64_t)(t - 1) * rows * H * 4) : b.none();
        cl.h = b.none();
        cl.dh = b.mk(A_WS, dh.off + (int64_t)t * rows * H * 4);
        cl.dc = dc;
        cl.dgates = b.mk(A_WS, dgates.off + (int64_t)t * rows * 4 * H * esize(adt));
        cl.rows = rows; cl.H = H; cl.hdt = adt; cl.gdt = adt; cl.first = t == TP - 1;
        if (t > 0) {
          RunGemm r = step_gemm(dgates, adt, rows, 4 * H, t, H, dh, H, (int64_t)(t - 1) * rows * H, DT_F32, kRunAccum);
          r.w = rb0.w;
          b.push(R, OP_RUNGEMM, tag).g = r;
        }
      }
      }
      // weight gradients over all steps: nothing needs them before UNPACK - on the weight-gradient lane (second stream) they run beside the
      // input-gradient GEMM and the NEXT layer's recurrence (343 workgroups of 48 sequences on 256 CUs: its second round leaves 2/3 of the chip idle)
      b.cur_lane = wg_lane;
      b.cur_hold = wg_hold;
      b.wg_rounds = wg_lane ? (tune_str("FSN_WG_ROUNDS") ? atoi(tune_str("FSN_WG_ROUNDS")) : 8) : 1;   // 3 -> 8 with the job-scheduled recurrences (r05 notes): 57.1 -> 56.6 ms
      RunGemm fw = L.gx;
      fw.ydt = adt;
      if (gru) set_y(fw, dgates, rows, NG * H, 0);        // the GRU's gradient slab is 3H wide (the forward slab keeps a 4th block for W_hn h + b_hn)
      Builder::Coef chh = [=](int nn, int s, int j) -> int32_t { return pe(*Whh, (int64_t)(um ? gate_torch_row(nn, H) : nn) * H + j, 1); };
      // A narrow input (the sub-band model's first layer: 32 features) next to H = 384 recurrent columns: [x_t | h_{t-1} | ones] is 64 + 384 + 64 =
      // 512 columns = exactly the two 256-wide k tiles the W_hh gradient alone occupies (its second tile half empty) - ONE weight-gradient GEMM
      // over dgates instead of two (the 1536 x 128 launch for W_ih and the bias, 1.6 ms at B = 64, and its pass over the 9.6 GB gate gradients are gone)
      const int xw = fw.nseg == 1 ? (int)rup(fw.seg[0].len, 64) : 0;
      const bool cat = !gru && L.rowsk && fw.nseg == 1 && fw.seg[0].src == 0 && xw == 64 && H % 64 == 0 && rup(xw + H + 64, 256) == rup(H, 256) &&
                       !(tune_str("FSN_WGCAT") && atoi(tune_str("FSN_WGCAT")) == 0);
      // The upper layer: [h1_t | h2_{t-1}] = 2 H = 768 columns = three whole 256-wide k tiles in ONE GEMM over dgates (W_ih and W_hh apart: 384 (+ 64 ones)
      // and 384 columns = 2 + 2 tiles, a quarter of them padding, and two passes over the gate gradients); the bias comes from the ones MFMA of k tile 0
      const bool cat2 = !cat && !gru && L.rowsk && fw.nseg == 1 && fw.seg[0].src == 0 && fw.seg[0].len == H && fw.seg[0].dt == 0 && (2 * H) % 256 == 0 &&
                        !(tune_str("FSN_WGCAT2") && atoi(tune_str("FSN_WGCAT2")) == 0) && !(tune_str("ONES_MFMA") && atoi(tune_str("ONES_MFMA")) == 0);
      if (cat || cat2) {
        RunGemm fc = fw;
        fc.x[1] = L.h; fc.bstride[1] = 0; fc.tstride[1] = (int)(rows * H); fc.base[1] = 0; fc.rowlen[1] = (int)(rows * H); fc.fstride[1] = H; fc.Tin[1] = TP;
        fc.seg[fc.nseg++] = Seg{1, -1, 0, H, 0};           // h_{t-1}
        const Builder::Coef cgx = L.cgx;
        Builder::Coef cc = [=](int nn, int s, int j) -> int32_t { return s == 0 ? cgx(nn, 0, j) : chh(nn, 0, j); };
        b.wgrad(R, fc, dgates, cc, tag, &L.bgx);
      } else {
      b.wgrad(R, fw, dgates, L.cgx, tag, &L.bgx);
      RunGemm fh = seq_gemm(L.h, adt, rows, H, 0, H, NG * H, adt);
      fh.seg[0].dt = -1;                                   // h_{t-1}
      set_y(fh, dgh, rows, NG * H, 0);
      b.wgrad(R, fh, dgh, chh, tag, gru ? &L.bhh : nullptr);           // GRU: b_hh belongs to this GEMM (bias "ones" run)
      }
      b.cur_lane = 0;
      b.cur_hold = 0;
      b.wg_rounds = 1;
      if (need_dx) {
        RunGemm g = seq_gemm(dgates, adt, rows, NG * H, 0, NG * H, dx_N, dx_dt);
        const Builder::Coef cf = L.cgx;
        Builder::Coef coef = [=](int nn, int s, int j) -> int32_t { return cf(j, 0, nn); };
        b.pack_weights(R, g, coef, L.nm + ".dx", tag);
        set_y(g, dx, rows, dx_ld, dx_off);
        b.push(R, OP_RUNGEMM, tag).g = g;
      }
    };
    auto fc_backward = [&](FcRt& fc, Ptr dy, Ptr x, int64_t rows, int H, int O, int ld, Ptr dh, int tag, const std::string& nm, bool head_fused = false) {
      RunGemm fw = fc.g;
      fw.ydt = adt; fw.flags = 0;
      // the sub-band head's weight gradient (a 1.2 ms pass over h of the upper layer in front of the first recurrence): on the weight-gradient
      // lane it is held back and runs in the CUs the recurrence's second dispatch round leaves idle
      if (head_fused) b.cur_lane = wg_lane;
      b.wgrad(R, fw, dy, fc.coef, tag, &fc.bias);
      b.cur_lane = 0;
      if (head_fused) return;                             // the row-block LSTM backward forms dh = dy . W_fc itself (O = 2: a rank-2 update)
      RunGemm g = seq_gemm(dy, adt, rows, ld, 0, O, H, DT_F32);
      const Builder::Coef cf = fc.coef;
      Builder::Coef coef = [=](int nn, int s, int j) -> int32_t { return cf(j, 0, nn); };
      b.pack_weights(R, g, coef, nm + ".fc.dg", tag);
      set_y(g, dh, rows, H, 0);
      b.push(R, OP_RUNGEMM, tag).g = g;
    };
    auto dropout_bwd = [&](LayerRt& L, Ptr dxd, int tag) -> Ptr {      // gradient wrt the un-dropped h (fp32, in place semantics via a copy)
      if (!(keep < 1.f)) return dxd;
      if (L.dropfused) { L.dropbwd = true; return dxd; }               // the row-block backward kernel multiplies dh by the mask as it loads it
      Ptr dhu = b.ws(L.nm + ".dhu", (int64_t)TP * L.rows * L.H, DT_F32);
      Op& op = b.push(R, OP_DROPOUT_BWD, tag);
      op.drop.x = dxd; op.drop.y = dhu; op.drop.seed = io_seed; op.drop.n = (int64_t)TP * L.rows * L.H; op.drop.keep = keep; op.drop.dt = DT_F32; op.drop.layer = L.lid;
      return dhu;
    };
    LayerRt &Lf0 = layers[0], &Lf1 = layers[1], &Ls0 = layers[2], &Ls1 = layers[3];
    // sub-band head
    Ptr d_sbo = b.ws("d_sbo", (int64_t)TP * rs * 2, adt);
    { Fsn f = fsn0(); f.in = io_gcrm; f.out = d_sbo; b.push(R, OP_FSN_OUT_BWD, 205).fsn = f; }
    // sub-band head: 2 outputs.  With the row-block kernels the [T x rows x H] fp32 gradient of h (4 GB written by a K = 2 GEMM, read back
    // by the recurrence) is never materialised: the kernel computes dh = d_sbo[.., 0] W_fc[0] + d_sbo[.., 1] W_fc[1] as it needs it
    Ls1.headfuse = Ls1.rowsk && !(tune_str("LSTM_HEADFUSE") && atoi(tune_str("LSTM_HEADFUSE")) == 0);
    Ptr dh3 = Ls1.headfuse ? b.none() : b.ws("dh3", (int64_t)TP * rs * Hs, DT_F32);       // (not even allocated then: 4.6 GB at B = 64)
    if (Ls1.headfuse) { Ls1.dyo = d_sbo; Ls1.wo = b.pptr("sb_model.fc_output_layer.weight"); }
    fc_backward(fcs, d_sbo, h3, rs, Hs, 2, 2, dh3, 204, "sb_model", Ls1.headfuse);
    // the gradient slab between the two sub-band layers ([T x rows x H]: 4.8 GB in fp32 at B = 64, written by the input-gradient GEMM and read once by
    // the row-block backward of the layer below): bf16 like every other activation gradient of the bf16 plans when nothing but that kernel
    // reads it (the inter-layer dropout fused into it, or no dropout); SEFD_FSN_DH16=0: fp32
    const bool dh16 = adt == DT_BF16 && Ls1.rowsk && Ls0.rowsk && (Ls0.dropfused || !(keep < 1.f)) && !(tune_str("FSN_DH16") && atoi(tune_str("FSN_DH16")) == 0);
    Ptr dh2d = b.ws("dh2d", (int64_t)TP * rs * Hs, dh16 ? adt : DT_F32);
    // round 6: with the upper layer's ONE weight-gradient GEMM (cat2, 3 k tiles) starting beside the input-gradient GEMM is 0.12 ms per step better than
    // waiting for the lower layer's recurrence (54.15 vs 54.28 ms, twice, one box); FSN_HOLD=1 restores the hold
    wg_hold = tune_str("FSN_HOLD") && atoi(tune_str("FSN_HOLD")) == 1;
    lstm_backward(Ls1, dh3, true, dh2d, Hs, 0, Hs, dh16 ? adt : DT_F32, 203);
    wg_hold = 0;
    if (dh16) Ls0.dhdt = adt;
    Ptr dh2 = dropout_bwd(Ls0, dh2d, 202);
    Ptr d_sbin = b.ws("d_sbin", (int64_t)TP * rs * W, DT_F32);
    lstm_backward(Ls0, dh2, true, d_sbin, W, 0, W, DT_F32, 202);
    // through the normalised concat into the full-band output
    Ptr sumS = b.ws("sum_S", (int64_t)B * F, DT_F32);
    Ptr Sm = b.ws("Sm", B, DT_F32);
    Ptr d_fb = b.ws("d_fb", (int64_t)TP * B * FP, adt);
    if (nmode == 0) {
      { Fsn f = fsn0(); f.in = d_sbin; f.aux = sb_in; f.sums = sumS; f.aux2 = Sm; b.push(R, OP_FSN_SBBWD_SUM, 201).fsn = f; }
      { Fsn f = fsn0(); f.in = d_sbin; f.aux = fbo; f.aux2 = mu_sb; f.sums = Sm; f.out = d_fb; b.push(R, OP_FSN_SBBWD_APPLY, 200).fsn = f; }
    } else {
      Ptr dpre = b.ws("d_fb_pre", (int64_t)TP * B * F, DT_F32);
      Ptr part = b.ws("normbwd_part", (int64_t)2 * B * F, DT_F32);
      { Fsn f = fsn0(); f.in = d_sbin; f.aux = fbo; f.aux2 = sb_in; f.stat = st_sb; f.sums = part; f.out = dpre; f.mode = nmode; f.src = 1;
        b.push(R, OP_FSN_NORMBWD, 201).fsn = f; }
      { Fsn f = fsn0(); f.in = dpre; f.aux = fbo; f.aux2 = mu_sb; f.sums = Sm; f.out = d_fb; f.mode = nmode; b.push(R, OP_FSN_SBBWD_APPLY, 200).fsn = f; }
    }
    if (fsn_buckets) b.flush_sums(R, 997, true);         // the sub-band folds: on the lane, behind the weight gradients they fold
    // full-band weight gradients (four 80 us launches): main stream.  Two gradient buckets need them there (no wait for the lane); since round 6 always: at
    // the end of the lane they ran 0.35 ms past the main stream, which idles beside the 8-workgroup cluster recurrences (FSN_FB_LANE=1: on the lane)
    if (fsn_buckets || !(tune_str("FSN_FB_LANE") && atoi(tune_str("FSN_FB_LANE")) == 1)) wg_lane = 0;
    Ptr dh1 = b.ws("dh1", (int64_t)TP * B * Hf, DT_F32);
    fc_backward(fcf, d_fb, h1, B, Hf, F, FP, dh1, 102, "fb_model");
    Ptr dh0d = b.ws("dh0d", (int64_t)TP * B * Hf, DT_F32);
    lstm_backward(Lf1, dh1, true, dh0d, Hf, 0, Hf, DT_F32, 101);
    Ptr dh0 = dropout_bwd(Lf0, dh0d, 100);
    lstm_backward(Lf0, dh0, false, b.none(), 0, 0, 0, DT_F32, 100);
    if (fsn_buckets) {
      const int64_t sb_lo = b.par("sb_model.sequence_model.weight_ih_l0").off;
      b.unpack_range(R, 0, sb_lo, 998, true);             // folds + UNPACK of the full-band range: no wait for the lane
      b.unpack_lo = sb_lo;
      P->bucket_elem = 0; P->bucket_end = sb_lo;
    }
    b.finish_unpack(R);
  }
  finalize_rungemms(b, P);
  for (size_t k = 0; k < P->bwd.size(); ++k)
    if (P->bwd[k].kind == OP_UNPACK && P->bwd[k].tag == 998) P->bucket_op = (int32_t)k;
  P->arena_bytes[A_WS] = b.ws_off;
  P->arena_bytes[A_PARAM] = nparam * 4;
  P->arena_bytes[A_GRAD] = nparam * 4;
  P->arena_bytes[A_STATE] = 4;
  P->arena_bytes[A_CONST] = (int64_t)P->consts.size();
  P->arena_bytes[A_IO] = b.io_off;
  return P;
}

// =================================================================================================================
// torch.stft front end of FullSubNet (model 4; tools_for_model.py:628-648): centre / reflect padding, hop = cfg.hop,
// periodic Hann(win_len) zero-padded to fft_len in the middle.  io.wav [B][L] -> io.spec = complex64 image [B][NF][T][2].
Plan* build_torchstft_plan(const ModelConfig& cfg) {
  Plan* P = new Plan();
  P->cfg = cfg;
  Builder b;
  b.P = P;
  b.c = cfg;
  const int B = cfg.B, L = cfg.L, W = cfg.win_len, hop = cfg.hop, NFFT = cfg.fft_len;
  const int pad = NFFT / 2, Lp = L + 2 * pad;
  const int T = 1 + L / hop;
  const int NF = NFFT / 2 + 1, NS = NF + 1, SW = NS * 2;
  P->T = T;
  P->NF = NF;
  if (hop % 4 != 0 || pad >= L) { P->error = "torch.stft plan: hop must be a multiple of 4 and the clip longer than fft_len/2"; return P; }
  Ptr io_wav = b.io("wav", (int64_t)B * L);
  Ptr io_spec = b.io("spec", (int64_t)B * NF * T * 2);
  Ptr wpad = b.ws("wpad", (int64_t)B * Lp, DT_F32);
  Ptr spec = b.ws("spec", (int64_t)B * T * SW, DT_F32);
  { Op& op = b.push(P->fwd, OP_REFLECTPAD, 1); op.rpad.src = io_wav; op.rpad.dst = wpad; op.rpad.B = B; op.rpad.L = L; op.rpad.pad = pad; }
  std::vector<double> win(NFFT, 0.0);
  const int left = (NFFT - W) / 2;
  for (int j = 0; j < W; ++j) win[left + j] = 0.5 - 0.5 * std::cos(2.0 * kPi * j / W);
  RunGemm g = Builder::gemm0();
  g.x[0] = wpad; g.xdt = DT_F32; g.ydt = DT_F32;
  g.bstride[0] = Lp; g.rowlen[0] = Lp; g.fstride[0] = hop; g.Tin[0] = 1;
  g.M = B * T; g.Tout = 1; g.Fo = T;
  g.nseg = 1; g.seg[0] = Seg{0, 0, 0, NFFT, 0};
  g.N = SW;
  Builder::layout_segs(g);
  {
    std::vector<float> wt((size_t)g.Npad * g.ldw, 0.f);
    for (int nn = 2; nn < g.N; ++nn)
      for (int j = 0; j < NFFT; ++j) {
        const double ang = 2.0 * kPi * (double)(((int64_t)(nn / 2 - 1) * j) % NFFT) / NFFT;
        wt[(size_t)nn * g.ldw + j] = (float)(((nn & 1) == 0 ? std::cos(ang) : -std::sin(ang)) * win[j]);
      }
    g.w = b.cst(wt.data(), (int64_t)wt.size() * 4);
  }
  g.y = spec; g.y_bstride = (int64_t)T * SW; g.y_fstride = SW;
  b.push(P->fwd, OP_RUNGEMM, 2).g = g;
  SpecOut so;
  std::memset(&so, 0, sizeof(so));
  so.est = spec; so.out_real = io_spec; so.out_imag = b.none(); so.B = B; so.T = T; so.NF = NF; so.mode = 3;
  b.push(P->fwd, OP_SPECOUT_FWD, 3).so = so;
  finalize_rungemms(b, P);
  P->arena_bytes[A_WS] = b.ws_off;
  P->arena_bytes[A_PARAM] = 4; P->arena_bytes[A_GRAD] = 4; P->arena_bytes[A_STATE] = 4;
  P->arena_bytes[A_CONST] = (int64_t)P->consts.size();
  P->arena_bytes[A_IO] = b.io_off;
  return P;
}

// =================================================================================================================
// torch.istft(n_fft, hop, win_length, hann_window(win_length), center=True, length=L) - the inverse front end of FullSubNet's
// validation path (tools_for_model.py:651-680, called at trainer.py:341-345).  IO: spec [B][NF][T][2] (memory image of the
// complex [B, NF, T] tensor, or the reference's real-pair [B, NF, T, 2]) -> wav [B][L].  Same two kernels as the ConviSTFT path:
// the inverse-FFT frame kernel (its rank-2 "correction" table set to the irfft's half weights of the DC and Nyquist bins:
//   irfft(X)[j] = (S[j] - Re X[0] / 2 - (-1)^j Re X[N/2] / 2) / (N/2),  S[j] = Re sum_{k <= N/2} X[k] e^{2 pi i k j / N})
// and the overlap-add kernel with the window-envelope normaliser, without the clamp.
Plan* build_torchistft_plan(const ModelConfig& cfg) {
  Plan* P = new Plan();
  P->cfg = cfg;
  Builder b;
  b.P = P;
  b.c = cfg;
  const int B = cfg.B, L = cfg.L, W = cfg.win_len, hop = cfg.hop, NFFT = cfg.fft_len;
  const int pad = NFFT / 2;
  const int T = 1 + L / hop;
  const int NF = NFFT / 2 + 1, NS = NF + 1, SW = NS * 2;
  P->T = T;
  P->NF = NF;
  if (NFFT != 512 || W > NFFT || pad >= L || (T - 1) * hop + NFFT < pad + L) { P->error = "torch.istft plan: fft_len must be 512 and the frames must cover the clip"; return P; }
  Ptr io_spec = b.io("spec", (int64_t)B * NF * T * 2);
  Ptr io_wav = b.io("wav", (int64_t)B * L);
  Ptr est = b.ws("est", (int64_t)B * T * SW, DT_F32);
  Ptr frames = b.ws("frames", (int64_t)B * T * NFFT, DT_F32);
  std::vector<double> win(NFFT, 0.0);
  const int left = (NFFT - W) / 2;
  for (int j = 0; j < W; ++j) win[left + j] = 0.5 - 0.5 * std::cos(2.0 * kPi * j / W);
  { Op& op = b.push(P->fwd, OP_MEMSET, 1); op.ms.dst = est; op.ms.bytes = (int64_t)B * T * SW * 4; }       // slot 0 of every frame stays 0
  SpecOut so;
  std::memset(&so, 0, sizeof(so));
  so.est = est; so.out_real = io_spec; so.out_imag = b.none(); so.B = B; so.T = T; so.NF = NF; so.mode = 3; so.accumulate = 0;
  b.push(P->fwd, OP_SPECOUT_BWD, 2).so = so;
  {
    std::vector<float> tw(1024);
    for (int k = 0; k < 512; ++k) { tw[2 * k] = (float)std::cos(2.0 * kPi * k / 512.0); tw[2 * k + 1] = (float)std::sin(2.0 * kPi * k / 512.0); }
    std::vector<float> c(4 * 257, 0.f);                  // [even | odd][re | im][k]
    c[(0 * 2 + 0) * 257 + 0] = 0.5f; c[(0 * 2 + 0) * 257 + 256] = 0.5f;
    c[(1 * 2 + 0) * 257 + 0] = 0.5f; c[(1 * 2 + 0) * 257 + 256] = -0.5f;
    Op& op = b.push(P->fwd, OP_ISTFT_FFT, 3);
    op.ifft.est = est; op.ifft.frames = frames; op.ifft.tw = b.cst(tw.data(), 4096); op.ifft.win = b.win512(win);
    op.ifft.corr = b.cst(c.data(), (int64_t)c.size() * 4); op.ifft.nframes = (int64_t)B * T; op.ifft.W = NFFT;
  }
  const int Lp = (T - 1) * hop + NFFT;
  std::vector<float> env(Lp, 0.f);
  for (int t = 0; t < T; ++t)
    for (int j = 0; j < NFFT; ++j) env[t * hop + j] += (float)(win[j] * win[j]);
  Ola ola;
  std::memset(&ola, 0, sizeof(ola));
  ola.frames = frames; ola.wav = io_wav; ola.coff = b.cst(env.data(), (int64_t)env.size() * 4); ola.dwav = ola.dpad = b.none();
  ola.B = B; ola.T = T; ola.L = L; ola.win = NFFT; ola.hop = hop; ola.trim = pad; ola.noclamp = 1;
  b.push(P->fwd, OP_OLA_FWD, 4).ola = ola;
  finalize_rungemms(b, P);
  P->arena_bytes[A_WS] = b.ws_off;
  P->arena_bytes[A_PARAM] = 4; P->arena_bytes[A_GRAD] = 4; P->arena_bytes[A_STATE] = 4;
  P->arena_bytes[A_CONST] = (int64_t)P->consts.size();
  P->arena_bytes[A_IO] = b.io_off;
  return P;
}

Plan* build_plan(const ModelConfig& cfg) {
  if (cfg.model == 5) return build_torchistft_plan(cfg);
  if (cfg.model == 4) return build_torchstft_plan(cfg);
  if (cfg.model == 3) return build_fsn_plan(cfg);
  return cfg.model == 2 ? build_frontend_plan(cfg) : (cfg.model == 1 ? build_crn_plan(cfg) : build_dccrn_plan(cfg));
}

}  // namespace sefd
