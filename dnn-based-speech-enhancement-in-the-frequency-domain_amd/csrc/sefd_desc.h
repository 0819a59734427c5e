// Launch descriptors for the MI355X speech-enhancement hot path.
//
// The host-side planner (plan.cpp, pure C++) turns a model configuration into a flat list of
// `Op`s over five memory arenas.  The HIP executor (exec.hip) launches one hand-written gfx950
// kernel per Op; the test-only host simulator (tests/hostsim) interprets the same list on the CPU
// so that the planner's index arithmetic can be checked without a GPU.  All structs are PODs.
#pragma once
#include <cstdlib>
#include <cstdint>

namespace sefd {

enum Arena : int32_t {
  A_NONE = -1,
  A_WS = 0,      // workspace: activations, gradients, packed weights, partial sums (caller allocated)
  A_PARAM = 1,   // flat fp32 trainable parameters, reference state_dict order
  A_GRAD = 2,    // flat fp32 gradients, same layout as A_PARAM
  A_STATE = 3,   // module buffers (BatchNorm running_mean / running_var), fp32
  A_CONST = 4,   // plan constants: STFT bases, OLA normaliser, pack/unpack index tables
  A_IO = 5,      // per-call I/O block: [wav | out_wav | out_real | out_imag | grad_wav | grad_real | grad_imag]
  A_COUNT = 6
};

enum DType : int32_t { DT_F32 = 0, DT_BF16 = 1 };

struct Ptr {
  int32_t arena;
  int32_t pad_;
  int64_t off;   // byte offset inside the arena
};

// ----------------------------------------------------------------------------------------------
// RUNGEMM:  y[row(b,u,fo)][n] = bias[n] + sum_seg sum_j  A_seg(b,u,fo)[j] * W[n][seg.koff + j]
// where A_seg is a *contiguous run* of `len` elements of source tensor `src`:
//     tt = u + seg.dt                     (valid iff 0 <= tt < Tin[src])
//     r  = seg.off + fo*fstride[src] + j  (valid iff 0 <= r < rowlen[src]);  invalid -> 0
//     elem = x[src][ b*bstride[src] + tt*tstride[src] + base[src] + r ]
// This single form covers: the strided (5,2) conv, both sub-pixel phases of the transposed conv, their
// input-gradients, the hoisted LSTM / Linear GEMMs, the conv-STFT framing GEMM and the iSTFT synthesis.
// seg.src == -1 is the "ones" run (value 1 at j == 0): it turns the weight-gradient GEMM into a bias-gradient too.
struct Seg {
  int32_t src, dt, off, len, koff;
};

constexpr int kMaxSeg = 8;

struct RunGemm {
  Ptr x[2];
  int32_t xdt;             // DType of x[*] and of w
  int32_t ydt;             // DType of y
  int64_t bstride[2];
  int32_t tstride[2], base[2], rowlen[2], fstride[2], Tin[2];
  int32_t M, Tout, Fo;     // M = B*Tout*Fo ; m -> b = m/(Tout*Fo), u = (m/Fo)%Tout, fo = m%Fo
  int32_t nseg;
  Seg seg[kMaxSeg];
  Ptr w;                   // packed weights [Npad][ldw], K-contiguous, zero padded   (WGRAD: fp32 partial output [nsplit][Npad][ldw])
  int32_t ldw, N, Npad;
  Ptr bias;                // fp32 [N] or A_NONE
  Ptr y;                   // output (WGRAD: the upstream gradient dy, read)
  int64_t y_bstride;
  int32_t y_tstride, y_fstride, y_off;   // y index = b*y_bstride + u*y_tstride + fo*y_fstride + y_off + n
  Ptr stats;               // A_NONE or fp32 partial sums [gridM][2][Npad] (sum, sum of squares of the stored value)
  int32_t nsplit;          // WGRAD only: number of row splits
  int32_t flags;           // bit 0: every run is 16-byte aligned and a whole number of 16-byte chunks -> LDS-DMA loader
  Ptr zero;                // >= 16 zero bytes (A_CONST): source of padding chunks for the LDS-DMA loader
  // Division by the two run-time row-decode divisors (Tout*Fo and Fo) as multiply-high + shift, filled by the planner
  // (fastdiv_make): for 0 <= x < 2^31, x / d == umulhi(x, m) >> s ; m == 0 encodes d == 1.  The kernels decode 4-8 row
  // indices per thread in their prologue; with 15 000-workgroup launches on the thin layers that was a visible cost.
  uint32_t div_tf_m, div_tf_s, div_fo_m, div_fo_s;
  // kRunBnBwd: this GEMM produces (a component of) the upstream gradient dz of a BatchNorm2d + PReLU layer, and its epilogue accumulates
  // that component's share of the layer's backward reductions - the separate pass over y and dz (BN_BWD_REDUCE) disappears.  For every
  // stored element dz[m][n] (the ROUNDED value the later passes read) with the layer's forward output yf = bnb_y[row(m) + n]:
  //   xh = (yf - mean[n]) * invstd[n];  bn = gamma[n] * xh + beta[n];  dbn = bn > 0 ? dz : slope * dz
  //   part[blk][0][n] += dbn;  part[blk][1][n] += dbn * xh;  part[blk][2][n] += bn > 0 ? 0 : bn * dz      (blk = m / kBM, `stats` = part)
  // row(m) = b * bnb_bstride + u * bnb_tstride + fo * bnb_fstride + bnb_off  (the decoder's y keeps one more frame than dz).
  // The sums are linear in dz: when dz = dz0 + dz1 (skip connection) each producer adds its own share.
  Ptr bnb_y, bnb_mi, bnb_gamma, bnb_beta, bnb_slope;
  int64_t bnb_bstride;
  int32_t bnb_tstride, bnb_fstride, bnb_off;
  // n2 > 0: two destinations - output columns n >= n2 are stored to y2 at column n - n2 with the same row offsets (one GEMM over the
  // upstream gradient produces the input gradients of BOTH sources of a decoder layer: previous layer's output and the skip connection)
  int32_t n2;
  Ptr y2;
  // kRunDyFromBn (WGRAD of the first encoder layer, enc0.hip): the upstream-gradient operand is not a stored tensor.  `y` points at dz0, the gradient with
  // respect to the layer's BatchNorm + PReLU OUTPUT (bnb_dz1: a second component that is added, the skip connection's; A_NONE = none), and each element is
  // taken through the layer's BatchNorm + PReLU backward on the way in:  xh = (yf - mean) * invstd, bn = gamma * xh + beta, dbn = bn > 0 ? dz : slope * dz,
  // dy = gamma * invstd * (dbn - t0 * inv_count - xh * t1 * inv_count)  with yf = bnb_y at the SAME index as dz, (t0, t1) = bnb_totals[0..C), [C..2C)
  // (what BN_BWD_APPLY would store, in fp32): the layer has no input gradient, so nothing else reads dy - the 127 MB store, its read back and the
  // BN_BWD_APPLY launch on the step's serial tail are gone.
  Ptr bnb_dz1, bnb_totals;
  float bnb_inv_count;
  int32_t pad2_;
};
static inline void fastdiv_make(uint32_t d, uint32_t* m, uint32_t* s) {
  if (d <= 1) { *m = 0; *s = 0; return; }
  uint32_t L = 0;
  while ((1ull << L) < d) ++L;                                   // L = ceil(log2 d) >= 1
  *m = (uint32_t)((((uint64_t)1 << (31 + L)) + d - 1) / d);     // ceil(2^(31+L) / d) < 2^32
  *s = L - 1;                                                    // (31 + L) - 32
}
constexpr int kRunAligned = 1;   // LDS-DMA loader usable
constexpr int kRunAccum = 2;     // y += result (fp32 y): recurrent term added onto the hoisted input GEMM / gradient accumulation
constexpr int kRunRelu = 4;      // y = max(result, 0)
constexpr int kRunYAligned = 8;  // bf16 output whose rows are whole 16-byte chunks: the tile is staged through LDS and stored wide
constexpr int kRunWTile32 = 16;  // packed weights are stored K-tile major, [ldw / 32][Npad][32]: the B operand of one 32-deep K tile is ONE
                                 // contiguous block, so every LDS-DMA of the wide-tile kernel (cgemm256.hip) moves whole 128-byte lines
constexpr int kRunBnBwd = 64;    // epilogue accumulates BatchNorm-backward partial sums against the layer's forward output (fields bnb_*); `stats` rows are [3][Npad]
constexpr int kRunDyFromBn = 1024; // WGRAD: upstream gradient formed on the fly from dz through the BatchNorm + PReLU backward (fields bnb_*, bnb_dz1, bnb_totals)
constexpr int kRunOnesMfma = 2048;  // WGRAD, 256 x 256 tile: the bias ones run (last 64 packed columns) is no k tile; k tile 0 multiplies dy with a constant ones operand
constexpr int kRunRank = 4096;      // WGRAD of a layer with N <= 4 outputs over one contiguous bf16 array (wgrad_rank_form below): streamed once by wgrad_rank_kernel (rungemm.hip)
constexpr int kRunEnc0 = 512;    // first encoder layer of a bf16 plan read from the fp32 spectrum itself (xdt = fp32, ydt = bf16, runs of 10 floats): enc0.hip
constexpr int kRunWgWide = 32;   // WGRAD: the planner sized the row splits for the 256 x 256 tile of the 8-wave kernel (rungemm.hip launch_wgrad_wide)
// element index of W[n][k] inside the packed weight buffer of `g`
static inline int64_t w_index(int flags, int ldw, int Npad, int n, int k) {
  return (flags & kRunWTile32) ? ((int64_t)(k >> 5) * Npad + n) * 32 + (k & 31) : (int64_t)n * ldw + k;
}

// element index of W[n][k] for any RUNGEMM descriptor
static inline int64_t w_index_g(const RunGemm& g, int n, int k) { return w_index(g.flags, g.ldw, g.Npad, n, k); }

// PACK: dst[i] = sum_{e < width} sign(tab[i*width+e]) * src[|tab[i*width+e]|-1]   (entry 0 -> nothing).  Table int32 in A_CONST.
// width 1: packed conv / LSTM weights ; width 2: combined biases (b_r - b_i | b_r + b_i), (b_ih + b_hh).
struct Pack {
  Ptr tab, src, dst;
  int64_t n;
  int32_t ddt, width;
};

// All PACK ops of one phase merged into one launch (they read only A_PARAM, so they can all run first): `entries` is an
// array of `count` Pack descriptors in A_CONST; workgroup row y serves entry y.
struct PackMulti {
  Ptr entries;
  int32_t count, pad_;
};

// SPLITSUM: part[0][i] = sum_s part[s*sstride + i]   (in place, fixed order -> deterministic)
// UNPACK:   grad[j] = sum_{e in [start[j], start[j+1])} sign(ent[e]) * src[|ent[e]|-1]    (CSR table in A_CONST)
struct Unpack {
  Ptr start, ent, part, dst;   // start: int32[n+1]; ent: int32[]; part: fp32 ; dst fp32
  int64_t n;                   // SPLITSUM: elements per split ; UNPACK: number of gradient elements written
  int64_t sstride;             // elements between splits
  int32_t nsplit;
  int32_t nseg;                // SPLITSUM with nseg > 0: `start` is an int64 [nseg][3] table (offset from `part` in elements, elements per
                               // split, splits; split stride = elements) - the folds of many weight gradients in one launch; n = largest segment
};

// BatchNorm2d (training) + PReLU on channels-last rows [R][C].
struct BnFinalize {            // partial sums -> mean, invstd ; running stats momentum update
  Ptr part;                    // [nblk][2][Cpad]
  Ptr mean_invstd;             // fp32 [2][C]  (workspace)
  Ptr running_mean, running_var;   // A_STATE (or A_NONE in eval)
  int32_t nblk, C, Cpad, mode; // mode 0: partials -> statistics.  SyncBN: 1: partials -> totals (this rank's sum, sum of squares);
                               //         2: totals (all-reduced by the caller in between) -> statistics
  double count;                // rows contributing (all ranks in mode 2)
  float eps, momentum;
  Ptr totals;                  // fp64 [2][C] (modes 1, 2)
  int32_t nsub, substride;     // nsub > 1: a partial row holds nsub column groups of the same channels (the two sub-pixel phases of a merged
                               // transposed-conv GEMM, columns phase * substride + c): channel c sums all of them
};
struct BnApply {               // z = prelu(gamma*(y-mean)*invstd + beta)
  Ptr y, z, mean_invstd, gamma, beta, slope;
  int64_t R;
  int32_t C, dt;               // dt: DType of y and z
};
struct BnBwdReduce {           // per-channel partial sums of (dbn, dbn*xhat) and of the PReLU slope gradient (row 2: per channel, summed at finalize)
  Ptr y, dz0, dz1;             // dz1 optional second upstream gradient (skip connection): dz = dz0 + dz1
  Ptr mean_invstd, gamma, beta, slope;
  Ptr part;                    // fp32 [nblk][3][C]  (sum dbn, sum dbn*xhat, slope-gradient share of channel c)
  int64_t R;
  int32_t C, dt, nblk, rows_per_blk;
  // y has `rpb` rows per batch element of which the first `skip` (the decoder frame dropped by `[..., 1:]`) receive no
  // upstream gradient; dz0 holds only the remaining rows:  y row r -> b = r / rpb, q = r % rpb; dz row = b*(rpb-skip) + q-skip.
  int64_t rpb;
  int32_t skip;
  int32_t ldp;                 // row pitch of `part` in floats (0: C): partial rows written by GEMM epilogues (kRunBnBwd) are Npad wide
};
struct BnBwdApply {            // FINALIZE: partials -> totals [3][C] + parameter gradients ; APPLY: dy = gamma*invstd*(dbn - mean(dbn) - xhat*mean(dbn*xhat))
  BnBwdReduce r;
  Ptr totals;                  // fp32 [3][C] workspace
  Ptr dy;                      // same shape/dtype as y
  Ptr dgamma, dbeta, dslope;   // A_GRAD
  double count;
};

// ComplexBatchNorm (tools_for_model.py:430-607; DCCRN(use_cbn=True)) + PReLU on channels-last rows [R][C]: channel k < h = C / 2 is the real part and
// channel k + h the imaginary part of complex channel k.  Statistics (2 means, 3 covariances per complex channel) come from a pass of their own
// (the GEMM epilogues give per-column sums, not the real x imaginary cross term); everything downstream is a per-complex-channel 2 x 2 affine map.
//   coef  fp32 [14][h]: Zrr Zri Zir Zii | br' bi' (y = Z x + b', b' = B - Z M) | Mr Mi | Urr Uri Uii (V^-1/2) | Vrr+eps Vri Vii+eps (backward finalize)
//   coefb fp32 [9][h]:  Zrr Zri Zir Zii | mean(dbn_r) mean(dbn_i) | 2 dVrr / N, dVri / N, 2 dVii / N
struct CbnFwd {                // OP_CBN_STATS (y -> part), OP_CBN_FINALIZE (part -> coef, running statistics), OP_CBN_APPLY (y, coef -> z)
  Ptr y, z, part, coef;        // part fp32 [nblk][5][h]: sum xr, sum xi, sum xr^2, sum xi^2, sum xr xi
  Ptr W[3], Bv[2], slope;      // Wrr Wri Wii, Br Bi (A_PARAM), PReLU slope
  Ptr RM[2], RV[3];            // RMr RMi, RVrr RVri RVii (A_STATE)
  int64_t R;
  int32_t C, dt, nblk, rows_per_blk, training, pad_;
  double count;
  float eps, momentum;
};
struct CbnBwd {                // OP_CBN_BWD_REDUCE (-> part), OP_CBN_BWD_FINALIZE (part -> coefb, parameter gradients), OP_CBN_BWD_APPLY (-> dy)
  Ptr y, dz0, dz1, dy;         // dz0 / dz1 / rpb / skip as in BnBwdReduce
  Ptr coef, coefb, part;       // part fp32 [nblk][7][h]: sum dbr, dbi, dbr x~r, dbr x~i, dbi x~r, dbi x~i, slope-gradient share (dbn = PReLU'(bn) dz, x~ = x - M)
  Ptr W[3], slope;
  Ptr dW[3], dB[2], dslope;    // A_GRAD
  int64_t R, rpb;
  int32_t C, dt, nblk, rows_per_blk, skip, pad_;
  double count;
};

// LSTM recurrence (input GEMM hoisted).  G independent groups, group g uses weight set g % nset.
// gx   [G][B][T][4H] fp32 gate pre-activations from the input GEMM (bias included), PyTorch gate order i,f,g,o
// whh  [nset][4H][H]  (param arena, fp32, reference layout weight_hh_l0)
// h    [G][B][T][H]   (dtype hdt) ; gates [G][B][T][4H] fp32 post-activation ; c [G][B][T][H] fp32
// Row (b,t) of group g:  gx  at gx  + gx_goff[g]  + (b*T+t)*gx_ld   (fp32, 4H wide)
//                        dgates at dgates + gx_goff[g] + (b*T+t)*gx_ld (dtype gdt, 4H wide)
// Gate-column order of gx / dgates (inside a group's 4H block) and of the saved gates [rows][H][4]: the four gates
// (q = 0 i, 1 f, 2 g, 3 o) of one hidden unit are adjacent, column = unit*4 + q, so an LSTM lane moves a (row, unit) cell
// with one 16-byte access.  PyTorch keeps gate-major rows: weight_ih / weight_hh / bias row of that column = q*H + unit.
static inline constexpr int gate_col(int q, int unit) { return unit * 4 + q; }
static inline constexpr int gate_torch_row(int col, int H) { return (col & 3) * H + (col >> 2); }

struct LstmRec {
  Ptr gx, whh[2], h, gates, c;
  Ptr dh, dgates;              // backward: dh [G][B][T][H] upstream (fp32) ; dgates out (same addressing as gx)
  int64_t gx_goff[4];
  int32_t gx_ld;
  int32_t G, nset, B, T, H, hdt, gdt;
  int32_t t0, t1;              // forward only: time range [t0, t1) of this launch (t1 == 0: the whole sequence).  t0 > 0 resumes
                               // from the saved h[t0-1], c[t0-1]: the planner chunks the sequence so that layer l+1 can start on
                               // a chunk while layer l works on the next one (two HIP streams)
  int32_t tmajor, impl;        // tmajor 0: buffers are [group][sequence][T][..] (DCCRN / CRN); 1: [group][T][sequence][..] (FullSubNet's
                               // time-major slabs; cluster kernels of lstm_cluster.hip and the host simulator only)
  // impl 1 (time-major, G == 1, thousands of sequences: FullSubNet's sub-band model): row-block kernels of lstm_rows.hip - a
  // workgroup owns 48 sequences and ALL hidden units, h_t / dgates_t live in LDS, the packed bf16 weights are streamed from L2 every
  // frame: wpk_f = W_hh as [4H gate columns (unit-major)][H], wpk_b = its transpose [H][4H] (both written by PACK ops of the plan)
  Ptr wpk_f, wpk_b;
  // impl 1, forward, xfeat == 32 (the sub-band model's first layer: 31 neighbour bins + the full-band output) or xfeat == H (a layer above it,
  // fed by the layer below's h - after dropout -, wpk_x then has H/32 k-steps: rows_wf_index(H, c, k)): the input projection is
  // fused - gate pre-activations = bias (b_ih + b_hh, fp32 [4H] unit-major) + x_t . W_ih^T (xin bf16 [T][rows][32], wpk_x packed like
  // wpk_f with ONE k-step: rows_wf_index(32, c, k)) + h_{t-1} . W_hh^T; `gx` is not read (no 8 GB pre-activation slab to write and re-read)
  Ptr xin, wpk_x, bias;
  int32_t xfeat, pad3_;
  // impl 1, forward, hd != A_NONE: the inverted dropout that follows the layer (struct Dropout: seed, keep, layer) is applied while h_t is
  // stored - hd [T][rows][H] (dtype hdt) = h * scale, no separate pass over the 2 GB h array
  // backward, seed != A_NONE: `dh` is the gradient w.r.t. the DROPPED h: it is multiplied by the same mask as it is loaded
  Ptr hd, seed;
  float keep;
  int32_t drop_layer;
  // impl 1, backward, no == 2: the upstream gradient is not read from `dh` but formed from the 2-output head that follows the layer:
  // dh[t][row][u] = dyo[t][row][0] * wo[0][u] + dyo[t][row][1] * wo[1][u]   (dyo dtype gdt [T][rows][2], wo fp32 [2][H] in A_PARAM)
  Ptr dyo, wo;
  int32_t no, dhdt;            // dhdt (impl 1, backward, no != 2): dtype of the `dh` slab - DT_BF16 when the input-gradient GEMM of the layer above writes bf16
                               // (FullSubNet's sub-band layers: 4.8 GB less written and read per step at B = 64), else fp32
  int32_t gxdt, pad2_;          // impl 1 only: dtype of the gx / gates slabs (DT_BF16 halves the HBM traffic that bounds these layers; the cell
                               // update itself uses the unrounded fp32 gate values, the backward reads the stored ones)
};

// Complex combine (tools_for_model.py:171-172): out[b,t, 0:H] = h[g0] - h[g3];  out[b,t,H:2H] = h[g2] + h[g1]
struct Combine {
  Ptr h, out;                  // h [4][B*T][H], out [B*T][2H]   (forward)   /  dout -> dh (backward)
  int64_t rows;                // B*T
  int32_t H, dt;
  int32_t T, t0, t1, pad_;     // forward only, t1 > 0: just the frames [t0, t1) of every batch item (rows are b*T + t)
};

// Mask application (models.py:253-276) on interleaved spectra.
// spec / est [B][T][NF+1][2] fp32 (r,i) with slot 0 a zero pad and slot k+1 = bin k (keeps bin 1 16-byte aligned);
// mask [..][NF-1][2] for bins 1..NF-1 (decoder output, dtype mdt).
struct Mask {
  Ptr spec, mask, est;         // forward
  Ptr dest, dmask;             // backward: dest = d(est) (fp32), dmask out (dtype mdt)
  int64_t frames;              // B*T
  int32_t NF, mode;            // mode 0=E 1=C 2=R
  int32_t mdt, pad0_;          // DType of mask / dmask
  int64_t mask_fstride;        // elements between consecutive frames of `mask` (decoder buffer has T+1 frames / batch)
  int64_t mask_bstride, mask_base;
  int32_t T;
  int32_t mch;                 // channels of the mask tensor: 2 (DCCRN complex mask) or 1 (CRN magnitude mask, mode 3)
  Ptr estm;                    // modes 3 / 5: CRN.forward's first output as [B*T][NF] fp32 (mode 3: tanh(mask) * |spec| ; mode 5, CRN
                               // 'Direct(None make)' models.py:506-517: the decoder output itself, bin 0 = 0, noisy phase re-attached)
  Ptr destm;                   // backward, modes 3 / 5, optional: gradient w.r.t. that first output (crn_direct_train's loss, trainer.py:169-170)
  // backward, colsum_rows > 0: the launch has exactly colsum_rows workgroups and workgroup w also writes colsum[w][8] (fp32): its share of
  // the column sums of the STORED dmask (channels 0 / 1, the rest 0) - the mask layer's conv-bias gradient without a pass over dmask
  Ptr colsum;
  int32_t colsum_rows, pad1_;
};

// |spec| for the CRN encoder (ConvSTFT 'real', tools_for_model.py:62-68: no eps): mags[f][MO + k] = sqrt(re^2 + im^2), k < NF;
// the MS-wide rows are zero padded; MO keeps bin 1 16-byte aligned for both storage dtypes.
struct Mags {
  Ptr spec, mags;
  int64_t frames;
  int32_t NF, MS, MO, dt;
};

// Overlap-add + 1/(coff+1e-8) + trim + clamp (tools_for_model.py:101-110, models.py:280-282)
// frames [B][T][win] fp32 ; wav [B][L] fp32 ; coff fp32 [(T-1)*hop+win] in A_CONST
struct Ola {
  Ptr frames, wav, coff;
  Ptr dwav, dpad;              // backward: dpad [B][(T-1)*hop+win] = clampmask*dwav/(coff+1e-8) (0 in the trimmed borders)
  int32_t B, T, L, win, hop, trim;
  int32_t noclamp, pad_;       // noclamp = 1: torch.istft semantics (no clamp_(-1, 1)); the ConviSTFT path of DCCRN / CRN clamps
};

// est spec [B][T][NF+1][2] (fp32, slot layout above) <-> reference layout out_real/out_imag [B][NF][T] fp32
// mode 1: out_real = |est| (magnitude of the pairs, CRN target_mags) ; mode 2: est is a plain [B*T][NF] fp32 array -> out_real
// mode 3: out_real receives the interleaved complex tensor [B][NF][T][2] (memory image of torch.complex64 [B, NF, T]);
//         SPECOUT_BWD with mode 3 is the inverse copy [B][NF][T][2] -> est (the input side of the torch.istft plan)
struct SpecOut {
  Ptr est, out_real, out_imag; // forward: est -> out_*   ; backward: dest += d(out_*)
  int32_t B, T, NF, accumulate;
  int32_t mode, pad_;
};

struct Memset {
  Ptr dst;
  int64_t bytes;
};
// Fused STFT: framing + window + 512-point FFT in LDS, one wavefront per frame (radix 8x8x8, two LDS transposes).
// frame (b, t) reads src[b][t*hop - off + n], n < 512 (0 outside [0, L)), times win[n]; writes bins 0..256 to the slot layout
// spec [B][T][258][2] (slot 0 = 0).  tw: fp32 (cos, sin)(2 pi k / 512), k < 512 ; win: fp32 [512]  (both A_CONST).
struct StftFft {
  Ptr src, spec, tw, win;
  int32_t B, L, T, hop, off;
  int32_t lp_dt;               // dtype of `lp`
  // Optional second output (plain STFT only): the channel-padded copy [frames][258][8] (re, im, 0 x 6) in the activation dtype that the
  // first encoder layer reads (16-byte runs for its LDS-DMA loader) - a separate SPECPAD pass over the spectrogram otherwise.  A_NONE: absent.
  Ptr lp;
  // Backward of the pinv synthesis (ConviSTFT, tools_for_model.py:64-112): d spec = Kinv . d frames is the SAME transform of
  // the padded waveform gradient, with a rank-2 term and the 1/256:  out[part][k] = scale * ( FFT(v)[part][k]
  //   - cE[part][k] * sum_{n even} v[n] - cO[part][k] * sum_{n odd} v[n] ),  v = windowed frame.  corr = A_NONE: plain STFT.
  Ptr corr;                    // fp32 [2 (even, odd)][2 (re, im)][257]  (A_CONST)
  float scale;
  int32_t pair;                // 1: two real frames per complex transform (bf16 plans: half the instructions); 0: one frame per transform (fp32 plans)
};
// iSTFT synthesis as an inverse 512-point FFT (fft_len == 512): frames[fr][j] = win[j] / 256 * ( Re sum_{k<=256} X[k] e^{+2 pi i k j / 512}
//   - C_parity(j) ),  C_even = sum_{part,k} X[part][k] cE[part][k], C_odd likewise with cO: the closed form of the reference's
// pinv(analysis basis) (init_kernels(invers=True), tools_for_model.py:16-33).  est [frames][258][2] -> frames [frames][W].
struct IstftFft {
  Ptr est, frames, tw, win, corr;
  int64_t nframes;
  int32_t W, pad_;
};
// torch.stft(center=True, pad_mode='reflect'): dst[b][i] = src[b][reflect(i - pad)], i < L + 2*pad
struct ReflectPad {
  Ptr src, dst;
  int32_t B, L, pad, pad_;
};

// ---------------------------------------------------------------------------------------------- FullSubNet (models.py:568-682)
// Time-major sequence tensors [T][rows][feat]: one LSTM time step is one dense GEMM over all rows (rows = B for the
// full-band model, B*257 for the sub-band model) followed by an element-wise cell kernel.
struct LstmCell {            // forward : gates slab holds pre-activations (input GEMM + recurrent GEMM) -> overwritten by i,f,g,o
  Ptr gates;                 //           fp32 [rows][4H]
  Ptr c_prev, c;             //           fp32 [rows][H]   (c_prev = A_NONE at t == 0)
  Ptr h;                     //           [rows][H] dtype hdt
  Ptr dh, dc, dgates;        // backward: dh fp32 [rows][H] (total upstream), dc fp32 [rows][H] carry (in/out), dgates out dtype gdt
  int64_t rows;
  int32_t H, hdt, gdt, first;     // first: t == 0 (no previous cell state) / backward: t == T-1 (no carry yet)
  // Strided form (G > 0), used by the per-time-step LSTM of DCCRN / CRN when rnn_units/2 > 128 (W_hh no longer fits the
  // register file of one CU): rows = G groups x Bg sequences addressed inside the batch-major [.., B*T, ..] buffers of the
  // persistent path; element (group g, sequence b) of buffer k sits at  base_k + go[k][g] + b * rs[k]  (elements of that
  // buffer's dtype; the time step is folded into base_k).  k: 0 gates (pre-activations in, i/f/g/o out, in place),
  // 1 c and c_prev, 2 h, 3 dh, 4 dgates.  unit_major: gate column order of sefd_desc.h gate_col instead of q*H + unit.
  // `dc` stays a dense [rows][H] carry.  G == 0: FullSubNet slabs (dense rows, gate-major columns).
  int32_t G, Bg, unit_major, pad_;
  int64_t rs[5];
  int64_t go[5][4];
  // kind == 1: GRU cell (torch.nn.GRU, gate order r, z, n; tools_for_model.py:748-756), dense gate-major slabs only.
  //   forward : gates[:, 0:3H] = W_ih x + b_ih (hoisted GEMM), gh [rows][3H] fp32 = W_hh h_{t-1} + b_hh (this step's GEMM),
  //             c_prev = h_{t-1} (dtype hdt; A_NONE at t == 0)  ->  h_t, and gates := (r, z, n, W_hn h + b_hn) for the backward
  //   backward: dh fp32 [rows][H] total gradient of h_t (in);  dgates [rows][3H] := d(W_ih x + b_ih), gh [rows][3H] := d(W_hh h + b_hh)
  //             (dtype gdt);  dc = dh slab of step t-1 (fp32, += dh_t * z; A_NONE at t == 0).  The recurrent GEMM adds gh . W_hh to it.
  Ptr gh;
  int32_t kind, pad2_;
};
// Inverted dropout between the LSTM layers (nn.LSTM(dropout=0.8), tools_for_model.py:746): counter-based hash RNG on
// (seed, element index); backward re-derives the mask from the same seed.  keep == 1 -> identity copy.
struct Dropout {
  Ptr x, y, seed;            // seed: two uint32 in A_IO ("io.seed"), advanced by the host every step
  int64_t n;
  float keep;
  int32_t dt, layer, pad_;   // dt of x / y ; layer id decorrelates the masks of different layers
};
// FSN_IN: noisy_mag [B][F][T] fp32 -> mag_t [TP][B][F] fp32 (zero for the look-ahead frames t >= T) and per-utterance sums.
// FSN_SCALE: fb_in[t][b][f < F] = mag_t / (sum_b / (F*TP) + 1e-5), zero padded to FP columns, dtype dt.
// FSN_SBSUM: per-utterance sum of the un-normalised sub-band input (31 reflected neighbours + full-band output).
// FSN_SBBUILD: sb_in [TP][B*F][NB+1] = that input / (mean + 1e-5).
// FSN_OUT: crm [B][F][T][2] = sb_out[t + LA][b*F + f][:] ; FSN_OUT_BWD the reverse (zero for t < LA).
// FSN_SBBWD_SUM / _APPLY: gradient of the normalised concat w.r.t. the full-band output (through value and mean), x ReLU'.
// cfg.norm_type (BaseModel.norm_wrapper, tools_for_model.py:1106-1118) = Fsn::mode: 0 offline_laplace_norm (the ops above), 1 cumulative_laplace_norm,
// 2 offline_gaussian_norm, 3 cumulative_layer_norm.  Modes 1-3 add
// FSN_NORMSTAT: statistics of the full-band input (src 0: mag_t, rows = B, F values per frame) or of the un-normalised sub-band input (src 1:
//   rows = B*F, NB+1 values per frame) into `stat`: mode 2 -> [2][B] (mean, unbiased std of the utterance); modes 1 / 3 -> [TP][rows][2]
//   (running mean over all values of the row up to and including frame t, sqrt(running variance + eps)).
// FSN_SCALE / FSN_SBBUILD with mode > 0 normalise with `stat`;  FSN_NORMBWD: gradient of the normalised sub-band input w.r.t. the full-band
//   output column (through the value and through the statistics) -> out fp32 [TP][B][F]; FSN_SBBWD_APPLY with mode > 0 takes that as `in`.
struct Fsn {
  Ptr in, out, aux, aux2, sums;
  int32_t B, F, T, TP, FP, NB, LA, dt;
  int32_t act, mode;
  Ptr stat;
  int32_t src, pad_;
};

enum OpKind : int32_t {
  OP_RUNGEMM = 1, OP_WGRAD, OP_PACK, OP_UNPACK, OP_BN_FINALIZE, OP_BN_APPLY, OP_BN_BWD_REDUCE, OP_BN_BWD_APPLY,
  OP_LSTM_FWD, OP_LSTM_BWD, OP_COMBINE_FWD, OP_COMBINE_BWD, OP_MASK_FWD, OP_MASK_BWD, OP_OLA_FWD, OP_OLA_BWD,
  OP_SPECOUT_FWD, OP_SPECOUT_BWD, OP_MEMSET, OP_SPLITSUM, OP_BN_BWD_FINALIZE, OP_MAGS,
  OP_CELL_FWD, OP_CELL_BWD, OP_DROPOUT_FWD, OP_DROPOUT_BWD, OP_FSN_IN, OP_FSN_SCALE, OP_FSN_SBSUM, OP_FSN_SBBUILD, OP_FSN_OUT,
  OP_FSN_OUT_BWD, OP_FSN_SBBWD_SUM, OP_FSN_SBBWD_APPLY, OP_REFLECTPAD,
  OP_SPECPAD,       // Mags struct reused: spec fp32 [frames][NF][2] -> mags [frames][NF][MS] (dtype dt), channels 2..MS-1 zero (NF = slots here)
  OP_STFT_FFT,
  OP_PACKMULTI,
  OP_ISTFT_FFT,
  OP_FSN_NORMSTAT, OP_FSN_NORMBWD,   // cfg.norm_type other than offline_laplace_norm (struct Fsn)
  OP_CBN_STATS, OP_CBN_FINALIZE, OP_CBN_APPLY, OP_CBN_BWD_REDUCE, OP_CBN_BWD_FINALIZE, OP_CBN_BWD_APPLY,   // ComplexBatchNorm (CbnFwd / CbnBwd)
};

constexpr int kOpHold = 2;     // Op::join of a lane-1 op
constexpr int kOpNoJoin = 4;   // Op::join of a main-stream UNPACK whose inputs were all produced on the main stream: it does not wait for the second lane

struct Op {
  int32_t kind;
  int32_t tag;                 // layer id for profiling / debugging
  int32_t lane;                // 0: critical path.  2: second stream, issued at its program position (waits for everything the main
                               // stream has been given so far); a lane-0 op with `join` set makes the main stream wait for the
                               // second one first.  3: third stream, behind the main stream's work so far and earlier lane-3 ops
                               // only; the next lane-2 op (and any join) waits for it.  1: off the critical path (weight gradients, their folds, the early UNPACK): a
                               // full-phase run holds the ones in front of the first LSTM backward back and runs them on a second HIP
                               // stream next to it (the recurrence occupies 32 of the 256 CUs); see api.hip plan_run
  int32_t join;                // lane 0: 1 = wait for the second stream first.  lane 1: kOpHold = issue behind the NEXT recurrence launch
  union {
    RunGemm g;
    Pack pack;
    Unpack unpack;
    BnFinalize bnf;
    BnApply bna;
    BnBwdReduce bnr;
    BnBwdApply bnb;
    LstmRec lstm;
    Combine comb;
    Mask mask;
    Ola ola;
    SpecOut so;
    Memset ms;
    Mags mags;
    LstmCell cell;
    Dropout drop;
    Fsn fsn;
    ReflectPad rpad;
    StftFft fft;
    IstftFft ifft;
    PackMulti packm;
    CbnFwd cbf;
    CbnBwd cbb;
  };
};

// ---- job order of the row-block recurrence launches that run as ticket-drawn jobs (lstm_rows.hip; checked on the CPU by tests/test_plan_hostsim.py)
#ifdef __HIPCC__
#define SEFD_HD __host__ __device__
#else
#define SEFD_HD
#endif
struct RowsJob { int layer, chunk, block, tb, te; };      // frames [tb, te) of row block `block`
// lstm_fwd_rows_pair_kernel: two stacked layers x C time chunks x nblk row blocks, wavefront order: segment 0 = L chunk 0; segments 1 + 2p, 2 + 2p =
// L chunk p + 1, U chunk p; last segment = U chunk C - 1.  L(c, j) reads L(c - 1, j); U(c, j) reads U(c - 1, j) and L(c, j): all in earlier segments.
SEFD_HD static inline RowsJob rows_pair_job(int job, int nblk, int C, int T) {
  RowsJob r;
  const int seg = job / nblk;
  r.block = job - seg * nblk;
  if (seg == 0) { r.layer = 0; r.chunk = 0; }
  else if (seg == 2 * C - 1) { r.layer = 1; r.chunk = C - 1; }
  else { r.layer = (seg - 1) & 1; r.chunk = (seg - 1) / 2 + (r.layer == 0 ? 1 : 0); }
  const int clen = (T + C - 1) / C;
  r.tb = r.chunk * clen < T ? r.chunk * clen : T;
  r.te = r.tb + clen < T ? r.tb + clen : T;
  return r;
}
// lstm_bwd_rows_jobs_kernel: C time chunks x nblk row blocks, chunk-major; chunk c covers frames [T - (c + 1) clen, T - c clen) and reads the carry of (c - 1, j)
SEFD_HD static inline RowsJob rows_bwd_job(int job, int nblk, int C, int T) {
  RowsJob r;
  r.layer = 0;
  r.chunk = job / nblk;
  r.block = job - r.chunk * nblk;
  const int clen = (T + C - 1) / C;
  r.te = T - r.chunk * clen > 0 ? T - r.chunk * clen : 0;
  r.tb = r.te - clen > 0 ? r.te - clen : 0;
  return r;
}

// Tile geometry shared by planner (padding) and kernels.
constexpr int kBM = 128;                                   // rows per RUNGEMM block == rows per statistics block
inline int bk_of(int dt) { return dt == DT_BF16 ? 64 : 32; }   // K-tile in elements: 128 bytes of either dtype
// WGRAD with N <= 4 outputs over ONE contiguous bf16 [M][fstride] array and a contiguous bf16 [M][y_fstride] gradient (FullSubNet's sub-band head: 2 outputs,
// 384 inputs, 3.2 M rows): a rank-N update - the planner marks it kRunRank (and sizes the row splits for a streaming kernel), rungemm.hip wgrad_rank_kernel runs it
inline bool wgrad_rank_form(const RunGemm& g) {
  if (g.xdt != DT_BF16 || g.ydt != DT_BF16 || g.N < 1 || g.N > 4 || g.nseg < 1 || g.nseg > 2) return false;
  const Seg& s = g.seg[0];
  if (s.src != 0 || s.dt != 0 || s.len % 8 != 0 || s.len < 8 || s.len > 512 || s.off % 8 != 0 || s.koff != 0) return false;
  if (g.nseg == 2 && (g.seg[1].src >= 0 || g.seg[1].koff < s.len)) return false;               // the second run can only be the bias ones run
  if (g.bstride[0] != 0 || g.base[0] != 0 || g.fstride[0] % 8 != 0 || s.off + s.len > g.fstride[0]) return false;
  if ((int64_t)g.tstride[0] != (int64_t)g.Fo * g.fstride[0] || (int64_t)g.M != (int64_t)g.Tout * g.Fo || g.Tin[0] != g.Tout) return false;
  if (g.y_bstride != 0 || (int64_t)g.y_tstride != (int64_t)g.Fo * g.y_fstride || g.y_fstride < g.N || g.y_off < 0) return false;
  return (g.x[0].off % 16) == 0 && (g.y.off % 2) == 0;
}

inline int bn_of(int N) { return N > 64 ? 128 : (N > 32 ? 64 : 32); }
constexpr int kWgTN = 64, kWgTK = 128, kWgRows = 32;       // WGRAD block tile (n x k) and reduction rows per step
// n extent of the WGRAD tile of the aligned bf16 kernel: 128 for wide layers, 32 for the thin ones (shared by the planner, which
// sizes the row splits for it, and the launcher).  Measured (MI355X, DCCRN B = 32): N = 32 layers 228 -> 210 us and 164 -> 149 us
// with the 32-wide tile; a 16-wide one made the mask layer (N = 8, M = 2 M rows) slower (318 -> 457 us: twice the workgroups, each
// bound by the same activation stream) and is not used.
static inline constexpr int wgrad_tn(int xdt, int N, int Npad) {
  return xdt != DT_BF16 ? kWgTN : Npad >= 128 ? 128 : N > 32 ? 64 : 32;
}
// Packed bf16 W_hh of the row-block LSTM kernels (lstm_rows.hip), in MFMA 16x16x32 B-fragment order: the 64 lanes of a wave read
// 64 x 16 contiguous bytes.  Forward: element (gate column c = 4 * unit + q, input k); backward: element (unit n, gate column k).
static inline int64_t rows_wf_index(int H, int c, int k) {
  const int u = c >> 2, q = c & 3, ub = u >> 4, ln = u & 15, ks = k >> 5, kq = (k & 31) >> 3, e = k & 7;
  return ((((int64_t)(ub * (H / 32) + ks) * 4 + q) * 64 + (kq * 16 + ln)) * 8) + e;
}
static inline int64_t rows_wb_index(int H, int n, int k) {
  const int nt = n >> 4, ln = n & 15, ks = k >> 5, kq = (k & 31) >> 3, e = k & 7;
  return (((int64_t)(nt * (4 * H / 32) + ks) * 64 + (kq * 16 + ln)) * 8) + e;
}
inline int esize(int dt) { return dt == DT_BF16 ? 2 : 4; }

}  // namespace sefd
