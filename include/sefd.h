/* sefd.h - C ABI of the MI355X (gfx950) speech-enhancement training hot path.
 *
 * Drop-in boundary: the reference exposes this path only as Python objects -
 *   models.DCCRN.__init__/forward/loss          /root/reference/models.py:15-323
 *   ConvSTFT / ConviSTFT                        /root/reference/tools_for_model.py:36-112
 *   ComplexConv2d / ComplexConvTranspose2d      /root/reference/tools_for_model.py:199-338
 *   NavieComplexLSTM                            /root/reference/tools_for_model.py:141-181
 *   sdr / si_snr / si_sdr / F.mse_loss          /root/reference/tools_for_loss.py:17-94, models.py:315-323
 *   torch.optim.Adam step                       /root/reference/train_interface.py:59, trainer.py:35-37
 * so this header is what a ctypes binding of those objects calls.
 * Plain pointers and sizes only; every function enqueues on the given HIP stream and returns 0 or a negative code.
 * No hidden device allocation: all device memory ("arenas") is owned by the caller.
 */
#ifndef SEFD_H_
#define SEFD_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sefd_plan sefd_plan;

/* Mirror of the config.py knobs that shape DCCRN (config.py:35-68) plus batch geometry. */
typedef struct sefd_model_config {
  int32_t model;          /* 0 DCCRN, 1 CRN, 2 ConvSTFT front end, 3 FullSubNet, 4 torch.stft front end, 5 torch.istft (see csrc/plan.cpp) */
  int32_t B, L;           /* batch, samples per clip */
  int32_t win_len, hop, fft_len;
  int32_t n_layers;
  int32_t kernel_num[12]; /* cfg.dccrn_kernel_num; FullSubNet (model 3): sb/fb neighbours, look_ahead, fb/sb hidden, fb/sb activation,
                             dropout keep in 1/1000, [8] sequence model (0 LSTM, 1 GRU), [9] norm type (0 offline_laplace_norm,
                             1 cumulative_laplace_norm, 2 offline_gaussian_norm, 3 cumulative_layer_norm) */
  int32_t rnn_layers, rnn_units;
  int32_t mask_mode;      /* 0 'E', 1 'C', 2 'R' (cfg.masking_mode) */
  int32_t lstm_complex;   /* cfg.lstm == 'complex' */
  int32_t skip;           /* cfg.skip_type */
  int32_t act_dtype;      /* 0 fp32, 1 bf16 storage / MFMA dtype */
  int32_t kernel_size;    /* 5 */
  int32_t training;       /* 1 train (batch statistics + backward plan), 0 eval */
  int32_t bn_world;       /* 0/1: BatchNorm statistics of this process only (standard DDP).  N > 1: SyncBN over N ranks - the plan
                             exposes sync points (sefd_plan_num_syncs / sefd_plan_sync) where the caller sum-all-reduces a small
                             statistics buffer between two op ranges; counts are scaled by N.  Makes N ranks x B/N utterances
                             equal to the reference's single process with batch B */
  int32_t grad_buckets;   /* 0/1: one UNPACK at the end of the backward phase.  2 (DCCRN): the gradients of decoder + LSTM (the flat
                             range [sefd_plan_grad_bucket elem, end)) are final at op `bucket op` of the backward phase, BEFORE the encoder
                             backward: a data-parallel caller starts their all-reduce there (sefd_plan_run_cb) and it rides under the
                             encoder's dgrad / wgrad kernels; the encoder range follows at the end of the phase (reverse layer order) */
  int32_t use_cbn;        /* DCCRN(use_cbn=True) (models.py:25, 76, 120): ComplexBatchNorm (tools_for_model.py:430-607) instead of nn.BatchNorm2d */
  int32_t window;         /* ConvSTFT / ConviSTFT window (tools_for_model.py:17-20): 0 periodic Hann (cfg.window = 'hanning'), 1 rectangular (win_type None),
                             2 the table `window_values` (any other scipy.signal.get_window(name, win_len, fftbins=True): the host evaluates it) */
  int32_t pad_;
  const double* window_values;   /* window == 2: win_len doubles, read during sefd_plan_create only */
} sefd_model_config;

enum { SEFD_ARENA_WS = 0, SEFD_ARENA_PARAM = 1, SEFD_ARENA_GRAD = 2, SEFD_ARENA_STATE = 3, SEFD_ARENA_CONST = 4, SEFD_ARENA_IO = 5,
       SEFD_ARENA_COUNT = 6 };
enum { SEFD_PHASE_FWD = 0, SEFD_PHASE_BWD = 1 };
enum { SEFD_LOSS_MSE = 0, SEFD_LOSS_SDR = 1, SEFD_LOSS_SISNR = 2, SEFD_LOSS_SISDR = 3 };

/* ---- plan life cycle (host only) ---------------------------------------------------------------- */
sefd_plan* sefd_plan_create(const sefd_model_config* cfg);        /* replaces models.DCCRN.__init__ (models.py:17-172) */
void sefd_plan_destroy(sefd_plan* p);
const char* sefd_plan_error(const sefd_plan* p);                  /* "" when the plan is valid */
int64_t sefd_plan_arena_bytes(const sefd_plan* p, int arena);
int32_t sefd_plan_frames(const sefd_plan* p);                     /* T */
/* parameters in reference state_dict order; kind 0 = trainable (A_PARAM / A_GRAD), 1 = buffer (A_STATE) */
int32_t sefd_plan_num_params(const sefd_plan* p, int kind);
const char* sefd_plan_param_name(const sefd_plan* p, int kind, int i);
int64_t sefd_plan_param_offset(const sefd_plan* p, int kind, int i);   /* in elements */
int64_t sefd_plan_param_numel(const sefd_plan* p, int kind, int i);
int32_t sefd_plan_param_shape(const sefd_plan* p, int kind, int i, int64_t* shape4);  /* returns ndim */
/* named workspace / io buffers (tests, debugging). returns 0 if found */
int32_t sefd_plan_buffer(const sefd_plan* p, const char* name, int32_t* arena, int64_t* off, int64_t* bytes, int32_t* dtype);
/* SyncBN (bn_world > 1): sync point i says "after op `op` of `phase` has been enqueued, sum-all-reduce `count` elements
   (dtype 0 fp32 / 1 fp64) at byte offset `off` of arena `arena` over the ranks, then continue with op + 1". */
int32_t sefd_plan_num_syncs(const sefd_plan* p);
int32_t sefd_plan_sync(const sefd_plan* p, int i, int32_t* phase, int32_t* op, int32_t* arena, int64_t* off, int64_t* count, int32_t* dtype);
int32_t sefd_plan_num_buffers(const sefd_plan* p);
const char* sefd_plan_buffer_name(const sefd_plan* p, int i);
/* host image of the constant arena (STFT bases, OLA normaliser, index tables): copy it to device memory once */
const void* sefd_plan_const_data(const sefd_plan* p);
/* op lists (POD array of sefd::Op, see csrc/sefd_desc.h) */
int32_t sefd_plan_num_ops(const sefd_plan* p, int phase);
const void* sefd_plan_ops(const sefd_plan* p, int phase);
int32_t sefd_op_size(void);
/* per-op summary for measurement: out = {kind, tag, M, N, K (true run length), operand dtype, algorithmic flops, algorithmic bytes} */
int32_t sefd_plan_op_info(const sefd_plan* p, int phase, int i, int64_t* out8);

/* ---- execution (device) --------------------------------------------------------------------------
 * arenas: SEFD_ARENA_COUNT device pointers.  Runs ops [first, last) of the phase on `stream`.
 * Forward  = DCCRN.forward  (models.py:176-284): IO.wav -> IO.out_wav, IO.out_real, IO.out_imag.
 * Backward = autograd of it: IO.grad_wav / grad_real / grad_imag -> A_GRAD (all parameters).
 * Returns 0, or < 0: -1 invalid plan, -2 a launch failed, -3 stream / event creation failed, -5 a cluster-LSTM launch of an EARLIER
 * call gave up waiting for a peer workgroup (bounded spin ran out: co-residency lost on a shared / preempted GPU) - that step's
 * results are invalid; the flag is cleared by the call that reports it. */
int32_t sefd_plan_run(const sefd_plan* p, int phase, int first, int last, void* const* arenas, void* stream);
/* grad_buckets == 2: index (backward phase) of the UNPACK op that completes the first gradient bucket and the first flat element
   of that bucket; returns -1 when the plan has a single bucket. */
int32_t sefd_plan_grad_bucket(const sefd_plan* p, int32_t* op, int64_t* elem);
/* The same as an explicit element range [lo, hi) of the flat gradient arena: DCCRN / CRN complete the TAIL of the arena first (decoder + LSTM,
 * hi = end), FullSubNet the FRONT (the full-band model: its weight gradients run on the main stream while the sub-band model's still
 * occupy the second lane, lo = 0). */
int32_t sefd_plan_grad_bucket_range(const sefd_plan* p, int32_t* op, int64_t* lo, int64_t* hi);
/* Whole phase as sefd_plan_run(first = 0, last = -1), and `cb(ctx)` is called on the host right after op `at` has been enqueued on
   `stream` (everything up to and including that op is ordered before whatever the callback enqueues behind an event on `stream`). */
int32_t sefd_plan_run_cb(const sefd_plan* p, int phase, void* const* arenas, void* stream, int at, void (*cb)(void*), void* ctx);
/* The same with run flags (cb may be NULL, at = -1).  SEFD_RUN_WAVE_ONLY: the caller's loss is a function of the enhanced WAVEFORM only
 * (trainer.py:27-39 without a perceptual term): the reference-layout copies out_real / out_imag [B][NF][T] are not produced and the
 * (all-zero) gradient with respect to them is not accumulated - DCCRN.forward's first two return values are not valid after such a run. */
#define SEFD_RUN_WAVE_ONLY 1
int32_t sefd_plan_run_flags(const sefd_plan* p, int phase, void* const* arenas, void* stream, int flags, int at, void (*cb)(void*), void* ctx);
/* Measurement only (bench.py's in-situ roofline leg): the whole phase in its real two-stream schedule with a HIP event pair around every
   op on the stream it is launched on; synchronises; ms[i] = duration of op i while the other lane runs beside it.  n >= number of ops. */
int32_t sefd_plan_run_timed(const sefd_plan* p, int phase, void* const* arenas, void* stream, float* ms, int32_t n);

/* ---- losses (tools_for_loss.py:17-94, models.py:315-323) ---------------------------------------
 * est, tgt: fp32 [B][L] device.  ws: fp32 device scratch of sefd_loss_ws_floats(B) floats.
 * forward writes the scalar loss to loss_out[0]; backward writes grad_est[B][L] = d(loss)/d(est) * grad_scale[0]. */
int64_t sefd_loss_ws_floats(int32_t B);
int32_t sefd_loss_forward(int kind, const float* est, const float* tgt, int32_t B, int32_t L, float* ws, float* loss_out, void* stream);
int32_t sefd_loss_backward(int kind, const float* est, const float* tgt, int32_t B, int32_t L, const float* ws,
                           const float* grad_scale, float* grad_est, void* stream);

/* Short rows: FullSubNet.loss (models.py:674-682) applies the same four losses to cRM / cIRM tensors [B, F, T, 2], whose last axis - the
 * reduction axis of tools_for_loss.py:17-94 - has two elements, and trainer.py:107 passes the network output in the `target` slot.
 * est, tgt: fp32 [R][L] device, L <= 16, same argument roles as above (SDR: est = s2, tgt = s1; SI-SNR: est = s1, tgt = s2; SI-SDR:
 * est = estimation, tgt = reference).  backward writes d(loss)/d(est) and / or d(loss)/d(tgt) (either pointer may be NULL), times grad_scale[0]. */
int64_t sefd_loss_rows_ws_floats(int64_t R);
int32_t sefd_loss_rows_forward(int kind, const float* est, const float* tgt, int64_t R, int32_t L, float* ws, float* loss_out, void* stream);
int32_t sefd_loss_rows_backward(int kind, const float* est, const float* tgt, int64_t R, int32_t L, const float* ws, const float* grad_scale,
                                float* grad_est, float* grad_tgt, void* stream);

/* SI-SDR under data parallelism (no counterpart in the reference, which is single-process; the arithmetic it must reproduce is
 * tools_for_loss.py:91-94: `ratio = torch.mean(ratio)` over the WHOLE batch, then the log).  After sefd_loss_forward / sefd_loss_rows_forward
 * with kind SI-SDR, ws + sefd_loss_dp_offset(n, rows) holds this rank's { sum of ratios, row count } (n = B, or R with rows = 1): the host
 * sum-all-reduces those TWO floats in place over the ranks and calls sefd_loss_dp_finish, which rewrites loss_out[0] to the global-batch
 * loss and re-scales the saved row coefficients so that the following sefd_loss_*_backward yields world x d(global loss)/d(est) for this
 * rank's rows (the gradient exchange sums over ranks, Adam applies 1 / world). */
int64_t sefd_loss_dp_offset(int64_t n, int32_t rows);
int32_t sefd_loss_dp_finish(int32_t rows, int64_t n, float* ws, int32_t world, float* loss_out, void* stream);

/* ---- LMS log-mel perceptual loss (tools_for_loss.py:120-249 + the magnitude step of models.py:306-312) ------------
 * clean_* / est_*: fp32 [B][NF][T] device (reference layout).  If the *_i pointers are NULL the *_r arrays are magnitudes
 * already (get_array_lms_loss(clean_mags, est_mags) signature); otherwise mag = sqrt(r^2 + i^2 + 1e-7) is fused in.
 * bands: int32 [nbands][4] = {first bin, taps, weight offset, scale index}; weights: the triangle taps (melFilterBank);
 * scale_sizes_host: HOST int32[nscales] = filters per scale (16, 32, 64).  rowloss_ws: fp32 [B*T] device scratch. */
int32_t sefd_lms_forward(const float* clean_r, const float* clean_i, const float* est_r, const float* est_i, int32_t B, int32_t NF, int32_t T,
                         const int32_t* bands, const float* weights, int32_t nbands, const int32_t* scale_sizes_host, int32_t nscales,
                         int32_t nfft, float* rowloss_ws, float* loss_out, void* stream);
int32_t sefd_lms_backward(const float* clean_r, const float* clean_i, const float* est_r, const float* est_i, int32_t B, int32_t NF, int32_t T,
                          const int32_t* bands, const float* weights, int32_t nbands, const int32_t* scale_sizes_host, int32_t nscales,
                          int32_t nfft, const float* grad_scale, float* grad_est_r, float* grad_est_i, void* stream);

/* ---- PMSQE perceptual loss (call chain tools_for_loss.py:253-269 `get_array_pmsqe_loss`, models.py:313-314) ----------------------
 * The arithmetic is third-party (asteroid SingleSrcPMSQE + PITLossWrapper('pw_pt') + asteroid_filterbanks STFTFB / Encoder / mag), absent
 * from the reference tree and unversioned there: PARITY UNPINNED; this is the published algorithm as oracle/pmsqe.py restates it.
 * est / clean: fp32 [B][L] device waves, L a whole number of seconds (the reference's view(N, -1, fs)), at most 6 seconds (PIT over
 * the seconds enumerates the permutations).  power: 1 = the loss works on the power spectrum re^2 + im^2 (the paper's definition; the host default),
 * 0 = on the magnitude sqrt(re^2 + im^2 + 1e-8) that transforms.mag hands over in the reference call chain, taken literally.
 * tab: fp32 [sefd_pmsqe_table_floats()] device = thr[49] zp[49] width[49] corr[49] aterm[49] mask[257] (at 245), then at 512 the
 * windowed DFT tables cos [512][257], -sin [512][257] and their transposes [257][512] x 2;  itab: int32 [64 + 257] device = prefix sums
 * of the FFT bins per Bark band [50] and, at 64, the band of every bin (-1: none).  ws: fp32 [sefd_pmsqe_ws_floats(B, L)] device, must
 * stay untouched between forward and backward.  backward writes d loss / d est * *grad_scale into grad_est [B][L]. */
int64_t sefd_pmsqe_table_floats(void);
int64_t sefd_pmsqe_ws_floats(int32_t B, int32_t L);
int32_t sefd_pmsqe_forward(const float* est, const float* clean, int32_t B, int32_t L, int32_t power, const float* tab, const int32_t* itab,
                           float* ws, float* loss_out, void* stream);
int32_t sefd_pmsqe_backward(int32_t B, int32_t L, int32_t power, const float* tab, const int32_t* itab, float* ws, const float* grad_scale,
                            float* grad_est, void* stream);

/* ---- FullSubNet training targets (trainer.py:100-104; tools_for_model.py:683-717) -------------------------------------
 * noisy_c64 / clean_c64: interleaved complex64 [n] (the torch.stft outputs).  Any of mag / phase / cirm may be NULL.
 * mag = |noisy| (mag_phase), phase = angle(noisy), cirm [n][2] = compress_cIRM(build_complex_ideal_ratio_mask(noisy, clean)). */
/* On-GPU SNR mixing of a batch (replaces the offline generate_noisy_data.py:46-67 `generate_noisy_wav`, the random segment start chosen by
   the caller): noisy[b] = speech[b] + alpha_b * noise[noise_start[b] : +L], alpha_b = sqrt(10^(-snr_db[b]/10) * P_speech / (P_noise + 1e-6)) with
   the powers taken after removing the DC bias; quantize != 0 adds the reference's int16 file round trip.  speech / noisy [B][L] fp32, noise a
   flat fp32 bank, ws: 4 * B doubles.  All device pointers. */
int32_t sefd_mix_snr(const float* speech, const float* noise, const int64_t* noise_start, const float* snr_db, int32_t B, int32_t L,
                     int32_t quantize, double* ws, float* noisy, void* stream);
int32_t sefd_fsn_targets(const float* noisy_c64, const float* clean_c64, int64_t n, float* mag, float* phase, float* cirm, void* stream);

/* ---- Adam (torch.optim.Adam defaults, train_interface.py:59) on flat fp32 buffers ------------------
 * step is 1-based. */
int32_t sefd_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step,
                       float lr, float beta1, float beta2, float eps, float grad_scale, void* stream);
/* The same update, skipped entirely (parameters and moments untouched) while *skip_if_set != 0.  skip_if_set = sefd_plan_status_word(plan)
 * of the plan that produced `grad`: a kernel of that plan that had to give up (bounded hand-over waits of the cluster LSTM kernels on a
 * shared / preempted GPU) sets the word, so a garbage gradient never reaches the parameters; NULL = unconditional. */
int32_t sefd_adam_step_guarded(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step,
                               float lr, float beta1, float beta2, float eps, float grad_scale, const int32_t* skip_if_set, void* stream);
/* Per-plan status word (0 = fine), kept twice: host-mapped (what sefd_plan_run / sefd_plan_status read) and in device memory (what
 * sefd_plan_status_word returns, for sefd_adam_step_guarded).  sefd_plan_run returns -5 while it is set (sticky); sefd_plan_status reads
 * it (after a stream synchronisation for a definite answer) and optionally clears both copies; sefd_plan_status_set sets both from the
 * host (test hook: what a kernel that gives up does). */
const int32_t* sefd_plan_status_word(const sefd_plan* p);
/* Data parallel: a rank whose status word is set must not hand its gradient to the others as if it were good, and the replicas must take the
 * same decision.  sefd_plan_status_poison (enqueued behind the backward, in front of the last gradient all-reduce) overwrites *grad_elem - an
 * element of that last bucket - with NaN when the word is set; the sum carries the NaN to every rank; sefd_adam_step_guarded_dp skips the
 * update on every rank while *skip_if_nan is NaN (and, as before, while *skip_if_set != 0).  No extra collective. */
int32_t sefd_plan_status_poison(const sefd_plan* p, float* grad_elem, void* stream);
int32_t sefd_adam_step_guarded_dp(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step, float lr, float beta1,
                                  float beta2, float eps, float grad_scale, const int32_t* skip_if_set, const float* skip_if_nan, void* stream);
int32_t sefd_plan_status(const sefd_plan* p, int32_t clear);
int32_t sefd_plan_status_set(const sefd_plan* p);

/* ---- tuning table ------------------------------------------------------------------------------------
 * The planner's and the launchers' tuning knobs (tile thresholds, ring depths, lane placement, ...: INTEGRATION.md section 6) live in ONE process-wide
 * table, not in the environment: a plan is a function of its sefd_model_config and of this table at the moment sefd_plan_create runs.  The table is
 * filled once from the single environment variable SEFD_TUNING="KNOB=value,KNOB=value" (read the first time the table is consulted) and by these
 * calls.  value NULL unsets a knob; sefd_tuning_get returns NULL for a knob that is not set (then the built-in default applies).  The reference has
 * no counterpart (its behaviour is fixed by config.py); nothing here changes results beyond floating-point summation order. */
void sefd_tuning_set(const char* knob, const char* value);
const char* sefd_tuning_get(const char* knob);
void sefd_tuning_clear(void);

#ifdef __cplusplus
}
#endif
#endif
