/* sefd_scorers.h - C ABI of the host-side objective scorers used by the validation loop.
 *
 * Replaces, for the path `trainer.model_validate` -> `cal_pesq` / `cal_stoi` (reference trainer.py:212-222):
 *   tools_for_estimate.py:91-99  cal_stoi  -> pystoi.stoi(clean, estimated, cfg.fs, extended=False)   (third-party, not vendored)
 *   tools_for_estimate.py:51-84  cal_pesq  -> ctypes call into the prebuilt x86 `PESQ.so` (ITU-T P.862 + P.862.2, no source)
 * Plain pointers and sizes, float32 waveforms in, float64 scores out, multi-threaded over the utterances of a batch. */
#ifndef SEFD_SCORERS_H_
#define SEFD_SCORERS_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* STOI (Taal et al. 2011) of B utterance pairs, clean / est row-major [B][n] at sampling rate fs; out[B].  nthreads <= 0: all cores. */
int32_t sefd_stoi_batch(const float* clean, const float* est, int32_t B, int32_t n, int32_t fs, double* out, int32_t nthreads);
/* Wide-band PESQ (ITU-T P.862 + P.862.2) MOS-LQO of B pairs at fs = 16000 (anything else: -1), n >= 512 samples each.  Replaces
   tools_for_estimate.py:68-84 (run_pesq_waveforms / cal_pesq -> PESQ.so pesq(clean f64[n], degraded f64[n], n, n)).  One utterance, one
   delay (see csrc_host/pesq.cpp); pinned to PESQ.so outputs on 34 pairs within 0.01 MOS. */
int32_t sefd_pesq_batch(const float* clean, const float* deg, int32_t B, int32_t n, int32_t fs, double* out, int32_t nthreads);
#ifdef __cplusplus
}
#endif
#endif
