"""Oracle (test infrastructure): the reference's offline SNR mixing restated (generate_noisy_data.py:46-67 `generate_noisy_wav`), with the
random segment start passed in instead of drawn.  Pinned by running the reference function itself under a patched np.random.randint
(tests/test_oracle_mixing.py, build container only) - the GPU test compares sefd_amd.dataloader.mix_snr with this."""
import numpy as np


def generate_noisy_wav(wav_speech, wav_noise, snr, start, quantize=True):
    n = wav_noise[start:start + len(wav_speech)]
    pow_speech = np.mean(np.power(wav_speech - np.mean(wav_speech), 2.0))
    pow_noise = np.mean(np.power(n - np.mean(n), 2.0))
    alpha = np.sqrt(10.0 ** (float(-snr) / 10.0) * pow_speech / (pow_noise + 1e-6))
    noisy = (wav_speech + alpha * n) * 32768
    return noisy.astype(np.int16) if quantize else noisy / 32768
