"""Configuration surface - same names, defaults and asserts as the reference's config.py (config.py:1-107).

Like the reference, it is a module of constants imported as `cfg`; constructor defaults of the models are bound at
import time and `cfg.loss` / `cfg.perceptual` / `cfg.lstm` / `cfg.skip_type` / `cfg.dccrn_kernel_num` are read at
construction / call time.  Differences: no banner print, DEVICE defaults to 'cuda' (the MI355X), and one build-side
knob (`act_dtype`) that the reference does not have.  (BatchNorm statistics are per rank under DDP; SyncBN is not built.)
"""
_CHOICES = dict(                      # the option lists the reference indexes into (config.py:35-45)
    model_list=['DCCRN', 'CRN', 'FullSubNet'], loss_list=['MSE', 'SDR', 'SI-SNR', 'SI-SDR'],
    perceptual_list=[False, 'LMS', 'PMSQE'], lstm_type=['real', 'complex'], main_net=['LSTM', 'GRU'],
    mask_type=['Direct(None make)', 'E', 'C', 'R'])

_DEFAULTS = [
    # (name, value)                                         what reads it
    ('job_dir', './models/'), ('logs_dir', './logs/'),    # checkpoint / log roots of train_interface.py
    ('chkpt_model', None), ('chkpt', 'EPOCH'), ('expr_num', 'EXPERIMENT_NUMBER'),
    ('DEVICE', 'cuda'),                                   # the MI355X; there is no CPU execution path
    ('model', _CHOICES['model_list'][0]), ('loss', _CHOICES['loss_list'][1]), ('perceptual', _CHOICES['perceptual_list'][0]),
    ('lstm', _CHOICES['lstm_type'][1]), ('sequence_model', _CHOICES['main_net'][0]),
    ('masking_mode', _CHOICES['mask_type'][1]), ('skip_type', True),
    ('max_epochs', 100), ('learning_rate', 0.001), ('batch', 10),
    ('dccrn_kernel_num', [32, 64, 128, 256, 256, 256]),
    # front end: 25 ms window, 6.25 ms hop at 16 kHz, 512-point transform
    ('fs', 16000), ('win_len', 400), ('win_inc', 100), ('ola_ratio', 0.75), ('fft_len', 512), ('window', 'hanning'),
    ('rnn_layers', 2), ('rnn_units', 256), ('rnn_input_size', 512),
    # FullSubNet
    ('sb_num_neighbors', 15), ('fb_num_neighbors', 0), ('look_ahead', 2),
    ('fb_output_activate_function', 'ReLU'), ('sb_output_activate_function', None),
    ('fb_model_hidden_size', 512), ('sb_model_hidden_size', 384), ('weight_init', False),
    ('norm_type', 'offline_laplace_norm'), ('num_groups_in_drop_band', 2),
    # build-side knob (not in the reference): 'fp32' = parity mode, 'bf16' = storage + MFMA dtype of the conv stack
    ('act_dtype', 'fp32'),
]
globals().update(_CHOICES)
globals().update(dict(_DEFAULTS))
sam_sec = fft_len / fs                 # noqa: F821  (derived values, config.py:62-63)
frm_samp = fs * sam_sec                # noqa: F821
num_freqs = fft_len // 2 + 1           # noqa: F821

# combinations the reference refuses at import time (config.py:86-89)
for _bad, _why in (((masking_mode == 'Direct(None make)' and perceptual is not False), "spectral mapping has no perceptual trainer"),   # noqa: F821
                   ((model == 'FullSubNet' and perceptual is not False), "FullSubNet has no perceptual trainer")):                    # noqa: F821
    assert not _bad, _why
