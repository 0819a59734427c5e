"""Configuration surface: the names, defaults and two asserts of the reference's config.py (config.py:1-107), because this
module IS the flag system its callers read (`import config as cfg`).

Like there, it is a module of plain constants; constructor defaults of the models are bound at import time and
`cfg.loss` / `cfg.perceptual` / `cfg.lstm` / `cfg.skip_type` / `cfg.dccrn_kernel_num` are read at construction / call time.
Differences: no banner print; build-side knobs (`act_dtype`, `pmsqe_power`); data-file paths for the loader (placeholders in the
reference's dataloader.py).  Under DDP BatchNorm statistics are per rank unless `GradientExchange(sync_bn=True)`.
"""
# paths (config.py:11-16)
job_dir = './models/'
logs_dir = './logs/'
chkpt_model = None          # directory name under job_dir of a run to resume
chkpt = str("EPOCH")
if chkpt_model is not None:
    chkpt_path = job_dir + chkpt_model + '/chkpt_' + chkpt + '.pt'

# option lists the current setting indexes into (config.py:22-27)
model_list = ['DCCRN', 'CRN', 'FullSubNet']
loss_list = ['MSE', 'SDR', 'SI-SNR', 'SI-SDR']
perceptual_list = [False, 'LMS', 'PMSQE']
lstm_type = ['real', 'complex']
main_net = ['LSTM', 'GRU']
mask_type = ['Direct(None make)', 'E', 'C', 'R']

expr_num = 'EXPERIMENT_NUMBER'
DEVICE = 'cuda'             # the MI355X; there is no CPU execution path

# current setting (config.py:35-50)
model = model_list[0]
loss = loss_list[1]
perceptual = perceptual_list[0]
lstm = lstm_type[1]
sequence_model = main_net[0]
masking_mode = mask_type[1]
skip_type = True
max_epochs = 100
learning_rate = 0.001
batch = 10
dccrn_kernel_num = [32, 64, 128, 256, 256, 256]

# front end: 25 ms window, 6.25 ms hop at 16 kHz, 512-point transform (config.py:54-61)
fs = 16000
win_len = 400
win_inc = 100
ola_ratio = 0.75
fft_len = 512
sam_sec = fft_len / fs
frm_samp = fs * (fft_len / fs)
window = 'hanning'

rnn_layers = 2              # DCCRN
rnn_units = 256
rnn_input_size = 512        # CRN

# FullSubNet (config.py:70-81)
sb_num_neighbors = 15
fb_num_neighbors = 0
num_freqs = fft_len // 2 + 1
look_ahead = 2
fb_output_activate_function = "ReLU"
sb_output_activate_function = None
fb_model_hidden_size = 512
sb_model_hidden_size = 384
weight_init = False
norm_type = "offline_laplace_norm"
num_groups_in_drop_band = 2

# ---- build-side knobs (not in the reference)
act_dtype = 'fp32'          # 'fp32' (parity mode) or 'bf16' (storage + MFMA operand dtype, fp32 accumulate)
pmsqe_power = False         # False (default) = the reference's literal call chain: PMSQE is handed transforms.mag(...) magnitudes (tools_for_loss.py:267-269).
                            # True = opt-in build-side variant on power spectra |X|^2 (the published loss's own definition; +0.29 PESQ on the synthetic held-out set, profiles/r02_heldout_eval.json)
train_data_path = None      # [N, 2, L] .npy files of (noisy, clean) pairs; dataloader.py:63-71 has placeholder paths
valid_data_path = None
test_data_path = None

# combinations the reference refuses at import time (config.py:86-89)
assert not (masking_mode == 'Direct(None make)' and perceptual is not False), "This setting is not created "
assert not (model == 'FullSubNet' and perceptual is not False), "This setting is not created "
