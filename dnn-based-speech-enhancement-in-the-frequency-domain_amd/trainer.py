"""Step driver mirroring the reference's trainer.py train functions (trainer.py:15-82).

`model_train(model, optimizer, train_loader, DEVICE)` has the reference's signature and epoch semantics (mean of the
per-batch losses).  With `sefd_amd.optim.Adam` the whole batch step is the fused HIP path (`model.train_step`);
with any other optimizer it is the literal reference loop (autograd Functions over the same HIP kernels)."""
import functools
import os

import torch

from . import config as cfg
from . import tools_for_loss as tfl
from .optim import Adam


def _sharded(fn):
    """A train function under `exchange`: for its duration the losses know the batch is sharded (tools_for_loss.set_data_parallel: SI-SDR's
    mean of ratios inside the log runs over all ranks).  Validation runs outside it - every rank scores its own, possibly ragged, shard."""
    @functools.wraps(fn)
    def run(model, optimizer, train_loader, DEVICE, exchange=None):
        prev = tfl.set_data_parallel(exchange)
        try:
            return fn(model, optimizer, train_loader, DEVICE, exchange=exchange)
        finally:
            tfl.set_data_parallel(prev)
    return run



@_sharded
def model_train(model, optimizer, train_loader, DEVICE, exchange=None):
    """trainer.py:15-42.  `exchange` (sefd_amd.ddp.GradientExchange, optional, not in the reference): data-parallel run."""
    train_loss = torch.zeros((), device=DEVICE)
    batch_num = 0
    model.train()
    fused = isinstance(optimizer, Adam)
    for inputs, targets in train_loader:
        batch_num += 1
        inputs = inputs.float().to(DEVICE, non_blocking=True)
        targets = targets.float().to(DEVICE, non_blocking=True)
        if fused:
            loss = model.train_step(inputs, targets, optimizer, exchange=exchange)
        else:
            _, _, outputs = model(inputs, targets)
            loss = model.loss(outputs, targets)
            optimizer.zero_grad()
            loss.backward()
            _exchange_grads(model, exchange, optimizer)
            optimizer.step()
        train_loss += loss.detach()          # the reference accumulates the graph-attached tensor (trainer.py:39)
    return train_loss / max(batch_num, 1)


@_sharded
def model_perceptual_train(model, optimizer, train_loader, DEVICE, exchange=None):
    """trainer.py:45-82: loss = (main + perceptual) / 2, forward called without targets.  With `sefd_amd.optim.Adam` the batch is the
    fused `train_step(perceptual=cfg.perceptual)` (same numbers as the autograd route below, tests/test_gpu_validate.py), which is also
    the data-parallel path (`exchange`)."""
    train_loss = train_main = train_perc = 0
    batch_num = 0
    model.train()
    fused = isinstance(optimizer, Adam)
    for inputs, targets in train_loader:
        batch_num += 1
        inputs = inputs.float().to(DEVICE)
        targets = targets.float().to(DEVICE)
        if fused:
            loss = model.train_step(inputs, targets, optimizer, exchange=exchange, perceptual=cfg.perceptual)
            main_loss, perceptual_loss = model._last_loss_parts
        else:
            real_spec, img_spec, outputs = model(inputs)
            main_loss = model.loss(outputs, targets)
            perceptual_loss = model.loss(outputs, targets, real_spec, img_spec, perceptual=True)
            loss = (main_loss + perceptual_loss) / 2
            optimizer.zero_grad()
            loss.backward()
            _exchange_grads(model, exchange, optimizer)
            optimizer.step()
        train_loss += loss.detach()
        train_main += main_loss.detach()
        train_perc += perceptual_loss.detach()
    n = max(batch_num, 1)
    return train_loss / n, train_main / n, train_perc / n


def _exchange_grads(model, exchange, optimizer=None):
    """Data-parallel step of the `loss.backward()` route: p.grad <- mean over ranks (ddp.GradientExchange.all_reduce_autograd).  The mean is
    formed here, so a fused-step 1/world left in the sefd Adam (models.train_step sets optimizer.grad_scale) must not be applied again."""
    if exchange is not None and exchange.active:
        exchange.all_reduce_autograd(list(model.parameters()))
        if optimizer is not None and hasattr(optimizer, "grad_scale"):
            optimizer.grad_scale = 1.0


@_sharded
def fullsubnet_train(model, optimizer, train_loader, DEVICE, exchange=None):
    """trainer.py:85-118."""
    from . import tools_for_model as tools
    train_loss = torch.zeros((), device=DEVICE)
    batch_num = 0
    model.train()
    fused = isinstance(optimizer, Adam)
    for inputs, targets in train_loader:
        batch_num += 1
        inputs = inputs.float().to(DEVICE)
        targets = targets.float().to(DEVICE)
        if fused:
            loss = model.train_step(inputs, targets, optimizer, exchange=exchange)
        else:
            noisy_complex = tools.stft(inputs)
            clean_complex = tools.stft(targets)
            noisy_mag, _ = tools.mag_phase(noisy_complex)
            cIRM = tools.build_complex_ideal_ratio_mask(noisy_complex, clean_complex)
            cRM = model(noisy_mag)
            loss = model.loss(cIRM, cRM)
            optimizer.zero_grad()
            loss.backward()
            _exchange_grads(model, exchange, optimizer)
            optimizer.step()
        train_loss += loss.detach()
    return train_loss / max(batch_num, 1)


@_sharded
def dccrn_direct_train(model, optimizer, train_loader, DEVICE, exchange=None):
    """trainer.py:121-150 (spectral mapping): loss = (loss(real) + loss(imag)) / 2 on the spectra.  With `sefd_amd.optim.Adam` the batch is the
    fused `model.train_step` (the spectral losses and their gradients go straight into the plan's spectrum-gradient inputs; same numbers as the
    autograd route below, tests/test_gpu_validate.py), which is also the bucketed / overlapped data-parallel path."""
    train_loss = torch.zeros((), device=DEVICE)
    batch_num = 0
    model.train()
    fused = isinstance(optimizer, Adam) and str(getattr(model, "masking_mode", "")).startswith("Direct")
    for inputs, targets in train_loader:
        batch_num += 1
        inputs = inputs.float().to(DEVICE)
        targets = targets.float().to(DEVICE)
        if fused:
            train_loss += model.train_step(inputs, targets, optimizer, exchange=exchange).detach()
            continue
        output_real, target_real, output_imag, target_imag, _ = model(inputs, targets)
        real_loss = model.loss(output_real, target_real)
        imag_loss = model.loss(output_imag, target_imag)
        loss = (real_loss + imag_loss) / 2
        optimizer.zero_grad()
        loss.backward()
        _exchange_grads(model, exchange, optimizer)
        optimizer.step()
        train_loss += loss.detach()
    return train_loss / max(batch_num, 1)


@_sharded
def crn_direct_train(model, optimizer, train_loader, DEVICE, exchange=None):
    """trainer.py:150-181: CRN spectral mapping ('Direct(None make)'): the loss compares the mapped magnitudes (first output
    of `CRN.forward`) with the target magnitudes; the waveform output carries no loss.  Fused `train_step` with `sefd_amd.optim.Adam`."""
    train_loss = torch.zeros((), device=DEVICE)
    batch_num = 0
    model.train()
    fused = isinstance(optimizer, Adam) and str(getattr(model, "masking_mode", "")).startswith("Direct")
    for inputs, targets in train_loader:
        batch_num += 1
        inputs = inputs.float().to(DEVICE, non_blocking=True)
        targets = targets.float().to(DEVICE, non_blocking=True)
        if fused:
            train_loss += model.train_step(inputs, targets, optimizer, exchange=exchange).detach()
            continue
        output_mag, target_mag, _ = model(inputs, targets)
        loss = model.loss(output_mag, target_mag)
        optimizer.zero_grad()
        loss.backward()
        _exchange_grads(model, exchange, optimizer)
        optimizer.step()
        train_loss += loss.detach()
    return train_loss / max(batch_num, 1)


# ------------------------------------------------------------------------------------------------ validation
SCORE_FILE_SUFFIX = ""            # ".rank<r>" on the ranks > 0 of a data-parallel run
LAST_VALIDATE_BATCHES = 0         # batches the last _validate call of this process went through


def _default_scorers():
    """(cal_pesq, cal_stoi) of sefd_amd.tools_for_estimate (reference tools_for_estimate.py:68-99 calls a closed x86 PESQ.so and
    pystoi): the C++ scorers of libsefd_scorers.so (wide-band P.862 PESQ, STOI).  None only when the library has not been built."""
    from . import tools_for_estimate as te
    if not os.path.exists(te.LIB_PATH):
        return None
    te.lib()                                     # a library that exists but does not load is an error, not "no scorers"
    return te.cal_pesq, te.cal_stoi


def _validate(model, validation_loader, writer, dir_to_save, epoch, DEVICE, batch_fn, n_losses, scorers):
    """Shared body of the reference's five validate functions (trainer.py:188-483): eval-mode plans (BatchNorm running
    statistics), no gradients, per-utterance PESQ / STOI lines in `<dir_to_save>/Epoch_<epoch>_SCORES`, `writer.log_wav` of the
    last batch's first utterance every 10th epoch, means over the batches.  `batch_fn(inputs, targets)` returns
    (tuple of n_losses loss tensors, enhanced waveforms [B, L])."""
    import numpy as np
    if scorers == "default":
        scorers = _default_scorers()
    sums = [torch.zeros((), device=DEVICE) for _ in range(n_losses)]
    avg_pesq = avg_stoi = 0.0
    batch_num = 0
    # data parallel: every rank scores its own shard into its own file (rank 0 keeps the reference's file name, SCORE_FILE_SUFFIX is set by
    # train_interface.run); the epoch driver then averages the scores over the ranks, weighted by their batch counts (LAST_VALIDATE_BATCHES)
    f_score = open(f"{dir_to_save}/Epoch_{epoch:d}_SCORES{SCORE_FILE_SUFFIX}", "a") if scorers is not None else None
    was_training = model.training
    model.eval()
    last = None
    try:
        with torch.no_grad():
            for inputs, targets in validation_loader:
                batch_num += 1
                inputs = inputs.float().to(DEVICE, non_blocking=True)
                targets = targets.float().to(DEVICE, non_blocking=True)
                losses, outputs = batch_fn(inputs, targets)
                for acc, l in zip(sums, losses):
                    acc += l
                last = (inputs, targets, outputs)
                if scorers is not None:
                    est, clean = outputs.cpu().numpy(), targets.cpu().numpy()
                    pesq, stoi = np.asarray(scorers[0](est, clean)).reshape(-1), np.asarray(scorers[1](est, clean)).reshape(-1)
                    for p, s in zip(pesq, stoi):
                        f_score.write("PESQ {:.6f} | STOI {:.6f}\n".format(p, s))
                    avg_pesq += float(pesq.sum()) / len(inputs)
                    avg_stoi += float(stoi.sum()) / len(inputs)
        if writer is not None and epoch % 10 == 0 and last is not None:
            writer.log_wav(last[0][0], last[1][0], last[2][0], epoch)
    finally:
        if f_score is not None:
            f_score.close()
        model.train(was_training)
    n = max(batch_num, 1)
    global LAST_VALIDATE_BATCHES
    LAST_VALIDATE_BATCHES = batch_num
    means = tuple(acc / n for acc in sums)
    if scorers is None:
        return means + (float("nan"), float("nan"))
    return means + (avg_pesq / n, avg_stoi / n)


def model_validate(model, validation_loader, writer, dir_to_save, epoch, DEVICE, scorers="default"):
    """trainer.py:188-241 (T-F masking).  Returns (validation_loss, avg_pesq, avg_stoi).  `scorers = (cal_pesq, cal_stoi)`, each
    `f(estimated[B, L] ndarray, clean[B, L] ndarray) -> per-utterance scores`; "default" = the package's C++ scorers;
    None: no scoring (both averages NaN, no score file)."""
    def batch(inputs, targets):
        _, _, outputs = model(inputs, targets)
        return (model.loss(outputs, targets),), outputs
    return _validate(model, validation_loader, writer, dir_to_save, epoch, DEVICE, batch, 1, scorers)


def model_perceptual_validate(model, validation_loader, writer, dir_to_save, epoch, DEVICE, scorers="default"):
    """trainer.py:242-306: loss = (main + perceptual) / 2.  Returns (loss, main, perceptual, avg_pesq, avg_stoi)."""
    def batch(inputs, targets):
        real_spec, img_spec, outputs = model(inputs)
        main_loss = model.loss(outputs, targets)
        perceptual_loss = model.loss(outputs, targets, real_spec, img_spec, perceptual=True)
        return ((main_loss + perceptual_loss) / 2, main_loss, perceptual_loss), outputs
    return _validate(model, validation_loader, writer, dir_to_save, epoch, DEVICE, batch, 3, scorers)


def fullsubnet_validate(model, validation_loader, writer, dir_to_save, epoch, DEVICE, scorers="default"):
    """trainer.py:309-373: loss on the compressed cIRM; the enhanced waveform is decompress_cIRM(cRM) x noisy spectrum through
    `tools.istft` (the reference passes a real-pair tensor to torch.istft there, which torch >= 2 rejects: SURVEY Q9)."""
    from . import tools_for_model as tools

    def batch(inputs, targets):
        noisy_complex = tools.stft(inputs)
        clean_complex = tools.stft(targets)
        noisy_mag, _ = tools.mag_phase(noisy_complex)
        cIRM = tools.build_complex_ideal_ratio_mask(noisy_complex, clean_complex)
        cRM = model(noisy_mag)
        loss = model.loss(cIRM, cRM)
        cRM = tools.decompress_cIRM(cRM)
        enhanced_real = cRM[..., 0] * noisy_complex.real - cRM[..., 1] * noisy_complex.imag
        enhanced_imag = cRM[..., 1] * noisy_complex.real + cRM[..., 0] * noisy_complex.imag
        enhanced_complex = torch.stack((enhanced_real, enhanced_imag), dim=-1)
        return (loss,), tools.istft(enhanced_complex, length=inputs.size(-1))
    return _validate(model, validation_loader, writer, dir_to_save, epoch, DEVICE, batch, 1, scorers)


def dccrn_direct_validate(model, validation_loader, writer, dir_to_save, epoch, DEVICE, scorers="default"):
    """trainer.py:377-430 (spectral mapping): loss = (loss(real) + loss(imag)) / 2 on the spectra."""
    def batch(inputs, targets):
        output_real, target_real, output_imag, target_imag, outputs = model(inputs, targets)
        return ((model.loss(output_real, target_real) + model.loss(output_imag, target_imag)) / 2,), outputs
    return _validate(model, validation_loader, writer, dir_to_save, epoch, DEVICE, batch, 1, scorers)


def crn_direct_validate(model, validation_loader, writer, dir_to_save, epoch, DEVICE, scorers="default"):
    """trainer.py:433-483: loss on the mapped magnitudes."""
    def batch(inputs, targets):
        output_mag, target_mag, outputs = model(inputs, targets)
        return (model.loss(output_mag, target_mag),), outputs
    return _validate(model, validation_loader, writer, dir_to_save, epoch, DEVICE, batch, 1, scorers)
