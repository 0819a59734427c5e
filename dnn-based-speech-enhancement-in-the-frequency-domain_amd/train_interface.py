"""Epoch driver with the reference's checkpoint / resume / logging contract (train_interface.py:49-239), as functions.

The reference file is a script (module-level code); here the same sequence is `run()`, so that it can be called under
torch.distributed (one process per GPU; only rank 0 writes files) and from tests.  File formats are the reference's:
  <job_dir>/<expr>_<m.d>_<model>_<loss>/chkpt_<epoch>.pt   = torch.save({'model', 'optimizer', 'epoch'})
  .../log.txt, .../mse_vali_total.npy, .../Epoch_<n>_SCORES, .../chkpt_opt.pt (copy of the best epoch)
and a checkpoint written by the reference loads here and vice versa (same state_dict keys, torch.optim.Adam state format).
"""
import os
import shutil
import time

import numpy as np
import torch

from . import config as cfg
from . import trainer as tr


def calculate_total_params(our_model):
    return sum(int(p.numel()) for p in our_model.parameters())


def write_status_to_log_file(fp, total_parameters):
    t = time.localtime()
    fp.write('%d-%d-%d %d:%d:%d\n' % (t.tm_year, t.tm_mon, t.tm_mday, t.tm_hour, t.tm_min, t.tm_sec))
    fp.write('total params   : %d (%.2f M, %.2f MBytes)\n' % (total_parameters, total_parameters / 1e6, total_parameters * 4.0 / 1e6))


def select_trainer_and_estimator():
    """train_interface.py:63-77."""
    if cfg.perceptual is not False:
        return tr.model_perceptual_train, tr.model_perceptual_validate
    if cfg.model == 'FullSubNet':
        return tr.fullsubnet_train, tr.fullsubnet_validate
    if cfg.masking_mode == 'Direct(None make)' and cfg.model == 'DCCRN':
        return tr.dccrn_direct_train, tr.dccrn_direct_validate
    if cfg.masking_mode == 'Direct(None make)' and cfg.model == 'CRN':
        return tr.crn_direct_train, tr.crn_direct_validate
    return tr.model_train, tr.model_validate


def build_model(DEVICE):
    from . import models
    return {'DCCRN': models.DCCRN, 'CRN': models.CRN, 'FullSubNet': models.FullSubNet}[cfg.model]().to(DEVICE)


def save_checkpoint(path, model, optimizer, epoch):
    """train_interface.py:166-171 / 204-210.  A step whose kernels gave up (Plan.status) never reached the parameters (guarded Adam), but it
    is an error all the same: checked here, behind a device synchronisation, BEFORE anything is written."""
    check_step_status(model)
    torch.save({'model': model.state_dict(), 'optimizer': optimizer.state_dict(), 'epoch': epoch}, path)


def check_step_status(model):
    """Raises when a kernel of the model's last plan gave up (this rank), or - data parallel - when ANY rank's did: the poisoned element of
    the all-reduced gradient (Plan.status_poison) is NaN on every rank, so all of them stop here instead of one raising and the others
    hanging in the next collective."""
    plan = getattr(model, "_status_plan", None)
    if plan is None:
        return
    torch.cuda.synchronize()
    if plan.status() != 0:
        raise RuntimeError("checkpoint not written" + plan._RC5)
    guard = getattr(model, "_dp_guard", None)
    if guard is not None and not bool(torch.isfinite(guard).all()):
        raise RuntimeError("checkpoint not written: another rank's plan gave up during the last step (its status word reached this rank "
                           "through the gradient all-reduce), or the gradient diverged at the guard element; every replica skipped that update")


def load_checkpoint(path, model, optimizer, map_location=None):
    """train_interface.py:107-111: returns the epoch to start from."""
    checkpoint = torch.load(path, map_location=map_location)
    model.load_state_dict(checkpoint['model'])
    optimizer.load_state_dict(checkpoint['optimizer'])
    return checkpoint['epoch'] + 1


class _NullWriter:
    """Stand-in for the tensorboardX `Writer` (write_on_tensorboard.py, out of scope): same calls, no output."""

    def log_loss(self, *a): pass
    def log_score(self, *a): pass
    def log_sub_loss(self, *a): pass
    def log_wav(self, *a): pass


def run(train_loader, validation_loader, model=None, optimizer=None, writer=None, DEVICE=None, exchange=None, rank=0, scorers="default",
        max_epochs=None):
    """The reference's main program: returns (model, optimizer, mse_vali_total, dir_to_save)."""
    from .optim import Adam
    DEVICE = torch.device(DEVICE or cfg.DEVICE)
    model = model if model is not None else build_model(DEVICE)
    optimizer = optimizer if optimizer is not None else Adam(model.parameters(), lr=cfg.learning_rate)
    total_params = calculate_total_params(model)
    trainer, estimator = select_trainer_and_estimator()
    max_epochs = max_epochs or cfg.max_epochs
    master = rank == 0
    if cfg.chkpt_model is not None:                                   # resume (train_interface.py:101-116)
        dir_to_save = cfg.job_dir + cfg.chkpt_model
        dir_to_logs = cfg.logs_dir + cfg.chkpt_model
        path = getattr(cfg, "chkpt_path", cfg.job_dir + cfg.chkpt_model + '/chkpt_' + cfg.chkpt + '.pt')
        epoch_start_idx = load_checkpoint(path, model, optimizer, map_location=DEVICE)
        mse_vali_total = np.load(str(dir_to_save + '/mse_vali_total.npy'))
        if len(mse_vali_total) < max_epochs:
            mse_vali_total = np.concatenate((mse_vali_total, np.zeros(max_epochs - len(mse_vali_total))), 0)
    else:
        if master:
            os.makedirs(cfg.job_dir, exist_ok=True)
            os.makedirs(cfg.logs_dir, exist_ok=True)
        epoch_start_idx = 1
        mse_vali_total = np.zeros(max_epochs)
        t = time.localtime()
        tag = cfg.expr_num + '_%d.%d' % (t.tm_mon, t.tm_mday) + '_%s' % cfg.model + '_%s' % cfg.loss
        if exchange is not None and exchange.world > 1:
            # one directory name for the whole job: rank 0's clock decides (a midnight rollover between the ranks would split it)
            import torch.distributed as dist
            box = [tag]
            dist.broadcast_object_list(box, src=0, group=exchange.pg)
            tag = box[0]
        dir_to_save, dir_to_logs = cfg.job_dir + tag, cfg.logs_dir + tag
    fp = None
    os.makedirs(dir_to_save, exist_ok=True)          # every rank: each one writes its own score file there (no shared file system assumed)
    if master:
        os.makedirs(dir_to_logs, exist_ok=True)
        log_fname = str(dir_to_save + '/log.txt')
        fresh = not os.path.exists(log_fname)
        fp = open(log_fname, 'w' if fresh else 'a')
        if fresh:
            write_status_to_log_file(fp, total_params)
    writer = writer if writer is not None else _NullWriter()
    perceptual = cfg.perceptual is not False
    kw = {"exchange": exchange} if exchange is not None else {}          # all five trainers take it
    if exchange is not None and exchange.world > 1:
        # replicas must START identical (each process built its model from its own RNG state, or loaded nothing): rank 0's parameters and
        # BatchNorm buffers go to everyone; optimizer moments are zero / come from the same checkpoint file on every rank
        if DEVICE.type == "cuda" and getattr(model, "_flat_param", 0) is None:
            model._flatten(DEVICE)
        exchange.broadcast_model(model)
    for epoch in range(epoch_start_idx, max_epochs + 1):
        start_time = time.time()
        if hasattr(train_loader, "set_epoch"):
            train_loader.set_epoch(epoch)
        res = trainer(model, optimizer, train_loader, DEVICE, **kw)
        if exchange is not None and exchange.world > 1 and not master:
            check_step_status(model)                                  # every rank looks before the collectives of the validation
        if master:                                                    # replicas are identical: same start (broadcast above), same averaged gradients
            save_checkpoint(str(dir_to_save + '/' + ('chkpt_%d.pt' % epoch)), model, optimizer, epoch)
        # every rank validates AND scores its shard; losses (what picks chkpt_opt), PESQ and STOI are averaged over the ranks weighted by
        # their batch counts - the numbers in log.txt are those of the whole validation set whatever the world size
        from . import trainer as _tr
        _tr.SCORE_FILE_SUFFIX = "" if master else ".rank%d" % rank
        val = estimator(model, validation_loader, writer if master else None, dir_to_save, epoch, DEVICE, scorers=scorers)
        if exchange is not None and exchange.world > 1:
            val = tuple(exchange.mean_scalars(val, weight=_tr.LAST_VALIDATE_BATCHES))
        if perceptual:
            train_loss, train_main_loss, train_perceptual_loss = res
            vali_loss, validation_main_loss, validation_perceptual_loss, vali_pesq, vali_stoi = val
        else:
            train_loss, (vali_loss, vali_pesq, vali_stoi) = res, val
        if master:
            writer.log_loss(train_loss, vali_loss, epoch)
            writer.log_score(vali_pesq, vali_stoi, epoch)
            if perceptual:
                writer.log_sub_loss(train_main_loss, train_perceptual_loss, validation_main_loss, validation_perceptual_loss, epoch)
                fp.write('Epoch [{}] | T {:.6f} | V {:.6}\n'.format(epoch, float(train_loss), float(vali_loss)))
                fp.write('          | T {:.6f} {:.6f} | V {:.6} {:.6f} takes {:.2f} seconds\n'.format(
                    float(train_main_loss), float(train_perceptual_loss), float(validation_main_loss), float(validation_perceptual_loss),
                    time.time() - start_time))
            else:
                fp.write('Epoch [{}] | T {:.6f} | V {:.6} takes {:.2f} seconds\n'.format(epoch, float(train_loss), float(vali_loss),
                                                                                         time.time() - start_time))
            fp.write('          | V PESQ: {:.6f} | STOI: {:.6f} \n'.format(vali_pesq, vali_stoi))
            fp.flush()
            mse_vali_total[epoch - 1] = float(vali_loss)
            np.save(str(dir_to_save + '/mse_vali_total.npy'), mse_vali_total)
    if master:
        fp.close()
        min_index = int(np.argmin(mse_vali_total[:max_epochs]))       # train_interface.py:233-239
        src_file = str(dir_to_save + '/' + ('chkpt_%d.pt' % (min_index + 1)))
        if os.path.exists(src_file):
            shutil.copy(src_file, str(dir_to_save + '/chkpt_opt.pt'))
    return model, optimizer, mse_vali_total, dir_to_save


if __name__ == "__main__":
    from .dataloader import create_dataloader
    run(create_dataloader(mode='train', device=cfg.DEVICE), create_dataloader(mode='valid', device=cfg.DEVICE))
