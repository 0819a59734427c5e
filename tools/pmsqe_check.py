"""PMSQE on the GPU: error against oracle/pmsqe.py and the time of the forward + backward at the bench batch (B = 32, 3 s)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sefd_amd  # noqa
from sefd_amd import config as cfg, tools_for_loss as tfl
from oracle import pmsqe
from test_oracle_pmsqe import speechlike
for power in (False, True):
    cfg.pmsqe_power = power
    c, n = speechlike(8, seed=1)
    e = n.cuda().requires_grad_()
    l = tfl.get_array_pmsqe_loss(c.cuda(), e); l.backward()
    eo = n.clone().double().requires_grad_(); lo = pmsqe.pmsqe_loss(c, eo, power); lo.backward()
    print(f"power={power} loss {float(l.detach()):.6f} oracle {float(lo):.6f} rel {abs(float(l.detach())-float(lo))/float(lo):.2e} grad rel L2 {float((e.grad.cpu().double()-eo.grad).norm()/eo.grad.norm()):.2e}")
cfg.pmsqe_power = False
c, n = speechlike(32, seed=2)
c, n = c.cuda(), n.cuda()
for _ in range(3):
    e = n.clone().requires_grad_(); tfl.get_array_pmsqe_loss(c, e).backward()
torch.cuda.synchronize(); t = time.time()
for _ in range(20):
    e = n.clone().requires_grad_(); tfl.get_array_pmsqe_loss(c, e).backward()
torch.cuda.synchronize(); print(f"B=32 fwd+bwd {(time.time()-t)/20*1e3:.3f} ms")
