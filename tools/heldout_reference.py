#!/usr/bin/env python3
"""Held-out evaluation, REFERENCE leg.  RUN ONLY IN THE BUILD CONTAINER (needs /root/reference; nothing here travels to the GPU box).

Trains the real reference DCCRN (models.DCCRN through the import shim of tests/golden/make_golden.py, the loop of trainer.py:15-42:
model(inputs, targets) -> model.loss -> zero_grad / backward / Adam.step) on the CPU with
  * the formula initial weights of oracle/weights.py (what `tools/heldout_eval.py train --init formula` gives the HIP model),
  * the same synthetic pool (tools/heldout_eval.make_set), the same batch order (numpy default_rng(order seed)), the same lr,
then enhances the held-out clips in eval mode and scores them with the reference's PESQ.so, this repo's C++ PESQ and C++ STOI.
Only the score table (data) is written: profiles/r04_heldout_reference.json.

    python tools/heldout_reference.py --orders 3,4,5,6 --steps 1000 --batch 8 --train-len 16000 --threads 4
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--orders", default="3,4,5,6")
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--pool", type=int, default=96)
    ap.add_argument("--heldout", type=int, default=32)
    ap.add_argument("--train-len", type=int, default=16000)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--json", default=os.path.join(ROOT, "profiles", "r04_heldout_reference.json"))
    ap.add_argument("--wavdir", default=os.path.join(ROOT, "gpurun_out", "heldout_ref"))
    ap.add_argument("--rescore", action="store_true", help="no training: recompute pesq_cpp / stoi of every stored run from its kept enhanced_reference_o*.npy "
                                                           "(after a change of the C++ scorers)")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from make_golden import import_reference
    from oracle.weights import fill_state_dict_
    import heldout_eval as he
    import sefd_amd  # noqa: F401
    from sefd_amd import tools_for_estimate as est
    cfg, models, _, _ = import_reference()
    cfg.masking_mode, cfg.loss, cfg.perceptual, cfg.lstm, cfg.skip_type = "E", "SI-SNR", False, "complex", True
    cfg.dccrn_kernel_num = [32, 64, 128, 256, 256, 256]
    pool_c, pool_n = he.make_set(1, a.pool, a.train_len)
    held_c, held_n = he.make_set(2, a.heldout)
    dll = ctypes.CDLL("/root/reference/PESQ.so")
    dll.pesq.restype = ctypes.c_double

    def pesq_ref(ref, deg):                      # the reference's own calling convention (tools_for_estimate.py:68-75)
        ref, deg = np.ascontiguousarray(ref, np.double), np.ascontiguousarray(deg, np.double)
        return float(dll.pesq(ctypes.c_void_p(ref.ctypes.data), ctypes.c_void_p(deg.ctypes.data), len(ref), len(deg)))

    os.makedirs(a.wavdir, exist_ok=True)
    # (the thread count of the first runs stays in the string: it keys the result file; later runs used more threads - not part of the protocol)
    out = dict(protocol=dict(model="reference DCCRN default, mask E, SI-SNR, fp32 CPU (torch %s, 3 threads)" % torch.__version__,
                             init="oracle/weights.py formula", steps=a.steps, batch=a.batch, lr=a.lr, pool=a.pool, heldout=a.heldout,
                             train_len=a.train_len, heldout_len=he.L), runs={})
    if os.path.exists(a.json):
        old = json.load(open(a.json))
        if old.get("protocol") == out["protocol"]:
            out = old
    clean16 = np.round(held_c * 32767).astype(np.int16).astype(np.float64)
    if a.rescore:
        for key, run in out["runs"].items():
            f = os.path.join(a.wavdir, f"enhanced_reference_o{key}.npy")
            if not os.path.exists(f):
                print("order", key, ": no kept output, scores unchanged")
                continue
            e = np.load(f).astype(np.float64)
            pc = est.cal_pesq([e[i] / 32768.0 for i in range(len(e))], [clean16[i] / 32768.0 for i in range(len(e))])
            st = est.cal_stoi([e[i] / 32768.0 for i in range(len(e))], [clean16[i] / 32768.0 for i in range(len(e))])
            for i, row in enumerate(run["rows"]):
                row["pesq_cpp"], row["stoi"] = float(pc[i]), float(st[i])
            run["mean"] = {k: float(np.mean([r[k] for r in run["rows"]])) for k in ("pesq", "pesq_cpp", "stoi")}
            print("order", key, run["mean"], flush=True)
        out["scorer_note"] = "pesq_cpp / stoi re-scored with the round-4 C++ scorers (PESQ input-stage fix); pesq = the reference's PESQ.so at training time"
        json.dump(out, open(a.json, "w"), indent=1)
        return
    for order_seed in [int(s) for s in a.orders.split(",")]:
        key = str(order_seed)
        if key in out["runs"]:
            continue
        torch.manual_seed(0)
        m = models.DCCRN(rnn_units=cfg.rnn_units, masking_mode="E")
        fill_state_dict_(m)
        m.train()
        opt = torch.optim.Adam(m.parameters(), lr=a.lr)
        order = np.random.default_rng(order_seed)
        losses = []
        t0 = time.time()
        for step in range(a.steps):
            idx = order.integers(0, a.pool, a.batch)
            x, y = torch.from_numpy(pool_n[idx]), torch.from_numpy(pool_c[idx])
            outputs = m(x, y)
            loss = m.loss(outputs[2], y)
            opt.zero_grad()
            loss.backward()
            opt.step()
            if step % 20 == 0 or step == a.steps - 1:
                losses.append((step, float(loss)))
                print(f"order {order_seed} step {step} loss {float(loss):.4f}  {time.time() - t0:.0f} s", flush=True)
        m.eval()
        outs = []
        with torch.no_grad():
            for i in range(0, a.heldout, 8):
                outs.append(m(torch.from_numpy(held_n[i:i + 8]), torch.from_numpy(held_c[i:i + 8]))[2].numpy())
        enh = np.round(np.clip(np.concatenate(outs), -1, 1) * 32767).astype(np.int16)
        np.save(os.path.join(a.wavdir, f"enhanced_reference_o{order_seed}.npy"), enh)
        e = enh.astype(np.float64)
        rows = []
        for i in range(a.heldout):
            rows.append(dict(utt=i, pesq=pesq_ref(clean16[i], e[i]),
                             pesq_cpp=float(est.cal_pesq([e[i] / 32768.0], [clean16[i] / 32768.0])[0]),
                             stoi=float(est.cal_stoi([e[i] / 32768.0], [clean16[i] / 32768.0])[0])))
        out["runs"][key] = dict(losses=losses, train_seconds=time.time() - t0, rows=rows,
                                mean={k: float(np.mean([r[k] for r in rows])) for k in ("pesq", "pesq_cpp", "stoi")})
        print("order", order_seed, out["runs"][key]["mean"], flush=True)
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
