#!/usr/bin/env python3
"""Held-out evaluation of the fp32 and bf16 training paths (north_star criterion: PESQ / STOI of the bf16-trained model within
+-0.02 of the fp32-trained one).

Two legs, because the reference's PESQ.so (its only PESQ implementation, x86 binary without source) cannot travel to the GPU box:

  GPU box:    python tools/heldout_eval.py train --steps 400 --out gpurun_out/heldout
      trains DCCRN (default sizes, mask E, SI-SNR) from the SAME initial weights on the SAME synthetic speech-like stream once
      with fp32 activations and once with bf16 (config.act_dtype), enhances a held-out set with both models (eval mode) and
      writes clean / noisy / enhanced_fp32 / enhanced_bf16 as int16 .npy plus the loss curves.
  container:  python tools/heldout_eval.py score --dir gpurun_out/heldout --json profiles/r02_heldout_eval.json
      scores every utterance with the reference's PESQ.so through its own ctypes contract (tools_for_estimate.py:68-84: float64
      arrays, returns WB MOS-LQO) when /root/reference is present, and with this repo's C++ STOI (tools_for_estimate.cal_stoi).

Synthetic data: there is no corpus in the image.  "Speech" = a harmonic source with a wandering pitch through three slowly moving
formant resonances, syllable-rate amplitude envelope and pauses; noise = coloured noise + a hum, mixed at 0..15 dB SNR."""
import argparse
import json
import os
import sys

import numpy as np
from scipy.signal import lfilter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FS, L = 16000, 48000


def speechlike(rng, n=L):
    t = np.arange(n) / FS
    f0 = 110 + 60 * rng.random() + 25 * np.sin(2 * np.pi * (0.7 + rng.random()) * t + rng.random() * 6)
    ph = 2 * np.pi * np.cumsum(f0) / FS
    src = sum(np.sin(k * ph) / k for k in range(1, 30))
    out = np.zeros(n)
    for fc, bw in ((500 + 300 * rng.random(), 90), (1500 + 600 * rng.random(), 120), (2600 + 500 * rng.random(), 160)):
        r = np.exp(-np.pi * bw / FS)
        y = np.empty(n)
        zi = np.zeros(2)
        for i in range(0, n, 800):               # two-pole resonator whose centre moves every 50 ms
            fm = fc * (1 + 0.15 * np.sin(2 * np.pi * 1.7 * t[i] + fc))
            y[i:i + 800], zi = lfilter([1.0], [1.0, -2 * r * np.cos(2 * np.pi * fm / FS), r * r], src[i:i + 800], zi=zi)
        out += y / (np.abs(y).max() + 1e-9)
    env = np.clip(np.sin(2 * np.pi * (3.5 + rng.random()) * t + rng.random() * 6), 0, None) ** 0.7
    gate = (np.sin(2 * np.pi * (0.45 + 0.2 * rng.random()) * t + rng.random() * 6) > -0.55).astype(float)
    gate = np.convolve(gate, np.hanning(801) / np.hanning(801).sum(), mode="same")
    s = out * env * gate
    return 0.25 * s / (np.abs(s).max() + 1e-9)


def noise(rng, n=L):
    w = rng.standard_normal(n + 64)
    k = np.exp(-np.arange(64) / (2 + 20 * rng.random()))
    c = np.convolve(w, k, mode="valid")[:n]
    c = c / (c.std() + 1e-9)
    hum = 0.3 * np.sin(2 * np.pi * (50 + 900 * rng.random()) * np.arange(n) / FS)
    return c + hum


def make_set(seed, count, n=L):
    rng = np.random.default_rng(seed)
    clean, noisy = [], []
    for _ in range(count):
        s, v = speechlike(rng, n), noise(rng, n)
        snr = 15 * rng.random()
        v = v * np.sqrt((s ** 2).mean() / ((v ** 2).mean() * 10 ** (snr / 10)))
        clean.append(s)
        noisy.append(np.clip(s + v, -1, 1))
    return np.stack(clean).astype(np.float32), np.stack(noisy).astype(np.float32)


def train(args):
    import torch
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models
    from sefd_amd.optim import Adam
    os.makedirs(args.out, exist_ok=True)
    pool_c, pool_n = make_set(1, args.pool, args.train_len)
    held_c, held_n = make_set(2, args.heldout)
    B = args.batch
    res, scores = {}, {}
    init = None
    # third leg: fp32 again with another batch order - the run-to-run spread of the metric, against which the bf16 delta is read
    # last leg: bf16 with the perceptual step (SI-SNR + PMSQE) / 2 - PMSQE is built to track PESQ, so a PESQ gain on the held-out set
    # is a consistency check of the (unpinned) loss and its hand-derived gradient
    for tag in args.legs.split(","):
        # "fp32:5" = leg fp32 with batch-order seed 5, saved as enhanced_fp32_o5.npy (the protocol of tools/heldout_reference.py)
        tag, _, oseed = tag.partition(":")
        dt = tag.split("_")[0]
        perc = "PMSQE" if "pmsqe" in tag else None
        cfg.pmsqe_power = tag.endswith("pmsqe_power")
        cfg.masking_mode, cfg.loss, cfg.act_dtype = "E", "SI-SNR", dt
        torch.manual_seed(0)
        m = models.DCCRN(rnn_units=cfg.rnn_units, masking_mode="E").to("cuda").train()
        if args.init == "formula":               # oracle/weights.py values: what the reference leg (tools/heldout_reference.py) starts from
            from oracle.weights import fill_state_dict_
            fill_state_dict_(m)
        elif init is None:
            init = {k: v.clone() for k, v in m.state_dict().items()}
        else:
            m.load_state_dict(init)
        opt = Adam(m.parameters(), lr=args.lr)
        order = np.random.default_rng(int(oseed) if oseed else (4 if tag.endswith("rerun") else 3))
        if oseed:
            tag = f"{tag}_o{oseed}"
        losses = []
        for step in range(args.steps):
            idx = order.integers(0, args.pool, B)
            x, y = torch.from_numpy(pool_n[idx]).cuda(), torch.from_numpy(pool_c[idx]).cuda()
            loss = m.train_step(x, y, opt, perceptual=perc)
            if step % 20 == 0 or step == args.steps - 1:
                losses.append((step, float(loss)))
        m.eval()
        outs = []
        with torch.no_grad():
            for i in range(0, args.heldout, B):
                o = m(torch.from_numpy(held_n[i:i + B]).cuda())
                outs.append((o[2] if isinstance(o, (tuple, list)) else o).float().cpu().numpy())
        enh = np.concatenate(outs)
        enh16 = np.round(np.clip(enh, -1, 1) * 32767).astype(np.int16)
        if not args.no_wav or tag.split("_o")[-1] in args.keep_wav.split(","):
            np.save(os.path.join(args.out, f"enhanced_{tag}.npy"), enh16)
        if args.score:                           # the shipped C++ scorers on the box (the merged gpurun_out is capped at 64 MiB: 12 legs of int16 clips are not)
            from sefd_amd import tools_for_estimate as est
            c16 = np.round(held_c * 32767).astype(np.int16).astype(np.float64) / 32768.0
            e = enh16.astype(np.float64) / 32768.0
            scores[tag] = dict(pesq_cpp=[float(v) for v in est.cal_pesq(e, c16)], stoi=[float(v) for v in est.cal_stoi(e, c16)])
            print(tag, "pesq_cpp", float(np.mean(scores[tag]["pesq_cpp"])), "stoi", float(np.mean(scores[tag]["stoi"])), flush=True)
        res[tag] = losses
        print(tag, "loss", losses[0], "->", losses[-1], flush=True)
    np.save(os.path.join(args.out, "clean.npy"), np.round(held_c * 32767).astype(np.int16))
    np.save(os.path.join(args.out, "noisy.npy"), np.round(held_n * 32767).astype(np.int16))
    json.dump(dict(steps=args.steps, batch=B, lr=args.lr, pool=args.pool, heldout=args.heldout, init=args.init, train_len=args.train_len, losses=res, scores=scores),
              open(os.path.join(args.out, "train_log.json"), "w"))


def score(args):
    import ctypes
    import sefd_amd  # noqa: F401
    from sefd_amd import tools_for_estimate as est
    ld = lambda n: np.load(os.path.join(args.dir, n)).astype(np.float64)
    clean, noisy, e32, e16, e32b = ld("clean.npy"), ld("noisy.npy"), ld("enhanced_fp32.npy"), ld("enhanced_bf16.npy"), ld("enhanced_fp32_rerun.npy")
    e16b = ld("enhanced_bf16_rerun.npy") if os.path.exists(os.path.join(args.dir, "enhanced_bf16_rerun.npy")) else None
    e16p = ld("enhanced_bf16_pmsqe.npy") if os.path.exists(os.path.join(args.dir, "enhanced_bf16_pmsqe.npy")) else None
    e16pp = ld("enhanced_bf16_pmsqe_power.npy") if os.path.exists(os.path.join(args.dir, "enhanced_bf16_pmsqe_power.npy")) else None
    pesq = None
    so = "/root/reference/PESQ.so"
    if os.path.exists(so):
        dll = ctypes.CDLL(so)
        dll.pesq.restype = ctypes.c_double

        def pesq(ref, deg):                      # the reference's own calling convention (tools_for_estimate.py:68-75)
            ref, deg = np.ascontiguousarray(ref, np.double), np.ascontiguousarray(deg, np.double)
            return float(dll.pesq(ctypes.c_void_p(ref.ctypes.data), ctypes.c_void_p(deg.ctypes.data), len(ref), len(deg)))
    rows = []
    for i in range(len(clean)):
        r = dict(utt=i)
        for name, sig in (("noisy", noisy), ("fp32", e32), ("bf16", e16), ("fp32_rerun", e32b)) + ((("bf16_rerun", e16b),) if e16b is not None else ()) + ((("bf16_pmsqe", e16p),) if e16p is not None else ()) + ((("bf16_pmsqe_power", e16pp),) if e16pp is not None else ()):
            r["stoi_" + name] = float(est.cal_stoi([sig[i] / 32768.0], [clean[i] / 32768.0])[0])
            if pesq:
                r["pesq_" + name] = pesq(clean[i], sig[i])
            r["pesq_cpp_" + name] = float(est.cal_pesq([sig[i] / 32768.0], [clean[i] / 32768.0])[0])   # the shipped C++ scorer beside it
        rows.append(r)
    mean = {k: float(np.mean([r[k] for r in rows])) for k in rows[0] if k != "utt"}
    out = dict(scorer_pesq="reference PESQ.so (WB MOS-LQO), run in the build container" if pesq else None,
               scorer_stoi="sefd_stoi_batch (C++ restatement of the published STOI, pystoi conventions; unpinned)",
               n_utts=len(rows), mean=mean,
               delta_bf16_minus_fp32={k: mean[k + "_bf16"] - mean[k + "_fp32"] for k in (("pesq", "stoi") if pesq else ("stoi",))},
               delta_fp32_rerun_minus_fp32={k: mean[k + "_fp32_rerun"] - mean[k + "_fp32"] for k in (("pesq", "stoi") if pesq else ("stoi",))},
               delta_bf16_mean_minus_fp32_mean={k: (mean[k + "_bf16"] + mean.get(k + "_bf16_rerun", mean[k + "_bf16"])) / 2 - (mean[k + "_fp32"] + mean[k + "_fp32_rerun"]) / 2
                                                for k in (("pesq", "stoi") if pesq else ("stoi",))},
               delta_bf16_pmsqe_minus_bf16_mean=({k: mean[k + "_bf16_pmsqe"] - (mean[k + "_bf16"] + mean.get(k + "_bf16_rerun", mean[k + "_bf16"])) / 2
                                                  for k in (("pesq", "stoi") if pesq else ("stoi",))} if e16p is not None else None),
               delta_bf16_pmsqe_power_minus_bf16_mean=({k: mean[k + "_bf16_pmsqe_power"] - (mean[k + "_bf16"] + mean.get(k + "_bf16_rerun", mean[k + "_bf16"])) / 2
                                                        for k in (("pesq", "stoi") if pesq else ("stoi",))} if e16pp is not None else None),
               per_utt_abs_delta_max={k: float(max(abs(r[k + "_bf16"] - r[k + "_fp32"]) for r in rows)) for k in (("pesq", "stoi") if pesq else ("stoi",))},
               train=json.load(open(os.path.join(args.dir, "train_log.json"))), rows=rows)
    json.dump(out, open(args.json, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("mean", "delta_bf16_minus_fp32", "delta_fp32_rerun_minus_fp32", "delta_bf16_mean_minus_fp32_mean", "delta_bf16_pmsqe_minus_bf16_mean", "delta_bf16_pmsqe_power_minus_bf16_mean", "per_utt_abs_delta_max")}, indent=1))


def score_vs_ref(args):
    """HIP legs (train --init formula --score, the C++ scorers on the GPU box) against the reference-trained models of
    tools/heldout_reference.py (same formula init, same pool, same batch orders, CPU fp32): paired by batch order.
    delta(order) = mean over the held-out clips of (HIP score - reference score); reported: mean over orders, standard error, 95 % interval
    (Student t), and the verdict against the two-sided +-0.02 criterion of north_star."""
    from scipy import stats
    ref = None
    for rp in args.ref.split(","):                     # several reference files (the runs were split over two processes): same protocol, runs merged
        if not os.path.exists(rp):
            continue
        r = json.load(open(rp))
        if ref is None:
            ref = r
        else:
            assert r["protocol"] == ref["protocol"], "reference protocols differ"
            ref["runs"].update(r["runs"])
    hip = json.load(open(os.path.join(args.dir, "train_log.json")))
    for extra in sorted(os.listdir(args.dir)):         # more HIP legs from a later GPU call: train_log_*.json
        if extra.startswith("train_log_") and extra.endswith(".json"):
            e = json.load(open(os.path.join(args.dir, extra)))
            assert all(e[k] == hip[k] for k in ("steps", "batch", "pool", "heldout", "train_len", "init"))
            hip["scores"].update(e["scores"]); hip["losses"].update(e["losses"])
    prot = ref["protocol"]
    assert (hip["steps"], hip["batch"], hip["pool"], hip["heldout"], hip.get("train_len"), hip.get("init")) == \
           (prot["steps"], prot["batch"], prot["pool"], prot["heldout"], prot["train_len"], "formula"), "protocols differ"
    out = dict(protocol=prot, scorer="C++ wide-band PESQ (csrc_host/pesq.cpp) and C++ STOI on both sides; the reference leg also carries PESQ.so scores",
               orders={}, summary={})
    pesq_so = None
    if os.path.exists("/root/reference/PESQ.so") and os.path.exists(os.path.join(args.dir, "clean.npy")):
        import ctypes
        dll = ctypes.CDLL("/root/reference/PESQ.so")
        dll.pesq.restype = ctypes.c_double

        def pesq_so(r, d):
            r, d = np.ascontiguousarray(r, np.double), np.ascontiguousarray(d, np.double)
            return float(dll.pesq(ctypes.c_void_p(r.ctypes.data), ctypes.c_void_p(d.ctypes.data), len(r), len(d)))
    for o, run in sorted(ref["runs"].items(), key=lambda kv: int(kv[0])):
        row = dict(reference={k: float(np.mean([r[k] for r in run["rows"]])) for k in ("pesq", "pesq_cpp", "stoi")},
                   reference_final_loss=run["losses"][-1][1])
        for dt in ("fp32", "bf16"):
            tag = f"{dt}_o{o}"
            if tag not in hip["scores"]:
                continue
            sc = hip["scores"][tag]
            row[dt] = dict(pesq_cpp=float(np.mean(sc["pesq_cpp"])), stoi=float(np.mean(sc["stoi"])), final_loss=hip["losses"][tag][-1][1],
                           delta_pesq_cpp=float(np.mean(sc["pesq_cpp"]) - row["reference"]["pesq_cpp"]),
                           delta_stoi=float(np.mean(sc["stoi"]) - row["reference"]["stoi"]),
                           per_utt_abs_delta_pesq_mean=float(np.mean(np.abs(np.array(sc["pesq_cpp"]) - np.array([r["pesq_cpp"] for r in run["rows"]])))))
            wav = os.path.join(args.dir, f"enhanced_{tag}.npy")
            if pesq_so and os.path.exists(wav):   # the reference's own scorer on the HIP model's clips (kept for one order: 64 MiB transfer cap)
                clean, e = np.load(os.path.join(args.dir, "clean.npy")).astype(np.float64), np.load(wav).astype(np.float64)
                v = float(np.mean([pesq_so(clean[i], e[i]) for i in range(len(e))]))
                row[dt]["pesq_so"] = v
                row[dt]["delta_pesq_so"] = v - row["reference"]["pesq"]
        out["orders"][o] = row
    for dt in ("fp32", "bf16"):
        for key in ("delta_pesq_cpp", "delta_stoi"):
            d = np.array([r[dt][key] for r in out["orders"].values() if dt in r])
            if len(d) < 2:
                continue
            se = float(d.std(ddof=1) / np.sqrt(len(d)))
            half = float(stats.t.ppf(0.975, len(d) - 1) * se)
            lim = 0.02
            verdict = "pass" if abs(d.mean()) + half <= lim else ("fail" if abs(d.mean()) - half > lim else "mean inside, interval straddles the limit" if abs(d.mean()) <= lim else "mean outside, interval straddles the limit")
            out["summary"][f"{dt}.{key}"] = dict(n_orders=int(len(d)), mean=float(d.mean()), std=float(d.std(ddof=1)), se=se, ci95=[float(d.mean() - half), float(d.mean() + half)],
                                                 criterion="+-0.02 two-sided", verdict=verdict, per_order=[float(v) for v in d])
    d = np.array([r["bf16"]["pesq_cpp"] - r["fp32"]["pesq_cpp"] for r in out["orders"].values() if "bf16" in r and "fp32" in r])
    if len(d) >= 2:
        out["summary"]["bf16_minus_fp32.pesq_cpp"] = dict(mean=float(d.mean()), se=float(d.std(ddof=1) / np.sqrt(len(d))), per_order=[float(v) for v in d])
    json.dump(out, open(args.json, "w"), indent=1)
    print(json.dumps(out["summary"], indent=1))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    t = sub.add_parser("train")
    t.add_argument("--steps", type=int, default=400)
    t.add_argument("--batch", type=int, default=16)
    t.add_argument("--lr", type=float, default=1e-3)
    t.add_argument("--pool", type=int, default=96)
    t.add_argument("--heldout", type=int, default=16)
    t.add_argument("--legs", default="fp32,bf16,fp32_rerun,bf16_rerun,bf16_pmsqe")
    t.add_argument("--init", default="torch", choices=("torch", "formula"))
    t.add_argument("--score", action="store_true", help="score every leg on the box with the C++ PESQ / STOI (train_log.json 'scores')")
    t.add_argument("--no-wav", action="store_true", help="do not keep the enhanced clips (except --keep-wav orders)")
    t.add_argument("--keep-wav", default="", help="comma-separated batch-order seeds whose enhanced clips are kept with --no-wav")
    t.add_argument("--train-len", type=int, default=L)
    t.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "heldout"))
    s = sub.add_parser("score")
    s.add_argument("--dir", default=os.path.join(ROOT, "gpurun_out", "heldout"))
    s.add_argument("--json", default=os.path.join(ROOT, "profiles", "r02_heldout_eval.json"))
    v = sub.add_parser("score_vs_ref")
    v.add_argument("--dir", default=os.path.join(ROOT, "gpurun_out", "heldout_r04"))
    v.add_argument("--ref", default=os.path.join(ROOT, "profiles", "r04_heldout_reference.json") + "," + os.path.join(ROOT, "profiles", "r04_heldout_reference_b.json"))
    v.add_argument("--json", default=os.path.join(ROOT, "profiles", "r04_heldout_eval.json"))
    a = ap.parse_args()
    {"train": train, "score": score, "score_vs_ref": score_vs_ref}[a.cmd](a)
