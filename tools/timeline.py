#!/usr/bin/env python3
"""Per-step stream timeline from a rocprofv3 kernel trace CSV: busy time per stream, idle gaps of the main stream, the kernels around them.
usage: timeline.py <kernel_trace.csv> [marker kernel name that starts a step: default stft_fft_kernel]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
mark = sys.argv[2] if len(sys.argv) > 2 else "stft_fft_kernel"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void sefd::", "").replace("sefd::", "")[:48], r.get("Stream_Id", r.get("Queue_Id", "0"))) for r in rows), key=lambda e: e[0])
starts = [i for i, e in enumerate(ev) if mark in e[2]]
# a training step = two STFT launches (noisy, clean): take every second marker
starts = starts[::2]
if len(starts) < 3:
    print("not enough steps", len(starts)); sys.exit()
a, b = starts[-2], starts[-1]
step = ev[a:b]
t0, t1 = step[0][0], ev[b][0]
print(f"step wall {(t1 - t0) / 1e3:.1f} us, {len(step)} launches")
by = collections.defaultdict(list)
for e in step: by[e[3]].append(e)
for sid, es in by.items():
    busy = sum(e[1] - e[0] for e in es)
    print(f" stream/queue {sid}: {len(es)} launches, busy {busy / 1e3:.1f} us, span {(max(e[1] for e in es) - min(e[0] for e in es)) / 1e3:.1f} us")
main = max(by.values(), key=len)
gaps = []
for p, q in zip(main, main[1:]):
    g = q[0] - p[1]
    if g > 3000: gaps.append((g, p[2], q[2], (p[1] - t0) / 1e3))
print(f" main-stream idle gaps > 3 us: {len(gaps)}, total {sum(g[0] for g in gaps) / 1e3:.1f} us")
for g in sorted(gaps, reverse=True)[:14]:
    print(f"   {g[0] / 1e3:7.1f} us at +{g[3]:8.1f} us  after {g[1]}  before {g[2]}")
# union busy time of all streams
iv = sorted((e[0], e[1]) for e in step)
cur_s, cur_e, tot = iv[0][0], iv[0][1], 0
for s_, e_ in iv[1:]:
    if s_ > cur_e: tot += cur_e - cur_s; cur_s, cur_e = s_, e_
    else: cur_e = max(cur_e, e_)
tot += cur_e - cur_s
print(f" GPU busy (any stream) {tot / 1e3:.1f} us of {(t1 - t0) / 1e3:.1f}")
