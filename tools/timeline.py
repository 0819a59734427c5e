#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 kernel trace: per kernel start / duration / queue, the time each kernel class runs
ALONE (nothing else in flight on the other queue) and the idle gaps.  usage: timeline.py kernel_trace.csv [step_index_from_end]"""
import csv, sys, re, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sefd::" in r["Kernel_Name"]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
verbose = len(sys.argv) > 3
def short(n):
    n = re.sub(r"\(.*", "", n.replace("void ", "").replace("sefd::", ""))
    return n
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in rows]
ks.sort()
import os
mark = os.environ.get("TIMELINE_MARK", "stft_fft")                             # DCCRN: two STFTs open each step; FullSubNet: TIMELINE_MARK=fsn_in_kernel:1
mark, per = (mark.split(":") + ["2"])[:2] if ":" in mark else (mark, "2")
starts = [i for i, k in enumerate(ks) if k[2].startswith(mark)]
steps = [starts[i] for i in range(0, len(starts), int(per))]
a = steps[-1 - back]; b = steps[-back] if back > 0 else len(ks)
ks = ks[a:b]
t0 = ks[0][0]; t1 = max(k[1] for k in ks)
print(f"step: {len(ks)} kernels, {(t1 - t0) / 1e3:.1f} us wall")
ev = sorted([(k[0], 1, i) for i, k in enumerate(ks)] + [(k[1], -1, i) for i, k in enumerate(ks)])
alone = collections.Counter(); busy2 = 0; idle = 0; active = set(); last = t0
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        if not active: idle += dt
        elif len(active) == 1: alone[ks[next(iter(active))][2]] += dt
        else: busy2 += dt
    last = t
    if d == 1: active.add(i)
    else: active.discard(i)
print(f"idle {idle / 1e3:.1f} us, two or more kernels in flight {busy2 / 1e3:.1f} us, exactly one in flight {sum(alone.values()) / 1e3:.1f} us:")
for n, v in alone.most_common(25): print(f"  {v / 1e3:9.1f} us  {n}")
if verbose:
    for k in ks: print(f"{(k[0] - t0) / 1e3:9.1f} +{(k[1] - k[0]) / 1e3:7.1f} q{k[3]} wg{k[4]:6d} {k[2]}")
