#!/usr/bin/env python
"""Summarise a rocprofv3 `--kernel-trace` result database (rocpd sqlite) into a per-kernel table
(calls, total / average / min / max duration, share) - the same columns as `--stats` CSV output."""
import re
import sqlite3
import sys


def main(db, steps=None):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
                     f"max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(d.group_segment_size) "
                     f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# {db}: {sum(r[1] for r in rows)} dispatches, {total/1e6:.3f} ms of kernel time")
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s} {'vgpr':>5s} {'agpr':>5s} {'lds':>7s}")
    for name, n, tot, mn, mx, vg, ag, lds in rows:
        short = re.sub(r"\(.*", "", name)
        short = short.replace("void sefd::", "").replace("sefd::", "")[:70]
        print(f"{short:70s} {n:7d} {tot/1e6:10.3f} {tot/n/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*tot/total:6.2f} {vg:5d} {ag:5d} {lds:7d}")


if __name__ == "__main__":
    main(sys.argv[1])
