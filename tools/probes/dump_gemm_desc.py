#!/usr/bin/env python3
"""Dump the wide-tile GEMM launch descriptors (RUNGEMM ops the planner marks kRunWTile32) of a plan to tools/probes/gemm_desc.bin for
tools/probes/gemm_ladder.hip (runs on the CPU: the planner is pure C++).  Layout: int64 magic, int64 op_size, int64 arena_bytes[6], int64 nops,
then nops x { int32 phase, int32 index, Op raw bytes }."""
import ctypes as C
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sefd_amd  # noqa: E402,F401
from sefd_amd.plan import PHASE_BWD, PHASE_FWD, Plan  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tools", "probes", "gemm_desc.bin")
    p = Plan(B, 48000, masking_mode="C", act_dtype="bf16")
    sz = p.lib.sefd_op_size()
    recs = []
    for ph in (PHASE_FWD, PHASE_BWD):
        base = p.ops_ptr(ph)
        for i in range(p.num_ops(ph)):
            o = p.op_info(ph, i)
            # the wide-tile GEMMs (kRunWTile32) and, for the 256 x 128 rung (tools/probes/ladder/cgemm128.hip), the LDS-DMA-able bf16 GEMMs of width 128
            if o["kind"] == 1 and ((o["flags"] & 16) or (o["dtype"] == 1 and (o["flags"] & 1) and o["N"] == 128 and o["M"] >= 65536)):
                raw = C.string_at(base + i * sz, sz)
                recs.append((ph, i, raw, o))
    with open(out, "wb") as f:
        f.write(struct.pack("<qq6qq", 0x53454644, sz, *[int(b) for b in p.arena_bytes], len(recs)))
        for ph, i, raw, o in recs:
            f.write(struct.pack("<ii", ph, i))
            f.write(raw)
    for ph, i, raw, o in recs:
        print(ph, i, o["tag"], o["M"], o["N"], o["K"], o["flags"])
    print("wrote", out, len(recs), "ops, op_size", sz)


if __name__ == "__main__":
    main()
