#!/usr/bin/env python
"""Compact view of the MFMA loops in a hipcc -save-temps .s file: per kernel, every basic block that holds a barrier, MFMAs and a
branch, as a string of M (mfma) d (ds_read) V (buffer_load ... lds) w (s_waitcnt) |B| (s_barrier) > (branch).  Used to check
the K2 GEMM's instruction interleave without a GPU."""
import re
import sys

s = open(sys.argv[1]).read()
lines = s.split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:\s", l)]
for a, b in zip(starts, starts[1:] + [len(lines)]):
    name = lines[a].split(":")[0]
    body = "\n".join(lines[a:b])
    for blk in re.split(r"\n(?=\.LBB\d+_\d+:)", body):
        if blk.count("v_mfma") >= 8 and "s_barrier" in blk and "s_cbranch" in blk:
            seq = []
            for l in blk.split("\n"):
                l = l.strip()
                if l.startswith("v_mfma"): seq.append("M")
                elif l.startswith("ds_read"): seq.append("d")
                elif l.startswith("buffer_load"): seq.append("V")
                elif l.startswith("s_barrier"): seq.append("|B|")
                elif l.startswith("s_waitcnt"): seq.append("w(" + l.split(None, 1)[1].replace("cnt", "") + ")")
                elif l.startswith("s_cbranch"): seq.append(">")
            print(re.findall(r"Li(\d)E", name)[:5], re.findall(r"Lb(\d)E", name), "".join(seq))
