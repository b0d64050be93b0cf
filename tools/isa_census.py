#!/usr/bin/env python
"""Per kernel instance of a hipcc --save-temps assembly file: registers, spills, and whether any scratch (spill) instruction
sits in a basic block that also issues MFMAs (the k-steps) -- a reload there is a VMEM op ahead of the x chunk in the in-order
vmcnt queue.   python tools/isa_census.py file.s [name-regex]"""
import re, sys
from collections import Counter

src = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else re.compile("resample_i8")
meta = {}
for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", src):
    meta[m.group(1)] = (int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)))
print(f"{'instance':58s} {'vgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratchB':>8s} | {'mfma blocks':>11s} {'mfma':>5s} {'valu':>5s} {'lds':>5s} {'scratch in mfma blocks':>22s}")
for m in re.finditer(r"^(_ZN3txm\S+):.*?^\.Lfunc_end\d+:", src, re.S | re.M):
    name = m.group(1)
    if not pat.search(name) or name not in meta:
        continue
    blocks, cur = [], None
    for l in m.group(0).split("\n"):
        if re.match(r"^\.LBB\S+:", l):
            cur = Counter(); blocks.append(cur)
        elif cur is not None and l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"):
            op = l.split()[0]
            k = ("mfma" if op.startswith("v_mfma") else "scratch" if op.startswith("scratch_") else "lds" if op.startswith("ds_")
                 else "valu" if op.startswith("v_") else "other")
            cur[k] += 1
    mb = [b for b in blocks if b["mfma"]]
    t = re.search(r"kernelI(.*?)EEvNS", name)
    short = re.sub(r"Li(\d+)E", r"\1,", t.group(1)).replace("Lb1E", "T,").replace("Lb0E", "F,") if t else name
    sc, ss, vg, vs = meta[name]
    print(f"{name.split('kernel')[0][-14:] + 'kernel<' + short.rstrip(',') + '>':58s} {vg:5d} {vs:6d} {ss:6d} {sc:8d} | {len(mb):11d} {sum(b['mfma'] for b in mb):5d} "
          f"{sum(b['valu'] for b in mb):5d} {sum(b['lds'] for b in mb):5d} {sum(b['scratch'] for b in mb):22d}")
