#!/usr/bin/env python
"""The sixteen-wave int8 kernel sets M0 once per k-step and its ds_write_addtid stores rely on it (txm_resample_i8w.hip).
This scans the kernel's ISA (hipcc --save-temps output) for any OTHER write of M0:  python tools/check_m0.py file.s"""
import re, sys
src = open(sys.argv[1]).read()
bad = 0
for m in re.finditer(r"^(_ZN3txm19resample_i8w_kernel\S+):.*?^\.Lfunc_end", src, re.S | re.M):
    body = m.group(0)
    writes = [l for l in body.split("\n") if re.search(r"\bm0\b", l) and not l.strip().startswith(";")]
    other = [l for l in writes if not re.match(r"\s*s_mov_b32 m0, s\d+", l)]
    n_mine = len(writes) - len(other)
    print(m.group(1)[:60], "m0 writes:", n_mine, "other m0 lines:", len(other))
    for l in other[:5]:
        print("   ", l.strip())
    bad += len(other)
sys.exit(1 if bad else 0)
