"""Build libtxmom.so (HIP, gfx950 only) in-tree with hipcc.

``python -m thermoextrap_amd._build`` or ``__graft_entry__.build()``.
hipcc cross-compiles for gfx950 without a GPU present.
"""

from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libtxmom.so"
SOURCES = ["txm_api.hip", "txm_reduce.hip", "txm_sampler.hip", "txm_small.hip", "txm_resample.hip", "txm_resample_i8.hip",
           "txm_resample_i8t.hip", "txm_resample_i8g.hip", "txm_resample_i8gn.hip", "txm_count_table.hip", "txm_perturb.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-pass-failed"]
# per-file flags.  txm_resample_i8t.hip: 11 int32 accumulator tiles (176 registers) per wave at two waves per SIMD only
# fit when the 256 registers are ONE file -- MFMA accumulators in VGPRs, no AGPR split (see the file's header)
# txm_sampler.hip: the binomial splits of sampler stream 3 are IEEE double arithmetic in a fixed order that the CPU
# restatement (oracle/philox_oracle.c) repeats bit for bit -- no fused multiply-add contraction
# (-Wno-inline-asm: its store asm names M0 as clobbered -- it writes it -- and clang remarks on every instance that M0 is
# a reserved register)
_I8T = ["-mllvm", "-amdgpu-mfma-vgpr-form", "-Wno-inline-asm"]
# txm_resample_i8g.hip: 12 accumulator tiles (192 of the 256 registers).  The greedy register allocator assigns "global" live
# ranges first and the sixteen-register tiles into what is left; a stray scalar in the middle of the file then leaves no aligned
# run of sixteen and whole tiles are spilled (any small edit flipped the kernel between 2 and 200+ spilled registers).  With
# the widest register classes assigned first every instance builds with 0-9 spills, none inside the k-steps.
_I8G = _I8T + ["-mllvm", "-greedy-regclass-priority-trumps-globalness=1"]
EXTRA_FLAGS = {"txm_resample_i8t.hip": _I8T, "txm_resample_i8g.hip": _I8G, "txm_resample_i8gn.hip": _I8G, "txm_sampler.hip": ["-ffp-contract=off"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (Path(cand).exists() or cand == "hipcc"):
            return cand
    return "hipcc"


HEADER = CSRC.parent.parent / "include" / "txmom.h"   # the public C ABI (txm_common.h includes it)


def csrc_sha() -> str:
    """sha256 (16 hex digits) over what decides the built library: the kernel sources, the public header they include and
    the compiler flags -- what a built library carries (txm_csrc_sha) and what ties a committed profile to the code it was
    measured on (bench.py)."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h"))) + [HEADER]:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    h.update(repr((FLAGS, sorted(EXTRA_FLAGS.items()), SOURCES)).encode())
    return h.hexdigest()[:16]


def built_sha() -> str | None:
    """The source hash embedded in the built library (read from the file, no dlopen), None when there is none."""
    if not LIB.exists():
        return None
    blob = LIB.read_bytes()
    i = blob.find(b"TXM_CSRC_SHA=")
    if i < 0:
        return None
    return blob[i + 13:i + 29].decode(errors="replace")


def needs_build() -> bool:
    """True when there is no library or it was not compiled from the sources in the tree -- decided by the hash the
    library carries, not by file times (a checkout, a copy or a snapshot keeps no usable times)."""
    return built_sha() != csrc_sha()


def build_library(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    cc = _hipcc()
    sha = csrc_sha()

    def compile_one(src: str) -> Path:
        obj = objdir / (src.replace(".hip", ".o"))
        extra = list(EXTRA_FLAGS.get(src, []))
        if src == "txm_api.hip":
            extra.append(f'-DTXM_CSRC_SHA="{sha}"')
        cmd = [cc, *FLAGS, *extra, "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    p = build_library(force="--force" in sys.argv, verbose=True)
    print("built", p)
