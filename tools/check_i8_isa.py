"""Verify the emitted ISA of resample_i8_kernel: the accumulator tiles that are pinned to
VGPRs / AGPRs through asm constraints must be touched by nothing but their own MFMA inside
the k-step loop (a compiler-made copy or spill right behind an asm MFMA would read the
destination before the matrix pipe wrote it -- the hazard recognizer skips asm).

    python tools/check_i8_isa.py            # compiles txm_resample_i8.hip to ISA and checks
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "thermoextrap_amd/csrc/txm_resample_i8.hip"
REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def kernels(asm):
    cur, name = [], None
    for line in asm.splitlines():
        m = re.match(r"^(_ZN3txm18resample_i8_kernel\w+):", line)
        if m:
            name, cur = m.group(1), []
        if name:
            cur.append(line)
            if "s_endpgm" in line:
                yield name, cur
                name = None


def check(name, lines):
    mf = [(i, l) for i, l in enumerate(lines) if "v_mfma_i32_32x32x32_i8" in l]
    pinned = set()
    asm_lines = set()
    for i, l in mf:
        ops = [o.strip() for o in l.split("v_mfma_i32_32x32x32_i8")[1].split(",")]
        # builtin tiles live in a[..] and come with hazard handling; the asm ones are those whose
        # dst is a VGPR range, plus the single AGPR tile that shares their A operand register
        if ops[0].startswith("v["):
            pinned |= regs(ops[0])
            asm_lines.add(i)
    # the one asm tile that sits in an AGPR: the MFMA emitted just before a run of VGPR-dst
    # asm MFMAs, with the same A operand register
    mf_idx = [i for i, _ in mf]
    for k, (i, l) in enumerate(mf):
        if i in asm_lines and k > 0 and mf_idx[k - 1] not in asm_lines:
            pl = mf[k - 1][1]
            ops = [o.strip() for o in pl.split("v_mfma_i32_32x32x32_i8")[1].split(",")]
            mine = [o.strip() for o in l.split("v_mfma_i32_32x32x32_i8")[1].split(",")]
            if ops[0].startswith("a[") and ops[1] == mine[1]:
                pinned |= regs(ops[0])
                asm_lines.add(mf_idx[k - 1])
    if not pinned:
        return f"{name}: no asm-pinned tiles (K < 4)", True
    bars = [i for i, l in enumerate(lines) if "s_barrier" in l]
    # each k-step is fenced by s_barrier: check every barrier-to-barrier span that holds asm MFMAs
    labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\w+):", l))}
    back = []  # backward branches close a loop body (the compiler rotates the end-of-step barrier to the loop top)
    for i, l in enumerate(lines):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\w+)", l)
        if m and labels.get(m.group(1), 1 << 30) < i:
            back.append(i)
    spans = set()
    for i in asm_lines:
        lo = max([b for b in bars if b < i], default=0)
        hi = min([b for b in bars + back if b > i], default=len(lines) - 1)
        spans.add((lo, hi))
    bad = []
    for lo, hi in sorted(spans):
        for i in range(lo, hi + 1):
            l = lines[i].split(";")[0]
            if i in asm_lines or not l.strip() or l.strip().startswith("."):
                continue
            if regs(l) & pinned:
                bad.append((i, l.strip()))
    lo, hi = min(s_[0] for s_ in spans), max(s_[1] for s_ in spans)
    ok = not bad
    msg = f"{name}: {len(pinned)} pinned registers, {len(spans)} k-step spans in lines {lo}-{hi}, {'clean' if ok else 'TOUCHED:'}"
    for i, l in bad[:10]:
        msg += f"\n    {i}: {l}"
    return msg, ok


def main():
    with tempfile.TemporaryDirectory() as d:
        out = Path(d) / "i8.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "-S",
                        "--cuda-device-only", str(SRC), "-o", str(out)], check=True, capture_output=True)
        asm = out.read_text()
    allok = True
    for name, lines in kernels(asm):
        msg, ok = check(name, lines)
        print(msg)
        allok &= ok
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main())
