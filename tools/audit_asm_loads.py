#!/usr/bin/env python3
"""Audit of the inline-asm LDS prefetch in hjbx_mlp.hip (guide section 5.7, item 1): between an asm `ds_read_b32 vX` /
`ds_read_b64 v[X:Y]` / `ds_read_b128 v[X:Y]` and the counted `s_waitcnt lgkmcnt(N)` that retires it, no
compiler-generated instruction may touch the destination register(s) (a copy or spill there would read them before the
data has landed).  lgkmcnt counts instructions, so a wide read is one entry.  Scans the device assembly of every kernel.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -S --cuda-device-only -DHJBX_MLP_ACT=0 -o /tmp/mlp.s csrc/hjbx_mlp.hip
    python tools/audit_asm_loads.py /tmp/mlp.s
"""
import re
import sys


def regs_of(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def audit(path):
    kernel, in_asm, pending, bad, nreads, nkern = None, False, [], 0, 0, 0
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel, pending = m.group(1), []
            nkern += 1
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        if in_asm:
            m = re.match(r"ds_read_b(?:32|64_tr_b16|64|128) (v\d+|v\[\d+:\d+\]),", s)
            if m:
                pending.append(frozenset(regs_of(m.group(1))))
                nreads += 1
            continue
        if s.startswith("s_endpgm"):
            pending = []
            continue
        if s.startswith("s_waitcnt"):   # the counted waits are the compiler-visible builtin; LDS returns in order
            m = re.search(r"lgkmcnt\((\d+)\)", s)
            if m:
                keep = int(m.group(1))
                pending = pending[len(pending) - keep:] if keep else []
            continue
        touched = regs_of(s.split(";")[0]) & set().union(*pending)
        if touched:
            bad += 1
            print(f"{path}:{ln}: [{kernel[:50]}] compiler instruction touches pending asm-load register(s) {sorted(touched)}: {s}")
    print(f"{nkern} functions, {nreads} asm ds_read, {bad} violations")
    return bad


if __name__ == "__main__":
    sys.exit(1 if audit(sys.argv[1]) else 0)
