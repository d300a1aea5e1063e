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


# ---- second audit: asm VALU statements reading MFMA results in flight (guide 5.7 item 2; DESIGN.md 4.5 "a hazard hipcc does not see") ----
# hipcc's hazard recogniser pads wait states between an MFMA and a COMPILER instruction that reads its result, but it does not look inside
# inline asm: an asm VALU statement (the mask helpers, v_fma_mix splits) placed behind the last MFMA of a chain reads half-written
# accumulators.  Rule checked here, per function, in program order: after `v_mfma... vDST` every asm VALU instruction that names a register
# of vDST must be separated from that MFMA by at least PASSES + 3 wait states (8-pass MFMA: 11, 16-pass: 19 -- what mfma_results_barrier
# provides), counting one per issued instruction and N + 1 per `s_nop N`.  (Counting an MFMA as one state is conservative: it issues for
# 4 x passes cycles when the pipe is busy.)
def _mfma_passes(mn):
    if "32x32x2_f32" in mn or "32x32x2f32" in mn:
        return 16
    if "32x32" in mn:
        return 8
    if "16x16x4_f32" in mn:
        return 8
    return 4


def audit_mfma_asm_reads(path):
    kernel, in_asm, inflight, bad, nasm, nkern, nmfma = None, False, [], 0, 0, 0, 0   # inflight: [regs, wait states still needed]
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel, inflight = m.group(1), []
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
        code = s.split(";")[0].strip()
        if not code or code.endswith(":"):
            continue
        if in_asm and code.startswith("v_") and not code.startswith("v_mfma"):
            nasm += 1
            ops = code.split(None, 1)[1] if " " in code else ""
            srcs = regs_of(ops.split(",", 1)[1]) if "," in ops else set()
            for regs, need in inflight:
                hit = srcs & regs
                if hit and need > 0:
                    bad += 1
                    print(f"{path}:{ln}: [{kernel[:50]}] asm VALU reads MFMA result register(s) {sorted(hit)[:4]} with {need} wait states still missing: {code}")
        states = 1
        m = re.match(r"s_nop (\d+)", code)
        if m:
            states = int(m.group(1)) + 1
        inflight = [[r, n - states] for r, n in inflight if n - states > 0]
        m = re.match(r"(v_mfma_\w+) (v\[\d+:\d+\]|a\[\d+:\d+\])", code)
        if m:
            nmfma += 1
            if m.group(2).startswith("v"):
                inflight.append([frozenset(regs_of(m.group(2))), _mfma_passes(m.group(1)) + 3])
    print(f"{nkern} functions, {nmfma} MFMAs, {nasm} asm VALU instructions, {bad} reads of MFMA results in flight")
    return bad


if __name__ == "__main__":
    sys.exit(1 if (audit(sys.argv[1]) + audit_mfma_asm_reads(sys.argv[1])) else 0)
