#!/usr/bin/env python3
"""Markdown tables for DESIGN.md section 6 from the report tests/test_gpu_f32_parity.py writes (profiles/r03_f32_parity_report.json).
    python tools/parity_tables.py [report.json]"""
import json
import sys

rep = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "profiles/r03_f32_parity_report.json"))
f = lambda v: f"{v:.1e}".replace("e-0", "e-")
print("| config (B), integrator, weights | at a ReLU kink: envs (matching neither side) | x′ kernel · CPU-f32 (max, relative to the element's term scale) | u | cost | residual | worst kernel / CPU-f32 ratio (max, p99.9) |")
print("|---|---|---|---|---|---|---|")
for key in sorted(k for k in rep if k.startswith("teacher_forced/f32/")):
    _, _, system, integ, weights = key.split("/")
    r = rep[key]
    cells, worst_m, worst_p = [], 0.0, 0.0
    for q in ("x_next", "u", "cost", "residual"):
        k, c = r[q]["kernel"], r[q]["cpu_f32"]
        cells.append(f"{f(k['max'])} · {f(c['max'])}")
        worst_m, worst_p = max(worst_m, k["max"] / c["max"]), max(worst_p, k["p999"] / c["p999"])
    print(f"| {system} (2^{r['B'].bit_length() - 1}), {integ}, {weights} | {r['at_kink_envs']} ({r['at_kink_matching_no_side']}) | " + " | ".join(cells) + f" | {worst_m:.2f}, {worst_p:.2f} |")
print()
for arith in ("bf16x3", "f16x2"):
    wm = wp = 0.0
    for key in (k for k in rep if k.startswith(f"teacher_forced/{arith}/")):
        for q in ("x_next", "u", "cost", "residual"):
            k, c = rep[key][q]["kernel"], rep[key][q]["cpu_f32"]
            wm, wp = max(wm, k["max"] / c["max"]), max(wp, k["p999"] / c["p999"])
    print(f"{arith}: worst kernel / CPU-f32 ratio over all cases and quantities: max {wm:.2f}, p99.9 {wp:.2f}")
print()
print("| system, integrator | environments compared (away from the box faces) | kernel mismatches (CPU-f32 mismatches) | of those explained by a kink side | inside the band: kernel (CPU-f32) |")
print("|---|---|---|---|---|")
for key in sorted(k for k in rep if k.startswith("done_step/f32/")):
    _, _, system, integ = key.split("/")
    r = rep[key]
    print(f"| {system}, {integ} | {100 * (1 - r['filtered_fraction']):.1f} % of 2^{r['B'].bit_length() - 1} | {r['mismatches_in_safe']} ({r['cpu_f32_mismatches_in_safe']}) | "
          f"{r['mismatches_in_safe_reproduced_by_the_other_side_of_a_kink']} | {r['mismatches_in_band']} ({r['cpu_f32_mismatches_in_band']}) |")
print()
print("| system, integrator | t | kernel median / p99 / max | CPU-f32 median / p99 (as fractions of the step's bound) | kernel median / p99 (fractions of the bound) |")
print("|---|---|---|---|---|")
for key in sorted(k for k in rep if k.startswith("closed_loop_T200/f32/")):
    _, _, system, integ = key.split("/")
    for t in ("1", "10", "50", "100", "200"):
        c = rep[key]["curve"].get(t)
        if c:
            print(f"| {system}, {integ} | {t} | {f(c['median'])} / {f(c['p99'])} / {f(c['max'])} | {c['cpu_f32_median_over_bound']:.3f} / {c['cpu_f32_p99_over_bound']:.3f} | "
                  f"{c['median_over_bound']:.3f} / {c['p99_over_bound']:.3f} |")
