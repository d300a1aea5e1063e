#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection CSVs (one directory per pass) into a JSON kept under
profiles/, and derive per-launch HBM traffic the way MI355X_MICROARCH.md prescribes for gfx950:
bytes = 2 * FETCH_SIZE * 1024 (FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads)
      + WRITE_SIZE * 1024.
usage: summarize_pmc.py <prof_dir> <out.json> <traffic.json> kernel_substring [...]"""
import collections
import csv
import glob
import json
import sys


def agg(prof_dir, kern):
    out = {}
    for path in glob.glob(f"{prof_dir}/*/*/*_counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                out["_meta"] = dict(vgpr=int(r["VGPR_Count"]), agpr=int(r["Accum_VGPR_Count"]), sgpr=int(r["SGPR_Count"]),
                                    lds=int(r["LDS_Block_Size"]), scratch=int(r["Scratch_Size"]), grid=int(r["Grid_Size"]),
                                    wg=int(r["Workgroup_Size"]))
        for k, v in acc.items():
            out[k] = dict(mean_per_launch=sum(v) / len(v), launches=len(v))
    return out


def main():
    prof_dir, out_json, traffic_json, *kernels = sys.argv[1:]
    summary, traffic = {}, {}
    for k in kernels:
        a = agg(prof_dir, k)
        d = dict(counters=a)
        if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
            rd = 2 * a["FETCH_SIZE"]["mean_per_launch"] * 1024
            wr = a["WRITE_SIZE"]["mean_per_launch"] * 1024
            d["hbm_bytes_per_launch"] = dict(read=rd, write=wr, total=rd + wr,
                                             note="read = 2 x FETCH_SIZE KB (gfx950 correction), write = WRITE_SIZE KB")
            traffic[k] = rd + wr
        if "GRBM_GUI_ACTIVE" in a:
            d["gpu_cycles_per_launch"] = a["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8  # counter sums the 8 XCDs
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "GRBM_GUI_ACTIVE" in a and a["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] > 0:
            d["mfma_pipe_busy_fraction"] = a["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / 1024 / d["gpu_cycles_per_launch"]
        summary[k] = d
    json.dump(summary, open(out_json, "w"), indent=1)
    json.dump(traffic, open(traffic_json, "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in summary.items()}, indent=1))


if __name__ == "__main__":
    main()
