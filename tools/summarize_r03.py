#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_r03.sh (gpurun_out/prof_r03) into the files kept under profiles/:
  r03_bench_kernel_stats.csv     --kernel-trace --stats of `python3 bench.py` (names truncated)
  r03_bench_kernel_launches.csv  every dispatch of the dominant kernel in that run (duration, grid, registers)
  r03_pmc_summary.json           the rollout kernel's counters at two launch lengths; HBM bytes per launch = 2 x FETCH_SIZE KB +
                                 WRITE_SIZE KB (gfx950: FETCH_SIZE counts 64 of each 128-byte request, MI355X_MICROARCH.md HBM section)
  traffic.json                   bytes per environment-step and per environment-launch fitted to those two lengths (read by bench.py)
  r03_kernel_bench.json          HBM-bound entry points with buffers rotated through a 640 MB pool: duration by rocprofv3, counter bytes
                                 against the algorithmic bytes
usage: summarize_r03.py [prof_dir] [profiles_dir]"""
import collections
import csv
import glob
import json
import re
import sys


def rows(pattern):
    out = []
    for path in glob.glob(pattern):
        out += list(csv.DictReader(open(path)))
    return out


def short(name, n=110):
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def counter_means(prof_dir, sub, kernel_sub):
    acc = collections.defaultdict(list)
    meta = {}
    for r in rows(f"{prof_dir}/{sub}/*/*_counter_collection.csv"):
        if kernel_sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = dict(vgpr=int(r["VGPR_Count"]), agpr=int(r["Accum_VGPR_Count"]), sgpr=int(r["SGPR_Count"]), lds=int(r["LDS_Block_Size"]),
                        scratch=int(r["Scratch_Size"]), grid=int(r["Grid_Size"]), wg=int(r["Workgroup_Size"]))
    return {k: dict(mean_per_launch=sum(v) / max(1, len(v)), launches=len(v)) for k, v in acc.items()}, meta


def main():
    prof = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r03"
    dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"
    B = 1 << 20
    ROLL = "k_vhjb_rollout_mfma<0, hjbx::Cartpole"
    # ---- bench trace ---------------------------------------------------------------------------------------------------------
    st = rows(f"{prof}/trace/*/*_kernel_stats.csv")
    st.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(f"{dst}/r03_bench_kernel_stats.csv", "w", newline="") as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py   (round 3; the timed region is 10 x one 200-step launch of "
                "k_vhjb_rollout_mfma<0, Cartpole, 8, 0, 0> = the float32 MFMA kernel (the library default); its other launches are the clock pre-warm (25 steps, no logs) and "
                "the 20 warm-up steps; <..., 0, 1> / <..., 0, 2> are the opt-in bf16x3 / f16x2 kernels of the secondary block)\n")
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in st:
            w.writerow([short(r["Name"])] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
    HEAD = "8, 0, 0>"      # ..., WAVES = 8, ACT = relu, ARITH = 0 (float32 MFMA, the default arithmetic = the headline kernel)
    tr = [r for r in rows(f"{prof}/trace/*/*_kernel_trace.csv") if ROLL in r["Kernel_Name"] and HEAD in r["Kernel_Name"].split("(")[0]]
    tr.sort(key=lambda r: int(r["Dispatch_Id"]))
    with open(f"{dst}/r03_bench_kernel_launches.csv", "w", newline="") as f:
        f.write("# every dispatch of k_vhjb_rollout_mfma<0, hjbx::Cartpole<float>, 8, 0, 2> (the float32 MFMA kernel) in the run above, in order (ns)\n")
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "DurationNs", "Grid_Size_X", "Workgroup_Size_X", "VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size"])
        for r in tr:
            w.writerow([r["Dispatch_Id"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Grid_Size_X"], r["Workgroup_Size_X"], r["VGPR_Count"],
                        r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"]])
    long = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr)[-10:]
    # ---- rollout kernel counters at two launch lengths -------------------------------------------------------------------------
    summ = {}
    per = {}
    for K in (100, 20):
        fe, meta = counter_means(prof, f"pmc{K}/FETCH_SIZE", ROLL)
        wr, _ = counter_means(prof, f"pmc{K}/WRITE_SIZE", ROLL)
        rd_b = 2 * fe["FETCH_SIZE"]["mean_per_launch"] * 1024
        wr_b = wr["WRITE_SIZE"]["mean_per_launch"] * 1024
        per[K] = rd_b + wr_b
        summ[f"k_vhjb_rollout_mfma cartpole, {K} steps per launch, B=2^20"] = dict(
            meta=meta, FETCH_SIZE_KB=fe["FETCH_SIZE"], WRITE_SIZE_KB=wr["WRITE_SIZE"],
            hbm_bytes_per_launch=dict(read=rd_b, write=wr_b, total=rd_b + wr_b, note="read = 2 x FETCH_SIZE KB (gfx950 correction), write = WRITE_SIZE KB"),
            bytes_per_env_step_inclusive=(rd_b + wr_b) / (B * K))
    a = (per[100] - per[20]) / (80.0 * B)
    b = (per[20] - 20 * a * B) / B
    n = 4
    for arith in ("f32",):
        sq, meta = counter_means(prof, f"sq_{arith}/a", ROLL)
        sq2, _ = counter_means(prof, f"sq_{arith}/b", ROLL)
        sq.update(sq2)
        if not sq:
            continue
        d = dict(counters=sq, meta=meta)
        tile_steps = 100.0 * B / 32
        if "GRBM_GUI_ACTIVE" in sq and "SQ_VALU_MFMA_BUSY_CYCLES" in sq:
            cyc = sq["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8          # the counter sums the 8 XCDs
            d["gpu_cycles_per_launch"] = cyc
            d["mfma_pipe_busy_fraction"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / 1024 / cyc
            d["gpu_cycles_per_tile_step_per_simd"] = cyc * 1024 / tile_steps
            try:
                with open(f"{prof}/sq_{arith}_a.json") as f:
                    ms = json.loads(f.read())["roofline"]["avg_launch_ms"]
                d["launch_ms_under_counters"] = ms
                d["shader_clock_GHz"] = cyc / (ms * 1e-3) / 1e9
            except (OSError, ValueError, KeyError):
                pass
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU"):
            if k in sq:
                d[k.lower() + "_per_tile_step"] = sq[k]["mean_per_launch"] / tile_steps
        if "SQ_WAIT_INST_ANY" in sq and "SQ_WAVE_CYCLES" in sq:
            d["wave_cycles_waiting_fraction"] = sq["SQ_WAIT_INST_ANY"]["mean_per_launch"] / sq["SQ_WAVE_CYCLES"]["mean_per_launch"]
        if "SQ_LDS_BANK_CONFLICT" in sq and "SQ_LDS_IDX_ACTIVE" in sq and sq["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"] > 0:
            d["lds_bank_conflict_fraction"] = sq["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"] / sq["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"]
        summ[f"k_vhjb_rollout_mfma cartpole, 100 steps per launch, arithmetic {arith}: SQ / GRBM passes"] = d
    summ["fit"] = dict(bytes_per_env_step=a, bytes_per_env_launch=b, algorithmic_per_env_step=4.0 * (n + 2), algorithmic_per_env_launch=4.0 * (3 * n + 2),
                       note="bytes(launch of k steps) = B (k a + b), from the 100- and 20-step passes")
    summ["bench_trace"] = dict(timed_launches_ns=long, median_ms_per_step=sorted(long)[len(long) // 2] / 200 / 1e6 if long else None)
    json.dump(summ, open(f"{dst}/r03_pmc_summary.json", "w"), indent=1)
    json.dump({"k_vhjb_rollout_mfma/n4": dict(bytes_per_env_step=a, bytes_per_env_launch=b,
                                              source="profiles/r03_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, cartpole B=2^20, 100- and 20-step launches)")},
              open(f"{dst}/traffic.json", "w"), indent=1)
    # ---- the cooperative parameter-gradient kernel (hjbx_train_coop.hip): HBM bytes per sample and matrix-pipe occupancy -------------
    train = {}
    for sysname_, nst in (("Cartpole", 4), ("NearHover", 10)):
        K = f"k_train_coop<0, 0, 1, hjbx::{sysname_}"             # <residual mode, activation, tile split, system>: split 1 = large batches
        K4 = f"k_train_coop<0, 0, 4, hjbx::{sysname_}"            # split 4 = the reference's minibatch (four workgroups per tile)
        big = lambda r: int(r["Grid_Size"]) >= 256 * 256          # the B = 2^20 launches (256 workgroups of 256 threads)
        d = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            v = [float(r["Counter_Value"]) for r in rows(f"{prof}/train/{c}/*/*_counter_collection.csv") if K in r["Kernel_Name"] and big(r)]
            if v:
                d[c] = sum(v) / len(v)
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            rd, wr = 2 * d["FETCH_SIZE"] * 1024, d["WRITE_SIZE"] * 1024
            d.update(hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr, bytes_per_sample=(rd + wr) / B, algorithmic_input_bytes_per_sample=4.0 * (nst + 2),
                     note="read = 2 x FETCH_SIZE KB (gfx950 correction), write = WRITE_SIZE KB; B = 2^20 samples per launch; the rest of the traffic is the "
                          "weights (104 KB per workgroup) and the per-workgroup partial sums (0.2 MB per workgroup)")
        sq = collections.defaultdict(list)
        for r in rows(f"{prof}/train/sq/*/*_counter_collection.csv"):
            if K in r["Kernel_Name"] and big(r):
                sq[r["Counter_Name"]].append(float(r["Counter_Value"]))
        sq = {k: sum(v) / len(v) for k, v in sq.items()}
        if "GRBM_GUI_ACTIVE" in sq and "SQ_VALU_MFMA_BUSY_CYCLES" in sq:
            cyc = sq["GRBM_GUI_ACTIVE"] / 8
            d.update(gpu_cycles_per_launch=cyc, mfma_pipe_busy_fraction=sq["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, counters=sq)
        trows = rows(f"{prof}/train/trace/*/*_kernel_trace.csv")
        tr_ = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trows if K in r["Kernel_Name"] or K4 in r["Kernel_Name"]]
        if tr_:
            tr_.sort()
            nbig = [t for t in tr_ if t > 1_000_000]
            nsmall = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trows if K4 in r["Kernel_Name"])
            if nbig:
                d["rocprof_ms_B_2p20"] = nbig[len(nbig) // 2] / 1e6
                d["samples_per_s_B_2p20"] = B / (nbig[len(nbig) // 2] * 1e-9)
            if nsmall:
                d["rocprof_us_B_256"] = nsmall[len(nsmall) // 2] / 1e3
        train[f"k_train_coop<0, 0, PS, hjbx::{sysname_}<float>>"] = d
    json.dump(train, open(f"{dst}/r03_train_coop_summary.json", "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in train.items()}, indent=1))
    st2 = rows(f"{prof}/train/trace/*/*_kernel_stats.csv")
    st2.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(f"{dst}/r03_train_kernel_stats.csv", "w", newline="") as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/dev/time_train.py nearhover 1048576  (hjbx_value_loss_grad_f32 in its three "
                "implementations at B = 2^20 and B = 256, then params_update at 256)\n")
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in st2:
            w.writerow([short(r["Name"])] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
    # ---- HBM-bound entry points ----------------------------------------------------------------------------------------------
    kb = json.load(open(f"{prof}/kb.json"))
    stats = {r["Name"]: r for r in rows(f"{prof}/kb/trace/*/*_kernel_stats.csv")}
    fe_rows = rows(f"{prof}/kb/FETCH_SIZE/*/*_counter_collection.csv")
    wr_rows = rows(f"{prof}/kb/WRITE_SIZE/*/*_counter_collection.csv")

    def mean_counter(rws, key):
        v = [float(r["Counter_Value"]) for r in rws if key(r["Kernel_Name"])]
        return (sum(v) / len(v) if v else None), len(v)

    sysname = dict(cartpole="Cartpole", acrobot="Acrobot", quad2d="Quad2D", nearhover="NearHover")
    kern = {"simulate (euler)": "k_simulate<0, ", "simulate (rk4)": "k_simulate<1, ", "vhjb_step (euler)": "k_vhjb_step<0, ", "vhjb_step (rk4)": "k_vhjb_step<1, ",
            "hjb_residual fwd+bwd+sums": "k_hjb_residual<0, ", "controller": "k_controller<"}
    out = []
    for r in kb:
        if r["kernel"] not in kern:
            continue
        sub, sn = kern[r["kernel"]], f"hjbx::{sysname[r['system']]}<float>"

        def key(name, sub=sub, sn=sn):
            return name.startswith("void " + sub) and sn in name.split("(")[0]
        ks = [v for k, v in stats.items() if key(k)]
        avg_ns = sum(float(v["TotalDurationNs"]) for v in ks) / max(1, sum(int(v["Calls"]) for v in ks)) if ks else None
        fe, nfe = mean_counter(fe_rows, key)
        wr, nwr = mean_counter(wr_rows, key)
        alg = r["bytes_per_env"] * B
        rec = dict(system=r["system"], kernel=r["kernel"], algorithmic_bytes_per_env=r["bytes_per_env"], rocprof_avg_us=avg_ns / 1e3 if avg_ns else None,
                   hip_event_us_under_trace=r["us"], rotating_sets=r.get("rotating_sets"))
        if avg_ns:
            rec["achieved_GBs"] = alg / (avg_ns * 1e-9) / 1e9
            rec["frac_of_8TBs"] = rec["achieved_GBs"] / 8000.0
        if fe is not None and wr is not None:
            tot = 2 * fe * 1024 + wr * 1024
            rec.update(counter_read_bytes=2 * fe * 1024, counter_write_bytes=wr * 1024, counter_bytes_over_algorithmic=tot / alg, counter_launches=[nfe, nwr])
        out.append(rec)
    json.dump(out, open(f"{dst}/r03_kernel_bench.json", "w"), indent=1)
    print(json.dumps(summ["fit"], indent=1))
    print(json.dumps(summ["bench_trace"]))
    for k, v in summ.items():
        if "SQ / GRBM" in k:
            print(k, {kk: vv for kk, vv in v.items() if kk not in ("counters", "meta")})
    for r in out:
        print(f"{r['system']:10s} {r['kernel']:28s} {r['rocprof_avg_us'] or 0:7.2f} us  {r.get('achieved_GBs', 0):7.0f} GB/s  {100 * r.get('frac_of_8TBs', 0):5.1f} %  "
              f"counter/alg {r.get('counter_bytes_over_algorithmic', 0):.3f}")


if __name__ == "__main__":
    main()
