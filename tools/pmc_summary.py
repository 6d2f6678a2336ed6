"""Aggregate rocprofv3 PMC passes (csv output) into profiles/rNN_pmc_counters.json — per kernel symbol, per-launch averages.

Collection (each pass on its own, never together with sys/hip/hsa traces):
  rocprofv3 --pmc FETCH_SIZE  --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python bench.py --steps 2 --warmup 1 --no-probe --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE  --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python bench.py ...
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES \\
            --kernel-trace --output-format csv -d gpurun_out/pmc_sq -o s -- python bench.py ...
  python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq profiles/r01_pmc_counters.json

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-reports wide
(16 B/lane) coalesced reads by 2x -> read bytes = 2 * FETCH_SIZE * 1024 (checked here on bn_bwd_apply / colreduce, whose traffic is
known exactly); Infinity-Cache hits are included in both.  mfma_pipe_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8
XCDs); eff_clock = GRBM_GUI_ACTIVE / 8 / duration."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

NOTE = ("rocprofv3 PMC passes over `bench.py --steps 2 --warmup 1 --no-probe --no-cpu-baseline --no-sub` (3 steps of the default workload: the joint segmenter+GAN step): pass 1 --pmc FETCH_SIZE, pass 2 --pmc "
        "WRITE_SIZE, pass 3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
        "SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES. KB units; gfx950 FETCH_SIZE under-reports wide (16 B/lane) coalesced reads by 2x "
        "(MI355X_MICROARCH.md, HBM section) -> read bytes = 2*FETCH_SIZE*1024 (calibrated on bn_bwd_apply / colreduce, whose traffic is known "
        "exactly); Infinity-Cache hits are included; per-launch averages per kernel symbol. mfma_pipe_util = SQ_VALU_MFMA_BUSY_CYCLES / "
        "(1024 SIMDs x GRBM_GUI_ACTIVE/8 XCDs); eff_clock = GRBM_GUI_ACTIVE/8/duration.  Built by tools/pmc_summary.py.")


def clean(name):
    """'void (anonymous namespace)::k<...>((anonymous namespace)::Args)' -> 'k<...>' (same shortening as tools/rocpd_summary.py)"""
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def read_pass(folder):
    """-> {kernel: {counter: [values]}, "_dur": {kernel: [ns]}}"""
    files = glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no *_counter_collection.csv under %s" % folder)
    vals = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = clean(row["Kernel_Name"])
                vals[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                dur[k][row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    return vals, dur


def avg(xs):
    return sum(xs) / len(xs) if xs else None


def main():
    fetch_dir, write_dir, sq_dir, out = sys.argv[1:5]
    fv, _ = read_pass(fetch_dir)
    wv, _ = read_pass(write_dir)
    sv, sdur = read_pass(sq_dir)
    kernels = {}
    for k in sorted(set(fv) | set(wv) | set(sv)):
        e = {}
        f, w = avg(fv.get(k, {}).get("FETCH_SIZE", [])), avg(wv.get(k, {}).get("WRITE_SIZE", []))
        e["launches"] = len(fv.get(k, {}).get("FETCH_SIZE", [])) or len(sv.get(k, {}).get("GRBM_GUI_ACTIVE", []))
        if f is not None:
            e["FETCH_SIZE_KB_avg"] = round(f, 1)
            e["hbm_read_MB_per_launch_corrected_x2"] = round(2 * f * 1024 / 1e6, 2)
        if w is not None:
            e["WRITE_SIZE_KB_avg"] = round(w, 1)
            e["hbm_write_MB_per_launch"] = round(w * 1024 / 1e6, 2)
        s = sv.get(k, {})
        if s.get("GRBM_GUI_ACTIVE"):
            d_us = avg(list(sdur[k].values())) / 1e3
            gui = avg(s["GRBM_GUI_ACTIVE"])
            e["avg_dur_us_profiled"] = round(d_us, 1)
            e["eff_clock_GHz"] = round(gui / 8 / (d_us * 1e3), 3)
            if s.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                e["mfma_pipe_util"] = round(avg(s["SQ_VALU_MFMA_BUSY_CYCLES"]) / (1024 * gui / 8), 3)
            if s.get("SQ_LDS_BANK_CONFLICT"):
                e["lds_bank_conflict_cycles"] = round(avg(s["SQ_LDS_BANK_CONFLICT"]), 1)
            for name in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES"):
                if s.get(name):
                    e[name + "_avg"] = round(avg(s[name]), 1)
        kernels[k] = e
    with open(out, "w") as fh:
        note = NOTE + ("  " + os.environ["PMC_NOTE"] if os.environ.get("PMC_NOTE") else "")
        json.dump({"note": note, "kernels": kernels}, fh, indent=1)
    print("wrote %s (%d kernels)" % (out, len(kernels)))


if __name__ == "__main__":
    main()
