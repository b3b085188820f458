"""Fold the rocprofv3 --pmc passes written by tools/pmc_conv.sh into one table per level.

usage: pmc_summary.py PMC_DIR [LEVEL ...] [--json OUT]
Per level: mean counter value per dispatch of the conv kernel (first dispatch dropped as warm-up), mean duration, and the
derived figures the roofline discussion uses (units per MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*
count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES count cycles; FETCH_SIZE / WRITE_SIZE are
KiB, and gfx950's FETCH_SIZE reads 1/2 of a wide coalesced stream -> doubled here, as that guide prescribes).
"""
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)


def load_pass(d, want=("conv_mfma_kernel", "conv3_pipe_kernel", "conv3_wreg_kernel")):
    vals, dur = {}, {}
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if not any(w in r["Kernel_Name"] for w in want):
                continue
            did = int(r["Dispatch_Id"])
            vals.setdefault(r["Counter_Name"], {})[did] = float(r["Counter_Value"])
            if r.get("End_Timestamp"):
                dur[did] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out = {}
    for name, by in vals.items():
        ids = sorted(by)[1:] or sorted(by)
        out[name] = sum(by[i] for i in ids) / len(ids)
    ids = sorted(dur)[1:] or sorted(dur)
    if ids:
        out["_dur_ns"] = sum(dur[i] for i in ids) / len(ids)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    root = args[0]
    levels = [int(a) for a in args[1:]] or [0, 1, 2]
    res = {}
    for lvl in levels:
        c = {}
        durs = {}
        for p in ("sq1", "sq2", "sq3", "grbm", "fetch", "write"):
            d = load_pass(os.path.join(root, f"l{lvl}_{p}"))
            if "_dur_ns" in d:
                durs[p] = d.pop("_dur_ns")
            c.update(d)
        if not c:
            continue
        row = {"counters": c, "dur_us_by_pass": {k: v / 1e3 for k, v in durs.items()}}
        dur = durs.get("sq1") or next(iter(durs.values()))
        n_simd = 256 * 4
        der = {}
        if "SQ_BUSY_CYCLES" in c:
            der["sq_busy_cycles_per_se"] = c["SQ_BUSY_CYCLES"]
        if "GRBM_GUI_ACTIVE" in c and "grbm" in durs:
            der["clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / durs["grbm"]  # summed over the 8 XCDs (guide: DVFS give-back)
        clk = der.get("clock_ghz", 2.4)
        cyc = dur * clk  # shader cycles of one dispatch
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            der["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (n_simd * cyc)
        if "SQ_ACTIVE_INST_VALU" in c:
            der["valu_active_frac"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (n_simd * cyc)
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if k in c:
                    der[k.lower() + "_of_wave_cycles"] = c[k] / wc
            der["waves_resident_per_simd"] = 4.0 * wc / (n_simd * cyc)
        if "SQ_INSTS_VALU" in c and "SQ_INSTS_VALU_MFMA_MOPS_BF16" in c:
            der["valu_insts_per_wave_total"] = c["SQ_INSTS_VALU"]
        if "FETCH_SIZE" in c:
            der["hbm_read_mb"] = 2.0 * c["FETCH_SIZE"] * 1024 / 1e6
        if "WRITE_SIZE" in c:
            der["hbm_write_mb"] = c["WRITE_SIZE"] * 1024 / 1e6
        if "hbm_read_mb" in der and "hbm_write_mb" in der:
            der["hbm_traffic_mb"] = der["hbm_read_mb"] + der["hbm_write_mb"]
        row["derived"] = der
        res[f"level{lvl}"] = row
        print(f"== level {lvl}: dispatch {dur / 1e3:.1f} us (sq1 pass)")
        for k in sorted(c):
            print(f"   {k:36s} {c[k]:16.0f}")
        for k, v in der.items():
            print(f"   -> {k:33s} {v:12.4f}")
    for a in sys.argv[1:]:
        if a.startswith("--json="):
            json.dump(res, open(a[7:], "w"), indent=1)


if __name__ == "__main__":
    main()
