"""Summarise single-counter rocprofv3 --pmc passes over tools/probe_unet.py into per-kernel SQ metrics.

usage: pmc_sq.py <dir with one sub-directory per counter, each holding *counter_collection.csv> <out.json>

MFMA busy: SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs (= 32 per v_mfma_f32_32x32x16_bf16, 16 per
16x16x32); GRBM_GUI_ACTIVE counts the cycles the launch was on the chip once per XCD (summed over the 8 XCDs: per
launch it comes to 8 x duration x 1.93 GHz on the big conv launches), so
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)
= the fraction of SIMD cycles, at the clock the launch actually ran at, in which the matrix pipe was busy.
SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles over all waves: wait_any and wait_inst_any are given as fractions of the
wave cycles; lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.
"""
import collections
import csv
import glob
import json
import os
import sys


def load(root, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(root, counter, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            agg[name][0] += 1
            agg[name][1] += float(r["Counter_Value"])
    return agg


def main():
    root, out = sys.argv[1:3]
    names = ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT",
             "SQ_LDS_IDX_ACTIVE"]
    data = {n: load(root, n) for n in names}
    kernels = {}
    for k in sorted(data["GRBM_GUI_ACTIVE"]):
        if "bsmi" not in k:
            continue
        g = lambda n: data[n].get(k, [0, 0.0])[1]  # noqa: E731
        launches = data["GRBM_GUI_ACTIVE"][k][0]
        row = {"launches": launches, "chip_cycles_per_launch": g("GRBM_GUI_ACTIVE") / 8 / max(launches, 1)}
        if g("GRBM_GUI_ACTIVE") > 0:
            row["mfma_busy"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") / 8 * 256 * 4)
        if g("SQ_WAVE_CYCLES") > 0:
            row["wait_any"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
            row["wait_inst_any"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
        if g("SQ_LDS_IDX_ACTIVE") > 0:
            row["lds_conflict"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
        kernels[k] = row
    json.dump({"source": "rocprofv3 --pmc <one counter per pass> -- python3 tools/probe_unet.py", "kernels": kernels}, open(out, "w"), indent=1)
    for k, r in kernels.items():
        if "conv" in k or "first_pass" in k or "wino" in k:
            print(k[:70].ljust(70), {a: round(b, 3) for a, b in r.items() if a != "chip_cycles_per_launch"})


if __name__ == "__main__":
    main()
