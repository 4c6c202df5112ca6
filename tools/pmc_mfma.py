"""Summarise a rocprofv3 --pmc pass of the matrix-busy counters into counted MFMA utilisation per kernel.

usage: pmc_mfma.py PMC_DIR OUT.json
PMC_DIR holds the counter_collection.csv of `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3
bench.py ...` (program directly after `--`; counters in their own pass, no trace domains).  Per kernel (summed over its
launches):
  mfma_busy      = SQ_VALU_MFMA_BUSY_CYCLES   cycles in which a SIMD's matrix pipe was busy, summed over all SIMDs
  gui_active     = GRBM_GUI_ACTIVE            kernel cycles, summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
  mfma_util      = mfma_busy / (gui_active / 8 * 1024 SIMDs)     fraction of SIMD-cycles with the matrix pipe busy
For an fp32 MFMA (64 cycles per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_16x16x4_f32, both 64 FLOP/clk/SIMD) mfma_util is
directly the fraction of the fp32 matrix peak at the clock the kernel ran at; FLOP / time (bench.py `roofline`) is the same
fraction at the 2.4 GHz peak clock, so the two differ by the clock the chip held.
"""
import csv, glob, json, os, sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"])
                if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    n[row["Kernel_Name"]] += 1
    out = {}
    for k, c in acc.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if not gui:
            continue
        out[k] = {"launches": n[k], "mfma_busy_cycles": busy, "gui_active_sum_over_xcds": gui, "sq_busy_cycles": c.get("SQ_BUSY_CYCLES", 0.0),
                  "kernel_cycles_per_launch": gui / 8.0 / max(1, n[k]), "mfma_util": busy / (gui / 8.0 * 1024.0)}
    with open(sys.argv[2], "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["gui_active_sum_over_xcds"])[:25]:
        print("mfma_util %.3f  cycles/launch %9.0f  (n=%d)  %s" % (v["mfma_util"], v["kernel_cycles_per_launch"], v["launches"], k[:110]))


if __name__ == "__main__":
    main()
