#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of the dominant kernel (B = A diag(d) A^T) into profiles/rNN_pmc_form_kernel.json.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-netlib
    rocprofv3 --kernel-trace --pmc WRITE_SIZE ...   (one pass per counter set: FETCH_SIZE and WRITE_SIZE do not fit together)
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE ...
    python3 tools/pmc_form_kernel.py --out profiles/r02_pmc_form_kernel.json --shape 4096 8192 gpurun_out/pmc_*

HBM-side traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes): on gfx950 FETCH_SIZE reports half of a wide
coalesced read (MI355X_MICROARCH.md, HBM section; calibrated here on gemv_n_kernel, whose FETCH_SIZE is A/2).  The
file carries the sha of the kernel source it was collected with; bench.py refuses it for any other source."""
import argparse
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--out", required=True)
    ap.add_argument("--shape", type=int, nargs=2, default=[4096, 8192])
    ap.add_argument("--kernel", default="adat_syrk_kernel", help="adat_syrk_kernel (serial path) or form_factor_kernel (fused path)")
    args = ap.parse_args()
    import bench
    acc = collections.defaultdict(list)
    calib = collections.defaultdict(list)
    for d in args.dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                nm = r["Kernel_Name"]
                if args.kernel in nm:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                elif "gemv_n_kernel" in nm and int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) >= 256 * 1024:
                    calib[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {k: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for k, v in acc.items()}
    m, n = args.shape
    der = {}
    if "FETCH_SIZE" in acc and "WRITE_SIZE" in acc:
        rd = 2.0 * out["FETCH_SIZE"]["mean_per_launch"] * 1024.0
        wr = out["WRITE_SIZE"]["mean_per_launch"] * 1024.0
        der["hbm_read_bytes_per_launch (2 x FETCH_SIZE x 1024)"] = rd
        der["hbm_write_bytes_per_launch (WRITE_SIZE x 1024)"] = wr
        der["traffic_bytes_per_launch"] = rd + wr
        der["algorithmic_bytes_per_launch (A once 8mn + lower B 4m^2)"] = 8.0 * m * n + 4.0 * m * m
    if "FETCH_SIZE" in calib:
        # gemv_n_kernel also serves the grouped triangular solves (reads of L, a fraction of A's bytes; at 16384 rows their grids pass
        # the size filter above too): the passes over ALL of A are the launches with the largest FETCH_SIZE -- median of those
        # within 2x of the maximum
        top = max(calib["FETCH_SIZE"])
        full = sorted(v for v in calib["FETCH_SIZE"] if v >= 0.5 * top)
        der["calibration: gemv_n_kernel (passes over A) FETCH_SIZE x 1024 / (8 mp np)"] = full[len(full) // 2] * 1024.0 / (8.0 * m * n)
        der["calibration_launches"] = len(full)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in acc and "GRBM_GUI_ACTIVE" in acc:
        der["mfma_busy_fraction (SQ_VALU_MFMA_BUSY_CYCLES/1024 SIMDs / (GRBM_GUI_ACTIVE/8 XCDs))"] = \
            out["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / 1024.0 / (out["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0)
    out["derived"] = der
    out["shape"] = [m, n]
    out["kernel"] = args.kernel
    out["kernel_source_sha"] = bench.kernel_source_sha(fused="form_factor" in args.kernel)
    if "form_factor" in args.kernel:
        out["note"] = ("the launch runs BESIDE the pivot chain's kernels and the residual stream (other queues): the per-dispatch SQ / GRBM "
                       "counters of overlapping dispatches are not separable, only the TCC byte counters are used; MFMA occupancy of this "
                       "launch: its in-kernel cycle profile (profiles/r03_ff_cycle_profile.txt)")
        out["derived"] = {k: v for k, v in out["derived"].items() if not k.startswith("mfma_busy")}
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
