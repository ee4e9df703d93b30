#!/usr/bin/env python3
"""Summarise rocprofv3 SQ / GRBM counter passes of `bench.py` per kernel (the evidence behind the "VALU x % busy, waves
wait y % of their life, z % of the LDS cycles are bank conflicts" sentences of DESIGN.md).

One counter group per rocprofv3 pass (8 SQ slots per pass on gfx950; asking for more aborts the tool), program directly
after `--`:

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
              SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d gpurun_out/sq1 -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS \
              SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/sq2 -- ...
    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE ... -d gpurun_out/sq3 -- ...
    python3 profiles/collect_sq.py <workload> gpurun_out/sq1 gpurun_out/sq2 [gpurun_out/sq3] > profiles/r02_<workload>_sq.json

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over all waves;
GRBM_GUI_ACTIVE is summed over the 8 XCDs.  Derived per kernel (averages over its launches):
  waves_per_simd   = SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs / 4)
  valu_busy        = SQ_ACTIVE_INST_VALU / that denominator       (one wave-instruction occupies a SIMD for one quad-cycle)
  wait_frac        = SQ_WAIT_ANY / SQ_WAVE_CYCLES                 (parked at s_waitcnt / barrier)
  issue_stall_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(dirname):
    f = sorted(glob.glob(f"{dirname}/*/*_counter_collection.csv"), key=os.path.getmtime)[-1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def main():
    workload, dirs = sys.argv[1], sys.argv[2:]
    merged = collections.defaultdict(dict)
    for d in dirs:
        for k, c in per_kernel(d).items():
            merged[k].update(c)
    out = {"workload": workload, "unit": "counter averages per launch", "kernels": {}}
    for k, c in sorted(merged.items()):
        rec = {"counters": c}
        gui = c.get("GRBM_GUI_ACTIVE")
        if gui:
            simd_quads = gui / 8.0 * 1024.0 / 4.0
            rec["cycles_per_xcd"] = gui / 8.0
            if "SQ_WAVE_CYCLES" in c:
                rec["waves_per_simd"] = c["SQ_WAVE_CYCLES"] / simd_quads
            if "SQ_ACTIVE_INST_VALU" in c:
                rec["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] / simd_quads
        if c.get("SQ_WAVE_CYCLES"):
            for name, key in (("wait_frac", "SQ_WAIT_ANY"), ("issue_stall_frac", "SQ_WAIT_INST_ANY"), ("active_frac", "SQ_ACTIVE_INST_ANY")):
                if key in c:
                    rec[name] = c[key] / c["SQ_WAVE_CYCLES"]
        if c.get("SQ_LDS_IDX_ACTIVE"):
            rec["lds_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        if c.get("SQ_WAVES") and c.get("SQ_INSTS_VALU"):
            rec["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        out["kernels"][k] = rec
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
