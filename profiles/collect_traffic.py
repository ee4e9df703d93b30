#!/usr/bin/env python3
"""Turn rocprofv3 PMC csv output (separate FETCH_SIZE and WRITE_SIZE passes of `bench.py`) into the per-kernel
HBM traffic file that bench.py reads for its `roofline.traffic` field.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python3 profiles/collect_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w cfg2 > profiles/r01_cfg2_traffic.json

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of coalesced streaming reads (verified here on k_mask_ge and k_transpose, whose reads
are known exactly), so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE matched known store volumes 1:1.
"""
import collections
import csv
import glob
import os
import json
import sys


def per_kernel(dirname, counter):
    f = sorted(glob.glob(f"{dirname}/*/*_counter_collection.csv"), key=os.path.getmtime)[-1]  # the latest run in that directory
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"workload": sys.argv[3], "unit": "bytes per launch", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    rd = 2.0 * fetch.get(k, 0.0) * 1024.0
    wr = write.get(k, 0.0) * 1024.0
    out["kernels"][k] = {"fetch_size_kib": fetch.get(k), "write_size_kib": write.get(k),
                         "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr}
json.dump(out, sys.stdout, indent=1)
