#!/bin/bash
# usage: variants.sh <workload> "ENV=.. ENV=.." "..."  -- one bench line per option set (kernel ms only)
w=$1; shift
for v in "$@"; do
  env $v python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/v.json 2> gpurun_out/v.err
  python -c "
import json
d=json.load(open('gpurun_out/v.json'))
print('$v', round(d['ms_per_step'],2), {k:round(v,3) for k,v in d['kernel_ms'].items()})
"
done
