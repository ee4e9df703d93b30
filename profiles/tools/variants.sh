for v in "" "MAREX_THR_TILE=16" "MAREX_THR_TILE=16 MAREX_THR_DD=61" "MAREX_THR_TILE=16 MAREX_THR_DD=32" "MAREX_THR_DD=61" "MAREX_THR_DD=366" "MAREX_THR_TALL=0"; do
  env $v python bench.py --workload cfg3band --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/v.json 2> gpurun_out/v.err
  python -c "
import json
d=json.load(open('gpurun_out/v.json'))
print('$v', round(d['ms_per_step'],2), {k:round(v,3) for k,v in d['kernel_ms'].items()})
"
done
