#!/bin/bash
# usage: sq_passes.sh <tag> <bench args...>   -- three SQ/GRBM counter passes of bench.py, summarised by profiles/collect_sq.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d gpurun_out/${tag}_sq1 -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_sq1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_sq2 -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_sq2.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH --output-format csv -d gpurun_out/${tag}_sq3 -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_sq3.log 2>&1 &&
python3 profiles/collect_sq.py "$tag" gpurun_out/${tag}_sq1 gpurun_out/${tag}_sq2 gpurun_out/${tag}_sq3 > gpurun_out/${tag}_sq.json
