#!/bin/bash
# usage: sq_cmd.sh <tag> <python script and args...>  -- two SQ counter passes of an arbitrary python3 command, summarised like sq_passes.sh
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d gpurun_out/${tag}_sq1 -- python3 "$@" > gpurun_out/${tag}_sq1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_sq2 -- python3 "$@" > gpurun_out/${tag}_sq2.log 2>&1 &&
python3 profiles/collect_sq.py "$tag" gpurun_out/${tag}_sq1 gpurun_out/${tag}_sq2 > gpurun_out/${tag}_sq.json
