#!/bin/bash
# usage: tools/pmc_one.sh <tag> <one_conv args...>   (three counter passes; output under gpurun_out/pmc_<tag>_*)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $R/gpurun_out/pmc_${tag}_a -- python3 $R/tools/one_conv.py "$@" > /dev/null 2>&1 || exit 1
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_${tag}_b -- python3 $R/tools/one_conv.py "$@" > /dev/null 2>&1 || exit 1
rocprofv3 --output-format csv --pmc TA_BUSY TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TCP_TA_DATA_STALL_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_${tag}_c -- python3 $R/tools/one_conv.py "$@" > /dev/null 2>&1 || exit 1
python3 $R/tools/pmc_table.py $R/gpurun_out/pmc_${tag}_a $R/gpurun_out/pmc_${tag}_b $R/gpurun_out/pmc_${tag}_c
