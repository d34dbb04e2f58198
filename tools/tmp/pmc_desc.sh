#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_desc
rm -rf $out; mkdir -p $out
B="python bench.py --no-cpu --no-reloc --steps 3 --warmup 1 --streams 1"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $out/a -o a --output-format csv -- $B > $out/a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum -d $out/b -o b --output-format csv -- $B > $out/b.log 2>&1 || exit 1
python tools/pmc_summary.py $(find $out/a -name "*counter_collection.csv" | head -1) $out/a_summary.csv
python tools/pmc_summary.py $(find $out/b -name "*counter_collection.csv" | head -1) $out/b_summary.csv
