#!/bin/bash
# SQ counters of the skinny absorption kernel (separate --pmc passes; the program itself follows `--`); only the
# engine's kernels are kept from the counter tables
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_skinny_$i -- python3 $R/tools/skinny_layouts.py > $R/gpurun_out/pmc_skinny_$i.log 2>&1
  echo "pass $i rc=$?"
  f=$(ls /tmp/pmc_skinny_$i/*/*counter_collection.csv | tail -1)
  head -1 $f > $R/gpurun_out/pmc_skinny_$i.csv
  grep "k_mfma_f32" $f >> $R/gpurun_out/pmc_skinny_$i.csv
  rm -rf /tmp/pmc_skinny_$i
done
