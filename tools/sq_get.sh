#!/bin/bash
# SQ counters of the get path (10 M reads): usage sq_get.sh
set -e
OUT=/root/repo/gpurun_out/sq_get
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
  --output-format csv -d $OUT/a -- python3 /root/repo/tools/get_bench.py 63 21 14 10000000 > /dev/null 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
  --output-format csv -d $OUT/b -- python3 /root/repo/tools/get_bench.py 63 21 14 10000000 > /dev/null 2> $OUT/b.err
python3 /root/repo/tools/sum_pmc.py $OUT/a $OUT/b > $OUT/summary.txt
cat $OUT/summary.txt
