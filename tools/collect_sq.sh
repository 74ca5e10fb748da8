#!/bin/bash
# Run ON THE GPU BOX (through gpurun): SQ issue/stall counters per kernel for a reduced bench (PMC only, no traces).
# usage: tools/collect_sq.sh TAG [bench args]   -> gpurun_out/sq_TAG/{a,b}/... + gpurun_out/sq_TAG/summary.txt
set -e
TAG=${1:-r01}; shift || true
ARGS=${@:---reads 10000000}
OUT=/root/repo/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
  --output-format csv -d $OUT/a -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --e2e-reads 0 $ARGS > /dev/null 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
  --output-format csv -d $OUT/b -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --e2e-reads 0 $ARGS > /dev/null 2> $OUT/b.err
python3 /root/repo/tools/sum_pmc.py $OUT/a $OUT/b > $OUT/summary.txt
echo collected $OUT
