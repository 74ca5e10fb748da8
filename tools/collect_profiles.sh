#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats and HBM traffic counters for the default bench.
# PMC passes are separate from the trace pass, and FETCH_SIZE / WRITE_SIZE are separate passes (TCC slots).
# usage: tools/collect_profiles.sh TAG     -> gpurun_out/prof_TAG/{stats,fetch,write}/...
set -e
TAG=${1:-r01}
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/write.err
echo collected $OUT
