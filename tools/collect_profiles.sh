#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats and HBM traffic counters for a bench.py workload.
# PMC passes are separate from the trace pass, and FETCH_SIZE / WRITE_SIZE are separate passes (TCC slots).
# usage: tools/collect_profiles.sh TAG [bench args]   -> gpurun_out/prof_TAG/{stats,fetch,write}/..., src.sha256, args.txt
set -e
TAG=${1:-r02}; shift || true
ARGS="$@"
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 /root/repo/tools/src_hash.py > $OUT/src.sha256   # the kernel sources these counters belong to
echo "$ARGS" > $OUT/args.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --e2e-reads 0 $ARGS > $OUT/bench_under_trace.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --e2e-reads 0 $ARGS > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --e2e-reads 0 $ARGS > /dev/null 2> $OUT/write.err
echo collected $OUT
