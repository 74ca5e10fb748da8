#!/bin/bash
# Run HERE after tools/collect_all.sh TAG came back through gpurun_out/: copies the summaries to judge into profiles/.
set -e
TAG=${1:-r03}
cd "$(dirname "$0")/.."
python3 tools/make_traffic_json.py gpurun_out/prof_${TAG} profiles/${TAG} 50000000 63 21 14
python3 tools/make_traffic_json.py gpurun_out/prof_${TAG}_k31 profiles/${TAG}_k31 10000000 31 11 11
python3 tools/make_traffic_json.py gpurun_out/prof_${TAG}_k31m15 profiles/${TAG}_k31m15 20000000 31 15 14
{ echo "# rocprofv3 --kernel-trace --pmc <SQ counters> (two passes, tools/collect_sq.sh) of: python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --reads 50000000 (k=63 m=21 b=14)"; cat gpurun_out/sq_${TAG}/summary.txt; } > profiles/${TAG}_sq_counters.txt
{ echo "# the same for: --k 31 --m 11 --b 11 --reads 10000000"; cat gpurun_out/sq_${TAG}_k31/summary.txt; } > profiles/${TAG}_k31_sq_counters.txt
{ echo "# the same for: --k 31 --m 15 --b 14 --reads 20000000 (the reference's default parameters)"; cat gpurun_out/sq_${TAG}_k31m15/summary.txt; } > profiles/${TAG}_k31m15_sq_counters.txt
cp gpurun_out/bench_${TAG}_50M.json profiles/${TAG}_bench_50M.json
cp gpurun_out/bench_${TAG}_k31_10M.json profiles/${TAG}_bench_k31_10M.json
cp gpurun_out/bench_${TAG}_k31m15_20M.json profiles/${TAG}_bench_k31m15_20M.json
cp gpurun_out/bench_${TAG}_50M_get.json profiles/${TAG}_bench_50M_get.json
[ -f gpurun_out/sq_get/summary.txt ] && { echo "# SQ counters of the get path, 10 M reads (insert once, three get_packed calls): tools/sq_get.sh"; cat gpurun_out/sq_get/summary.txt; } > profiles/${TAG}_get_sq_counters.txt
ls -la profiles/ | grep ${TAG}
