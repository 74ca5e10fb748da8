#!/bin/bash
# Run ON THE GPU BOX (through gpurun): everything profiles/ holds for one round -- kernel stats, PMC traffic and SQ counters
# for the headline config (#3) and for config #2' (k31 m11 b11, 10 M reads), and the bench lines of both.
# usage: tools/collect_all.sh TAG      then, here:  tools/publish_profiles.sh TAG
set -e
TAG=${1:-r03}
cd /root/repo
bash tools/collect_profiles.sh ${TAG}
bash tools/collect_profiles.sh ${TAG}_k31 --k 31 --m 11 --b 11 --reads 10000000
bash tools/collect_profiles.sh ${TAG}_k31m15 --k 31 --m 15 --b 14 --reads 20000000
bash tools/collect_sq.sh ${TAG} --reads 50000000
bash tools/collect_sq.sh ${TAG}_k31 --k 31 --m 11 --b 11 --reads 10000000
bash tools/collect_sq.sh ${TAG}_k31m15 --k 31 --m 15 --b 14 --reads 20000000
python3 bench.py --steps 10 --warmup 3 > gpurun_out/bench_${TAG}_50M.json 2> gpurun_out/bench_${TAG}_50M.err
python3 bench.py --k 31 --m 11 --b 11 --reads 10000000 --steps 10 --warmup 3 --e2e-reads 0 > gpurun_out/bench_${TAG}_k31_10M.json 2> gpurun_out/bench_${TAG}_k31_10M.err
python3 bench.py --k 31 --m 15 --b 14 --reads 20000000 --steps 10 --warmup 3 --e2e-reads 0 > gpurun_out/bench_${TAG}_k31m15_20M.json 2> gpurun_out/bench_${TAG}_k31m15_20M.err
python3 bench.py --steps 5 --warmup 2 --get --no-cpu-baseline --e2e-reads 0 > gpurun_out/bench_${TAG}_50M_get.json 2> gpurun_out/bench_${TAG}_50M_get.err
bash tools/sq_get.sh > gpurun_out/sq_get_${TAG}.log 2>&1 || true
echo done
