#!/bin/bash
# Round 6's closing evidence at the final library (run on the GPU box through gpurun): HBM traffic (kernel trace + FETCH_SIZE /
# WRITE_SIZE passes) of the workloads bench.py quotes besides the headline and the clustered collection (those:
# tools/profile_bench.sh), kernel statistics of single-query and 1250-query launches, and the bench lines themselves.
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"
export SGPU_INDEX_CACHE=/tmp
T=gpurun_out/r06
C5="--docs 5000000 --dim 200000 --comp-width 4 --k 100 --query-cut 10 --heap-factor 0.9 --n-postings 2000 --centroid-fraction 0.1 --summary-energy 0.4 --max-fraction 4 --min-cluster-size 10 --queries 2000 --steps 5 --warmup 1"
tools/profile_traffic.sh ${T}_traffic_r90 --n-postings 4000 --max-fraction 3 --query-cut 6 > /dev/null
tools/profile_traffic.sh ${T}_traffic_r95 --n-postings 3000 --max-fraction 4 --query-cut 11 > /dev/null
tools/profile_traffic.sh ${T}_traffic_r99 --n-postings 6000 --max-fraction 4 --query-cut 13 > /dev/null
tools/profile_traffic.sh ${T}_traffic_fixedu8 --value-type fixedu8 > /dev/null
tools/profile_traffic.sh ${T}_traffic_dotvbyte --value-type dotvbyte > /dev/null
tools/profile_traffic.sh ${T}_traffic_first_sorted --first-sorted 1 > /dev/null
tools/profile_traffic.sh ${T}_traffic_c5 $C5 > /dev/null
# single-query (cooperative) and 1250-query (streamed) launches: kernel statistics
mkdir -p ${T}_single
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/${T}_single/trace" -o trace -- \
   python "$REPO/tools/latency_probe.py" 8800000 > "$REPO/${T}_single/latency_probe.txt" 2> "$REPO/${T}_single/err.txt")
mkdir -p ${T}_1250
(cd /tmp && TMPDIR=/tmp LATENCY_PROBE_SIZES=1250 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/${T}_1250/trace" -o trace -- \
   python "$REPO/tools/latency_probe.py" 8800000 > "$REPO/${T}_1250/latency_probe.txt" 2> "$REPO/${T}_1250/err.txt")
find ${T}_single ${T}_1250 -name "*.db" -delete; find ${T}_single ${T}_1250 -type f -size +400k -delete
python tools/latency_probe.py 8800000 > gpurun_out/r06_latency_probe.txt 2>&1
python tools/shard_probe.py > gpurun_out/r06_shard_probe.txt 2>&1
# the bench lines (the traffic entries above are not in pmc_traffic.json yet when these run: their lines are re-read by
# tools/record_profile.py on the build host and say so)
python bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err
python bench.py --collection clustered > gpurun_out/r06_bench_clustered.json 2> gpurun_out/r06_bench_clustered.err
python bench.py --value-type fixedu8 --no-cpu --target-recall= > gpurun_out/r06_bench_fixedu8.json 2>/dev/null
python bench.py --value-type dotvbyte --no-cpu --target-recall= > gpurun_out/r06_bench_dotvbyte.json 2>/dev/null
python bench.py --first-sorted 1 --no-cpu --target-recall= > gpurun_out/r06_bench_first_sorted.json 2>/dev/null
python bench.py $C5 --target-recall= > gpurun_out/r06_bench_c5_5M_200K.json 2>/dev/null
du -sh gpurun_out
