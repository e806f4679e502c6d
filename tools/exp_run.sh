export SGPU_INDEX_CACHE=/tmp
O=gpurun_out/r05u; mkdir -p $O
python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; tail -n 4 $O/gpu_suite.log
rm -rf gpurun_out/r05_prof gpurun_out/r05_traffic_* gpurun_out/r05_single
tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1; tail -n 3 gpurun_out/r05_profile_round.log
C5="--docs 5000000 --dim 200000 --comp-width 4 --k 100 --query-cut 10 --heap-factor 0.9 --n-postings 2000 --centroid-fraction 0.1 --summary-energy 0.4 --max-fraction 4 --min-cluster-size 10 --queries 2000 --steps 5 --warmup 1 --target-recall="
tools/profile_traffic.sh gpurun_out/r05_traffic_c5 $C5 > $O/traffic_c5.txt 2>&1; tail -n 4 $O/traffic_c5.txt
