export SGPU_INDEX_CACHE=/tmp
tools/profile_traffic.sh gpurun_out/r05_traffic_r90 --n-postings 4000 --max-fraction 3 --query-cut 6 > gpurun_out/r05_tr90.txt 2>&1; tail -n 9 gpurun_out/r05_tr90.txt
tools/profile_traffic.sh gpurun_out/r05_traffic_r95 --n-postings 3000 --max-fraction 4 --query-cut 11 > gpurun_out/r05_tr95.txt 2>&1; tail -n 9 gpurun_out/r05_tr95.txt
tools/profile_traffic.sh gpurun_out/r05_traffic_r99 --n-postings 6000 --max-fraction 4 --query-cut 13 > gpurun_out/r05_tr99.txt 2>&1; tail -n 9 gpurun_out/r05_tr99.txt
