#!/bin/bash
# One round's profiling evidence in one go (run on the GPU box through gpurun): the headline workload with the full
# counter set (tools/profile_bench.sh), then HBM traffic (kernel trace + FETCH_SIZE / WRITE_SIZE passes,
# tools/profile_traffic.sh) of the other workloads bench.py quotes - the three fixed-recall operating points, the
# fixed-u8 and DotVByte forward indexes, first_sorted - and the kernel statistics of single-query (cooperative) launches.
# Usage: tools/profile_round.sh <tag, e.g. r04>     -> gpurun_out/<tag>_prof*, then tools/record_profile.py on each
set -u
TAG="$1"
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"
tools/profile_bench.sh gpurun_out/${TAG}_prof
tools/profile_traffic.sh gpurun_out/${TAG}_traffic_r90 --n-postings 4000 --max-fraction 3 --query-cut 6
tools/profile_traffic.sh gpurun_out/${TAG}_traffic_r95 --n-postings 3000 --max-fraction 4 --query-cut 11
tools/profile_traffic.sh gpurun_out/${TAG}_traffic_r99 --n-postings 6000 --max-fraction 4 --query-cut 13
tools/profile_traffic.sh gpurun_out/${TAG}_traffic_fixedu8 --value-type fixedu8
tools/profile_traffic.sh gpurun_out/${TAG}_traffic_dotvbyte --value-type dotvbyte
tools/profile_traffic.sh gpurun_out/${TAG}_traffic_first_sorted --first-sorted 1
# single-query launches (the cooperative variant): kernel statistics of the reference's sequential loop
mkdir -p gpurun_out/${TAG}_single
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/${TAG}_single/trace" -o trace -- \
   python "$REPO/tools/latency_probe.py" 8800000 > "$REPO/gpurun_out/${TAG}_single/latency_probe.txt" 2> "$REPO/gpurun_out/${TAG}_single/err.txt")
find gpurun_out/${TAG}_single -name "*.db" -delete; find gpurun_out/${TAG}_single -type f -size +400k -delete
ls gpurun_out/${TAG}_single/trace 2>/dev/null | head
du -sh gpurun_out
