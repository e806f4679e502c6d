#!/bin/bash
# A/B of small-launch latency between builds of the library on ONE box (boxes differ by +-2 %): runs
# tools/latency_probe.py once per named build, in the order given (name a build twice to see the run-to-run spread),
# and prints one line per run: kernel time of 1 / 8 / 64 / 256-query launches and the sequential single-query loop.
#   tools/latency_ab.sh <out dir> <build> [<build> ...]      build = "base" (the product library) or the <name> of
#   seismic_amd/libseismic_hip_<name>.so (tools/exp_build.sh <name> ...).   r05: profiles/r05_coop_in_flight.txt
set -u
O=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$O"
export SGPU_INDEX_CACHE=${SGPU_INDEX_CACHE:-/tmp} SGPU_TEST_HOOKS=1
i=0
for n in "$@"; do
  i=$((i + 1))
  if [ "$n" = base ]; then L=$R/seismic_amd/libseismic_hip.so; else L=$R/seismic_amd/libseismic_hip_$n.so; fi
  SGPU_LIB=$L timeout 400 python "$R/tools/latency_probe.py" 8800000 > "$O/$i.$n.txt" 2>&1
  echo "$n: $(grep -E 'nq=   1:|nq=   8:|nq=  64:|nq= 256:' "$O/$i.$n.txt" | sed 's/ us wall per pass,//; s/us\/query//' | tr '\n' ' ') $(grep -o 'sgpu_search_sequential: [0-9.]* us' "$O/$i.$n.txt")"
done
