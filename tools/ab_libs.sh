#!/bin/bash
export SGPU_TEST_HOOKS=1   # (the SGPU_* knobs these runs set are test hooks)
# A/B of experiment builds of the library on the bench's kernel leg (run on the GPU box through gpurun):
#   tools/ab_libs.sh <out-tag> "<lib suffixes, '' = the product library>" "<value types>" [extra bench args...]
# e.g. tools/ab_libs.sh ab_s4 "default s4" "f16 fixedu8"     (libseismic_hip_s4.so from `make exp NAME=s4 EXTRA=...`)
TAG=$1; LIBS=$2; VTS=$3; shift 3
export SGPU_INDEX_CACHE=/tmp/idx; mkdir -p /tmp/idx gpurun_out/$TAG
F="--no-entry --no-cpu --no-recall --no-latency --no-e2e --no-accounting --target-recall= --steps 20 --warmup 3 $*"
for rep in 1 2; do for vt in $VTS; do for lib in $LIBS; do
  L=seismic_amd/libseismic_hip.so; [ $lib != default ] && L=seismic_amd/libseismic_hip_$lib.so
  SGPU_LIB=$PWD/$L python bench.py $F --value-type $vt 2>gpurun_out/$TAG/err_${lib}_$vt.txt | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $vt rep$rep kernel_ms %.4f ms_per_step %.4f' % (l['roofline']['kernel_ms'], l['ms_per_step']))" | tee -a gpurun_out/$TAG/result.txt
done; done; done
