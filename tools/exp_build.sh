#!/bin/bash
# Experiment builds of the product library: only the named kernel families are recompiled (with extra flags and / or a
# patched copy of search_kernel.inc); every other object is taken from the default build. The result is
# seismic_amd/libseismic_hip_<name>.so (run with SGPU_LIB=<path>). Nothing of this ships in the default library.
#   tools/exp_build.sh <name> "<extra hipcc flags>" [patched search_kernel.inc] [unit ...]   (default unit: sk_u16_dense)
set -e
NAME=$1; EXTRA=$2; INC=$3; shift 3 || true
UNITS=${@:-sk_u16_dense}
C=$(cd "$(dirname "$0")/../seismic_amd/csrc" && pwd)
B=$C/build_$NAME
make -s -C "$C" -j8 >/dev/null
mkdir -p "$B/src"
cp "$C"/build/*.o "$B"/
cp "${INC:-$C/search_kernel.inc}" "$B/src/search_kernel.inc"
for u in $UNITS; do
  cp "$C/$u.hip" "$B/src/"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-parameter -Wno-unknown-pragmas \
    $EXTRA -I"$C" -I"$C/../../include" -c "$B/src/$u.hip" -o "$B/$u.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$C/../libseismic_hip_$NAME.so" "$B"/*.o -L/usr/lib/gcc/x86_64-linux-gnu/11 -lgomp
ls -la "$C/../libseismic_hip_$NAME.so"
