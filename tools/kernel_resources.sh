#!/bin/bash
# Register / spill / scratch figures of every search kernel variant (code object notes).
# Usage: tools/kernel_resources.sh > profiles/<round>_kernel_resources.txt
cd "$(dirname "$0")/../seismic_amd/csrc" || exit 1
TMP=$(mktemp -d)
for f in sk_*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --offload-device-only --no-gpu-bundle-output -c "$f" -o "$TMP/${f%.hip}.co" -Wno-unknown-pragmas -Wno-unused-parameter &
done
wait
printf "%-62s %5s %5s %6s %6s %8s\n" "kernel<component, threads, heap regs, lookup, counted, value type>" vgpr sgpr vspill sspill scratch
for f in "$TMP"/*.co; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" | awk '
    /\.name:/ {name=$2}
    /\.private_segment_fixed_size:/ {scr=$2}
    /\.sgpr_count:/ {sg=$2}
    /\.sgpr_spill_count:/ {ss=$2}
    /\.vgpr_count:/ {vg=$2}
    /\.vgpr_spill_count:/ {vs=$2; printf "%s %s %s %s %s %s\n", name, vg, sg, vs, ss, scr}'
done | while read n vg sg vs ss scr; do
  printf "%-62s %5s %5s %6s %6s %8s\n" "$(echo "$n" | c++filt | sed 's/void sgpu::seismic_search_kernel//; s/(sgpu::DevView.*//; s/unsigned short/u16/; s/unsigned int/u32/')" "$vg" "$sg" "$vs" "$ss" "$scr"
done | sort
rm -rf "$TMP"
