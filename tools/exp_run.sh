export SGPU_INDEX_CACHE=/tmp SGPU_TEST_HOOKS=1
O=gpurun_out/r05nd3; mkdir -p $O
for n in base nd3a nd3b base2 nd3a2; do
  case $n in base|base2) L=seismic_amd/libseismic_hip.so;; nd3a|nd3a2) L=seismic_amd/libseismic_hip_nd3a.so;; nd3b) L=seismic_amd/libseismic_hip_nd3b.so;; esac
  SGPU_LIB=$PWD/$L timeout 400 python tools/latency_probe.py 8800000 > $O/$n.txt 2>&1
  echo "$n: $(grep -E 'nq=   1:|nq=  64:|nq= 256:' $O/$n.txt | sed 's/ us wall per pass,//; s/us\/query//' | tr '\n' ' ') $(grep -o 'sgpu_search_sequential: [0-9.]* us' $O/$n.txt)"
done
