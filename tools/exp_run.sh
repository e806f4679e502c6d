export SGPU_INDEX_CACHE=/tmp SGPU_TEST_HOOKS=1
O=gpurun_out/r05x; mkdir -p $O
python tools/shard_probe.py --threads 1,2,3 > $O/shard_probe.txt 2>&1; cat $O/shard_probe.txt | tail -n 14
