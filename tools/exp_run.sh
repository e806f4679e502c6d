export SGPU_INDEX_CACHE=/tmp SGPU_TEST_HOOKS=1
O=gpurun_out/r05o; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_coop.py tests/test_gpu_knn.py tests/test_gpu_fixedu8.py -x -q -m gpu > $O/pytest_subset.txt 2>&1; tail -n 3 $O/pytest_subset.txt
B="python bench.py --no-cpu --no-e2e --no-entry --target-recall= --index-cache /tmp"
for i in 1 2; do
$B > $O/bench_dir$i.json 2> $O/bench_dir$i.err; python -c "import json;d=json.load(open('$O/bench_dir$i.json'));print('directory', d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['index']['hbm_bytes'], d.get('mean_latency_us_single_query'), d['timing_s'])"
SGPU_ROW_DIR=0 $B > $O/bench_nodir$i.json 2> $O/bench_nodir$i.err; python -c "import json;d=json.load(open('$O/bench_nodir$i.json'));print('search   ', d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['index']['hbm_bytes'], d.get('mean_latency_us_single_query'), d['timing_s'])"
done
$B --collection clustered > $O/bench_clu_dir.json 2> $O/bench_clu_dir.err; python -c "import json;d=json.load(open('$O/bench_clu_dir.json'));print('clustered directory', d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('mean_latency_us_single_query'))"
SGPU_ROW_DIR=0 $B --collection clustered > $O/bench_clu_nodir.json 2> $O/bench_clu_nodir.err; python -c "import json;d=json.load(open('$O/bench_clu_nodir.json'));print('clustered search   ', d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('mean_latency_us_single_query'))"
python tools/mid_probe.py 64 256 1250 > $O/mid_dir.txt 2>&1; cat $O/mid_dir.txt
SGPU_ROW_DIR=0 python tools/mid_probe.py 64 256 1250 > $O/mid_nodir.txt 2>&1; cat $O/mid_nodir.txt
