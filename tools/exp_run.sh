export SGPU_INDEX_CACHE=/tmp SGPU_TEST_HOOKS=1
O=gpurun_out/r05q; mkdir -p $O
T1=$PWD/seismic_amd/libseismic_hip_t1.so
SGPU_LIB=$T1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_coop.py tests/test_gpu_api_and_scale.py -x -q -m gpu > $O/pytest_subset.txt 2>&1; tail -n 3 $O/pytest_subset.txt
B="python bench.py --no-cpu --no-e2e --no-entry --no-latency --target-recall= --index-cache /tmp"
for i in 1 2 3; do
SGPU_LIB=$T1 $B > $O/bench_t1_$i.json 2> $O/bench_t1_$i.err; python -c "import json;d=json.load(open('$O/bench_t1_$i.json'));print('t1     ', d['roofline']['kernel_ms'], d['roofline']['frac'])"
$B > $O/bench_base_$i.json 2> $O/bench_base_$i.err; python -c "import json;d=json.load(open('$O/bench_base_$i.json'));print('base   ', d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
SGPU_LIB=$T1 $B --collection clustered > $O/bench_clu_t1.json 2> $O/bench_clu_t1.err; python -c "import json;d=json.load(open('$O/bench_clu_t1.json'));print('clustered t1  ', d['roofline']['kernel_ms'], d['roofline']['frac'])"
$B --collection clustered > $O/bench_clu_base.json 2> $O/bench_clu_base.err; python -c "import json;d=json.load(open('$O/bench_clu_base.json'));print('clustered base', d['roofline']['kernel_ms'], d['roofline']['frac'])"
SGPU_LIB=$T1 $B --value-type fixedu8 > $O/bench_u8_t1.json 2> $O/bench_u8_t1.err; python -c "import json;d=json.load(open('$O/bench_u8_t1.json'));print('u8 t1  ', d['roofline']['kernel_ms'], d['roofline']['frac'])"
$B --value-type fixedu8 > $O/bench_u8_base.json 2> $O/bench_u8_base.err; python -c "import json;d=json.load(open('$O/bench_u8_base.json'));print('u8 base', d['roofline']['kernel_ms'], d['roofline']['frac'])"
