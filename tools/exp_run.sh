export SGPU_INDEX_CACHE=/tmp SGPU_TEST_HOOKS=1
O=gpurun_out/r05h; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fixedu8.py tests/test_gpu_knn.py tests/test_gpu_lookup_layouts.py -x -q -m gpu > $O/pytest_subset.txt 2>&1; tail -n 3 $O/pytest_subset.txt
B="python bench.py --no-cpu --no-latency --no-e2e --no-entry --target-recall= --index-cache /tmp"
$B > $O/bench_sliced.json 2> $O/bench_sliced.err; python -c "import json;d=json.load(open('$O/bench_sliced.json'));print('sliced', d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['index']['hbm_bytes'], d.get('recall_at_k'))"
SGPU_FWD_STREAM=plain $B > $O/bench_plain.json 2> $O/bench_plain.err; python -c "import json;d=json.load(open('$O/bench_plain.json'));print('plain', d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['index']['hbm_bytes'], d.get('recall_at_k'))"
$B > $O/bench_sliced2.json 2> $O/bench_sliced2.err; python -c "import json;d=json.load(open('$O/bench_sliced2.json'));print('sliced', d['roofline']['kernel_ms'], d['roofline']['frac'])"
SGPU_FWD_STREAM=plain $B > $O/bench_plain2.json 2> $O/bench_plain2.err; python -c "import json;d=json.load(open('$O/bench_plain2.json'));print('plain', d['roofline']['kernel_ms'], d['roofline']['frac'])"
run() { name=$1; sizes=$2; shift 2; env "$@" python tools/mid_probe.py $sizes > $O/$name.txt 2>&1; echo "== $name"; cat $O/$name.txt | tail -n 3; }
S="64 256"
run s_default "$S" A=1
run s_reach128 "$S" SGPU_COOP_FIRST_REACH=128
run s_reach1024 "$S" SGPU_COOP_FIRST_REACH=1024
run s_init64 "$S" SGPU_COOP_ITEMS_INIT=64
run s_init256 "$S" SGPU_COOP_ITEMS_INIT=256
run s_chunk64 "$S" SGPU_COOP_CHUNK=64
run s_chunk256 "$S" SGPU_COOP_CHUNK=256
run s_minitems0 "$S" SGPU_COOP_MIN_ITEMS=0
run s_idlemin1 "$S" SGPU_COOP_IDLE_MIN=1
