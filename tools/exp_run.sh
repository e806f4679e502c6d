export SGPU_INDEX_CACHE=/tmp
O=gpurun_out/r05t; mkdir -p $O
python -m pytest tests/test_gpu_fixedu8.py tests/test_gpu_fuzz.py tests/test_gpu_coop.py tests/test_gpu_knn.py tests/test_gpu_lookup_layouts.py -x -q -m gpu > $O/pytest_subset.txt 2>&1; tail -n 3 $O/pytest_subset.txt
B="python bench.py --no-cpu --no-e2e --no-entry --no-latency --target-recall= --index-cache /tmp"
for i in 1 2; do
$B --value-type dotvbyte > $O/dvb$i.json 2> /dev/null; python -c "import json;d=json.load(open('$O/dvb$i.json'));print('dvb', d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['index']['hbm_bytes'])"
$B --value-type fixedu8 > $O/u8$i.json 2> /dev/null; python -c "import json;d=json.load(open('$O/u8$i.json'));print('u8 ', d['roofline']['kernel_ms'], d['roofline']['frac'])"
$B > $O/f16$i.json 2> /dev/null; python -c "import json;d=json.load(open('$O/f16$i.json'));print('f16', d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
