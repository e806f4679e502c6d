export SGPU_INDEX_CACHE=/tmp
O=gpurun_out/r05final2; mkdir -p $O
python bench.py > $O/bench_final.json 2> $O/bench_final.err; tail -c 200 $O/bench_final.err
python - <<PY
import json
d=json.load(open('$O/bench_final.json'))
print('value',d['value'],'frac',d['roofline']['frac'],'kernel_ms',d['roofline']['kernel_ms'],'traffic',d['roofline']['traffic'],'lat',d.get('mean_latency_us_single_query'),'resident',d['device_resident']['value'])
for p in d.get('operating_points',[]):
    print(p['target_recall'],p['reached'],p.get('recall_heldout'),p.get('value'),p.get('device_resident_qps'),p.get('roofline_frac'),p.get('entry_point_calls'))
print(d['cpu_baseline']['value'], d['gpu_over_cpu_allcore'], d['timing_s'])
PY
python -m pytest tests/test_gpu_bench_multirank.py -q -m gpu > $O/pytest.txt 2>&1; tail -n 2 $O/pytest.txt
