export SGPU_INDEX_CACHE=/tmp
O=gpurun_out/r05l; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; tail -n 5 $O/gpu_suite.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; python - <<PY
import json
d=json.load(open('$O/bench.json'))
print('value',d['value'],'frac',d['roofline']['frac'],'kernel_ms',d['roofline']['kernel_ms'],'lat',d.get('mean_latency_us_single_query'),d.get('latency_percentiles_us_single_query'))
print('recall',d.get('recall_at_k'),d.get('recall_heldout'))
for p in d.get('operating_points',[]):
    print(p['target_recall'],p['reached'],p.get('recall_selection_sample'),p.get('recall_heldout'),p.get('value'),p.get('query_cut'),p.get('latency_percentiles_us'))
print(d['timing_s'], d.get('operating_points_error'))
PY
SGPU_TEST_HOOKS=1 SGPU_LIB=$PWD/seismic_amd/libseismic_hip_prof.so python tools/phase_profile.py --docs 8800000 --n-postings 2000 --queries 10000 --collection 1 --no-save > $O/phase_clustered.txt 2>&1; tail -n 22 $O/phase_clustered.txt
SGPU_TEST_HOOKS=1 SGPU_LIB=$PWD/seismic_amd/libseismic_hip_prof.so python tools/phase_profile.py --docs 8800000 --n-postings 2000 --queries 10000 --collection 0 --no-save > $O/phase_survey.txt 2>&1; tail -n 22 $O/phase_survey.txt
