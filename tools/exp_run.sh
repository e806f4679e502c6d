export SGPU_INDEX_CACHE=/tmp
O=gpurun_out/r05m; mkdir -p $O
python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; tail -n 4 $O/gpu_suite.log
python bench.py --collection clustered --target-recall= > $O/bench_clustered.json 2> $O/bench_clustered.err; tail -c 300 $O/bench_clustered.err
python -c "import json;d=json.load(open('$O/bench_clustered.json'));r=d['roofline'];print('clustered', d['value'], r['kernel_ms'], r['frac'], r['bytes_per_query'], d.get('recall_at_k'), d.get('mean_latency_us_single_query'), d['cpu_baseline']['value'], d['cpu_baseline']['single_thread_us_per_query'])"
# footprint option: document-major forward store only (headline index and the 0.99-recall index)
SGPU_FWD_LAYOUT=doc tools/profile_traffic.sh gpurun_out/r05m/traffic_doc_headline > $O/traffic_doc_headline.txt 2>&1; tail -n 12 $O/traffic_doc_headline.txt
SGPU_FWD_LAYOUT=doc tools/profile_traffic.sh gpurun_out/r05m/traffic_doc_r99 --n-postings 6000 --max-fraction 4 --query-cut 12 > $O/traffic_doc_r99.txt 2>&1; tail -n 12 $O/traffic_doc_r99.txt
