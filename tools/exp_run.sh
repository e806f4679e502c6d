export SGPU_INDEX_CACHE=/tmp
O=gpurun_out/r05z4; mkdir -p $O
B="python bench.py --no-cpu --no-latency --no-recall --no-e2e --target-recall= --index-cache /tmp"
for w in 3 10 3 10; do
$B --warmup $w > $O/v_w$w.json 2> /dev/null
python -c "import json;d=json.load(open('$O/v_w$w.json'));print('warmup $w value',round(d['value']),'resident',round(d['device_resident']['value']),'ms_per_step',d['ms_per_step'])"
done
SGPU_CHUNK_MAX=2 $B --warmup 3 > $O/v_cm2.json 2> /dev/null; python -c "import json;d=json.load(open('$O/v_cm2.json'));print('CHUNK_MAX=2 warmup 3 value',round(d['value']))"
SGPU_CHUNK_MAX=4 $B --warmup 3 > $O/v_cm4.json 2> /dev/null; python -c "import json;d=json.load(open('$O/v_cm4.json'));print('CHUNK_MAX=4 warmup 3 value',round(d['value']))"
