export SGPU_INDEX_CACHE=/tmp
O=gpurun_out/r05w; mkdir -p $O
ND5=$PWD/seismic_amd/libseismic_hip_nd5.so
B="python bench.py --no-cpu --no-e2e --no-entry --no-latency --target-recall= --index-cache /tmp"
for i in 1 2; do
SGPU_LIB=$ND5 $B --value-type fixedu8 > $O/u8_nd5_$i.json 2> /dev/null; python -c "import json;d=json.load(open('$O/u8_nd5_$i.json'));print('u8 nd5 ', d['roofline']['kernel_ms'])"
$B --value-type fixedu8 > $O/u8_base_$i.json 2> /dev/null; python -c "import json;d=json.load(open('$O/u8_base_$i.json'));print('u8 base', d['roofline']['kernel_ms'])"
SGPU_LIB=$ND5 $B --value-type dotvbyte > $O/dvb_nd5_$i.json 2> /dev/null; python -c "import json;d=json.load(open('$O/dvb_nd5_$i.json'));print('dvb nd5 ', d['roofline']['kernel_ms'])"
$B --value-type dotvbyte > $O/dvb_base_$i.json 2> /dev/null; python -c "import json;d=json.load(open('$O/dvb_base_$i.json'));print('dvb base', d['roofline']['kernel_ms'])"
done
