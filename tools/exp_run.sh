export SGPU_INDEX_CACHE=/tmp SGPU_TEST_HOOKS=1
O=gpurun_out/r05j; mkdir -p $O
python tools/clustered_tune.py 8800000 "96,0.8,0.25,0.3" "256,0.8,0.3,0.4" "512,0.7,0.35,0.5" "256,0.6,0.4,0.6" > $O/clustered_tune.txt 2>&1; cat $O/clustered_tune.txt
X=gpurun_out/r05j/lds
export PMC_SETS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS;SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES"
tools/profile_bytes.sh $X u8_default --index-cache /tmp --value-type fixedu8
SGPU_LIB=$PWD/seismic_amd/libseismic_hip_ldsctl.so tools/profile_bytes.sh $X u8_control --index-cache /tmp --value-type fixedu8
SGPU_LIB=$PWD/seismic_amd/libseismic_hip_ldsnc.so tools/profile_bytes.sh $X u8_noconflict --index-cache /tmp --value-type fixedu8
cat $X/results.jsonl
