mkdir -p gpurun_out/r5i; O=gpurun_out/r5i
T='tests/test_gpu_e2e.py -m gpu -q --tb=line -s -k "multi_condition_router or gamma_outliers or two_blocks"'
run() { echo "== $1"; env $1 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=line -s -k "multi_condition_router or gamma_outliers or two_blocks" 2>&1 | grep -E "config-5 chain|gamma_outliers|two_blocks|passed|failed" | cut -c1-260; }
(run "X=1"; run "CTRL_SMALLCONV_MFMA=0"; run "CTRL_SMALL_TILES=0"; run "CTRL_QKV_ONE=0"; run "CTRL_GROUP=0"; run "CTRL_ADAPTER_TOK_F16=0"; run "CTRL_SMALLCONV_MFMA=0 CTRL_SMALL_TILES=0"; run "CTRL_IGEMM8=0") > $O/sweep.txt 2>&1
cat $O/sweep.txt
