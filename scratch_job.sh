cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03w
timeout 1500 python -m pytest tests/test_gibbs_gpu.py tests/test_gibbs_long_gpu.py tests/test_cli_gpu.py tests/test_comm_gpu.py -x -q -m gpu --timeout=600 -k "noise or C5 or thirty or wide or ranks" > gpurun_out/r03w/tests.log 2>&1
timeout 600 python tools/perf_noise_classes.py 10 100000 > gpurun_out/r03w/perf10.log 2>&1
timeout 600 python tools/perf_noise_classes.py 30 2000 > gpurun_out/r03w/perf30.log 2>&1
