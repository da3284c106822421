cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03u
timeout 1500 python bench.py > gpurun_out/r03u/bench.json 2> gpurun_out/r03u/bench.err
timeout 2400 python -m pytest tests -x -q -m gpu --timeout=900 > gpurun_out/r03u/tests.log 2>&1
