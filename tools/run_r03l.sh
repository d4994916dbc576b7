mkdir -p gpurun_out/r03l
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r03l/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03l/tests.log
tail -4 gpurun_out/r03l/tests.log
python bench.py > gpurun_out/r03l/bench_default.json 2> gpurun_out/r03l/bench_default.err; echo "bench rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03l/stats_default -- python3 bench.py > gpurun_out/r03l/stats_default.log 2>&1
cp gpurun_out/r03l/stats_default/*/*kernel_stats.csv gpurun_out/r03l/kernel_stats_default.csv 2>/dev/null
python tools/probe_depths.py 2>&1 | grep points > gpurun_out/r03l/depths.txt
./nns-cuda_amd/nns_driver --repeat 3 > gpurun_out/r03l/driver.txt 2>&1
cut -c1-300 gpurun_out/r03l/bench_default.json; cat gpurun_out/r03l/depths.txt
