mkdir -p gpurun_out/r03p
python -m pytest tests -m gpu -x -q > gpurun_out/r03p/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03p/tests.log
tail -n 4 gpurun_out/r03p/tests.log
python tools/fuzz_parity.py --seconds 400 --seed 221 > gpurun_out/r03p/fuzz221.txt 2>&1
tail -n 2 gpurun_out/r03p/fuzz221.txt
python tools/probe_streams.py 2>&1 | grep "k=" > gpurun_out/r03p/streams.txt
python tools/probe_crossover.py 2>&1 | grep "k=" > gpurun_out/r03p/crossover.txt
python tools/probe_shapes.py 2>&1 | grep "k=" > gpurun_out/r03p/shapes.txt
cat gpurun_out/r03p/streams.txt gpurun_out/r03p/crossover.txt gpurun_out/r03p/shapes.txt
