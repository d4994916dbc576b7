# round 3, session 2: the whole GPU suite on the build with the eight-wave 512-deep tile, then a fuzz slice
mkdir -p gpurun_out/r03x
python -m pytest tests -m gpu -x -q > gpurun_out/r03x/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03x/tests.log
tail -n 5 gpurun_out/r03x/tests.log
python tools/fuzz_parity.py --seconds 150 --seed 271 > gpurun_out/r03x/fuzz271.txt 2>&1; tail -n 1 gpurun_out/r03x/fuzz271.txt
