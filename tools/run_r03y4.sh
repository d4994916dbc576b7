mkdir -p gpurun_out/r03y
python -m pytest tests -m gpu -x -q -k "k1f or c2 or low_dim or golden or k1a" > gpurun_out/r03y/tests4.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03y/tests4.log
tail -n 4 gpurun_out/r03y/tests4.log
python tools/fuzz_parity.py --lowdim --seconds 240 --seed 311 > gpurun_out/r03y/fuzz311_lowdim.txt 2>&1; tail -n 2 gpurun_out/r03y/fuzz311_lowdim.txt
