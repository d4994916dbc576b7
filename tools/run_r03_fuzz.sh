# final build of the round: long randomised parity sweeps (general + K1f's domain)
mkdir -p gpurun_out/r03f
python tools/fuzz_parity.py --seconds 600 --seed 321 > gpurun_out/r03f/fuzz321.txt 2>&1; tail -n 1 gpurun_out/r03f/fuzz321.txt
python tools/fuzz_parity.py --lowdim --seconds 420 --seed 331 > gpurun_out/r03f/fuzz331_lowdim.txt 2>&1; tail -n 1 gpurun_out/r03f/fuzz331_lowdim.txt
