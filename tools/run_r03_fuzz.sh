# final build of the round: long randomised parity sweeps (general + K1f's domain)
mkdir -p gpurun_out/r03f
python tools/fuzz_parity.py --seconds ${1:-600} --seed ${2:-321} > gpurun_out/r03f/fuzz${2:-321}.txt 2>&1; tail -n 1 gpurun_out/r03f/fuzz${2:-321}.txt
python tools/fuzz_parity.py --lowdim --seconds ${3:-420} --seed ${4:-331} > gpurun_out/r03f/fuzz${4:-331}_lowdim.txt 2>&1; tail -n 1 gpurun_out/r03f/fuzz${4:-331}_lowdim.txt
