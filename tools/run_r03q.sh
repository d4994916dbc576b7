mkdir -p gpurun_out/r03q
python -m pytest tests -m gpu -x -q -k "k1024 or error_model or rounding_model or deep_tile or dimensionality_beyond or k512 or bf16" > gpurun_out/r03q/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03q/tests.log
tail -n 4 gpurun_out/r03q/tests.log
python tools/probe_depths.py --deep 2>&1 | grep -E "points|exact" > gpurun_out/r03q/deep.txt
cat gpurun_out/r03q/deep.txt
python tools/fuzz_parity.py --seconds 200 --seed 231 > gpurun_out/r03q/fuzz231.txt 2>&1; tail -n 2 gpurun_out/r03q/fuzz231.txt
