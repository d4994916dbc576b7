mkdir -p gpurun_out/r03s
python -m pytest tests -m gpu -x -q -k "k512 or error_model or rounding_model or deep_tile or bf16 or k1024" > gpurun_out/r03s/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03s/tests.log
tail -n 4 gpurun_out/r03s/tests.log
python tools/probe_depths.py --k512 2>&1 | grep -E "points" > gpurun_out/r03s/k512.txt
cat gpurun_out/r03s/k512.txt
python tools/fuzz_parity.py --seconds 240 --seed 261 > gpurun_out/r03s/fuzz261.txt 2>&1; tail -n 1 gpurun_out/r03s/fuzz261.txt
