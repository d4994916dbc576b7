mkdir -p gpurun_out/r03y
python -m pytest tests -m gpu -x -q -k "c2 or low_dim or golden or k1a or determinism or adversarial or midsize or extreme or offset" > gpurun_out/r03y/tests2.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03y/tests2.log
tail -n 4 gpurun_out/r03y/tests2.log
sh tools/abc2.sh k1f k1foff 2>&1 | tee gpurun_out/r03y/ab_k1f3.txt
