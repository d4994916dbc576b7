mkdir -p gpurun_out/r03y
python -m pytest tests -m gpu -x -q -k "k1f or c2 or low_dim or golden or k1a" > gpurun_out/r03y/tests5.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03y/tests5.log
tail -n 4 gpurun_out/r03y/tests5.log
sh tools/abc2.sh k1f k1fsc k1fsc4 2>&1 | tee gpurun_out/r03y/ab_k1f9.txt
