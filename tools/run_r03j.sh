mkdir -p gpurun_out/r03j
python -m pytest tests -m gpu -x -q -k "k1c or driver or golden_recipe or small_whole or low_dim or foreign or k1024 or k512 or whole_call or pool" > gpurun_out/r03j/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03j/tests.log
tail -4 gpurun_out/r03j/tests.log
./nns-cuda_amd/nns_driver --repeat 5 > gpurun_out/r03j/driver.txt 2>&1
python tools/probe_depths.py --k512 2>&1 | grep points > gpurun_out/r03j/deep.txt
python tools/probe_depths.py --deep 2>&1 | grep -E "points" >> gpurun_out/r03j/deep.txt
python tools/probe_smallcall.py > gpurun_out/r03j/smallcall.txt 2>&1
cat gpurun_out/r03j/driver.txt gpurun_out/r03j/deep.txt; tail -20 gpurun_out/r03j/smallcall.txt
