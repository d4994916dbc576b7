mkdir -p gpurun_out/r03k
python -m pytest tests -m gpu -x -q -k "driver or golden or whole_call or small_whole or foreign or pool or reentrant or multi or restore or errors" > gpurun_out/r03k/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03k/tests.log
tail -4 gpurun_out/r03k/tests.log
for i in 1 2; do for v in prod synccopies; do
 echo "== $v" >> gpurun_out/r03k/ab.txt
 NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python tools/probe_smallcall.py 2>&1 | grep "k=" >> gpurun_out/r03k/ab.txt
 NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python tools/probe_midcall.py 2>&1 | grep -v amdgpu >> gpurun_out/r03k/ab.txt
done; done
./nns-cuda_amd/nns_driver --repeat 5 > gpurun_out/r03k/driver.txt 2>&1
python tools/wholecall_c3.py 2>&1 | grep -v amdgpu > gpurun_out/r03k/wholecall_c3.txt
cat gpurun_out/r03k/ab.txt; grep CudaCall gpurun_out/r03k/driver.txt | awk 'NR%5==0'; cat gpurun_out/r03k/wholecall_c3.txt
