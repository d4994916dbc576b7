# K4's DMA pieces on consecutive steps (prod) against two steps apart (sp2 = -DNNS_F_DMA_SP=2, the old rule): parity, A/B on C5
mkdir -p gpurun_out/r03b
python -m pytest tests -m gpu -x -q -k "bf16 or c5" > gpurun_out/r03b/tests_sp.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03b/tests_sp.log
tail -n 3 gpurun_out/r03b/tests_sp.log
bash tools/ab5.sh prod sp2 2>&1 | tee gpurun_out/r03b/ab_sp_c5_2.txt
