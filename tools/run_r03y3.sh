mkdir -p gpurun_out/r03y
python -m pytest tests -m gpu -x -q -k "k1f or c2 or low_dim or golden or k1a or largest_ref_count_int32 or largest_query or index_base or search_indices or chunked_upload_exact" > gpurun_out/r03y/tests3.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03y/tests3.log
tail -n 6 gpurun_out/r03y/tests3.log
python tools/fuzz_parity.py --lowdim --seconds 200 --seed 301 > gpurun_out/r03y/fuzz301_lowdim.txt 2>&1; tail -n 3 gpurun_out/r03y/fuzz301_lowdim.txt
