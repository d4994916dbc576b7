# round 3, session 2: K1f (VALU filter + exact re-rank for k <= 3) — parity first, then C2 / C1 / driver shapes
mkdir -p gpurun_out/r03y
python -m pytest tests -m gpu -x -q -k "c2 or low_dim or golden or k1a or index_base or determinism or adversarial or short_randomised or chunked_upload_exact or search_indices or midsize or extreme or offset or device_api" > gpurun_out/r03y/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03y/tests.log
tail -n 6 gpurun_out/r03y/tests.log
python bench.py --workload c2 --steps 200 --warmup 20 --no-also > gpurun_out/r03y/bench_c2.json 2> gpurun_out/r03y/bench_c2.err; tail -c 1500 gpurun_out/r03y/bench_c2.json
python tools/probe_shapes.py > gpurun_out/r03y/shapes.txt 2>&1; tail -n 12 gpurun_out/r03y/shapes.txt
