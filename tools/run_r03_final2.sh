# final build: whole GPU suite + smoke
mkdir -p gpurun_out/r03h
python -m pytest tests -m gpu -x -q > gpurun_out/r03h/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03h/tests.log
tail -n 4 gpurun_out/r03h/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03h/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r03h/smoke.log; tail -n 2 gpurun_out/r03h/smoke.log
