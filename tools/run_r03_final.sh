# round 3, final build of the second session: whole GPU suite + smoke, then the evidence collection (part 1)
mkdir -p gpurun_out/r03f
python -m pytest tests -m gpu -x -q > gpurun_out/r03f/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03f/tests.log
tail -n 4 gpurun_out/r03f/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03f/smoke.log 2>&1; tail -n 1 gpurun_out/r03f/smoke.log
bash tools/collect_profiles.sh r03f 1 > gpurun_out/r03f/collect1.log 2>&1; tail -n 25 gpurun_out/r03f/collect1.log
