# the round-end check in one call: the GPU suite, smoke(), every bundled sample through the CLI, the default bench line
R=${GRAFT_REPO_ROOT:-.}; cd $R; mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/full_gpu.log 2>&1; tail -2 gpurun_out/full_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/run_samples.sh gpurun_out/cli_all_samples.txt > /dev/null 2>&1; cat gpurun_out/cli_all_samples.txt
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; cat gpurun_out/final_bench.json
