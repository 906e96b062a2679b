# One GPU call at the end of a round: the GPU suite, the randomised sweeps, the profile collection on the final build, the
# side measurements the docs quote.   usage (GPU box): bash tools/final_collect.sh <tag>
set -u
TAG=${1:-final}
mkdir -p gpurun_out
(timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_gputest.log)
tail -3 gpurun_out/${TAG}_gputest.log
for sd in 21 22; do python tests/fuzz_blend.py 1500 $sd > gpurun_out/${TAG}_fuzz_blend_$sd.log 2>&1; tail -2 gpurun_out/${TAG}_fuzz_blend_$sd.log; done
python tests/fuzz_adjust.py 600 24 > gpurun_out/${TAG}_fuzz_adjust.log 2>&1; tail -1 gpurun_out/${TAG}_fuzz_adjust.log
bash tools/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1; tail -5 gpurun_out/${TAG}_collect.log
python tools/virtual_scaling.py > gpurun_out/${TAG}_virtual_scaling.json 2> gpurun_out/${TAG}_virtual_scaling.err; tail -c 600 gpurun_out/${TAG}_virtual_scaling.json
python tools/process_timing.py > gpurun_out/${TAG}_process.json 2> gpurun_out/${TAG}_process.err; tail -c 400 gpurun_out/${TAG}_process.json
bash tools/pmc_diag.sh $TAG > gpurun_out/${TAG}_pmc.log 2>&1; tail -3 gpurun_out/${TAG}_pmc.log
