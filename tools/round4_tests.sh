# Round-4 GPU test pass: the whole -m gpu suite in one process, log kept under gpurun_out/<tag>/.
#   gpurun --timeout 1190 -- 'bash tools/round4_tests.sh r4a'
TAG=${1:-r4a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=15 > $O/gpu_tests.log 2>&1
rc=$?
tail -25 $O/gpu_tests.log
exit $rc
