# Round-3 GPU test pass: the whole -m gpu suite in one process, log kept under gpurun_out/<tag>/.
#   gpurun --timeout 1190 -- 'bash tools/round3_tests.sh r3a'
TAG=${1:-r3a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=15 > $O/gpu_tests.log 2>&1
rc=$?
tail -25 $O/gpu_tests.log
exit $rc
