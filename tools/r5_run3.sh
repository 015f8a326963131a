#!/bin/bash
# GPU box: the long-template tests, the round-5 tests again, the api_large shapes with lazy lists.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_run3; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_long.py -x -q > $OUT/long.txt 2>&1; rc=$?
tail -40 $OUT/long.txt
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python3 -m pytest tests/test_gpu_round5.py tests/test_gpu_api.py tests/test_gpu_round2.py tests/test_gpu_round4.py -x -q > $OUT/r5.txt 2>&1; rc=$?
tail -5 $OUT/r5.txt
[ $rc -ne 0 ] && { grep -n "Error\|assert" $OUT/r5.txt | tail -20; exit 1; }
timeout -k 10 400 python3 tools/r5_api_large_probe.py > $OUT/api_large_probe.txt 2>&1 || { tail -20 $OUT/api_large_probe.txt; exit 1; }
cat $OUT/api_large_probe.txt
