#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 2400 python3 -m pytest tests -m gpu -q -x > gpurun_out/r04/gpu_tests.log 2>&1; echo "suite rc=$?" >> gpurun_out/r04/gpu_tests.log
tail -5 gpurun_out/r04/gpu_tests.log
bash tools/profile_round4.sh > gpurun_out/r04/profile.log 2>&1
tail -70 gpurun_out/r04/profile.log
