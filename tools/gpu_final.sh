#!/bin/bash
# round-3 final GPU session: whole GPU suite, then the bench lines / kernel stats / PMC figures under profiles/r03_*
export TMPDIR=/tmp
out=gpurun_out/final; mkdir -p $out
timeout 3000 python3 -m pytest tests -m gpu -q -x > $out/gpu_tests.log 2>&1; echo "suite rc=$?" >> $out/gpu_tests.log
tail -4 $out/gpu_tests.log
timeout 3000 bash tools/profile_round3.sh > $out/profile.log 2>&1; echo "profile rc=$?" >> $out/profile.log
tail -5 $out/profile.log
