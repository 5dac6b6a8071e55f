#!/bin/bash
# (1) GPU suite for the modules touched by the re-entrancy refactor; (2) ten fresh processes of the shipped forward;
# (3) tools/ab_streams.py with the default number of hardware queues and with 8
export TMPDIR=/tmp
out=gpurun_out/r04f; mkdir -p $out
timeout 1800 python3 -m pytest tests/test_gpu_modules.py tests/test_gpu_robustness.py tests/test_gpu_train_step.py tests/test_gpu_bench_contract.py -m gpu -q -x > $out/tests.log 2>&1; tail -4 $out/tests.log
for i in 1 2 3 4 5 6 7 8 9 10; do python3 tools/ab_forward.py shipped 2>/dev/null; done | tee $out/ten_fresh.log
python3 tools/ab_streams.py 2>/dev/null | tee $out/ab_streams_default.log
GPU_MAX_HW_QUEUES=8 python3 tools/ab_streams.py 2>/dev/null | tee $out/ab_streams_q8.log
for i in 1 2 3; do GPU_MAX_HW_QUEUES=8 python3 tools/ab_forward.py shipped 2>/dev/null; done | tee $out/fresh_q8.log
