#!/bin/bash
# Dev tool (GPU box): long runs of tests/fuzzers.py on the round's final library -> gpurun_out/r06/fuzz_long_run.txt
out=gpurun_out/r06; mkdir -p $out
: > $out/fuzz_long_run.txt
i=301
for spec in "search 12000" "rows64 25000" "split_gemm 8000" "attention 3000" "small_width 6000" "multi_search 5000" "prepared 1500" "wide_k 1500" "soak 1500"; do
  set -- $spec
  s=$(date +%s)
  timeout 600 python tools/fuzz_search.py $1 $2 $i 2>&1 | tail -2 | tr '\n' ' ' >> $out/fuzz_long_run.txt
  echo " (seed $i; $(( $(date +%s) - s )) s)" >> $out/fuzz_long_run.txt
  i=$((i+1))
done
python tools/r05/repeat_large_searches.py 300 2>&1 | tail -3 >> $out/fuzz_long_run.txt
cat $out/fuzz_long_run.txt
