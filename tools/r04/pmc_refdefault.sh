export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_refdefault_$c -o p -- python3 bench.py --workload refdefault --steps 1 --warmup 1 --cpu-rows 0 --exact-steps 0 > $out/pmc_refdefault_$c.log 2>&1
  f=$(find $out/pmc_refdefault_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/pmc_refdefault_$c.csv
  rm -rf $out/pmc_refdefault_$c
done
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmcr_$c -o p -- python3 bench.py --workload refdefault --steps 1 --warmup 1 --cpu-rows 0 --exact-steps 0 > $out/pmcr_$c.log 2>&1
  f=$(find $out/pmcr_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/pmcr_refdefault_$c.csv
  rm -rf $out/pmcr_$c
done
python3 tools/pmc_summary.py $out refdefault 2>&1 | head -8
