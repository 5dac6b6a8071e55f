"""Dev tool: command line of tests/fuzzers.py (the time-boxed versions run inside `-m gpu`: tests/test_gpu_fuzz.py).
usage: python tools/fuzz_search.py {search|rows64|attention|split_gemm|soak|small_width|multi_search|prepared|wide_k} [cases] [seed]      (DBGLIB=<path>: a variant build)"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
from medtok_amd import _lib
if os.environ.get("DBGLIB"):           # (a variant build of the library, e.g. tools/r05/build_mutants.py)
    _lib.use_library(os.environ["DBGLIB"])
import fuzzers
which = sys.argv[1] if len(sys.argv) > 1 else "search"
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
fn = {"search": fuzzers.fuzz_search, "rows64": fuzzers.fuzz_rows64, "attention": fuzzers.fuzz_attention, "split_gemm": fuzzers.fuzz_split_gemm,
      "soak": fuzzers.soak_forward, "small_width": fuzzers.fuzz_small_width, "multi_search": fuzzers.fuzz_multi_search, "prepared": fuzzers.fuzz_prepared, "wide_k": fuzzers.fuzz_wide_k}[which]
ran, bad = fn(cases, seed, log=lambda s: print("MISMATCH", s, flush=True))
print(f"{which}: {ran} cases, {len(bad)} mismatches", flush=True)
sys.exit(1 if bad else 0)
