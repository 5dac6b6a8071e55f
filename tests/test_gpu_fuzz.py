"""The randomised loops of tests/fuzzers.py inside `-m gpu`, time-boxed and with fixed seeds (round-4 verdict: three wrong-result
bugs were found by these loops while they lived outside the suite).  Budgets: search 60 s, narrow rows 30 s, attention 60 s,
split GEMM 30 s, soak 30 s -- a box that is slow just runs fewer cases; a floor on the case count keeps the test meaningful."""
import pytest

pytestmark = pytest.mark.gpu


def _report(name, ran, bad, floor):
    print(f"{name}: {ran} cases, {len(bad)} mismatches")
    assert not bad, "\n".join(bad[:10])
    assert ran >= floor, f"{name}: only {ran} cases ran inside the budget (floor {floor})"


def test_fuzz_search_filter_equals_exact_path(dev):
    import fuzzers
    ran, bad = fuzzers.fuzz_search(cases=400, seed=20251, budget_s=60, max_rows=70001)
    _report("fuzz_search", ran, bad, 25)


def test_fuzz_rows64_equals_exact_path_and_general_kernel(dev):
    import fuzzers
    ran, bad = fuzzers.fuzz_rows64(cases=400, seed=20252, budget_s=30, max_rows=70001)
    _report("fuzz_rows64", ran, bad, 15)


def test_fuzz_attention_all_variants_match_oracle(dev, oracle):
    import fuzzers
    ran, bad = fuzzers.fuzz_attention(cases=400, seed=20253, budget_s=60)
    _report("fuzz_attention", ran, bad, 10)


def test_fuzz_split_gemm_matches_fp64(dev):
    import fuzzers
    ran, bad = fuzzers.fuzz_split_gemm(cases=1000, seed=20254, budget_s=30)
    _report("fuzz_split_gemm", ran, bad, 40)


def test_soak_forward_multi_stream_is_bit_stable(dev):
    import fuzzers
    ran, bad = fuzzers.soak_forward(runs=60, seed=0, budget_s=30)
    _report("soak_forward", ran, bad, 5)


def test_fuzz_small_width_cross_attention_equals_the_layer_path(dev):
    import fuzzers
    ran, bad = fuzzers.fuzz_small_width(cases=400, seed=20255, budget_s=40)
    _report("fuzz_small_width", ran, bad, 15)


def test_fuzz_batched_searches_equal_the_single_calls(dev):
    import fuzzers
    ran, bad = fuzzers.fuzz_multi_search(cases=400, seed=20256, budget_s=30)
    _report("fuzz_multi_search", ran, bad, 15)


def test_fuzz_prepared_codebook_equals_unprepared_searches(dev):
    import fuzzers
    ran, bad = fuzzers.fuzz_prepared(cases=300, seed=20257, budget_s=30)
    _report("fuzz_prepared", ran, bad, 10)


def test_fuzz_more_than_eight_codes_per_row_equal_the_oracle(dev, oracle):
    import fuzzers
    ran, bad = fuzzers.fuzz_wide_k(cases=200, seed=20258, budget_s=40)
    _report("fuzz_wide_k", ran, bad, 10)
