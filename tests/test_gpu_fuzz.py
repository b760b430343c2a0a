"""Randomised shape sweeps (run with -m gpu on an MI355X): the tests/sweeps/fuzz_*.py generators at a size that takes seconds -- random bit widths, group
sizes, ragged N, every dispatch boundary of M, random mixed-bit band structures, random layer lists -- each configuration against the CPU
restatement.  A configuration the library REFUSES (RuntimeError with its message) is acceptable; a wrong value, a NaN or a crash is not.
The long runs (2000 / 800 / 600 cases, no finding) are recorded in profiles/r05_fuzz.txt."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests", "sweeps"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


@pytest.mark.parametrize("seed", [101, 102])
def test_randomised_mpq_forward_shapes_against_the_oracle(seed):
    import fuzz_mpq_forward
    r = fuzz_mpq_forward.run(cases=120, seed=seed)
    assert not r["bad"], r["bad"][:3]
    assert r["ok"] >= 100, r  # the sweep must mostly land on configurations the library serves


@pytest.mark.parametrize("seed", [201])
def test_randomised_exl2_band_structures_against_the_oracle(seed):
    import fuzz_exl2_forward
    r = fuzz_exl2_forward.run(cases=80, seed=seed)
    assert not r["bad"], r["bad"][:3]
    assert r["ok"] >= 70, r


@pytest.mark.parametrize("seed", [301])
def test_randomised_layer_lists_against_the_oracle(seed):
    import fuzz_mpq_lists
    r = fuzz_mpq_lists.run(cases=60, seed=seed)
    assert not r["bad"], r["bad"][:3]
    assert r["ok"] >= 50, r


@pytest.mark.parametrize("seed", [401])
def test_randomised_binary_integer_uniform_mbwq_and_grouped_calls_against_the_oracle(seed):
    import fuzz_other_ops
    r = fuzz_other_ops.run(cases=40, seed=seed)
    for op, v in r["ops"].items():
        assert not v["bad"], (op, v["bad"][:3])
        assert not v["refused"], (op, v["refused"])  # every configuration this generator draws is one the reference's layers accept
        assert v["ok"] >= 30, (op, v)


@pytest.mark.parametrize("seed", [501])
def test_random_call_programs_never_get_another_tensors_result_from_the_sibling_grouping(seed):
    """tests/sweeps/fuzz_sibling_groups.py: parents with 3..6 quantised children, random programs of calls (shared tensor, freed temporaries, clones,
    slices, in-place updates, a second tensor), several rounds, programs that change between rounds -- every output against the layer's own
    launch on the same input.  Groups must actually form (the sweep is not vacuous)."""
    import fuzz_sibling_groups
    r = fuzz_sibling_groups.run(parents=30, seed=seed)
    assert not r["bad"], r["bad"][:2]
    assert r["calls"] > 500 and r["grouped_launches"] > 0 and r["served_from_group"] > 0, r
