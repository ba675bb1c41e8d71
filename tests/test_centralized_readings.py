"""C2 (BASELINE configs[1]: centralised, H = 100) is measured on the block-diagonal reading of the stacked problem; the reference's
literal algorithm finite-differences the SUMMED cost (centralized.hpp:18-38, multi_agent_problem.hpp:94-122) and is computable up to
16 agents.  VERDICT r2 item 6 asked for the deviation between the two AT C2's HORIZON.  It is large -- tens of per cent of the total
cost, different iteration counts -- and it is not a property of the reading: the literal solve moves by the same amount when ONE input
coordinate changes by ONE ulp (second-difference Hessians of a sum of N costs divide rounding noise by eps^2; both solves stall after
3-12 iterations on that noise).  So the statement that can be tested is: the block-diagonal reading lies inside the band the literal
solve itself spans under one-ulp inputs -- widened by the factor stated here -- and bench.py prints the numbers
(`also_config2.parity.vs_reference_stacked_solve`).  At T = 10 the two agree to 5e-3 (tests/test_gpu_parity.py)."""
import numpy as np
import pytest

from oracle.population import centralized_reading_deviation

BAND_FACTOR = 3.0        # block-diagonal vs literal <= BAND_FACTOR x (largest one-ulp self-deviation of the literal solve over N = 4..16)
ABSOLUTE_BOUND = 0.6     # and never more than this fraction of the total cost


@pytest.fixture(scope="module")
def curve(oracle):
    return [centralized_reading_deviation(oracle, n) for n in (4, 8, 12, 16)]


def test_the_literal_solve_is_chaotic_at_h100(curve):
    worst = max(max(c["literal_under_one_ulp_inputs_total_rel"]) for c in curve)
    assert worst > 0.2, worst                       # one ulp of one input coordinate: more than 20 % of the total cost somewhere
    assert any(len(set(c["literal_under_one_ulp_inputs_iterations"] + [c["iterations_literal"]])) > 1 for c in curve)     # and other iteration counts


def test_block_diagonal_reading_lies_inside_the_literal_solves_own_band(curve):
    band = max(max(c["literal_under_one_ulp_inputs_total_rel"]) for c in curve)
    for c in curve:
        assert c["block_diagonal_vs_literal_total_rel"] <= min(ABSOLUTE_BOUND, BAND_FACTOR * band), c
        assert np.isfinite(c["block_diagonal_vs_literal_per_agent_max_rel"])


def test_committed_curve_to_64_agents(curve):
    """profiles/r3_c2_readings_curve.json (tools/c2_readings_curve.py, minutes of CPU at 64 agents) extends the curve to the sizes VERDICT r2
    item 6 named (32-64 agents).  Its rows up to 16 agents must be what the oracle computes here (so the file is the oracle's, not an
    edited, curve); the statement about the band is asserted on all of its rows."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r3_c2_readings_curve.json")
    rows = {r["agents"]: r for r in json.load(open(path))["curve"]}
    assert {2, 4, 8, 16, 32, 64} <= set(rows)
    for c in curve:
        if c["agents"] in rows:
            for key in ("block_diagonal_vs_literal_total_rel", "iterations_literal", "iterations_block_diagonal", "literal_under_one_ulp_inputs_total_rel"):
                assert rows[c["agents"]][key] == c[key], (c["agents"], key)
    band = max(max(r["literal_under_one_ulp_inputs_total_rel"]) for r in rows.values())
    for r in rows.values():
        assert r["block_diagonal_vs_literal_total_rel"] <= min(ABSOLUTE_BOUND, BAND_FACTOR * band), r
