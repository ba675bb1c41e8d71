"""The tolerance that CAN be true -- CPU only.

north_star asks for <= 1e-5 relative cost error per agent against the CPU reference.  On this workload no two implementations
that differ in ONE rounding anywhere can meet that: tests/test_oracle_sensitivity.py moves v0 by one ulp and the converged costs
of single agents move by per cents (second-difference Hessians divide rounding noise by eps^2, the x10 ladder is discontinuous).
What IS stable is the population: the C3 workload is 4096 agents on a circle, and total cost, iteration counts, converged
fraction and the distribution of the per-agent costs barely move under every perturbation of the arithmetic that a different
but equally valid evaluation of the reference could have:
  * glibc's sin/cos/tan instead of the shared pure-IEEE ones                       (`libm`)
  * Eigen's SSE2 packet-pairwise product sums instead of index order              (`pairwise`: the reading DESIGN 2 cannot verify)
  * v_xx = 0.5 (v_xx + v_xx^T) through a temporary instead of aliased in place     (`mean`: the other unverifiable reading)
  * every agent's v0 moved by one ulp up / down                                    (`ulp+`, `ulp-`)
  * all of them at once                                                            (`all`)
This file states the population tolerance and tests it on every 8th agent of the C3 population (512 agents, H = 100, sequential,
max_outer 10); bench.py evaluates the same statistics for the GPU result against every variant on ALL 4096 agents
(`parity.population_tolerance`) with the tighter bounds of POPULATION_BOUNDS_4096.  The HIP path itself is bit-exact against the
default variant (tests/test_gpu_parity.py); this bounds how far ANY faithful evaluation may sit from it.
"""
import numpy as np
import pytest

from oracle.population import (POPULATION_BOUNDS_512, VARIANTS, oracle_variant, population_deviation,  # noqa: E402
                               population_statistics)


@pytest.fixture(scope="module")
def population(oracle):
    x0 = oracle.single_track_x0(4096)[::8]
    return x0, population_statistics(oracle_variant(oracle, x0))


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_population_statistics_hold_under_every_faithful_arithmetic(oracle, population, variant):
    x0, base = population
    r = oracle_variant(oracle, x0, **VARIANTS[variant])
    dev = population_deviation(population_statistics(r), base)
    for key, bound in POPULATION_BOUNDS_512.items():
        assert dev[key] <= bound, (variant, key, dev[key], bound)


def test_the_per_agent_tolerance_of_north_star_is_not_attainable(oracle, population):
    """the reason a population tolerance is needed at all: under the same variants most agents move by far more than 1e-5"""
    x0, _ = population
    base = oracle_variant(oracle, x0)
    for variant in ("libm", "pairwise", "ulp+"):
        r = oracle_variant(oracle, x0, **VARIANTS[variant])
        rel = np.abs(r["cost"] - base["cost"]) / np.abs(base["cost"])
        assert np.mean(rel > 1e-5) > 0.5 and rel.max() > 0.1, (variant, float(np.mean(rel > 1e-5)), float(rel.max()))


def test_the_default_variant_is_what_the_parity_tests_use(oracle, population):
    """variant (index order, aliased) with explicit arguments == the library's defaults, bit for bit"""
    x0, _ = population
    a = oracle_variant(oracle, x0[:32])
    b = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0[:32], None, 100, 0.5, strategy="sequential", max_outer=10)
    assert np.array_equal(a["cost"], b["cost"]) and np.array_equal(a["X"], b["X"])
