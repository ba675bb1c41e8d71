"""The two readings of the centralised strategy at C2's horizon (T = 100), CPU oracle only: the literal stacked solve
(centralized.hpp:18-38 + multi_agent_problem.hpp:94-122, finite differences of the SUMMED cost) against the block-diagonal reading the
product runs beyond 16 agents, and the literal solve against itself under one-ulp inputs, for 2 ... 64 agents.  The oracle's literal
solve is bit-equal to the device's (mas_hip_run_centralized_dense, <= 16 agents: tests/test_gpu_parity.py); beyond 16 agents only the
oracle runs it (4 (4N)^2 evaluations of the N-agent cost per time step: 64 agents take minutes per solve on 8 cores).
usage: python tools/c2_readings_curve.py [max_agents] > profiles/r3_c2_readings_curve.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle.population import centralized_reading_deviation  # noqa: E402

top = int(sys.argv[1]) if len(sys.argv) > 1 else 64
O.set_trig_mode("det")
curve = []
for n in (2, 4, 8, 16, 32, 64):
    if n > top:
        break
    t0 = time.time()
    c = centralized_reading_deviation(O, n, T=100, dt=0.5)
    c["oracle_seconds"] = round(time.time() - t0, 1)
    curve.append(c)
    print(n, c["oracle_seconds"], "s", file=sys.stderr, flush=True)
json.dump({"_what": __doc__.split("usage:")[0].strip(), "workload": "single-track, centralised strategy, H = 100, dt = 0.5 (BASELINE configs[1] at 2 ... 64 agents)",
           "trig": "det", "curve": curve}, sys.stdout, indent=1)
