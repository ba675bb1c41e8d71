"""multi_agent_solver_amd -- MI355X-native batched iLQR behind the `--solver ilqr` surface of
markomiz/multi_agent_solver.  HIP kernels + C ABI live in csrc/ (built to lib/libmas_hip.so);
`batch` is the thin handle over the C ABI, `api` mirrors the reference's OCP / Solver /
Strategy interface.  There is no CPU fallback: without libmas_hip.so and a HIP device every
solve raises."""
from .batch import (BatchSolver, MODEL_BICYCLE6, MODEL_LQR_IDENTITY, MODEL_PENDULUM, MODEL_SINGLE_TRACK,
                    MODEL_SINGLE_TRACK_LANE)

__all__ = ["BatchSolver", "MODEL_SINGLE_TRACK", "MODEL_LQR_IDENTITY", "MODEL_PENDULUM",
           "MODEL_SINGLE_TRACK_LANE", "MODEL_BICYCLE6"]
