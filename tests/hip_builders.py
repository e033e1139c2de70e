"""HIP-side twins of the oracle problems in tests/scenarios.py (the builders live in the package so
that bench.py can use them without importing test code)."""
from stoch_gpmp_amd.workloads import (hip_panda_cost, hip_panda_planner, hip_planar_cost,  # noqa: F401
                                      hip_planar_planner)
