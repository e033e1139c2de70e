"""CPU oracle for the StochGPMP hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in plain torch-CPU / numpy, the algorithm of the reference
`anindex/stoch_gpmp` for the path `StochGPMP.optimize()` -> `CostComposite.eval()`.
It is the *checker* for the HIP kernels in `stoch_gpmp_amd/csrc`, never the product:

  * only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
    may import it;
  * nothing under `stoch_gpmp_amd/` imports it, and the product path raises when the
    HIP library is missing instead of falling back to this code.

Pinning status (see DESIGN.md "Oracle"):
  * planner / prior / sampler / GP, goal-prior and grid-collision costs / sphere and
    self-distance fields / softmax update: PINNED against the reference itself, imported
    in the build container from /root/reference by `oracle/gen_golden.py`, whose outputs
    are committed under `tests/golden/` and re-checked by `tests/test_oracle_golden.py`.
  * Panda forward kinematics (`oracle/fk.py`): PARITY UNPINNED.  The reference obtains FK
    from the un-vendored, un-versioned third-party package `torch_robotics`
    (reference README.md:19, costs/fields.py:4, examples/panda_environment.py:13-20,47,98)
    which is absent here; FK is restated from the URDF constants
    (assets/franka_description/robots/panda_arm_no_gripper.urdf:41-235) with standard
    URDF semantics.  Everything downstream of FK is pinned by handing the oracle FK to the
    *reference's* CostComposite as its injected `FK` callable.
"""
