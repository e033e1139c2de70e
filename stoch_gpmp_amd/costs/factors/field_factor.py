"""FieldFactor -- same interface as reference costs/factors/field_factor.py (host-side mirror)."""


class FieldFactor:
    def __init__(self, n_dof, sigma, traj_range):
        self.sigma = sigma
        self.n_dof = n_dof
        self.traj_range = traj_range
        self.length = traj_range[1] - traj_range[0]
        self.K = 1. / (sigma ** 2)                      # field_factor.py:16

    def get_error(self, q_trajs, field, x_trajs=None, calc_jacobian=False, **observations):
        if calc_jacobian:
            raise NotImplementedError("Jacobians belong to the GPMP planner (out of scope, SURVEY.md 8f)")
        batch = q_trajs.shape[0]
        a, b = self.traj_range
        if x_trajs is not None:
            states = x_trajs[:, a:b]
        else:
            states = q_trajs[:, a:b, :self.n_dof].reshape(-1, self.n_dof)
        return field.compute_cost(states, **observations).reshape(batch, self.length)
