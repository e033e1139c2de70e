"""FieldFactor -- same interface as reference costs/factors/field_factor.py (host-side mirror)."""


class FieldFactor:
    def __init__(self, n_dof, sigma, traj_range):
        self.sigma = sigma
        self.n_dof = n_dof
        self.traj_range = traj_range
        self.length = traj_range[1] - traj_range[0]
        self.K = 1. / (sigma ** 2)                      # field_factor.py:16

    def get_error(self, q_trajs, field, x_trajs=None, calc_jacobian=True, fk_chain=None, **observations):
        """field_factor.py:18-40 -> (error [B, length], H [B, length, n_dof]) with calc_jacobian (the reference's default),
        else error alone.  The reference differentiates the field through the FK callable with autograd; here the
        Jacobian is analytic and needs the URDF chain itself (`fk_chain`, e.g. CostComposite.chain; a field may carry
        its own as `field.fk_chain`) -- link frames `x_trajs` are then not used."""
        batch = q_trajs.shape[0]
        a, b = self.traj_range
        if calc_jacobian:
            if fk_chain is None:
                fk_chain = getattr(field, "fk_chain", None)
            if fk_chain is None:
                raise ValueError("calc_jacobian=True (the default, as in the reference) needs fk_chain= (the URDF chain of the "
                                 "composite's FK): the analytic Jacobian replaces the reference's autograd pass")
            if not hasattr(field, "compute_cost_and_grad"):
                raise NotImplementedError(f"{type(field).__name__} has no analytic Jacobian")
            q = q_trajs[:, a:b, :self.n_dof].reshape(-1, self.n_dof)
            err, grad = field.compute_cost_and_grad(q, fk_chain, **observations)
            return err.reshape(batch, self.length), -grad.reshape(batch, self.length, self.n_dof)
        if x_trajs is not None:
            states = x_trajs[:, a:b]
        else:
            states = q_trajs[:, a:b, :self.n_dof].reshape(-1, self.n_dof)
        return field.compute_cost(states, **observations).reshape(batch, self.length)
