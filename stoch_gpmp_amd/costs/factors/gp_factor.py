"""GPFactor -- same interface as reference costs/factors/gp_factor.py (host-side mirror).

The kernels rebuild Phi and Q^-1 from (dim, sigma | Q_c_inv, d_t) themselves (K1 in
csrc/prior_factor.hip, the GP term of csrc/cost_sweep.hip); `phi`, `Q_c_inv`, `Q_inv` are exposed as
small tensors for API parity only."""
import torch


class GPFactor:
    def __init__(self, dim, sigma, d_t, num_factors, tensor_args=None, Q_c_inv=None):
        self.dim = dim
        self.sigma = sigma
        self.d_t = d_t
        self.tensor_args = tensor_args
        self.state_dim = self.dim * 2
        self.num_factors = num_factors
        self.user_Q_c_inv = Q_c_inv

    def calc_phi(self):
        """gp_factor.py:36-42: Phi = [[I, d_t I], [0, I]]."""
        phi = torch.eye(self.state_dim, **self.tensor_args)
        phi[:self.dim, self.dim:] = torch.eye(self.dim, **self.tensor_args) * self.d_t
        return phi

    @property
    def phi(self):                                      # (the reference keeps calc_phi()'s result as an attribute)
        return self.calc_phi()

    @property
    def Q_c_inv(self):                                  # gp_factor.py:25-27
        q = self.user_Q_c_inv
        if q is None:
            q = torch.eye(self.dim, **self.tensor_args) / self.sigma ** 2
        return torch.zeros(self.num_factors, self.dim, self.dim, **self.tensor_args) + q

    def calc_Q_inv(self):
        """gp_factor.py:44-52: Q^-1 = [[12 d_t^-3, -6 d_t^-2], [-6 d_t^-2, 4 d_t^-1]] (x) Q_c^-1 -> [num_factors, state_dim, state_dim]."""
        qc = self.Q_c_inv
        m1, m2, m3 = 12. * (self.d_t ** -3.) * qc, -6. * (self.d_t ** -2.) * qc, 4. * (self.d_t ** -1.) * qc
        return torch.cat((torch.cat((m1, m2), dim=-1), torch.cat((m2, m3), dim=-1)), dim=-2)

    @property
    def Q_inv(self):                                    # (the reference keeps calc_Q_inv()'s result as an attribute)
        return self.calc_Q_inv()

    @property
    def H1(self):                                       # gp_factor.py:30
        return self.phi.unsqueeze(0).repeat(self.num_factors, 1, 1)

    @property
    def H2(self):                                       # gp_factor.py:31-34
        return -torch.eye(self.state_dim, **self.tensor_args).unsqueeze(0).repeat(self.num_factors, 1, 1)

    def get_error(self, x_traj, calc_jacobian=True):
        """gp_factor.py:54-67: e_i = x_{i+1} - Phi x_i ([B, num_factors, state_dim, 1]); with calc_jacobian (the reference's
        default) also the constant Jacobians H1 = Phi, H2 = -I."""
        error = (x_traj[:, 1:] - x_traj[:, :-1] @ self.phi.t()).unsqueeze(-1)
        if calc_jacobian:
            H1 = self.phi.unsqueeze(0).repeat(self.num_factors, 1, 1)
            H2 = -torch.eye(self.state_dim, **self.tensor_args).unsqueeze(0).repeat(self.num_factors, 1, 1)
            return error, H1, H2
        return error
