"""UnaryFactor -- same interface as reference costs/factors/unary_factor.py (host-side mirror).

On the HIP path only `sigma` and `mean` are consumed (K = I / sigma^2 is applied inside the
kernels); `K` and `get_error` exist for API parity and are tiny setup-time torch expressions."""
import torch


class UnaryFactor:
    def __init__(self, dim, sigma, mean=None, tensor_args=None):
        self.sigma = sigma
        self.mean = torch.zeros(dim, **tensor_args) if mean is None else mean
        self.tensor_args = tensor_args
        self.dim = dim

    @property
    def K(self):
        return torch.eye(self.dim, **self.tensor_args) / self.sigma ** 2   # unary_factor.py:19

    def get_error(self, x, calc_jacobian=False):
        """unary_factor.py:22-29: error = mean - x; Jacobian H = I."""
        error = self.mean - x
        if calc_jacobian:
            H = torch.eye(self.dim, **self.tensor_args).unsqueeze(0).repeat(x.shape[0], 1, 1)
            return error.reshape(x.shape[0], self.dim, 1), H
        return error

    def set_mean(self, x):
        self.mean = x.clone().detach()
