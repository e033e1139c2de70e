"""UnaryFactor: a Gaussian factor pulling one state towards `mean` with isotropic standard deviation
`sigma` (interface of reference costs/factors/unary_factor.py, host-side mirror).

The HIP kernels consume only (sigma, mean): the weight I / sigma^2 is applied inside them.  The
tensors below (`K`, errors, the identity Jacobian) are small setup-time / inspection values."""
import torch


class UnaryFactor:
    def __init__(self, dim, sigma, mean=None, tensor_args=None):
        self.dim, self.sigma, self.tensor_args = dim, sigma, tensor_args
        self.mean = mean if mean is not None else torch.zeros(dim, **tensor_args)

    @property
    def precision(self):
        """Scalar weight 1 / sigma^2 of every state component."""
        return 1.0 / (self.sigma * self.sigma)

    @property
    def K(self):
        """[dim, dim] weight matrix (unary_factor.py:19)."""
        return self.precision * torch.eye(self.dim, **self.tensor_args)

    def get_error(self, x, calc_jacobian=True):
        """mean - x, and with `calc_jacobian` (the reference's default) also H = -d error / d x = I per batch entry
        (unary_factor.py:22-29): -> error [B, dim, 1], H [B, dim, dim]."""
        residual = torch.sub(self.mean, x)
        if not calc_jacobian:
            return residual
        batch = x.shape[0]
        jac = torch.eye(self.dim, **self.tensor_args).expand(batch, self.dim, self.dim).clone()
        return residual.reshape(batch, self.dim, 1), jac

    def set_mean(self, x):
        self.mean = x.detach().clone()
