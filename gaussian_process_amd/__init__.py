"""MI355X-native exact Gaussian-process regression hot path behind the reference's Python API
(emakryo/gaussian_process: kernels.py, regression.py, expectation_propagation.py, distillation.py)."""
from .regression import GaussianProcessRegression
from .expectation_propagation import GaussianProcessExpectationPropagation
from .distillation import GaussianProcessDistillation
from .kernels import RBF, ARD, KernelValue

__all__ = ["GaussianProcessRegression", "GaussianProcessExpectationPropagation",
           "GaussianProcessDistillation", "RBF", "ARD",
           "KernelValue"]
