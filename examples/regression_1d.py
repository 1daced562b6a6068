"""regression_1d.ipynb as a script: 1-D RBF regression, fixed hyper-parameters, then empirical Bayes."""
import numpy as np

from gaussian_process_amd import GaussianProcessRegression, RBF
from gaussian_process_amd.datasets import generate_gp

X, y = generate_gp(n_samples=15, kernel=RBF(beta=1.), random_state=1, noise=0.1)
Xs = np.linspace(-3, 3).reshape(-1, 1)
gp = GaussianProcessRegression(cov=RBF(), sigma=0.1)
gp.fit(X, y)
mean, cov = gp.predict_with_variance(Xs)
print("fixed theta : LML %.4f, predictive sd in [%.3f, %.3f]" % (gp.log_marginal_likelihood(),
      np.sqrt(np.diag(cov)).min(), np.sqrt(np.diag(cov)).max()))
gp.empirical_bayes()
print("optimised   : LML %.4f, beta %.4f, cov sigma %.4f, noise %.5f" % (gp.log_marginal_likelihood(),
      gp.cov.beta, gp.cov.sigma, gp.sigma))
