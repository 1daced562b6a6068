"""ARD_regression_2d.ipynb as a script: 2-D ARD regression, RMSE before and after empirical Bayes."""
import numpy as np
from sklearn.model_selection import train_test_split

from gaussian_process_amd import ARD, GaussianProcessRegression
from gaussian_process_amd.datasets import generate_gp

X, y = generate_gp(n_samples=200, n_features=2, kernel=ARD(beta=[1, 3]), random_state=0, noise=0.01)
Xtr, Xte, ytr, yte = train_test_split(X, y, train_size=0.5, random_state=0)
gp = GaussianProcessRegression(cov=ARD(beta=[1, 1]))
gp.fit(Xtr, ytr)
print("RMSE with beta = [1, 1]      : %.4f" % np.sqrt(np.mean((gp.predict(Xte) - yte) ** 2)))
gp.empirical_bayes()
print("RMSE after empirical Bayes   : %.4f, beta %s" % (np.sqrt(np.mean((gp.predict(Xte) - yte) ** 2)),
                                                      np.round(gp.cov.beta, 4)))
