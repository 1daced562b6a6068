"""expectation_propagation.ipynb as a script: EP classification of make_moons (site loop on the GPU)."""
import numpy as np
from sklearn.datasets import make_moons
from sklearn.model_selection import train_test_split

from gaussian_process_amd import GaussianProcessExpectationPropagation, RBF

X, y = make_moons(100, noise=0.1, random_state=0)
y = 2 * y - 1
Xtr, Xte, ytr, yte = train_test_split(X, y, train_size=0.5, random_state=0)
np.random.seed(0)
ep = GaussianProcessExpectationPropagation(RBF())
ep.fit(Xtr, ytr)
print("test accuracy %.2f, LML %.4f" % (ep.score(Xte, yte), ep.log_marginal_likelihood()))
