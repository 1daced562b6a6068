"""distillation_illustration.ipynb as a script: hard labels y and a teacher's soft labels s, tied by rho."""
import numpy as np
from sklearn.model_selection import train_test_split

from gaussian_process_amd import GaussianProcessDistillation, RBF
from gaussian_process_amd.datasets import generate_gp

X, s = generate_gp(n_samples=100, random_state=6)
y = np.sign(s)
Xtr, Xte, ytr, yte, s_tr, s_te = train_test_split(X, y, s, train_size=0.5, random_state=0)
np.random.seed(0)
gpd = GaussianProcessDistillation(cov=RBF(1, 1), sigma=0.01, rho=1.)
gpd.fit(Xtr, ytr, s_tr)
print("rho = 1: test accuracy %.2f" % gpd.score(Xte, yte))
gpd.set_params(rho=0)
gpd.fit(Xtr, ytr, s_tr)
print("rho = 0: test accuracy %.2f" % gpd.score(Xte, yte))
