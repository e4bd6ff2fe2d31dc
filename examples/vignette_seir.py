"""The reference's vignette (vignette.ipynb cells 3-8) run through this package: same calls, numpy in place of tf.*.

    python examples/vignette_seir.py            # needs an MI355X and the built library (python -m magi_v2_amd.build)

Data: the 81 thinned rows of data/SEIR_seed=0.csv held in tests/golden/g3_pipeline.npz (every 50th row, t <= 4)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import magi_v2  # noqa: E402  (the drop-in module)


def f_vec(t, X, thetas):
    """SEIR with S implicit (vignette.ipynb cell 3); theta = (beta, gamma, sigma)."""
    S = 1.0 - np.reshape(np.sum(X, axis=1), (-1, 1))
    return np.concatenate([(thetas[0] * S * X[:, 1:2]) - (thetas[2] * X[:, 0:1]),
                           (thetas[2] * X[:, 0:1]) - (thetas[1] * X[:, 1:2]),
                           (thetas[1] * X[:, 1:2])], axis=1)


def main():
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "g3_pipeline.npz"))
    ts_obs, X_obs = g["seir3_ts_obs"], g["seir3_X_obs"]

    model = magi_v2.MAGI_v2(D_thetas=3, ts_obs=ts_obs, X_obs=X_obs, bandsize=80, f_vec=f_vec)
    # the vignette's call; hparam_fit_on="observed" is this package's documented alternative
    model.initial_fit(discretization=1, verbose=True)
    print("phi1", model.phi1s, "phi2", model.phi2s, "sigma_init", np.sqrt(model.sigma_sqs_init), "theta_init", model.thetas_init)

    results = model.predict(num_results=1000, num_burnin_steps=1000, verbose=True)
    th = results["thetas_samps"]
    print("theta posterior mean", th.mean(axis=0), "sd", th.std(axis=0), "(truth 6, 0.6, 1.8)")
    print("mean tree depth", results["kernel_results"]["tree_depth"].mean(), "minutes", results["minutes_elapsed"])


if __name__ == "__main__":
    main()
