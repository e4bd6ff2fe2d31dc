"""BASELINE config 4 across the GPUs of a node: D datasets x C chains = D*C independent (dataset, chain) units.

The reference fits one dataset with one chain per process (magi_v2.py:383-395; data/SEIR_beta=6_gamma=0.6_sigma=1.8_alpha=*_seed=*.csv);
nothing couples two datasets or two chains, so the path shards with no data-path collective: whole datasets are
dealt round-robin to the ranks (shard.shard_units) -- the chains of a dataset share a GPU and its matrices -- every rank builds the
matrices of ITS datasets on ITS GPU (one handle per dataset: all of them resident before the timed region), samples them, and ONE
gather (shard.gather_samples: RCCL on GPUs, gloo in the rehearsals) brings the post-burn-in samples of all units to rank 0 in global
unit order.  A unit's Philox stream is keyed by its global id (dataset * C + chain), so its samples do not depend on the number of GPUs.
"""
from __future__ import annotations

import threading
from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import host
from .shard import gather_samples, shard_units


def problem_setup(I, X_obs, P: int) -> dict:
    """Host-side constants of a dataset at the reference's STARTING hyper-parameters (magi_v2.py:85-100, 105, 114, 277, 299-300)."""
    Xi = host.linear_interpolate(X_obs)
    hp = host.hparams_initial(Xi)
    N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
    Xhat = host.cubic_smoother(I, Xi)
    LB = host.sigma_sqs_lower_bound(Xhat)
    sig_pre0, th_pre0 = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(P), LB)
    return dict(I=np.asarray(I, dtype=np.float64).reshape(-1), Xi=Xi, hp=hp, N_ds=N_ds.astype(np.float64), beta=float(beta), idx=idx, y=y, Xhat=Xhat, LB=LB,
                sig_pre0=sig_pre0, th_pre0=th_pre0, mu=Xi.mean(axis=0))


def alpha_sweep_datasets(npz_path: str, discretization: int = 1) -> List[Tuple[str, dict]]:
    """The ten alpha-sweep datasets (vignette thinning: 81 rows each, columns t, S, E, I, R observed, clipped at 0 as vignette.ipynb:100-112
    does) in a fixed order -- alpha = 0.05 seeds 0..4, then alpha = 0.15 seeds 0..4 -- discretised as the vignette is (N = 161)."""
    z = np.load(npz_path)
    names = sorted(k for k in z.files if k.startswith("alpha="))
    out = []
    for name in names:
        rows = z[name]
        I, X = host.discretize(rows[:, 0], np.clip(rows[:, 1:5], 0.0, None), discretization)
        out.append((name, problem_setup(np.asarray(I).reshape(-1), X, 3)))
    return out


class SweepRunner:
    """The datasets one rank owns: a MagiEngine (handle) per dataset on the rank's GPU, matrices built there."""

    def __init__(self, device: int, datasets: Sequence[Tuple[str, dict]], chains_per_dataset: int, rank: int, world: int,
                 bandsize=80, drift: str = "seir4"):
        from .engine import MagiEngine
        self.chains = chains_per_dataset
        self.units = shard_units(len(datasets), chains_per_dataset, rank, world)          # [(dataset index, [global unit ids])]
        self.engines, self.pbs, self.names = [], [], []
        self._captured = False
        for ds, _ in self.units:
            name, pb = datasets[ds]
            eng = MagiEngine(device)
            eng.build_matrices(pb["I"], pb["hp"]["phi1s"], pb["hp"]["phi2s"], 2.01, bandsize=bandsize, want_host=False)
            eng.set_problem(pb["mu"], pb["N_ds"], pb["idx"], pb["y"], pb["beta"], pb["LB"], drift)
            self.engines.append(eng); self.pbs.append(pb); self.names.append(name)

    @property
    def unit_ids(self) -> List[int]:
        return [u for _, ids in self.units for u in ids]

    def init(self, seed: int, **cfg_kw):
        for eng, pb, (_, ids) in zip(self.engines, self.pbs, self.units):
            cfg = eng.default_cfg(**cfg_kw)
            rep = lambda v: np.repeat(np.asarray(v)[None], self.chains, axis=0)
            eng.sampler_init(cfg, rep(pb["Xhat"]), rep(pb["sig_pre0"]), rep(pb["th_pre0"]), seed=seed, chain_ids=ids)
        self._captured = False

    def run(self, n_steps: int) -> int:
        """n_steps transitions of every chain of every owned dataset; returns the leapfrogs taken.  The handles are independent: each is
        driven from its own host thread (ctypes releases the GIL), so a rank that owns two small datasets keeps both on the GPU at once."""
        lf = [0] * len(self.engines)
        err: Dict[int, BaseException] = {}

        def work(k):
            try:
                lf[k] = self.engines[k].sampler_run(n_steps)[0]
            except BaseException as e:          # noqa: BLE001 (re-raised on the caller's thread)
                err[k] = e

        if len(self.engines) == 1 or not self._captured:
            # (the first run after an initialisation captures each handle's leapfrog graph: a stream capture in one thread makes HIP refuse the
            #  blocking copies of the other threads -- "would make the legacy stream depend on a capturing stream" -- so that run is sequential)
            for k in range(len(self.engines)):
                work(k)
            self._captured = True
        else:
            ths = [threading.Thread(target=work, args=(k,)) for k in range(len(self.engines))]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        if err:
            raise next(iter(err.values()))
        return int(sum(lf))

    def samples(self) -> Tuple[np.ndarray, List[int]]:
        """[units of this rank, results, N*D + D + P] (X flattened, sigma_pre, theta_pre) and their global unit ids."""
        blocks = []
        for eng in self.engines:
            Xs, sp, tp = eng.sampler_samples()
            blocks.append(np.concatenate([Xs.reshape(Xs.shape[0], Xs.shape[1], -1), sp, tp], axis=2))
        tail = blocks[0].shape[1:] if blocks else (0, 0)
        flat = np.concatenate(blocks, axis=0) if blocks else np.zeros((0,) + tuple(tail))
        return flat, self.unit_ids

    def gather(self, dst: int = 0):
        flat, ids = self.samples()
        return gather_samples(flat, ids, dst=dst)

    def close(self):
        for eng in self.engines:
            eng.close()
        self.engines = []
