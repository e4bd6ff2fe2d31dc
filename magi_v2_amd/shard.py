"""Chain / dataset sharding over the GPUs of one node and the final sample gather.

The reference runs one chain in one process (magi_v2.py:383-395); chains and datasets are
independent, so the path shards with no data-path collective.  A unit is
(dataset, chain).  Philox streams are keyed by the GLOBAL chain id, so results do not depend on
how units are placed.  The only collective is one gather of the post-burn-in samples to rank 0
(RCCL when the tensors live on the GPU; gloo in the CPU tests)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def chain_ids_for_rank(rank: int, world: int, chains_total: int) -> List[int]:
    """Contiguous block partition of global chain ids 0..chains_total-1 (BASELINE config 3:
    64 chains over 8 GPUs -> 8 per GPU); the first ``chains_total % world`` ranks get one extra."""
    base, extra = divmod(chains_total, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def family_chains_for(chains_total: int, world: int) -> int:
    """The largest per-rank share of ``chains_total`` chains over ``world`` ranks.  Every rank passes it to its handle
    (``set_option("family_chains", ...)`` / ``predict(family_chains=...)``): the streaming-kernel family is then chosen for that count on
    all ranks, so an uneven shard (20 chains over 8 GPUs: 3, 3, 3, 3, 2, 2, 2, 2) does not put some chains on the matrix-core kernel and
    others on the VALU kernel -- a chain's samples are bit-identical wherever it runs (the families agree to rounding only)."""
    return -(-int(chains_total) // max(1, int(world)))


def shard_units(n_datasets: int, chains_per_dataset: int, rank: int, world: int) -> List[Tuple[int, List[int]]]:
    """BASELINE config 4 (alpha sweep: 10 datasets x 8 chains): whole datasets are dealt
    round-robin to ranks so that the chains of one dataset share a GPU and its matrices.
    Returns [(dataset index, [global chain ids]), ...] for this rank."""
    out = []
    for ds in range(rank, n_datasets, world):
        out.append((ds, [ds * chains_per_dataset + c for c in range(chains_per_dataset)]))
    return out


def gather_samples(local: np.ndarray, chain_ids: Sequence[int], dst: int = 0, device=None):
    """Gather per-chain sample blocks [n_local_chains, ...] from every rank to ``dst`` with ONE
    collective; returns (samples ordered by global chain id, ids) on dst and (None, None) elsewhere.
    Ranks may hold different numbers of chains (blocks are padded to the maximum)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        order = np.argsort(np.asarray(chain_ids))
        return np.asarray(local)[order], np.asarray(chain_ids)[order]
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    n_local = torch.tensor([len(chain_ids)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    tail = tuple(local.shape[1:])
    block = np.zeros((nmax,) + tail + (), dtype=np.float64)
    block[: len(chain_ids)] = local
    ids = np.full((nmax,), -1, dtype=np.int64)
    ids[: len(chain_ids)] = np.asarray(chain_ids, dtype=np.int64)
    # ids ride along as one extra fp64 "column" so a single gather moves everything
    payload = torch.from_numpy(np.concatenate([block.reshape(nmax, -1), ids[:, None].astype(np.float64)], axis=1)).to(dev)
    bufs = [torch.empty_like(payload) for _ in range(world)] if rank == dst else None
    dist.gather(payload, bufs, dst=dst)
    if rank != dst:
        return None, None
    allp = torch.cat(bufs, dim=0).cpu().numpy()
    gid = allp[:, -1].astype(np.int64)
    keep = gid >= 0
    allp, gid = allp[keep], gid[keep]
    order = np.argsort(gid)
    return allp[order, :-1].reshape((len(order),) + tail), gid[order]
