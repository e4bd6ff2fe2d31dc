"""world_size-2 gloo test of the N>1 path: chain ids per rank + the single final gather."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, chains_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from magi_v2_amd.shard import chain_ids_for_rank, gather_samples
    ids = chain_ids_for_rank(rank, world, chains_total)
    # a chain's "samples" depend only on its global id, as the Philox keying guarantees on the GPU
    local = np.stack([np.full((3, 5), float(i)) + np.arange(5) for i in ids]) if ids else np.zeros((0, 3, 5))
    out, gid = gather_samples(local, ids, dst=0)
    if rank == 0:
        q.put((out, gid))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_orders_chains_by_global_id_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    out, gid = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert list(gid) == [0, 1, 2, 3, 4]                  # 3 chains on rank 0, 2 on rank 1 (uneven)
    assert out.shape == (5, 3, 5)
    for i in range(5):
        np.testing.assert_array_equal(out[i], np.full((3, 5), float(i)) + np.arange(5))


def test_gather_single_process_passthrough():
    from magi_v2_amd.shard import gather_samples
    out, gid = gather_samples(np.arange(6.0).reshape(3, 2), [2, 0, 1])
    assert list(gid) == [0, 1, 2]
    np.testing.assert_array_equal(out, np.array([[2.0, 3.0], [4.0, 5.0], [0.0, 1.0]]))


def _sweep_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from magi_v2_amd.shard import gather_samples, shard_units
    units = shard_units(10, 8, rank, world)                      # BASELINE config 4: whole datasets per rank
    ids = [u for _, us in units for u in us]
    local = np.stack([np.full((2, 4), float(u)) for u in ids])   # a unit's block depends only on its global id
    out, gid = gather_samples(local, ids, dst=0)
    if rank == 0:
        q.put((out, gid, [ds for ds, _ in units]))
    dist.barrier()
    dist.destroy_process_group()


def test_alpha_sweep_units_arrive_once_and_in_order_world2():
    """Config 4's placement (shard_units: datasets round-robin over the ranks, a dataset's 8 chains together) + the one gather: rank 0 holds
    datasets 0, 2, 4, 6, 8 -- unit ids 0-7, 16-23, .. -- rank 1 the others; rank 0 receives all 80 units once, sorted by global id."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sweep_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out, gid, ds0 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ds0 == [0, 2, 4, 6, 8] and list(gid) == list(range(80)) and out.shape == (80, 2, 4)
    np.testing.assert_array_equal(out[:, 0, 0], np.arange(80.0))


def test_alpha_sweep_datasets_are_the_ten_vignette_thinnings():
    from magi_v2_amd.sweep import alpha_sweep_datasets
    ds = alpha_sweep_datasets(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seir_alpha_sweep.npz"))
    assert [n for n, _ in ds] == [f"alpha={a}_seed={s}" for a in ("0.05", "0.15") for s in range(5)]
    for _, pb in ds:
        assert pb["I"].shape == (161,) and pb["Xhat"].shape == (161, 4) and list(pb["N_ds"]) == [81.0] * 4 and np.isfinite(pb["Xhat"]).all()
