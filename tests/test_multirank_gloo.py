"""CPU suite, part 3: the N > 1 host path with world_size = 2 over gloo.  The GPU compute is stood in by
the oracle (allowed in tests); what is exercised is the slab partition, the one broadcast of the
read-only state from the factorising rank, and the gather of the disjoint output slabs."""
import importlib
import os
import sys

import numpy as np
import pytest

from conftest import ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import gp_oracle as orc
    sh = importlib.import_module("gaussian-object-modelling_amd.sharding")
    ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        n, g = 80, 7
        nq = g ** 3
        # rank 0 owns the model inputs; the others receive the state by broadcast
        state = torch.zeros(5 * n, dtype=torch.float64)
        if rank == 0:
            state[:] = torch.from_numpy(np.concatenate(ds.fibonacci_training_set(n)))
        sh.broadcast_state(dist, [state], src=0)
        x, y, z, lab, s2 = (state[i * n:(i + 1) * n].numpy() for i in range(5))
        model = orc.Model(orc.make_kernel("matern52", 1, 1), x, y, z, lab, s2)
        qx, qy, qz = ds.query_grid(g)
        lo, hi = sh.slab_range(nq, rank, world)
        out = model.evaluate(qx[lo:hi], qy[lo:hi], qz[lo:hi], want_v=True)
        f = sh.gather_slabs(dist, torch, torch.from_numpy(out["f"]), nq, rank, world)
        v = sh.gather_slabs(dist, torch, torch.from_numpy(out["v"]), nq, rank, world)
        # max-over-ranks timing reduction as bench.py does it
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            ref = model.evaluate(qx, qy, qz, want_v=True)
            q.put((float(np.max(np.abs(f.numpy() - ref["f"]))), float(np.max(np.abs(v.numpy() - ref["v"]))),
                   float(t.item())))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_grid_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ef, ev, tmax = res
    assert ef == 0.0 and ev == 0.0  # same arithmetic, disjoint slabs: bit-identical
    assert tmax == 2.0
