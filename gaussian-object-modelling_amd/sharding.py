"""Host logic of the multi-GPU path (one process per GPU, torch.distributed; backend "nccl" = RCCL over
xGMI on the GPU node, "gloo" in the CPU tests).

Two partitions:
  * independent models: rank r owns model r and its own queries -- no data-path collective at all;
  * one model, sharded query grid: contiguous slabs of the query list per rank; ONE exchange step, the
    broadcast of the factorising rank's read-only state (points, alpha, 1/D [, inverse factor]); outputs
    are disjoint slabs, gathered only if the caller wants them in one place.
"""


def slab_range(nq, rank, world):
    """Contiguous slab [lo, hi) of nq queries for `rank` (remainder spread over the low ranks)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(int(nq), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def broadcast_state(dist, buffers, src=0):
    """Broadcast every tensor of `buffers` from `src` in place (the factor / state blob parts)."""
    for b in buffers:
        dist.broadcast(b, src=src)


def device_blob_as_tensor(torch, ptr, nbytes, device):
    """Zero-copy uint8 view of a device allocation owned by libgpx (gpx_model_state_blob)."""

    class _Blob:
        pass

    b = _Blob()
    b.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                  "version": 3, "strides": None}
    return torch.as_tensor(b, device=device)


def gather_slabs(dist, torch, local, nq, rank, world, dst=0):
    """Gather the per-rank output slabs (1-D tensors) on `dst` in query order; returns None elsewhere."""
    sizes = [slab_range(nq, r, world)[1] - slab_range(nq, r, world)[0] for r in range(world)]
    width = max(sizes)
    pad = torch.zeros(width, dtype=local.dtype, device=local.device)
    pad[: local.numel()] = local
    bufs = [torch.zeros_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, gather_list=bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([bufs[r][: sizes[r]] for r in range(world)])


VAR_NCORR = 14  # row-correction vectors of the variance contraction (csrc/gpx_internal.hpp)
BLOB_META = 8   # doubles at the end of part 0: centre x y z, 1 / (sx sk) of the split operands, reserved


def state_blob_layout(npad, esz):
    """Byte layout of state blob part 0 of a model (gpx_model_state_blob(m, 0), csrc/gpx_model.hpp): name -> (offset,
    count, bytes per element).  fp64 x y z alpha (internal = pivot order, zero padded to npad), fp64 1/D ("dinv64"),
    the 14 fp64 row-correction vectors X b_c of the variance contraction ("corr"), then in the working type
    (esz = 4 or 8) the points relative to the model's centre ("tx", "ty", "tz") and 1/D ("dinv"), then the meta block
    (8 doubles: centre x y z, 1 / (sx sk), reserved).  Part 1 is the npad x npad inverse factor X = L^-1 (row-major,
    working type)."""
    out, off = {}, 0
    for name in ("x", "y", "z", "alpha", "dinv64"):
        out[name] = (off, npad, 8)
        off += 8 * npad
    out["corr"] = (off, VAR_NCORR * npad, 8)
    off += 8 * VAR_NCORR * npad
    for name in ("tx", "ty", "tz", "dinv"):
        out[name] = (off, npad, esz)
        off += esz * npad
    out["meta"] = (off, BLOB_META, 8)
    off += 8 * BLOB_META
    out["bytes"] = off
    return out


def shard_record(state, per_rank):
    """The `shard` record of bench.py --mode shard from the per-rank phase times of one step.
    per_rank: list over ranks of (t_model, t_exchange, t_commit, t_predict) in seconds, where t_model is the time to have a
    local model object (rank 0: train; other ranks: train when state == "recompute", else allocate a shell), t_exchange
    the time inside the broadcast of the two state blobs (0 for "recompute": no communication at all), t_commit the
    commit of the received state, t_predict the evaluation of the rank's slab.
    Returns ms figures: the training rank's train time, the broadcast as the SOURCE sees it (the other ranks are already
    waiting in it, so this is the transfer), the slowest commit, and the longest time a rank spent waiting for the
    factorising rank (its time in the exchange minus the transfer itself)."""
    if state not in ("broadcast", "recompute"):
        raise ValueError("state must be 'broadcast' or 'recompute'")
    ms = lambda t: 1e3 * float(t)
    t_train = per_rank[0][0]
    t_bcast = per_rank[0][1] if state == "broadcast" else 0.0
    others = per_rank[1:]
    idle = max([max(0.0, r[1] - t_bcast) for r in others], default=0.0) if state == "broadcast" else 0.0
    return {"state": state, "t_train_ms": ms(t_train), "t_bcast_ms": ms(t_bcast),
            "t_commit_ms": ms(max([r[2] for r in others], default=0.0)),
            "t_train_other_ranks_ms": ms(max([r[0] for r in others], default=0.0)) if state == "recompute" else 0.0,
            "idle_max_ms": ms(idle), "t_predict_max_ms": ms(max(r[3] for r in per_rank)),
            "t_step_max_ms": ms(max(sum(r) for r in per_rank))}
