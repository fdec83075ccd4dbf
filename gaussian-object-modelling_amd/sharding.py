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


# ------------------------------------------------------------------ BASELINE configs 4 and 5 in their multi-GPU form
# The two drivers below hold ALL of the rank logic (who trains, what travels, which x-planes / which objects a rank
# owns, how the phase times are reduced); the compute sits behind a small duck-typed backend, so that bench.py runs
# them on libgpx models over RCCL and tests/test_multirank_gloo.py runs the SAME functions on the CPU oracle over gloo.

def grid_x_slab(g, rank, world):
    """Rank's share of the g^3 lattice (x slowest, z fastest: src/gp_node.cpp:1025-1036) as whole x-planes:
    (x_lo, x_hi, idx_lo, idx_hi) -- planes [x_lo, x_hi), i.e. lattice indices [x_lo g^2, x_hi g^2).  Remainder planes
    go to the low ranks; a rank beyond the g-th plane gets an empty slab."""
    x_lo, x_hi = slab_range(g, rank, world)
    return x_lo, x_hi, x_lo * g * g, x_hi * g * g


def objects_of_rank(n_objects, rank, world):
    """Independent models (BASELINE config 5): object o is trained and evaluated by rank o mod world -- one object per
    rank when world == n_objects, round-robin queues below that, idle ranks above it."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, int(n_objects), world))


def _gather_rows(dist, torch, row, world, device):
    mine = torch.tensor([float(v) for v in row], dtype=torch.float64, device=device)
    if dist is None or world == 1:
        return [mine.tolist()]
    rows = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(rows, mine)
    return [r.tolist() for r in rows]


def _fence(dist, be):
    be.sync()
    if dist is not None:
        dist.barrier()
    be.sync()


def sharded_grid_step(dist, torch, rank, world, g, state, be, device, clock):
    """ONE model, its g^3 query lattice cut into x-slabs (BASELINE config 4).  state == "broadcast": rank 0
    trains, the others allocate a shell, every state blob goes through ONE dist.broadcast, the receivers commit.
    state == "recompute": every rank trains the same model itself -- no collective on the data path.
    Backend: be.train() / be.shell() -> model; be.blobs(m) -> list of uint8 tensors (in place views of the state);
    be.commit(m); be.predict(m, g, x_lo, x_hi) -> (n_queries, sum f, sum v, min v, max v) of the rank's planes;
    be.close(m); be.sync().  Returns the record of the step (same on every rank)."""
    if state not in ("broadcast", "recompute"):
        raise ValueError("state must be 'broadcast' or 'recompute'")
    x_lo, x_hi, lo, hi = grid_x_slab(g, rank, world)
    _fence(dist, be)
    t0 = clock()
    if state == "broadcast":
        m = be.train() if rank == 0 else be.shell()
        be.sync()
        t1 = clock()
        bufs = be.blobs(m)
        if dist is not None:
            broadcast_state(dist, bufs, src=0)
        be.sync()
        t2 = clock()
        if rank != 0:
            be.commit(m)
            be.sync()
        t3 = clock()
        nbytes = int(sum(int(b.numel()) * int(b.element_size()) for b in bufs))
    else:
        m = be.train()
        be.sync()
        t1 = t2 = t3 = clock()
        nbytes = 0
    nq, sf, sv, vmin, vmax = be.predict(m, g, x_lo, x_hi) if x_hi > x_lo else (0, 0.0, 0.0, float("inf"), float("-inf"))
    be.sync()
    t4 = clock()
    be.close(m)
    _fence(dist, be)
    t5 = clock()
    rows = _gather_rows(dist, torch, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t0, x_lo, x_hi, nq, sf, sv,
                                      min(vmin, 1e300), max(vmax, -1e300)), world, device)
    rec = shard_record(state, [tuple(r[:4]) for r in rows])
    wall = max(r[4] for r in rows)
    rec.update({"grid": g, "n_query": g ** 3, "world": world, "state_bytes": nbytes, "ms_per_step": 1e3 * wall,
                "value": g ** 3 / wall, "unit": "query-points/s (whole grid / max-over-ranks wall time, barrier to barrier)",
                "slabs": [{"rank": r, "x_planes": [int(rows[r][5]), int(rows[r][6])], "n_query": int(rows[r][7]),
                           "t_predict_ms": 1e3 * rows[r][3]} for r in range(world)],
                "sum_f": sum(r[8] for r in rows), "sum_v": sum(r[9] for r in rows),
                "v_min": min(r[10] for r in rows), "v_max": max(r[11] for r in rows)})
    if sum(s["n_query"] for s in rec["slabs"]) != g ** 3:
        raise RuntimeError("slabs do not tile the lattice")
    return rec


def objects_per_rank_step(dist, torch, rank, world, n_objects, run_object, be, device, clock):
    """Independent models, one per GPU (BASELINE config 5): rank r trains and evaluates the objects of
    objects_of_rank(n_objects, r, world), one after the other; no data-path collective.  run_object(o) -> (n_train,
    n_query, sum f, sum v) after the object's results are complete.  Returns the record (same on every rank): wall time
    = barrier to barrier, value = all queries of all objects / that time."""
    mine = objects_of_rank(n_objects, rank, world)
    _fence(dist, be)
    t0 = clock()
    done = []
    for o in mine:
        ta = clock()
        n_train, nq, sf, sv = run_object(o)
        done.append((o, n_train, nq, sf, sv, clock() - ta))
    be.sync()
    t_mine = clock() - t0
    _fence(dist, be)
    wall = clock() - t0
    # fixed-width rows: per object slot (o, n_train, nq, sum f, sum v, seconds), padded with -1
    slots = (n_objects + world - 1) // world
    row = [t_mine, wall]
    for s in range(slots):
        row += list(done[s]) if s < len(done) else [-1, 0, 0, 0.0, 0.0, 0.0]
    rows = _gather_rows(dist, torch, row, world, device)
    objs = {}
    for r, rw in enumerate(rows):
        for s in range(slots):
            o, n_train, nq, sf, sv, sec = rw[2 + 6 * s: 8 + 6 * s]
            if o >= 0:
                if int(o) in objs:
                    raise RuntimeError("object %d ran on two ranks" % int(o))
                objs[int(o)] = {"rank": r, "n_train": int(n_train), "n_query": int(nq), "sum_f": sf, "sum_v": sv, "ms": 1e3 * sec}
    if sorted(objs) != list(range(n_objects)):
        raise RuntimeError("objects %s were not all run" % sorted(objs))
    wall = max(rw[1] for rw in rows)
    total_q = sum(o["n_query"] for o in objs.values())
    return {"world": world, "n_objects": n_objects, "ms_per_step": 1e3 * wall, "value": total_q / wall,
            "unit": "query-points/s (all objects / max-over-ranks wall time, barrier to barrier)", "n_query": total_q,
            "t_rank_ms": [1e3 * rw[0] for rw in rows], "objects": [dict(objs[o], object=o) for o in range(n_objects)]}
