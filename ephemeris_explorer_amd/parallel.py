"""One process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" on CPU for the tests).

Three ways the path uses more than one GPU:
  * replicas of a massive-body system (ensembles, forward/backward): no data-path collective, only the timing is
    reduced (max over ranks);
  * the massless sweep: spacecraft are independent given the ephemeris -> `shard_range` over craft;
  * ONE massive-body system partitioned by target body (`shard_nbody`): one all-gather of the packed positions per
    force evaluation, RCCL over xGMI inside the library (`eph_nbody_shard`), or a caller-supplied exchange
    (`host_staged_exchange`: through host memory and any torch.distributed backend -- what the tests use to run
    two ranks on one GPU).
"""
import ctypes as C
import os


def env_rank():
    """(rank, local_rank, world_size) from the torch.distributed.run environment; (0, 0, 1) when run directly."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(n_items, rank, world):
    """Contiguous block partition of n_items independent work items; blocks differ by at most one item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def replica_seed(base_seed, rank):
    return base_seed + rank


def reduce_timing(elapsed_s, units_local, dist=None, device="cpu"):
    """Whole-job throughput = (sum of units over ranks) / (max elapsed over ranks). `dist` is torch.distributed
    (initialised) or None for a single process. Returns (total_units, max_elapsed)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(units_local), float(elapsed_s)
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units_local)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()), float(t.item())


def gather_craft_states(state, n_total, dist=None, device="cpu"):
    """The result exchange of the massless sweep (`ncclAllGather` of the final states): every rank holds the
    final (t, position, velocity) of its contiguous block of spacecraft (`shard_range`); returns the [n_total, 7] array of
    all of them on every rank. One all-gather of equal, zero-padded slices -- RCCL over xGMI with backend "nccl" and
    device="cuda" (56 B per craft: 56 MB for 1e6), gloo on CPU in the tests. `state` = SpacecraftBatch.state()."""
    import numpy as np
    import torch
    if isinstance(state, np.ndarray) and state.dtype.names and state.dtype.itemsize == 80:
        # SpacecraftBatch.summary(): t, pos, vel are the first seven doubles of every 80-byte record -- a view, no copy
        mine = state.view(np.float64).reshape(-1, 10)[:, :7]
    else:                                                          # the dict of arrays from SpacecraftBatch.state()
        mine = np.concatenate([np.asarray(state["t"])[:, None], np.asarray(state["pos"]), np.asarray(state["vel"])], axis=1)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        assert len(mine) == n_total
        return mine
    rank, world = dist.get_rank(), dist.get_world_size()
    width = -(-n_total // world)                                  # largest block of shard_range
    lo, hi = shard_range(n_total, rank, world)
    assert hi - lo == len(mine)
    send = torch.zeros((width, 7), dtype=torch.float64, device=device)
    send[: hi - lo] = torch.from_numpy(np.ascontiguousarray(mine)).to(device)
    recv = torch.empty((world * width, 7), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(recv, send)
    recv = recv.cpu().numpy().reshape(world, width, 7)
    return np.concatenate([recv[r, : shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0]]
                           for r in range(world)], axis=0)


def _hip():
    from . import hip_runtime
    return hip_runtime()          # the runtime the library itself uses (a torch process carries a second one)


def host_staged_exchange(dist, group=None):
    """An eph_exchange_fn body: in-place all-gather of equal slices through host memory and torch.distributed
    (any backend, CPU tensors). Slow by construction (two PCIe copies and a host sync per force evaluation): it is
    the portable fallback and the test transport; production runs pass an RCCL unique id instead."""
    import numpy as np
    import torch
    hip = _hip()
    D2H, H2D = 2, 1

    def exchange(dev_ptr, slice_bytes, rank, world, stream):
        if hip.hipStreamSynchronize(stream):
            return 2
        mine = np.empty(slice_bytes, dtype=np.uint8)
        if hip.hipMemcpy(mine.ctypes.data, dev_ptr + rank * slice_bytes, slice_bytes, D2H):
            return 3
        parts = [torch.empty(slice_bytes, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(mine), group=group)
        for r, part in enumerate(parts):
            if r == rank:
                continue
            a = part.numpy()
            if hip.hipMemcpy(dev_ptr + r * slice_bytes, a.ctypes.data, slice_bytes, H2D):
                return 4
        return 0

    return exchange


def broadcast_unique_id(dist, make_id, device="cpu", group=None):
    """Rank 0 makes the 128-byte RCCL id (make_id()), every rank returns it."""
    import torch
    buf = torch.zeros(128, dtype=torch.uint8, device=device)
    if dist.get_rank() == 0:
        buf = torch.tensor(list(make_id()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=0, group=group)
    return bytes(buf.cpu().tolist())


def broadcast_ephemeris(build, dist=None, device="cpu", group=None, from_image=None):
    """rank 0 builds the massive bodies' table ONCE (build() -> Ephemeris), exports it as one contiguous image
    (eph_ephemeris_export) and broadcasts it; every other rank imports it (eph_ephemeris_import) instead of integrating the bodies
    again. `device`: where the broadcast buffers live ("cuda" with the nccl = RCCL backend: over xGMI; "cpu" with gloo).
    Returns (ephemeris, {"build_s", "export_s", "broadcast_s", "import_s", "bytes"}); the table is bit-identical on every rank
    (tests/test_gpu_bench.py compares the images). `from_image`: Ephemeris.from_image unless given (the CPU test of this plumbing
    passes a stand-in)."""
    import time

    import numpy as np
    if from_image is None:
        from . import Ephemeris
        from_image = Ephemeris.from_image
    t = {"build_s": 0.0, "export_s": 0.0, "broadcast_s": 0.0, "import_s": 0.0, "bytes": 0}
    world = 1 if dist is None or not dist.is_initialized() else dist.get_world_size(group)
    rank = 0 if world == 1 else dist.get_rank(group)
    eph, image = None, None
    if rank == 0:
        t0 = time.perf_counter()
        eph = build()
        t["build_s"] = time.perf_counter() - t0
        if world > 1:
            t0 = time.perf_counter()
            image = eph.export_image()
            t["export_s"] = time.perf_counter() - t0
            t["bytes"] = int(image.size)
    if world > 1:
        import torch
        t0 = time.perf_counter()
        size = torch.tensor([image.size if rank == 0 else 0], dtype=torch.int64, device=device)
        dist.broadcast(size, src=0, group=group)
        n = int(size.item())
        buf = torch.from_numpy(image).to(device) if rank == 0 else torch.empty(n, dtype=torch.uint8, device=device)
        dist.broadcast(buf, src=0, group=group)
        if device != "cpu":
            torch.cuda.synchronize()
        t["broadcast_s"] = time.perf_counter() - t0
        t["bytes"] = n
        if rank != 0:
            t0 = time.perf_counter()
            eph = from_image(np.ascontiguousarray(buf.cpu().numpy()))
            t["import_s"] = time.perf_counter() - t0
    return eph, t


def peer_transport(dist, slot_bytes=1 << 22, group=None):
    """A connected PeerTransport over the ranks of the process group: the 64-byte hipIpc handles travel through
    torch.distributed (all_gather_object: any backend), the data never does."""
    from . import PeerTransport
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    errors = []
    for memory in ("auto", "coarse"):                # a second round in plain device memory if ANY rank could not map a mailbox
        err, t = None, None
        try:
            t = PeerTransport(rank, world, slot_bytes, memory=memory)
        except Exception as e:                       # (the collective below must still be entered by every rank)
            err = f"rank {rank} create[{memory}]: {e}"
        handles = [None] * world
        dist.all_gather_object(handles, t.handle if t is not None else None, group=group)
        if err is None and all(h is not None for h in handles):
            try:
                t.connect(handles)
            except Exception as e:
                err = f"rank {rank} connect[{memory}]: {e}"
        outcome = [None] * world
        dist.all_gather_object(outcome, (err, t.memory if t is not None else None), group=group)
        if all(o[0] is None for o in outcome):       # every rank has mapped every mailbox before the first push
            t.forms = [o[1] for o in outcome]
            t.attempts = errors
            return t
        errors += [o[0] for o in outcome if o[0] is not None]
        del t
    raise RuntimeError("peer transport could not be set up: " + " | ".join(errors))


def shard_nbody(integration, dist=None, transport="rccl", device="cpu"):
    """Partition `integration` (an NBodyIntegration every rank created identically) over the ranks of the
    initialised process group. transport "rccl": RCCL inside the library; "peer": direct writes into hipIpc-mapped
    mailboxes (csrc/peer.hip); "host": host_staged_exchange."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return integration
    from . import rccl_unique_id
    rank, world = dist.get_rank(), dist.get_world_size()
    if transport == "rccl":
        uid = broadcast_unique_id(dist, rccl_unique_id, device=device)
        integration.shard(rank, world, unique_id=uid)
    elif transport == "peer":
        integration.shard_peer(peer_transport(dist))
    elif transport == "host":
        integration.shard(rank, world, exchange=host_staged_exchange(dist))
    else:
        raise ValueError(transport)
    return integration
