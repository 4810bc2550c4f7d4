"""One massive-body system partitioned by target body over ranks (eph_nbody_shard): every rank's
results must be bit-identical to the single-device run, because each target's all-pairs sum keeps the order of
NewtonianGravity::eval (ephemeris/src/propagators/nbody.rs:22-38) whatever rank evaluates it.

Only one GPU is available to the tests, so the two-rank cases run two processes on that GPU with the host-staged
exchange (gloo); the RCCL transport is exercised on a one-rank communicator."""
import os
import socket

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
H = 1.0 / 1024.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _single(n, method, steps):
    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd.workloads import plummer
    pos, vel, mu = plummer(n)
    nb = ea.NBodyIntegration(pos, vel, mu, 0.0, H, method)
    nb.advance(steps)
    return nb.state(), nb.acc()


def test_rccl_one_rank_communicator(gpu):
    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd.workloads import plummer
    pos, vel, mu = plummer(1024)
    nb = ea.NBodyIntegration(pos, vel, mu, 0.0, H).shard(0, 1, unique_id=ea.rccl_unique_id())
    nb.advance(12 + 30)
    (p, v, t, sc), a = nb.state(), nb.acc()
    lo, hi, gathers = nb.shard_info()
    assert (lo, hi) == (0, 1024)
    assert gathers >= 12 * 4 * 7 + 30                # one per published position set
    (p0, v0, t0, sc0), a0 = _single(1024, "QuinlanTremaine12", 12 + 30)
    assert t == t0 and sc == sc0
    assert np.array_equal(p, p0) and np.array_equal(v, v0) and np.array_equal(a, a0)


def test_shard_argument_errors(gpu):
    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd.workloads import plummer
    pos, vel, mu = plummer(32)
    with pytest.raises(ea.EphemerisError):           # one-workgroup systems: replicas only
        ea.NBodyIntegration(pos, vel, mu, 0.0, H).shard(0, 2, exchange=lambda *a: 0)
    pos, vel, mu = plummer(192)                      # padded to 192: not a multiple of 64 * 2
    with pytest.raises(ea.EphemerisError):
        ea.NBodyIntegration(pos, vel, mu, 0.0, H).shard(0, 2, exchange=lambda *a: 0)
    pos, vel, mu = plummer(128)
    nb = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
    with pytest.raises(ea.EphemerisError):           # world > 1 needs a transport
        nb.shard(0, 2)
    with pytest.raises(ea.EphemerisError):
        nb.shard(2, 2, exchange=lambda *a: 0)
    nb.shard(1, 2, exchange=lambda *a: 0)
    assert nb.shard_info()[:2] == (64, 128)
    with pytest.raises(ea.EphemerisError):           # a handle is sharded once
        nb.shard(1, 2, exchange=lambda *a: 0)


def test_failing_exchange_is_reported(gpu):
    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd.workloads import plummer
    pos, vel, mu = plummer(128)
    nb = ea.NBodyIntegration(pos, vel, mu, 0.0, H).shard(0, 2, exchange=lambda *a: 7)
    with pytest.raises(ea.EphemerisError) as e:
        nb.advance(1)
    assert e.value.status == -6 and "7" in str(e.value)


def _worker(rank, world, port, n, method, steps, out, transport="host"):
    import sys
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd import parallel
    from ephemeris_explorer_amd.workloads import plummer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pos, vel, mu = plummer(n)
    nb = ea.NBodyIntegration(pos, vel, mu, 0.0, H, method)
    parallel.shard_nbody(nb, dist, transport=transport)
    half = steps // 2
    nb.advance(half)
    twin = nb.clone()                                 # collective: every rank clones, the clones share the ranks
    nb.advance(steps - half)
    twin.advance(steps - half)
    p, v, t, sc = nb.state()
    a = nb.acc()
    tp = twin.state()[0]
    lo, hi, gathers = nb.shard_info()
    out[rank] = (p, v, a, t, sc, lo, hi, gathers, np.array_equal(tp, p))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,world,method,steps,transport", [
    (1024, 2, "QuinlanTremaine12", 12 + 40, "host"),  # wave kernel, start-up and steady state sharded
    (1000, 2, "QuinlanTremaine12", 12 + 9, "host"),   # ragged: the last rank owns 488 bodies
    (512, 4, "BlanesMoan6B", 5, "host"),              # SRKN only
    # the REAL device-to-device transport between processes: hipIpc-mapped mailboxes, direct writes, flags (peer.hip)
    (1024, 2, "QuinlanTremaine12", 12 + 40, "peer"),
    (1000, 2, "QuinlanTremaine12", 12 + 9, "peer"),
    (4096, 4, "QuinlanTremaine12", 12 + 20, "peer"),  # the metric's system: 1024 targets per rank, workgroup kernel
    (512, 4, "BlanesMoan6B", 5, "peer"),
])
def test_ranks_on_one_gpu_match_single_device(gpu, n, world, method, steps, transport, monkeypatch):
    import torch.multiprocessing as mp
    monkeypatch.setenv("EPH_PEER_TIMEOUT_MS", "5000")   # a rank that never delivers costs seconds, not the GPU
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, method, steps, out, transport), nprocs=world, join=True)
    (p0, v0, t0, sc0), a0 = _single(n, method, steps)
    assert set(out.keys()) == set(range(world))
    npad = (n + 63) // 64 * 64
    for r in range(world):
        p, v, a, t, sc, lo, hi, gathers, twin_ok = out[r]
        assert (lo, hi) == (r * npad // world, min(n, (r + 1) * npad // world))
        assert t == t0 and sc == sc0 and gathers > 0 and twin_ok
        assert np.array_equal(p, p0) and np.array_equal(v, v0) and np.array_equal(a, a0), (r, n, method)


def _prop_worker(rank, world, port, n, out, transport="host"):
    import sys
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd import parallel
    from ephemeris_explorer_amd.workloads import plummer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pos, vel, mu = plummer(n)
    count = (np.arange(n) % 3 + 1).astype(np.uint32)       # ragged sampling periods: the ranks own unequal window counts
    degree = (np.arange(n) % 3 + 5).astype(np.uint32)
    p = ea.NBodyPropagator(pos, vel, mu, 0.0, H, ea.FORWARD, count, degree)
    if transport == "peer":
        p.shard_peer(parallel.peer_transport(dist, slot_bytes=4096))   # small slots: the record gather goes in rounds
    else:
        parallel.shard_nbody(p, dist, transport=transport)
    sol = p.propagate(100 * H)
    rows = [(sol.info(b), sol.coeffs(b)) for b in (0, 1, n // 2 - 1, n // 2, n - 2, n - 1)]
    out[rank] = (p.time(), rows)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["host", "peer"])
def test_sharded_propagator_builds_the_same_ephemeris(gpu, transport, monkeypatch):
    """eph_prop_shard: two ranks on one GPU (host-staged exchange) each sample and fit the bodies they own, the
    polynomials are all-gathered: every rank ends with the single-device Vec<UniformSpline>, bit for bit."""
    import torch.multiprocessing as mp
    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd.workloads import plummer
    n, world = 256, 2
    monkeypatch.setenv("EPH_PEER_TIMEOUT_MS", "5000")
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_prop_worker, args=(world, _free_port(), n, out, transport), nprocs=world, join=True)
    pos, vel, mu = plummer(n)
    count = (np.arange(n) % 3 + 1).astype(np.uint32)
    degree = (np.arange(n) % 3 + 5).astype(np.uint32)
    p = ea.NBodyPropagator(pos, vel, mu, 0.0, H, ea.FORWARD, count, degree)
    sol = p.propagate(100 * H)
    want = [(sol.info(b), sol.coeffs(b)) for b in (0, 1, n // 2 - 1, n // 2, n - 2, n - 1)]
    for r in range(world):
        t, rows = out[r]
        assert t == p.time()
        for (info, (co, nc)), (winfo, (wco, wnc)) in zip(rows, want):
            assert info == winfo and info[2] > 0
            assert np.array_equal(nc, wnc) and np.array_equal(co, wco)


def test_two_ranks_in_one_process_workgroup_kernel_at_an_offset(gpu):
    """2048 and 4096 targets per rank of 4096- and 8192-body systems: the role-specialised workgroup kernel (default layout: 152 KB of
    LDS per workgroup) evaluating a target range that does not start at body 0. Two ranks as two THREADS of this process, each
    with its own handle and stream; the exchange callback is a device-to-device copy of the peer's slice between two thread
    barriers. (Two PROCESSES sharing one GPU thrash on this kernel -- every alternation swaps 152 KB of LDS per CU -- which
    says nothing about one process per GPU.)"""
    import threading
    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd.workloads import plummer
    hip = ea.hip_runtime()        # the library's own HIP runtime (earlier tests of this module import torch, which has another)
    for n, steps in ((4096, 12 + 5), (8192, 12 + 2)):
        pos, vel, mu = plummer(n)
        world = 2
        bar = threading.Barrier(world)
        bufs = [None] * world
        results = [None] * world
        errors = []

        def exchange_for(rank):
            def exchange(dev_ptr, slice_bytes, r, w, stream):
                if hip.hipStreamSynchronize(stream):
                    return 2
                bufs[rank] = dev_ptr
                bar.wait()                                   # both slices written, both buffer addresses known
                peer = 1 - rank
                if hip.hipMemcpy(dev_ptr + peer * slice_bytes, bufs[peer] + peer * slice_bytes, slice_bytes, 3):   # D2D
                    return 3
                bar.wait()                                   # nobody overwrites its slice before the peer has copied it
                return 0
            return exchange

        def worker(rank):
            try:
                nb = ea.NBodyIntegration(pos, vel, mu, 0.0, H).shard(rank, world, exchange=exchange_for(rank))
                nb.advance(steps)
                results[rank] = (nb.state(), nb.acc(), nb.shard_info())
            except Exception as e:                           # noqa: BLE001
                errors.append(e)
                bar.abort()

        threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        (p0, v0, t0, sc0), a0 = _single(n, "QuinlanTremaine12", steps)
        for r in range(world):
            (p, v, t, sc), a, (lo, hi, gathers) = results[r]
            assert (lo, hi) == (r * n // 2, (r + 1) * n // 2) and gathers > 0
            assert t == t0 and sc == sc0
            assert np.array_equal(p, p0) and np.array_equal(v, v0) and np.array_equal(a, a0), (n, r)


def test_propagator_sharded_after_steps_keeps_the_polynomial_order(gpu):
    """eph_prop_shard on a propagator that has already stepped (device-resident pending polynomials and a queue of
    deferred steps): the older windows must reach the host splines before the sharded branch pushes newer ones."""
    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd.workloads import plummer
    n = 256
    pos, vel, mu = plummer(n)
    count = (np.arange(n) % 3 + 1).astype(np.uint32)
    degree = (np.arange(n) % 3 + 5).astype(np.uint32)
    for direction in (ea.FORWARD, ea.BACKWARD):
        p = ea.NBodyPropagator(pos, vel, mu, 0.0, H, direction, count, degree)
        q = ea.NBodyPropagator(pos, vel, mu, 0.0, H, direction, count, degree)
        p.step_n(70)
        for _ in range(7):
            p.step()                                  # queued, not yet run
        p.shard(0, 1, unique_id=ea.rccl_unique_id())
        p.step_n(90)
        q.step_n(70 + 7 + 90)
        assert p.time() == q.time()
        sp, sq = p.take_solution(), q.take_solution()
        for b in range(n):
            assert sp.info(b) == sq.info(b) and sp.info(b)[2] > 1
            (cp, np_), (cq, nq) = sp.coeffs(b), sq.coeffs(b)
            assert np.array_equal(np_, nq) and np.array_equal(cp, cq), (direction, b)


def _lost_peer_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), EPH_PEER_TIMEOUT_MS="300")
    import torch.distributed as dist
    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd import parallel
    from ephemeris_explorer_amd.workloads import plummer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pos, vel, mu = plummer(256)
    nb = ea.NBodyIntegration(pos, vel, mu, 0.0, H, "BlanesMoan6B")
    parallel.shard_nbody(nb, dist, transport="peer")
    if rank == 0:                                      # rank 1 never steps: rank 0's wait must end by itself
        try:
            nb.advance(1)
            nb.sync()
            out[0] = "no error"
        except ea.EphemerisError as e:
            out[0] = (e.status, str(e))
    dist.barrier()
    dist.destroy_process_group()


def test_peer_that_never_delivers_is_an_error_not_a_hang(gpu):
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_lost_peer_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    status, text = out[0]
    assert status == -6 and "rank 1" in text


def _fallback_worker(rank, world, port, force, out):
    import sys
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), EPH_PEER_TIMEOUT_MS="5000")
    if force:
        os.environ["EPH_PEER_FORCE_FAIL"] = force
    import torch.distributed as dist
    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd import parallel
    from ephemeris_explorer_amd.workloads import plummer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pos, vel, mu = plummer(1024)
    nb = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
    t = parallel.peer_transport(dist, slot_bytes=1 << 16)
    nb.shard_peer(t)
    nb.advance(12 + 8)
    p, v = nb.state()[:2]
    out[rank] = (t.memory, list(t.forms), list(t.attempts), p, v)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("force,memory", [("", None), ("alloc", "coarse"), ("export", "coarse"), ("open", "coarse")])
def test_peer_mailbox_fallbacks(gpu, force, memory):
    """Every branch of the mailbox's fallback (csrc/peer.hip, parallel.peer_transport), forced with EPH_PEER_FORCE_FAIL: the
    fine-grained allocation fails, its hipIpc export fails (both: plain device memory inside eph_peer_create), a peer cannot map
    it (every rank re-creates in plain device memory and connects again) -- the exchange works and reports which form is live."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_fallback_worker, args=(2, _free_port(), force, out), nprocs=2, join=True)
    (p0, v0, t0, sc0), a0 = _single(1024, "QuinlanTremaine12", 12 + 8)
    for r in range(2):
        mem, forms, attempts, p, v = out[r]
        assert np.array_equal(p, p0) and np.array_equal(v, v0), (force, r)
        assert mem in ("fine", "coarse") and forms == [mem, mem]
        if memory:
            assert mem == memory
        if force == "open":
            assert attempts and all("connect[auto]" in a for a in attempts)      # the first round failed on every rank, and says so
        else:
            assert attempts == []


def _lost_peer_state_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), EPH_PEER_TIMEOUT_MS="300")
    import torch.distributed as dist
    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd import parallel
    from ephemeris_explorer_amd.workloads import plummer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pos, vel, mu = plummer(256)
    nb = ea.NBodyIntegration(pos, vel, mu, 0.0, H, "BlanesMoan6B")
    parallel.shard_nbody(nb, dist, transport="peer")
    if rank == 0:                                      # rank 1 never joins the gather of get_state: THIS call must report it
        try:
            nb.state()
            out[0] = "no error"
        except ea.EphemerisError as e:
            out[0] = (e.status, str(e))
    dist.barrier()
    dist.destroy_process_group()


def test_get_state_reports_a_lost_peer_itself(gpu):
    """(advisor, round 3) a timed-out wait used to surface at the NEXT exchange, with the stale mailbox slot copied into the gathered
    buffer and EPH_OK returned: the call that synchronises the stream now polls the transport, and the kernel skips the copy-out"""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_lost_peer_state_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    status, text = out[0]
    assert status == -6 and "rank 1" in text


# ---- BASELINE configs[4] as stated: binary32 pair arithmetic ON a target partition ------------------------------------------------
def _single_f32(n, steps_f32, steps_exact):
    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd.workloads import plummer
    pos, vel, mu = plummer(n)
    nb = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
    nb.set_path(ea.PATH_F32_PAIRS)
    nb.advance(12 + steps_f32)
    mid = nb.state(), nb.acc()
    nb.set_path(0)
    nb.advance(steps_exact)
    return mid, (nb.state(), nb.acc())


def _f32_worker(rank, world, port, n, steps_f32, steps_exact, out, transport):
    import sys
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd import parallel
    from ephemeris_explorer_amd.workloads import plummer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pos, vel, mu = plummer(n)
    nb = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
    nb.set_path(ea.PATH_F32_PAIRS)
    if transport == "host":
        # the host-staged exchange costs ~0.2 s per gather between processes that share a GPU and the start-up has 336 of them
        # (covered sharded by test_ranks_on_one_gpu_match_single_device): every rank starts the whole system up, then the partition
        nb.advance(12)
        parallel.shard_nbody(nb, dist, transport=transport)
        nb.advance(1)
    else:
        parallel.shard_nbody(nb, dist, transport=transport)
        nb.advance(12 + 1)                            # start-up (f64, sharded) and the first binary32 step
    g0 = nb.shard_info()[2]
    nb.advance(steps_f32 - 1)                         # one batch: one f64 gather (the batch's prediction) + one f32 gather per step
    g1 = nb.shard_info()[2]
    mid = nb.state(), nb.acc()
    twin = nb.clone()
    nb.set_path(0)                                    # back to the exact path on the same partition: nothing stale is read
    nb.advance(steps_exact)
    twin.set_path(0)
    twin.advance(steps_exact)
    end = nb.state(), nb.acc()
    out[rank] = (mid, end, g1 - g0, np.array_equal(twin.state()[0], end[0][0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,world,transport", [(4096, 2, "host"), (4096, 4, "peer"), (4050, 2, "peer"), (16384, 2, "peer"),
                                               (16384, 4, "host")])
def test_f32_pairs_on_a_target_partition(gpu, n, world, transport, monkeypatch):
    """EPH_PATH_F32_PAIRS on a sharded handle (BASELINE configs[4]: "65 536-body f32 system, 8 x MI355X shard"): each rank converts
    ITS rows to binary32, one all-gather of 16 B per body, then k_fast_partial_f32 over its own targets. The slices of the f64
    accumulation are cut on GLOBAL source indices, so the result is bit-identical to the single-device f32 path for any world
    size (the reference has no f32 path -- ephemeris/src/propagators/nbody.rs:13,19 -- so that is what there is to compare with).
    Then the same handles go back to the exact path and stay identical to the single device that did the same."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("EPH_PEER_TIMEOUT_MS", "5000")
    steps_f32, steps_exact = (30, 3) if n <= 4096 and transport != "host" else (8, 2)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_f32_worker, args=(world, _free_port(), n, steps_f32, steps_exact, out, transport), nprocs=world, join=True)
    mid0, end0 = _single_f32(n, steps_f32, steps_exact)
    assert set(out.keys()) == set(range(world))
    for r in range(world):
        mid, end, gathers, twin_ok = out[r]
        for (st, a), (st0, a0), what in ((mid, mid0, "f32 leg"), (end, end0, "exact leg after it")):
            assert st[2:] == st0[2:], what
            assert np.array_equal(st[0], st0[0]) and np.array_equal(st[1], st0[1]) and np.array_equal(a, a0), (r, what)
        assert twin_ok
        assert gathers == 1 + (steps_f32 - 1)        # the batch's first prediction in f64 + the binary32 rows once per step


def test_configs4_at_full_width_on_a_partition(gpu, monkeypatch):
    """BASELINE configs[4] as stated, at its stated size: 65 536 bodies, binary32 pair arithmetic (EPH_PATH_F32_PAIRS) on a target
    partition, two ranks (sharing this box's one GPU), direct peer writes into hipIpc-mapped mailboxes; the sharded start-up (302
    exact force evaluations), 4 binary32 steps and one exact step afterwards, bit-identical to the single-device run of the same."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("EPH_PEER_TIMEOUT_MS", "20000")
    n, world, steps_f32, steps_exact = 65536, 2, 4, 1
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_f32_worker, args=(world, _free_port(), n, steps_f32, steps_exact, out, "peer"), nprocs=world, join=True)
    mid0, end0 = _single_f32(n, steps_f32, steps_exact)
    assert set(out.keys()) == set(range(world))
    for r in range(world):
        mid, end, gathers, twin_ok = out[r]
        for (st, a), (st0, a0), what in ((mid, mid0, "f32 leg"), (end, end0, "exact leg after it")):
            assert st[2:] == st0[2:], what
            assert np.array_equal(st[0], st0[0]) and np.array_equal(st[1], st0[1]) and np.array_equal(a, a0), (r, what)
        assert twin_ok and gathers == 1 + (steps_f32 - 1)


def test_f32_pairs_sharded_through_rccl(gpu):
    """the same through ncclAllGather (a one-rank communicator: the collective still runs on the 16-byte rows)"""
    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd.workloads import plummer
    pos, vel, mu = plummer(4096)
    nb = ea.NBodyIntegration(pos, vel, mu, 0.0, H).shard(0, 1, unique_id=ea.rccl_unique_id())
    nb.set_path(ea.PATH_F32_PAIRS)
    nb.advance(12 + 25)
    one = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
    one.set_path(ea.PATH_F32_PAIRS)
    one.advance(12 + 25)
    assert np.array_equal(nb.state()[0], one.state()[0]) and np.array_equal(nb.acc(), one.acc())
    for path in (ea.PATH_FAST, ea.PATH_FAST_RSQ):    # the f64 reordered paths stay single-device
        nb.set_path(path)
        with pytest.raises(ea.EphemerisError) as e:
            nb.advance(1)
        assert e.value.status == ea.ERR_UNSUPPORTED
