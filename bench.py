#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric).

Workload (config.workload): BASELINE.json configs[2], the configuration the metric is quoted on -- a synthetic
4096-body Plummer sphere, f64, QuinlanTremaine12 (the reference's N-body method), h = 1/1024 N-body time units
(SURVEY.md §8(d)3). A "step" is one integrator step of the whole system: one pass of the hot path
(predictor -> all-pairs acceleration in the reference's summation order -> Cowell velocity) over all bodies.
Start-up (12 macro steps through 4x BlanesMoan6B sub-steps) happens before the warm-up; state is resident in HBM.

Multi-GPU: the time stepping of one N-body system is serial and every step needs every position, so the path
does not shard without a per-step exchange (DESIGN.md "Multi-GPU"): `--gpus N` runs N independent replicas
(weak scaling, e.g. the reference's concurrent forward/backward propagators or an ensemble); value = bodies x
steps summed over ranks / max-over-ranks time. No data-path collective.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

N_BODIES = 4096
H = 1.0 / 1024.0
# SURVEY.md §8(d) algorithmic counts
BYTES_PER_BODY_STEP = 680.0            # 12 y + 12 a levels (576) + mu (8) + own position (24) + write y, v, a (72)
FLOP_PER_INTERACTION = 20.0
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP64_VECTOR_PEAK_TFLOPS = 78.6         # MI355X vector FP64 (no MFMA on this path)
FP32_VECTOR_PEAK_TFLOPS = 157.3        # MI355X vector FP32 (packed v_pk_fma_f32): the roof of --path f32-pairs' pair arithmetic


def cpu_baseline(pos, vel, mu, steps):
    """The oracle (a port of the reference algorithm: single thread, scalar, triangular N(N-1)/2 pair loop) timed
    on this host on a bounded sample of the same workload. Also returns the oracle state for the parity figure."""
    from oracle import orc
    orc.build(native=True)
    o = orc.NBody(pos, vel, mu, 0.0, H, native=True)
    t0 = time.perf_counter()
    assert o.advance(12) == 0
    t_start = time.perf_counter() - t0
    t0 = time.perf_counter()
    assert o.advance(steps) == 0
    dt = time.perf_counter() - t0
    pairs = N_BODIES * (N_BODIES - 1) / 2
    base = {
        "value": N_BODIES * steps / dt, "unit": "body-steps/s", "cores": 1, "kind": "port",
        "sample": f"{steps} steady-state QuinlanTremaine12 steps of the same 4096-body system "
                  f"(after the 12-step start-up, {t_start:.1f} s, not counted)",
        "ns_per_pair": dt / steps / pairs * 1e9, "seconds": dt,
    }
    # beside it, clearly labelled: what a parallel CPU could do -- the same sums partitioned by target body over all
    # host threads (OpenMP, all N^2 directed interactions, same bits); the reference itself is single-threaded per
    # propagator
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                               # a container's CPU quota, if any (cgroup v2 / v1)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = max(1, min(usable, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            if q > 0:
                usable = max(1, min(usable, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except (OSError, ValueError):
            pass
    if usable > 1:
        best = None
        for t in sorted({usable, min(usable, 64), min(usable, 16)}):     # affinity can overstate what is schedulable
            orc.set_gravity_threads(t, native=True)
            orc.gravity(pos, mu, native=True)
            t0 = time.perf_counter()
            orc.gravity(pos, mu, native=True)
            e = time.perf_counter() - t0
            if best is None or e < best[0]:
                best = (e, t)
        threads = best[1]
        orc.set_gravity_threads(threads, native=True)
        try:
            t0 = time.perf_counter()
            assert o.advance(steps) == 0
            dtp = time.perf_counter() - t0
        finally:
            orc.set_gravity_threads(0, native=True)
        base["all_cores"] = {"value": N_BODIES * steps / dtp, "unit": "body-steps/s", "cores": threads,
                             "usable_cores": usable,
                             "kind": "port, target-partitioned OpenMP (not how the reference runs)", "seconds": dtp}
    return base, o


def sharded_cpu_baseline(pos, mu, n, evals=2):
    """One steady QuinlanTremaine12 step of a large system is one NewtonianGravity::eval plus O(N) work, and the oracle's
    start-up alone would take half an hour at 65 536 bodies: the bounded sample is `evals` force evaluations of the same
    system by the oracle (single thread, triangular pair loop like the reference)."""
    from oracle import orc
    orc.build(native=True)
    t0 = time.perf_counter()
    for _ in range(evals):
        orc.gravity(pos, mu, native=True)
    dt = time.perf_counter() - t0
    return {"value": n * evals / dt, "unit": "body-steps/s", "cores": 1, "kind": "port",
            "sample": f"{evals} NewtonianGravity::eval of the same {n}-body system (a steady multistep step = one "
                      "evaluation + O(N)); start-up not run on the CPU", "seconds": dt,
            "ns_per_pair": dt / evals / (n * (n - 1) / 2) * 1e9}


def craft_main(args):
    """BASELINE.json configs[3] (bounded): full_solar_system ephemeris + spacecraft sharded over the ranks in
    contiguous blocks (independent given the ephemeris, which every rank rebuilds bit-identically: no data-path
    collective). One step = one sweep of all craft over `--craft-days` days (Verner87, tol 1e-3 km)."""
    import numpy as np
    import torch

    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd.parallel import env_rank, reduce_timing, shard_range
    from ephemeris_explorer_amd.systems import load_ship, load_system

    rank, local_rank, world = env_rank()
    local_rank %= max(ea.device_count(), 1)            # (lets the N > 1 flow be exercised on a one-GPU box)
    torch.cuda.set_device(local_rank)
    ea.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("EPH_BENCH_BACKEND", "nccl")   # "gloo" only to test this flow with ranks sharing a GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    backend_is_nccl = world > 1 and os.environ.get("EPH_BENCH_BACKEND", "nccl") == "nccl"
    sysdir = ROOT / "tests/golden/systems/full_solar_system_2433282.5"
    s = load_system(sysdir)
    ship = load_ship(sysdir / "ships" / "Mars Transfer Ship.json")
    # The MB-scale ephemeris, both ways, both timed (max over ranks): every rank integrating the bodies itself, bit-identically
    # (rounds 2-5), and SURVEY 8(e)'s design -- rank 0 builds the table once, exports ONE contiguous image, broadcasts it (RCCL over
    # xGMI for world > 1), the others import it. The sweep runs against the BROADCAST table; the two are compared image for image.
    from ephemeris_explorer_amd.parallel import broadcast_ephemeris

    def build_table():
        sol = ea.NBodyPropagator.from_system(s).propagate(s.epoch + (args.craft_days + 40.0) * 86400.0)
        return ea.Ephemeris(sol, s.mu)

    build_table()                                        # (the process's first kernel launches, code-object load: neither leg pays it)
    t_eph0 = time.perf_counter()
    eph_local = build_table()
    eph_rebuild_s = time.perf_counter() - t_eph0
    if dist is not None:
        dist.barrier()
    t_eph0 = time.perf_counter()
    eph, eph_parts = broadcast_ephemeris(build_table, dist, device="cuda" if backend_is_nccl else "cpu")
    eph_broadcast_s = time.perf_counter() - t_eph0
    same_table = bool(np.array_equal(eph.export_image(), eph_local.export_image()))
    assert same_table, "the broadcast table differs from the locally integrated one"
    del eph_local
    from ephemeris_explorer_amd.workloads import craft_population, wave_divergence
    pos, vel, family = craft_population(args.population, args.craft, s, ship, order=args.population_order)
    lo, hi = shard_range(args.craft, rank, world)
    t_end = ship.start + args.craft_days * 86400.0
    max_knots = int((1200 if args.population == "transfer" else 400) * args.craft_days) + 64   # (a low orbit: ~215 knots a day)

    def make():          # one SpacecraftPropagator per craft: states, timelines and knot slabs resident in HBM
        return ea.SpacecraftBatch(eph, ship.start, pos[lo:hi], vel[lo:hi], "Verner87", max_knots=max_knots)

    def sweep(b=None):
        b = b or make()
        b.propagate(t_end)
        return b

    from ephemeris_explorer_amd.parallel import gather_craft_states
    t_pre = time.perf_counter()
    for _ in range(args.warmup if args.warmup < 3 else 2):
        sweep()
    while time.perf_counter() - t_pre < args.prewarm:   # untimed: bring the board from idle to its steady clock (see main())
        sweep()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    steps_local, attempts_local, ms = 0, 0, 0.0
    nsweeps = max(1, min(args.steps, 5))
    table = None
    batches = [make() for _ in range(nsweeps)]          # inputs resident before the timed region (a batch is ~20 KB per craft)
    # The record array the sweeps' read-backs land in is the caller's and is kept between sweeps (its pages are mapped), and
    # every batch is read back once BEFORE the timed region: the first read-back after a burst of batch creations starts 15-40 ms
    # late on the device, once (profiles/r04_sweep_evidence.md) -- set-up, like the creations themselves.
    rec = np.zeros(hi - lo, dtype=ea.SpacecraftBatch.RECORD)
    for b in batches:
        b.summary(rec)
    barrier()
    t0 = time.perf_counter()
    for b in batches:
        sweep(b)
        st = b.summary(rec)                             # ONE device-packed record per craft, one copy
        assert (st["status"] == 0).all()
        steps_local += int(st["steps"].sum())
        attempts_local += int(st["attempts"].sum())
        ms += b.kernel_ms()
        # the sweep's one exchange step: final states of all craft on every rank -- ncclAllGather over
        # xGMI for world > 1 (north_star: "RCCL over xGMI only for the embarrassingly-parallel spacecraft sweep")
        table = gather_craft_states(st, args.craft, dist, device="cuda" if backend_is_nccl else "cpu")
    barrier()
    elapsed = time.perf_counter() - t0
    assert table.shape == (args.craft, 7) and np.isfinite(table).all() and (table[:, 0] >= t_end).all()
    units, elapsed = reduce_timing(elapsed, steps_local, dist, device="cuda")
    _, eph_rebuild_s = reduce_timing(eph_rebuild_s, 0, dist, device="cuda")
    _, eph_broadcast_s = reduce_timing(eph_broadcast_s, 0, dist, device="cuda")
    if rank == 0:
        # per craft-attempt 13 evaluations x B bodies x (Horner + index + point mass) ~ 74 flop; 56 B knot
        # written per ACCEPTED step (the ephemeris rows stay in L1/L2)
        flop = attempts_local * 13.0 * s.n * 74.0
        launch_s = ms * 1e-3 / nsweeps
        out = {
            "metric": "craft-steps/s", "value": units / elapsed, "unit": "accepted integrator steps/s",
            "n_gpus": world, "steps": nsweeps, "warmup": (args.warmup if args.warmup < 3 else 2), "ms_per_step": elapsed / nsweeps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"full_solar_system ephemeris + {args.craft} craft x {args.craft_days} d, Verner87 "
                                   "tol 1e-3 (BASELINE.json configs[3], bounded)"
                                   + ("" if args.population == "transfer" else
                                      f"; population '{args.population}' ({args.population_order}): LEO / GTO / lunar transfer / "
                                      "heliocentric in equal numbers"),
                       "parallelism": f"craft sharded x{world}",
                       "exchange": "1 all-gather of the final states per sweep (56 B per craft)" if world > 1 else "none (1 rank)"},
            "kernel_ms_rank0": ms, "includes": "sweep kernel + per-craft status / final state read-back + result all-gather "
                                              "(batches created and read back once before the timed region: state resident in HBM)",
            "roofline": {"bound": "hbm", "achieved": 56.0 * steps_local / nsweeps / launch_s / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": 56.0 * steps_local / nsweeps / launch_s / 1e9 / HBM_PEAK_GBS,
                         "traffic": None,
                         "kernel": "k_craft_propagate<13,false,false,2>, every batch dealt to the lanes by orbital time scale "
                                   "(craft.hip craft_sort; k_craft_queue<13,false> only with EPH_CRAFT_SORT=0|1)",
                         "launch_us": launch_s * 1e6,
                         "algorithmic_bytes_per_launch": 56.0 * steps_local / nsweeps,
                         "binding": "fp64_valu",
                         "fp64": {"achieved": flop / nsweeps / launch_s / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS,
                                  "unit": "TFLOP/s", "frac": flop / nsweeps / launch_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                  "flop_per_launch": flop / nsweeps, "attempts_per_launch": attempts_local / nsweeps,
                                  "count": "13 stages x 32 bodies x 74 flop per attempt (SURVEY 8(d))"},
                         "note": "rank 0's shard; the sweep is f64-VALU bound (every ephemeris row is an L1 hit), see roofline.fp64"},
            "wall_over_kernel": elapsed / nsweeps / launch_s,
            "ephemeris_rebuild_s": {"max_over_ranks": eph_rebuild_s, "days": args.craft_days + 40.0,
                                    "what": "NBodyPropagator of the 32-body system to the sweep's horizon + 40 d and its device table, "
                                            "done by EVERY rank (after one untimed build that pays the process's first launches)"},
            "ephemeris_broadcast_s": {"max_over_ranks": eph_broadcast_s, "rank0_parts": eph_parts,
                                      "what": "SURVEY 8(e): rank 0 builds the table once, eph_ephemeris_export -> one broadcast of the "
                                              "image (RCCL for world > 1) -> eph_ephemeris_import on the other ranks; the sweep runs "
                                              "against THIS table", "identical_to_the_local_rebuild": same_table},
        }
        out["fp64"] = dict(out["roofline"]["fp64"], bound="fp64_valu")
        # counters of the committed rocprofv3 passes of this kernel (scripts/prof_craft.sh), scaled by the attempts of THIS run;
        # dropped when the kernel's sources changed since they were taken
        tj = ROOT / "profiles" / "traffic_craft.json"
        if tj.exists() and args.population == "transfer":
            from ephemeris_explorer_amd.workloads import profile_is_current
            tinfo = json.loads(tj.read_text())
            current, why = profile_is_current(tinfo, "craft")
            out["roofline"]["traffic_profile_commit"] = tinfo.get("profile_commit")
            if current and tinfo.get("traffic_bytes_per_attempt"):
                att_launch = attempts_local / nsweeps
                out["roofline"]["traffic"] = tinfo["traffic_bytes_per_attempt"] * att_launch
                out["roofline"]["traffic_source"] = (tinfo.get("source", "") + ": raw FETCH_SIZE + WRITE_SIZE per attempt x this launch's attempts "
                                                     "(1.5 x the 56 B per accepted step since round 6 -- the stage derivatives' velocity halves live in LDS; round 5: "
                                                     "5.9 x, the scratch stores of 100 spilled VGPRs)")
                lane_ops = tinfo["valu_wave_insts_per_attempt"] * att_launch * 64.0 / launch_s
                out["roofline"]["fp64"]["valu_issue"] = {
                    "achieved": lane_ops / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS / 2.0, "unit": "T lane-ops/s",
                    "frac": lane_ops / 1e12 / (FP64_VECTOR_PEAK_TFLOPS / 2.0),
                    "valu_wave_insts_per_attempt": tinfo["valu_wave_insts_per_attempt"],
                    "f64_wave_insts_per_attempt": tinfo.get("f64_wave_insts_per_attempt"),
                    "active_inst_valu_over_wave_cycles": tinfo.get("active_inst_valu_over_wave_cycles"),
                    "waves_per_simd": tinfo.get("waves_per_simd"), "vgpr_count": tinfo.get("vgpr_count"),
                    "vgpr_spill_count": tinfo.get("vgpr_spill_count"), "scratch_bytes_per_lane": tinfo.get("scratch_bytes_per_lane"),
                    "lds_bytes_per_workgroup": tinfo.get("lds_bytes_per_workgroup")}
                out["fp64"]["valu_issue"] = out["roofline"]["fp64"]["valu_issue"]
            elif not current:
                out["roofline"]["traffic_stale"] = f"{tj.name} not used: {why}"
        # lane idling of a static craft -> lane assignment: per wave max / mean attempts (1.0 = none), and what the
        # kernel did about it (the work queue of k_craft_propagate refills finished lanes)
        att = st["attempts"].astype(np.float64)
        out["divergence"] = {"attempts_max_over_mean_per_wave": wave_divergence(att),
                             "order": "craft index -- what an UNDEALT batch would idle; the kernel runs the dealt order (1.01, "
                                      "profiles/r03_craft_queue.md)",
                             "attempts_per_craft_min": float(att.min()), "attempts_per_craft_max": float(att.max()),
                             "steps_by_family": {str(f): float(st["steps"][family[lo:hi] == f].mean())
                                                 for f in sorted(set(family[lo:hi].tolist()))}}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = craft_cpu_baseline(s, ship, pos, vel, t_end, args.craft_days)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def craft_cpu_baseline(s, ship, pos, vel, t_end, craft_days, sample=1000):
    """The oracle (port of SpacecraftPropagator::step, one thread) on the first `sample` craft of the same sweep against
    the oracle's own ephemeris of the same system; the per-craft rate is what a CPU core does, the whole-workload figure
    would be an extrapolation and is not printed."""
    from oracle import orc
    orc.build()
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree)
    assert o.step_to(s.epoch + (craft_days + 40.0) * 86400.0) == 0
    eph = o.take_solution()
    n = min(sample, len(pos))
    t0 = time.perf_counter()
    steps = 0
    for i in range(n):
        c = orc.Craft(eph, s.mu, ship.start, pos[i], vel[i], "Verner87")
        assert c.step_to(t_end) == 0
        steps += len(c.knots()[0]) - 1
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "accepted integrator steps/s", "cores": 1, "kind": "port",
            "sample": f"the first {n} craft of the same sweep ({craft_days} d each, Verner87 tol 1e-3), single thread; "
                      "craft are independent, so the rate per core carries over -- not extrapolated to the full batch here",
            "seconds": dt, "craft": n}


def transport_preflight(dist, world, rank, backend_is_nccl, record):
    """Before any sharded timing: every transport of the exchange step (RCCL when each rank has its own device, direct peer
    writes, host staging) runs start-up + 5 steps of a 1024-body system partitioned over the job's ranks, and every rank compares
    the gathered state bit for bit with its own single-device run. `record` (filled in place, so that a watchdog that fires
    sees what was established so far) gets {"rccl": "ok" | "<what failed, per rank>", "peer": ..., "host": ..., "peer_memory":
    [form per rank]}. A transport that HANGS on hardware it has never met is the watchdog's business (main())."""
    import numpy as np

    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd.parallel import host_staged_exchange, peer_transport, shard_nbody
    from ephemeris_explorer_amd.workloads import plummer
    pos, vel, mu = plummer(1024, seed=20260927)
    ref = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
    ref.advance(12 + 5)
    rp, rv = ref.state()[:2]
    del ref
    host_group = dist.new_group(backend="gloo") if backend_is_nccl else None      # CPU tensors need a gloo group
    for transport in (["rccl"] if backend_is_nccl else []) + ["peer", "host"]:
        status, forms = "ok", None
        try:
            g = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
            if transport == "host":
                g.shard(rank, world, exchange=host_staged_exchange(dist, group=host_group))
            elif transport == "peer":
                t = peer_transport(dist, slot_bytes=1 << 20)
                forms = t.forms
                g.shard_peer(t)
            else:
                shard_nbody(g, dist, transport="rccl", device="cuda")
            g.advance(12 + 5)
            p, v = g.state()[:2]
            if not (np.array_equal(p, rp) and np.array_equal(v, rv)):
                status = "bits differ from the single-device run"
            del g
        except Exception as e:
            status = f"{type(e).__name__}: {e}"[:200]
        every = [None] * world
        dist.all_gather_object(every, status, group=host_group)        # (the gloo group under nccl: no device tensors for a string)
        bad = [f"rank {r}: {st}" for r, st in enumerate(every) if st != "ok"]
        record[transport] = "ok" if not bad else "; ".join(bad)[:400]
        if forms is not None:
            record["peer_memory"] = forms
    return record


def sharded_4096(dist, world, rank, steps, backend_is_nccl, usable=None):
    """"2/4/8-GPU scaling of config 3 is expected to be poor and must be reported as measured". The metric's
    own 4096-body system as ONE system partitioned by target body over the ranks of this job, once per transport that can
    run here (direct peer writes always; RCCL when every rank has its own device), timed like the main line (repeated
    K-step blocks, median) and compared bit for bit with a single-device run of the same steps on rank 0."""
    import numpy as np
    import torch

    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd.parallel import reduce_timing, shard_nbody
    from ephemeris_explorer_amd.workloads import plummer
    pos, vel, mu = plummer(N_BODIES, seed=20260926)
    out = {"ranks": world, "bodies": N_BODIES, "bodies_per_gpu": N_BODIES // world, "transports": {}}

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    total = 0
    # RCCL first (when every rank has its own device), then the direct-write transport; EPH_BENCH_SHARDED=rccl|peer|0 narrows it
    want = os.environ.get("EPH_BENCH_SHARDED", "rccl,peer").split(",")
    for transport in [t for t in (["rccl", "peer"] if backend_is_nccl else ["peer"]) if t in want]:
        if usable is not None and usable.get(transport) != "ok":
            out["transports"][transport] = {"error": "failed the preflight: " + str(usable.get(transport))[:300]}
            continue
        try:
            g = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
            shard_nbody(g, dist, transport=transport, device="cuda" if backend_is_nccl else "cpu")
            g.advance(12 + 5)
            g.sync()

            def block():
                g.advance(steps)
                g.sync()

            blocks = timed_blocks(block, barrier, lambda t: reduce_timing(t, 0, dist, device="cuda")[1])
            _, med = reduce_timing(blocks[len(blocks) // 2], 0, dist, device="cuda")
            p, v, t, sc = g.state()                     # collective
            total = 12 + 5 + steps * len(blocks)
            out["transports"][transport] = {"ms_per_step": med / steps * 1e3, "blocks": len(blocks),
                                            "gathers": g.shard_info()[2], "body_steps_per_s": N_BODIES * steps / med}
            if rank == 0:
                ref = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
                ref.advance(total)
                same = bool(np.array_equal(ref.state()[0], p) and np.array_equal(ref.state()[1], v))
                out["transports"][transport]["bit_identical_to_single_device"] = same
            del g
        except Exception as e:                          # the main line must survive a transport that cannot run here
            out["transports"][transport] = {"error": f"{type(e).__name__}: {e}"[:300]}
        barrier()
    ok = {k: v for k, v in out["transports"].items() if "ms_per_step" in v}
    if ok:
        best = min(ok, key=lambda k: ok[k]["ms_per_step"])
        out.update(transport=best, ms_per_step=ok[best]["ms_per_step"], gathers=ok[best]["gathers"],
                   bit_identical_to_single_device=all(v.get("bit_identical_to_single_device", True) for v in ok.values()))
    return out


def configs4_f32_sharded(dist, world, rank, steps, backend_is_nccl, usable, bodies):
    """BASELINE.json configs[4] AS STATED, measured by the default N > 1 command: ONE system of `bodies` (65 536) partitioned by target
    body over the ranks with binary32 pair arithmetic (EPH_PATH_F32_PAIRS on a sharded handle: each rank converts its rows, one
    all-gather of 16 B per body per step, f64 accumulation in the global slice order, f64 integrator), timed like the main line and
    compared bit for bit with rank 0's own single-device f32 run of the same steps -- the reference has no f32 path
    (ephemeris/src/propagators/nbody.rs:13,19), so that is what there is to compare with; the single-device time beside it is the
    strong-scaling denominator."""
    import numpy as np
    import torch

    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd.parallel import reduce_timing, shard_nbody
    from ephemeris_explorer_amd.workloads import plummer
    pos, vel, mu = plummer(bodies, seed=20260926)
    out = {"ranks": world, "bodies": bodies, "bodies_per_gpu": bodies // world, "exchange_bytes_per_step": 16 * bodies,
           "path": "f32-pairs on a target partition"}
    order = [t for t in (["rccl", "peer"] if backend_is_nccl else ["peer"]) if usable is None or usable.get(t) == "ok"]
    if not order:
        out["error"] = "no transport passed the preflight"
        return out
    transport = order[0]
    out["transport"] = transport

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    g = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
    g.set_path(ea.PATH_F32_PAIRS)
    shard_nbody(g, dist, transport=transport, device="cuda" if backend_is_nccl else "cpu")
    g.advance(12 + 3)
    g.sync()

    def block():
        g.advance(steps)
        g.sync()

    blocks = timed_blocks(block, barrier, lambda t: reduce_timing(t, 0, dist, device="cuda")[1], blocks=3)
    _, med = reduce_timing(blocks[len(blocks) // 2], 0, dist, device="cuda")
    p, v, t, sc = g.state()                               # collective
    total = 12 + 3 + steps * len(blocks)
    out.update(ms_per_step=med / steps * 1e3, body_steps_per_s=bodies * steps / med, blocks=len(blocks), gathers=g.shard_info()[2])
    if rank == 0:
        one = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
        one.set_path(ea.PATH_F32_PAIRS)
        one.advance(total - steps)
        one.sync()
        t0 = time.perf_counter()
        one.advance(steps)
        one.sync()
        single = time.perf_counter() - t0
        sp, sv = one.state()[:2]
        out["bit_identical_to_single_device_f32"] = bool(np.array_equal(sp, p) and np.array_equal(sv, v))
        out["single_device_ms_per_step"] = single / steps * 1e3
        out["speedup_over_one_gpu"] = single / med
    del g
    barrier()
    return out


def other_variants(pos, vel, mu, steps=400):
    """us per step of the same system under the OTHER evaluation orders of the unpinned point-mass term (csrc/pair_term.h): the
    headline is measured on order `pair_variant` (0 unless EPH_PAIR_VARIANT says otherwise); should the Rust binary follow another
    order (tools/identify_pair_variant.py), this is what the step costs there. HIP events on each handle's stream."""
    import ephemeris_explorer_amd as ea
    out, keep = {}, ea.pair_variant()
    try:
        for k in range(7):
            if k == keep:
                continue
            ea.set_pair_variant(k)
            g = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
            g.advance(12 + 50)
            g.enable_timing(True)
            g.advance(steps)
            g.sync()
            ms, launches = g.kernel_time()
            out[str(k)] = {"us_per_step": ms / launches * 1e3, "body_steps_per_s": len(mu) / (ms / launches * 1e-3)}
            del g
    except Exception as e:
        out["error"] = f"{type(e).__name__}: {e}"[:300]
    finally:
        ea.set_pair_variant(keep)
    return out


def other_configs():
    """The default line's companions, taken AFTER the headline measurement is complete (N = 1 only) so that one driver command
    records them too: BASELINE.json configs[1] -- the reference's shipping system, full_solar_system (32 bodies), 1e6 steps of the
    propagator with its solout and every least-squares fit, in this process -- and configs[3], the massless sweep, as a child
    process of this file (`--workload craft`) under a time limit. Neither can cost the headline: failures are recorded."""
    import subprocess
    import numpy as np
    out = {}
    try:
        import ephemeris_explorer_amd as ea
        from ephemeris_explorer_amd.systems import load_system
        s = load_system(ROOT / "tests/golden/systems/full_solar_system_2433282.5")
        p = ea.NBodyPropagator.from_system(s)
        p.step_n(20000)                                  # start-up + clock
        t = time.perf_counter()
        p.step_n(1_000_000)
        w = time.perf_counter() - t
        c1 = {"bodies": int(s.n), "steps": 1000000, "seconds": w, "us_per_step": w / 1e6 * 1e6,
              "body_steps_per_s": s.n * 1e6 / w,
              "includes": "k_lm_small steps + solout sampling + every least-squares fit"}
        out["configs1_full_solar_system"] = c1
        # the same 1 020 000 steps by the CPU restatement (NBodyPropagator with its solout and fits; native build, one thread --
        # the reference steps this system on one thread, ephemeris_explorer/src/load/mod.rs:673-687), the last 1e6 timed
        from oracle import orc
        orc.build(native=True)
        o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree, native=True)
        assert o.step_n(20000) == 0
        t = time.perf_counter()
        assert o.step_n(1_000_000) == 0
        wc = time.perf_counter() - t
        c1["cpu_baseline"] = {"value": s.n * 1e6 / wc, "unit": "body-steps/s", "cores": 1, "kind": "port",
                              "sample": "the same 1e6 steps of the same system with solout and fits (oracle Propagator, -O2 "
                                        "-march=native, one thread)", "seconds": wc, "us_per_step": wc}
        pg, po = p.state(), o.state()
        c1["parity"] = {"max_abs_dpos": float(np.abs(pg[0] - po[0]).max()), "max_abs_dvel": float(np.abs(pg[1] - po[1]).max()),
                        "steps": int(po[3]), "vs": "oracle (port)"}
        # Latency roofline: a step of 32 bodies is ONE dependent chain through one workgroup, not throughput. Its shortest form,
        # priced with the measured dependent-issue interval of v_add_f64 on this part (5.9 cycles, profiles/r02_chain2_ubench.txt;
        # multiplies and fmas taken at the same figure -- optimistic) at the 2.4 GHz shader clock:
        #   pair term   25 dependent f64 operations (difference, three-term square, the IEEE square root's and reciprocal's
        #               refinement sequences, mu * inv, d * (...)) + v_rsq_f64 (16)
        #   row sums    31 ordered additions of the longest chain (bodies 0 and 31) + the two halves' addition
        #   predictor   a * w_beta[0], eleven more ordered additions, * h^2/beta_d, + sum1 = 15 (sum1's twelve run beside the pairs)
        #   two LDS round trips (positions out and in, contributions out and in): 2 x (write 64 + barrier + read 64)
        dep = 5.9
        floor = (25 * dep + 16) + 32 * dep + 15 * dep + 2 * 128
        ticks = w / 1e6 * 2.4e9
        c1["latency_roofline"] = {"floor_ticks_per_step": floor, "ticks_per_step": ticks, "frac": floor / ticks, "clock_ghz": 2.4,
                                  "floor": "(25 x 5.9 + 16) pair chain + 32 x 5.9 ordered row sum + 15 x 5.9 predictor + 2 x 128 LDS "
                                           "round trips; where the rest goes: profiles/r05_small_kernel_evidence.md"}
    except Exception as e:
        out["configs1_full_solar_system"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    try:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
        r = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--workload", "craft", "--steps", "5",
                            "--no-other-configs"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                           text=True, timeout=float(os.environ.get("EPH_BENCH_CRAFT_TIMEOUT", "240")))
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][-1]
        d = json.loads(line)
        out["configs3_craft_sweep"] = {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                                       "workload": d["config"]["workload"], "kernel": d["roofline"]["kernel"],
                                       "launch_us": d["roofline"]["launch_us"], "wall_over_kernel": d["ms_per_step"] * 1e3 / d["roofline"]["launch_us"],
                                       "fp64": d["roofline"]["fp64"], "traffic": d["roofline"].get("traffic"),
                                       "traffic_profile_commit": d["roofline"].get("traffic_profile_commit"),
                                       "algorithmic_bytes_per_launch": d["roofline"].get("algorithmic_bytes_per_launch"),
                                       "cpu_baseline": d.get("cpu_baseline"),
                                       "divergence": d["divergence"]["attempts_max_over_mean_per_wave"]}
    except Exception as e:
        out["configs3_craft_sweep"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher environment: become the launcher -- one rank per GPU through
    torch.distributed.run on 127.0.0.1 with a free port -- and pass rank 0's single JSON line through."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in lines[-1:]:
        print(ln, flush=True)
    if r.returncode or not lines:
        sys.stderr.write(r.stdout[-4000:])
        raise SystemExit(r.returncode or 1)


def timed_blocks(run_block, barrier, agree, blocks=0, min_region_s=0.05, max_blocks=64):
    """The K-step block, timed R times, each block bracketed by barrier + device sync on both sides. R = `blocks`, or
    (0) chosen after the first block so that the timed region is at least `min_region_s` (a 20-step block of a 40 us
    kernel is 0.8 ms: one scheduling hiccup moves it by percents), at least 3. `agree(t)` returns the max of t over the
    ranks so that every rank times the same number of blocks. Returns the sorted block times [s]."""
    times, want = [], blocks
    while True:
        barrier()
        t0 = time.perf_counter()
        run_block()
        barrier()
        times.append(time.perf_counter() - t0)
        if want <= 0:
            want = min(max_blocks, max(3, int(min_region_s / max(agree(times[0]), 1e-9)) + 1))
        if len(times) >= want:
            return sorted(times)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cpu-steps", type=int, default=60, help="oracle steps timed for cpu_baseline (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the companions of the default line (configs[1] in-process, the configs[3] sweep as a child process)")
    ap.add_argument("--bodies", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--workload", choices=["nbody", "craft", "nbody-sharded"], default="nbody",
                    help="nbody (default, the BASELINE metric: per-rank replicas) | craft: the massless sweep of "
                         "configs[3], sharded over the ranks (steps = sweeps of `--craft-days` days over `--craft` "
                         "spacecraft) | nbody-sharded: ONE system of --bodies (default 65536, configs[4] in f64) "
                         "partitioned by target body over the ranks, one RCCL all-gather per step (strong scaling)")
    ap.add_argument("--transport", choices=["rccl", "host", "peer"], default="rccl",
                    help="nbody-sharded exchange: rccl (ncclAllGather) | peer (direct writes into IPC-mapped peer "
                         "mailboxes, csrc/peer.hip) | host (staged through host memory; tests)")
    ap.add_argument("--blocks", type=int, default=0,
                    help="how many times the K-step block is timed (0 = as many as make the timed region >= 50 ms, at "
                         "least 3); ms_per_step is the MEDIAN block")
    ap.add_argument("--path", choices=["exact", "fast", "fast-rsq", "f32-pairs"], default="exact",
                    help="exact (default): the reference's summation order, bit-identical to the CPU path | fast: the "
                         "opt-in slice-parallel sums (EPH_PATH_FAST) -- a second, separately labelled line "
                         "(config.workload ..._fast) with its measured divergence from the reference order")
    ap.add_argument("--horizon", type=int, default=100000,
                    help="parity horizon: max |dpos| vs the oracle's committed positions at every 10^k-th step up to this "
                         "many steps (tests/golden/plummer4096_horizon.npz; 0 = skip)")
    ap.add_argument("--prewarm", type=float, default=1.0,
                    help="seconds of untimed load on a scratch clone before the warm-up steps, so the timed region runs at the "
                         "steady shader clock instead of the ramp from idle (0 = off)")
    ap.add_argument("--craft", type=int, default=262144)
    ap.add_argument("--craft-days", type=float, default=0.25)
    ap.add_argument("--population", choices=["transfer", "mixed"], default="transfer",
                    help="craft workload: transfer (SURVEY 8(d)4: one heliocentric arc +- 100 km) | mixed: LEO / GTO / lunar "
                         "transfer / heliocentric in equal numbers (step counts spanning > 10x)")
    ap.add_argument("--population-order", choices=["interleaved", "blocked"], default="interleaved")
    args = ap.parse_args()
    # `--gpus N` without a launcher environment: this process becomes the launcher (every workload)
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus > 1 and "RANK" not in os.environ:
        return self_launch(args.gpus)
    if args.workload == "craft":
        return craft_main(args)
    sharded = args.workload == "nbody-sharded"
    if args.bodies is None:
        args.bodies = 65536 if sharded else N_BODIES

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world

    import numpy as np
    import torch

    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd.workloads import plummer

    if not torch.cuda.is_available() or ea.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    local_rank %= max(ea.device_count(), 1)            # (lets the N > 1 flow be exercised on a one-GPU box)
    torch.cuda.set_device(local_rank)
    ea.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("EPH_BENCH_BACKEND", "nccl")   # "gloo" only to test this flow with ranks sharing a GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    n = args.bodies
    # rank r integrates its own replica (seed + r): independent systems, no exchange -- or, sharded, every rank
    # builds the SAME system and owns n/world target bodies of it
    pos, vel, mu = plummer(n, seed=20260926 + (0 if sharded else rank))
    fast = args.path in ("fast", "fast-rsq", "f32-pairs")
    fast_path = {"fast": ea.PATH_FAST, "fast-rsq": ea.PATH_FAST_RSQ, "f32-pairs": ea.PATH_F32_PAIRS}.get(args.path)
    if fast and sharded and args.path != "f32-pairs":
        raise SystemExit("--path fast / fast-rsq are single-device paths; on a target partition: exact or f32-pairs")
    g = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
    if fast:
        g.set_path(fast_path)
    if sharded:
        from ephemeris_explorer_amd.parallel import shard_nbody
        shard_nbody(g, dist, transport=args.transport, device="cuda")
    g.advance(12)                       # multistep start-up, reported separately in DESIGN.md
    if args.prewarm > 0:
        # The board idles at 100 MHz / 250 W and takes ~0.6 s of load to reach its 2.4 GHz shader clock
        # (profiles/r02_step_kernel_evidence.md section 1): a short timed region entered straight from idle measures the ramp,
        # not the kernel. Load the device first with a SCRATCH clone of the system (the measured handle's own history
        # stays exactly start-up + W + K steps); untimed, like the W warm-up steps that follow.
        scratch = g.clone()
        chunk = max(1, int(2000 * (N_BODIES / n) ** 2))            # ~0.1 s of work per chunk at any size
        t_pre = time.perf_counter()
        while True:
            scratch.advance(chunk)
            scratch.sync()
            out_of_time = time.perf_counter() - t_pre >= args.prewarm
            if sharded and dist is not None:
                # a sharded advance is a collective: every rank must run the same number of chunks (a per-rank clock let one rank
                # start a chunk its peers never entered -- "no data from rank 2 within the time limit", round 5)
                from ephemeris_explorer_amd.parallel import reduce_timing as _rt
                out_of_time = _rt(1.0 if out_of_time else 0.0, 0, dist, device="cuda")[1] > 0.0
            if out_of_time:
                break
        del scratch
    g.advance(args.warmup)
    g.enable_timing(True)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def block():
        g.advance(args.steps)           # EXACTLY K steps
        g.sync()

    from ephemeris_explorer_amd.parallel import reduce_timing
    blocks = timed_blocks(block, barrier, lambda t: reduce_timing(t, 0, dist, device="cuda")[1], blocks=args.blocks)
    elapsed = blocks[len(blocks) // 2]                   # the median block: K steps
    units_local = n * args.steps / world if sharded else n * args.steps
    units, elapsed = reduce_timing(elapsed, units_local, dist, device="cuda")       # sum of units, MAX of time
    _, t_min = reduce_timing(blocks[0], 0, dist, device="cuda")
    _, t_max = reduce_timing(blocks[-1], 0, dist, device="cuda")
    ms_kernel, launches = g.kernel_time()
    # The contract's region is ONE K-step block between synchronisations; a 20-step block of this kernel is 0.8 ms, of which the
    # launch ramp and the synchronisation are about 40 us (6 %). The same K steps ten times per synchronisation, once, beside it:
    long_reps = 10
    g.enable_timing(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(long_reps):
        g.advance(args.steps)
    g.sync()
    barrier()
    _, t_long = reduce_timing(time.perf_counter() - t0, 0, dist, device="cuda")
    want_strong = world > 1 and not sharded and not fast and n == N_BODIES
    out = None

    if rank == 0:
        value = units / elapsed
        launch_s = ms_kernel * 1e-3 / max(launches, 1)      # HIP events on the handle's stream, timed region only
        nt = n // world if sharded else n                   # target bodies one launch of this rank advances
        achieved_gbs = BYTES_PER_BODY_STEP * nt / launch_s / 1e9
        flops = (FLOP_PER_INTERACTION * (n - 1) + 231.0) * nt
        # HBM-side bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, collected and corrected
        # as MI355X_MICROARCH.md prescribes) of this same command, committed under profiles/ -- not measurable live
        traffic, traffic_src, valu_insts = None, None, None
        tj = ROOT / "profiles" / ("traffic_fast.json" if fast else "traffic.json")
        traffic_commit, traffic_stale = None, None
        if tj.exists() and n == N_BODIES and not sharded:                   # the committed PMC passes of this path
            from ephemeris_explorer_amd.workloads import profile_is_current
            tinfo = json.loads(tj.read_text())
            # the figures are only as good as the kernel they were counted on: the file carries the sha256 of the kernel's sources
            # (scripts/summarize_profile.py), recomputed here; a source that changed since drops them from the line
            current, why = profile_is_current(tinfo, "fast" if fast else "nbody")
            traffic_commit = tinfo.get("profile_commit")
            if current:
                traffic, traffic_src = tinfo.get("traffic_bytes_per_launch"), tinfo.get("source")
                valu_insts = tinfo.get("valu_wave_insts_per_launch")
            else:
                traffic_stale = f"{tj.name} not used: {why} (re-take with scripts/round_profile.sh + scripts/summarize_profile.py)"
        out = {
            "metric": "body-steps/s", "value": value, "unit": "body-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "blocks": len(blocks), "ms_per_step_min": t_min / args.steps * 1e3, "ms_per_step_max": t_max / args.steps * 1e3,
            "timing": f"the {args.steps}-step block timed {len(blocks)} times (barrier + device sync around each); "
                      "value and ms_per_step are the median block",
            "long_region": {"steps": long_reps * args.steps, "ms_per_step": t_long / (long_reps * args.steps) * 1e3,
                            "value": units / (t_long / long_reps),
                            "note": f"{long_reps} x the K-step block per synchronisation, one region: the block's ramp and "
                                    "synchronisation amortised (value / ms_per_step above stay the contract's single block)"},
            "pair_variant": ea.pair_variant(),
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": ("f32 pair arithmetic, f64 accumulation and integrator (mixed)" if args.path == "f32-pairs" else "f64"),
            "data": "synthetic",
            "config": ({"workload": (f"plummer_{n}_qt12_f32pairs, one system partitioned by target body (BASELINE.json configs[4] AS "
                                     "STATED: binary32 pair arithmetic on a shard; f64 accumulation in global slice order and f64 "
                                     "integrator; bit-identical to this library's single-device f32 path for any world size -- the "
                                     "reference has no f32 path; h=1/1024, seed 20260926)" if args.path == "f32-pairs" else
                                     f"plummer_{n}_f64_qt12, one system partitioned by target body "
                                     f"(BASELINE.json configs[4] in f64; h=1/1024, seed 20260926)"),
                        "bodies_per_gpu": nt, "method": "QuinlanTremaine12", "path": args.path,
                        "parallelism": f"target-partition x{world}, 1 all-gather of {(16 if args.path == 'f32-pairs' else 32) * n} B "
                                       f"per step ({args.transport})"} if sharded else
                       {"workload": (f"plummer_{n}_qt12_f32pairs (BASELINE.json configs[4] on one GPU; h=1/1024, seed 20260926+rank); "
                                     "OPT-IN mixed precision: pair arithmetic in binary32, f64 accumulation in slice order, f64 "
                                     "integrator -- the reference has no f32 path, no parity claim" if args.path == "f32-pairs" else
                                     f"plummer_{n}_f64_qt12{'_' + args.path.replace('-', '_') if fast else ''} (BASELINE.json configs[2]; h=1/1024, "
                                     "seed 20260926+rank)" + ("; OPT-IN fast path: slice-parallel partial sums, NOT the "
                                                              "reference's summation order" if fast else "")),
                        "bodies_per_gpu": n, "method": "QuinlanTremaine12", "parallelism": f"replicas x{world}",
                        "path": args.path}),
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_profile_commit": traffic_commit, "traffic_stale": traffic_stale,
                         "kernel": ("k_fast_partial_f32 + k_fast_finish<12>" if args.path == "f32-pairs" else
                                    "k_fast_partial + k_fast_finish<12>" if fast else
                                    f"k_lm_step_wg<12,{16 if nt > 2048 else 8 if nt > 1024 else 4}>" if nt > 512 else "k_lm_step<BPW,12>"),
                         "launch_us": launch_s * 1e6, "launches": launches,
                         "algorithmic_bytes_per_launch": BYTES_PER_BODY_STEP * nt,
                         # what actually binds the kernel: f64 VALU issue (`bound` stays "hbm" because the contract's roofline
                         # object offers hbm | mfma; the working set is cache resident, so that fraction is tiny by construction)
                         "binding": "fp64_valu",
                         "fp64": {"achieved": flops / launch_s / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": flops / launch_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS, "flop_per_launch": flops,
                                  "count": "20 (N - 1) + 231 flop per body-step (SURVEY 8(d)); the peak counts an FMA as two, "
                                           "and parity forbids contraction: the reachable roof is half of it"},
                         "note": "working set (912 B/body) is L2-resident; the path is f64-VALU bound, see roofline.fp64"
                                 + ("; launch_us includes the per-step all-gather" if sharded else "")},
        }
        if args.path == "f32-pairs":
            # the pair arithmetic of this opt-in path is binary32: the same flop count against the packed-f32 vector roof
            out["roofline"]["binding"] = "fp32_valu"
            out["roofline"]["fp32"] = {"achieved": flops / launch_s / 1e12, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                       "frac": flops / launch_s / 1e12 / FP32_VECTOR_PEAK_TFLOPS, "flop_per_launch": flops,
                                       "count": "the exact path's 20 (N - 1) + 231 flop per body-step (TFLOP/s-EQUIVALENT: the f32 "
                                                "loop issues 18 flop per interaction, v_rsq_f32 counted as one)"}
            out["roofline"]["note"] = "working set is L2-resident; the pair loop is f32-VALU bound (v_pk_* + v_rsq_f32), see roofline.fp32"
            del out["roofline"]["fp64"]                    # (an f64 figure would compare binary32 arithmetic with the f64 roof)
        else:
            out["fp64"] = dict(out["roofline"]["fp64"], bound="fp64_valu")   # (round 3's top-level key, kept for readers of older lines)
        if valu_insts and "fp64" in out["roofline"]:
            # issue-slot view of the same launch: wave64 VALU instructions (SQ_INSTS_VALU of the committed profile) x 64
            # lanes / live launch time, against 256 CU x 4 SIMD x 16 f64 lanes per clock at 2.4 GHz
            lane_ops = valu_insts * 64.0 / launch_s
            out["roofline"]["fp64"]["valu_issue"] = {"achieved": lane_ops / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS / 2.0,
                                                     "unit": "T lane-ops/s", "frac": lane_ops / 1e12 / (FP64_VECTOR_PEAK_TFLOPS / 2.0),
                                                     "valu_wave_insts_per_launch": valu_insts}
            out["fp64"]["valu_issue"] = out["roofline"]["fp64"]["valu_issue"]
        if world == 1 and not args.no_cpu_baseline and n != N_BODIES and not sharded:
            out["cpu_baseline"] = sharded_cpu_baseline(pos, mu, n)
        if world == 1 and not args.no_cpu_baseline and n == N_BODIES and args.path != "f32-pairs":
            base, o = cpu_baseline(pos, vel, mu, args.cpu_steps)
            out["cpu_baseline"] = base
            # parity beside the number: a fresh GPU run of the same steps vs the oracle
            c = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
            if fast:
                c.set_path(fast_path)
            nsteps = o.state()[3]                            # the oracle went on for the all-cores sample
            c.advance(nsteps)
            dp = np.abs(c.state()[0] - o.state()[0]).max()
            out["parity"] = {"max_abs_dpos": float(dp), "steps": int(nsteps), "vs": "oracle (port)"}
        if world == 1 and not args.no_cpu_baseline and sharded:
            out["cpu_baseline"] = sharded_cpu_baseline(pos, mu, n)
        if world == 1 and args.path == "f32-pairs":
            # what the mixed path is to be judged against: the build's own exact f64 path on the same system
            ex = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
            mx = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
            mx.set_path(fast_path)
            done, dv = 0, {}
            for k in (12 + 10, 12 + 100, 12 + 1000):
                if n > 16384 and k > 12 + 100:
                    break                                   # (an exact step of 65 536 bodies is 8 ms; keep the default run short)
                ex.advance(k - done); mx.advance(k - done)
                done = k
                dv[str(k - 12)] = float(np.abs(mx.state()[0] - ex.state()[0]).max())
            a_ex, a_mx = ex.acc(), mx.acc()
            out["parity"] = {"vs": "this library's exact f64 path on the same system (the reference has no f32 path)",
                             "max_abs_dpos_after_steady_steps": dv,
                             "max_rel_dacc": float(np.abs(a_mx - a_ex).max() / np.abs(a_ex).max())}
        fx = ROOT / "tests" / "golden" / "plummer4096_horizon.npz"
        if world == 1 and n == N_BODIES and not sharded and args.horizon > 0 and fx.exists() and args.path != "f32-pairs":
            # north_star: "positions within 1e-9 AU of the reference over 1e5 steps": max |dpos| against the oracle's
            # committed positions (generator tests/golden/make_plummer_horizon.py) at every 10^k-th step
            ref = np.load(fx)
            c = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
            if fast:
                c.set_path(fast_path)
            done, hz = 0, {}
            for k in (int(x) for x in ref["checkpoints"]):
                if k > args.horizon:
                    break
                c.advance(k - done)
                done = k
                hz[str(k)] = float(np.abs(c.state()[0] - ref[f"pos_{k}"]).max())
            out.setdefault("parity", {})["horizon_max_abs_dpos"] = hz
            out["parity"]["horizon_vs"] = "oracle positions, tests/golden/plummer4096_horizon.npz (N-body units: length scale 1)"
    if want_strong:
        # The strong-scaling leg runs AFTER the main line is complete and under a watchdog: a transport that hangs on hardware
        # it has never met (instead of raising) must not cost the run its number. On expiry rank 0 prints the line with the
        # leg marked as timed out and every rank leaves without waiting for the others.
        import threading
        limit = float(os.environ.get("EPH_BENCH_SHARDED_TIMEOUT", "420"))

        preflight = {}
        if rank == 0:
            out["transports"] = preflight          # (filled in place: what the preflight established, whatever happens next)

        def expired():
            if rank == 0:
                out.setdefault("sharded_4096", {"error": f"no result within {limit:.0f} s (EPH_BENCH_SHARDED_TIMEOUT)"})
                out.setdefault("configs4_f32_sharded", {"error": f"no result within {limit:.0f} s (EPH_BENCH_SHARDED_TIMEOUT)"})
                print(json.dumps(out), flush=True)
            os._exit(0)

        dog = threading.Timer(limit + (0.0 if rank == 0 else 5.0), expired)
        dog.daemon = True
        dog.start()
        try:
            nccl = os.environ.get("EPH_BENCH_BACKEND", "nccl") == "nccl"
            transport_preflight(dist, world, rank, nccl, preflight)
            strong = sharded_4096(dist, world, rank, args.steps, nccl, usable=preflight)
            # configs[4] as stated rides along when every rank has its own device (or when asked for: a rehearsal on a shared
            # device would spend its time slicing the GPU between the ranks' 65 536-body start-ups)
            c4_bodies = int(os.environ.get("EPH_BENCH_CONFIGS4_BODIES", "65536" if nccl else "0"))
            c4 = None
            if c4_bodies > 0 and c4_bodies % (64 * world) == 0:
                if rank == 0:
                    out["sharded_4096"] = dict(strong, replica_ms_per_step=elapsed / args.steps * 1e3)   # (kept if the next leg times out)
                c4 = configs4_f32_sharded(dist, world, rank, args.steps, nccl, preflight, c4_bodies)
        except BaseException as e:                       # (a peer gone, a collective torn down: the group is not usable any more)
            if rank == 0:
                out.setdefault("sharded_4096", {"error": f"{type(e).__name__}: {e}"[:300]})
                out.setdefault("configs4_f32_sharded", {"error": f"{type(e).__name__}: {e}"[:300]})
                print(json.dumps(out), flush=True)
            os._exit(0)
        dog.cancel()
        if rank == 0:
            strong["replica_ms_per_step"] = elapsed / args.steps * 1e3
            out["sharded_4096"] = strong
            if c4 is not None:
                out["configs4_f32_sharded"] = c4
    if rank == 0 and world == 1 and n == N_BODIES and not sharded and not fast and not args.no_other_configs:
        out["other_variants"] = other_variants(pos, vel, mu)
    if (rank == 0 and world == 1 and n == N_BODIES and not sharded and not fast and not args.no_cpu_baseline
            and not args.no_other_configs):
        out["other_configs"] = other_configs()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
