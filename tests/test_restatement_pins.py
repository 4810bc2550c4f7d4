"""CPU: known answers for the two restatements added in round 2 that the GPU tests lean on -- so that they are not only
checked against each other.

* ERKN / Tsitouras75Nystrom (integration/src/runge_kutta/nystrom/explicit.rs, methods.rs:1417-1520): on the Kepler problem
  of integration/examples/plot_work_precision.rs (point mass, closed-form orbit) the fixed-step method converges with its
  stated order 7 and its embedded error estimate with order 5+1.
* PlotPoints::new (ephemeris_explorer/src/ui/world/plot.rs:93-149): end points, monotone epochs, the max_points cut, and the
  property the loop enforces -- every accepted step's extrapolation error, seen from the camera, is within the target.
"""
import math

import numpy as np

from oracle import pyoracle as po


def kepler_rhs(t, y):
    r2 = y[0] * y[0] + y[1] * y[1] + y[2] * y[2]
    inv = 1.0 / (r2 * math.sqrt(r2))
    return [y[3], y[4], y[5], -y[0] * inv, -y[1] * inv, -y[2] * inv]


def kepler_error(method_cls, name, steps, ecc=0.3):
    """one revolution of an e = 0.3 orbit (mu = 1, a = 1) in `steps` fixed steps -> position error at the end"""
    y = [1.0 - ecc, 0.0, 0.0, 0.0, math.sqrt((1.0 + ecc) / (1.0 - ecc)), 0.0]      # periapsis
    rk = method_cls(name, y)
    h, t = 2.0 * math.pi / steps, 0.0
    est = 0.0
    for _ in range(steps):
        t, y = rk.advance(h, t, y, kepler_rhs)
        est = max(est, max(abs(e) for e in rk.error(h)[:3]))
    return math.dist(y[:3], [1.0 - ecc, 0.0, 0.0]), est


def test_erkn_converges_with_its_stated_orders():
    e1, est1 = kepler_error(po.Erkn, "Tsitouras75Nystrom", 60)
    e2, est2 = kepler_error(po.Erkn, "Tsitouras75Nystrom", 120)
    assert e1 < 1e-8 and e2 < 1e-10
    assert 2.0 ** 6 < e1 / e2 < 2.0 ** 8.5                  # global error of an order-7 method: ~2^7 per halving
    assert 2.0 ** 5 < est1 / est2 < 2.0 ** 7.5              # local error estimate of the order-5 embedded solution: ~h^6
    # the general Nystrom pair (Fine45, ERKNG; orders 4(5)) on the same problem, for scale
    f1, _ = kepler_error(po.Erkng, "Fine45", 60)
    f2, _ = kepler_error(po.Erkng, "Fine45", 120)
    assert f1 / f2 > 2.0 ** 4 and e1 < f1 / 20 and e2 < f2 / 20


def test_plot_points_properties():
    radius, omega = 1.0e5, 1.0e-3

    def evaluate(t):
        return (po.Vec(radius * math.cos(omega * t), radius * math.sin(omega * t), 0.0),
                po.Vec(-radius * omega * math.sin(omega * t), radius * omega * math.cos(omega * t), 0.0))
    cam = po.Vec(0.0, 0.0, 4.0e5)
    tmin, tmax = 100.0, 100.0 + 2.0 * math.pi / omega
    counts = []
    for res in (4e-3, 1e-3, 2.5e-4):
        kind, pts = po.plot_points_new(evaluate, tmin, tmax, cam, res, 100000)
        assert kind == "ok" and pts[0][0] == tmin and pts[-1][0] == tmax
        ts = [p[0] for p in pts]
        assert all(b > a for a, b in zip(ts, ts[1:]))
        assert pts[3][1] == tuple(po._f32(c) for c in evaluate(ts[3])[0])          # positions are the f32 of the evaluation
        target = res * res
        for a, b in zip(ts, ts[1:]):                                                # the invariant of the accept test
            pa, va = evaluate(a)
            err = po.angular_distance(cam, pa + va * (b - a), evaluate(b)[0]) / 16.0
            assert err <= target
        counts.append(len(pts))
    # error ~ (curvature * dt^2)^2 and the target is res^2: dt ~ sqrt(res), points ~ res^-1/2 -> x2 per quartering
    assert 1.6 < counts[1] / counts[0] < 2.4 and 1.6 < counts[2] / counts[1] < 2.4
    kind, pts = po.plot_points_new(evaluate, tmin, tmax, cam, 1e-3, 7)
    assert kind == "ok" and len(pts) == 7 and pts[-1][0] < tmax                     # max_points cuts the curve short
    assert po.plot_points_new(evaluate, tmin, tmax, cam, 1e-3, 0) == ("ok", [])
    assert po.plot_points_new(lambda t: None if t > 200.0 else evaluate(t), tmin, tmax, cam, 1e-3, 100)[0] == "err"
    # the window clamp of compute_plot_points_parallel
    assert po.plot_window((0.0, 100.0, 5), None, -50.0, 500.0, 0, 30.0) == (0.0, 100.0)
    assert po.plot_window((0.0, 100.0, 5), (20.0, 80.0, 3), -50.0, 500.0, 1, 30.0) == (30.0, 80.0)     # Start: from `current`
    assert po.plot_window((0.0, 100.0, 5), (20.0, 80.0, 3), -50.0, 500.0, 2, 30.0) == (20.0, 30.0)     # End: up to `current`
    assert po.plot_window((0.0, 100.0, 0), None, 0.0, 50.0, 0, 30.0) is None                            # empty trajectory
    assert po.plot_window((0.0, 100.0, 5), None, 60.0, 40.0, 0, 30.0) is None                           # min >= max
