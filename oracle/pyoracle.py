"""Second, independent restatement of the reference hot path in pure Python. TEST INFRASTRUCTURE ONLY.

Purpose: (a) catch transcription errors in oracle/eph_oracle.c (the two were written separately from the
reference text and must agree bit for bit in float mode), (b) with `num=mpmath.mpf` quantify how far any
correct f64 implementation may sit from the exact recurrence (justifies the parity tolerance).

Python floats are IEEE binary64 with round-to-nearest-even and no FMA contraction, i.e. the same
arithmetic the Rust reference performs. Reference citations are relative to /root/reference.
"""
import json
import math
from pathlib import Path

_GOLD = Path(__file__).resolve().parent.parent / "tests" / "golden" / "coeff_tables.json"
_tables = None


def tables():
    global _tables
    if _tables is None:
        _tables = json.loads(_GOLD.read_text())
    return _tables


def _ratio(nd):
    # Mul<Ratio> for f64: numer as f64 / denom as f64  (integration/src/ratio.rs:221-228)
    return float(int(nd[0])) / float(int(nd[1]))


class Vec(tuple):
    """3-vector with the component-wise operators glam::DVec3 provides."""

    __slots__ = ()

    def __new__(cls, x, y, z):
        return tuple.__new__(cls, (x, y, z))

    def __add__(self, o):
        return Vec(self[0] + o[0], self[1] + o[1], self[2] + o[2])

    def __sub__(self, o):
        return Vec(self[0] - o[0], self[1] - o[1], self[2] - o[2])

    def __mul__(self, s):
        return Vec(self[0] * s, self[1] * s, self[2] * s)

    def __truediv__(self, s):
        return Vec(self[0] / s, self[1] / s, self[2] / s)

    def __neg__(self):
        return Vec(-self[0], -self[1], -self[2])


def accel_paired(pi, mui, pj, muj, sqrt=math.sqrt):
    """`particular` acceleration_paired with softening 0 -- source absent, PARITY UNPINNED (see eph_oracle.c)."""
    d = pj - pi
    n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
    inv = 1 / (n2 * sqrt(n2))
    return d * (muj * inv), (-d) * (mui * inv)


def gravity(y, mu, zero, sqrt=math.sqrt):
    """NewtonianGravity::eval into a fresh zeroed vector (ephemeris/src/propagators/nbody.rs:22-38)."""
    n = len(y)
    ddy = [Vec(zero, zero, zero) for _ in range(n)]
    for i in range(n):
        out = Vec(zero, zero, zero)
        for j in range(i + 1, n):
            ai, aj = accel_paired(y[i], mu[i], y[j], mu[j], sqrt)
            out = out + ai
            ddy[j] = ddy[j] + aj
        ddy[i] = ddy[i] + out
    return ddy


class Srkn:
    """SRKN<C, V> (integration/src/runge_kutta/nystrom/symplectic.rs:36-102)."""

    def __init__(self, name, num=float):
        t = tables()["methods"][name]
        self.A = [num(_ratio(r)) for r in t["A"]["ratio"]]
        self.B = [num(_ratio(r)) for r in t["B"]["ratio"]]
        self.fsal = t["FSAL"]
        self.i = 0
        self.ddy = None

    def advance(self, h, p):
        for s in range(len(self.A)):
            if not self.fsal or s > 0 or self.i == 0:
                self.ddy = p.eval(p.y)
            hb, ha = h * self.B[s], h * self.A[s]
            p.dy = [v + a * hb for v, a in zip(p.dy, self.ddy)]
            p.y = [y + v * ha for y, v in zip(p.y, p.dy)]
        p.time = p.time + h
        self.i += 1


class Problem:
    def __init__(self, pos, vel, mu, t0, num=float, sqrt=math.sqrt):
        self.num, self.sqrt = num, sqrt
        self.y = [Vec(*(num(c) for c in r)) for r in pos]
        self.dy = [Vec(*(num(c) for c in r)) for r in vel]
        self.mu = [num(m) for m in mu]
        self.time = num(t0)
        self.evals = 0

    def eval(self, y):
        self.evals += 1
        return gravity(y, self.mu, self.num(0), self.sqrt)


class LinearMultistep2:
    """LinearMultistep<ELM2 coefficients, f64, Substepper<4, BlanesMoan6B>> (integration/src/multistep/mod.rs,
    second_order/mod.rs, second_order/cowell.rs, buffer.rs; aliases methods.rs:37-40).

    The ring is kept as an explicit list of 'levels' newest-first instead of the reference's head-index ring,
    so an indexing slip in the C restatement cannot be shared."""

    def __init__(self, name, h, problem, num=float):
        t = tables()
        m = t["methods"][name]
        self.order = int(m["ORDER"])
        self.wa = [num(float(-int(a))) for a in m["ALPHA"][1:]]
        self.wb = [num(float(int(b))) for b in m["BETA_N"][1:]]
        self.inv_bd = num(1.0 / float(int(m["BETA_D"])))
        c = t["cowell"][f"Cowell<{self.order}>"]
        self.cw = [num(float(int(b))) for b in c["BETA_N"]]
        self.inv_cd = num(1.0 / float(int(c["BETA_D"])))
        self.h = num(h)
        self.hs = self.h * num(1.0 / 4.0)
        self.starter = Srkn("BlanesMoan6B", num)
        self.p = problem
        self.past = []           # [(y, a)] newest first; at most order-1 entries are ever read
        self.cur_a = None
        self.lm_i = 0

    def step_count(self):
        return self.starter.i // 4 + self.lm_i

    def advance(self):
        p = self.p
        if self.starter.i // 4 < self.order:
            if self.starter.i == 0:
                self.cur_a = p.eval(p.y)                       # advance_with(no-op)
            # advance_with(starter): the level being left becomes the newest past level
            self.past.insert(0, (list(p.y), self.cur_a))
            del self.past[self.order - 1:]
            for _ in range(4):
                self.starter.advance(self.hs, p)
            self.cur_a = p.eval(p.y)
            return
        h = self.h
        levels = [(p.y, self.cur_a)] + self.past             # j = 0 .. order-1
        assert len(levels) == self.order
        n = len(p.y)
        z = p.num(0)
        s1 = [Vec(z, z, z)] * n
        s2 = [Vec(z, z, z)] * n
        for j, (yy, aa) in enumerate(levels):
            s1 = [s + y * self.wa[j] for s, y in zip(s1, yy)]
            s2 = [s + a * self.wb[j] for s, a in zip(s2, aa)]
        y_prev = p.y
        self.past.insert(0, (p.y, self.cur_a))
        del self.past[self.order - 1:]
        hh = h * h * self.inv_bd
        p.y = [a + b * hh for a, b in zip(s1, s2)]
        p.time = p.time + h
        self.cur_a = p.eval(p.y)
        w = [Vec(z, z, z)] * n
        for j, aa in enumerate([self.cur_a] + [lv[1] for lv in self.past]):
            w = [s + a * self.cw[j] for s, a in zip(w, aa)]
        hc = h * self.inv_cd
        p.dy = [(y - ym) / h + ww * hc for y, ym, ww in zip(p.y, y_prev, w)]
        self.lm_i += 1


def horner(coeffs, t, zero):
    r = zero
    for c in reversed(coeffs):
        r = r * t + c
    return r


def least_squares_fit(degree, ts, xs):
    """LeastSquaresFit::interpolate (ephemeris_explorer/src/dynamics/celestial.rs:24-135), scalars for the
    quantities whose DVec3 components coincide. Returns the trimmed coefficient list of Vec."""
    m = len(ts)
    d0 = Vec(0.0, 0.0, 0.0)
    g0 = 0.0
    b0 = 0.0
    for t, x in zip(ts, xs):
        d0 = d0 + x
        g0 += 1.0
        b0 += t
    if g0 == 0.0:
        return None
    degree = min(degree, m - 1)
    b0 /= g0
    d0 = d0 / g0
    if degree == 0:
        return [d0]
    pdata = [Vec(0.0, 0.0, 0.0) for _ in range(degree + 1)]
    p_km1 = [0.0] * (degree + 2)
    p_k = [0.0] * (degree + 2)
    pdata[0] = d0
    p_k[0] = 1.0
    g_k, b_k, mc_k, kp1 = g0, b0, 0.0, 1
    while True:
        for i in range(kp1):
            p_km1[i] = mc_k * p_km1[i] - b_k * p_k[i]
        for i in range(kp1):
            p_km1[i + 1] += p_k[i]
        d = Vec(0.0, 0.0, 0.0)
        g = 0.0
        b = 0.0
        for t, x in zip(ts, xs):
            px = horner(p_km1[: kp1 + 1], t, 0.0)
            d = d + x * px
            pp = px * px
            g += pp
            b += t * pp
        if g == 0.0:
            break
        d = d / g
        for i in range(kp1 + 1):
            pdata[i] = pdata[i] + d * p_km1[i]
        if kp1 == degree:
            break
        b /= g
        kp1 += 1
        b_k = b
        mc_k = -(g / g_k)
        g_k = g
        p_k, p_km1 = p_km1, p_k
    while pdata and pdata[-1] == (0.0, 0.0, 0.0):
        pdata.pop()
    return pdata


def spline_eval(start, interval, polys, at):
    """UniformSpline::state_vector (ephemeris/src/trajectory.rs:459-470,551-561,600-617)."""
    local = at - start
    if math.copysign(1.0, local) < 0 or local > interval * float(len(polys)):
        return None
    idx = max(int(math.ceil(local / interval)) - 1, 0)
    if idx >= len(polys):
        return None
    tau = (local - interval * float(idx)) / interval
    c = polys[idx]
    zero = Vec(0.0, 0.0, 0.0)
    first = c[0] if c else zero
    last = c[-1] if c else zero
    e, d = last, last
    for ci in list(reversed(c[1:]))[1:]:
        e = e * tau + ci
        d = d * tau + e
    e = e * tau + first
    return e, d / interval


class Propagator:
    """NBodyPropagator + SplineInterpolators solout (ephemeris/src/propagators/nbody.rs:243-517)."""

    def __init__(self, pos, vel, mu, t0, dt, direction, count, degree, method="QuinlanTremaine12"):
        self.p = Problem(pos, vel, mu, t0)
        self.dir = 1 if direction > 0 else -1
        self.dt = dt
        self.integ = LinearMultistep2(method, abs(dt) * self.dir, self.p)
        self.period = [dt * float(c) for c in count]
        self.degree = list(degree)
        self.last = [0.0] * len(mu)
        self.window = [[self.p.y[b]] for b in range(len(mu))]
        self.taus = [i / 8.0 if self.dir > 0 else 1.0 - i / 8.0 for i in range(9)]
        self.solution = self._new_solution()

    def _new_solution(self):
        out = []
        for b in range(len(self.period)):
            tm = self.last[b] + self.period[b] * float(len(self.window[b]) - 1)
            start = self.p.time + (-tm) if self.dir > 0 else self.p.time - (-tm)
            out.append({"start": start, "interval": self.period[b] * 8.0, "polys": []})
        return out

    def step(self):
        self.integ.advance()
        for b in range(len(self.period)):
            self.last[b] += self.dt
            if self.last[b] == self.period[b]:
                self.last[b] = 0.0
                self.window[b].append(self.p.y[b])
                if len(self.window[b]) == 9:
                    poly = least_squares_fit(self.degree[b], self.taus, self.window[b])
                    s = self.solution[b]
                    if self.dir > 0:
                        s["polys"].append(poly)
                    else:
                        s["polys"].insert(0, poly)
                        s["start"] -= s["interval"]
                    self.window[b] = [self.window[b][8]]

    def take_solution(self):
        old, self.solution = self.solution, self._new_solution()
        return old
