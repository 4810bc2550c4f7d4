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


PAIR_VARIANT = 0   # evaluation order of the unpinned point-mass term, same numbering as eph_oracle.c / pair_term.h


def set_pair_variant(v):
    global PAIR_VARIANT
    PAIR_VARIANT = int(v)


def point_mass_term(d, n2, mu, sqrt=math.sqrt):
    """Acceleration of a point mass mu along d (n2 = |d|^2) in the selected operation order (eph_oracle.c
    point_mass_term): 0-3 one reciprocal then d * (mu * inv); 4 (d * mu) / p; 5 d * (mu / p); 6 (d / p) * mu,
    p = n2 * sqrt(n2), vector / scalar component-wise as glam's DVec3 / f64."""
    v = PAIR_VARIANT
    if v == 4:
        return (d * mu) / (n2 * sqrt(n2))
    if v == 5:
        return d * (mu / (n2 * sqrt(n2)))
    if v == 6:
        return (d / (n2 * sqrt(n2))) * mu
    if v == 1:
        r = sqrt(n2)
        inv = 1 / (r * r * r)
    elif v == 2:
        s = 1 / sqrt(n2)
        inv = s * s * s
    elif v == 3:
        inv = (1 / n2) * (1 / sqrt(n2))
    else:
        inv = 1 / (n2 * sqrt(n2))
    return d * (mu * inv)


def accel_paired(pi, mui, pj, muj, sqrt=math.sqrt):
    """`particular` acceleration_paired with softening 0 -- source absent, PARITY UNPINNED (see eph_oracle.c)."""
    d = pj - pi
    n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
    return point_mass_term(d, n2, muj, sqrt), point_mass_term(-d, n2, mui, sqrt)


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
        """FixedRungeKuttaIntegrator::advance (runge_kutta/mod.rs:112-125) around SRKN::advance: 0, or the StepError (3 = BoundReached,
        1 = StepSizeUnderflow) with the problem untouched"""
        if p.time >= p.bound:
            return 3
        if p.time + h == p.time:
            return 1
        for s in range(len(self.A)):
            if not self.fsal or s > 0 or self.i == 0:
                self.ddy = p.eval(p.y)
            hb, ha = h * self.B[s], h * self.A[s]
            p.dy = [v + a * hb for v, a in zip(p.dy, self.ddy)]
            p.y = [y + v * ha for y, v in zip(p.y, p.dy)]
        p.time = p.time + h
        self.i += 1
        return 0


class Problem:
    def __init__(self, pos, vel, mu, t0, num=float, sqrt=math.sqrt):
        self.num, self.sqrt = num, sqrt
        self.y = [Vec(*(num(c) for c in r)) for r in pos]
        self.dy = [Vec(*(num(c) for c in r)) for r in vel]
        self.mu = [num(m) for m in mu]
        self.time = num(t0)
        self.evals = 0
        self.bound = math.inf                 # NBodyProblem::bound (f64::INFINITY until set_bound)

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
        """LinearMultistepIntegrator::advance (multistep/mod.rs:194-224): 0 or the StepError. An error out of the starter's sub-steps
        leaves the ring pushed and the sub-steps before the failing one applied, exactly where `?` leaves them in the reference."""
        p = self.p
        if p.time >= p.bound:
            return 3
        if p.time + self.h == p.time:
            return 1
        if self.starter.i // 4 < self.order:
            if self.starter.i // 4 == 0:
                # `if self.starter.step_count() == 0 { lm.advance_with(no-op) }` (multistep/mod.rs:212-214): the STARTER's macro-step
                # count, so a first macro step abandoned between two sub-steps (StepSizeUnderflow) is re-entered through here. It
                # pushes the ring like every advance_with; in an undisturbed run that entry is the one that falls off the ring's end
                # before the multistep formula first reads it. (Until round 5 this read `self.starter.i == 0` and pushed nothing: the
                # same bits everywhere except in that abandoned first step -- found by tests/test_oracle.py::
                # test_underflow_inside_the_starter_c_vs_python against the C restatement, which had it right.)
                self.past.insert(0, (list(p.y), self.cur_a))
                del self.past[self.order - 1:]
                self.cur_a = p.eval(p.y)
            # advance_with(starter): the level being left becomes the newest past level
            self.past.insert(0, (list(p.y), self.cur_a))
            del self.past[self.order - 1:]
            for _ in range(4):                                 # SubstepperIntegrator::advance  :98-108
                st = self.starter.advance(self.hs, p)
                if st:
                    return st
            self.cur_a = p.eval(p.y)
            return 0
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
        return 0


class Double:
    """Double<T> of the reference's convergence test (ephemeris/tests/solar_system_convergence.rs:12-110): a value
    with a running error term; Add/Sub are compensated (two_sum + fast_two_sum), Mul/Div by f64 act on both parts.
    One instance per DVec3 component (every operator the steppers use is component-wise)."""

    __slots__ = ("value", "error")

    def __init__(self, value, error=0.0):
        self.value, self.error = value, error

    @staticmethod
    def two_sum(a, b):            # :30-40
        value = a + b
        v = value - a
        return value, (a - (value - v)) + (b - v)

    @staticmethod
    def fast_two_sum(a, b):       # :50-59
        value = a + b
        return Double(value, b - (value - a))

    def __add__(self, o):         # :62-73
        sv, se = Double.two_sum(self.value, o.value)
        return Double.fast_two_sum(sv, (se + self.error) + o.error)

    def __sub__(self, o):         # :75-86 (two_sub(a, b) = two_sum(a, -b) :42-48)
        sv, se = Double.two_sum(self.value, -o.value)
        return Double.fast_two_sum(sv, (se + self.error) - o.error)

    def __mul__(self, r):         # :88-98
        return Double(self.value * r, self.error * r)

    def __truediv__(self, r):     # :100-110
        return Double(self.value / r, self.error / r)


class DoubleProblem:
    """NBodyProblem<Vec<Double<DVec3>>> + the test's NewtonianGravity (solar_system_convergence.rs:112-216): the force
    reads `.value` and accumulates into `.value`; error parts of the accelerations stay zero."""

    def __init__(self, pos, vel, mu, t0):
        self.y = [Vec(*(Double(float(c)) for c in r)) for r in pos]
        self.dy = [Vec(*(Double(float(c)) for c in r)) for r in vel]
        self.mu = [float(m) for m in mu]
        self.time = float(t0)
        self.evals = 0
        self.bound = math.inf

    @staticmethod
    def num(v):
        return Double(float(v))

    def eval(self, y):
        self.evals += 1
        vals = [Vec(c[0].value, c[1].value, c[2].value) for c in y]
        return [Vec(Double(a[0]), Double(a[1]), Double(a[2])) for a in gravity(vals, self.mu, 0.0)]


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
        st = self.integ.advance()                              # Integration::advance: Err(..)? before the solout
        if st:
            return st
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
        return 0

    def take_solution(self):
        old, self.solution = self.solution, self._new_solution()
        return old


# =====================================================================================================
# Massless path (independent restatement; float mode only)
# =====================================================================================================
def spline_position(start, interval, polys, at):
    """UniformSpline::position: eval_slice_horner (ephemeris/src/trajectory.rs:398-410,459-462,551-561)."""
    local = at - start
    if math.copysign(1.0, local) < 0 or local > interval * float(len(polys)):
        return None
    idx = max(int(math.ceil(local / interval)) - 1, 0)
    if idx >= len(polys):
        return None
    tau = (local - interval * float(idx)) / interval
    r = Vec(0.0, 0.0, 0.0)
    for c in reversed(polys[idx]):
        r = r * tau + c
    return r


class Erk:
    """ERK + embedded error (integration/src/runge_kutta/explicit.rs:54-141)."""

    def __init__(self, name, state):
        t = tables()["methods"][name]
        self.A = [[_ratio(r) for r in row] for row in t["A"]["ratio"]]
        self.B = [_ratio(r) for r in t["B"]["ratio"]]
        self.C = [_ratio(r) for r in t["C"]["ratio"]]
        self.E = [_ratio(r) for r in t["E"]["ratio"]]
        self.fsal = t["FSAL"]
        self.lower = min(int(t["ORDER"]), int(t["ORDER_EMBEDDED"]))
        self.i = 0
        self.k = [list(state) for _ in self.B]

    def advance(self, h, t, y, f):
        S = len(self.B)
        for s in range(S):
            if self.fsal and s == 0 and self.i > 0:
                self.k[0], self.k[S - 1] = self.k[S - 1], self.k[0]
                continue
            ti = t + h * self.C[s]
            yi = list(y)
            for j in range(s):
                ha = h * self.A[s][j]
                yi = [a + kk * ha for a, kk in zip(yi, self.k[j])]
            out = f(ti, yi)
            if out is None:                      # `self.k[s].zero()` ran before eval returned Err  explicit.rs:92
                self.k[s] = [0.0] * len(yi)
                return None
            self.k[s] = out
        for i in range(S):
            hb = h * self.B[i]
            y = [a + kk * hb for a, kk in zip(y, self.k[i])]
        self.i += 1
        return t + h, y

    def error(self, h):
        e = [0.0] * len(self.k[0])
        for i in range(len(self.B)):
            he = h * self.E[i]
            e = [a + kk * he for a, kk in zip(e, self.k[i])]
        return e


class Erkng:
    """ERKNG + embedded error on SecondOrderState<[DVec3; 1]> (nystrom/explicit_generalized.rs:59-170); state = y(3) + dy(3),
    k[s] = dk[s] (3 values). Same interface as Erk; f returns the 6-vector derivative, whose last three are ddy."""

    def __init__(self, name, state):
        t = tables()["methods"][name]
        self.AP = [[_ratio(r) for r in row] for row in t["AP"]["ratio"]]
        self.AV = [[_ratio(r) for r in row] for row in t["AV"]["ratio"]]
        self.BP, self.BV = [_ratio(r) for r in t["BP"]["ratio"]], [_ratio(r) for r in t["BV"]["ratio"]]
        self.EP, self.EV = [_ratio(r) for r in t["EP"]["ratio"]], [_ratio(r) for r in t["EV"]["ratio"]]
        self.C = [_ratio(r) for r in t["C"]["ratio"]]
        self.fsal = t["FSAL"]
        self.lower = min(int(t["ORDER"]), int(t["ORDER_EMBEDDED"]))
        self.i = 0
        self.k = [list(state[3:]) for _ in self.C]

    def advance(self, h, t, state, f):
        S = len(self.C)
        y, dy = list(state[:3]), list(state[3:])
        for s in range(S):
            if self.fsal and s == 0 and self.i > 0:
                self.k[0], self.k[S - 1] = self.k[S - 1], self.k[0]
                continue
            ti = t + h * self.C[s]
            hc = h * self.C[s]
            yi = [a + v * hc for a, v in zip(y, dy)]
            dyi = list(dy)
            for j in range(s):
                hhap, hav = h * h * self.AP[s][j], h * self.AV[s][j]
                yi = [a + kk * hhap for a, kk in zip(yi, self.k[j])]
                dyi = [a + kk * hav for a, kk in zip(dyi, self.k[j])]
            out = f(ti, yi + dyi)
            if out is None:                      # the zeroed dk[s] stays
                self.k[s] = [0.0] * 3
                return None
            self.k[s] = out[3:]
        y = [a + v * h for a, v in zip(y, dy)]
        for i in range(S):
            hhbp, hbv = h * h * self.BP[i], h * self.BV[i]
            y = [a + kk * hhbp for a, kk in zip(y, self.k[i])]
            dy = [a + kk * hbv for a, kk in zip(dy, self.k[i])]
        self.i += 1
        return t + h, y + dy

    def error(self, h):
        ey, edy = [0.0] * 3, [0.0] * 3
        for i in range(len(self.C)):
            hhep, hev = h * h * self.EP[i], h * self.EV[i]
            ey = [a + kk * hhep for a, kk in zip(ey, self.k[i])]
            edy = [a + kk * hev for a, kk in zip(edy, self.k[i])]
        return ey + edy


class Erkn(Erkng):
    """ERKN + embedded error (integration/src/runge_kutta/nystrom/explicit.rs:53-157; Tsitouras75Nystrom): y'' = f(t, y).
    Stage positions only -- the right-hand side is shown the unchanged current velocity (a SecondOrderODE never reads
    it). Update and error are ERKNG's (inherited)."""

    def __init__(self, name, state):
        t = tables()["methods"][name]
        self.A = [[_ratio(r) for r in row] for row in t["A"]["ratio"]]
        self.BP, self.BV = [_ratio(r) for r in t["BP"]["ratio"]], [_ratio(r) for r in t["BV"]["ratio"]]
        self.EP, self.EV = [_ratio(r) for r in t["EP"]["ratio"]], [_ratio(r) for r in t["EV"]["ratio"]]
        self.C = [_ratio(r) for r in t["C"]["ratio"]]
        self.fsal = t["FSAL"]
        self.lower = min(int(t["ORDER"]), int(t["ORDER_EMBEDDED"]))
        self.i = 0
        self.k = [list(state[3:]) for _ in self.C]

    def advance(self, h, t, state, f):
        S = len(self.C)
        y, dy = list(state[:3]), list(state[3:])
        for s in range(S):
            if self.fsal and s == 0 and self.i > 0:
                self.k[0], self.k[S - 1] = self.k[S - 1], self.k[0]
                continue
            ti = t + h * self.C[s]
            hc = h * self.C[s]
            yi = [a + v * hc for a, v in zip(y, dy)]
            for j in range(s):
                hha = h * h * self.A[s][j]
                yi = [a + kk * hha for a, kk in zip(yi, self.k[j])]
            out = f(ti, yi + list(dy))
            if out is None:                      # the zeroed dk[s] stays
                self.k[s] = [0.0] * 3
                return None
            self.k[s] = out[3:]
        y = [a + v * h for a, v in zip(y, dy)]
        for i in range(S):
            hhbp, hbv = h * h * self.BP[i], h * self.BV[i]
            y = [a + kk * hhbp for a, kk in zip(y, self.k[i])]
            dy = [a + kk * hbv for a, kk in zip(dy, self.k[i])]
        self.i += 1
        return t + h, y + dy


class Craft:
    """SpacecraftPropagator with an adaptive ERK pair and the CubicHermiteSpline solout
    (ephemeris/src/propagators/spacecraft.rs:415-695, integration/src/runge_kutta/mod.rs:188-285,396-440,
    ephemeris_explorer/src/dynamics/spacecraft.rs:70-74,218-293,609-641). `eph` = list of dicts
    {start, interval, polys} per body; burns = (start, end, acc, ref_index or -1)."""

    EMIN, EMAX = -1.7976931348623157e308, 1.7976931348623157e308

    def __init__(self, eph, mu, t0, pos, vel, method, tol, burns, h_init=60.0, n_max=1_000_000, soi=None):
        self.eph, self.mu, self.method, self.tol = eph, list(mu), method, tol
        self.soi = soi                      # SpacecraftSolout: sphere-of-influence radii, or None
        self.h_init, self.n_max = h_init, n_max
        self.fac_min, self.fac_max, self.fac, self.h_max = 1.0 / 5.0, 5.0 / 1.0, 9.0 / 10.0, self.EMAX
        self.tol_pos = self.tol_vel = tol        # AbsTol: position / velocity tolerances (dynamics/spacecraft.rs:609-641); settable
        segs, cursor = [], self.EMIN
        for s, e, acc, ref in sorted(burns, key=lambda b: b[0]):
            if s > cursor:
                segs.append((cursor, s, None, -1))
            cursor = e
            segs.append((s, e, Vec(*acc), ref))
        if cursor < self.EMAX:
            segs.append((cursor, self.EMAX, None, -1))
        self.segs = segs
        self.t = t0
        self.y = list(pos) + list(vel)
        self.cur = next(i for i, sg in enumerate(segs) if not sg[1] <= t0)
        self.bound = segs[self.cur][1]
        self.reset()
        self.knots = [(self.t, tuple(self.y))]
        self.transitions, self.apsides = [], []      # [(time, body)], [(time, distance, body, kind)]
        if soi is not None:                 # new_solution  dynamics/spacecraft.rs:525-537
            cur = self._soi_at_except(self.t, Vec(*self.y[:3]), -1)
            if cur is not None:
                self.transitions.append((self.t, cur))

    # ---- SpacecraftSolout (ephemeris_explorer/src/dynamics/spacecraft.rs:77-221,296-451,539-586) ----
    def _body_pos(self, b, t):
        e = self.eph[b]
        return spline_position(e["start"], e["interval"], e["polys"], t)

    def _soi_at_except(self, t, position, except_):
        best = None
        for b in range(len(self.eph)):
            if b == except_:
                continue
            bp = self._body_pos(b, t)
            if bp is None:
                continue
            d = position - bp
            d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
            if d2 < self.soi[b] * self.soi[b] and (best is None or d2 < best[0]):
                best = (d2, b)
        return None if best is None else best[1]

    @staticmethod
    def _signum(x):
        return x if x != x else math.copysign(1.0, x)

    def _zero_crossing(self, f, t0, t1):
        f0, f1 = f(t0), f(t1)
        if f0 is None or f1 is None or self._signum(f0) == self._signum(f1):
            return None
        x0, x1, g0 = t0, t1, f0
        for _ in range(100):
            mid = x0 + (x1 - x0) / 2.0
            fm = f(mid)
            if self._signum(g0) != self._signum(fm):
                x1 = mid
            else:
                x0, g0 = mid, fm
            if abs(x1 - x0) < 1e-3:
                return x0, math.copysign(1.0, f0) < 0.0           # (time, ascending)
        return None

    def _tr_insert(self, time, body):
        ts = [t for t, _ in self.transitions]
        import bisect
        i = bisect.bisect_left(ts, time)
        if i < len(ts) and ts[i] == time:
            self.transitions[i] = (time, body)
        elif i > 0 and self.transitions[i - 1][1] == body:
            pass
        else:
            self.transitions.insert(i, (time, body))

    def _events(self):
        (t0, y0), (t1, y1) = self.knots[-2], self.knots[-1]
        p0, p1, d0, d1 = Vec(*y0[:3]), Vec(*y1[:3]), Vec(*y0[3:]), Vec(*y1[3:])
        dt = t1 - t0
        r1 = 1.0 / dt
        r2 = r1 * r1
        r3 = r1 * r2
        dv = p1 - p0
        a2 = dv * r2 * 3.0 - (d0 * 2.0 + d1) * r1
        a3 = dv * r3 * -2.0 + (d0 + d1) * r2

        def pos(t):
            s = t - t0
            return (((a3 * s + a2) * s) + d0) * s + p0

        def vel(t):
            s = t - t0
            return ((a3 * s * 3.0 + a2 * 2.0) * s) + d0

        def f_soi(b):
            def f(t):
                bp = self._body_pos(b, t)
                if bp is None:
                    return None
                d = pos(t) - bp
                return (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) - self.soi[b] * self.soi[b]
            return f

        def f_radial(b):
            def f(t):
                e = self.eph[b]
                sv = spline_eval(e["start"], e["interval"], e["polys"], t)
                if sv is None:
                    return None
                rp, rv = pos(t) - sv[0], vel(t) - sv[1]
                return rp[0] * rv[0] + rp[1] * rv[1] + rp[2] * rv[2]
            return f

        for b in range(len(self.eph)):
            ev = self._zero_crossing(f_soi(b), t0, t1)
            if ev is None:
                continue
            time, ascending = ev
            if not ascending:
                self._tr_insert(time, b)
            else:
                entered = self._soi_at_except(time, pos(time), b)
                if entered is not None:
                    self._tr_insert(time, entered)
        ts = [t for t, _ in self.transitions]
        import bisect
        i = bisect.bisect_left(ts, t0)
        i0 = i if (i < len(ts) and ts[i] == t0) else max(i - 1, 0)
        for i in range(i0, len(self.transitions)):
            t, soi = self.transitions[i]
            ta = max(t, t0)
            tb = self.transitions[i + 1][0] if i + 1 < len(self.transitions) else t1
            ev = self._zero_crossing(f_radial(soi), ta, tb)
            if ev is None:
                continue
            time, ascending = ev
            bp = self._body_pos(soi, time)
            if bp is None:
                continue
            d = bp - pos(time)
            rec = (time, math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), soi, 0 if ascending else 1)
            at = [a[0] for a in self.apsides]
            j = bisect.bisect_left(at, time)
            if j < len(at) and at[j] == time:
                self.apsides[j] = rec
            else:
                self.apsides.insert(j, rec)

    def reset(self):
        tab = tables()["methods"][self.method]
        self.rk = (Erkng if "AP" in tab else Erkn if "BP" in tab else Erk)(self.method, self.y)
        self.next_h = self.h_init
        self.n = 0

    def rhs(self, t, y):
        pos, vel = Vec(*y[:3]), Vec(*y[3:])
        acc = Vec(0.0, 0.0, 0.0)
        for b, e in enumerate(self.eph):
            bp = spline_position(e["start"], e["interval"], e["polys"], t)
            if bp is None:
                return None
            d = bp - pos
            n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
            acc = acc + point_mass_term(d, n2, self.mu[b])
        man = Vec(0.0, 0.0, 0.0)
        _, _, bacc, ref = self.segs[self.cur]
        if bacc is not None:
            if ref >= 0:
                e = self.eph[ref]
                sv = spline_eval(e["start"], e["interval"], e["polys"], t)
                if sv is None:
                    return None
                rp, rv = pos - sv[0], vel - sv[1]

                def norm(v):
                    rcp = 1.0 / math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
                    return v * rcp if math.isfinite(rcp) and rcp > 0.0 else None

                def cross(a, b):
                    return Vec(a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1])

                x = norm(rv)
                yv = norm(cross(rp, rv)) if x is not None else None
                if x is None or yv is None:
                    return None
                xy = cross(x, yv)
                z = xy * (1.0 / math.sqrt(xy[0] * xy[0] + xy[1] * xy[1] + xy[2] * xy[2]))
                man = x * bacc[0] + z * bacc[1] + yv * bacc[2]
            else:
                man = Vec(1.0, 0.0, 0.0) * bacc[0] + Vec(0.0, 1.0, 0.0) * bacc[1] + Vec(0.0, 0.0, 1.0) * bacc[2]
        a = acc + man
        return [vel[0], vel[1], vel[2], a[0], a[1], a[2]]

    def step(self):
        if self.t >= self.segs[self.cur][1]:
            self.cur += 1
            self.bound = self.segs[self.cur][1]
            self.reset()
        prev = (self.t, list(self.y), self.rk.i, list(self.rk.k[-1]))
        while True:
            if self.n > self.n_max:
                return 2
            if self.t + self.next_h > self.bound:
                self.next_h = self.bound - self.t
            h = self.next_h
            if self.t >= self.bound:
                return 3
            if self.t + h == self.t:
                return 1
            r = self.rk.advance(h, self.t, self.y, self.rhs)
            if r is None:
                return 4
            self.t, self.y = r
            self.n += 1
            e = self.rk.error(h)
            err = max(max(abs(e[0] / self.tol_pos), max(abs(e[1] / self.tol_pos), abs(e[2] / self.tol_pos))),
                      max(abs(e[3] / self.tol_vel), max(abs(e[4] / self.tol_vel), abs(e[5] / self.tol_vel))))
            m = self.fac * math.pow(err, -(1.0 / float(self.rk.lower))) if err != 0.0 else math.inf
            c = self.fac_min if m < self.fac_min else (self.fac_max if m > self.fac_max else m)
            nh = self.next_h * c
            self.next_h = self.h_max if nh > self.h_max else nh
            if err <= 1.0:
                break
            self.t, self.y, self.rk.i = prev[0], list(prev[1]), prev[2]
            if self.rk.fsal:
                self.rk.k[-1] = list(prev[3])
        self.knots.append((self.t, tuple(self.y)))
        if self.soi is not None:
            self._events()
        return 0


# ---- Timeline::{new, common_times, divergence_time_before}  ephemeris/src/propagators/spacecraft.rs:129-213 ----
EPOCH_MIN, EPOCH_MAX = -1.7976931348623157e308, 1.7976931348623157e308


def timeline_new(burns):
    """burns: (start, end, acc[3], ref) -> segments (start, end, thrust) with thrust = None for a coast."""
    segs, cursor = [], EPOCH_MIN
    for s, e, acc, ref in sorted(burns, key=lambda b: b[0]):        # sort_by is stable, like sorted()
        if s > cursor:
            segs.append((cursor, s, None))
        cursor = e
        segs.append((s, e, (tuple(float(x) for x in acc), int(ref))))
    if cursor < EPOCH_MAX:
        segs.append((cursor, EPOCH_MAX, None))
    return segs


def timeline_common_times(a, b):
    out, done = [], False
    for s1, s2 in zip(a, b):
        if done or s1[0] != s2[0]:
            break
        out.append(s1[0])
        if s1[2] != s2[2]:
            done = True
    return out


def timeline_divergence_time_before(new_burns, old_burns, before):
    """self = the new timeline (flight_plan.rs:289-291). None where the reference would panic (unwrap on empty)."""
    times = []
    for t in timeline_common_times(timeline_new(new_burns), timeline_new(old_burns)):
        if not t < before:
            break
        times.append(t)
    return times[-1] if times else None


# ---- UniformSpline container operations  ephemeris/src/trajectory.rs:484-617 (host logic) ----
def _usize(x):
    """Rust `as usize` from f64: saturating, NaN -> 0"""
    if not x > 0.0:
        return 0
    return (1 << 64) - 1 if x >= 18446744073709551616.0 else int(x)


class Spline:
    """UniformSpline<V> with opaque polynomials: start, interval, list"""

    def __init__(self, start, interval, polys):
        self.start, self.interval, self.polys = start, interval, list(polys)

    def span(self):
        return self.interval * float(len(self.polys))

    def _idx(self, time):
        if math.copysign(1.0, time) < 0.0 or time >= self.span():
            return None
        return _usize(time / self.interval)

    def _idx_excl(self, time):
        if math.copysign(1.0, time) < 0.0 or time > self.span():
            return None
        return max(_usize(math.ceil(time / self.interval)) - 1, 0)

    def clear_before(self, at):
        idx = self._idx_excl((at + self.interval) - self.start)
        if idx is not None:
            self.start += self.interval * float(idx)
            del self.polys[:idx]

    def clear_after(self, at):
        idx = self._idx(at - self.start)
        if idx is not None:
            del self.polys[idx:]

    def between(self, start, end):
        if not self.polys:
            return None
        a, b = self._idx_excl(start - self.start), self._idx_excl(end - self.start)
        if a is None or b is None:
            return None
        return Spline(self.start + self.interval * float(a), self.interval,
                      [self.polys[i] for i in range(a, b + 1) if i < len(self.polys)])


def hermite_join(lhs, rhs):
    """SpacecraftPropagator::join (ephemeris/src/propagators/spacecraft.rs:558-561): lhs.clear_after(rhs.start()) --
    retain the knots with `at > t` (trajectory.rs:842-845), rhs.start() = first knot or Epoch::MIN (:756-758) -- then
    lhs.extend(rhs) (:847-849). Splines are lists of (t, pos, vel)."""
    at = rhs[0][0] if rhs else EPOCH_MIN
    return [k for k in lhs if at > k[0]] + list(rhs)


def _rust_binary_search(times, t):
    """slice::binary_search_by on a sorted list: ("ok", i) when times[i] == t, else ("err", insertion index)."""
    import bisect
    i = bisect.bisect_left(times, t)
    return ("ok", i) if i < len(times) and times[i] == t else ("err", i)


def transitions_clear_after(tr, at):
    """SoiTransitions::clear_after (ephemeris_explorer/src/dynamics/spacecraft.rs:341-346). tr: list of (time, body)."""
    kind, i = _rust_binary_search([t for t, _ in tr], at)
    return tr[:i + 1] if kind == "ok" else tr[:i]


def transitions_insert(tr, time, body):
    """SoiTransitions::insert (:331-337): replace at an equal time; no entry when the previous one is the same body."""
    kind, i = _rust_binary_search([t for t, _ in tr], time)
    if kind == "ok":
        tr[i] = (time, body)
    elif i > 0 and tr[i - 1][1] == body:
        pass
    else:
        tr.insert(i, (time, body))


def transitions_join(lhs, rhs, at):
    """item.transitions.clear_after(at); item.transitions.extend(rhs)   (:838-839, extend = insert each, :356-361)"""
    out = transitions_clear_after(list(lhs), at)
    for time, body in rhs:
        transitions_insert(out, time, body)
    return out


def apsides_join(lhs, rhs, at):
    """item.apsides.clear_after(at); item.apsides.extend(rhs)   (:836-837; clear_after :431-436, extend appends :426-428).
    Entries are (time, distance, kind, body); times distinct at the cut."""
    kind, i = _rust_binary_search([a[0] for a in lhs], at)
    return list(lhs[:i + 1] if kind == "ok" else lhs[:i]) + list(rhs)


# ---- adaptive plot sampling (ephemeris_explorer/src/ui/world/plot.rs:93-149,318-374,429-436) ---------------------------
def _f32(x):
    import struct
    return struct.unpack("f", struct.pack("f", x))[0]            # DVec3::as_vec3: `as f32`, round to nearest


def angular_distance(cam, p1, p2):
    """plot.rs:429-436 -- tan^2 of the angle between p1 and p2 seen from the camera. glam DVec3::normalize =
    self * (1 / self.length()), length = sqrt(x*x + y*y + z*z) (glam 0.30.10, Cargo.lock:2890-2892)."""
    def normalize(v):
        r = 1.0 / math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
        return v * r
    v1, v2 = normalize(p1 - cam), normalize(p2 - cam)
    w = Vec(v1[1] * v2[2] - v2[1] * v1[2], v1[2] * v2[0] - v2[2] * v1[0], v1[0] * v2[1] - v2[0] * v1[1])
    d = v1[0] * v2[0] + v1[1] * v2[1] + v1[2] * v2[2]
    return (w[0] * w[0] + w[1] * w[1] + w[2] * w[2]) / (d * d)   # wedge.length_squared() / v1.dot(v2).powi(2)


def plot_points_new(evaluate, tmin, tmax, cam, tan2_angular_resolution, max_points, max_inner=100000):
    """PlotPoints::new (plot.rs:93-149). evaluate(t) -> (position Vec, velocity Vec) in global space, or None.
    Returns ("ok", [(t, (x, y, z) as f32)]) or ("err", t) where the reference returns Err(t). The reference's inner
    loop never ends if the error is NaN; `max_inner` turns that into ("stuck", t)."""
    if max_points == 0:
        return "ok", []
    target = tan2_angular_resolution * tan2_angular_resolution
    previous_time = tmin
    previous = evaluate(previous_time)
    if previous is None:
        return "err", previous_time
    delta = tmax - previous_time
    estimated = None
    points = [(previous_time, tuple(_f32(c) for c in previous[0]))]
    while previous_time < tmax and len(points) < max_points:
        inner = 0
        while True:
            if estimated is not None and estimated > 0.0:
                delta = delta * 0.9 * math.sqrt(math.sqrt(target / estimated))
            t = previous_time + delta
            if t > tmax:
                t = tmax
            delta = t - previous_time
            extrapolated = previous[0] + previous[1] * delta
            current = evaluate(t)
            if current is None:
                return "err", t
            error = angular_distance(cam, extrapolated, current[0]) / 16.0
            if error <= target:
                break
            estimated = error
            inner += 1
            if inner >= max_inner:
                return "stuck", t
        previous_time, previous, estimated = t, current, error
        points.append((t, tuple(_f32(c) for c in previous[0])))
    return "ok", points


def plot_window(traj_bounds, ref_bounds, plot_start, plot_end, bound, current):
    """compute_plot_points_parallel :329-352: (min, max) of the plotted span, or None when nothing is drawn.
    traj_bounds / ref_bounds = (start, end, segment_count) (ref_bounds None without a reference); bound 0 None, 1 Start,
    2 End. Ord::clamp / max / min on Epoch (total order on f64 seconds, ftime/src/duration.rs:162-180)."""
    clamp = lambda x, lo, hi: lo if x < lo else (hi if x > hi else x)      # noqa: E731
    start, end, segs = traj_bounds
    if ref_bounds is not None:
        start, end, segs = max(start, ref_bounds[0]), min(end, ref_bounds[1]), min(segs, ref_bounds[2])
    if segs == 0 or start > end:                    # is_empty(); clamp(min > max) would panic in the reference
        return None
    current_clamped = clamp(current, start, end)
    tmin, tmax = clamp(plot_start, start, end), clamp(plot_end, start, end)
    if bound == 1:
        tmin = tmin if current_clamped < tmin else current_clamped          # Ord::max
    elif bound == 2:
        tmax = current_clamped if current_clamped < tmax else tmax          # Ord::min
    if tmin >= tmax:
        return None
    return tmin, tmax
