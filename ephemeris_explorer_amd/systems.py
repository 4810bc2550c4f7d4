"""Readers for the reference's on-disk inputs: `state.json`, `ephemeris.json`, `ships/*.json`.

Formats: ephemeris_explorer/src/load/solar_system/loaders.rs:210-270 (state), :286-340 (ephemeris),
load/solar_system/mod.rs:208-226 (ships); epoch strings: ftime/src/epoch.rs:19-44,154-217
(f64 seconds since 1958-01-01 TAI); durations: ftime/src/duration.rs:279-345.
"""
import json
import math
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

SEC_PER_DAY = 86400.0


def _days_from_civil(y, m, d):
    # proleptic Gregorian day number relative to 1970-01-01 (ftime/src/epoch.rs days_from_civil)
    y -= m <= 2
    era = (y if y >= 0 else y - 399) // 400
    yoe = y - era * 400
    doy = (153 * (m + (-3 if m > 2 else 9)) + 2) // 5 + d - 1
    doe = yoe * 365 + yoe // 4 - yoe // 100 + doy
    return era * 146097 + doe - 719468


def parse_epoch(s):
    """"YYYY-MM-DD HH:MM:SS[.frac]" (TAI) -> f64 seconds since 1958-01-01 (epoch.rs:19-44,158-217)."""
    date, time = s.split(" ", 1)
    y, mo, d = (int(x) for x in date.split("-", 2))
    if "." in time:
        hms, frac = time.split(".", 1)
        if not frac or not frac.isdigit():
            raise ValueError(s)
        digits = frac[:3]
        millis = int(digits) * 10 ** (3 - len(digits))
    else:
        hms, millis = time, 0
    h, mi, sec = (int(x) for x in hms.split(":", 2))
    if not (1 <= mo <= 12) or h > 23 or mi > 59 or sec > 59 or millis > 999:
        raise ValueError(s)
    days = _days_from_civil(y, mo, d) - _days_from_civil(1958, 1, 1)
    secs = days * 86400 + h * 3600 + mi * 60 + sec
    return float(secs) + float(millis) / 1000.0


def _civil_from_days(z):
    z += 719468
    era = (z if z >= 0 else z - 146096) // 146097
    doe = z - era * 146097
    yoe = (doe - doe // 1460 + doe // 36524 - doe // 146096) // 365
    y = yoe + era * 400
    doy = doe - (365 * yoe + yoe // 4 - yoe // 100)
    mp = (5 * doy + 2) // 153
    d = doy - (153 * mp + 2) // 5 + 1
    m = mp + (3 if mp < 10 else -9)
    return y + (m <= 2), m, d


def format_epoch(seconds):
    """f64 seconds since 1958-01-01 TAI -> "YYYY-MM-DD HH:MM:SS.mmm" (the form state.json / ships use)."""
    ms_total = int(round(seconds * 1000.0))
    days, ms = divmod(ms_total, 86400000)
    y, m, d = _civil_from_days(days + _days_from_civil(1958, 1, 1))
    h, ms = divmod(ms, 3600000)
    mi, ms = divmod(ms, 60000)
    sec, ms = divmod(ms, 1000)
    return f"{y:04d}-{m:02d}-{d:02d} {h:02d}:{mi:02d}:{sec:02d}.{ms:03d}"


_UNITS_MS = {}
for _names, _ms in (
    (("y", "yr", "yrs", "year", "years"), int(365.25 * 86400.0 * 1000.0)),
    (("d", "day", "days"), 86400000),
    (("h", "hr", "hrs", "hour", "hours"), 3600000),
    (("m", "min", "mins", "minute", "minutes"), 60000),
    (("s", "sec", "secs", "second", "seconds"), 1000),
    (("ms", "msec", "msecs", "millisecond", "milliseconds"), 1),
):
    for _n in _names:
        _UNITS_MS[_n] = _ms


def parse_duration(s):
    """Whitespace-separated <unsigned integer> <unit> pairs with an optional leading sign; summed in integer
    milliseconds and converted once: `(total_ms as f64) * 1e-3` (ftime/src/duration.rs:279-345)."""
    s = s.strip()
    if not s:
        raise ValueError("empty duration")
    sign = 1.0
    if s[0] == "+":
        s = s[1:].lstrip()
    elif s[0] == "-":
        sign, s = -1.0, s[1:].lstrip()
    toks = s.split()
    total_ms = 0
    # zip(tokens, tokens.skip(1)).step_by(2): a trailing unpaired token is silently ignored, as in the reference
    for num, unit in list(zip(toks, toks[1:]))[::2]:
        if not num.isdigit():
            raise ValueError(f"invalid number {num!r}")
        u = unit.strip().lower()
        if u not in _UNITS_MS:
            raise ValueError(f"unknown unit {unit!r}")
        total_ms += int(num) * _UNITS_MS[u]
    return sign * (float(total_ms) * 1e-3)


@dataclass
class System:
    name: str
    epoch: float                       # seconds since 1958-01-01 TAI
    names: list
    mu: np.ndarray                     # [N] km^3/s^2
    pos: np.ndarray                    # [N,3] km
    vel: np.ndarray                    # [N,3] km/s
    dt: float = 0.0                    # ephemeris.json "dt" in seconds
    count: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    degree: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    path: Path = None

    @property
    def n(self):
        return len(self.mu)


def load_system(directory):
    directory = Path(directory)
    st = json.loads((directory / "state.json").read_text())
    bodies = st["bodies"]
    sysm = System(
        name=st.get("name", directory.name),
        epoch=parse_epoch(st["epoch"]),
        names=[b["name"] for b in bodies],
        mu=np.array([b["mu"] for b in bodies], dtype=np.float64),
        pos=np.array([b["position"] for b in bodies], dtype=np.float64),
        vel=np.array([b["velocity"] for b in bodies], dtype=np.float64),
        path=directory,
    )
    ep = directory / "ephemeris.json"
    if ep.exists():
        e = json.loads(ep.read_text())
        sysm.dt = parse_duration(e["dt"])
        # a body missing from `settings` is an error in the app (load/mod.rs:313-319)
        sysm.count = np.array([e["settings"][n]["count"] for n in sysm.names], dtype=np.uint32)
        sysm.degree = np.array([e["settings"][n]["degree"] for n in sysm.names], dtype=np.uint32)
    return sysm


def soi_radii(system):
    """Sphere-of-influence radius per body, in body order (ephemeris_explorer/src/load/mod.rs:283-307,
    dynamics/spacecraft.rs:33-38): bodies by decreasing mu (stable sort); a body's parent is, among the heavier bodies
    whose sphere contains it at the epoch, the one giving the smallest r = a * (mu / mu_parent)^(2/5) with a = their
    distance at the epoch; no such body -> infinity (the root). The power is the host libm's pow, as in the
    reference (f64::powf)."""
    order = sorted(range(system.n), key=lambda i: -system.mu[i])          # sorted() is stable, like sort_by
    radius = {}
    for k, i in enumerate(order):
        best = math.inf
        for j in order[:k]:
            d = system.pos[i] - system.pos[j]
            a = math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
            if a < radius[j]:
                r = a * math.pow(system.mu[i] / system.mu[j], 2.0 / 5.0)
                if r < best:                                              # min_by keeps the first minimum
                    best = r
        radius[i] = best
    return np.array([radius[i] for i in range(system.n)], dtype=np.float64)


@dataclass
class Burn:
    start: float
    duration: float
    acceleration: np.ndarray           # km/s^2 in the burn frame axes
    reference: str = None              # body name (TNB frame relative to it) or None = inertial


@dataclass
class Ship:
    name: str
    integrator: str
    tolerance: float
    start: float
    end: float
    pos: np.ndarray
    vel: np.ndarray
    burns: list


def load_ship(path):
    d = json.loads(Path(path).read_text())
    return Ship(
        name=d["name"], integrator=d["integrator"], tolerance=float(d["tolerance"]),
        start=parse_epoch(d["start"]), end=parse_epoch(d["end"]),
        pos=np.array(d["position"], dtype=np.float64), vel=np.array(d["velocity"], dtype=np.float64),
        burns=[Burn(parse_epoch(b["start"]), parse_duration(b["duration"]),
                    np.array(b["acceleration"], dtype=np.float64), b.get("reference")) for b in d.get("burns", [])],
    )
