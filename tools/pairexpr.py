"""Evaluation orders of the point-mass term as expression trees over IEEE binary64, evaluated in pure Python.

The reference's innermost arithmetic -- `particular::gravity::newtonian` `acceleration_paired` / `acceleration_at::<false>`
(call sites ephemeris/src/propagators/nbody.rs:29, ephemeris_explorer/src/dynamics/spacecraft.rs:73; crate `particular`
0.8.0-dev @ d490707a, Cargo.lock:4277-4285) -- is not on disk. This module spells out every order a Rust implementation of
  a = d * mu / |d|^3,   d = p_other - p_self
could plausibly use, as (n2 form, denominator form, application form) triples. Python floats ARE binary64 and `+ - * /
math.sqrt` are correctly rounded, so each tree evaluates to exactly the bits the Rust expression would give; `fma` is done in
exact rational arithmetic. `libm pow` forms are platform dependent and marked so.

The library's seven built orders (csrc/pair_term.h, EPH_PAIR_VARIANT) are BUILT[k].  Used by tools/pair_probe.py (generates
the probe set) and tools/identify_pair_variant.py (names the order a print-out from the real crate follows)."""
import math
import struct
from fractions import Fraction


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def from_bits(u):
    return struct.unpack("<d", struct.pack("<Q", u))[0]


def hexbits(x):
    return f"{bits(x):016x}"


def fma(a, b, c):
    """correctly rounded a*b + c (Fraction -> float conversion rounds to nearest even)"""
    if not (math.isfinite(a) and math.isfinite(b) and math.isfinite(c)):
        return a * b + c
    r = Fraction(a) * Fraction(b) + Fraction(c)
    if r == 0:
        # sign of an exact zero sum: +0 unless both addends are -0
        prod_neg = (math.copysign(1.0, a) * math.copysign(1.0, b)) < 0
        return -0.0 if (prod_neg and math.copysign(1.0, c) < 0) else 0.0
    return float(r)


def div(a, b):
    """IEEE division (Python raises on a zero divisor)"""
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0.0:
            return math.nan
        return math.copysign(math.inf, a) * math.copysign(1.0, b)


def sqrt(x):
    return math.sqrt(x) if x >= 0.0 else math.nan


# ---- |d|^2 ------------------------------------------------------------------------------------------------------------------
N2_FORMS = {
    "dot_lr": ("(d.x*d.x + d.y*d.y) + d.z*d.z   [glam DVec3::length_squared / dot, left to right]",
               lambda x, y, z: (x * x + y * y) + z * z),
    "dot_rl": ("d.x*d.x + (d.y*d.y + d.z*d.z)",
               lambda x, y, z: x * x + (y * y + z * z)),
    "dot_fma": ("d.z.mul_add(d.z, d.y.mul_add(d.y, d.x*d.x))   [dot with fused multiply-adds]",
                lambda x, y, z: fma(z, z, fma(y, y, x * x))),
    "dot_fma_rev": ("d.x.mul_add(d.x, d.y.mul_add(d.y, d.z*d.z))",
                    lambda x, y, z: fma(x, x, fma(y, y, z * z))),
}

# ---- the scalar every component is scaled by, from n2 and mu ----------------------------------------------------------------
# kind "scale": returns s with a = d * s applied component-wise by APPLY; kind "den": returns p for the division forms


def _pow15(n2):
    return math.pow(n2, 1.5)          # the platform's libm


def _powm15(n2):
    try:
        return math.pow(n2, -1.5)
    except (ZeroDivisionError, ValueError):
        return math.inf if n2 == 0.0 else math.nan


INV_FORMS = {      # 1 / r^3 as ONE scalar `inv`
    "recip(n2*sqrt)": ("1.0 / (n2 * n2.sqrt())", lambda n2: div(1.0, n2 * sqrt(n2)), False),
    "recip(r*r*r)": ("let r = n2.sqrt(); 1.0 / (r * r * r)   [also r.powi(3)]", lambda n2: div(1.0, (sqrt(n2) * sqrt(n2)) * sqrt(n2)), False),
    "s*s*s,s=1/sqrt": ("let s = 1.0 / n2.sqrt(); s * s * s   [n2.sqrt().recip().powi(3)]", lambda n2: (div(1.0, sqrt(n2)) * div(1.0, sqrt(n2))) * div(1.0, sqrt(n2)), False),
    "recip(n2)*recip(sqrt)": ("(1.0 / n2) * (1.0 / n2.sqrt())", lambda n2: div(1.0, n2) * div(1.0, sqrt(n2)), False),
    "recip(n2)/sqrt": ("(1.0 / n2) / n2.sqrt()", lambda n2: div(div(1.0, n2), sqrt(n2)), False),
    "s*s*s,s=sqrt(1/n2)": ("let s = (1.0 / n2).sqrt(); s * s * s", lambda n2: (sqrt(div(1.0, n2)) * sqrt(div(1.0, n2))) * sqrt(div(1.0, n2)), False),
    "s/n2,s=1/sqrt": ("(1.0 / n2.sqrt()) / n2", lambda n2: div(div(1.0, sqrt(n2)), n2), False),
    "recip(powf1.5)": ("1.0 / n2.powf(1.5)   [libm pow: platform dependent]", lambda n2: div(1.0, _pow15(n2)), True),
    "powf(-1.5)": ("n2.powf(-1.5)   [libm pow: platform dependent]", lambda n2: _powm15(n2), True),
}
DEN_FORMS = {      # r^3 as a denominator p
    "n2*sqrt": ("n2 * n2.sqrt()", lambda n2: n2 * sqrt(n2), False),
    "r*r*r": ("let r = n2.sqrt(); r * r * r", lambda n2: (sqrt(n2) * sqrt(n2)) * sqrt(n2), False),
    "powf1.5": ("n2.powf(1.5)   [libm pow: platform dependent]", _pow15, True),
}
APPLY_INV = {      # (d component, mu, inv) -> a component
    "d*(mu*inv)": lambda c, mu, inv: c * (mu * inv),
    "(d*mu)*inv": lambda c, mu, inv: (c * mu) * inv,
    "(d*inv)*mu": lambda c, mu, inv: (c * inv) * mu,
}
APPLY_DEN = {      # (d component, mu, p) -> a component
    "(d*mu)/p": lambda c, mu, p: div(c * mu, p),
    "d*(mu/p)": lambda c, mu, p: c * div(mu, p),
    "(d/p)*mu": lambda c, mu, p: div(c, p) * mu,
    "d/(p/mu)": lambda c, mu, p: div(c, div(p, mu)),
}


class Tree:
    def __init__(self, n2_form, scalar_kind, scalar_form, apply_form):
        self.n2_form, self.kind, self.scalar_form, self.apply_form = n2_form, scalar_kind, scalar_form, apply_form
        table = INV_FORMS if scalar_kind == "inv" else DEN_FORMS
        self.text, self.scalar, self.platform_dependent = table[scalar_form]
        self.apply = (APPLY_INV if scalar_kind == "inv" else APPLY_DEN)[apply_form]
        self.n2 = N2_FORMS[n2_form][1]

    def name(self):
        return f"n2 = {self.n2_form}; {'inv' if self.kind == 'inv' else 'p'} = {self.scalar_form}; a = {self.apply_form}"

    def describe(self):
        return (f"n2  = {N2_FORMS[self.n2_form][0]}\n"
                f"{'inv' if self.kind == 'inv' else 'p  '} = {self.text}\n"
                f"a   = {self.apply_form}   (component-wise; the other body's half uses -d and the other mu)")

    def directed(self, p_self, p_other, mu_other):
        """acceleration of the body at p_self caused by (p_other, mu_other)"""
        d = [p_other[c] - p_self[c] for c in range(3)]
        n2 = self.n2(*d)
        s = self.scalar(n2)
        return [self.apply(d[c], mu_other, s) for c in range(3)]

    def paired(self, pi, mui, pj, muj):
        """acceleration_paired: (a on i, a on j) with ONE direction vector d = pj - pi and its negation"""
        d = [pj[c] - pi[c] for c in range(3)]
        n2 = self.n2(*d)
        s = self.scalar(n2)
        ai = [self.apply(d[c], muj, s) for c in range(3)]
        aj = [self.apply(-d[c], mui, s) for c in range(3)]
        return ai, aj


def all_trees():
    out = []
    for n2f in N2_FORMS:
        for sf in INV_FORMS:
            for af in APPLY_INV:
                out.append(Tree(n2f, "inv", sf, af))
        for sf in DEN_FORMS:
            for af in APPLY_DEN:
                out.append(Tree(n2f, "den", sf, af))
    return out


# the seven orders the library builds (EPH_PAIR_VARIANT / eph_set_pair_variant)
BUILT = {
    0: Tree("dot_lr", "inv", "recip(n2*sqrt)", "d*(mu*inv)"),
    1: Tree("dot_lr", "inv", "recip(r*r*r)", "d*(mu*inv)"),
    2: Tree("dot_lr", "inv", "s*s*s,s=1/sqrt", "d*(mu*inv)"),
    3: Tree("dot_lr", "inv", "recip(n2)*recip(sqrt)", "d*(mu*inv)"),
    4: Tree("dot_lr", "den", "n2*sqrt", "(d*mu)/p"),
    5: Tree("dot_lr", "den", "n2*sqrt", "d*(mu/p)"),
    6: Tree("dot_lr", "den", "n2*sqrt", "(d/p)*mu"),
}


def ulp_distance(a, b):
    """distance in units in the last place between two finite doubles (by their ordered integer images)"""
    def key(x):
        u = bits(x)
        return u if u < 1 << 63 else (1 << 63) - u
    return abs(key(a) - key(b))
