#!/usr/bin/env python3
"""Builds the identification kit for the one formula on the path whose reference source is not on disk.

`particular::gravity::newtonian` (`acceleration_paired`, ephemeris/src/propagators/nbody.rs:29; `acceleration_at::<false>`,
ephemeris_explorer/src/dynamics/spacecraft.rs:73; crate `particular` 0.8.0-dev @ d490707a, Cargo.lock:4277-4285) and the
platform `powf` of the step-size controller (integration/src/runge_kutta/mod.rs:238-239) are the two places where "bit-identical
to this repository's CPU restatement" is not yet "bit-identical to the Rust binary". Only someone who can build the reference
can close that; this tool hands them a one-minute way to do it:

  python tools/pair_probe.py            writes tests/golden/pair_probe.json  (operands + expected bits per built order k)
                                        and    tools/particular_probe.rs     (a Rust #[test] with the operands baked in)
  cargo test -p ephemeris --test particular_probe -- --nocapture > printout.txt      (on a machine with the reference)
  python tools/identify_pair_variant.py printout.txt                                 -> "k = 5", or the matching tree

Probe operands: pairs (p_i, mu_i, p_j, mu_j) on which the seven built orders give PAIRWISE different bits in at least one of
the six result components (searched at random over realistic solar-system scales, deterministic seed), followed by edge
operands (zero separation components, equal masses, a massless partner, extreme separations) that need not separate anything.
Pure Python (tools/pairexpr.py): no oracle, no library -- tests/test_pair_probe.py checks the file against both."""
import json
import math
import platform
import random
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(Path(__file__).resolve().parent))
import pairexpr as pe  # noqa: E402

N_SEPARATING = 64
POW_ORDERS = (4, 5, 7, 8)      # min(order, embedded order) of the selectable pairs (flight_plan.rs:175-184): 4, 5, 7, 8
N_POW = 64


def separates_all(outs):
    ks = sorted(outs)
    for a in range(len(ks)):
        for b in range(a + 1, len(ks)):
            if outs[ks[a]] == outs[ks[b]]:
                return False
    return True


def evaluate(op):
    pi, mui, pj, muj = op
    outs = {}
    for k, t in pe.BUILT.items():
        ai, aj = t.paired(pi, mui, pj, muj)
        outs[k] = tuple(pe.bits(v) for v in ai + aj)
    return outs


def random_operand(rng):
    # barycentric positions in km, mu in km^3/s^2: the reference's units (systems/*/state.json)
    scale = 10.0 ** rng.uniform(3.0, 9.5)
    pi = [rng.uniform(-1.0, 1.0) * scale for _ in range(3)]
    sep = 10.0 ** rng.uniform(2.0, 9.5)
    u = [rng.gauss(0.0, 1.0) for _ in range(3)]
    nu = math.sqrt(sum(c * c for c in u))
    pj = [pi[c] + sep * u[c] / nu for c in range(3)]
    mui = 10.0 ** rng.uniform(-3.0, 11.2)
    muj = 10.0 ** rng.uniform(-3.0, 11.2)
    return pi, mui, pj, muj


def edge_operands():
    out = []
    out.append(([0.0, 0.0, 0.0], 1.32712440041e11, [1.495978707e8, 0.0, 0.0], 398600.435436))          # two zero components
    out.append(([1.0e8, -2.0e7, 3.0e6], 4902.800066, [1.0e8, -2.0e7, 3.0e6 + 384400.0], 398600.435436))  # along one axis
    out.append(([7.0, 11.0, 13.0], 1.0, [8.0, 12.0, 14.0], 1.0))                                          # equal masses, small numbers
    out.append(([1.0e9, 1.0e9, 1.0e9], 0.0, [1.0e9 + 6778.0, 1.0e9, 1.0e9 - 1.0], 398600.435436))      # a massless partner
    out.append(([1.0e-3, 2.0e-3, -1.0e-3], 1.0e-9, [1.5e-3, 2.5e-3, -0.5e-3], 2.0e-9))                    # metre-scale separation
    out.append(([-4.5e9, 1.0e9, 2.0e8], 6836527.100580, [4.4e9, -2.0e9, -1.0e8], 5793939.0))             # across the system
    out.append(([0.1, 0.2, 0.3], 1.0 / 4096.0, [0.4, 0.6, 0.9], 1.0 / 4096.0))                            # N-body units (Plummer workload)
    out.append(([3.0, 4.0, 0.0], 2.0, [0.0, 0.0, 0.0], 8.0))                                              # n2 = 25 exactly: r = 5
    return out


def pow_operands(rng, k):
    """64 values of err / tol per controller order: 8 fixed, 40 random, and 16 on which THIS host's libm pow is not the
    correctly rounded value (found by search: about one call in a thousand) -- the operands that tell libms apart."""
    y = -(1.0 / float(k))
    errs = [1.0, 0.5, 2.0, 1.0e-3, 1.0e3, 0.999999999, 1.000000001, 1.0e-12]
    while len(errs) < 48:
        errs.append(10.0 ** rng.uniform(-9.0, 4.0))          # err / tol of accepted and rejected attempts
    tried = 0
    while len(errs) < N_POW and tried < 400000:
        e = 10.0 ** rng.uniform(-9.0, 4.0)
        tried += 1
        if math.pow(e, y) != cr_pow(e, y):
            errs.append(e)
    while len(errs) < N_POW:
        errs.append(10.0 ** rng.uniform(-9.0, 4.0))
    return errs


def main():
    rng = random.Random(20260927)
    ops, tried = [], 0
    while len(ops) < N_SEPARATING:
        op = random_operand(rng)
        tried += 1
        if separates_all(evaluate(op)):
            ops.append(op)
    n_sep = len(ops)
    ops += edge_operands()
    expected = {str(k): [] for k in pe.BUILT}
    at_expected = {str(k): [] for k in pe.BUILT}
    for op in ops:
        outs = evaluate(op)
        for k in pe.BUILT:
            expected[str(k)].append([f"{u:016x}" for u in outs[k]])
            # acceleration_at::<false>: the acceleration AT p_i caused by (p_j, mu_j) -- the first half of the pair
            at_expected[str(k)].append([f"{u:016x}" for u in outs[k][:3]])
    # the library's controller evaluates err^(-1/k) correctly rounded (cr_pow); the reference calls the
    # platform's powf. Correctly rounded values here from 80-digit decimal arithmetic.
    pow_cr, pow_host, errs = {}, {}, {}
    for k in POW_ORDERS:
        y = -(1.0 / float(k))                               # `-k.inv()` with k = U::from(order)
        errs[k] = pow_operands(rng, k)
        pow_cr[str(k)] = [f"{pe.bits(cr_pow(e, y)):016x}" for e in errs[k]]
        pow_host[str(k)] = [f"{pe.bits(math.pow(e, y)):016x}" for e in errs[k]]
    doc = {
        "what": "operands and expected result bits for the seven built evaluation orders of the point-mass term, and for powf "
                "of the step-size controller; generated by tools/pair_probe.py (seed 20260927), checked by tests/test_pair_probe.py",
        "layout": {"pairs": "pi[3], mui, pj[3], muj as IEEE-754 binary64 bit patterns (hex)",
                   "expected[k][i]": "acceleration_paired(&(pi, mui), &(pj, muj), &0.0): a_i.x a_i.y a_i.z a_j.x a_j.y a_j.z",
                   "at_expected[k][i]": "(pj, muj).acceleration_at::<false>(&pi, &0.0): x y z",
                   "pow": "err.powf(-(1.0 / k as f64)) for k in orders"},
        "n_separating": n_sep, "random_operands_tried": tried,
        "pairs": [{"pi": [pe.hexbits(c) for c in op[0]], "mui": pe.hexbits(op[1]),
                   "pj": [pe.hexbits(c) for c in op[2]], "muj": pe.hexbits(op[3])} for op in ops],
        "built_orders": {str(k): t.name() for k, t in pe.BUILT.items()},
        "expected": expected, "at_expected": at_expected,
        "pow": {"orders": list(POW_ORDERS), "err": {str(k): [pe.hexbits(e) for e in errs[k]] for k in POW_ORDERS}, "correctly_rounded": pow_cr,
                "generating_host_libm": pow_host, "generating_host": f"{platform.libc_ver()[0]} {platform.libc_ver()[1]} {platform.machine()}"},
    }
    out = ROOT / "tests" / "golden" / "pair_probe.json"
    text = json.dumps(doc, indent=1)
    import re
    text = re.sub(r'\[\s+((?:"[0-9a-f]{16}",?\s*)+)\]', lambda m: "[" + " ".join(m.group(1).split()) + "]", text)   # one row per line
    out.write_text(text + "\n")
    (Path(__file__).resolve().parent / "particular_probe.rs").write_text(rust_test(ops, errs))
    differing = sum(1 for k in POW_ORDERS for a, b in zip(pow_cr[str(k)], pow_host[str(k)]) if a != b)
    print(f"{out}: {n_sep} separating operands (of {tried} tried) + {len(ops) - n_sep} edge operands; "
          f"powf: {differing} of {N_POW * len(POW_ORDERS)} differ between correctly rounded and this host's libm")


def cr_pow(x, y):
    """x^y correctly rounded for x > 0 and y = -1/k exactly representable?  No: y is the DOUBLE nearest -1/k, as in the reference,
    so x^y is evaluated for that double, to 200 bits with decimal-free integer arithmetic, then rounded once."""
    from fractions import Fraction
    import decimal
    decimal.getcontext().prec = 80
    d = (decimal.Decimal(Fraction(x).numerator) / decimal.Decimal(Fraction(x).denominator)).ln() * \
        (decimal.Decimal(Fraction(y).numerator) / decimal.Decimal(Fraction(y).denominator))
    v = d.exp()
    # 80 significant digits: the rounding to binary64 is safe unless v lies within 1e-60 of a rounding boundary
    f = Fraction(v)
    return float(f)


def rust_test(ops, errs):
    def arr(vals):
        return ", ".join(f"0x{pe.bits(v):016x}" for v in vals)
    rows = ",\n    ".join(f"[{arr(op[0] + [op[1]] + op[2] + [op[3]])}]" for op in ops)
    erows = ",\n    ".join(f"[{arr(errs[k])}]" for k in POW_ORDERS)
    return f"""// Generated by tools/pair_probe.py -- prints the bits the real `particular` crate and the platform's `powf` produce on the
// probe operands of tests/golden/pair_probe.json. Drop into ephemeris/tests/particular_probe.rs of the reference and run
//   cargo test -p ephemeris --test particular_probe -- --nocapture > printout.txt
// then  python tools/identify_pair_variant.py printout.txt  in this repository.
use glam::DVec3;
use particular::gravity::newtonian::{{AccelerationAt, AccelerationPaired}};

const OPS: [[u64; 8]; {len(ops)}] = [
    {rows}
];
const ORDERS: [u16; {len(POW_ORDERS)}] = [{", ".join(str(k) for k in POW_ORDERS)}];
const ERR: [[u64; {N_POW}]; {len(POW_ORDERS)}] = [
    {erows}
];

#[test]
fn particular_probe() {{
    let f = f64::from_bits;
    for (i, o) in OPS.iter().enumerate() {{
        let p_i = (DVec3::new(f(o[0]), f(o[1]), f(o[2])), f(o[3]));
        let p_j = (DVec3::new(f(o[4]), f(o[5]), f(o[6])), f(o[7]));
        let (a, b): (DVec3, DVec3) = p_i.acceleration_paired(&p_j, &0.0);
        println!("pair {{}} {{:016x}} {{:016x}} {{:016x}} {{:016x}} {{:016x}} {{:016x}}", i,
            a.x.to_bits(), a.y.to_bits(), a.z.to_bits(), b.x.to_bits(), b.y.to_bits(), b.z.to_bits());
        let c: DVec3 = p_j.acceleration_at::<false>(&p_i.0, &0.0);
        println!("at {{}} {{:016x}} {{:016x}} {{:016x}}", i, c.x.to_bits(), c.y.to_bits(), c.z.to_bits());
    }}
    for (k, errs) in ORDERS.iter().zip(ERR.iter()) {{
        for (i, e) in errs.iter().enumerate() {{
            // IController::step, integration/src/runge_kutta/mod.rs:238-239: `err.pow(-k.inv())` with k = U::from(order)
            println!("pow {{}} {{}} {{:016x}}", k, i, f(*e).powf(-(f64::from(*k).recip())).to_bits());
        }}
    }}
}}
"""


if __name__ == "__main__":
    main()
