#!/usr/bin/env python3
"""Names the evaluation order of the point-mass term a print-out from the REAL `particular` crate follows.

  python tools/identify_pair_variant.py printout.txt        (printout: tools/particular_probe.rs run inside the reference)
  python tools/identify_pair_variant.py --emulate 4         (self-check: fabricates the print-out order k would give)

Answers, in this order:
  1. "k = <n>": one of the seven orders the library builds (eph_set_pair_variant(n) / EPH_PAIR_VARIANT=n at run time) reproduces
     every printed bit -- set it, and the library is bit-identical to the Rust binary at this boundary;
  2. otherwise a search over the grammar of tools/pairexpr.py (how |d|^2 is summed, r^3 as n2*sqrt / r*r*r / powf, reciprocal
     or division, where mu multiplies, fused multiply-adds): every tree that reproduces all printed bits is written out as the
     Rust expression to add as an eighth order (csrc/pair_term.h pair_den / pair_apply, oracle/eph_oracle.c point_mass_term);
  3. otherwise the closest trees, with how many of the printed values they reproduce and the largest distance in ulp.
Separately: whether `acceleration_at::<false>` follows the same order as `acceleration_paired`, and how the platform's powf
relates to the correctly rounded value the library's step-size controller uses (integration/src/runge_kutta/mod.rs:238-239).
Pure Python; reads tests/golden/pair_probe.json.

The walk-through (on a machine that builds the reference; the choice moves positions by 1.2-2.4e-8 AU over 1e5 steps):
  a. tools/pair_probe.py wrote tests/golden/pair_probe.json (64 operand pairs on which the seven orders give pairwise different
     bits + 8 edge operands, with the expected bits per k; 256 operands of the controller's powf) and tools/particular_probe.rs, a
     Rust test with the operands baked in that prints `pair i <6 words>` (acceleration_paired, nbody.rs:29), `at i <3 words>`
     (acceleration_at::<false>, dynamics/spacecraft.rs:73) and `pow k i <word>` (err.powf(-1/k), runge_kutta/mod.rs:238-239);
  b. cp tools/particular_probe.rs <reference>/ephemeris/tests/ ; cargo test -p ephemeris --test particular_probe -- --nocapture > printout.txt
  c. python tools/identify_pair_variant.py printout.txt"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(Path(__file__).resolve().parent))
import pairexpr as pe  # noqa: E402


def load_probe():
    d = json.loads((ROOT / "tests" / "golden" / "pair_probe.json").read_text())
    ops = [([pe.from_bits(int(h, 16)) for h in p["pi"]], pe.from_bits(int(p["mui"], 16)),
            [pe.from_bits(int(h, 16)) for h in p["pj"]], pe.from_bits(int(p["muj"], 16))) for p in d["pairs"]]
    return d, ops


def parse(text):
    pair, at, pw = {}, {}, {}
    for line in text.splitlines():
        f = line.split()
        if len(f) == 8 and f[0] == "pair":
            pair[int(f[1])] = [int(h, 16) for h in f[2:]]
        elif len(f) == 5 and f[0] == "at":
            at[int(f[1])] = [int(h, 16) for h in f[2:]]
        elif len(f) == 4 and f[0] == "pow":
            pw[(int(f[1]), int(f[2]))] = int(f[3], 16)
    return pair, at, pw


def emulate(doc, k):
    lines = []
    for i, row in enumerate(doc["expected"][str(k)]):
        lines.append(f"pair {i} " + " ".join(row))
        lines.append(f"at {i} " + " ".join(doc["at_expected"][str(k)][i]))
    for o in doc["pow"]["orders"]:
        for i, h in enumerate(doc["pow"]["generating_host_libm"][str(o)]):
            lines.append(f"pow {o} {i} {h}")
    return "\n".join(lines)


def score(tree, ops, pair):
    """(values reproduced, values printed, largest ulp distance)"""
    hit = tot = 0
    worst = 0
    for i, want in pair.items():
        ai, aj = tree.paired(*ops[i])
        for got, w in zip(ai + aj, want):
            tot += 1
            gb = pe.bits(got)
            # the library accumulates into +0, so it cannot tell -0 from +0; the print-out can: compare bits as printed
            if gb == w:
                hit += 1
            else:
                wf = pe.from_bits(w)
                if wf == wf and got == got:
                    worst = max(worst, pe.ulp_distance(got, wf))
                else:
                    worst = max(worst, 1 << 62)
    return hit, tot, worst


def main(argv):
    doc, ops = load_probe()
    if len(argv) == 3 and argv[1] == "--emulate":
        text = emulate(doc, int(argv[2]))
    elif len(argv) == 2 and argv[1] not in ("-h", "--help"):
        text = Path(argv[1]).read_text()
    else:
        print(__doc__)
        return 2
    pair, at, pw = parse(text)
    if not pair:
        print("no `pair <i> <6 hex words>` lines found in the print-out")
        return 2
    print(f"print-out: {len(pair)} pair lines, {len(at)} at lines, {len(pw)} pow lines "
          f"({doc['n_separating']} of the {len(ops)} probe operands separate all seven built orders)")
    # 1. the built orders
    matches = []
    for k, t in pe.BUILT.items():
        hit, tot, worst = score(t, ops, pair)
        if hit == tot:
            matches.append(k)
    verdict = 1
    if matches:
        k = matches[0]
        print(f"\nk = {k}   -- `{pe.BUILT[k].name()}` reproduces all {6 * len(pair)} printed values")
        print(f"  eph_set_pair_variant({k}) before creating handles (or EPH_PAIR_VARIANT={k} in the environment): the library is then")
        print("  bit-identical to `particular` at nbody.rs:29; tests/test_gpu_variants.py covers that order on every kernel family.")
        verdict = 0
        chosen = pe.BUILT[k]
    else:
        print("\nnone of the seven built orders reproduces the print-out; searching the grammar of tools/pairexpr.py ...")
        scored = []
        for t in pe.all_trees():
            hit, tot, worst = score(t, ops, pair)
            scored.append((tot - hit, worst, t))
        scored.sort(key=lambda r: (r[0], r[1]))
        exact = [r for r in scored if r[0] == 0]
        chosen = None
        if exact:
            print(f"{len(exact)} expression tree(s) reproduce every printed value:")
            for _, _, t in exact:
                print("  ---" + (" (platform-dependent libm pow)" if t.platform_dependent else ""))
                for line in t.describe().splitlines():
                    print("  " + line)
            print("add it as an eighth order in csrc/pair_term.h (pair_den / pair_apply; kPairVariants in eph_internal.h, the table in "
                  "dispatch.cpp, N_PAIR_VARIANTS in build.py) and oracle/eph_oracle.c (point_mass_term).")
            chosen = exact[0][2]
        else:
            print("no tree of the grammar reproduces every value; the closest:")
            for miss, worst, t in scored[:5]:
                print(f"  {6 * len(pair) - miss} of {6 * len(pair)} values, worst distance {worst} ulp: {t.name()}")
            print("send the print-out back: the formula has a shape the grammar does not hold (softening folded in, another vector type, ...)")
    # acceleration_at vs acceleration_paired
    if at and chosen is not None:
        same = sum(1 for i, w in at.items() if [pe.bits(v) for v in chosen.directed(ops[i][0], ops[i][2], ops[i][3])] == w)
        print(f"\nacceleration_at::<false> (dynamics/spacecraft.rs:73): {same} of {len(at)} lines follow the same order as acceleration_paired"
              + ("" if same == len(at) else "  <-- the spacecraft path uses ANOTHER order: identify it separately"))
        if same != len(at):
            verdict = 1
    # powf
    if pw:
        cr = host = tot = 0
        for (o, i), got in pw.items():
            tot += 1
            cr += got == int(doc["pow"]["correctly_rounded"][str(o)][i], 16)
            host += got == int(doc["pow"]["generating_host_libm"][str(o)][i], 16)
        print(f"\npowf of the step-size controller (runge_kutta/mod.rs:238-239), {tot} values, a quarter of them chosen where libms disagree:")
        print(f"  {cr} equal the correctly rounded value (what the library computes, cr_pow)")
        print(f"  {host} equal the generating host's libm ({doc['pow']['generating_host']})")
        if cr != tot:
            print(f"  => on this platform {tot - cr} of these adversarial operands round differently from the library; in a propagation one")
            print("     controller call in about a thousand does, which changes that step's size by one ulp (DESIGN.md section 2).")
    return verdict


if __name__ == "__main__":
    sys.exit(main(sys.argv))
