#!/usr/bin/env python3
"""Development-time generator for the integrator coefficient tables.

Reads the coefficient *data* (Butcher tableaux, multistep alpha/beta rows) out of the reference's
`integration/src/methods.rs` and `integration/src/multistep/second_order/cowell.rs`, pushes every entry
through an emulation of the reference's `Ratio` arithmetic (`integration/src/ratio.rs:44-49` const_new +
normalize `:153-177`, `from_f64` `:75-103`, `const_sub` `:123-139`) so that each coefficient ends up as the
exact (numer, denom) i128 pair the Rust constant holds, and writes a C include file with those integer
pairs.  The f64 value used at run time is `(double)numer / (double)denom` (`ratio.rs:221-228`), computed by
the consumer (oracle and product each do the conversion themselves).

Only run in the build container (it needs /root/reference); the generated .inc files are committed.
No reference source text is emitted: only integers.

usage: python tools/gen_coeffs.py [/root/reference] -> writes
    oracle/coeff_tables.inc
    ephemeris_explorer_amd/csrc/coeff_tables.inc
    tests/golden/coeff_tables.json   (same data + f64 hex values, for the tests)
"""
import json
import math
import re
import sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
ROOT = Path(__file__).resolve().parent.parent

I128_MAX = (1 << 127) - 1
I128_MIN = -(1 << 127)


def _chk(v):
    assert I128_MIN <= v <= I128_MAX, "i128 overflow in Ratio emulation"
    return v


class Ratio:
    """Emulation of integration::ratio::Ratio<i128> (ratio.rs)."""

    __slots__ = ("n", "d")

    def __init__(self, n, d, normalize=True):
        self.n, self.d = int(n), int(d)
        if normalize:
            self._normalize()

    def _normalize(self):  # ratio.rs:153-177
        if self.d == 0:
            return
        if self.n == 0:
            self.d = 1
            return
        if self.n == self.d:
            self.n = self.d = 1
            return
        g = math.gcd(abs(self.n), abs(self.d))
        # Rust: `self.numer /= g` truncating division on i128; exact here since g divides both
        self.n = int(self.n / g) if False else (abs(self.n) // g) * (1 if self.n > 0 else -1)
        self.d = (abs(self.d) // g) * (1 if self.d > 0 else -1)
        if self.d < 0:
            self.n, self.d = -self.n, -self.d

    def const_sub(self, rhs):  # ratio.rs:123-139 (result is NOT normalized)
        if self.d == rhs.d:
            return Ratio(_chk(self.n - rhs.n), self.d, normalize=False)
        l = abs(self.d) * abs(rhs.d) // math.gcd(abs(self.d), abs(rhs.d))
        _chk(l)
        ln = _chk(self.n * (l // self.d))
        rn = _chk(rhs.n * (l // rhs.d))
        return Ratio(_chk(ln - rn), l, normalize=False)

    def const_add(self, rhs):
        if self.d == rhs.d:
            return Ratio(_chk(self.n + rhs.n), self.d, normalize=False)
        l = abs(self.d) * abs(rhs.d) // math.gcd(abs(self.d), abs(rhs.d))
        ln = _chk(self.n * (l // self.d))
        rn = _chk(rhs.n * (l // rhs.d))
        return Ratio(_chk(ln + rn), l, normalize=False)

    def to_f64(self):  # ratio.rs:221-228: numer as f64 / denom as f64
        return float(self.n) / float(self.d)

    def __repr__(self):
        return f"Ratio({self.n}/{self.d})"


def frac(n, d):  # methods.rs:53-57
    return Ratio(_chk(n), _chk(d))


def frac_f64(val):  # methods.rs:59-65 -> ratio.rs:75-103
    val = float(val)
    assert not (math.isnan(val) or math.isinf(val))
    p = 0
    new_val = val
    while True:
        a = abs(new_val)
        # `(new_val.abs() as u64) as f64 == new_val.abs()`  (saturating float->u64 cast)
        au = int(a) if a < 18446744073709551616.0 else 18446744073709551615
        if float(au) == a:
            break
        p += 1
        new_val = val * float(10 ** p)  # `ten.pow(p) as f64`
        assert not math.isinf(new_val)
    return Ratio(int(new_val), 10 ** p)


def strip_comments(src):
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return src


def find_block(src, start):
    """src[start] == '{' -> index one past the matching '}'."""
    depth = 0
    for i in range(start, len(src)):
        c = src[i]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return i + 1
    raise ValueError("unbalanced braces")


def split_const_items(body):
    """Yield (name, value_text) for each `const NAME: TYPE = VALUE;` at depth 0 of body."""
    i = 0
    out = []
    pat = re.compile(r"const\s+([A-Z_][A-Z0-9_]*)\s*:\s*([^=]+?)=\s*", re.S)
    while True:
        m = pat.search(body, i)
        if not m:
            break
        j = m.end()
        depth = 0
        k = j
        while k < len(body):
            c = body[k]
            if c in "{[(":
                depth += 1
            elif c in "}])":
                depth -= 1
            elif c == ";" and depth == 0:
                break
            k += 1
        out.append((m.group(1), body[j:k].strip()))
        i = k + 1
    return out


def rust_to_py(expr):
    e = expr
    e = e.replace("frac_f64!(", "frac_f64(").replace("frac!(", "frac(")
    e = e.replace("&[", "[")
    e = re.sub(r"Self::([A-Z_]+)", r"SELF['\1']", e)
    e = re.sub(r"(?<![A-Za-z_.])(\d[\d_]*)(?![\d.eE_])", lambda m: m.group(1).replace("_", ""), e)
    e = e.replace("true", "True").replace("false", "False")
    return e


def eval_value(text, selfenv):
    text = text.strip()
    env = {"frac": frac, "frac_f64": frac_f64, "SELF": selfenv}
    if text.startswith("{"):
        inner = text[1 : find_block(text, 0) - 1]
        last_end = 0
        for name, val in split_const_items(inner):
            env[name] = eval_value(val, selfenv) if val.lstrip().startswith(("{", "&[")) else eval(rust_to_py(val), env)
        # the tail expression is whatever follows the last ';' at depth 0
        depth = 0
        for k, c in enumerate(inner):
            if c in "{[(":
                depth += 1
            elif c in "}])":
                depth -= 1
            elif c == ";" and depth == 0:
                last_end = k + 1
        return eval(rust_to_py(inner[last_end:].strip()), env)
    return eval(rust_to_py(text), env)


def parse_impls(src, traits):
    """-> {struct_name: {const_name: value}} merged over all matching `impl <Trait> for <Name>` blocks."""
    out = {}
    for m in re.finditer(r"impl(?:<[^>]*>)?\s+([A-Za-z0-9_]+)\s+for\s+([A-Za-z0-9_<>]+)\s*\{", src):
        trait, name = m.group(1), m.group(2)
        if trait not in traits:
            continue
        end = find_block(src, m.end() - 1)
        body = src[m.end() : end - 1]
        d = out.setdefault(name, {})
        for cname, val in split_const_items(body):
            d[cname] = eval_value(val, d)
    return out


def main():
    methods = strip_comments((REF / "integration/src/methods.rs").read_text())
    cowell = strip_comments((REF / "integration/src/multistep/second_order/cowell.rs").read_text())

    tabs = parse_impls(
        methods,
        {
            "ERKCoefficients", "EERKCoefficients", "SRKNCoefficients", "ELM2Coefficients", "ELM1Coefficients",
            "ERKNCoefficients", "EERKNCoefficients", "ERKNGCoefficients", "EERKNGCoefficients",
        },
    )
    cow = parse_impls(cowell, {"CowellVelocityCoefficients"})

    def rat(r):
        assert isinstance(r, Ratio), r
        return [r.n, r.d]

    def conv(v):
        if isinstance(v, Ratio):
            return rat(v)
        if isinstance(v, bool):
            return v
        if isinstance(v, int):
            return v
        if isinstance(v, list):
            return [conv(x) for x in v]
        raise TypeError(v)

    data = {"methods": {k: {c: conv(v) for c, v in d.items()} for k, d in tabs.items()},
            "cowell": {k: {c: conv(v) for c, v in d.items()} for k, d in cow.items()}}

    # ---- JSON for tests (with f64 hex of every rational) -------------------------------------------
    def hexify(v):
        if isinstance(v, list) and len(v) == 2 and all(isinstance(x, int) and not isinstance(x, bool) for x in v) and v[1] != 0 and False:
            return v
        return v

    (ROOT / "tests/golden").mkdir(parents=True, exist_ok=True)
    jd = {"methods": {}, "cowell": data["cowell"]}
    for name, d in tabs.items():
        e = {}
        for c, v in d.items():
            if isinstance(v, list) and v and isinstance(v[0], Ratio):
                e[c] = {"ratio": [[str(r.n), str(r.d)] for r in v], "f64hex": [r.to_f64().hex() for r in v]}
            elif isinstance(v, list) and v and isinstance(v[0], list):
                e[c] = {"ratio": [[[str(r.n), str(r.d)] for r in row] for row in v],
                        "f64hex": [[r.to_f64().hex() for r in row] for row in v]}
            elif isinstance(v, list):
                e[c] = [str(x) for x in v]
            else:
                e[c] = v if isinstance(v, bool) else str(v)
        jd["methods"][name] = e
    jd["cowell"] = {k: {c: ([str(x) for x in v] if isinstance(v, list) else str(v)) for c, v in d.items()}
                    for k, d in cow.items()}
    (ROOT / "tests/golden/coeff_tables.json").write_text(json.dumps(jd, indent=1) + "\n")

    # ---- C include ---------------------------------------------------------------------------------
    def i128(v):
        """C initializer for an i128 split as {int64 hi, uint64 lo} (two's complement)."""
        u = v & ((1 << 128) - 1)
        hi = u >> 64
        if hi >= 1 << 63:
            hi -= 1 << 64
        return f"{{{hi}LL, {u & ((1 << 64) - 1)}ULL}}"

    def ratio_c(r):
        return f"{{{i128(r.n)}, {i128(r.d)}}}"

    L = []
    L.append("/* GENERATED by tools/gen_coeffs.py -- integer (numer, denom) pairs of every integrator coefficient,")
    L.append(" * exactly as the reference's `Ratio` constants hold them (integration/src/methods.rs,")
    L.append(" * integration/src/multistep/second_order/cowell.rs, integration/src/ratio.rs). Data only.")
    L.append(" * Consumer must define: EPH_I128 {int64 hi; uint64 lo}, EPH_RATIO {EPH_I128 n, d}. */")

    def emit_ratio_array(sym, rs):
        L.append(f"static const EPH_RATIO {sym}[{max(len(rs), 1)}] = {{")
        for r in rs:
            L.append(f"  {ratio_c(r)},")
        if not rs:
            L.append("  {{0,0},{0,1}},")
        L.append("};")

    def emit_int_array(sym, vs):
        L.append(f"static const EPH_I128 {sym}[{len(vs)}] = {{")
        for v in vs:
            L.append(f"  {i128(v)},")
        L.append("};")

    erk, srkn, elm2, elm1, erkn, erkng = [], [], [], [], [], []
    for name, d in tabs.items():
        if "A" in d and d["A"] and isinstance(d["A"][0], list) or (name == "RK4"):
            if "BP" in d:  # ERKN
                erkn.append(name)
                continue
            s = len(d["B"])
            flat = [r for row in d["A"] for r in row]
            assert [len(row) for row in d["A"]] == list(range(s)), name
            emit_ratio_array(f"eph_{name}_A", flat)
            emit_ratio_array(f"eph_{name}_B", d["B"])
            emit_ratio_array(f"eph_{name}_C", d["C"])
            if "E" in d:
                assert len(d["E"]) == s
                emit_ratio_array(f"eph_{name}_E", d["E"])
            erk.append(name)
        elif "AP" in d:
            erkng.append(name)
        elif "ALPHA" in d:
            emit_int_array(f"eph_{name}_ALPHA", d["ALPHA"])
            emit_int_array(f"eph_{name}_BETA_N", d["BETA_N"])
            (elm2 if name in ("QuinlanTremaine12", "Stormer13") else elm1).append(name)
        elif "A" in d:
            assert len(d["A"]) == len(d["B"])
            emit_ratio_array(f"eph_{name}_A", d["A"])
            emit_ratio_array(f"eph_{name}_B", d["B"])
            srkn.append(name)
    for name in erkn:
        d = tabs[name]
        s = len(d["BP"])
        assert [len(row) for row in d["A"]] == list(range(s)), name
        emit_ratio_array(f"eph_{name}_A", [r for row in d["A"] for r in row])
        for k in ("BP", "BV", "C", "EP", "EV"):
            emit_ratio_array(f"eph_{name}_{k}", d[k])
    for name in erkng:
        d = tabs[name]
        s = len(d["BP"])
        for k in ("AP", "AV"):
            assert [len(row) for row in d[k]] == list(range(s)), (name, k)
            emit_ratio_array(f"eph_{name}_{k}", [r for row in d[k] for r in row])
        for k in ("BP", "BV", "C", "EP", "EV"):
            emit_ratio_array(f"eph_{name}_{k}", d[k])
    for cname, d in cow.items():
        order = int(re.search(r"<(\d+)>", cname).group(1))
        emit_int_array(f"eph_Cowell{order}_BETA_N", d["BETA_N"])

    L.append("typedef struct { const char *name; int stages, order, order_embedded, fsal;")
    L.append("  const EPH_RATIO *A, *B, *C, *E; } EPH_ERK_TABLE;")
    L.append(f"static const EPH_ERK_TABLE eph_erk_tables[{len(erk)}] = {{")
    for name in erk:
        d = tabs[name]
        E = f"eph_{name}_E" if "E" in d else "0"
        L.append(f'  {{"{name}", {len(d["B"])}, {d["ORDER"]}, {d.get("ORDER_EMBEDDED", 0)}, {int(d["FSAL"])}, '
                 f"eph_{name}_A, eph_{name}_B, eph_{name}_C, {E}}},")
    L.append("};")
    L.append(f"#define EPH_N_ERK_TABLES {len(erk)}")

    L.append("typedef struct { const char *name; int stages, fsal; const EPH_RATIO *A, *B; } EPH_SRKN_TABLE;")
    L.append(f"static const EPH_SRKN_TABLE eph_srkn_tables[{len(srkn)}] = {{")
    for name in srkn:
        d = tabs[name]
        L.append(f'  {{"{name}", {len(d["A"])}, {int(d["FSAL"])}, eph_{name}_A, eph_{name}_B}},')
    L.append("};")
    L.append(f"#define EPH_N_SRKN_TABLES {len(srkn)}")

    L.append("typedef struct { const char *name; int order; const EPH_I128 *ALPHA, *BETA_N; EPH_I128 BETA_D;")
    L.append("  const EPH_I128 *COWELL_N; EPH_I128 COWELL_D; } EPH_ELM2_TABLE;")
    L.append(f"static const EPH_ELM2_TABLE eph_elm2_tables[{len(elm2)}] = {{")
    for name in elm2:
        d = tabs[name]
        o = d["ORDER"]
        c = cow[f"Cowell<{o}>"]
        assert len(d["ALPHA"]) == o + 1 and len(d["BETA_N"]) == o + 1 and len(c["BETA_N"]) == o
        L.append(f'  {{"{name}", {o}, eph_{name}_ALPHA, eph_{name}_BETA_N, {i128(d["BETA_D"])}, '
                 f'eph_Cowell{o}_BETA_N, {i128(c["BETA_D"])}}},')
    L.append("};")
    L.append(f"#define EPH_N_ELM2_TABLES {len(elm2)}")

    L.append("typedef struct { const char *name; int order; const EPH_I128 *ALPHA, *BETA_N; EPH_I128 BETA_D; } EPH_ELM1_TABLE;")
    L.append(f"static const EPH_ELM1_TABLE eph_elm1_tables[{len(elm1)}] = {{")
    for name in elm1:
        d = tabs[name]
        L.append(f'  {{"{name}", {d["ORDER"]}, eph_{name}_ALPHA, eph_{name}_BETA_N, {i128(d["BETA_D"])}}},')
    L.append("};")
    L.append(f"#define EPH_N_ELM1_TABLES {len(elm1)}")

    L.append("typedef struct { const char *name; int stages, order, order_embedded, fsal;")
    L.append("  const EPH_RATIO *A, *BP, *BV, *C, *EP, *EV; } EPH_ERKN_TABLE;")
    L.append(f"static const EPH_ERKN_TABLE eph_erkn_tables[{max(len(erkn),1)}] = {{")
    for name in erkn:
        d = tabs[name]
        L.append(f'  {{"{name}", {len(d["BP"])}, {d["ORDER"]}, {d["ORDER_EMBEDDED"]}, {int(d["FSAL"])}, '
                 f"eph_{name}_A, eph_{name}_BP, eph_{name}_BV, eph_{name}_C, eph_{name}_EP, eph_{name}_EV}},")
    L.append("};")
    L.append(f"#define EPH_N_ERKN_TABLES {len(erkn)}")

    L.append("typedef struct { const char *name; int stages, order, order_embedded, fsal;")
    L.append("  const EPH_RATIO *AP, *AV, *BP, *BV, *C, *EP, *EV; } EPH_ERKNG_TABLE;")
    L.append(f"static const EPH_ERKNG_TABLE eph_erkng_tables[{max(len(erkng),1)}] = {{")
    for name in erkng:
        d = tabs[name]
        L.append(f'  {{"{name}", {len(d["BP"])}, {d["ORDER"]}, {d["ORDER_EMBEDDED"]}, {int(d["FSAL"])}, '
                 f"eph_{name}_AP, eph_{name}_AV, eph_{name}_BP, eph_{name}_BV, eph_{name}_C, eph_{name}_EP, eph_{name}_EV}},")
    L.append("};")
    L.append(f"#define EPH_N_ERKNG_TABLES {len(erkng)}")

    text = "\n".join(L) + "\n"
    for p in (ROOT / "oracle/coeff_tables.inc", ROOT / "ephemeris_explorer_amd/csrc/coeff_tables.inc"):
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(text)
    print("erk:", erk)
    print("srkn:", srkn)
    print("elm2:", elm2, "elm1:", elm1, "erkn:", erkn, "erkng:", erkng)
    print("cowell orders:", sorted(int(re.search(r'<(\d+)>', k).group(1)) for k in cow))


if __name__ == "__main__":
    main()
