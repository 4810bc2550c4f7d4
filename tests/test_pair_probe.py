"""The identification kit of the unpinned point-mass term (tools/pair_probe.py, tools/identify_pair_variant.py,
tests/golden/pair_probe.json): the committed expected bits are what the CPU oracle gives in each of its seven orders, the
operands really do tell the orders apart, the identifier names each order from a fabricated print-out and finds an expression
tree outside the seven; the controller's pow column is the oracle's correctly rounded cr_pow."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import identify_pair_variant as ident  # noqa: E402
import pairexpr as pe  # noqa: E402

from oracle import orc  # noqa: E402


@pytest.fixture(scope="module")
def probe():
    return ident.load_probe()


def _norm(u):
    return 0 if u == 1 << 63 else u          # the oracle accumulates into +0: a -0 term arrives as +0


def test_expected_bits_are_the_oracles_in_every_order(probe):
    doc, ops = probe
    try:
        for k in range(7):
            orc.set_pair_variant(k)
            for i, (pi, mui, pj, muj) in enumerate(ops):
                a = orc.gravity(np.array([pi, pj]), np.array([mui, muj]))
                got = [int(v) for v in a.reshape(-1).view(np.uint64)]
                want = [_norm(int(h, 16)) for h in doc["expected"][str(k)][i]]
                assert got == want, (k, i)
    finally:
        orc.set_pair_variant(0)


def test_separating_operands_separate_every_pair_of_orders(probe):
    doc, _ = probe
    n = doc["n_separating"]
    assert n >= 64
    for i in range(n):
        rows = [tuple(doc["expected"][str(k)][i]) for k in range(7)]
        assert len(set(rows)) == 7, i


def test_identifier_names_each_built_order(probe, capsys):
    doc, _ = probe
    for k in range(7):
        assert ident.main(["identify", "--emulate", str(k)]) == 0
        assert f"\nk = {k} " in capsys.readouterr().out


def test_identifier_finds_a_tree_outside_the_seven(probe, tmp_path, capsys):
    doc, ops = probe
    tree = pe.Tree("dot_lr", "inv", "recip(r*r*r)", "(d*inv)*mu")          # not one of the built orders
    lines = []
    for i, op in enumerate(ops):
        ai, aj = tree.paired(*op)
        lines.append(f"pair {i} " + " ".join(pe.hexbits(v) for v in ai + aj))
    f = tmp_path / "printout.txt"
    f.write_text("\n".join(lines))
    assert ident.main(["identify", str(f)]) == 1
    out = capsys.readouterr().out
    assert "none of the seven built orders" in out and "reproduce every printed value" in out
    assert "let r = n2.sqrt(); 1.0 / (r * r * r)" in out and "(d*inv)*mu" in out


def test_pow_column_is_the_oracles_correctly_rounded_pow(probe):
    doc, _ = probe
    differ = 0
    for k in doc["pow"]["orders"]:
        y = -(1.0 / float(k))
        for e, want, host in zip(doc["pow"]["err"][str(k)], doc["pow"]["correctly_rounded"][str(k)],
                                 doc["pow"]["generating_host_libm"][str(k)]):
            assert pe.bits(orc.cr_pow(pe.from_bits(int(e, 16)), y)) == int(want, 16)
            differ += want != host
    assert differ >= 32            # the operands chosen to tell libms apart do


def test_generator_is_deterministic(tmp_path):
    before = (ROOT / "tests/golden/pair_probe.json").read_text()
    subprocess.check_call([sys.executable, str(ROOT / "tools/pair_probe.py")], stdout=subprocess.DEVNULL)
    assert (ROOT / "tests/golden/pair_probe.json").read_text() == before
