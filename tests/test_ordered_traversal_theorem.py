"""The exactness statement behind the planned near-first traversal, checked against the CPU
oracle's fixed-order BVH walk on millions of rays: see tests/native/ordered_theorem_check.c."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_argmin_with_walk_order_ties_is_the_reference_result_when_the_winner_is_safe(tmp_path, orc):
    exe = str(tmp_path / "ordered_theorem_check")
    odir = os.path.join(ROOT, "oracle")
    subprocess.run(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-Wall", "-Wextra", "-I" + odir,
                    os.path.join(ROOT, "tests", "native", "ordered_theorem_check.c"), "-L" + odir, "-loracle", "-lm",
                    "-Wl,-rpath," + odir, "-o", exe], check=True)
    r = subprocess.run([exe, "150000"], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = re.findall(r"(\S+)\s+(\d+) prims\s+(\d+) rays:\s+(\d+) hits,\s+(\d+) exact-t ties, unsafe winners (\d+) .*safe mismatches (\d+)", r.stdout)
    assert len(rows) == 6
    for name, _n, _rays, hits, ties, unsafe, bad in rows:
        assert int(bad) == 0, name
        assert int(hits) > 1000, name
        assert int(unsafe) <= 0.02 * int(hits), name               # the fallback must stay rare
    m = re.search(r"hit point outside its box by ([0-9.e+-]+) D", r.stdout)
    assert m and float(m.group(1)) <= 5.5e-3                                          # the spatial bound behind walk_ordered's culling band
    m = re.search(r"worst backward error of Sphere::hit: .* = ([0-9.e+-]+) ", r.stdout)
    assert m and float(m.group(1)) <= 1.43e-6                                         # the derived constant 24 u of DESIGN.md section 10
    by = {row[0]: row for row in rows}
    assert int(by["cornell"][4]) > 0 and int(by["coincident"][4]) > 0          # exact ties did occur and were resolved by walk order
