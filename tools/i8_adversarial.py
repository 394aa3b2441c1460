"""The int8-sliced GEMM tile (csrc/nlc_i8gemm.h) on ADVERSARIAL rows, against exact rational arithmetic.

`tools/i8gemm_check.bin --adversarial FILE` runs the device code on rows built to break a fixed-point product -- cancelling sums,
a 2^+-40 spread inside a row with large weights on small states, h = +-1 exactly, every |h| < 2^-54, tiny states among ordinary
ones -- and dumps inputs and outputs; this module recomputes every output with python `fractions` (exact) and reports, per family,
the error in the two units that matter:

  rel  = |err| / (2^-53 sum_k |w_k h_k|)                 what an FP64 fused-multiply-add chain is bounded in (<= K units, ~5 seen)
  abs  = |err| / (2^-54 (sum_k |w_k| + s_m sum_k |h_k|))  the fixed-point operands' own quantisation: |dh| <= 2^-55, |dw| <= 2^-55 s_m

The sliced product's guarantee is ABSOLUTE (|err| <= abs unit + 5 rel units, asserted for every family): for ordinary states it is
also within 5 rel units like the FP64 chain (families 0, 2, 5); where sum |w h| is far below s_m -- states below the 2^-55 grid --
it cannot be, and is not, relatively accurate (families 1, 3, 4), while its absolute error there is still below 2^-48 s_m, i.e.
below the rounding of the bias the GRU adds next.  Run as a script on the GPU box: prints the table as JSON
(profiles/r6_i8_adversarial.json).
"""
import json
import os
import subprocess
import sys
from fractions import Fraction

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(REPO, "tools", "i8gemm_check.bin")
FAMILIES = ["cancelling rows", "2^+-40 spread, large w on small h", "h = +-1, |w| = row scale", "all |h| < 2^-54",
            "tiny h among ordinary ones", "random (control)"]
RELATIVE_FAMILIES = (0, 2, 5)  # ordinary state magnitudes: the FP64 chain's relative bound must hold as well


def build_tool():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "neurallaplacecontrol_amd", "csrc"), "tools"])
    return EXE


def run_dump(path):
    res = subprocess.run([build_tool(), "--adversarial", path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert res.returncode == 0, res.stdout.decode()


def read_dump(path):
    raw = open(path, "rb").read()
    n = int(np.frombuffer(raw, dtype=np.int32, count=1)[0])
    pos, tiles = [4], []

    def take(count, shape):
        a = np.frombuffer(raw, dtype=np.float64, count=count, offset=pos[0]).reshape(shape)
        pos[0] += 8 * count
        return a

    for _ in range(n):
        fam = int(np.frombuffer(raw, dtype=np.int32, count=1, offset=pos[0])[0])
        pos[0] += 4
        tiles.append(dict(family=fam, W=take(16 * 64, (16, 64)), scale=take(16, (16,)), h=take(64 * 16, (64, 16)),
                          merged=take(256, (16, 16)), plain=take(256, (16, 16))))
    assert pos[0] == len(raw)
    return tiles


def check(tiles):
    """Per family: outputs, max rel / abs units over both recombination forms, and the worst value of |err| / (abs + 5 rel)."""
    F = Fraction
    rows = {f: dict(family=FAMILIES[f], outputs=0, max_rel_units=0.0, max_abs_units=0.0, max_of_bound=0.0, max_abs_err_over_scale=0.0)
            for f in range(len(FAMILIES))}
    for t in tiles:
        Wf = [[F(float(x)) for x in row] for row in t["W"]]
        hf = [[F(float(t["h"][k, n])) for k in range(64)] for n in range(16)]
        for r in range(16):
            s = F(float(t["scale"][r]))
            sum_w = sum(abs(w) for w in Wf[r])
            for n in range(16):
                exact = sum(w * h for w, h in zip(Wf[r], hf[n]))
                mag = sum(abs(w * h) for w, h in zip(Wf[r], hf[n]))
                u_rel = mag * F(1, 2**53)
                u_abs = (sum_w + s * sum(abs(h) for h in hf[n])) * F(1, 2**54)
                row = rows[t["family"]]
                for form in ("merged", "plain"):
                    err = abs(F(float(t[form][r, n])) - exact)
                    row["max_rel_units"] = max(row["max_rel_units"], float(err / u_rel) if u_rel else 0.0)
                    row["max_abs_units"] = max(row["max_abs_units"], float(err / u_abs))
                    row["max_of_bound"] = max(row["max_of_bound"], float(err / (u_abs + 5 * u_rel)))
                    row["max_abs_err_over_scale"] = max(row["max_abs_err_over_scale"], float(err / s))
                row["outputs"] += 1
    return [rows[f] for f in sorted(rows)]


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/i8_adversarial.bin"
    run_dump(out)
    table = check(read_dump(out))
    print(json.dumps(dict(tool="tools/i8gemm_check.bin --adversarial + tools/i8_adversarial.py (python fractions)",
                          units=dict(rel="2^-53 sum|w h|", abs="2^-54 (sum|w| + s_m sum|h|)", bound="abs + 5 rel",
                                     abs_err_over_scale="|err| / s_m; 2^-48 = 3.55e-15"), families=table), indent=1))
