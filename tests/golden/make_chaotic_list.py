"""
Which pairs of the fixture zoo are CHAOTIC IN THE ORACLE: their bandwidth goes through TNC (kde_bandwidth.py:276-299) and the
oracle's own get_h moves by more than 1e-6 when its functionals are perturbed by +-1..12 x 1e-15 (oracle.get_h_ensemble) --
the only pairs whose device grid may differ from the oracle's by more than 1e-6.  Oracle only (numpy /
scipy; no reference import, no GPU):

    python tests/golden/make_chaotic_list.py   ->   tests/golden/tnc_chaotic_pairs.json

tests/test_oracle_golden.py recomputes the list and compares it with the committed one; tests/test_gpu_densities.py admits a
loose pair only if its name is on it.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import kde_oracle as ko  # noqa: E402

PATH = os.path.join(HERE, "tnc_chaotic_pairs.json")


def chaotic_pairs_of(fx):
    """[(key, oracle_moves_by)] of one fixture; key = '<fixture>/p2d/<x>/<y>/<kwargs key>' as the GPU test spells it."""
    import golden_util as gu

    out = []
    if not fx["pairs"]:
        return out
    orc = ko.OracleSamples(fx["samples"], fx["weights"], names=fx["names"], ranges=fx["ranges"])
    for kw in fx["kw2"]:
        for (a, b) in fx["pairs"]:
            tr = {}
            orc.density_2d(a, b, trace=tr, **kw)
            if "p_13" not in tr:  # no TNC on this pair (a bounded parameter, the rule-of-thumb branch, a fixed scale)
                continue
            psi = (tr["p_02"], tr["p_20"], tr["p_11"], tr["p_00"], tr["p_13"], tr["p_31"])
            ens = ko.get_h_ensemble(psi, tr["opt_N"], tr["opt_corr"])
            moved = float(np.max(np.abs(ens - ens[0])) / np.max(np.abs(ens[0])))
            if moved > 1e-6:
                out.append(("%s/p2d/%s/%s/%s" % (fx["name"], fx["names"][a], fx["names"][b], gu.kwkey(kw)), moved))
    return out


def main():
    from oracle.fixtures import fixture_zoo

    rows = []
    for fx in fixture_zoo():
        rows += chaotic_pairs_of(fx)
    json.dump(dict(note="pairs of the fixture zoo on which the ORACLE's get_h moves by more than 1e-6 under +-1..12e-15 "
                        "perturbations of its own functionals (tests/golden/make_chaotic_list.py)",
                   pairs=sorted(k for k, _ in rows), oracle_moves_by={k: v for k, v in rows}), open(PATH, "w"), indent=1, sort_keys=True)
    print(len(rows), "chaotic pairs ->", PATH)


if __name__ == "__main__":
    main()
