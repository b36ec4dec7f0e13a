"""
Worker process of the host TNC pool: `python -m getdist_amd._tnc_worker`.

Reads length-prefixed pickled job lists from stdin, writes length-prefixed pickled result lists to stdout.  It is
started with subprocess (NOT multiprocessing), so a user script without an `if __name__ == "__main__"` guard is
never re-imported in the workers, and no HIP state is inherited.
"""

import pickle
import struct
import sys

import numpy as np


def main():
    from getdist_amd.mcsamples import _get_h  # numpy/scipy only; creates no GPU context

    inp, out = sys.stdin.buffer, sys.stdout.buffer
    while True:
        head = inp.read(8)
        if len(head) < 8:
            return
        (n,) = struct.unpack("<q", head)
        jobs = pickle.loads(inp.read(n))
        res = []
        for psi, neff, corr, do_corr in jobs:
            try:
                # plain floats travel over the pipe (cheap to pickle); numpy scalar semantics are restored here
                # (np.float64 ** gives nan where a Python float would raise / go complex, as in the reference)
                h = _get_h(tuple(np.float64(v) for v in psi), np.float64(neff), np.float64(corr) if corr else corr, do_corr)
                res.append(tuple(float(v) for v in h))
            except Exception as e:  # e.g. "bias not positive definite": re-raised in the parent
                res.append(e)
        blob = pickle.dumps(res, protocol=pickle.HIGHEST_PROTOCOL)
        out.write(struct.pack("<q", len(blob)))
        out.write(blob)
        out.flush()


if __name__ == "__main__":
    main()
