"""
FFT-size helper of getdist/convolve.py:5-193 for API compatibility.  The reference picks its zero-padded FFT
sizes from a literal table; the same numbers are generated here from their rule (2^a 3^b 5^c, c <= 1, for the row
ranges below, plus the table's two stragglers).  The device pipeline plans its own frame sizes
(mcsamples.next_fft_size); any zero padding gives the same linear convolution.
"""

import numpy as np

_ROWS = {(0, 0): (1, 29), (0, 1): (1, 28), (1, 0): (1, 29), (1, 1): (5, 27), (2, 0): (4, 27), (2, 1): (4, 25), (3, 0): (4, 26)}
fastFFT = np.array(sorted([7 * 2**25, 81 * 2**24] + [2**a * 3**b * 5**c for (b, c), (lo, hi) in _ROWS.items()
                                                      for a in range(lo, hi + 1)]), dtype=np.int64)


def nearestFFTnumber(x):
    return np.maximum(x, fastFFT[np.searchsorted(fastFFT, x)])
