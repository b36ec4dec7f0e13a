"""
Chain ingestion: GetDist's plain-text chain format -- ``root_1.txt, root_2.txt, ...`` (or
``root.txt``) with columns ``weight  -log(posterior)  param_1 ... param_n``, ``root.paramnames`` (``name[*]  label``
per line, ``*`` = derived) and ``root.ranges`` (``name  lower  upper``, ``N`` = unbounded) -- read into a getdist_amd
MCSamples.  Follows chains.py:77-125 (file matching, loadNumpyTxt), chains.py:228-246 / mcsamples.py:501-528
(ignore_rows burn-in per chain, fixed-parameter deletion, makeSingle) and paramnames.py / parampriors.py for the two
side files.  The parse itself is host work (pandas' C tokenizer when present: ~10x np.loadtxt); the columns go to the
device in the same SoA layout as array input.
"""

import os
import re

import numpy as np


def chainFiles(root, chain_indices=None, ext=".txt", separator="_", first_chain=0, last_chain=-1, chain_exclude=None):
    """chains.py:77-108: the chain files of ``root`` (``root.txt`` counts as index 0), sorted by name."""
    folder = os.path.dirname(root) or "."
    if root.endswith((os.sep, "/")):
        reg_exp = re.compile("(?P<num>[0-9]+)?" + re.escape(ext))
    else:
        reg_exp = re.compile(re.escape(os.path.basename(root)) + "(" + re.escape(separator) + "(?P<num>[0-9]+))?" + re.escape(ext))
    files = []
    for f in sorted(os.listdir(folder)):
        m = reg_exp.fullmatch(f)
        if m:
            index = int(m.group("num") or 0)
            if ((chain_indices is None or index in chain_indices) and (chain_exclude is None or index not in chain_exclude)
                    and index >= first_chain and (last_chain < 0 or index <= last_chain)):
                files.append(os.path.join(folder, f))
    return files


def loadNumpyTxt(fname, skiprows=None):
    """chains.py:115-125: a 2D float array from a whitespace-separated text file (``#`` comments allowed)."""
    try:
        import pandas as pd

        # round_trip: correctly rounded decimal -> double, bit-equal to np.loadtxt (the default fast parser is not)
        df = pd.read_csv(fname, sep=r"\s+", header=None, comment="#", skiprows=skiprows or 0, dtype=np.float64,
                         engine="c", float_precision="round_trip")
        return np.atleast_2d(df.to_numpy())
    except ImportError:
        return np.atleast_2d(np.loadtxt(fname, skiprows=skiprows or 0))
    except ValueError:
        print("Error reading %s" % fname)
        raise


def readParamNames(fname):
    """paramnames.py:95-110, 250-270: (names, labels, derived flags) of a .paramnames file."""
    names, labels, derived = [], [], []
    with open(fname, encoding="utf-8-sig") as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            parts = line.split(None, 1)
            name = parts[0]
            is_derived = name.endswith("*")
            names.append(name[:-1] if is_derived else name)
            derived.append(is_derived)
            labels.append(parts[1].strip() if len(parts) > 1 else None)
    return names, labels, derived


def readRanges(fname):
    """parampriors.py:30-60: name -> (lower, upper) with None for 'N'."""
    ranges = {}
    with open(fname, encoding="utf-8-sig") as f:
        for line in f:
            parts = line.split()
            if len(parts) >= 3 and not parts[0].startswith("#"):
                lo, hi = (None if v in ("N", "None") else float(v) for v in parts[1:3])
                ranges[parts[0]] = (lo, hi)
    return ranges


def loadMCSamples(file_root, settings=None, ignore_rows=None, chain_exclude=None, device=0, **kwargs):
    """
    mcsamples.py:47-146 for the plain-text format: read every chain file of ``file_root``, drop ``ignore_rows`` rows
    (a count if >= 1, else a fraction of each chain) as burn-in, delete the parameters that never move
    (chains.py:1029-1045), name / bound the rest from the side files and return an MCSamples holding the chains.
    """
    from .mcsamples import MCSamples, WeightedSampleError

    files = chainFiles(file_root, chain_exclude=chain_exclude) or chainFiles(file_root, separator=".", chain_exclude=chain_exclude)
    if not files:
        raise WeightedSampleError("loadChains - no chains found for " + file_root)
    if ignore_rows is None:
        ignore_rows = float((settings or {}).get("ignore_rows", 0))
    chains = []
    for fname in files:
        cols = loadNumpyTxt(fname, skiprows=int(ignore_rows) if ignore_rows >= 1 else None)
        if cols.shape[0] == 0 or cols.shape[1] < 3:
            continue  # "Ignored file (likely empty)" (chains.py:1400-1403)
        if 0 < ignore_rows < 1:
            cols = cols[int(round(cols.shape[0] * ignore_rows)):]
        chains.append(cols)
    if not chains:
        raise WeightedSampleError("loadChains - no chains found for " + file_root)
    n = chains[0].shape[1] - 2
    names = labels = derived = None
    if os.path.isfile(file_root + ".paramnames"):
        names, labels, derived = readParamNames(file_root + ".paramnames")
        if len(names) != n:
            raise WeightedSampleError("paramnames file does not match the number of chain columns")
    else:
        names = ["param%d" % (i + 1) for i in range(n)]
    # deleteFixedParams over the combined set (chains.py:1029-1045)
    allrows = np.vstack([c[:, 2:] for c in chains]) if len(chains) > 1 else chains[0][:, 2:]
    keep = []
    for i in range(n):
        col = allrows[:, i]
        fixed = np.isclose(col[0], col[-1], equal_nan=True) and np.allclose(col, np.average(col), rtol=1e-12, atol=0, equal_nan=True)
        if not fixed:
            keep.append(i)
    kept_names = [names[i] for i in keep]
    ranges = readRanges(file_root + ".ranges") if os.path.isfile(file_root + ".ranges") else {}
    s = {k: v for k, v in (settings or {}).items() if k != "ignore_rows"}
    mc = MCSamples(samples=[np.asfortranarray(c[:, 2:][:, keep]) for c in chains], weights=[c[:, 0] for c in chains],
                   loglikes=[c[:, 1] for c in chains], names=kept_names,
                   labels=None if labels is None else [labels[i] for i in keep],
                   ranges={k: v for k, v in ranges.items() if k in kept_names}, settings=s or None, device=device,
                   name_tag=os.path.basename(file_root), **kwargs)
    if derived is not None:
        for par, i in zip(mc.paramNames.names, keep):
            par.isDerived = derived[i]
    mc.root = file_root
    return mc
