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
    if os.path.getsize(fname) == 0:
        return np.zeros((0, 0))
    try:
        import pandas as pd

        # round_trip: correctly rounded decimal -> double, bit-equal to np.loadtxt (the default fast parser is not)
        df = pd.read_csv(fname, sep=r"\s+", header=None, comment="#", skiprows=skiprows or 0, dtype=np.float64,
                         engine="c", float_precision="round_trip")
        arr = np.atleast_2d(df.to_numpy())
        if np.isnan(arr).any():
            # pandas pads a short row with NaN where np.loadtxt -- the reference -- raises: a chain file that is still
            # being written must fail loudly, not yield a NaN sample (and poison the binary cache).  NaN written in the
            # file itself parses identically in np.loadtxt, which then decides.
            # (a ValueError of the reference's parser is reported once, by the handler below: chains.py:121-125)
            return np.atleast_2d(np.loadtxt(fname, skiprows=skiprows or 0))
        return arr
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


CACHE_MAGIC = b"GDAMDSOA1\n"
CACHE_ALIGN = 4096


def cache_path(file_root):
    return file_root + ".gdamd_soa"


def write_soa_cache(path, chains):
    """
    Binary chain cache (the role of the reference's ``.py_mcsamples`` pickle, mcsamples.py:83-126, in a layout made for
    the device): magic, one JSON header line (rows per chain, columns), zero padding to a 4096-byte boundary, then the
    stacked chain rows as ONE column-major fp64 block -- column 0 = weight, 1 = -log(posterior), 2.. = parameters --
    i.e. exactly the SoA layout of the device (ctx.hpp), so a load is one sequential read into page-locked memory
    followed by per-column DMA with no transpose and no text parse.
    """
    import json

    rows = [int(c.shape[0]) for c in chains]
    ncol = int(chains[0].shape[1])
    head = json.dumps(dict(rows=rows, ncol=ncol, dtype="<f8")).encode() + b"\n"
    pad = (-(len(CACHE_MAGIC) + len(head))) % CACHE_ALIGN
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(CACHE_MAGIC + head + b"\0" * pad)
        for j in range(ncol):
            for c in chains:
                np.ascontiguousarray(c[:, j], dtype="<f8").tofile(f)
    os.replace(tmp, path)


def read_soa_cache(path, alloc=None):
    """(column-major (N, ncol) array, rows per chain) from a cache file.  ``alloc(shape, dtype)`` supplies the host
    buffer (page-locked memory from the device context); the file is read straight into it."""
    import json

    with open(path, "rb") as f:
        if f.read(len(CACHE_MAGIC)) != CACHE_MAGIC:
            raise ValueError("not a getdist_amd chain cache: " + path)
        head = json.loads(f.readline().decode())
        pos = f.tell()
        f.seek(pos + (-pos) % CACHE_ALIGN)
        N, ncol = int(sum(head["rows"])), int(head["ncol"])
        flat = alloc((N * ncol,), np.float64) if alloc is not None else np.empty(N * ncol)
        got = f.readinto(memoryview(flat).cast("B"))
        if got != N * ncol * 8:
            raise ValueError("truncated chain cache: " + path)
    return flat.reshape((ncol, N)).T, head["rows"]


def read_root(file_root, chain_exclude=None, no_cache=False, alloc=None):
    """
    Everything ``MCSamples(root=...)`` / ``loadMCSamples`` read from disk (mcsamples.py:47-146, chains.py:1368-1405):
    the chain files (or the binary cache when it is newer than all of them), ``.paramnames`` and ``.ranges``.
    Returns dict(samples=[per-chain (rows, n)], weights=[...], loglikes=[...], names, labels, derived, ranges,
    from_cache).  No burn-in is removed here.
    """
    from .mcsamples import WeightedSampleError

    files = chainFiles(file_root, chain_exclude=chain_exclude) or chainFiles(file_root, separator=".", chain_exclude=chain_exclude)
    if not files:
        raise OSError("No chains found: " + file_root)
    if chain_exclude:
        no_cache = True  # mcsamples.py:73-74
    cpath = cache_path(file_root)
    chains = None
    from_cache = False
    if not no_cache and os.path.isfile(cpath) and max(os.path.getmtime(f) for f in files) < os.path.getmtime(cpath):
        try:
            block, rows = read_soa_cache(cpath, alloc)
            offs = np.cumsum([0] + list(rows))
            chains = [block[a:b] for a, b in zip(offs[:-1], offs[1:])]
            from_cache = True
        except (ValueError, OSError, KeyError):
            chains = None
    if chains is None:
        chains = []
        for fname in files:
            cols = loadNumpyTxt(fname)
            if cols.shape[0] == 0 or cols.shape[1] < 3:
                continue  # "Ignored file (likely empty)" (chains.py:1400-1403)
            chains.append(cols)
        if not chains:
            raise WeightedSampleError("loadChains - no chains found for " + file_root)
        if not no_cache:
            try:
                write_soa_cache(cpath, chains)
            except OSError:
                pass  # read-only chain directory: the cache is an optimisation only
    n = chains[0].shape[1] - 2
    labels = derived = None
    if os.path.isfile(file_root + ".paramnames"):
        names, labels, derived = readParamNames(file_root + ".paramnames")
        if len(names) != n:
            raise WeightedSampleError("paramnames file does not match the number of chain columns")
    else:
        names = ["param%d" % (i + 1) for i in range(n)]
    ranges = readRanges(file_root + ".ranges") if os.path.isfile(file_root + ".ranges") else {}
    return dict(samples=[c[:, 2:] for c in chains], weights=[c[:, 0] for c in chains], loglikes=[c[:, 1] for c in chains],
                names=names, labels=labels, derived=derived, ranges={k: v for k, v in ranges.items() if k in names},
                from_cache=from_cache)


def loadMCSamples(file_root, ini=None, jobItem=None, no_cache=False, settings=None, chain_exclude=None, **kwargs):
    """
    mcsamples.py:47-126: an MCSamples from the chain files of ``file_root`` (``ignore_rows`` burn-in per chain --
    a row count if >= 1, else a fraction --, per-chain minimum-weight filter, deletion of the parameters that do not
    move, names / labels / derived flags and hard bounds from the side files).  The first load writes the binary
    column cache next to the chains; later loads stream it (``no_cache`` or ``chain_exclude`` bypass it, as in the
    reference).
    """
    from .mcsamples import MCSamples

    if jobItem is not None:
        raise NotImplementedError("grid job items are outside the accelerated path")
    return MCSamples(root=file_root, ini=ini, settings=settings, _chain_exclude=chain_exclude, _no_cache=no_cache, **kwargs)
