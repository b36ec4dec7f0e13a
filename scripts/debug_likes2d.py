"""Debug aid: compare the pieces of the 2D mean-likelihood path on the GPU with numpy (tests/fake_ctx.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from fake_ctx import FakeBuf, FakeContext
    from getdist_amd.mcsamples import MCSamples
    from oracle.fixtures import fixture_zoo, loglikes_for

    zoo = {f["name"]: f for f in fixture_zoo()}
    fx = zoo["c1_bounded"]
    ll = loglikes_for(fx["samples"])
    mc = MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"], loglikes=ll)
    ctx = mc.ctx
    mc._init_params([0, 3])
    F = 256
    names = mc.paramNames.names
    fwx, xmin, _ = mc._bin_edges(names[0], F)
    fwy, ymin, _ = mc._bin_edges(names[3], F)
    ix = mc._index_column(0, F, xmin, fwx)
    iy = mc._index_column(3, F, ymin, fwy)
    d_h = ctx.hist2d_prebinned([ix], [iy], F)
    d_lh = mc._like_histograms(0, lambda: ctx.hist2d_prebinned([ix], [iy], F))
    H = d_h.to_host((F, F))
    LH = d_lh.to_host((F, F))
    s = np.asarray(fx["samples"])
    bx = ((s[:, 0] - xmin) / fwx + 0.5).astype(int)
    by = ((s[:, 3] - ymin) / fwy + 0.5).astype(int)
    w = np.ones(len(s)) if fx["weights"] is None else fx["weights"]
    ml = w.dot(ll) / w.sum()
    lw = w * np.exp(ml - ll)
    H0 = np.bincount(bx + by * F, weights=w, minlength=F * F).reshape(F, F)
    LH0 = np.bincount(bx + by * F, weights=lw, minlength=F * F).reshape(F, F)
    print("hist err", np.abs(H - H0).max(), "likehist rel err", np.abs(LH - LH0).max() / LH0.max(), "mean_loglike", mc.mean_loglike - ml)
    fake = FakeContext(0)
    for mbc in (0, 1):
        for args in ((12.3, 9.1, 0.0, 31, 64 | 4), (12.3, 9.1, 0.4, 31, 0)):
            rx, ry, c, winw, fl = args
            d_L, st = ctx.likes2d(d_h, d_lh, 1, F, [rx], [ry], [c], [winw], [fl], mbc)
            L = d_L.to_host((F, F))
            L0, _ = fake.likes2d(FakeBuf(H0[None]), FakeBuf(LH0[None]), 1, F, [rx], [ry], [c], [winw], [fl], mbc)
            e = np.abs(L - L0.a[0])
            print("mbc", mbc, args, "likes err", e.max(), "at", np.unravel_index(np.argmax(e), e.shape), "status", st)


if __name__ == "__main__":
    main()
