import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from getdist_amd.mcsamples import MCSamples
from oracle import kde_oracle as ko
from oracle.fixtures import example_mask_function, fixture_zoo
zoo = {fx["name"]: fx for fx in fixture_zoo()}
fx = zoo["shapes"]
mc = MCSamples(samples=fx["samples"], weights=fx["weights"], names=fx["names"], ranges=fx["ranges"])
orc = ko.OracleSamples(fx["samples"], fx["weights"], names=fx["names"], ranges=fx["ranges"])
a, b = 0, 1
tr = {}
o = orc.density_2d(a, b, mask_function=example_mask_function, trace=tr)
d = mc.get2DDensities([(a, b)], mask_function=example_mask_function)[0]
print("device bw", [repr(float(v)) for v in d.bandwidth], "oracle", tr["hx"], tr["hy"], tr["c"])
o2 = orc.density_2d(a, b, mask_function=example_mask_function, _bandwidths=tuple(d.bandwidth))
err = np.abs(d.P - o2["P"])
print("vs oracle@device-bw: max %.3e  n>1e-6 %d  n>1e-3 %d  median %.2e  argmax %s" % (err.max(), (err > 1e-6).sum(), (err > 1e-3).sum(), np.median(err), np.unravel_index(err.argmax(), err.shape)))
for eps in (1e-13, -1e-13, 1e-11):
    bw = tuple(float(v) * (1 + eps) for v in d.bandwidth)
    o3 = orc.density_2d(a, b, mask_function=example_mask_function, _bandwidths=bw)
    e3 = np.abs(o3["P"] - o2["P"])
    print("oracle self-sensitivity eps=%g: max %.3e n>1e-6 %d argmax %s" % (eps, e3.max(), (e3 > 1e-6).sum(), np.unravel_index(e3.argmax(), e3.shape)))
