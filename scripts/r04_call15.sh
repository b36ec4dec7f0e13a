#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; mkdir -p gpurun_out/r04
GDHIP_BATCH_LOG=1 GETDIST_AMD_HOSTLOG=1 timeout 300 python scripts/host_timeline.py > gpurun_out/r04/ht1.out 2> gpurun_out/r04/ht1.err
grep " ms " gpurun_out/r04/ht1.out | tail -12
python - <<'PY'
txt=open("gpurun_out/r04/ht1.err").read().split("---- gd_density2d_batch host timeline (ms)\n")
print(len(txt)-1,"calls"); t=txt[-1]
print("\n".join(l for l in t.splitlines() if "conv: enqueued" not in l and "WARNING" not in l)[:4000])
PY
