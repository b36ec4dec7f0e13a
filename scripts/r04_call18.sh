#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
GDHIP_BATCH_LOG=1 python scripts/r04_c5_probe.py 1 > gpurun_out/r04/c5_probe.out 2> gpurun_out/r04/c5_probe.err
tail -1 gpurun_out/r04/c5_probe.out
python - <<'PY'
txt=open("gpurun_out/r04/c5_probe.err").read().split("---- gd_density2d_batch host timeline (ms)\n")
print(len(txt)-1,"calls")
for k in (1,2,6):
    print("==== call",k); print("\n".join(l for l in txt[k].splitlines() if "WARNING" not in l and "conv: enqueued" not in l)[:2200])
PY
