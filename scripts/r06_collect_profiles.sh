#!/bin/bash
# copies the judged summaries of gpurun_out/r06/ (scripts/r06_evidence_run.sh) into profiles/ under their round-6 names
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06
cp $O/bench_default.json profiles/r06_bench_default.json
cp $O/bench_40.json profiles/r06_bench.json
cp $O/bench_kernel_stats.csv profiles/r06_bench_kernel_stats.csv
cp $O/bench_stream_timeline.txt profiles/r06_bench_stream_timeline.txt
cp $O/latency_timeline.txt profiles/r06_latency_timeline.txt
cp $O/kernels.json profiles/r06_kernels.json 2>/dev/null || cp gpurun_out/kernels_isolated.json profiles/r06_kernels.json
cp $O/pmc_prep.json profiles/r06_pmc_prep.json
cp $O/cov_ab.json profiles/r06_cov_one_pass_ab.json
cp $O/conv_kernel_stats.csv profiles/r06_conv_bench_kernel_stats.csv
cp $O/config_c2.json profiles/r06_config_c2.json
cp $O/config_c5.json profiles/r06_config_c5.json
cp $O/c4_kernel_stats.csv profiles/r06_c4_kernel_stats.csv
grep '^{' $O/c4.log | tail -1 > profiles/r06_config_c4.json
for W in 2 4 8; do tail -1 $O/emulate_w$W.json; done > profiles/r06_emulate_world.jsonl
cat $O/emulate_c4_w*.json 2>/dev/null | grep '^{' > profiles/r06_emulate_world_c4.jsonl
tail -3 $O/pytest.log > profiles/r06_pytest_gpu_tail.txt
grep -E 'nccl smoke ok|RCCL version' $O/rccl_smoke.log > profiles/r06_rccl_smoke.txt
cp $O/bench_gloo2_shared_gpu.json profiles/r06_bench_gloo2_shared_gpu.json 2>/dev/null
cp gpurun_out/r06_parity_2d.json profiles/r06_parity_2d.json 2>/dev/null
ls -la profiles/r06_*
