#!/bin/bash
# gpurun_out/r05/* (scratch) -> profiles/r05_* (tracked): run in the build container after scripts/r05_evidence_run.sh.
cd "$(dirname "$0")/.." || exit 1
S=gpurun_out/r05; P=profiles
cp $S/bench_default.json $P/r05_bench_default.json
cp $S/bench_40.json $P/r05_bench.json
cp $S/bench_kernel_stats.csv $P/r05_bench_kernel_stats.csv
cp $S/bench_stream_timeline.txt $P/r05_bench_stream_timeline.txt
cp $S/conv_kernel_stats.csv $P/r05_conv_bench_kernel_stats.csv
cp $S/pmc_conv.json $P/r05_pmc_conv.json
cp $S/pmc_c4.json $P/r05_pmc_c4.json
cp $S/c4_kernel_stats.csv $P/r05_c4_kernel_stats.csv
cp $S/config_c2.json $P/r05_config_c2.json
grep -v '^[WE]2026' $S/c4.log | tail -1 > $P/r05_config_c4.json
cp $S/kernels.json $P/r05_kernels.json
cp $S/emulate_c3.json $P/r05_emulate_world.json
cat $S/emulate_c4_w*.json > $P/r05_emulate_world_c4.jsonl
cp $S/emulate_c5.json $P/r05_emulate_world_c5.json
cp $S/emu8_stream_timeline.txt $P/r05_emu8_timeline.txt
cp $S/emu8_kernel_stats.csv $P/r05_emu8_kernel_stats.csv
cp $S/lds_atomic_f64_roof.txt $P/r05_lds_atomic_f64_roof.txt
grep -v "^[WE]2026" $S/rccl_smoke.log | grep -E "library communicator not available|nccl smoke ok|RCCL version|Librccl path" > $P/r05_rccl_smoke.txt
cp $S/bench_gloo2_shared_gpu.json $P/r05_bench_gloo2_shared_gpu.json
tail -5 $S/pytest.log > $P/r05_pytest_gpu_tail.txt
cp gpurun_out/r05_parity_2d.json $P/r05_parity_2d.json
ls -la $P | grep r05_
