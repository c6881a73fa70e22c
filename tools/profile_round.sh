#!/bin/bash
# Run on the GPU box from the repo root:  bash tools/profile_round.sh
# Three separate rocprofv3 runs of bench.py (the program itself follows `--`): kernel trace + stats on the default
# workload, then one PMC pass per counter (FETCH_SIZE, WRITE_SIZE) on a shorter run.  Raw output -> gpurun_out/,
# (kernel trace at 20 steps: with 30+ steps, 16 groups and the device pose stage this rocprofv3 build segfaults inside
# hipMemcpyAsync; the same command runs clean without the profiler and with --pmc)
# summaries are made from it by tools/summarize_profiles.py and committed under profiles/.
set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out
rm -rf $O/prof_kt $O/prof_fetch $O/prof_write $O/prof_sq
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -- python3 bench.py --steps 20 --warmup 3 --cpu-frames 0 --host-steps 0 --sustained-frames 0 > $O/prof_kt.json 2> $O/prof_kt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_fetch -- python3 bench.py --steps 4 --warmup 1 --cpu-frames 0 --host-steps 0 --sustained-frames 0 > $O/prof_fetch.json 2> $O/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_write -- python3 bench.py --steps 4 --warmup 1 --cpu-frames 0 --host-steps 0 --sustained-frames 0 > $O/prof_write.json 2> $O/prof_write.err
# SQ counters in a pass of their own (8 SQ slots; MI355X_MICROARCH.md "rocprofv3 PMC slots"): LDS-array cycles and the extra cycles
# bank conflicts cost, LDS instructions, waves launched, VALU instructions issued, cycles waves spent waiting / busy
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/prof_sq -- python3 bench.py --steps 4 --warmup 1 --cpu-frames 0 --host-steps 0 --sustained-frames 0 > $O/prof_sq.json 2> $O/prof_sq.err
find $O/prof_sq -name "*kernel_trace.csv" -delete
# the stats files are small; the raw traces are not needed back
find $O/prof_kt -name "*kernel_trace.csv" -delete
find $O/prof_fetch $O/prof_write -name "*kernel_trace.csv" -delete
# the other two configurations DESIGN §7 quotes: the reference's mapper inside every step, and BASELINE's configuration C
python3 bench.py --mapper --steps 20 --warmup 4 --cpu-frames 0 --host-steps 0 > $O/bench_mapper.json 2> $O/bench_mapper.err
python3 bench.py --workload S-C --seqs 512 --steps 30 --warmup 4 --cpu-frames 60 --host-steps 0 --sustained-frames 0 > $O/bench_config_c.json 2> $O/bench_config_c.err
python3 bench.py --workload S-C --steps 30 --warmup 4 --cpu-frames 0 --host-steps 0 --sustained-frames 0 > $O/bench_config_c_64seq.json 2> $O/bench_config_c_64seq.err
# kernel stats of configuration C (the workload whose dominant kernel VERDICT r01 asked to halve)
rm -rf $O/sc_kt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sc_kt -- python3 bench.py --workload S-C --seqs 512 --steps 10 --warmup 3 --cpu-frames 0 --host-steps 0 --sustained-frames 0 > $O/sc_kt.json 2> $O/sc_kt.err
find $O/sc_kt -name "*kernel_trace.csv" -delete
# the line earlier rounds quoted (2048 sequences, 100 steps), for comparison across rounds
python3 bench.py --seqs 2048 --steps 100 --warmup 5 --cpu-frames 0 --host-steps 0 --sustained-frames 0 > $O/bench_2048x100.json 2> $O/bench_2048x100.err
# (A/B lines: tools/ab_r05.sh "<ENV=VAL>" "<extra args>" PAIRS STEPS, a call of its own; round 5 per-workload profiles: tools/profile_r05.sh)
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_peak_probe tools/valu_peak_probe.cpp 2>/dev/null && timeout -k 5 120 /tmp/valu_peak_probe > $O/valu_peak_probe.txt 2>&1
python3 tools/kernel_bench.py 256 6 > $O/kernel_bench_isolated.txt 2>&1
python3 tools/pcie_probe.py > $O/pcie_probe.txt 2>&1
python3 tools/summarize_profiles.py $O $O/summary_r
cp $O/bench_mapper.json $O/bench_config_c.json $O/bench_config_c_64seq.json $O/bench_2048x100.json $O/valu_peak_probe.txt $O/kernel_bench_isolated.txt $O/pcie_probe.txt $O/summary_r/
ks=$(ls -S $O/sc_kt/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$ks" ] && cp "$ks" $O/summary_r/kernel_stats_config_c.csv
ls -la $O/summary_r
