#!/bin/bash
# Run on the GPU box from the repo root:  bash tools/profile_bench.sh TAG [bench.py arguments, e.g. --texture camera | --workload S-B --texture camera]
# One workload / texture per call: the full bench line, then rocprofv3 passes of a short run of the same command (the program itself
# follows `--`; counters in passes of their own, never with a trace domain beyond the kernel trace), condensed by
# tools/summarize_profiles.py into gpurun_out/TAG/summary (copy what should be judged into profiles/rNN/TAG/).
#   SKIP_FULL=1: no full bench line (only the profiler passes);  SKIP_PMC=1: kernel trace only
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/$TAG
mkdir -p $O
rm -rf $O/prof_kt $O/prof_fetch $O/prof_write $O/prof_sq $O/summary
# every profiled run: no child processes (the latency legs exec track_sequence: under rocprofv3 the GPU is already initialised), no extra farms
SHORT="--cpu-frames 0 --host-steps 0 --sustained-frames 0 --latency-frames 0 --lost-mix-steps 0"
if [ -z "${SKIP_FULL:-}" ]; then
  python3 bench.py "$@" > $O/bench_default.json 2> $O/bench_default.err || { echo "bench.py failed"; tail -5 $O/bench_default.err; exit 1; }
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -- python3 bench.py "$@" --steps 20 --warmup 3 $SHORT > $O/prof_kt.json 2> $O/prof_kt.err
find $O/prof_kt -name "*kernel_trace.csv" -delete
if [ -z "${SKIP_PMC:-}" ]; then
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_fetch -- python3 bench.py "$@" --steps 4 --warmup 1 $SHORT > $O/prof_fetch.json 2> $O/prof_fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_write -- python3 bench.py "$@" --steps 4 --warmup 1 $SHORT > $O/prof_write.json 2> $O/prof_write.err
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/prof_sq -- python3 bench.py "$@" --steps 4 --warmup 1 $SHORT > $O/prof_sq.json 2> $O/prof_sq.err
  find $O/prof_fetch $O/prof_write $O/prof_sq -name "*kernel_trace.csv" -delete
fi
python3 tools/summarize_profiles.py $O $O/summary
rm -rf $O/prof_kt $O/prof_fetch $O/prof_write $O/prof_sq   # the raw counter files are tens of MB; gpurun_out/ travels back only below 64 MiB
ls -la $O/summary
