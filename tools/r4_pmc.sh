# the three PMC passes of tools/profile_round.sh alone (FETCH_SIZE, WRITE_SIZE, SQ counters) + their summary -> gpurun_out/summary_pmc
set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out
rm -rf $O/prof_fetch $O/prof_write $O/prof_sq $O/prof_kt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_fetch -- python3 bench.py --steps 4 --warmup 1 --cpu-frames 0 --host-steps 0 --sustained-frames 0 > $O/prof_fetch.json 2> $O/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_write -- python3 bench.py --steps 4 --warmup 1 --cpu-frames 0 --host-steps 0 --sustained-frames 0 > $O/prof_write.json 2> $O/prof_write.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/prof_sq -- python3 bench.py --steps 4 --warmup 1 --cpu-frames 0 --host-steps 0 --sustained-frames 0 > $O/prof_sq.json 2> $O/prof_sq.err
find $O/prof_sq $O/prof_fetch $O/prof_write -name "*kernel_trace.csv" -delete
python3 tools/summarize_profiles.py $O $O/summary_pmc
cut -d, -f1-6 $O/summary_pmc/pmc_hbm_traffic.csv
cut -d, -f1-12 $O/summary_pmc/pmc_sq_lds.csv
