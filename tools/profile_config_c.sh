set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out
rm -rf $O/sc_kt $O/sc_sq
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sc_kt -- python3 bench.py --workload S-C --seqs 512 --steps 10 --warmup 3 --cpu-frames 0 --host-steps 0 > $O/sc_kt.json 2> $O/sc_kt.err
find $O/sc_kt -name "*kernel_trace.csv" -delete
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/sc_sq -- python3 bench.py --workload S-C --seqs 512 --steps 4 --warmup 1 --cpu-frames 0 --host-steps 0 > $O/sc_sq.json 2> $O/sc_sq.err
find $O/sc_sq -name "*kernel_trace.csv" -delete
ls -R $O/sc_kt | head; ls -R $O/sc_sq | head
