cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/ic_bench gpurun_out/ic_alone
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/ic_bench -- python3 bench.py --steps 4 --warmup 1 --cpu-frames 0 --host-steps 0 > gpurun_out/ic_bench.json 2> gpurun_out/ic_bench.err
find gpurun_out/ic_bench -name "*kernel_trace.csv" -delete
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/ic_alone -- python3 tools/kernel_bench.py 256 3 > gpurun_out/ic_alone.txt 2>&1
find gpurun_out/ic_alone -name "*kernel_trace.csv" -delete
ls gpurun_out/ic_bench/* gpurun_out/ic_alone/*
