cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/kb_pmc
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/kb_pmc -- python3 tools/kernel_bench.py 256 3 > gpurun_out/kb_pmc.txt 2>&1
find gpurun_out/kb_pmc -name "*kernel_trace.csv" -delete
