{
echo "== hardware queues: GPU_MAX_HW_QUEUES=16 (16 streams on 16 queues) vs the runtime's default 4"
bash tools/ab_r05.sh "GPU_MAX_HW_QUEUES=16" - 3 60
echo "== group shape: 32 groups x 256 sequences (8192 sequences) vs 16 x 256"
bash tools/ab_r05.sh - "--seqs 8192 --groups 32" 3 60
echo "== group shape: 32 groups x 128 sequences vs 16 x 256"
bash tools/ab_r05.sh - "--groups 32" 3 60
echo "== FAST path choice: rounds 2-4's threshold (32 of 64 probed pixels) vs 16, camera texture"
AB_COMMON_ARGS="--texture camera" bash tools/ab_r05.sh "SDVL_FAST_DENSE_NUM=32" - 3 60
} > gpurun_out/ab_round5.txt 2>&1
cat gpurun_out/ab_round5.txt
