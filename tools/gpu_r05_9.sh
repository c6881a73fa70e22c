T=./slam-sdvl_amd/host/track_sequence
A="--synthetic 300 --prerender --quiet --json --texture camera"
echo "b1 default"; $T $A | cut -c1-170
echo "b1 fused"; SDVL_IA_TRACK_FUSED=1 $T $A | cut -c1-170
echo "b1 fused one wave"; SDVL_IA_TRACK_FUSED=1 SDVL_IA_SMALL_WAVES=1 $T $A | cut -c1-170
SDVL_IA_TRACK_FUSED=1 python3 -m pytest tests/test_gpu_camera_texture.py tests/test_gpu_tracker.py -x -q -m gpu -k "closed_loop or farm or handleframe or chunks" 2>&1 | tail -3
echo "== image alignment of a tracked step: fused kernel, items in LDS (SDVL_IA_TRACK_FUSED=1) vs precompute launch + one-wave chain, camera texture"
AB_COMMON_ARGS="--texture camera" bash tools/ab_r05.sh "SDVL_IA_TRACK_FUSED=1" - 3 60
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/ia_fused; mkdir -p $O
SHORT="--texture camera --steps 4 --warmup 1 --cpu-frames 0 --host-steps 0 --sustained-frames 0 --latency-frames 0"
SDVL_IA_TRACK_FUSED=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_fetch -- python3 bench.py $SHORT > $O/prof_fetch.json 2> $O/prof_fetch.err
SDVL_IA_TRACK_FUSED=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_write -- python3 bench.py $SHORT > $O/prof_write.json 2> $O/prof_write.err
find $O -name "*kernel_trace.csv" -delete
python3 tools/summarize_profiles.py $O $O/summary > /dev/null 2>&1
grep "image_align" $O/summary/pmc_hbm_traffic.csv | cut -d, -f1-6
rm -rf $O/prof_fetch $O/prof_write
