set -x
python -m pytest tests/test_gpu_camera_texture.py tests/test_gpu_tracker.py -x -q -m gpu > gpurun_out/t1.log 2>&1; echo "pytest rc $?" ; tail -15 gpurun_out/t1.log
for t in plane camera; do
  ./slam-sdvl_amd/host/track_sequence --synthetic 300 --texture $t --prerender --quiet --json > gpurun_out/lat_${t}_b1.json 2> gpurun_out/lat_${t}_b1.err
  SDVL_HANDLEFRAME_ONE_SHOT=1 ./slam-sdvl_amd/host/track_sequence --synthetic 300 --texture $t --prerender --quiet --json > gpurun_out/lat_${t}_b1_oneshot.json 2> gpurun_out/lat_${t}_b1_oneshot.err
  ./slam-sdvl_amd/host/track_sequence --synthetic 300 --texture $t --prerender --quiet --json --trackers 16 > gpurun_out/lat_${t}_b16.json 2> gpurun_out/lat_${t}_b16.err
done
./slam-sdvl_amd/host/track_sequence --synthetic 300 --texture camera --prerender --quiet --profile > /dev/null 2> gpurun_out/lat_camera_profile.err
cat gpurun_out/lat_*.json; cat gpurun_out/lat_camera_profile.err
python tools/fast_density_probe.py 256 > gpurun_out/fast_density_probe.txt 2>&1; cat gpurun_out/fast_density_probe.txt
SDVL_KB_TEXTURE=camera python tools/kernel_bench.py 256 6 > gpurun_out/kernel_bench_camera.txt 2>&1; cat gpurun_out/kernel_bench_camera.txt
