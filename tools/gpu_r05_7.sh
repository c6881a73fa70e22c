bash tools/profile_r05.sh sa_plane > gpurun_out/p_sa_plane.log 2>&1; tail -2 gpurun_out/p_sa_plane.log
bash tools/profile_r05.sh sa_camera --texture camera > gpurun_out/p_sa_camera.log 2>&1; tail -2 gpurun_out/p_sa_camera.log
bash tools/profile_r05.sh sb_camera --workload S-B --texture camera > gpurun_out/p_sb_camera.log 2>&1; tail -2 gpurun_out/p_sb_camera.log
python3 tools/fast_density_probe.py 256 > gpurun_out/fast_density_probe.txt 2>&1
SDVL_KB_TEXTURE=camera python3 tools/kernel_bench.py 256 6 > gpurun_out/kernel_bench_isolated_camera.txt 2>&1
python3 tools/kernel_bench.py 256 6 > gpurun_out/kernel_bench_isolated_plane.txt 2>&1

