bash tools/profile_r05.sh sa_plane > gpurun_out/p_sa_plane.log 2>&1; tail -1 gpurun_out/p_sa_plane.log
bash tools/profile_r05.sh sa_camera --texture camera > gpurun_out/p_sa_camera.log 2>&1; tail -1 gpurun_out/p_sa_camera.log
bash tools/profile_r05.sh sb_camera --workload S-B --texture camera > gpurun_out/p_sb_camera.log 2>&1; tail -1 gpurun_out/p_sb_camera.log
