# (historical: the -D switches it sets existed during round 6's experiment only; kept as the record of how profiles/r06/fast_cells_lds_experiment/ was produced)
# bash tools/fc_lds_ab.sh: fast_cells with different LDS row pitches / read widths — time alone (kernel_bench) and the SQ LDS counters
# (are the bank conflicts hidden behind VALU issue?).  Rebuilds csrc/libsdvl_hip.so per variant ON THE BOX; restores the default at the end.
set -u
F="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function"
for v in "20 0" "20 0 -DSDVL_FC_LDS_TWICE"; do
  set -- $v
  touch slam-sdvl_amd/csrc/sdvl_detect.hip
  make -s -C slam-sdvl_amd/csrc HIPFLAGS="$F -DSDVL_FC_PITCH=$1 -DSDVL_FC_ALL64=$2 ${3:-}" > /dev/null 2>&1 || { echo "build failed for $v"; continue; }
  t=$(for i in 1 2 3; do python3 tools/kernel_bench.py 256 10 2>/dev/null | grep "fast_cells " | awk '{print $2}'; done | tr '\n' ' ')
  c=$(bash tools/pmc_kernel_bench.sh fast_cells_wave 2>/dev/null | cut -c1-330)
  echo "pitch $1 all64 $2 ${3:-} : alone us/launch $t | $c"
done
touch slam-sdvl_amd/csrc/sdvl_detect.hip
make -s -C slam-sdvl_amd/csrc > /dev/null 2>&1
rm -rf gpurun_out/kb_pmc
