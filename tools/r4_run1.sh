set -u
O=gpurun_out
hipcc --offload-arch=gfx950 -O2 -o /tmp/icache_probe tools/icache_probe.cpp 2>&1 | tail -2
timeout -k 5 120 /tmp/icache_probe > $O/icache_probe.txt 2>&1; echo "rc=$?"; cat $O/icache_probe.txt
