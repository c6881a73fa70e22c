import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print("value %.1f fps  ms/step %.3f  B=%s threads=%s" % (d["value"], d["ms_per_step"], d["config"]["sequences_per_gpu"], (d["config"]["groups_per_gpu"], d["config"].get("host_worker_threads"))))
    print("  host stages:", d.get("host_stage_ms_per_group_step"))
    print("  kernels    :", d.get("kernel_ms_per_step"))
    print("  roofline   :", d.get("roofline"))
    print("  cpu        :", d.get("cpu_baseline"), d.get("speedup_vs_cpu_1core"))
