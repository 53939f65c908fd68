"""Copy the evidence of one gpurun call from gpurun_out/<tag>/ into profiles/ (tracked):
trimmed rocprofv3 kernel stats, the bench JSON line, parity log."""
import csv, json, os, shutil, sys
tag, name = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
for ks, dst in ((os.path.join(src, "bench_kernel_stats.csv"), f"profiles/{name}_rocprofv3_kernel_stats.csv"),
                (os.path.join(src, "train_kernel_stats.csv"), f"profiles/{name}_train_rocprofv3_kernel_stats.csv"),
                (os.path.join(src, "unet_kernel_stats.csv"), f"profiles/{name}_unet_rocprofv3_kernel_stats.csv")):
    if not os.path.exists(ks):
        continue
    rows = list(csv.DictReader(open(ks)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:40]:
            n = r["Name"]
            n = n if len(n) <= 110 else n[:107] + "..."
            w.writerow([n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
for fn, out in (("bench.log", f"{name}_bench.json"), ("bench_train.log", f"{name}_bench_train.json"), ("bench_unet.log", f"{name}_bench_unet.json"),
                ("bench_adapter_cache.log", f"{name}_bench_adapter_cache.json"), ("gpu_parity.log", f"{name}_gpu_parity.log"),
                ("pytest.log", None), ("smoke.log", f"{name}_smoke.log")):
    p = os.path.join(src, fn)
    if not os.path.exists(p):
        continue
    if fn.startswith("bench"):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines:
            open(f"profiles/{out}", "w").write(json.dumps(json.loads(lines[-1]), indent=1) + "\n")
    elif fn == "pytest.log":
        tail = [l for l in open(p) if "passed" in l or "failed" in l or "error" in l.lower()]
        open(f"profiles/{name}_pytest_summary.txt", "w").write("".join(tail[-5:]))
    else:
        shutil.copy(p, f"profiles/{out}")
print("saved", name)
