"""Copy the evidence of one gpurun call from gpurun_out/<tag>/ into profiles/ (tracked):
trimmed rocprofv3 kernel stats, the bench JSON line, parity log."""
import csv, json, os, shutil, sys
tag, name = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
ks = os.path.join(src, "bench_kernel_stats.csv")
if os.path.exists(ks):
    rows = list(csv.DictReader(open(ks)))
    with open(f"profiles/{name}_rocprofv3_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:40]:
            n = r["Name"]
            n = n if len(n) <= 110 else n[:107] + "..."
            w.writerow([n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
for fn, out in (("bench.log", f"{name}_bench.json"), ("gpu_parity.log", f"{name}_gpu_parity.log"),
                ("pytest.log", None), ("smoke.log", f"{name}_smoke.log")):
    p = os.path.join(src, fn)
    if not os.path.exists(p):
        continue
    if fn == "bench.log":
        lines = [l for l in open(p) if l.startswith("{")]
        if lines:
            open(f"profiles/{out}", "w").write(json.dumps(json.loads(lines[-1]), indent=1) + "\n")
    elif fn == "pytest.log":
        tail = [l for l in open(p) if "passed" in l or "failed" in l or "error" in l.lower()]
        open(f"profiles/{name}_pytest_summary.txt", "w").write("".join(tail[-5:]))
    else:
        shutil.copy(p, f"profiles/{out}")
print("saved", name)
