#!/bin/bash
# HBM traffic of the bench's kernels from the TCC counters (separate --pmc passes, MI355X_MICROARCH.md §HBM):
#   bytes_read  = FETCH_SIZE [KiB] * 1024 * 2   (gfx950: FETCH_SIZE reports half the bytes of wide streaming reads)
#   bytes_write = WRITE_SIZE [KiB] * 1024
# usage: scripts/pmc_traffic.sh <tag>   -> gpurun_out/<tag>/pmc_traffic.json
TAG=${1:-r1}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp
# (round 6: a fourth pass with the L2's memory-side request counters split by destination - TCC_EA0_RDREQ / _WRREQ count every request that
#  leaves the L2 towards the fabric, the _DRAM forms only those that go to HBM, i.e. NOT the ones the Infinity Cache (MALL) serves:
#  dram_fraction = *_DRAM / total tells how much of FETCH_SIZE / WRITE_SIZE is real HBM traffic)
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum"; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmct_${TAG}_${C%% *} -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-text-only-leg > $OUT/pmc_${C%% *}.log 2>&1
done
python - "$OUT" /tmp/pmct_${TAG}_FETCH_SIZE /tmp/pmct_${TAG}_WRITE_SIZE /tmp/pmct_${TAG}_SQ_VALU_MFMA_BUSY_CYCLES /tmp/pmct_${TAG}_TCC_EA0_RDREQ_sum <<'PY'
import csv, glob, json, sys, collections
out, dirs = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = next((n for n in ("gemm4w_kernel", "gemm_bf16_kernel", "attn_stream_kernel", "attn_res_kernel", "attn_fwd_kernel",
                                  "attn_group_lds_kernel", "attn_small_kernel", "layernorm_kernel") if n in k), None)
            if k is None: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "FETCH_SIZE": calls[k] += 1
res = {}
for k, d in agg.items():
    n = max(calls[k], 1)
    rd, wr = d.get("FETCH_SIZE", 0) * 1024 * 2, d.get("WRITE_SIZE", 0) * 1024
    res[k] = {"launches": n, "hbm_read_bytes_per_launch": rd / n, "hbm_write_bytes_per_launch": wr / n,
              "hbm_bytes_per_launch": (rd + wr) / n, "note": "L2 -> fabric bytes (MALL hits included): FETCH_SIZE KiB*1024*2 (gfx950 wide-load correction) + WRITE_SIZE KiB*1024; 2 bench steps (1 warmup + 1)"}
    if d.get("TCC_EA0_RDREQ_sum"):
        fr = d.get("TCC_EA0_RDREQ_DRAM_sum", 0.0) / d["TCC_EA0_RDREQ_sum"]
        fw = d.get("TCC_EA0_WRREQ_DRAM_sum", 0.0) / d["TCC_EA0_WRREQ_sum"] if d.get("TCC_EA0_WRREQ_sum") else None
        res[k]["dram"] = {"read_request_fraction_to_dram": fr, "write_request_fraction_to_dram": fw,
                          "dram_read_bytes_per_launch_est": fr * rd / n, "dram_write_bytes_per_launch_est": None if fw is None else fw * wr / n,
                          "dram_bytes_per_launch_est": (fr * rd + (fw if fw is not None else 1.0) * wr) / n,
                          "note": "TCC_EA0_RDREQ_DRAM / TCC_EA0_RDREQ (and the WRREQ pair) of a separate pass: the share of the L2's fabric requests that HBM, not the Infinity Cache, serves; bytes = share x the FETCH / WRITE figures"}
# MFMA utilisation (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts pipe cycles summed over the 1024 SIMDs,
# GRBM_GUI_ACTIVE is summed over the 8 XCDs): util = busy / (1024 * gui_active / 8)
for k, d in agg.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("GRBM_GUI_ACTIVE"):
        n = max(calls[k], 1)
        res[k]["mfma"] = {"insts_mfma_per_launch": d.get("SQ_INSTS_MFMA", 0) / n,
                          "mfma_busy_cycles_per_launch": d["SQ_VALU_MFMA_BUSY_CYCLES"] / n,
                          "gui_active_cycles_per_xcd_per_launch": d["GRBM_GUI_ACTIVE"] / 8 / n,
                          "mfma_pipe_utilisation": d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * d["GRBM_GUI_ACTIVE"] / 8),
                          "note": "one rocprofv3 --pmc pass of its own; busy cycles / (1024 SIMDs x kernel cycles)"}
json.dump(res, open(out + "/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
