#!/bin/bash
# usage: scripts/pmc.sh <tag> <one_kernel args...>   -- PMC passes for one kernel config
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp
pass() { n=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$n -o p -- python $GRAFT_REPO_ROOT/scripts/one_kernel.py $ARGS > $OUT/pass$n.log 2>&1; f=$(find /tmp/pmc_${TAG}_$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python - "$f" <<'PY' | tee -a $OUT/summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "attn" in k or "gemm" in k:
        print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
}
ARGS="$@"
pass 1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
pass 2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC
pass 3 GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL
tail -3 $OUT/pass1.log
