#!/bin/bash
# usage: tools/pmc_kernel.sh <kernel-substring> <out-file> -- <command...>
# Two separate rocprofv3 --pmc passes (SQ issue / wait breakdown; LDS + L2) averaged per dispatch of
# the kernels whose name contains the substring.
K="$1"; OUTF="$2"; shift 3
cd /tmp; export TMPDIR=/tmp
: > $OUTF
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum"; do
  rm -rf /tmp/pmc_k
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_k -o p -- "$@" > /tmp/pmc_k.log 2>&1 || true
  python3 - "$K" <<'PY' >> $OUTF
import csv, collections, sys, glob
f = glob.glob("/tmp/pmc_k/*counter_collection.csv")
if not f: print("no counter file:", open("/tmp/pmc_k.log").read()[-800:]); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(f[0])):
    if sys.argv[1] in r["Kernel_Name"]:
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in agg:
    print(k, {c: round(v / n[k][c]) for c, v in agg[k].items()}, "dispatches", max(n[k].values()))
PY
done
cat $OUTF
