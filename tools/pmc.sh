#!/bin/bash
# usage: tools/pmc.sh "<counters>" <kernel-substring> -- <command...>   (one PMC pass, averaged per dispatch)
set -e
C="$1"; K="$2"; shift 3
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pmc_out
rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_out -o p -- "$@" > /tmp/pmc_out.log 2>&1 || true
python3 - "$K" <<'PY'
import csv, collections, sys, glob
f = glob.glob("/tmp/pmc_out/*counter_collection.csv")
if not f: print("no counter file; log tail:"); print(open("/tmp/pmc_out.log").read()[-1500:]); sys.exit(0)
rows = [r for r in csv.DictReader(open(f[0])) if sys.argv[1] in r["Kernel_Name"]]
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for r in rows: agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print({k: round(v / n[k]) for k, v in agg.items()}, "dispatches", max(n.values()) if n else 0)
PY
