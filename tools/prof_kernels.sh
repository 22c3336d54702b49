#!/bin/bash
# tools/prof_kernels.sh PATTERN -- CMD...: run CMD under rocprofv3 --kernel-trace --stats and print the per-kernel rows matching PATTERN
pat=$1; shift; shift
export TMPDIR=/tmp
d=/tmp/prof_$$
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- "$@" 2>&1 | grep -v "^\[rocprofv3\]\|^W2\|^E2\|^I2" | tail -5)
f=$(find $d -name '*kernel_stats.csv' | head -1)
python - "$f" "$pat" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("%-60s calls %6s  avg %9.1f us  total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
