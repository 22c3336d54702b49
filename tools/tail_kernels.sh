#!/bin/bash
# tools/tail_kernels.sh "EXTRA flags" tag: rebuild with the flags, kernel-trace a sequential 160-frame run, print the mean
# duration of the tail's small launches (steady state) and the frame period
mkdir -p gpurun_out/tk
( cd rampvo_amd/csrc && make -s clean && make -s -j 12 EXTRA="$1" libramp_hip.so 2>&1 | grep -i " error" )
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tk_$2
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tk_$2 -- python $GRAFT_REPO_ROOT/tools/frame_timeline.py run sequential > /dev/null 2>&1
python - $(ls /tmp/tk_$2/*/*kernel_trace.csv | head -1) "$2" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
keys = [k for k in d if any(t in k for t in ("trk_", "plan_", "motionmag", "frame_commit", "transform_kernel", "ba_"))]
tot = 0
for k in sorted(keys):
    v = d[k]; print("%-12s %-52s n=%4d mean %6.2f us" % (sys.argv[2], k[:52], len(v), sum(v) / len(v)))
P
cd $GRAFT_REPO_ROOT
