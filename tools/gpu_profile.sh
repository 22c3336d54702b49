#!/bin/bash
# tools/gpu_profile.sh TAG [bench args...]: bench.py under rocprofv3 --kernel-trace --stats on the GPU box; writes
# gpurun_out/TAG_kernel_stats.md (+ the csv) and gpurun_out/TAG_bench.json (the un-profiled line of the same command)
tag=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
python bench.py "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
d=/tmp/prof_$tag
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/bench.py "$@" --cpu-steps 0 --parity 0 --fp32-leg 0 > /tmp/prof_$tag.out 2>&1)
f=$(find $d -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv
python tools/summarize_rocprof.py "$f" gpurun_out/${tag}_kernel_stats.md "bench.py $* (rocprofv3 --kernel-trace --stats)"
tail -c 600 gpurun_out/${tag}_bench.json
