bash tools/tail_kernels.sh "" base 2>&1 | grep -v "^$"
bash tools/tail_kernels.sh "-DTRK_EB=256" eb256 2>&1 | grep "trk_"
bash tools/tail_kernels.sh "-DPLAN_EPB=256" epb256 2>&1 | grep "plan_"
bash tools/tail_kernels.sh "-DPLAN_EPB=512 -DTRK_EB=512" b512 2>&1 | grep "plan_\|trk_"
