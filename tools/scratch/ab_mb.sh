#!/bin/bash
for lib in build_variants/*.so; do echo "$lib $(RAMP_HIP_LIB=$PWD/$lib python tools/scratch/mb_update.py 2>&1 | tail -1)"; done
