#!/bin/bash
# scan-grid hand-off clocks (library built with -DKAI_PROF_VICTIM): publish / own slice / wait for the fold / commands = prof slots 36..39
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for args in "--config C5 --mixed" "--config C3 --fractions 0.3"; do
  echo "== $args"
  KAI_CORE_LIB=build/libkai_core_vprof.so KAI_PROF=1 timeout 600 python bench.py $args --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -E "kai prof" | awk '{print "total="$10" publish="$39" own="$40" wait="$41" cmds="$42}'
done
