#!/bin/bash
# round 6: where the victim search's time goes on the reference's ReclaimLargeJobs shapes (the round-5 verdict's item 1; no change to the engine this round — evidence for DESIGN.md section 10):
# (1) the control lane's phase clocks from a -DKAI_PROF_VICTIM build of the library (tools/micro/libkai_core_profv.so, built by hand: see DESIGN.md), (2) a PMC pass of k_action<victim>
TAG=${1:-r06v}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
for shape in ReclaimLargeJobs_200 ReclaimLargeJobs_500 ConsolidationAction_Medium; do
  if [ -f tools/micro/libkai_core_profv.so ]; then
    KAI_CORE_LIB="$R/tools/micro/libkai_core_profv.so" KAI_PROF=1 timeout 120 python tools/prof_reclaim.py $shape > gpurun_out/${TAG}_phase_clocks_$shape.txt 2>&1; echo "phase clocks $shape rc=$?"; tail -4 gpurun_out/${TAG}_phase_clocks_$shape.txt | cut -c1-400
  fi
done
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-48)
  cd /tmp
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_${TAG}_$tag" -- python "$R/tools/prof_reclaim.py" ReclaimLargeJobs_200 > "$R/gpurun_out/pmc_${TAG}_$tag.log" 2>&1; echo "$set rc=$?"
  cd "$R"
  f=$(find gpurun_out/pmc_${TAG}_$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" "$set" <<'PY' >> gpurun_out/${TAG}_victim_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:60]
    if "k_action" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k, " ".join(f"{c} {v:.0f} (x{n[(k, c)]} launches)" for c, v in sorted(d.items())))
PY
  rm -rf gpurun_out/pmc_${TAG}_$tag
done
echo "# rocprofv3 --pmc, separate passes, python tools/prof_reclaim.py ReclaimLargeJobs_200 (2 iterations: open + reclaim): counters of k_action<victim> summed over its launches" >> gpurun_out/${TAG}_victim_pmc.txt
cat gpurun_out/${TAG}_victim_pmc.txt | cut -c1-400
