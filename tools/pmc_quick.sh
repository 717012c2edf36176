#!/bin/bash
# development: one SQ counter pass over a few first frames of scene7 4K, default and generation pipeline (GPU box)
cd /tmp && export TMPDIR=/tmp
for gen in 0 1; do
  export RTX_GENERATIONS=$gen
  D=$GRAFT_REPO_ROOT/gpurun_out/pmcq_$gen
  rm -rf $D; mkdir -p $D
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/prof_frame.py scene7 3840 2160 6 5 > /dev/null 2>&1 || exit 1
  python3 - $D $gen <<'PY'
import csv, glob, sys, collections
per = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_frame" in r["Kernel_Name"] and "clear" not in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
v = {k: sum(x) / len(x) for k, x in per.items()}
print("generations", sys.argv[2], {k: f"{x:.4g}" for k, x in sorted(v.items())})
print("   lane utilisation %.3f  waiting %.3f" % (v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"]), v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]))
PY
done
