#!/bin/bash
# development: SQ counters of the frame kernel while three frames are in flight (the benchmark's headline loop without
# its gather), per library variant: mean per launch over all launches of the run.   tools/pmc_inflight.sh "<name> ..."
cd /tmp && export TMPDIR=/tmp
export RTX_TILE_FEEDBACK=0
for v in $1; do
  if [ "$v" = tree ]; then export RTX_LIB_PATH=$GRAFT_REPO_ROOT/raytracer_amd/librtx.so; else export RTX_LIB_PATH=$GRAFT_REPO_ROOT/raytracer_amd/variants/$v.so; fi
  D=$GRAFT_REPO_ROOT/gpurun_out/pmcif_$v
  rm -rf $D; mkdir -p $D
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/gpu_inflight.py 1 3 2 6 > $D/run.log 2>&1 || exit 1
  tail -1 $D/run.log
  python3 - $D $v <<'PY'
import csv, glob, sys, collections
per = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_frame" in r["Kernel_Name"] and "clear" not in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
v = {k: sum(x) / len(x) for k, x in per.items()}
print(sys.argv[2], len(per["SQ_INSTS_VALU"]), "launches", {k: f"{x:.4g}" for k, x in sorted(v.items())})
PY
done
