#!/bin/bash
# development: SQ counter passes of the generation pipeline over a few first frames of scene7 4K, per library variant
#   tools/pmc_ab.sh "<name> ..."   (names as in tools/ab_r3.sh)
cd /tmp && export TMPDIR=/tmp
export RTX_GENERATIONS=1
for v in $1; do
  if [ "$v" = tree ]; then export RTX_LIB_PATH=$GRAFT_REPO_ROOT/raytracer_amd/librtx.so; else export RTX_LIB_PATH=$GRAFT_REPO_ROOT/raytracer_amd/variants/$v.so; fi
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"; do
    D=$GRAFT_REPO_ROOT/gpurun_out/pmcab_$v
    rm -rf $D; mkdir -p $D
    timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/prof_frame.py scene7 3840 2160 6 5 > /dev/null 2>&1 || exit 1
    python3 - $D $v <<'PY'
import csv, glob, sys, collections
per = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_frame" in r["Kernel_Name"] and "clear" not in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
v = {k: sum(x) / len(x) for k, x in per.items()}
print(sys.argv[2], {k: f"{x:.4g}" for k, x in sorted(v.items())})
PY
  done
done
