#!/bin/bash
# development A/B on the GPU box over library variants built with tools/variant.sh: cache counters of the scene7 4K
# first frame per variant (L2 hits/misses, bytes fetched from the fabric).   tools/ab_cache.sh "<name> ..."
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in $1; do
  export RTX_LIB_PATH=$R/raytracer_amd/variants/$v.so
  D=$R/gpurun_out/abc_$v
  rm -rf $D
  timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $D -- python3 $R/tools/prof_frame.py scene7 3840 2160 6 5 > /dev/null 2>&1 || exit 1
  python3 - "$D" "$v" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if row["Kernel_Name"].startswith("void k_frame"):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print(sys.argv[2], {k: round(x) for k, x in m.items()}, "L2 hit %.3f" % (m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])), flush=True)
PY
done
