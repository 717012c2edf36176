#!/bin/bash
# development: one rocprofv3 counter pass per argument (a quoted list of counters each) over a few first frames of
# scene7 @ 4K in the generation pipeline; prints the mean per launch of the frame kernel.
#   tools/pmc_pass.sh "SQ_A SQ_B ..." "TCP_X ..."        (RTX_LIB_PATH / RTX_GENERATIONS are taken from the environment)
cd /tmp && export TMPDIR=/tmp
export RTX_GENERATIONS=${RTX_GENERATIONS:-1}
for pass in "$@"; do
  D=$GRAFT_REPO_ROOT/gpurun_out/pmcpass
  rm -rf $D; mkdir -p $D
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/prof_frame.py scene7 3840 2160 6 5 > $D/run.log 2>&1 || { tail -3 $D/run.log; exit 1; }
  python3 - $D <<'PY'
import csv, glob, sys, collections
per = collections.defaultdict(list)
dur = []
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_frame" in r["Kernel_Name"] and "clear" not in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
v = {k: sum(x) / len(x) for k, x in per.items()}
print(f"kernel {sum(dur) / max(len(dur), 1):.3f} ms", {k: f"{x:.4g}" for k, x in sorted(v.items())})
PY
done
