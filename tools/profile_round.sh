#!/bin/bash
# Collects the round's profiles on the GPU box.  Usage: tools/profile_round.sh <tag>
#   1. rocprofv3 kernel stats of bench.py's headline loop (frames in flight, as the benchmark runs them: the launches of
#      the frame kernel overlap, bench.py reports how many run at once beside their mean duration);
#   2. per benchmark configuration: rocprofv3 kernel stats and PMC passes (one run each) over a few first frames;
#   3. the per-configuration roofline table (tools/config_roofline.py).
# tools/summarize_profile.py turns the output directory into the summaries kept under profiles/.
set -u
TAG=${1:-r2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --headline-only > $OUT/bench_under_rocprof.json 2> $OUT/bench.err || exit 1
while read name scene w h sub gen; do
  export RTX_GENERATIONS=$gen          # (tools/prof_frame.py: 0 = per-lane paths, what a lone frame runs; 1 = generations)
  D=$OUT/$name
  mkdir -p $D
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $GRAFT_REPO_ROOT/tools/prof_frame.py $scene $w $h $sub 6 > /dev/null 2>&1 || exit 1
  pass() { p=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $D/$p -- python3 $GRAFT_REPO_ROOT/tools/prof_frame.py $scene $w $h $sub 5 > /dev/null 2>&1 || exit 1; }
  pass pmc_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
  pass pmc_sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT
  pass pmc_cache TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
  pass pmc_fetch FETCH_SIZE
  pass pmc_write WRITE_SIZE
  echo "$name done"
done <<CONFIGS
scene7_4k scene7 3840 2160 6 0
scene7_4k_generations scene7 3840 2160 6 1
scene1 scene1 640 480 0 0
scene4-specular scene4-specular 640 480 0 0
scene5 scene5 640 480 0 0
scene6 scene6 640 480 0 0
CONFIGS
unset RTX_GENERATIONS
timeout -k 10 120 python3 $GRAFT_REPO_ROOT/tools/config_roofline.py > $OUT/configs.jsonl 2> $OUT/configs.err || exit 1
ls $OUT
