"""Turns a tools/profile_round.sh output directory into the small text/CSV summaries kept under profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, f"{tag}_bench_under_rocprof.json"))
lines = [f"PMC summary ({tag}): k_frame on scene7 @ 3840x2160 depth 5, nested search sub 6, per launch (mean over the launches after the first).",
         "Collected with rocprofv3 --pmc, one pass per line group (tools/profile_round.sh)."]
vals = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        per, dur = collections.defaultdict(list), {}
        for r in csv.DictReader(open(f)):
            if "k_frame<" in r["Kernel_Name"]:
                per[r["Counter_Name"]].append((int(r["Start_Timestamp"]), float(r["Counter_Value"])))
                dur[int(r["Start_Timestamp"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        # the first launch runs without the previous frame's scheduling hint: leave it out
        starts = sorted(dur)[1:] if len(dur) > 1 else sorted(dur)
        agg = {k: sum(val for t, val in rows if t in set(starts)) / max(1, sum(1 for t, _ in rows if t in set(starts))) for k, rows in per.items()}
        for k in agg:
            vals[k] = agg[k]
        lines.append(f"[{os.path.basename(d)}] kernel duration under this pass: {sum(dur[t] for t in starts) / max(len(starts), 1) / 1e6:.3f} ms "
                     f"({len(starts)} launches)")
        lines += [f"    {k:32s} {agg[k]:.6g}" for k in sorted(agg)]
v = vals
lines.append("derived:")
if "SQ_THREAD_CYCLES_VALU" in v:
    lines.append(f"    VALU lane utilisation = THREAD_CYCLES_VALU / (64 * ACTIVE_INST_VALU) = {v['SQ_THREAD_CYCLES_VALU'] / (64 * v['SQ_ACTIVE_INST_VALU']):.3f}")
    lines.append(f"    wave cycles waiting (WAIT_ANY / WAVE_CYCLES) = {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.3f}; issue stalls (WAIT_INST_ANY / WAVE_CYCLES) = {v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f}")
if "TCC_HIT_sum" in v:
    lines.append(f"    L2 hit rate = {v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.3f}; L1 hit rate = {1 - v['TCP_TCC_READ_REQ_sum'] / v['TCP_TOTAL_CACHE_ACCESSES_sum']:.3f}")
    lines.append(f"    effective clock = GRBM_GUI_ACTIVE / 8 / duration")
if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
    # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads half the bytes of wide coalesced streams, so the
    # doubled figure is an upper bound for this gather-heavy kernel
    fetch, write = v["FETCH_SIZE"] * 1024, v["WRITE_SIZE"] * 1024
    lines.append(f"    HBM-side traffic per launch: WRITE_SIZE {write / 1e6:.1f} MB, FETCH_SIZE {fetch / 1e6:.1f} MB as counted "
                 f"({2 * fetch / 1e6:.1f} MB with the gfx950 x2 correction); algorithmic bytes = 99.5 MB framebuffer write")
    json.dump({"traffic_bytes_per_launch": write + 2 * fetch, "write_bytes": write, "fetch_bytes_counted": fetch},
              open(os.path.join(dst, f"{tag}_traffic.json"), "w"))
open(os.path.join(dst, f"{tag}_pmc_frame.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
