"""Turns a tools/profile_round.sh output directory into the small text/CSV summaries kept under profiles/:
   <tag>_kernel_stats.csv, <tag>_bench_under_rocprof.json   rocprofv3 kernel stats of bench.py and its JSON line
   <tag>_<config>_kernel_stats.csv, <tag>_<config>_pmc.txt    per configuration: kernel stats and PMC summary
   <tag>_traffic.json                                         memory-side bytes per launch of the 4K frame kernel
   <tag>_configs.jsonl                                        roofline row per configuration

    python tools/summarize_profile.py gpurun_out/prof_<tag> <tag>"""
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
if os.path.exists(os.path.join(src, "configs.jsonl")):
    shutil.copy(os.path.join(src, "configs.jsonl"), os.path.join(dst, f"{tag}_configs.jsonl"))


def summarize(cfg_dir, name):
    for f in glob.glob(os.path.join(cfg_dir, "trace", "*", "*kernel_stats.csv")):
        shutil.copy(f, os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"))
    kernel = "k_frame_gen (generation pipeline)" if name.endswith("_generations") else "k_frame (per-lane paths)"
    lines = [f"PMC summary ({tag}, {name}): {kernel}, one frame at a time, nested search, depth 5, every frame scheduled as a first frame; per "
             "launch (mean over the launches of one run).",
             "Collected with rocprofv3 --pmc, one run per bracketed group (tools/profile_round.sh)."]
    vals = {}
    for d in sorted(glob.glob(os.path.join(cfg_dir, "pmc_*"))):
        for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
            per, dur = collections.defaultdict(list), {}
            for r in csv.DictReader(open(f)):
                if "k_frame<" in r["Kernel_Name"] or "k_frame_gen<" in r["Kernel_Name"]:
                    per[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    dur[int(r["Start_Timestamp"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            agg = {k: sum(v) / max(len(v), 1) for k, v in per.items()}
            vals.update(agg)
            lines.append(f"[{os.path.basename(d)}] kernel duration under this pass: {sum(dur.values()) / max(len(dur), 1) / 1e6:.3f} ms ({len(dur)} launches)")
            lines += [f"    {k:32s} {agg[k]:.6g}" for k in sorted(agg)]
    v = vals
    lines.append("derived:")
    if "SQ_THREAD_CYCLES_VALU" in v:
        lines.append(f"    VALU lane utilisation = THREAD_CYCLES_VALU / (64 * ACTIVE_INST_VALU) = {v['SQ_THREAD_CYCLES_VALU'] / (64 * v['SQ_ACTIVE_INST_VALU']):.3f}")
        lines.append(f"    wave cycles waiting (WAIT_ANY / WAVE_CYCLES) = {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.3f}; issue stalls (WAIT_INST_ANY / WAVE_CYCLES) = {v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f}")
    if "TCC_HIT_sum" in v:
        lines.append(f"    L2 hit rate = {v['TCC_HIT_sum'] / max(v['TCC_HIT_sum'] + v['TCC_MISS_sum'], 1):.3f}; L1 hit rate = {1 - v['TCP_TCC_READ_REQ_sum'] / max(v['TCP_TOTAL_CACHE_ACCESSES_sum'], 1):.3f}")
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        # FETCH_SIZE / WRITE_SIZE are in KiB and count the L2's requests to the fabric, Infinity Cache hits included; on
        # gfx950 FETCH_SIZE reads half the bytes of wide coalesced streams, so the doubled figure is an upper bound for
        # this gather-heavy kernel
        fetch, write = v["FETCH_SIZE"] * 1024, v["WRITE_SIZE"] * 1024
        lines.append(f"    memory-side traffic per launch (L2 <-> fabric, Infinity Cache hits included): WRITE_SIZE {write / 1e6:.1f} MB, FETCH_SIZE "
                     f"{fetch / 1e6:.1f} MB as counted ({2 * fetch / 1e6:.1f} MB with the gfx950 x2 correction)")
        if name == "scene7_4k_generations":
            json.dump({"traffic_bytes_per_launch": write + 2 * fetch, "write_bytes": write, "fetch_bytes_counted": fetch},
                      open(os.path.join(dst, f"{tag}_traffic.json"), "w"))
    open(os.path.join(dst, f"{tag}_{name}_pmc.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


for cfg_dir in sorted(glob.glob(os.path.join(src, "*"))):
    if os.path.isdir(cfg_dir) and glob.glob(os.path.join(cfg_dir, "pmc_*")):
        summarize(cfg_dir, os.path.basename(cfg_dir))
