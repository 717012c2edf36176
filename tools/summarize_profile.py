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
    vals, durations = {}, {}
    for d in sorted(glob.glob(os.path.join(cfg_dir, "pmc_*"))):
        for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
            per, dur = collections.defaultdict(list), {}
            for r in csv.DictReader(open(f)):
                if "k_frame<" in r["Kernel_Name"] or "k_frame_gen<" in r["Kernel_Name"]:
                    per[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    dur[int(r["Start_Timestamp"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            agg = {k: sum(v) / max(len(v), 1) for k, v in per.items()}
            vals.update(agg)
            durations[os.path.basename(d)] = sum(dur.values()) / max(len(dur), 1) / 1e6
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
    if name == "scene7_4k_generations" and "SQ_INSTS_VALU" in v and "TCC_HIT_sum" in v:
        # what bench.py reports beside the roofline: the headline kernel's counters against the hardware's capacity
        # (1024 SIMDs at 2.4 GHz, a wave64 VALU instruction holds its SIMD for 4 cycles)
        ms = durations.get("pmc_sq1", 0.0)
        json.dump({"kernel": "k_frame_gen, scene7 @ 3840x2160 depth 5 sub 6, one launch at a time",
                   "kernel_ms_under_profiler": ms,
                   "valu_insts_per_launch": v["SQ_INSTS_VALU"], "salu_insts_per_launch": v["SQ_INSTS_SALU"],
                   "valu_issue_busy": v["SQ_INSTS_VALU"] * 4 / (1024 * 2.4e9 * ms * 1e-3) if ms else None,
                   "lane_utilisation": v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"]),
                   "waves_waiting": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"],
                   "issue_stalls": v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"],
                   "l2_hit_rate": v["TCC_HIT_sum"] / max(v["TCC_HIT_sum"] + v["TCC_MISS_sum"], 1),
                   "l1_hit_rate": 1 - v["TCP_TCC_READ_REQ_sum"] / max(v["TCP_TOTAL_CACHE_ACCESSES_sum"], 1),
                   "lds_bank_conflict_share_of_lds_cycles": (v["SQ_LDS_BANK_CONFLICT"] / v["SQ_ACTIVE_INST_LDS"]) if "SQ_LDS_BANK_CONFLICT" in v else None},
                  open(os.path.join(dst, f"{tag}_counters.json"), "w"))
    open(os.path.join(dst, f"{tag}_{name}_pmc.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


for cfg_dir in sorted(glob.glob(os.path.join(src, "*"))):
    if os.path.isdir(cfg_dir) and glob.glob(os.path.join(cfg_dir, "pmc_*")):
        summarize(cfg_dir, os.path.basename(cfg_dir))
