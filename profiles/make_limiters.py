#!/usr/bin/env python3
"""Turns the rocprofv3 passes of profiles/profile.sh into
   profiles/<round>_<workload>_summary.txt   kernel table + counters per launch (text, for reading)
   profiles/<round>_limiters.json            per workload and kernel: duration, HBM traffic, TA / vector-issue / LDS occupancy;
                                             "_meta": sha256 of the kernel sources the passes were taken on (bench.csrc_sha256)
bench.py quotes the newest JSON in its roofline objects (traffic, limiter) and marks the quote stale when the kernel sources
have changed since, so every number in the bench line can be recomputed from a file in this directory.

    python3 profiles/make_limiters.py r03 gpurun_out/<tag> fixed variable full

Formulas (MI355X: 256 CUs = 256 TAs = 256 LDS, 1024 SIMDs, 32 shader engines):
  cycles      = SQ_BUSY_CYCLES / 32                     (the kernel's duration in shader clocks)
  valu_issue  = SQ_INSTS_VALU * 4 / 1024 / cycles       (a wave64 fp64 instruction holds its SIMD for 4 clocks)
  valu_active = SQ_ACTIVE_INST_VALU * 4 / 1024 / cycles (SQ_ACTIVE_* count quad-cycles)
  ta_busy     = TA_TA_BUSY_sum / 256 / cycles           (cycles of the TA pass)
  lds_busy    = SQ_LDS_IDX_ACTIVE / 256 / cycles
  traffic     = 2 * FETCH_SIZE + WRITE_SIZE  (KB -> bytes; gfx950 tallies wide reads at half their bytes,
                MI355X_MICROARCH.md "HBM"; an upper bound for the 16..96-byte gathers)
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"sph::\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"rocprim::.*?detail::", "rocprim::", name)
    return name[:70]


def rows(d, pat):
    out = []
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        with open(f) as fh:
            out += list(csv.DictReader(fh))
    return out


def counters(d):
    acc = defaultdict(lambda: defaultdict(list))
    for r in rows(d, "*counter_collection.csv"):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}, {k: len(next(iter(cs.values()))) for k, cs in acc.items()}


def main():
    rnd, base = sys.argv[1], sys.argv[2]
    workloads = sys.argv[3:] or ["fixed", "variable", "full"]
    path = os.path.join(ROOT, "profiles", f"{rnd}_limiters.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    sys.path.insert(0, ROOT)
    import bench
    sha = bench.csrc_sha256()
    if out.get("_meta", {}).get("csrc_sha256") not in (None, sha):
        out = {}                                        # passes of different code are not mixed in one file
    git = None
    try:
        import subprocess
        git = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        pass
    out["_meta"] = {"csrc_sha256": sha, "git": git or out.get("_meta", {}).get("git"),
                    "note": "git = HEAD of the tree the passes ran on when known (the GPU box has no .git: see the commit that adds this file)"}
    for w in workloads:
        o = f"{base}_{w}"
        stats = rows(o + "_kt", "*kernel_stats.csv")
        if not stats:
            print(f"no kernel stats under {o}_kt", file=sys.stderr)
            continue
        lines = [f"== {w}: rocprofv3 --kernel-trace --stats ==", f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'%':>6s}"]
        kern = {}
        for r in stats:
            k = short(r["Name"])
            kern[k] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6,
                       "pct": float(r["Percentage"])}
            lines.append(f"{k:70s} {kern[k]['calls']:6d} {kern[k]['avg_us']:10.1f} {kern[k]['total_ms']:10.3f} {kern[k]['pct']:6.2f}")
        fetch, _ = counters(o + "_fetch")
        write, _ = counters(o + "_write")
        sq, nsq = counters(o + "_sq")
        ta, _ = counters(o + "_ta")
        lines.append("== per launch (mean over launches): traffic, occupancy of the TA, the vector ALUs and the LDS ==")
        rec = {}
        for k, info in sorted(kern.items(), key=lambda kv: -kv[1]["total_ms"]):
            e = dict(info)
            if k in fetch and k in write:
                e["fetch_size_raw_bytes"] = fetch[k]["FETCH_SIZE"] * 1024
                e["write_size_bytes"] = write[k]["WRITE_SIZE"] * 1024
                e["traffic_bytes_per_launch"] = 2 * e["fetch_size_raw_bytes"] + e["write_size_bytes"]
            if k in sq and sq[k].get("SQ_BUSY_CYCLES", 0) > 0:
                cyc = sq[k]["SQ_BUSY_CYCLES"] / 32.0
                e["cycles"] = cyc
                e["valu_insts"] = sq[k]["SQ_INSTS_VALU"]
                e["valu_issue"] = sq[k]["SQ_INSTS_VALU"] * 4 / 1024 / cyc
                e["valu_active"] = sq[k]["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc
                e["lds_busy"] = sq[k]["SQ_LDS_IDX_ACTIVE"] / 256 / cyc
                e["salu_insts"] = sq[k]["SQ_INSTS_SALU"]
                e["lds_insts"] = sq[k]["SQ_INSTS_LDS"]
                e["vmem_rd_insts"] = sq[k]["SQ_INSTS_VMEM_RD"]
            if k in ta and ta[k].get("SQ_BUSY_CYCLES", 0) > 0:
                cyc = ta[k]["SQ_BUSY_CYCLES"] / 32.0
                e["ta_busy"] = ta[k]["TA_TA_BUSY_sum"] / 256 / cyc
                e["lds_bank_conflict_cycles"] = ta[k].get("SQ_LDS_BANK_CONFLICT", 0.0)
                if ta[k].get("GRBM_GUI_ACTIVE", 0) > 0 and info["avg_us"] > 0:
                    e["clock_GHz_est"] = ta[k]["GRBM_GUI_ACTIVE"] / 8 / (info["avg_us"] * 1e3)
            rec[k] = e
            if info["pct"] >= 0.5:
                parts = [f"{x}={e[x]:.3g}" for x in ("traffic_bytes_per_launch", "valu_issue", "valu_active", "ta_busy", "lds_busy", "clock_GHz_est") if x in e]
                lines.append(f"{k:70s} " + "  ".join(parts))
        out[w] = {"command": f"bash profiles/profile.sh {rnd} <tag> {w}", "kernels": rec}
        with open(os.path.join(ROOT, "profiles", f"{rnd}_{w}_summary.txt"), "w") as fh:
            fh.write("\n".join(lines) + "\n")
        print("\n".join(lines[:14]))
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
