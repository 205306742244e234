"""Digest the rocprofv3 output of tools/profile_round.sh into the small files profiles/ keeps.

  <tag>_kernel_stats_<dtype>.csv : per-kernel calls / total / average / share (from --kernel-trace --stats)
  <tag>_pmc_traffic.json         : per-kernel HBM bytes per launch from FETCH_SIZE / WRITE_SIZE

Corrections (MI355X_MICROARCH.md, "HBM"): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB-like units
of 1024 B, and on gfx950 FETCH_SIZE tallies the 128-B requests of a wide coalesced read at 64 B, so
fetch bytes = FETCH_SIZE * 1024 * 2.  WRITE_SIZE is uncalibrated and is reported as measured * 1024.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def stats(odir, tag, dt):
    files = glob.glob(os.path.join(odir, f"stats_{dt}", "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        return
    rows = list(csv.DictReader(open(files[0])))
    out = os.path.join(odir, f"{tag}_kernel_stats_{dt}.csv")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "percent"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], f'{float(r["TotalDurationNs"]) / 1e3:.1f}',
                        f'{float(r["AverageNs"]) / 1e3:.2f}', f'{float(r["MinNs"]) / 1e3:.2f}',
                        f'{float(r["MaxNs"]) / 1e3:.2f}', r["Percentage"]])


def pmc(odir, counter, dt):
    files = glob.glob(os.path.join(odir, f"pmc_{counter}_{dt}", "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: [0, 0.0])
    for fn in files:
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in acc.items()}


def main():
    odir, tag = sys.argv[1], sys.argv[2]
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "proxytransformation_amd", "libproxyt_hip.so")
    traffic = {"config": "cfg2", "scenes_per_gpu": 4, "units": "bytes per launch", "correction": "FETCH_SIZE*1024*2 (gfx950 half-count), WRITE_SIZE*1024",
               # the binary these counters were taken with (bench.py reports traffic_stale when it runs another one)
               "so_sha16": hashlib.sha256(open(so, "rb").read()).hexdigest()[:16]}
    for dt in ("bf16", "f32"):
        stats(odir, tag, dt)
        fetch, write = pmc(odir, "FETCH_SIZE", dt), pmc(odir, "WRITE_SIZE", dt)
        traffic[dt] = {k: {"fetch_bytes": fetch[k] * 2048.0, "write_bytes": write.get(k, 0.0) * 1024.0,
                           "raw_FETCH_SIZE": fetch[k], "raw_WRITE_SIZE": write.get(k)}
                       for k in sorted(fetch) if k.startswith("k_") or "k_" in k}
    # matrix-pipe occupancy at 32 scenes per GPU (profile_round.sh step 4): SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024
    # SIMDs' matrix pipes, SQ_BUSY_CYCLES the busy cycles of the 32 SQs (one per shader engine), so
    #   mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 * 1024)
    # is the share of the launch during which an average SIMD's matrix pipe was executing (VERDICT r05 "next" #3)
    files = glob.glob(os.path.join(odir, "pmc_mfma_b32", "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for fn in files:
        for r in csv.DictReader(open(fn)):
            a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    mf = {}
    for k, c in acc.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and any(t in k for t in ("k_gemm", "k_proxy_attn", "k_mlp", "k_attn32")):
            busy, sq = c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / c["SQ_VALU_MFMA_BUSY_CYCLES"][0], c["SQ_BUSY_CYCLES"][1] / c["SQ_BUSY_CYCLES"][0]
            mf[k] = {"SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_BUSY_CYCLES": sq, "mfma_busy_frac": round(busy / (sq / 32.0 * 1024.0), 4) if sq else None,
                     "launches": c["SQ_BUSY_CYCLES"][0]}
    if mf:
        traffic["mfma_at_32_scenes"] = {"what": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 * 1024): share of the launch an average "
                                                "SIMD's matrix pipe was busy (cfg2, 32 scenes per GPU, bf16 features)", "kernels": dict(sorted(mf.items()))}
    json.dump(traffic, open(os.path.join(odir, f"{tag}_pmc_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
