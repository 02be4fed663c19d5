#!/usr/bin/env python3
"""Copy the summaries scripts/profile.sh left under gpurun_out/prof_<tag>/ into profiles/<tag>/ (tracked) and merge their
hbm_traffic_entry.json records into profiles/hbm_traffic.json under the bench config's name.

    python scripts/collect_profiles.py r04_cfg2:cfg2 r04_cfg3:cfg3 r04_cfg2_bwd:cfg2_bwd ...
"""
import json
import pathlib
import shutil
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
KEEP = ("kernel_stats.csv", "kernel_stats_timed.csv", "dispatch_info.txt", "bench_line_under_rocprof.json", "hbm_traffic_entry.json",
        "pmc_sq1.csv", "pmc_sq2.csv", "pmc_fetch.csv", "pmc_write.csv", "pmc_l2.csv", "pmc_grbm.csv")
tfile = ROOT / "profiles" / "hbm_traffic.json"
traffic = json.loads(tfile.read_text())
for spec in sys.argv[1:]:
    tag, cfg = spec.split(":")
    src = ROOT / "gpurun_out" / f"prof_{tag}"
    dst = ROOT / "profiles" / tag
    dst.mkdir(parents=True, exist_ok=True)
    for name in KEEP:
        if (src / name).exists():
            shutil.copy(src / name, dst / name)
    rec = json.loads((src / "hbm_traffic_entry.json").read_text())
    if rec.get("hbm_bytes_per_launch") is None:
        print(f"{tag}: no traffic record", file=sys.stderr)
        continue
    traffic[cfg] = rec
    print(f"{tag} -> {cfg}: {rec['kernel']} {rec['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch, sources {rec['csrc_sha256_16']}")
tfile.write_text(json.dumps(traffic, indent=1) + "\n")
