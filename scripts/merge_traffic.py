"""Merge the hbm_traffic_entry.json records written by scripts/profile.sh into profiles/hbm_traffic.json.
usage: python scripts/merge_traffic.py cfg2=profiles/r02_cfg2 cfg3=profiles/r02_cfg3 ..."""
import json, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
out = ROOT / "profiles" / "hbm_traffic.json"
data = {"_comment": "HBM/fabric bytes per launch of the dominant kernel from rocprofv3 PMC passes (scripts/profile.sh): FETCH_SIZE and "
        "WRITE_SIZE in KiB from separate --pmc passes; FETCH_SIZE doubled (gfx950 reports half the bytes of a wide coalesced read, "
        "MI355X_MICROARCH.md HBM section); Infinity-Cache hits are counted, so this is fabric traffic, an upper bound on HBM traffic. "
        "bench.py reports a record only while csrc_sha256_16 (sha256 over the kernel sources, bench.py --print-csrc-fingerprint) and the "
        "kernel name still match the build it runs."}
for spec in sys.argv[1:]:
    name, _, d = spec.partition("=")
    rec = json.loads((ROOT / d / "hbm_traffic_entry.json").read_text())
    rec["source"] = d
    data[name] = rec
out.write_text(json.dumps(data, indent=1) + "\n")
print(out.read_text())
