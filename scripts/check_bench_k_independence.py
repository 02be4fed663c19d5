"""Timing check, NOT a correctness test (it left the GPU suite in round 3: a timing-sensitive assertion on a device event nobody
controls does not belong in a parity gate).  bench.py's headline must not depend on K: a 20-step timed region against a 200-step
one, both behind the untimed ramp; bench.py's JSON line carries `ramp.slowest_group_ms_per_step`, so a stall that lands in the
ramp stays visible.  Exit code 1 when the short region reads more than 10 % below the long one.

    python scripts/check_bench_k_independence.py
"""
import json
import pathlib
import subprocess
import sys

root = pathlib.Path(__file__).resolve().parent.parent
vals = {}
for k in (20, 200):
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", "--steps", str(k), "--warmup", "5",
                          "--no-secondary", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    vals[k] = line["value"]
    print(k, "steps:", line["value"], "TFLOP/s; ramp:", line.get("ramp"))
sys.exit(0 if vals[20] >= 0.90 * vals[200] else 1)
