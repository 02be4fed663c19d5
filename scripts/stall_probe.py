"""What happens ~0.35-0.5 s after load onset?  (the bench's untimed ramp steps over a one-time ~40 ms device stall.)
Runs the headline kernel back to back from an idle chip for 1.5 s, timing groups of 8 launches with HIP events, while a host thread
samples what an ordinary user may read of the device's power management - sysfs sclk / mclk level, average power, busy percent,
hwmon frequency - every ~2 ms with host timestamps.  Prints the slowest group and the samples around it.  Read-only: no setting is
changed.

    python scripts/stall_probe.py
"""
import glob
import pathlib
import sys
import threading
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch  # noqa: E402
import flash_attention_amd as fa  # noqa: E402


def my_cards():
    """sysfs nodes of the card this process computes on (a box shows every card of its host in sysfs); all of them if it cannot be told"""
    cards = sorted(glob.glob("/sys/class/drm/card*/device"))
    try:
        pr = torch.cuda.get_device_properties(0)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        mine = [c for c in cards if bdf in str(pathlib.Path(c).resolve())]
        return mine or cards
    except AttributeError:
        return cards


def readers():
    out = {}
    for card in my_cards():
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "gpu_busy_percent", "pp_dpm_fclk", "pp_dpm_socclk"):
            p = pathlib.Path(card) / name
            if p.exists():
                out[f"{pathlib.Path(card).parent.name}/{name}"] = p
        for hw in glob.glob(card + "/hwmon/hwmon*"):
            for name in ("power1_average", "power1_input", "freq1_input", "temp1_input", "temp2_input", "power1_cap"):
                p = pathlib.Path(hw) / name
                if p.exists():
                    out[f"{pathlib.Path(card).parent.name}/{name}"] = p
    return out


def current(p):
    try:
        txt = p.read_text().strip()
    except OSError as e:
        return f"<{e.errno}>"
    if "\n" in txt:       # a pp_dpm_* table: the active level is starred
        star = [ln for ln in txt.splitlines() if ln.endswith("*")]
        return star[0] if star else txt.splitlines()[-1]
    return txt


qkv = torch.randn(3, 128, 4096, 64, device="cuda").bfloat16()
fa.flash_attention_qkv_packed(qkv, 8, 8)
torch.cuda.synchronize()
rd = readers()
print("readable:", sorted(rd))
time.sleep(3.0)           # let the chip go idle
samples, stop = [], False


def sampler():
    while not stop:
        samples.append((time.perf_counter(), {k: current(p) for k, p in rd.items()}))
        time.sleep(0.002)


th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(0.05)
t0 = time.perf_counter()
groups = []
while time.perf_counter() - t0 < 1.5:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = time.perf_counter()
    e0.record()
    for _ in range(8):
        fa.flash_attention_qkv_packed(qkv, 8, 8)
    e1.record()
    e1.synchronize()
    groups.append((ts - t0, time.perf_counter() - t0, e0.elapsed_time(e1) / 8))
stop = True
th.join()
med = sorted(g[2] for g in groups)[len(groups) // 2]
slow = max(groups[6:], key=lambda g: g[2])
print(f"{len(groups)} groups of 8 launches; median {med * 1e3:.1f} us per launch; slowest after the first 6: {slow[2] * 1e3:.1f} us per launch "
      f"({slow[2] * 8:.1f} ms for the group) from t = {slow[0] * 1e3:.1f} ms to {slow[1] * 1e3:.1f} ms after load onset")
print("groups around it:", [(round(a * 1e3, 1), round(c * 1e3, 1)) for a, b, c in groups if abs(a - slow[0]) < 0.03])
print(f"host sampler: {len(samples)} samples, largest gap between two samples {max(b[0] - a[0] for a, b in zip(samples, samples[1:])) * 1e3:.1f} ms "
      f"(a host-side stall would show here)")
last = None
for t, vals in samples:
    rel = (t - t0) * 1e3
    if rel < -20 or rel > slow[1] * 1e3 + 60:
        continue
    key = tuple(sorted(vals.items()))
    near = slow[0] * 1e3 - 30 <= rel <= slow[1] * 1e3 + 30
    if key != last or (near and int(rel) % 4 == 0):
        print(f"  t={rel:8.1f} ms  " + "  ".join(f"{k.split('/')[-1]}={v}" for k, v in sorted(vals.items())))
        last = key
