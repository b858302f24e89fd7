"""Poll a GPU's clocks, power and temperature while another process keeps it busy: one JSON line per sample on stdout.

Started as a CHILD by bench.py's `sustained` extra and tools/live_soak (never exec'd from a process that has initialised the GPU, and it
never initialises one itself: it reads the amdgpu driver's sysfs files -- /sys/class/drm/card*/device/{pp_dpm_sclk, pp_dpm_mclk,
gpu_busy_percent, hwmon/hwmon*/{freq1_input, power1_average | power1_input, temp*_input}} -- and, where those are not readable, asks
`rocm-smi --json`, which talks to the same driver through librocm_smi, not through HIP).

    python tools/smi_poll.py [--pci 0000:75:00.0 | --device N] [--interval 0.2] [--seconds 0]        (runs until killed)

A box may show the cards of every GPU of its host (and one more per compute partition): --pci names the one the caller computes on
(hipDeviceGetPCIBusId / torch.cuda.get_device_properties(i).pci_bus_id); --device N is the N-th PCI amdgpu card.
"""
import argparse
import glob
import json
import os
import re
import subprocess
import sys
import time


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def cards():
    """amdgpu cards in PCI order: [(card directory, hwmon directory or None)]"""
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device"), key=lambda p: int(re.findall(r"card(\d+)", p)[0])):
        if "amdgpu" not in (os.path.realpath(os.path.join(d, "driver")) or ""):
            continue
        if "/platform/" in os.path.realpath(d):          # amdgpu_xcp_N: a compute partition's render node, not a card with sensors
            continue
        hw = sorted(glob.glob(os.path.join(d, "hwmon", "hwmon*")))
        out.append((d, hw[0] if hw else None))
    return out


def _current_level_mhz(text):
    """pp_dpm_sclk: lines like '1: 2400Mhz *' -- the starred level"""
    if not text:
        return None
    for line in text.splitlines():
        if line.rstrip().endswith("*"):
            m = re.search(r"(\d+)\s*[Mm][Hh]z", line)
            if m:
                return int(m.group(1))
    return None


def sample_sysfs(card, hwmon):
    s = {}
    v = _current_level_mhz(_read(os.path.join(card, "pp_dpm_sclk")))
    if v is not None:
        s["sclk_mhz"] = v
    v = _current_level_mhz(_read(os.path.join(card, "pp_dpm_mclk")))
    if v is not None:
        s["mclk_mhz"] = v
    v = _read(os.path.join(card, "gpu_busy_percent"))
    if v is not None and v.isdigit():
        s["busy_pct"] = int(v)
    if hwmon:
        v = _read(os.path.join(hwmon, "freq1_input"))
        if v and v.isdigit():
            s["sclk_mhz_hwmon"] = int(v) / 1e6
        for name in ("power1_average", "power1_input"):
            v = _read(os.path.join(hwmon, name))
            if v and v.isdigit():
                s["power_w"] = int(v) / 1e6
                break
        temps = {}
        for p in glob.glob(os.path.join(hwmon, "temp*_input")):
            v = _read(p)
            label = _read(p.replace("_input", "_label")) or os.path.basename(p)
            if v and v.lstrip("-").isdigit():
                temps[label] = int(v) / 1e3
        if temps:
            s["temp_c"] = max(temps.values())
            s["temps_c"] = temps
    return s


def sample_rocm_smi(device):
    """one `rocm-smi --json` call (hundreds of milliseconds): the fallback where sysfs shows nothing"""
    try:
        p = subprocess.run(["rocm-smi", "-d", str(device), "--showclocks", "--showpower", "--showtemp", "--showuse", "--json"],
                           capture_output=True, text=True, timeout=20)
        d = json.loads(p.stdout)
    except Exception:
        return {}
    card = next(iter(d.values())) if d else {}
    s = {}
    for k, v in card.items():
        kl = k.lower()
        m = re.search(r"([-+]?\d+(\.\d+)?)", str(v))
        if not m:
            continue
        x = float(m.group(1))
        if "sclk clock speed" in kl:
            s["sclk_mhz"] = x
        elif "mclk clock speed" in kl:
            s["mclk_mhz"] = x
        elif "power" in kl and "(w)" in kl and "power_w" not in s:
            s["power_w"] = x
        elif "temperature" in kl and "(c)" in kl:
            s["temp_c"] = max(s.get("temp_c", -1e9), x)
        elif kl.startswith("gpu use"):
            s["busy_pct"] = x
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--pci", default=None, help="PCI address of the card (domain:bus:device.function), e.g. 0000:75:00.0")
    ap.add_argument("--interval", type=float, default=0.2)
    ap.add_argument("--seconds", type=float, default=0.0)
    args = ap.parse_args()
    cs = cards()
    card, hwmon = cs[args.device] if args.device < len(cs) else (None, None)
    if args.pci:
        want = args.pci.lower()
        hit = [c for c in cs if os.path.basename(os.path.realpath(c[0])).lower() == want]
        if hit:
            card, hwmon = hit[0]
        elif cs:
            # not among the cards shown (a container may renumber the bus): take the card that is busiest over the first half second --
            # the caller is keeping exactly one of them busy
            t_end = time.time() + 0.5
            busy = [0] * len(cs)
            while time.time() < t_end:
                for k, (c, _) in enumerate(cs):
                    v = _read(os.path.join(c, "gpu_busy_percent"))
                    busy[k] += int(v) if v and v.isdigit() else 0
                time.sleep(0.05)
            card, hwmon = cs[max(range(len(cs)), key=lambda k: busy[k])]
    source = "sysfs"
    if card is None or not sample_sysfs(card, hwmon):
        source = "rocm-smi"
    print(json.dumps({"source": source, "card": card, "hwmon": hwmon}), flush=True)
    t_end = time.time() + args.seconds if args.seconds > 0 else None
    while t_end is None or time.time() < t_end:
        t = time.time()
        s = sample_sysfs(card, hwmon) if source == "sysfs" else sample_rocm_smi(args.device)
        s["t"] = t
        try:
            print(json.dumps(s), flush=True)
        except BrokenPipeError:
            return
        time.sleep(max(0.0, args.interval - (time.time() - t)))


if __name__ == "__main__":
    try:
        main()
    except KeyboardInterrupt:
        pass
