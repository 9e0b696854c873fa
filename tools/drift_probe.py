"""Developer probe (GPU): does the X-pass time of ONE plan drift with time under sustained load?  Three bursts of 80 executes
(512^3 fp64, stage events on every execute) separated by idle pauses, with the GPU's clocks / power sampled from sysfs."""
import glob, os, sys, threading, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["DFFT_TUNE"] = "0"
from distributedfft_amd import api

dev = torch.device("cuda:0")
n = 512
a = torch.complex(torch.rand(n ** 3, device=dev, dtype=torch.float64), torch.rand(n ** 3, device=dev, dtype=torch.float64))
b = torch.zeros_like(a)
p = api.Plan(n, n, n, a, b, None, 0, 1, api.FORWARD, api.PLAN_INPUT_FROM_IN)


def read(path):
    try:
        return Path(path).read_text()
    except Exception:
        return ""


cards = [c for c in glob.glob("/sys/class/drm/card*/device") if os.path.exists(c + "/pp_dpm_sclk")]
print("sysfs cards:", cards, flush=True)
samples, stop = [], False


def active(txt):
    for l in txt.splitlines():
        if l.strip().endswith("*"):
            return l.split(":")[1].replace("*", "").strip()
    return "?"


def sampler():
    c = cards[0] if cards else None
    hw = glob.glob(c + "/hwmon/hwmon*/power1_average") if c else []
    hw_in = glob.glob(c + "/hwmon/hwmon*/power1_input") if c else []
    while not stop:
        if c:
            pw = read(hw[0]).strip() if hw else (read(hw_in[0]).strip() if hw_in else "")
            samples.append((time.perf_counter(), active(read(c + "/pp_dpm_sclk")), active(read(c + "/pp_dpm_mclk")),
                            active(read(c + "/pp_dpm_fclk")), pw))
        time.sleep(0.004)


th = threading.Thread(target=sampler, daemon=True)
th.start()
for burst, pause in enumerate((0.0, 1.0, 0.05)):
    time.sleep(pause)
    t_b = time.perf_counter()
    ts = []
    for i in range(80):
        p.execute(api.EXEC_ASYNC)
        ts.append(p.stage_times())
    ts = np.array(ts) * 1e3
    print(f"burst {burst} (after {pause} s idle), t3 per execute [us]:", " ".join(f"{v * 1e3:.0f}" for v in ts[:, 3]), flush=True)
    print(f"burst {burst} t0 per execute [us]:", " ".join(f"{v * 1e3:.0f}" for v in ts[:, 0]), flush=True)
    t_e = time.perf_counter()
    seg = [s for s in samples if t_b <= s[0] <= t_e]
    print(f"burst {burst} clocks (sclk, mclk, fclk, power) every ~20 ms:", [s[1:] for s in seg[::5]][:12], flush=True)
# untimed back-to-back loop, X pass bracketed by nothing: whole-step time per group of 10
for rep in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(40):
        p.execute(api.EXEC_NO_TIMING)
    p.sync()
    print(f"untimed loop {rep}: {(time.perf_counter() - t) / 40 * 1e3:.4f} ms per step", flush=True)
stop = True
p.destroy()
