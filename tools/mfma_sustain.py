#!/usr/bin/env python3
"""Clock / power log beside a seconds-long bare fp32-MFMA loop (the 'sustained 121-129 TFLOP/s' ceiling needs
the clock and power beside it) -- at 4, 3, 2 and 1 waves per SIMD: the ceiling turned out to be a matter of how many waves
of a SIMD issue MFMAs (>= 3: 123.5 TFLOP/s; <= 2: 154.6), not of the chip's clock or power.  The parent never touches the GPU: it starts tools/mfma_sustain as a child and samples
freq1_input / power1_input of every amdgpu hwmon directory it can read, every 50 ms; the busy card is the one whose power rises.
usage (GPU box): python tools/mfma_sustain.py [seconds]   (builds the child with hipcc if it is missing)"""
import glob, os, subprocess, sys, threading, time
here = os.path.dirname(os.path.abspath(__file__))
exe = os.path.join(here, "mfma_sustain")
if not os.path.exists(exe):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", os.path.join(here, "mfma_sustain.hip"), "-o", exe])
secs = sys.argv[1] if len(sys.argv) > 1 else "5"
hw = [d for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*") if os.path.exists(os.path.join(d, "freq1_input"))]
def read(d):
    try:
        return int(open(os.path.join(d, "freq1_input")).read()) / 1e6, int(open(os.path.join(d, "power1_input")).read()) / 1e6
    except (OSError, ValueError):
        return None
# (operands, workgroups of 4 waves per CU = waves per SIMD): the rate depends on how many waves of a SIMD issue MFMAs
for mode, wgs in (("1", "4"), ("1", "3"), ("1", "2"), ("1", "1"), ("0", "4")):
    idle = {d: read(d) for d in hw}
    samples, stop = {d: [] for d in hw}, [False]
    def loop():
        t0 = time.time()
        while not stop[0]:
            for d in hw:
                r = read(d)
                if r:
                    samples[d].append((time.time() - t0,) + r)
            time.sleep(0.05)
    th = threading.Thread(target=loop, daemon=True); th.start()
    out = subprocess.run([exe, secs, mode, "0", wgs], capture_output=True, text=True).stdout
    stop[0] = True; th.join()
    print(out.strip())
    busy = max(hw, key=lambda d: (sum(s[2] for s in samples[d]) / max(len(samples[d]), 1)) - (idle[d][1] if idle[d] else 0)) if hw else None
    if busy:
        s = samples[busy]
        clk = sorted(x[1] for x in s); pw = [x[2] for x in s]
        print(f"sensors ({busy.split('/device')[0]}): idle {idle[busy]}; under load n={len(s)} sclk MHz min/median/max "
              f"{clk[0]:.0f}/{clk[len(clk)//2]:.0f}/{clk[-1]:.0f}  power W mean/max {sum(pw)/len(pw):.0f}/{max(pw):.0f}")
        print("  trace (t s, MHz, W): " + " ".join(f"({a:.1f},{b:.0f},{c:.0f})" for a, b, c in s[::10]))
