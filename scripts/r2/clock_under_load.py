"""Shader clock under load: scripts/diag/libclkmon.so times a chain of 4096 dependent scalar additions (one wave, priority 3,
its own stream) every 2 ms while different workloads run.  The chain's duration is inversely proportional to the clock."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
mon = C.CDLL(os.path.join(ROOT, "scripts", "diag", "libclkmon.so"))
W, H, NP = 1920, 1080, 32
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.surf_clip(W, H, NP + 1, device=torch.device("cuda"))
flows = torch.empty((NP, H, W, 2), dtype=torch.float32, device="cuda")
ctx = Context(W, H)
x = torch.randn(4096, 4096, device="cuda")

def sample(name, work, seconds=2.0, period_us=2000):
    n = int(seconds * 1e6 / period_us)
    torch.cuda.synchronize()
    assert mon.clkmon_start(n, period_us) == 0
    t0 = time.perf_counter(); it = 0
    while time.perf_counter() - t0 < seconds * 0.95:
        work(); it += 1
        if it % 4 == 0: torch.cuda.current_stream().synchronize()       # (not the device: the monitor is still running)
    torch.cuda.current_stream().synchronize()
    buf = (C.c_longlong * (2 * n))()
    assert mon.clkmon_read(buf) == n
    a = np.array(buf, dtype=np.float64).reshape(n, 2)
    t = a[n // 4:, 0]                                    # ticks of 10 ns per chain; skip the ramp
    print("%-34s chain of 4096 s_add: ticks p5 %6.0f  p50 %6.0f  p95 %6.0f  -> %.2f shader cycles per instruction at 2.4 GHz, "
          "i.e. %.0f MHz if a lone wave issues one per %.1f cycles   (%d work calls)"
          % (name, *np.percentile(t, [5, 50, 95]), np.median(t) * 24 / 4096, 2400 * BASE / np.median(t) if BASE else 0, CYC, it), flush=True)
    return float(np.median(t))

BASE, CYC = 0.0, 0.0
torch.mm(x, x); torch.cuda.current_stream().synchronize()
BASE = sample("light load (reference)", lambda: (x[:64].add_(1.0), time.sleep(0.002)))
CYC = BASE * 24 / 4096
sample("farneback clip (bench workload)", lambda: ctx.farneback_clip(frames, flows, **P))
sample("synthetic VALU burn, 32 waves per CU", lambda: mon.clkmon_burn(20000, 0))
sample("synthetic VALU + SALU burn", lambda: mon.clkmon_burn(20000, 1))
sample("torch fp32 matmul 4096^3", lambda: torch.mm(x, x))
sample("device copy 128 MB", lambda: flows[:8].copy_(flows[8:16]))
sample("light load again", lambda: (x[:64].add_(1.0), time.sleep(0.002)))
