"""Package power and shader clock of the device THIS process computes on (found by PCI address, not by index), sampled from hwmon / sysfs at
20 Hz while one fp32 GEMM launch repeats for a few seconds: ytvln_gemm_f32 beside the vendor library (torch.matmul), random-normal beside
all-zero operands.  Answers whether the library's lead on a shape is cycles or clock (energy per flop).  A yardstick, not a dependency.
usage: python tools/gemm_power.py [seconds per cell]"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import torch
from ytvln import ops

dev = torch.device("cuda", 0)
torch.backends.cuda.matmul.allow_tf32 = False
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
pr = torch.cuda.get_device_properties(0)
bdf = None
try:
    bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
except Exception as e:
    print("no PCI address in device properties:", e)
print("device:", pr.name, "pci", bdf)
base = f"/sys/bus/pci/devices/{bdf}" if bdf else None
cands = []
if base and os.path.isdir(base):
    cands = glob.glob(base + "/hwmon/hwmon*")
if not cands:       # fall back: every drm card, print what is there so the log says why nothing matched
    for c in sorted(glob.glob("/sys/class/drm/card*/device")):
        try:
            slot = [l.split("=")[1].strip() for l in open(c + "/uevent") if l.startswith("PCI_SLOT_NAME")]
        except Exception:
            slot = []
        print("  drm", c, slot, glob.glob(c + "/hwmon/hwmon*"))
        if bdf and slot and slot[0].lower() == bdf.lower():
            cands = glob.glob(c + "/hwmon/hwmon*")
hw = cands[0] if cands else None
print("hwmon:", hw, sorted(os.listdir(hw))[:40] if hw else "none visible in this container")

def rd(name):
    try:
        return float(open(os.path.join(hw, name)).read().strip())
    except Exception:
        return None

pname = next((n for n in ("power1_average", "power1_input") if hw and rd(n) is not None), None)
fname = next((n for n in ("freq1_input",) if hw and rd(n) is not None), None)
print("power file:", pname, " clock file:", fname, " caps (W):", {n: (rd(n) or 0) / 1e6 for n in ("power1_cap", "power1_cap_default", "power1_cap_max", "power1_cap_min")})

class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True); self.on = True; self.p = []; self.f = []
    def run(self):
        while self.on:
            if pname: self.p.append(rd(pname) / 1e6)
            if fname: self.f.append(rd(fname) / 1e6)
            time.sleep(0.05)

def cell(label, f, flop):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s = Sampler(); s.start()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(20): f()
        n += 20
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    s.on = False; s.join()
    us = e0.elapsed_time(e1) * 1000 / n
    tf = flop / us / 1e6
    p = s.p[len(s.p) // 3:]; fr = s.f[len(s.f) // 3:]       # drop the ramp
    pw = sum(p) / len(p) if p else None; mhz = sum(fr) / len(fr) if fr else None
    print(f"{label:46s} {us:8.1f} us {tf:6.1f} TF/s" + (f"  {pw:6.0f} W  {pw / tf:5.2f} pJ/flop" if pw else "  power n/a") + (f"  sclk {mhz:5.0f} MHz" if mhz else "  sclk n/a"), flush=True)

for M, N, K, ta, tb in ((16128, 3072, 1024, 0, 1), (4480, 3072, 768, 0, 1), (3072, 1024, 16128, 1, 0), (16128, 1024, 1024, 0, 1)):
    for data in ("randn", "zeros"):
        mk = torch.randn if data == "randn" else torch.zeros
        A = mk((K, M) if ta else (M, K), device=dev); B = mk((N, K) if tb else (K, N), device=dev)
        C = torch.empty(M, N, device=dev)
        lda = M if ta else K; ldb = K if tb else N
        Ao, Bo = (A.t() if ta else A), (B.t() if tb else B)
        fl = 2.0 * M * N * K
        cell(f"{M}x{N}x{K} tA{ta} tB{tb} {data:5s} ytvln", lambda: ops._gemm(A, lda, ta, B, ldb, tb, C, N, M, N, K), fl)
        cell(f"{M}x{N}x{K} tA{ta} tB{tb} {data:5s} torch", lambda: torch.matmul(Ao, Bo, out=C), fl)
        if data == "randn":       # the same launch over 12 rotating operand sets (~1-3 GB): nothing is left in the 256 MB Infinity Cache from the last use
            sets = [(mk(A.shape, device=dev), mk(B.shape, device=dev), torch.empty(M, N, device=dev)) for _ in range(12)]
            it = {"i": 0}
            def cold():
                a_, b_, c_ = sets[it["i"] % 12]; it["i"] += 1
                ops._gemm(a_, lda, ta, b_, ldb, tb, c_, N, M, N, K)
            cell(f"{M}x{N}x{K} tA{ta} tB{tb} {data:5s} ytvln, rotating operands", cold, fl)
            def cold_t():
                a_, b_, c_ = sets[it["i"] % 12]; it["i"] += 1
                torch.matmul(a_.t() if ta else a_, b_.t() if tb else b_, out=c_)
            cell(f"{M}x{N}x{K} tA{ta} tB{tb} {data:5s} torch, rotating operands", cold_t, fl)
            del sets
