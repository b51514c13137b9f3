"""Independent reading of the shader clock while one kernel runs in a loop: launches attention (variant 12 / 3) or the
GATE_RES GEMM back to back for a few seconds on random data and samples `rocm-smi --showclocks` meanwhile.
    python tools/clock_probe.py"""
import os, subprocess, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualcloze_amd import hip
dev = "cuda:0"
L, H, D = 3968, 24, 3072
qkv = torch.randn(L, 3 * D, device=dev).to(torch.bfloat16)
vt = torch.randn(H, 128, L, device=dev).to(torch.bfloat16)
o = torch.empty(L, D, dtype=torch.bfloat16, device=dev)
a4 = torch.randn(L, 4 * D, device=dev).to(torch.bfloat16)
w4 = (torch.randn(D, 4 * D, device=dev) * (4 * D) ** -0.5).to(torch.bfloat16)
b = torch.zeros(D, dtype=torch.bfloat16, device=dev)
gate = torch.randn(D, device=dev).to(torch.bfloat16)
x = torch.randn(L, D, device=dev).to(torch.bfloat16)
prob = hip.make_problem(a4, w4, b, x, res=x, gate=gate)

def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        return [ln.strip() for ln in out.splitlines() if "sclk" in ln.lower()][:2]
    except Exception as e:
        return [repr(e)]

def loop(name, fn, seconds=4.0):
    stop = [False]
    samples = []
    def sampler():
        time.sleep(1.0)
        while not stop[0]:
            samples.append(smi())
            time.sleep(0.6)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop[0] = True; th.join()
    print(f"{name}: {e0.elapsed_time(e1) * 1e3 / n:.1f} us per launch; rocm-smi sclk samples: {samples}", flush=True)

qz, vz = torch.zeros_like(qkv), torch.zeros_like(vt)
print("idle:", smi(), flush=True)
if os.environ.get("VC_PROBE", "all") == "all":
    loop("attention variant 12, ZERO operands (same instruction stream, no toggling)", lambda: hip.attention(qz, vz, o, L, H, variant=12))
    loop("attention variant 3, ZERO operands", lambda: hip.attention(qz, vz, o, L, H, variant=3))
    loop("attention variant 12", lambda: hip.attention(qkv, vt, o, L, H, variant=12))
    loop("attention variant 3", lambda: hip.attention(qkv, vt, o, L, H, variant=3))
loop("GEMM GATE_RES 3968x3072x12288", lambda: hip.gemm(prob, epi=hip.EPI_GATE_RES))
az, wz, xz = torch.zeros_like(a4), torch.zeros_like(w4), torch.zeros_like(x)
probz = hip.make_problem(az, wz, b, xz, res=xz, gate=gate)
loop("GEMM GATE_RES 3968x3072x12288, ZERO operands", lambda: hip.gemm(probz, epi=hip.EPI_GATE_RES))
a1, w1 = torch.randn(L, D, device=dev).to(torch.bfloat16), (torch.randn(3 * D, D, device=dev) * D ** -0.5).to(torch.bfloat16)
y1, b1 = torch.empty(L, 3 * D, dtype=torch.bfloat16, device=dev), torch.zeros(3 * D, dtype=torch.bfloat16, device=dev)
loop("GEMM BIAS (qkv) 3968x9216x3072", lambda: hip.gemm(hip.make_problem(a1, w1, b1, y1)))
loop("GEMM BIAS (qkv) 3968x9216x3072, ZERO operands", lambda: hip.gemm(hip.make_problem(torch.zeros_like(a1), torch.zeros_like(w1), b1, y1)))
