"""Per-phase timeline of the attention main loop (needs the profiling build: make -C visualcloze_amd/csrc debug;
VC_HIP_LIB=visualcloze_amd/lib/libvcloze_hip_dbg.so python tools/attn_ts.py)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualcloze_amd import hip
dev = "cuda:0"
L, H = 3968, 24
qkv = torch.randn(L, 3 * H * 128, device=dev).to(torch.bfloat16)
vt = torch.randn(H, 128, L, device=dev).to(torch.bfloat16)
o = torch.empty(L, H * 128, dtype=torch.bfloat16, device=dev)
for variant in (1, 0):
    ts = torch.zeros(2, 4096, dtype=torch.int64, device=dev)
    for _ in range(2):
        hip.attention(qkv, vt, o, L, H, variant=variant)
    torch.cuda.synchronize()
    hip.lib().vc_debug_set_attn_ts(ctypes.c_void_p(ts.data_ptr()))
    hip.attention(qkv, vt, o, L, H, variant=variant)
    torch.cuda.synchronize()
    hip.lib().vc_debug_set_attn_ts(ctypes.c_void_p(0))
    t = ts.cpu()
    print("variant", variant)
    for blk in range(2):
        v = [int(x) for x in t[blk] if int(x) != 0]
        print(f" block {blk}: {len(v)} stamps")
        for kt in range(20, 26):
            r = v[kt * 4:(kt + 1) * 4 + 1]
            if len(r) < 5: break
            print(f"  kt{kt}: stage+QK {r[1]-r[0]:5d} | softmax {r[2]-r[1]:5d} | PV {r[3]-r[2]:5d} | vmcnt+barrier {r[4]-r[3]:5d} | total {r[4]-r[0]}")
