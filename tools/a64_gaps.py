"""Per-gap listing of attention64's instruction stream: every instruction between consecutive MFMAs of the kernel, with an
issue-slot estimate from the measured prices (kernel header: MFMA 1, plain VALU 1, v_exp 2, ds_read_b128 3.2, SALU ~0.3).
    python tools/a64_gaps.py file.s kernel_substring [--loop]      (--loop: only the largest loop body)"""
import re, sys
path, name = sys.argv[1], sys.argv[2]
txt = open(path).read()
m = re.search(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)\n\s*s_endpgm" % name, txt, re.S | re.M)
body = m.group(2)
cost = {'v_exp_f32': 2, 'ds_read_b128': 3.2, 'global_load_lds_dwordx4': 6}
cur = None
out = []
for ln in body.splitlines():
    s = ln.strip()
    if s.endswith(":"):
        out.append(("LABEL", s))
        continue
    if not s or s.startswith((";", ".")):
        continue
    if s.startswith("v_mfma"):
        if cur is not None:
            out.append(("GAP", cur))
        cur = []
    elif cur is not None:
        cur.append(s.split()[0])
if cur:
    out.append(("GAP", cur))
g = 0
for k, v in out:
    if k == "LABEL":
        print("----", v)
    else:
        c = sum(cost.get(o, 0.3 if o.startswith('s_') else 1) for o in v)
        short = [o.replace('v_', '').replace('_f32', '').replace('_b32', '') for o in v]
        print(f"{g:4d} n={len(v):2d} slots={c:5.1f}  {' '.join(short)}")
        g += 1
