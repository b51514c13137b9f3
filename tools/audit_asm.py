"""Audit of a hipcc .s for a kernel whose inline asm owns registers by name (attention64.hip): the kernel must have
no spills and no scratch, and every v_accvgpr_* must sit inside an ;;#ASMSTART / ;;#ASMEND block.  Also prints the
instruction mix of the kernel's largest loop.     python tools/audit_asm.py file.s kernel_substring"""
import re, sys
path, name = sys.argv[1], sys.argv[2]
txt = open(path).read()
ms = re.search(r"\.amdhsa_kernel\s+(_Z\w*%s\w*)" % name, txt)
if not ms:
    sys.exit(f"kernel *{name}* not found")
sym = ms.group(1)
m = re.search(r"^%s:[^\n]*\n(.*?)\n\s*s_endpgm" % re.escape(sym), txt, re.S | re.M)
if not m:
    sys.exit(f"body of {sym} not found")
body = m.group(1)
meta = txt[txt.index(".amdhsa_kernel " + sym):]
meta = meta[:meta.index(".end_amdhsa_kernel")]
def field(k):
    mm = re.search(r"\.amdhsa_%s\s+(\S+)" % k, meta)
    return mm.group(1) if mm else None
yaml = txt[txt.rindex("amdhsa.kernels"):]
ky = yaml[yaml.index(sym):]
def yfield(k):
    mm = re.search(r"\.%s:\s+(\S+)" % k, ky)
    return mm.group(1) if mm else None
print(f"{sym}: next_free_vgpr={field('next_free_vgpr')} accum_offset={field('accum_offset')} vgpr_count={yfield('vgpr_count')} "
      f"agpr_count={yfield('agpr_count')} sgpr_count={yfield('sgpr_count')} spill_vgpr={yfield('vgpr_spill_count')} "
      f"spill_sgpr={yfield('sgpr_spill_count')} scratch={yfield('private_segment_fixed_size')} lds={yfield('group_segment_fixed_size')}")
bad = []
inasm = False
n_insn = 0
for ln in body.splitlines():
    s = ln.strip()
    if s.startswith(";;#ASMSTART"): inasm = True
    elif s.startswith(";;#ASMEND"): inasm = False
    elif s and not s.startswith((";", ".", "s_nop")) and not s.endswith(":"):
        n_insn += 1
        if not inasm and ("v_accvgpr" in s or "scratch_" in s or "buffer_store" in s and "offen" in s):
            bad.append(s)
print(f"{n_insn} instructions; compiler-generated accvgpr / scratch instructions outside asm blocks: {len(bad)}")
for b in bad[:20]:
    print("   ", b)
ok = not bad and yfield('vgpr_spill_count') == '0' and yfield('private_segment_fixed_size') == '0'
# between the kernel's first and last s_barrier (the tile stream) every LDS wait is a COUNTED lgkmcnt and every LDS-DMA wait a
# counted vmcnt: a scalar load (s_load: lgkmcnt, returns out of order), a scratch access or a compiler-placed global load there
# would break the counts silently
lines = body.splitlines()
bar = [i for i, ln in enumerate(lines) if ln.strip() == "s_barrier"]
stray = []
if len(bar) >= 2:
    inasm = False
    for ln in lines[bar[0]:bar[-1]]:
        t = ln.strip()
        if t.startswith(";;#ASMSTART"): inasm = True
        elif t.startswith(";;#ASMEND"): inasm = False
        elif not inasm and re.match(r"(s_load|s_buffer_load|scratch_)", t):
            stray.append(t)
print(f"scalar loads / scratch accesses between the first and the last s_barrier: {len(stray)}")
for t in stray[:10]:
    print("   ", t)
ok = ok and not stray
# per-gap histogram of the main loop: instructions between consecutive MFMAs
gaps, cur = [], None
for ln in body.splitlines():
    s = ln.strip()
    if not s or s.startswith((";", ".")) or s.endswith(":"):
        continue
    if s.startswith("v_mfma"):
        if cur is not None: gaps.append(cur)
        cur = 0
    elif cur is not None:
        cur += 1
import collections
h = collections.Counter(min(g, 12) for g in gaps)
print("instructions between consecutive MFMAs (12 = 12 or more):", dict(sorted(h.items())))
sys.exit(0 if ok else 1)
