import csv, sys, collections
rows = collections.OrderedDict()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if "gemm_bf16" not in r["Kernel_Name"]: continue
        d = rows.setdefault(int(r["Dispatch_Id"]), {"dur": float(r["End_Timestamp"]) - float(r["Start_Timestamp"])})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows); n = 12
for g in range(len(ids) // n):
    grp = [rows[i] for i in ids[g * n + 4:(g + 1) * n]]
    dur = sum(x["dur"] for x in grp) / len(grp)
    cyc = sum(x["GRBM_GUI_ACTIVE"] for x in grp) / len(grp)
    print(f"group {g}: {dur/1e3:7.1f} us  GRBM_GUI_ACTIVE {cyc:10.0f}  -> {cyc/dur:5.2f} GHz (if counter is per-SE/XCD summed, divide accordingly)")
