"""Summarise rocprofv3 counter_collection CSVs of tools/gemm_pmc_modes.py: per mode (dispatch order, groups of 3)."""
import csv, sys, collections
modes = ["normal", "A hot", "W hot", "both hot"]
for path in sys.argv[1:]:
    disp = collections.OrderedDict()
    with open(path) as f:
        for r in csv.DictReader(f):
            if "gemm_bf16" not in r["Kernel_Name"]:
                continue
            d = disp.setdefault(int(r["Dispatch_Id"]), {})
            d[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(disp)
    for g in range(len(ids) // 3):
        grp = [disp[i] for i in ids[3 * g + 1:3 * g + 3]]   # skip the first launch of each mode
        names = sorted(grp[0])
        print(f"{modes[g % 4]:9s} " + "  ".join(f"{n}={sum(x[n] for x in grp) / len(grp):.4g}" for n in names))
