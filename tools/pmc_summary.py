"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch) -> JSON on stdout.

    python tools/pmc_summary.py gpurun_out/pmc2/f/f_counter_collection.csv gpurun_out/pmc2/w/w_counter_collection.csv ...
"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0] if "<" not in name else name[: name.index(">") + 1]


def main(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        per_dispatch = collections.defaultdict(dict)
        with open(p) as f:
            for r in csv.DictReader(f):
                n = short(r["Kernel_Name"])
                if not any(k in n for k in ("gemm_bf16", "attn", "qknorm", "ln_modulate", "splitk", "pack_")):
                    continue
                per_dispatch[(n, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
                per_dispatch[(n, r["Dispatch_Id"])]["_dur_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        for (n, _), v in per_dispatch.items():
            for k, x in v.items():
                agg[n][k].append(x)
    out = {}
    for n, v in agg.items():
        out[n] = {k: sum(x) / len(x) for k, x in v.items()}
        out[n]["dispatches"] = len(next(iter(v.values())))
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1:])
