"""Turn a rocprofv3 results .db (--kernel-trace --stats) into a short per-kernel summary (CSV on stdout).

    python tools/rocprof_summary.py gpurun_out/prof1/r01_results.db > profiles/r01_kernel_stats.csv
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    if name.startswith("Cijk_"):
        return "hipBLASLt:" + name[:40] + "... (one-time LoRA merge via torch)"
    if "at::native" in name:
        inner = re.findall(r"(normal_kernel|uniform_kernel|FillFunctor|MulFunctor|CUDAFunctor_add|bfloat16tofloat32_copy|bfloat16_copy)", name)
        return "torch:" + (inner[0] if inner else "elementwise") + " (model setup / LoRA merge, not in step graph)"
    return name.split("(")[0] if "<" not in name else name[: name.index(">") + 1]


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    agg = {}
    for n, calls, tot, avg, pct in rows:
        k = short(n)
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += calls; a[1] += tot; a[2] += pct
    print("kernel,calls,total_us,avg_us,percent")
    for k, (calls, tot, pct) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"\"{k}\",{calls},{tot:.1f},{tot / calls:.3f},{pct:.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
