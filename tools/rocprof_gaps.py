"""Per-step busy/idle analysis of a rocprofv3 --kernel-trace results .db: finds the repeating step (euler_kernel marks the
end of each evaluation), and prints per-kernel time inside a steady-state step plus the idle gaps between kernels.

    python tools/rocprof_gaps.py /tmp/prof/x_results.db
"""
import re, sqlite3, sys, collections

def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0] if "<" not in name else name[: name.index(">") + 1]

def main(path):
    c = sqlite3.connect(path)
    try:
        rows = c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
    except sqlite3.OperationalError:
        names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
        print("no 'kernels' view; tables/views:", names); return
    rows = [(short(n), s, e, g, w) for n, s, e, g, w in rows]
    ends = [i for i, r in enumerate(rows) if r[0].startswith("euler_kernel")]
    if len(ends) < 6:
        print("too few steps", len(ends)); return
    # steady state: steps between the (n-11)th and the last euler kernel
    lo, hi = ends[-11], ends[-1]
    nsteps = 10
    seg = rows[lo + 1: hi + 1]
    span = seg[-1][2] - seg[0][1]
    busy = sum(e - s for _, s, e, _, _ in seg)
    gaps = [seg[i + 1][1] - seg[i][2] for i in range(len(seg) - 1)]
    print(f"steady-state: {nsteps} steps, {len(seg)/nsteps:.0f} kernels/step, span {span/nsteps/1e6:.3f} ms/step, busy {busy/nsteps/1e6:.3f} ms/step, "
          f"idle {(span-busy)/nsteps/1e6:.3f} ms/step ({100*(span-busy)/span:.1f} %), median gap {sorted(gaps)[len(gaps)//2]/1e3:.2f} us")
    agg = collections.OrderedDict()
    for n, s, e, g, w in seg:
        k = f"{n} grid={g // max(w, 1)}x{w}"
        a = agg.setdefault(k, [0, 0])
        a[0] += 1; a[1] += e - s
    print("kernel,calls_per_step,avg_us,ms_per_step,percent_of_busy")
    for k, (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"\"{k}\",{cnt/nsteps:.1f},{tot/cnt/1e3:.2f},{tot/nsteps/1e6:.3f},{100*tot/busy:.1f}")

if __name__ == "__main__":
    main(sys.argv[1])
