"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table.

    python tools/rocpd_stats.py gpurun_out/prof/run_results.db [--last N]  > profiles/rNN_kernels.txt

--last N: per kernel, only its last N dispatches (the steady-state tail of a long run).
"""
import re
import sqlite3
import sys


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    suffix = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))[len("rocpd_kernel_dispatch"):]
    kd, ks = "rocpd_kernel_dispatch" + suffix, "rocpd_info_kernel_symbol" + suffix
    cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    q = f"select s.{name_col}, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"
    rows = db.execute(q).fetchall()
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
    if last > 0:
        per = {}
        for name, dur in rows:
            per.setdefault(name, []).append(dur)
        rows = [(name, dur) for name, durs in per.items() for dur in durs[-last:]]
    agg = {}
    for name, dur in rows:
        name = re.sub(r"\(.*", "", name)
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name[:70]:70s} {a[0]:6d} {a[1]/1e6:10.3f} {a[1]/a[0]/1e3:10.1f} {a[2]/1e3:10.1f} {a[3]/1e3:10.1f} {100*a[1]/total:6.2f}")


if __name__ == "__main__":
    main()
