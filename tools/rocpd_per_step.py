"""Per-STEP kernel table from a rocprofv3 rocpd database: launches per step and microseconds per step of every kernel, over the last N
steps (a step = from one dispatch of the anchor kernel to the next).
    python tools/rocpd_per_step.py run_results.db [--anchor encode_m16] [--steps 20]"""
import re
import sqlite3
import sys


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def main():
    db = sqlite3.connect(sys.argv[1])
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    suffix = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))[len("rocpd_kernel_dispatch"):]
    kd, ks = "rocpd_kernel_dispatch" + suffix, "rocpd_info_kernel_symbol" + suffix
    scols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    rows = db.execute(f"select s.{name_col}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    anchor, n = arg("--anchor", "encode_m16"), int(arg("--steps", "20"))
    marks = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(marks) < n + 1:
        raise SystemExit(f"only {len(marks)} dispatches of {anchor!r}")
    lo, hi = marks[-n - 1], marks[-1]
    agg = {}
    for name, t0, t1 in rows[lo:hi]:
        name = re.sub(r"\(.*", "", name)
        name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
        a = agg.setdefault(name, [0, 0])
        a[0] += 1; a[1] += t1 - t0
    span = (rows[hi][1] - rows[lo][1]) / n / 1e3
    tot = sum(a[1] for a in agg.values()) / n / 1e3
    print(f"steps {n}: span {span:.1f} us/step, kernel time {tot:.1f} us/step, {sum(a[0] for a in agg.values()) / n:.1f} launches/step")
    print(f"{'kernel':64s} {'per step':>9s} {'us/step':>9s} {'avg us':>8s}")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name[:64]:64s} {a[0] / n:9.2f} {a[1] / n / 1e3:9.1f} {a[1] / a[0] / 1e3:8.1f}")


if __name__ == "__main__":
    main()
