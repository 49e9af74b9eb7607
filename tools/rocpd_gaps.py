"""Where the time BETWEEN kernels goes: per-step span vs sum of kernel durations from a rocprofv3 kernel trace.

    python tools/rocpd_gaps.py run_results.db [--anchor encode_m16] [--last N] [--skip-last M]

A step is the interval between consecutive dispatches of the anchor kernel (default: the fused encoder).  For the last
N steps (default 200; --skip-last M drops the final M first) prints the mean span, the mean sum of kernel durations, and
the idle time on the queue attributed to the kernel that FOLLOWS each gap (gap = start[i] - max(end of earlier kernels)).
"""
import re
import sqlite3
import sys


def arg(name, default):
    return type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def main():
    db = sqlite3.connect(sys.argv[1])
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    suffix = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))[len("rocpd_kernel_dispatch"):]
    kd, ks = "rocpd_kernel_dispatch" + suffix, "rocpd_info_kernel_symbol" + suffix
    scols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    rows = db.execute(f"select s.{name_col}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    anchor = arg("--anchor", "encode_m16")
    last, skip = arg("--last", 200), arg("--skip-last", 0)
    short = lambda n: re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", re.sub(r"\(.*", "", n))[:48]  # noqa: E731
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(idx) < 3:
        raise SystemExit(f"anchor kernel '{anchor}' has {len(idx)} dispatches")
    if skip:
        idx = idx[:-skip]
    idx = idx[-(last + 1):]
    spans, sums, gaps, counts = [], [], {}, {}
    for a, b in zip(idx[:-1], idx[1:]):
        seg = rows[a:b]
        spans.append(rows[b][1] - seg[0][1])
        sums.append(sum(e - s for _, s, e in seg))
        hi = seg[0][2]
        for (n, s, e) in seg[1:] + [rows[b]]:
            g = max(0, s - hi)
            key = short(n)
            gaps[key] = gaps.get(key, 0) + g
            counts[key] = counts.get(key, 0) + 1
            hi = max(hi, e)
    n = len(spans)
    ms = lambda v: v / n / 1e6  # noqa: E731
    print(f"steps analysed: {n} (anchor '{anchor}', launches per step: {sum(counts.values()) / n:.1f})")
    print(f"mean span per step      {ms(sum(spans)):8.4f} ms")
    print(f"mean kernel-time sum    {ms(sum(sums)):8.4f} ms")
    print(f"mean idle on the queue  {ms(sum(spans) - sum(sums)):8.4f} ms  ({100 * (sum(spans) - sum(sums)) / sum(spans):.1f} % of the span)")
    print(f"{'idle before kernel':50s} {'per step':>9s} {'us/step':>9s} {'us/gap':>8s}")
    for k, g in sorted(gaps.items(), key=lambda kv: -kv[1])[:25]:
        print(f"{k:50s} {counts[k] / n:9.2f} {g / n / 1e3:9.2f} {g / counts[k] / 1e3:8.2f}")


if __name__ == "__main__":
    main()
