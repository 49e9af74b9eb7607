"""Summarise the SQ counter passes of tools/collect_stalls.sh: per-kernel means of every counter for the kernels that
take the most time, plus the derived fractions the MI355X guide defines (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY
~= WAVE_CYCLES, all in quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs).

    python tools/stall_summary.py <workdir> <out.txt> <tag>
"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    workdir, out_txt, tag = sys.argv[1:4]
    vals = defaultdict(lambda: defaultdict(list))
    durs = defaultdict(list)
    for f in glob.glob(f"{workdir}/pmc_g*/**/*counter_collection.csv", recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"{workdir}/pmc_g1/**/*kernel_trace.csv", recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                durs[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    mean = lambda v: sum(v) / len(v) if v else float("nan")  # noqa: E731
    order = sorted(durs, key=lambda k: -sum(durs[k]))
    order = [k for k in order if "at::native" not in k and "rocprim" not in k][:8]
    lines = [f"rocprofv3 --kernel-trace --pmc <group> passes ({tag}; tools/collect_stalls.sh; bench.py --steps 3 --warmup 1), means per launch.",
             "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs.", ""]
    for k in order:
        c = {n: mean(v) for n, v in vals[k].items()}
        lines.append(f"== {k[:110]}")
        lines.append(f"   launches {len(durs[k])}, mean duration under the counter pass {mean(durs[k]) / 1e3:.1f} us")
        for n in sorted(c):
            lines.append(f"   {n:32s} {c[n]:16.4e}")
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                      "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_VMEM"):
                if n in c:
                    lines.append(f"   {n + ' / SQ_WAVE_CYCLES':48s} {100 * c[n] / wc:6.1f} %")
        gui = c.get("GRBM_GUI_ACTIVE")
        if gui and durs[k]:
            lines.append(f"   effective shader clock {gui / 8 / (mean(durs[k]) / 1e9) / 1e9:.2f} GHz")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                lines.append(f"   MFMA pipe busy {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui / 8 * 1024):.1f} % of SIMD cycles")
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            lines.append(f"   LDS bank-conflict cycles / LDS active cycles {100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.1f} %")
        lines.append("")
    open(out_txt, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
