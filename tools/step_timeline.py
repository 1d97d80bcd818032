#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV: start offset, duration, kernel, grid.

    python tools/step_timeline.py <..._kernel_trace.csv> [step-index-from-the-middle]
A step starts at each neg_draw_kernel launch (the first kernel bench.py queues per step)."""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "neg_draw" in r["Kernel_Name"]]
    k = len(idx) // 2 + (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    s, e = idx[k], idx[k + 1]
    t0 = int(rows[s]["Start_Timestamp"])
    busy = 0
    for r in rows[s:e]:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += en - st
        print("%8.1f %7.1f  %-72s grid=%s vgpr=%s lds=%s" % ((st - t0) / 1e3, (en - st) / 1e3, r["Kernel_Name"][:72], r["Grid_Size_X"],
                                                              r["VGPR_Count"], r["LDS_Block_Size"]))
    print("step: %.1f us wall, %.1f us summed kernel time, %d launches" % ((int(rows[e]["Start_Timestamp"]) - t0) / 1e3, busy / 1e3, e - s))


if __name__ == "__main__":
    main()
