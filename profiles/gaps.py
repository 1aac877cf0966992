#!/usr/bin/env python3
"""profiles/gaps.py <trace_kernel_trace.csv> [first_step] [nsteps] [first kernel of a step: k_check | k_slab_head | k_density_list ...] — the timeline of a few consecutive steps out of a
rocprofv3 --kernel-trace: per kernel its duration and the idle gap in front of it (end of the previous kernel on the
stream to its start), and per step the sum of kernel time, of gaps, and the step's span.  A step = everything from one
k_check launch to the next."""
import csv
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").replace("sph::", "")


def main():
    rows = []
    with open(sys.argv[1]) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    head = sys.argv[4] if len(sys.argv) > 4 else "k_check"
    starts = [k for k, r in enumerate(rows) if r[2].startswith(head)]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    tot_k = tot_g = tot_span = 0.0
    agg = {}
    for s in range(first, min(first + nsteps, len(starts) - 1)):
        a, b = starts[s], starts[s + 1]
        ksum = gsum = 0.0
        for k in range(a, b):
            st, en, nm = rows[k]
            gap = (st - rows[k - 1][1]) / 1e3 if k > 0 else 0.0
            dur = (en - st) / 1e3
            ksum += dur
            gsum += gap
            d = agg.setdefault(nm, [0, 0.0, 0.0])
            d[0] += 1; d[1] += dur; d[2] += gap
            if s < first + 2:
                print("  %-28s gap %6.2f us   run %7.2f us" % (nm, gap, dur))
        span = (rows[b][0] - rows[a][0]) / 1e3
        print("step %d: kernels %.2f us, gaps %.2f us, span %.2f us" % (s, ksum, gsum, span))
        tot_k += ksum; tot_g += gsum; tot_span += span
    # averages over a long stretch
    lo, hi = starts[len(starts) // 4], starts[3 * len(starts) // 4]
    n = 3 * len(starts) // 4 - len(starts) // 4
    agg = {}
    for k in range(lo, hi):
        st, en, nm = rows[k]
        d = agg.setdefault(nm, [0, 0.0, 0.0])
        d[0] += 1; d[1] += (en - st) / 1e3; d[2] += (st - rows[k - 1][1]) / 1e3
    print("\naverages over %d steps (middle half of the trace): span %.2f us per step" % (n, (rows[hi][0] - rows[lo][0]) / 1e3 / n))
    for nm, (cnt, dur, gap) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %-28s %6.2f launches/step   run %7.2f us/step   gap in front %6.2f us/step" % (nm, cnt / n, dur / n, gap / n))


if __name__ == "__main__":
    main()
