#!/usr/bin/env python3
"""profiles/summarize.py <gpurun_out/prof_TAG> [out.md] — condense the rocprofv3 CSVs written by
profiles/collect.sh into one markdown summary (per-kernel time from --kernel-trace --stats, PMC
counters averaged per dispatch, HBM traffic corrected as MI355X_MICROARCH.md §HBM prescribes:
FETCH_SIZE x2 for wide coalesced reads on gfx950, WRITE_SIZE as is; both are reported in KiB)."""
import csv
import collections
import os
import sys


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("sph::", "")
    return n


def kernel_stats(path):
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((short(r["Name"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]),
                         float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    return rows


def pmc(path):
    """mean counter value per dispatch for each (kernel, counter)."""
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
    out = collections.defaultdict(dict)
    for (kn, cn), (s, n) in acc.items():
        out[kn][cn] = s / n
    return out


def write_traffic(merged, path, workload):
    """per-launch HBM traffic of the step kernels in bench.py's naming (read side doubled, see above) -> traffic.json[workload]"""
    names = {"k_kick_drift<false>": "kick_drift", "k_key_hist<false>": "key_hist", "k_reorder": "reorder",
             "k_build_list": "build_list", "k_density_list<1, 0>": "density_eos", "k_density_list<1, 0, true>": "density_eos",
             "k_density_list<1, 0, false>": "density_eos_plain", "k_force_list<2, 0>": "force_kick", "k_check": "check",
             "k_scan_reduce": "scan_reduce", "k_scan_apply": "scan_apply", "k_rebuild<0>": "rebuild"}
    traffic, valu = {}, {}
    for kn, bn in names.items():
        cs = merged.get(kn, {})
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            traffic[bn] = int(2 * cs["FETCH_SIZE"] * 1024 + cs["WRITE_SIZE"] * 1024)
        if "SQ_INSTS_VALU" in cs:      # wave-instructions per launch (all SIMDs); the transcendental ones issue at a quarter of the rate
            valu[bn] = {"insts": int(cs["SQ_INSTS_VALU"]), "trans": int(cs.get("SQ_INSTS_VALU_TRANS", 0))}
    if valu:
        traffic["_valu"] = valu
    if path and traffic:
        import json
        try:
            allt = json.load(open(path))
        except Exception:
            allt = {}
        allt[workload] = traffic
        allt["_note"] = ("HBM-side bytes per launch from rocprofv3 PMC passes (profiles/collect.sh): 2 x FETCH_SIZE KiB + "
                         "WRITE_SIZE KiB; the x2 is the gfx950 correction for wide coalesced reads "
                         "(MI355X_MICROARCH.md, HBM) and is an upper bound where reads are 8 B/lane; "
                         "Infinity-Cache hits are counted")
        json.dump(allt, open(path, "w"), indent=1)


def main():
    if sys.argv[1] == "--trace-only":      # <dir> <out.md> <the profiled command, verbatim>: another program than bench.py (the C slab host)
        d, out, cmd = sys.argv[2:5]
        lines = ["# rocprofv3 summary: %s" % os.path.basename(d.rstrip("/")), "", "## per-kernel time (`rocprofv3 --kernel-trace --stats -- %s`)" % cmd, "",
                 "| kernel | calls | avg us | % | min us | max us |", "|---|---|---|---|---|---|"]
        for n, c, a, p, mn, mx in kernel_stats(os.path.join(d, "trace", "trace_kernel_stats.csv")):
            lines.append("| %s | %d | %.2f | %.2f | %.2f | %.2f |" % (n, c, a, p, mn, mx))
        open(out, "w").write("\n".join(lines) + "\n")
        print("\n".join(lines))
        return
    if sys.argv[1] == "--from-summary":    # <summary.md> <traffic.json> <workload>: the traffic entry again, from a committed summary's counter table
        md, path, workload = sys.argv[2:5]
        rows, hdr = {}, None
        for ln in open(md):
            cells = [c.strip() for c in ln.strip().strip("|").split("|")]
            if ln.startswith("| kernel | ") and "SQ_INSTS_VALU" in ln:
                hdr = cells
            elif hdr and ln.startswith("| k_") and len(cells) == len(hdr):
                rows[cells[0]] = {h: float(v) for h, v in zip(hdr[1:], cells[1:]) if v}
            elif hdr and not ln.startswith("|"):
                hdr = None
        write_traffic(rows, path, workload)
        return
    d = sys.argv[1]
    lines = ["# rocprofv3 summary: %s" % os.path.basename(d.rstrip("/")), ""]
    ks = os.path.join(d, "trace", "trace_kernel_stats.csv")
    if os.path.exists(ks):
        desc = sys.argv[5] if len(sys.argv) > 5 else "--steps 300 --warmup 100"
        lines += ["## per-kernel time (`rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu --no-also %s`)" % desc, "",
                  "| kernel | calls | avg us | % | min us | max us |", "|---|---|---|---|---|---|"]
        for n, c, a, p, mn, mx in kernel_stats(ks):
            lines.append("| %s | %d | %.2f | %.2f | %.2f | %.2f |" % (n, c, a, p, mn, mx))
        lines.append("")
    merged = collections.defaultdict(dict)
    for sub in sorted(os.listdir(d)):
        f = os.path.join(d, sub, "pmc_counter_collection.csv")
        if os.path.exists(f):
            for kn, cs in pmc(f).items():
                merged[kn].update(cs)
    if merged:
        counters = sorted({c for cs in merged.values() for c in cs})
        lines += ["## PMC counters, mean per dispatch (separate `--pmc` passes: 10 steps after the warm-up of the regime; with a long warm-up only those dispatches are counted, `--kernel-iteration-range`)", "",
                  "| kernel | " + " | ".join(counters) + " |", "|---|" + "---|" * len(counters)]
        for kn in sorted(merged):
            lines.append("| %s | " % kn + " | ".join("%.4g" % merged[kn][c] if c in merged[kn] else "" for c in counters) + " |")
        lines.append("")
        lines += ["## HBM traffic per launch (FETCH_SIZE, WRITE_SIZE in KiB; gfx950: wide coalesced reads are tallied at 1/2,",
                  "so the read side is doubled as MI355X_MICROARCH.md §HBM prescribes — an upper bound where reads are narrow)", "",
                  "| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | read MB (x2) | write MB | total MB |", "|---|---|---|---|---|---|"]
        for kn in sorted(merged):
            cs = merged[kn]
            if "FETCH_SIZE" in cs or "WRITE_SIZE" in cs:
                fs, ws = cs.get("FETCH_SIZE", 0.0), cs.get("WRITE_SIZE", 0.0)
                rd, wr = 2 * fs * 1024 / 1e6, ws * 1024 / 1e6
                lines.append("| %s | %.0f | %.0f | %.2f | %.2f | %.2f |" % (kn, fs, ws, rd, wr, rd + wr))
        lines.append("")
    if len(sys.argv) > 3:
        write_traffic(merged, sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "cfg2")
    text = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
