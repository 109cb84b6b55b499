#!/usr/bin/env python3
"""usage: tools/pmc_table.py <label> <run_counter_collection.csv> [more csvs of the same run mode ...]

Per-kernel table of rocprofv3 --pmc counters (mean per dispatch) with the dispatch's own duration (End - Start timestamp of
the counter record) next to it and, where the counters are there, the derived stall attribution:

  dur_us            mean dispatch duration under the counters (tells whether dispatches still overlapped: compare with
                    profiles/r0x_cfg2_{serial,fif4}_kernel_stats.csv)
  waves/SIMD        4 x SQ_WAVE_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES-derived span): resident waves, time average
  VALU-busy         4 x SQ_ACTIVE_INST_VALU / (1024 x cycles)
  wait/wave         SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: share of a resident wave's time spent waiting for an instruction's operands
  vmem/wave         SQ_INST_CYCLES_VMEM / SQ_WAVE_CYCLES
  L2 hit            TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)

SQ_* cycle counters are in quad-cycles summed over the chip; cycles = dur_us x clock (2.4 GHz assumed for the ratios)."""
import collections
import csv
import sys

CLOCK_GHZ = 2.4


def short(name):
    name = name.replace("void ", "")
    cut = name.find("(")
    return (name if cut < 0 else name[:cut]).replace("msplat::", "")[:46]


def main():
    label, files = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.Counter())
    dur = collections.defaultdict(float)
    ndur = collections.Counter()
    seen = set()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            c = r["Counter_Name"]
            agg[k][c] += float(r["Counter_Value"])
            cnt[k][c] += 1
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
                ndur[k] += 1
    names = sorted({c for k in agg for c in agg[k]})
    print("# %s" % label)
    print("# counters (mean per dispatch): " + " ".join(names))
    hdr = "%-46s %6s %8s" % ("kernel", "calls", "dur_us") + "".join(" %14s" % n[-14:] for n in names)
    print(hdr)
    rows = sorted(agg, key=lambda k: -dur[k])
    for k in rows:
        if ndur[k] < 2 and dur[k] < 50.0:
            continue
        m = {c: agg[k][c] / max(cnt[k][c], 1) for c in names}
        print("%-46s %6d %8.1f" % (k, ndur[k], dur[k] / ndur[k]) + "".join(" %14.4g" % m[c] for c in names))
    print("# derived")
    print("%-46s %8s %10s %9s %9s %9s %9s %8s" % ("kernel", "dur_us", "waves/SIMD", "VALU-busy", "clk/VALU", "wait/wave", "vmem/wave", "L2 hit"))
    for k in rows:
        if ndur[k] < 2 and dur[k] < 50.0:
            continue
        m = {c: agg[k][c] / max(cnt[k][c], 1) for c in names}
        d_us = dur[k] / ndur[k]
        cyc = d_us * 1e3 * CLOCK_GHZ

        def f(v, fmt):
            return (fmt % v) if v is not None else "-"
        wps = 4.0 * m["SQ_WAVE_CYCLES"] / (1024.0 * cyc) if "SQ_WAVE_CYCLES" in m else None
        vb = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (1024.0 * cyc) if "SQ_ACTIVE_INST_VALU" in m else None
        cpv = 4.0 * m["SQ_ACTIVE_INST_VALU"] / max(m.get("SQ_INSTS_VALU", 0.0), 1.0) if "SQ_ACTIVE_INST_VALU" in m and "SQ_INSTS_VALU" in m else None
        ww = m["SQ_WAIT_INST_ANY"] / max(m["SQ_WAVE_CYCLES"], 1.0) if "SQ_WAIT_INST_ANY" in m and "SQ_WAVE_CYCLES" in m else None
        vw = m["SQ_INST_CYCLES_VMEM"] / max(m["SQ_WAVE_CYCLES"], 1.0) if "SQ_INST_CYCLES_VMEM" in m and "SQ_WAVE_CYCLES" in m else None
        hit = m["TCC_HIT_sum"] / max(m["TCC_HIT_sum"] + m["TCC_MISS_sum"], 1.0) if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m else None
        print("%-46s %8.1f %10s %9s %9s %9s %9s %8s" % (k, d_us, f(wps, "%.2f"), f(vb and 100 * vb, "%.1f%%"), f(cpv, "%.2f"),
                                                        f(ww, "%.3f"), f(vw, "%.3f"), f(hit, "%.3f")))


if __name__ == "__main__":
    main()
