#!/usr/bin/env python3
"""usage: tools/pcsamp_summary.py <dir with rocprofv3 --pc-sampling-beta-enabled csv output>

Summarises PC samples per kernel: share of samples by instruction class / stall reason (stochastic sampling) or by the sampled
instruction's mnemonic class (host-trap sampling).  PC sampling does not serialise dispatches, so it describes frames in flight.
The column names differ between rocprofiler-sdk versions: everything is looked up defensively and the header is printed."""
import collections
import csv
import glob
import os
import re
import sys


def klass(instr):
    op = instr.strip().split(" ")[0] if instr else "?"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("v_pk_") or op.startswith("v_exp") or op.startswith("v_fma") or op.startswith("v_mul") or op.startswith("v_add") or op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_"):
        return "salu/branch"
    return op[:12]


def main():
    d = sys.argv[1]
    disp = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            disp[r.get("Dispatch_Id")] = re.sub(r"\(.*", "", r.get("Kernel_Name", "?").replace("void ", "")).replace("msplat::", "")[:44]
    files = glob.glob(os.path.join(d, "**", "*pc_sampling*.csv"), recursive=True)
    if not files:
        print("no pc sampling csv under", d)
        return 1
    for f in files:
        rd = csv.DictReader(open(f))
        print("#", os.path.basename(f), "columns:", rd.fieldnames)
        per = collections.defaultdict(collections.Counter)
        stall = collections.defaultdict(collections.Counter)
        itype = collections.defaultdict(collections.Counter)
        issued = collections.defaultdict(collections.Counter)
        n = 0
        for r in rd:
            n += 1
            k = disp.get(r.get("Dispatch_Id"), "?")
            per[k][klass(r.get("Instruction", ""))] += 1
            for col, tgt in (("Stall_Reason", stall), ("Instruction_Type", itype), ("Wave_Issued_Instruction", issued)):
                if col in r and r[col] != "":
                    tgt[k][r[col]] += 1
        print("# samples:", n)
        for title, tab in (("sampled instruction class", per), ("Wave_Issued_Instruction", issued), ("Instruction_Type", itype), ("Stall_Reason", stall)):
            if not tab:
                continue
            print("## share of a kernel's samples by", title)
            for k in sorted(tab, key=lambda k: -sum(tab[k].values())):
                tot = sum(tab[k].values())
                if tot < 50:
                    continue
                print("%-44s n=%-8d " % (k, tot) + "  ".join("%s %.1f%%" % (c, 100.0 * v / tot) for c, v in tab[k].most_common(9)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
