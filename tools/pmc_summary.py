#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes per kernel: MFMA-busy fraction, effective clock, LDS conflict share, HBM-side bytes.

  python tools/pmc_summary.py <dir with *_counter_collection.csv [+ *_kernel_trace.csv]> [more dirs ...] [--match substr]

busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs)      (MI355X_MICROARCH.md, cycle constants)
clock = GRBM_GUI_ACTIVE / 8 / duration;   FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B), unit KB."""
import collections
import csv
import glob
import sys


def main():
    args = [a for a in sys.argv[1:]]
    match = None
    if "--match" in args:
        i = args.index("--match")
        match = args[i + 1]
        del args[i:i + 2]
    split, labels = 1, None
    if "--split" in args:       # dispatches of a kernel fall into N equal consecutive groups (e.g. window methods run in turn)
        i = args.index("--split")
        labels = args[i + 1].split(",")
        split = len(labels)
        del args[i:i + 2]
    raw = collections.defaultdict(lambda: collections.defaultdict(list))
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> [values per dispatch]
    dur = collections.defaultdict(list)
    for d in args:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if match and match not in k:
                    continue
                dd = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9 if r.get("End_Timestamp") else None
                raw[k][r["Counter_Name"]].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"]), dd))
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if match and match not in k:
                    continue
                dur[(k, "trace")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    for k, byc in raw.items():
        for name, rows in byc.items():
            rows.sort()
            n = len(rows)
            for gi in range(split):
                part = rows[gi * n // split:(gi + 1) * n // split]
                kk = k if split == 1 else f"[{labels[gi]}] {k}"
                cnt[kk][name] += [v for _, v, _ in part]
                dur[(kk, name)] += [d for _, _, d in part if d is not None]
    for k, c in cnt.items():
        avg = lambda name: (sum(c[name]) / len(c[name])) if c.get(name) else None
        n = max(len(v) for v in c.values())
        line = [f"{k[:90]}", f"  dispatches {n}"]
        gui, mfma = avg("GRBM_GUI_ACTIVE"), avg("SQ_VALU_MFMA_BUSY_CYCLES")
        d = None
        for key in ("GRBM_GUI_ACTIVE", "trace"):
            if dur.get((k, key)):
                d = sum(dur[(k, key)]) / len(dur[(k, key)])
                break
        if d:
            line.append(f"  duration (this pass) {d * 1e3:.3f} ms")
        if gui and d:
            line.append(f"  effective clock {gui / 8 / d / 1e9:.2f} GHz")
        if gui and mfma:
            line.append(f"  MFMA busy {100 * mfma / 1024 / (gui / 8):.1f} %  (SQ_VALU_MFMA_BUSY_CYCLES {mfma:.4g}, GRBM_GUI_ACTIVE {gui:.4g})")
        if avg("SQ_BUSY_CYCLES") and gui:
            line.append(f"  SQ busy / GUI active {avg('SQ_BUSY_CYCLES') / gui:.3f}")
        if avg("SQ_LDS_BANK_CONFLICT") is not None and avg("SQ_LDS_IDX_ACTIVE"):
            line.append(f"  LDS bank-conflict cycles / LDS active {100 * avg('SQ_LDS_BANK_CONFLICT') / avg('SQ_LDS_IDX_ACTIVE'):.1f} %")
        for name in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_MFMA"):
            if avg(name) is not None:
                line.append(f"  {name} {avg(name):.4g}")
        if avg("FETCH_SIZE") is not None:
            line.append(f"  FETCH_SIZE x2 = {avg('FETCH_SIZE') * 2 * 1024 / 1e9:.3f} GB per launch")
        if avg("WRITE_SIZE") is not None:
            line.append(f"  WRITE_SIZE = {avg('WRITE_SIZE') * 1024 / 1e9:.3f} GB per launch")
        print("\n".join(line))


if __name__ == "__main__":
    main()
