#!/usr/bin/env python3
"""Aggregate two rocprofv3 PMC passes of bench.py (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) into
profiles/r0x_hbm_traffic_cfg3.json: HBM-side bytes per kernel family and per launch.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- \\
      python3 bench.py --workload cfg3 --steps 1 --warmup 0 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE ... -d gpurun_out/pmc_write ...
  python tools/hbm_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_hbm_traffic_cfg3.json

FETCH_SIZE is doubled (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section); counter unit KB."""
import collections
import csv
import json
import sys

FAMILIES = [
    ("conv3d halo-resident (conv_halo*_kernel)", lambda n: "conv_halo" in n),
    ("conv3d tap-staged (A_CONV of gemm.hip + gemm2.hip)", lambda n: "gemm2_kernel<1" in n or "gemm_kernel<1" in n),
    ("dense gemm", lambda n: "gemm2_kernel<0" in n or "gemm_kernel<0" in n or "gemm4_kernel" in n),
    ("gn_partial", lambda n: "gn_partial" in n), ("gn_apply", lambda n: "gn_apply" in n),
    ("gn_finalize", lambda n: "gn_finalize" in n), ("attn_d512", lambda n: "attn_d512" in n),
    ("window_attn", lambda n: "flash_attn" in n), ("im2col3d", lambda n: "im2col" in n),
    ("qkv_prep", lambda n: "qkv_prep" in n),
    ("copies (carry / cat)", lambda n: "copyBuffer" in n or "CatArray" in n or "direct_copy" in n),
]


def load(d, cname):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(d + "/p_counter_collection.csv")):
        if r["Counter_Name"] != cname:
            continue
        fam = next((k for k, f in FAMILIES if f(r["Kernel_Name"])), "other")
        agg[fam][0] += 1
        agg[fam][1] += float(r["Counter_Value"])
    return agg


def main():
    fdir, wdir, out = sys.argv[1:4]
    f, w = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    res = {"source": __doc__.strip().splitlines()[0] + " One bench step (cfg3: first temporal batch, 21 frames); FETCH_SIZE x2, KB -> bytes; "
           "per launch = total / launches.", "kernels": {}}
    tot = 0.0
    for k in sorted(set(f) | set(w), key=lambda k: -(f[k][1] * 2 + w[k][1])):
        n = f[k][0] or w[k][0]
        fb, wb = f[k][1] * 1024 * 2, w[k][1] * 1024
        res["kernels"][k] = {"launches": n, "fetch_bytes_per_launch": fb / n, "write_bytes_per_launch": wb / n,
                             "hbm_bytes_per_launch": (fb + wb) / n, "total_GB": (fb + wb) / 1e9}
        tot += fb + wb
        print(f"{k:55s} n={n:5d} fetch={fb / 1e9:8.1f} GB write={wb / 1e9:8.1f} GB")
    res["total_GB_per_step"] = tot / 1e9
    json.dump(res, open(out, "w"), indent=1)
    print("total GB per step:", tot / 1e9)


if __name__ == "__main__":
    main()
