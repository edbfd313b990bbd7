#!/usr/bin/env python
"""HBM traffic of the attention kernels inside bench.py from rocprofv3 PMC passes.

Run on the GPU box (separate passes for FETCH_SIZE and WRITE_SIZE, kernel trace only):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_r -o r --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o w --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python scripts/pmc_traffic.py /tmp/pmc_r/r_counter_collection.csv /tmp/pmc_w/w_counter_collection.csv

FETCH_SIZE / WRITE_SIZE are KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports
half the bytes of a wide coalesced 16 B/lane read stream, so read bytes = 2 * FETCH_SIZE * 1024.
"""
import collections
import csv
import json
import sys


def per_kernel(path):
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        for key in ("attn_fwd_kernel", "attn_bwd2_kernel", "attn_bwd_kernel"):
            if key in name:
                tot[key] += float(r["Counter_Value"])
                cnt[key] += 1
                break
    return tot, cnt


def main():
    """pmc_traffic.py <fetch csv> <write csv> [config dtype batch commit]"""
    rd, rc = per_kernel(sys.argv[1])
    wr, wc = per_kernel(sys.argv[2])
    meta = sys.argv[3:7] + [None] * 4
    out = {"config": meta[0], "dtype": meta[1], "batch": int(meta[2]) if meta[2] else None, "commit": meta[3],
           "measured": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes) on bench.py --steps 3 --warmup 1",
           "method": "read = 2*FETCH_SIZE*1024 (gfx950: FETCH_SIZE counts 64 B per 128 B request), write = WRITE_SIZE*1024",
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on bench.py --steps 3 --warmup 1; "
                   "read = 2*FETCH_SIZE*1024 (gfx950 correction), write = WRITE_SIZE*1024; averages per launch over the "
                   "launch mix of the GAN step"}
    for key, tag in (("attn_fwd_kernel", "attn_fwd"), ("attn_bwd_kernel", "attn_bwd"), ("attn_bwd2_kernel", "attn_bwd2")):
        if not rc[key] or not wc[key]:
            continue
        read_b = 2.0 * rd[key] * 1024 / rc[key]
        write_b = wr[key] * 1024 / wc[key]
        out[f"{tag}_bytes_per_launch"] = read_b + write_b
        out[f"{tag}_read_bytes_per_launch"] = read_b
        out[f"{tag}_write_bytes_per_launch"] = write_b
        out[f"{tag}_launches"] = rc[key]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
