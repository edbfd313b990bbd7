#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 PMC passes (csv output).  Usage:
    python scripts/pmc_kernels.py pass1_counter_collection.csv [pass2 ...]
Prints, per kernel name (templates kept, arguments cut), the mean of every counter per dispatch and a few
derived ratios (MFMA busy share, LDS conflict share)."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = name.split("(")[0]
    return name.replace("void ", "").replace("dg::", "")[:70]


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if "Cijk" in k or "at::" in k or "elementwise" in k or "amd_rocclr" in k:
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    for k in sorted(acc):
        m = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
        n = max(cnt[k].values())
        line = [f"{k}  (n={n})"]
        for c in sorted(m):
            line.append(f"    {c:34s} {m[c]:16.0f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CU_CYCLES" in m and m["SQ_BUSY_CU_CYCLES"]:
            # SQ_VALU_MFMA_BUSY_CYCLES sums the four SIMDs' matrix pipes, SQ_BUSY_CU_CYCLES counts per CU: / 4
            line.append(f"    -> MFMA pipe busy share (MFMA busy / (4 SIMDs x CU busy cycles))  "
                        f"{m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * m['SQ_BUSY_CU_CYCLES']):.3f}")
        if "SQ_LDS_BANK_CONFLICT" in m and "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"]:
            line.append(f"    -> LDS bank-conflict cycles / LDS active  {m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:.3f}")
        print("\n".join(line))


if __name__ == "__main__":
    main()
