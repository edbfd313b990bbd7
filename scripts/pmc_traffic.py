#!/usr/bin/env python
"""HBM traffic per launch of the edge-level kernels inside bench.py, from rocprofv3 PMC passes.

Run on the GPU box (separate passes for FETCH_SIZE and WRITE_SIZE, kernel trace only -- gpurun refuses other mixes):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_r -o r --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o w --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra
    python scripts/pmc_traffic.py <fetch csv> <write csv> c2 f32 256 <commit> [profiles/traffic.json] [edge_min_bytes]

FETCH_SIZE / WRITE_SIZE are KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the
bytes of a wide coalesced 16 B/lane read stream, so read bytes = 2 * FETCH_SIZE * 1024.  Launches are matched to
bench.py's kernel keys by name; row-GEMM / LayerNorm / weight-gradient kernels serve node-level launches too, so only
dispatches that move more than `edge_min_bytes` count as edge-level (the same split as DG_EDGE_ROWS in the library's
profiler; default: a quarter of one [B,N,N,128] tensor pass of the configuration, 8th argument).
With a 7th argument the record is merged into that JSON file (one record per (config, dtype, batch))."""
import collections
import csv
import json
import re
import sys

# bench.py key -> regex on the (mangled or demangled) kernel name
KEYS = [
    ("attn_half_fwd", r"attn_half(_f32)?_fwd"),
    ("attn_half_bwd", r"attn_half(_f32)?_bwd"),
    ("attn_bwd2", r"attn_bwd2_kernel"),
    ("attn_bwd", r"attn_bwd_kernel"),
    ("attn_fwd", r"attn_fwd_kernel"),
    ("row_gemm_e_k384", r"row_gemm_k384_kernel|row_gemm_h3_k384_kernel|row_gemm_bf16_kernel(<3, 1>|ILi3ELi1E)"),
    ("row_gemm_e_n384", r"row_gemm_n384_kernel|row_gemm_h3_kernel(<1, 1, (false|true), 6>|ILi1ELi1ELb[01]ELi6E)|row_gemm_bf16_kernel(<1, 3>|ILi1ELi3E)"),
    ("row_gemm_e128", r"row_gemm_h3_kernel(<1, 1, (false|true), 4>|ILi1ELi1ELb[01]ELi4E)|row_gemm_bf16_kernel(<1, 1>|ILi1ELi1E)"),
    ("ffn_f32", r"ffn_fused_f32_kernel"),
    ("ffn", r"ffn_fwd_bf16|ffn_bwd_dx_bf16"),
    ("ffn_wgrad", r"ffn_bwd_dw_bf16"),
    ("linear_wgrad_e_n384", r"wgrad_(stream_)?kernel(<(float, )?12, 4,|I(f)?Li12ELi4E)"),
    ("linear_wgrad_e_k384", r"wgrad_(stream_)?kernel(<(float, )?4, 12,|I(f)?Li4ELi12E)"),
    ("linear_wgrad_e128", r"wgrad_(stream_)?kernel(<(float, )?4, 4,|I(f)?Li4ELi4E)"),
    ("linear_wgrad", r"wgrad_kernel|wgrad_stream_kernel"),
    ("ln_bwd", r"ln_bwd_kernel"),
]


def per_dispatch(path, scale):
    out = {}
    for r in csv.DictReader(open(path)):
        key = r.get("Dispatch_Id") or r.get("Correlation_Id")
        out[key] = (r["Kernel_Name"], float(r["Counter_Value"]) * scale)
    return out


def main():
    rd = per_dispatch(sys.argv[1], 2.0 * 1024)       # gfx950: FETCH_SIZE counts 64 B per 128 B request
    wr = per_dispatch(sys.argv[2], 1024.0)
    meta = sys.argv[3:7] + [None] * 4
    edge_min = float(sys.argv[8]) if len(sys.argv) > 8 else 20e6
    # the two passes run the same deterministic launch sequence: pair the n-th launch of a kernel name in each
    seq_r, seq_w = collections.defaultdict(list), collections.defaultdict(list)
    for name, v in rd.values():
        seq_r[name].append(v)
    for name, v in wr.values():
        seq_w[name].append(v)
    acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for name, reads in seq_r.items():
        writes = seq_w.get(name, [])
        key = next((k for k, pat in KEYS if re.search(pat, name)), None)
        if key is None:
            continue
        for i, rbytes in enumerate(reads):
            wbytes = writes[i] if i < len(writes) else 0.0
            if rbytes + wbytes < edge_min:
                continue
            a = acc[key]
            a[0] += rbytes
            a[1] += wbytes
            a[2] += 1
    rec = {"config": meta[0], "dtype": meta[1], "batch": int(meta[2]) if meta[2] else None, "commit": meta[3],
           "measured": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes) on bench.py --steps 3 --warmup 1",
           "method": "read = 2*FETCH_SIZE*1024 (gfx950: FETCH_SIZE counts 64 B per 128 B request), write = WRITE_SIZE*1024; "
                     f"per-launch averages over the launch mix of the GAN step; dispatches under {edge_min / 1e6:.0f} MB (node-level) excluded",
           "kernels": {k: {"bytes_per_launch": (v[0] + v[1]) / v[2], "read_bytes_per_launch": v[0] / v[2],
                           "write_bytes_per_launch": v[1] / v[2], "launches": v[2]} for k, v in sorted(acc.items()) if v[2]}}
    if len(sys.argv) > 7:
        try:
            doc = json.load(open(sys.argv[7]))
            if "records" not in doc:
                doc = {"records": []}
        except Exception:
            doc = {"records": []}
        doc["records"] = [r for r in doc["records"]
                          if (r.get("config"), r.get("dtype"), r.get("batch")) != (rec["config"], rec["dtype"], rec["batch"])]
        doc["records"].append(rec)
        json.dump(doc, open(sys.argv[7], "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
