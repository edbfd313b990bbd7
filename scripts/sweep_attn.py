import os, sys, subprocess
for rw in (4, 8):
    env = dict(os.environ, DG_ATTN_BWD_RW=str(rw))
    out = subprocess.run([sys.executable, "scripts/bench_kernels.py", "attn"], env=env, capture_output=True, text=True).stdout
    print("RW", rw, [l for l in out.splitlines() if l.startswith("attn")])
