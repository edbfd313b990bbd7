"""GPU tests at the sizes of BASELINE configs[3] (B = 2048 per GPU, N = 45) and configs[4] (N = 90, E = 10,
L = 8, B = 64 per GPU), through size-independent properties (the fp64 oracle does not finish at these sizes),
and of the data-parallel path with the HIP modules: two processes sharing ONE GPU (gloo) must reproduce the
single-process full-batch step (what replaces reference train.py:220-223's nn.DataParallel)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _nets(vertexes, edges, depth, seed=0):
    from druggen_amd.model import Discriminator, Generator
    torch.manual_seed(seed)
    kw = dict(dim=128, depth=depth, heads=8, mlp_ratio=3)
    G = Generator("relu", vertexes, edges, 13, 0.0, **kw).cuda()
    D = Discriminator("relu", vertexes, edges, 13, 0.0, **kw).cuda()
    return G, D


def _batch(B, N, E, seed):
    from druggen_amd import synth
    a, x, _, _ = synth.molecule_batch(B, N, E, 13, seed=seed)
    return torch.from_numpy(a).cuda(), torch.from_numpy(x).cuda()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("name,B,N,E,L", [("configs[3] B=2048", 2048, 45, 5, 4), ("configs[4] N=90 L=8", 64, 90, 10, 8)])
def test_full_size_shard_consistency_and_permutation_equivariance(name, B, N, E, L, dtype):
    """(i) per-molecule outputs do not depend on the rest of the batch: the logits of the full batch equal the
    concatenation of two half-batch shards and the mean-loss gradient the average of the shard gradients (the
    property the data-parallel sharding relies on); (ii) no positional encoding: relabelling the atoms permutes
    the generator's outputs.  fp32: 1e-5 / 1e-4; bf16 activations: rows are still computed independently, so
    (i) holds to the same tolerances; (ii) reorders sums, hence 3e-2."""
    from druggen_amd import functional as dgf
    G, D = _nets(N, E, L)
    a, x = _batch(B, N, E, seed=5)
    with dgf.activations(dtype):
        full = D(a, x)
        (-full.mean()).backward()
        g_full = {k: p.grad.clone() for k, p in D.named_parameters() if p.grad is not None}
        D.zero_grad(set_to_none=True)
        parts = []
        h = B // 2
        for sl in (slice(0, h), slice(h, B)):
            out = D(a[sl], x[sl])
            (-out.mean() / 2).backward()
            parts.append(out.detach())
        assert torch.isfinite(full).all()
        assert (torch.cat(parts) - full.detach()).abs().max().item() <= 1e-5 * max(1.0, full.abs().max().item())
        tot = torch.sqrt(sum((g ** 2).sum() for g in g_full.values())).item()
        gtol = 1e-4 if dtype == "f32" else 2e-3       # fp32 accumulation order of the row split differs between B and B/2
        for k, p in D.named_parameters():
            if p.grad is None:
                assert k not in g_full
                continue
            assert (p.grad - g_full[k]).norm().item() <= gtol * max(g_full[k].norm().item(), tot / len(g_full) ** 0.5), k
        D.zero_grad(set_to_none=True)
        del g_full, parts, full
        nb = min(B, 128)
        perm = torch.randperm(N, device="cuda")
        with torch.no_grad():
            _, _, ns, es = G(a[:nb], x[:nb])
            _, _, ns_p, es_p = G(a[:nb][:, perm][:, :, perm], x[:nb][:, perm])
        tol = 1e-4 if dtype == "f32" else 3e-2
        scale = max(1.0, es.abs().max().item())
        assert (ns[:, perm] - ns_p).abs().max().item() < tol * scale
        assert (es[:, perm][:, :, perm] - es_p).abs().max().item() < tol * scale


@pytest.mark.parametrize("name,B,N,E,L", [("configs[4] N=90 L=8 B=64", 64, 90, 10, 8)])
def test_full_size_gan_step_runs_and_is_reproducible(name, B, N, E, L):
    """One whole WGAN-GP iteration at configs[4]'s per-GPU size (attention kernels on the JPL = 12 geometry, 8
    encoder blocks): finite losses and bit-identical results when repeated from the same state (every reduction in
    the library runs in a fixed order, no atomics)."""
    from druggen_amd.trainer import GANStep
    runs = []
    for _ in range(2):
        G, D = _nets(N, E, L, seed=3)
        a, x = _batch(B, N, E, seed=7)
        da, dx = _batch(B, N, E, seed=8)
        eps = (torch.rand(B, 1, 1, 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)),
               torch.rand(B, 1, 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2)))
        st = GANStep(G, D, lambda_gp=10.0)
        d_loss, g_loss = st.step(da, dx, a, x, eps=eps)
        assert torch.isfinite(d_loss) and torch.isfinite(g_loss)
        runs.append((d_loss.item(), g_loss.item(), torch.cat([p.detach().reshape(-1) for p in D.parameters()]).clone()))
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1]
    assert torch.equal(runs[0][2], runs[1][2])


@pytest.mark.parametrize("name,dtype,B", [("configs[2] bf16", "bf16", 2048), ("configs[3] f32 low-memory", "f32", 2048)])
def test_full_gan_step_at_batch_2048_is_finite_and_reproducible(name, dtype, B):
    """The WHOLE iteration -- both losses, the gradient penalty's second order, FlatAdamW -- at the per-GPU sizes of BASELINE
    configs[2] (bf16 activations) and configs[3] (float32: the low-memory step with the shared generator graph), through
    GANStep.step as bench.py runs it: finite losses, and bit-identical parameters when repeated from the same state."""
    from druggen_amd import functional as dgf
    from druggen_amd.trainer import GANStep
    a, x = _batch(B, 45, 5, seed=11)
    da, dx = _batch(B, 45, 5, seed=12)
    eps = (torch.rand(B, 1, 1, 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)),
           torch.rand(B, 1, 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2)))
    runs = []
    with dgf.activations(dtype):
        for _ in range(2):
            G, D = _nets(45, 5, 4, seed=5)
            st = GANStep(G, D, lambda_gp=10.0)
            if dtype == "f32":
                assert st._low_memory(a)
            d_loss, g_loss = st.step(da, dx, a, x, eps=eps)
            assert torch.isfinite(d_loss) and torch.isfinite(g_loss)
            runs.append((float(d_loss), float(g_loss), torch.cat([p.detach().reshape(-1) for p in list(G.parameters()) + list(D.parameters())]).clone()))
            del st, G, D
            torch.cuda.empty_cache()
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1]
    assert torch.equal(runs[0][2], runs[1][2])


def test_low_memory_step_equals_fast_step_at_batch_512():
    """memory="low" (the three Discriminator terms differentiated one after the other, the generator's graph shared) against the
    default step at B = 512, the headline model: same losses, D gradient bucket within 1e-3, parameters after the step within
    5 % of their movement."""
    from druggen_amd.trainer import GANStep
    B = 512
    a, x = _batch(B, 45, 5, seed=21)
    da, dx = _batch(B, 45, 5, seed=22)
    eps = (torch.rand(B, 1, 1, 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)),
           torch.rand(B, 1, 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4)))
    outs = []
    for mode in ("fast", "low"):
        G, D = _nets(45, 5, 4, seed=6)
        start = torch.cat([p.detach().reshape(-1) for p in list(G.parameters()) + list(D.parameters())]).clone()
        st = GANStep(G, D, g_lr=1e-3, d_lr=1e-3, lambda_gp=10.0, memory=mode)
        losses = st.step(da, dx, a, x, eps=eps)
        outs.append((torch.cat([p.detach().reshape(-1) for p in list(G.parameters()) + list(D.parameters())]).clone(),
                     st.d_optimizer.flat_grad.clone(), [float(v) for v in losses]))
        del st, G, D
        torch.cuda.empty_cache()
    assert float((outs[0][1] - outs[1][1]).norm() / outs[0][1].norm()) < 1e-3
    for u, v in zip(outs[0][2], outs[1][2]):
        assert abs(u - v) <= 1e-3 * max(1.0, abs(u))
    assert float((outs[0][0] - outs[1][0]).norm()) <= 0.05 * float((outs[0][0] - start).norm())


def test_bench_n8_gloo_line_on_one_gpu(tmp_path):
    """N = 8 readiness without the hardware: bench.py exactly as the driver's scaling run launches it -- torch.distributed.run,
    eight ranks, `--gpus 8` -- with the `gloo` backend and every rank on cuda:0: ONE parseable line with n_gpus = 8, identical
    replicas and the all-reduce record populated, so that the first real SCALE run cannot die on rendezvous or device-index
    plumbing (under `nccl` rank r takes device r: bench.py `dev_index`)."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DG_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-extra", "--eager", "--batch", "8", "--vertexes", "9", "--depth", "1"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["config"]["global_batch"] == 64 and out["config"]["parallelism"] == "dp8"
    assert out["replicas_identical"] is True
    ar = out["allreduce"]
    assert len(ar["per_rank_ms_per_step"]) == 8 and ar["collectives_per_step"] == 2 and ar["max_ms_per_step"] > 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import cases, harness
from druggen_amd.model import Discriminator, Generator
from druggen_amd.trainer import GANStep, broadcast_parameters
rank, world, out = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1]
torch.cuda.set_device(0)
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
case = cases.CASES["c1_b4"]
cfg = cases.net_config(case)
gp, dp = cases.build_params(case)
args = (cfg.act, cfg.vertexes, cfg.edges, cfg.nodes, cfg.dropout)
kw = dict(dim=cfg.dim, depth=cfg.depth, heads=cfg.heads, mlp_ratio=cfg.mlp_ratio)
G, D = Generator(*args, **kw), Discriminator(*args, **kw)
G.load_state_dict({k: torch.from_numpy(v) for k, v in gp.items()})
D.load_state_dict({k: torch.from_numpy(v) for k, v in dp.items()})
G, D = G.cuda(), D.cuda()
if rank != 0:                      # the initial broadcast must repair this
    with torch.no_grad():
        for p in list(G.parameters()) + list(D.parameters()):
            p.add_(0.25)
broadcast_parameters(G)
broadcast_parameters(D)
inp = harness.torch_inputs(case, torch.float32, "cuda")
per = case["batch"] // world
sl = slice(rank * per, (rank + 1) * per)
st = GANStep(G, D, g_lr=1e-3, d_lr=1e-3, lambda_gp=case["lambda_gp"])       # FlatAdamW: the flat bucket IS the all-reduce buffer
grads = []
shard = (inp["disc_edge"][sl].contiguous(), inp["disc_node"][sl].contiguous(), inp["gen_edge"][sl].contiguous(),
         inp["gen_node"][sl].contiguous())
eps = (inp["eps_edge"][sl].contiguous(), inp["eps_node"][sl].contiguous())
graphed = os.environ.get("DG_TEST_GRAPHED") == "1"
tag = "g" if graphed else ""
if graphed:      # the data-parallel step as three hipGraphs cut at the two all-reduces; its one warm-up step is iteration 0
    from druggen_amd.trainer import GraphedGANStep
    gst = GraphedGANStep(st, *shard, warmup=1, eps=eps)
    assert gst.segments is not None and len(gst.segments[0]) == 3 and len(gst.segments[1]) == 2
    grads = [st.d_optimizer.flat_grad.detach().cpu().clone(), st.g_optimizer.flat_grad.detach().cpu().clone()]
    for it in range(1, 3):
        gst.step(*[t.clone() for t in shard], eps=eps)
else:
    for it in range(3):
        if os.environ.get("DG_TEST_SLOW_RANK") == str(rank):      # uneven arrival at the all-reduces
            import time
            time.sleep(0.3 * (it + 1))
        st.step(*shard, eps=eps)
        if it == 0:      # the rank-averaged gradient buckets of the first iteration (same weights in every run)
            grads = [st.d_optimizer.flat_grad.detach().cpu().clone(), st.g_optimizer.flat_grad.detach().cpu().clone()]
torch.cuda.synchronize()
torch.save({"G": [p.detach().cpu() for p in G.parameters()], "D": [p.detach().cpu() for p in D.parameters()], "grads": grads},
           os.path.join(out, f"w{world}{tag}_r{rank}.pt"))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
"""


def _launch(world, out_dir, port, **extra_env):
    script = os.path.join(out_dir, "worker.py")
    with open(script, "w") as f:
        f.write(f"ROOT = {ROOT!r}\n" + _WORKER)
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
        procs.append(subprocess.Popen([sys.executable, script, out_dir], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out.decode()[-3000:]


def test_two_process_hip_data_parallel_step_equals_single_process(tmp_path):
    """Two processes on cuda:0 (gloo), each running the HIP `GANStep` on its half of the batch -- sharded inputs
    and eps, `broadcast_parameters`, FlatAdamW's flat bucket as the all-reduce buffer with the Discriminator's dead
    parameters excluded -- end on the same parameters as one process on the full batch (1e-5 of the parameter
    movement; fp32 sums over the batch are grouped differently), and the two ranks stay bit-identical."""
    out = str(tmp_path)
    _launch(2, out, _free_port())
    _launch(1, out, _free_port())
    r0, r1, full = (torch.load(os.path.join(out, f)) for f in ("w2_r0.pt", "w2_r1.pt", "w1_r0.pt"))
    for a, b in zip(r0["G"] + r0["D"], r1["G"] + r1["D"]):
        assert torch.equal(a, b), "ranks diverged"
    import cases
    gp, dp = cases.build_params(cases.CASES["c1_b4"])
    start = [torch.from_numpy(v) for v in list(gp.values()) + list(dp.values())]
    # the all-reduced (averaged) gradient buckets of the first iteration equal the full-batch gradients.  fp32: the
    # weight gradients sum the rows in a different grouping (2 + 2 molecules vs 4), and the WGAN loss subtracts the
    # real from the fake term, so the bucket agrees to ~1e-4 of its norm (measured 1.3e-4), not to 1e-6; a wrong
    # shard / eps / averaging would show up as O(1).  The exact (fp64, 1e-9) version of this test is the gloo test
    # on the oracle nets in tests/test_host.py.
    for gb, gf in zip(r0["grads"], full["grads"]):
        assert gb.shape == gf.shape and float((gb - gf).norm() / gf.norm()) < 1e-3
    # ... and so do the parameters after three AdamW steps (an Adam step is ~ lr sign(g): elements whose gradient is
    # rounding noise may move differently, hence the looser bound on the parameter movement)
    moved = torch.sqrt(sum(((a - s) ** 2).sum() for a, s in zip(full["G"] + full["D"], start)))
    diff = torch.sqrt(sum(((a - b) ** 2).sum() for a, b in zip(r0["G"] + r0["D"], full["G"] + full["D"])))
    assert moved > 0 and diff <= 5e-2 * moved, (float(diff), float(moved))
    # dead discriminator parameters were never touched on any rank (not even by weight decay)
    names = list(dp.keys())
    n_g = len(gp)
    untouched = [k for k, a, s in zip(names, r0["D"], start[n_g:]) if torch.equal(a, s)]
    assert len(untouched) == 10 and all(".attn.out_e." in k or ".ln4." in k or ".mlp2." in k or ".ln6." in k for k in untouched)


def test_two_ranks_replaying_the_step_as_three_graphs_follow_the_eager_ranks(tmp_path):
    """Data parallelism without ~750 launches per step from every rank's Python: `GraphedGANStep` under a
    process group captures the iteration as THREE hipGraphs cut at the two gradient all-reduces -- the shared generator
    forward's autograd graph is built in the first capture and consumed in the second -- and replays graph, all-reduce,
    graph, all-reduce, graph.  Two ranks on cuda:0 (gloo): the ranks stay bit-identical, and after one eager + two replayed
    iterations on fixed interpolation weights they hold the parameters of the eagerly stepping pair."""
    out = str(tmp_path)
    _launch(2, out, _free_port(), DG_TEST_GRAPHED="1")
    _launch(2, out, _free_port())
    g0, g1, e0 = (torch.load(os.path.join(out, f)) for f in ("w2g_r0.pt", "w2g_r1.pt", "w2_r0.pt"))
    for a, b in zip(g0["G"] + g0["D"], g1["G"] + g1["D"]):
        assert torch.equal(a, b), "ranks diverged"
    for a, b in zip(g0["grads"], e0["grads"]):
        assert torch.equal(a, b)
    import cases
    gp, dp = cases.build_params(cases.CASES["c1_b4"])
    start = [torch.from_numpy(v) for v in list(gp.values()) + list(dp.values())]
    moved = torch.sqrt(sum(((a - s) ** 2).sum() for a, s in zip(e0["G"] + e0["D"], start)))
    diff = torch.sqrt(sum(((a - b) ** 2).sum() for a, b in zip(g0["G"] + g0["D"], e0["G"] + e0["D"])))
    assert moved > 0 and diff <= 1e-6 * moved, (float(diff), float(moved))


def test_four_ranks_with_a_late_rank_stay_bit_identical(tmp_path):
    """Four processes on cuda:0 (gloo), one molecule each, rank 2 arriving 0.3 - 0.9 s late at every iteration's all-reduces:
    the two flat-bucket collectives per step are the only synchronisation points, so a late rank may delay the others but
    never changes what they compute -- all four ranks end on bit-identical parameters, equal to the two-rank run's to the
    rounding of another summation order."""
    out = str(tmp_path)
    _launch(4, out, _free_port(), DG_TEST_SLOW_RANK="2")
    ranks = [torch.load(os.path.join(out, f"w4_r{r}.pt")) for r in range(4)]
    for other in ranks[1:]:
        for a, b in zip(ranks[0]["G"] + ranks[0]["D"], other["G"] + other["D"]):
            assert torch.equal(a, b), "ranks diverged"
        for a, b in zip(ranks[0]["grads"], other["grads"]):
            assert torch.equal(a, b)
    _launch(2, out, _free_port())
    two = torch.load(os.path.join(out, "w2_r0.pt"))
    for gb, gf in zip(ranks[0]["grads"], two["grads"]):
        assert float((gb - gf).norm() / gf.norm()) < 1e-3


_NCCL_WORKER = r"""
import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)      # exactly bench.py's call
t = torch.arange(8, dtype=torch.float32, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.AVG)                                   # GradBucket's reduction on RCCL
dist.broadcast(t, src=0)                                                   # broadcast_parameters
dist.barrier()
u = torch.tensor([1.5], device=dev, dtype=torch.float64)
dist.all_reduce(u, op=dist.ReduceOp.MAX)                                   # bench.py's max-over-ranks timing
torch.cuda.synchronize()
assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32)) and float(u) == 1.5
dist.destroy_process_group()
print("rccl ok")
"""


def test_rccl_calls_of_the_multi_gpu_path_run_on_this_build(tmp_path):
    """The N > 1 path cannot run on a one-GPU box, but every RCCL call it makes can: a one-rank `nccl` group with
    bench.py's init arguments, GradBucket's AVG all-reduce, broadcast_parameters' broadcast, the barrier and the MAX
    reduction of the timing.  Catches a missing / mismatched RCCL, an unsupported reduce op or init signature."""
    script = os.path.join(str(tmp_path), "nccl_worker.py")
    with open(script, "w") as f:
        f.write(_NCCL_WORKER)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert p.returncode == 0 and b"rccl ok" in p.stdout, p.stdout.decode()[-2000:]


_NCCL_GRAPH_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import cases, harness
from druggen_amd.model import Discriminator, Generator
from druggen_amd.trainer import GANStep, GraphedGANStep
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
case = cases.CASES["c1_b4"]
cfg = cases.net_config(case)
gp, dp = cases.build_params(case)
args = (cfg.act, cfg.vertexes, cfg.edges, cfg.nodes, cfg.dropout)
kw = dict(dim=cfg.dim, depth=cfg.depth, heads=cfg.heads, mlp_ratio=cfg.mlp_ratio)
inp = harness.torch_inputs(case, torch.float32, "cuda")
batch = (inp["disc_edge"], inp["disc_node"], inp["gen_edge"], inp["gen_node"])
eps = (inp["eps_edge"], inp["eps_node"])
ends = []
for graphed in (False, True):
    G, D = Generator(*args, **kw), Discriminator(*args, **kw)
    G.load_state_dict({k: torch.from_numpy(v) for k, v in gp.items()})
    D.load_state_dict({k: torch.from_numpy(v) for k, v in dp.items()})
    G, D = G.cuda(), D.cuda()
    st = GANStep(G, D, g_lr=1e-3, d_lr=1e-3, lambda_gp=case["lambda_gp"])
    if graphed:
        gst = GraphedGANStep(st, *batch, warmup=1, eps=eps, segmented=True)
        assert len(gst.segments[0]) == 3
        st.time_collectives(True)
        for _ in range(3):
            gst.step(*[t.clone() for t in batch], eps=eps)
        n, ms = st.collective_ms()
        assert n == 6, n          # two RCCL all-reduces per replayed step, between the graphs
    else:
        for _ in range(4):
            st.step(*batch, eps=eps)
    torch.cuda.synchronize()
    ends.append(torch.cat([p.detach().reshape(-1) for p in list(G.parameters()) + list(D.parameters())]).cpu())
start = torch.cat([torch.from_numpy(v).reshape(-1) for v in list(gp.values()) + list(dp.values())])
moved, diff = float((ends[0] - start).norm()), float((ends[0] - ends[1]).norm())
assert moved > 0 and diff <= 1e-6 * moved, (diff, moved)
dist.destroy_process_group()
print("rccl graphs ok")
"""


def test_three_graph_step_with_rccl_all_reduces_between_the_graphs(tmp_path):
    """The N > 1 replay sequence -- graph, RCCL all-reduce, graph, RCCL all-reduce, graph -- on this build's RCCL with a
    one-rank `nccl` group (the capture runs while the process group's watchdog thread is alive; the collectives are real
    ncclAllReduce calls on the flat buckets): three replayed steps end on the parameters of four eager steps."""
    script = os.path.join(str(tmp_path), "nccl_graph_worker.py")
    with open(script, "w") as f:
        f.write(f"ROOT = {ROOT!r}\n" + _NCCL_GRAPH_WORKER)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0 and b"rccl graphs ok" in p.stdout, p.stdout.decode()[-3000:]


def test_two_gpu_rccl_bench_keeps_replicas_identical(tmp_path):
    """On a box with >= 2 GPUs (the driver's 8-GPU scaling node; skipped on the 1-GPU test boxes): bench.py under
    torch.distributed.run with the `nccl` backend, one rank per GPU, two timed steps; every rank must end on the same
    parameters (bench.py --check-replicas compares per-tensor checksums with MIN / MAX all-reduces) and rank 0 must
    report n_gpus = 2 with weak scaling."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-extra", "--check-replicas", "--batch", "64"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["replicas_identical"] is True
    assert out["config"]["global_batch"] == 128


@pytest.mark.parametrize("graph", [False, True])
def test_bench_n2_line_on_one_gpu_reports_allreduce_time_and_identical_replicas(tmp_path, graph):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one rank per process), with both ranks on
    cuda:0 and the `gloo` backend (DG_DIST_BACKEND: the functional twin of the RCCL run, which needs two GPUs): the line
    must carry n_gpus = 2, the per-rank time of the two gradient all-reduces per step and replicas_identical = true
    without any extra flag.  `--graph`: every rank replays its step as three hipGraphs cut at the two all-reduces."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DG_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-extra", "--batch", "8", "--vertexes", "9", "--depth", "1"] + (["--graph"] if graph else [])
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    out = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["global_batch"] == 16
    assert out["replicas_identical"] is True
    if graph:
        assert out["config"]["hip_graph_replay"] is True and out["timed_region"].startswith("three hipGraphs")
    else:      # the default line times the eager launches (`value`) AND, beside them, the three-graph replay
        replay = out["hip_graph_replay_same_step"]
        assert replay["value"] > 0 and out["config"]["hip_graph_replay"] is False and out["timed_region"].startswith("eager launches")
        assert abs(out["value"] - 16 * out["steps"] / (out["ms_per_step"] * out["steps"] / 1e3)) < 1e-3 * out["value"]
    ar = out["allreduce"]
    assert len(ar["per_rank_ms_per_step"]) == 2 and ar["collectives_per_step"] == 2 and ar["max_ms_per_step"] > 0


def test_bench_line_is_one_short_parseable_record(tmp_path):
    """The driver stores bench.py's stdout line; round 4's 21 KB line was recorded as unparseable.  The default single-GPU
    line (with the CPU baseline, the secondary blocks switched off for speed) must be ONE JSON object under 6 KB that
    carries the contract fields, `roofline` (with its kernel symbol; `frac` = the SURVEY 8d floor bytes -- inputs + outputs of
    the launch -- over the HIP-event time, `frac_of_moved` beside it) and `cpu_baseline`."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "64", "--no-extra",
           "--cpu-batch", "2"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 6144, (len(lines), [len(l) for l in lines])
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    rf = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_symbol", "bytes_floor_per_launch", "frac_of_moved", "avg_us"):
        assert k in rf, k
    # the driver-parsed fraction IS the section-8d definition: the launch's algorithmic work / its average duration / the peak of
    # the roof it is bound by -- floor bytes (inputs + outputs) against 8 TB/s, or useful flops against the MFMA peak of its arithmetic
    hbm_frac = rf["bytes_floor_per_launch"] / (rf["avg_us"] * 1e-6) / 8e12
    if rf["bound"] == "hbm":
        assert abs(rf["frac"] - hbm_frac) < 2e-3 * rf["frac"] and 0 < rf["frac"] <= rf["frac_of_moved"] < 1.0
    else:
        assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac_of_hbm_floor"] - hbm_frac) < 2e-3 * hbm_frac
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6 and hbm_frac < rf["frac"] < 1.0
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] == "port"
    assert "kernels" not in out and "roofline_all" not in out
    # `value` = the eagerly launched region, always; beside it the same steps replayed from the hipGraph and the strict-float32
    # hidden-storage run the default mode's precision trade is measured against, and the committed parity margin of the mode
    graph = out["hip_graph_replay_same_step"]
    assert graph["steps"] == 2 and graph["value"] > 0 and all(v == v for v in graph["losses"])
    assert out["config"]["hip_graph_replay"] is False and out["timed_region"].startswith("eager launches")
    assert abs(out["value"] - 64 * 1e3 / out["ms_per_step"]) < 1e-3 * out["value"]
    strict = out["strict_f32_hidden"]
    assert strict["value"] > 0 and strict["finite_losses"] is True
    assert out["config"]["hidden_storage"] == "dh16"
    pm = out["parity_margin"]
    assert pm is None or (pm["hidden_storage"] == "dh16" and 0 < pm["worst_golden_tensor_error"] < pm["bar"])


def test_dataparallel_replicas_on_one_device_run_the_gradient_penalty():
    """The reference's --parallel path (train.py:220-223) wraps D in nn.DataParallel: replicas are THREADS.  Two replicas
    on cuda:0 (device_ids=[0, 0]) go through discriminator_loss -- the batched D(real, fake) forward, the gradient
    penalty's second-order forward, its inputs-only first backward and the double backward -- concurrently; the
    process-wide pass flags and the per-thread workspaces must give the gradients of the unwrapped module."""
    import cases
    import harness
    from druggen_amd.model import Discriminator, Generator, discriminator_loss
    case = cases.CASES["c1_b4"]
    cfg = cases.net_config(case)
    gp, dp = cases.build_params(case)
    args = (cfg.act, cfg.vertexes, cfg.edges, cfg.nodes, cfg.dropout)
    kw = dict(dim=cfg.dim, depth=cfg.depth, heads=cfg.heads, mlp_ratio=cfg.mlp_ratio)
    inp = harness.torch_inputs(case, torch.float32, "cuda")
    eps = (inp["eps_edge"], inp["eps_node"])
    grads = []
    for parallel in (False, True):
        G, D = Generator(*args, **kw), Discriminator(*args, **kw)
        G.load_state_dict({k: torch.from_numpy(v) for k, v in gp.items()})
        D.load_state_dict({k: torch.from_numpy(v) for k, v in dp.items()})
        G, D = G.cuda(), D.cuda()
        Dw = torch.nn.DataParallel(D, device_ids=[0, 0]) if parallel else D
        for _ in range(3 if parallel else 1):          # repeat: a race would not show up every time
            D.zero_grad(set_to_none=True)
            _, _, d_loss = discriminator_loss(G, Dw, inp["disc_edge"], inp["disc_node"], inp["gen_edge"], inp["gen_node"],
                                              case["batch"], inp["gen_node"].device, case["lambda_gp"], eps=eps)
            d_loss.backward()
            torch.cuda.synchronize()
            cur = {k: (None if p.grad is None else p.grad.clone()) for k, p in D.named_parameters()}
            if parallel:
                grads.append((float(d_loss), cur))
        if not parallel:
            ref = (float(d_loss), cur)
    for loss, cur in grads:
        assert abs(loss - ref[0]) <= 1e-4 * max(1.0, abs(ref[0]))
        # nn.DataParallel's Broadcast backward materialises zeros for parameters no replica used: the Discriminator's
        # dead last-block edge branch (grad None on the bare module, reference models.py:202-207) reads all-zero here
        for k, v in ref[1].items():
            if v is None:
                assert cur[k] is None or float(cur[k].abs().max()) == 0.0, k
        num = sum(float(((cur[k] - v) ** 2).sum()) for k, v in ref[1].items() if v is not None)
        den = sum(float((v ** 2).sum()) for v in ref[1].values() if v is not None)
        assert num <= (1e-3 ** 2) * den, (num, den)
