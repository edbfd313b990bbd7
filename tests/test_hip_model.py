"""GPU parity of the drop-in modules (druggen_amd.model == src.model API) against
the reference-generated golden vectors and against the oracle."""
import pytest
import torch

import cases
import harness
from oracle import druggen_oracle as orc

pytestmark = pytest.mark.gpu

# north_star: outputs and gradients within 1e-3 rel (fp32) of the reference path
TOL_OUT = 1e-3
TOL_GRAD = 1e-3


def _build(case, device="cuda"):
    from druggen_amd.model import Generator, Discriminator
    cfg = cases.net_config(case)
    args = (cfg.act, cfg.vertexes, cfg.edges, cfg.nodes, cfg.dropout)
    kw = dict(dim=cfg.dim, depth=cfg.depth, heads=cfg.heads, mlp_ratio=cfg.mlp_ratio)
    G, D = Generator(*args, **kw), Discriminator(*args, **kw)
    gp, dp = cases.build_params(case)
    G.load_state_dict({k: torch.from_numpy(v) for k, v in gp.items()}, strict=True)
    D.load_state_dict({k: torch.from_numpy(v) for k, v in dp.items()}, strict=True)
    return cfg, G.to(device), D.to(device)


def _d_loss(G, D, de, dn, ge, gn, lam, ee, en):
    from druggen_amd.model import discriminator_loss
    return discriminator_loss(G, D, de, dn, ge, gn, gn.shape[0], gn.device, lam, eps=(ee, en))[2]


def _g_loss(G, D, ge, gn):
    from druggen_amd.model import generator_loss
    return generator_loss(G, D, ge, gn, gn.shape[0])[0]


@pytest.mark.parametrize("name", list(cases.CASES))
def test_gan_step_matches_reference_golden(name):
    case = cases.CASES[name]
    fx = harness.load_fixture(name)
    cfg, G, D = _build(case)
    inp = harness.torch_inputs(case, torch.float32, "cuda")
    with torch.no_grad():
        node, edge, ns, es = G(inp["gen_edge"], inp["gen_node"])
        real = D(inp["disc_edge"], inp["disc_node"])
        fake = D(es, ns)
    for tag in ("ref64", "ref32"):
        assert harness.rel_err(ns.cpu().numpy(), fx[f"{tag}/G.node_sample"]) < TOL_OUT
        assert harness.rel_err(es.cpu().numpy(), fx[f"{tag}/G.edge_sample"]) < TOL_OUT
        assert harness.rel_err(real.cpu().numpy(), fx[f"{tag}/D.real_logits"]) < TOL_OUT
        assert harness.rel_err(fake.cpu().numpy(), fx[f"{tag}/D.fake_logits"]) < TOL_OUT
        if case["full"]:
            assert harness.rel_err(node.cpu().numpy(), fx[f"{tag}/G.node"]) < TOL_OUT
            assert harness.rel_err(edge.cpu().numpy(), fx[f"{tag}/G.edge"]) < TOL_OUT
    from druggen_amd.model import gradient_penalty
    gp = gradient_penalty(D, inp["disc_node"], inp["disc_edge"], ns, es, case["batch"], ns.device,
                          eps=(inp["eps_edge"], inp["eps_node"]))
    harness.compare_scalar(gp, fx["ref64/gp"], TOL_OUT, "gp")
    res = harness.run_step(G, D, _d_loss, _g_loss, inp, case["lambda_gp"])
    worst = harness.compare_step(case, fx, "ref64", res, TOL_OUT, TOL_GRAD,
                                 rtol_delta=0.05 if case["full"] else None)
    print(name, {k: (f"{v[0]:.2e}", v[1]) for k, v in worst.items()})


def test_reference_loss_code_path_works_on_these_modules():
    """The reference's own gradient_penalty structure (autograd.grad with
    create_graph=True, then backward) drives the HIP double-backward kernels:
    restated here by the oracle's loss functions, which take any callables."""
    case = cases.CASES["tiny_relu"]
    fx = harness.load_fixture("tiny_relu")
    cfg, G, D = _build(case)
    inp = harness.torch_inputs(case, torch.float32, "cuda")
    _, _, d_loss = orc.discriminator_loss(G, D, inp["disc_edge"], inp["disc_node"], inp["gen_edge"],
                                          inp["gen_node"], case["lambda_gp"], inp["eps_edge"], inp["eps_node"])
    d_loss.backward()
    harness.compare_scalar(d_loss, fx["ref64/d_loss"], TOL_OUT, "d_loss")
    table = {k: p.grad for k, p in D.named_parameters()}
    harness.compare_grad_table(case, fx, "ref64", "D.grad", table, TOL_GRAD)


def test_state_dict_schema_and_checkpoint_roundtrip(tmp_path):
    """train.py:259-263 saves bare state_dicts; keys/shapes must equal the reference's."""
    case = cases.CASES["c1_b4"]
    cfg, G, D = _build(case, device="cpu")
    assert [(k, tuple(v.shape)) for k, v in G.state_dict().items()] == [(k, tuple(s)) for k, s in orc.generator_schema(cfg)]
    assert [(k, tuple(v.shape)) for k, v in D.state_dict().items()] == [(k, tuple(s)) for k, s in orc.discriminator_schema(cfg)]
    path = tmp_path / "1-100-G.ckpt"
    torch.save(G.state_dict(), path)
    from druggen_amd.model import Generator
    G2 = Generator(cfg.act, cfg.vertexes, cfg.edges, cfg.nodes, cfg.dropout, dim=cfg.dim, depth=cfg.depth,
                   heads=cfg.heads, mlp_ratio=cfg.mlp_ratio)
    G2.load_state_dict(torch.load(path, map_location=lambda storage, loc: storage))
    for a, b in zip(G.state_dict().values(), G2.state_dict().values()):
        assert torch.equal(a, b)


def test_inference_mode_forward():
    """inference.py:180-195 runs G under torch.inference_mode()."""
    case = cases.CASES["c1_b4"]
    fx = harness.load_fixture("c1_b4")
    cfg, G, D = _build(case)
    G.eval()
    inp = harness.torch_inputs(case, torch.float32, "cuda")
    with torch.inference_mode():
        _, _, ns, es = G(inp["gen_edge"], inp["gen_node"])
    assert harness.rel_err(es.cpu().numpy(), fx["ref64/G.edge_sample"]) < TOL_OUT
    assert harness.rel_err(ns.cpu().numpy(), fx["ref64/G.node_sample"]) < TOL_OUT


def test_full_size_batch_shard_consistency():
    """configs[1] (B=256, N=45, L=4): the data-parallel property the multi-GPU
    path relies on -- per-molecule outputs do not depend on what else is in the
    batch, so logits of the full batch equal the concatenation of two shards, and
    the mean-loss gradient equals the average of the shard gradients."""
    from druggen_amd import synth
    from druggen_amd.model import Discriminator
    torch.manual_seed(0)
    D = Discriminator("relu", 45, 5, 13, 0.0, dim=128, depth=4, heads=8, mlp_ratio=3).cuda()
    a, x, _, _ = synth.molecule_batch(256, 45, 5, 13, seed=5)
    a, x = torch.from_numpy(a).cuda(), torch.from_numpy(x).cuda()
    full = D(a, x)
    (-full.mean()).backward()
    g_full = {k: p.grad.clone() for k, p in D.named_parameters() if p.grad is not None}
    D.zero_grad(set_to_none=True)
    parts = []
    for sl in (slice(0, 128), slice(128, 256)):
        out = D(a[sl], x[sl])
        (-out.mean() / 2).backward()
        parts.append(out.detach())
    assert (torch.cat(parts) - full.detach()).abs().max().item() <= 1e-5 * max(1.0, full.abs().max().item())
    tot = torch.sqrt(sum((g ** 2).sum() for g in g_full.values())).item()
    for k, p in D.named_parameters():
        if p.grad is None:
            assert k not in g_full
            continue
        assert (p.grad - g_full[k]).norm().item() <= 1e-4 * max(g_full[k].norm().item(), tot / len(g_full) ** 0.5), k


def test_generator_is_node_permutation_equivariant_full_size():
    """No positional encoding anywhere in G: relabelling the atoms permutes the
    outputs (checked at N=45, L=4, B=64)."""
    from druggen_amd import synth
    from druggen_amd.model import Generator
    torch.manual_seed(1)
    G = Generator("relu", 45, 5, 13, 0.0, dim=128, depth=4, heads=8, mlp_ratio=3).cuda()
    a, x, _, _ = synth.molecule_batch(64, 45, 5, 13, seed=6)
    a, x = torch.from_numpy(a).cuda(), torch.from_numpy(x).cuda()
    perm = torch.randperm(45, device="cuda")
    with torch.no_grad():
        _, _, ns, es = G(a, x)
        _, _, ns_p, es_p = G(a[:, perm][:, :, perm], x[:, perm])
    assert (ns[:, perm] - ns_p).abs().max().item() < 1e-4
    assert (es[:, perm][:, :, perm] - es_p).abs().max().item() < 1e-4


def test_deep_variant_shape_matches_oracle():
    """BASELINE configs[4] geometry (N=90 atoms, 10 bond types; depth cut to 2 so the CPU oracle
    finishes in seconds): forward outputs, D-step loss and all D gradients vs the oracle in fp64."""
    from druggen_amd import synth
    from druggen_amd.model import Discriminator, Generator, discriminator_loss
    cfg = orc.NetConfig(act="relu", vertexes=90, edges=10, nodes=13, dropout=0.0, dim=128, depth=2, heads=8, mlp_ratio=3)
    gp = synth.fill_parameters(orc.generator_schema(cfg), 71, 1.7)
    dp = synth.fill_parameters(orc.discriminator_schema(cfg), 72, 1.7)
    args = (cfg.act, cfg.vertexes, cfg.edges, cfg.nodes, cfg.dropout)
    kw = dict(dim=cfg.dim, depth=cfg.depth, heads=cfg.heads, mlp_ratio=cfg.mlp_ratio)
    G, D = Generator(*args, **kw), Discriminator(*args, **kw)
    G.load_state_dict({k: torch.from_numpy(v) for k, v in gp.items()})
    D.load_state_dict({k: torch.from_numpy(v) for k, v in dp.items()})
    G, D = G.cuda(), D.cuda()
    OG = orc.OracleNet("G", cfg, {k: torch.from_numpy(v).double() for k, v in gp.items()})
    OD = orc.OracleNet("D", cfg, {k: torch.from_numpy(v).double() for k, v in dp.items()})
    B = 2
    a, x, _, _ = synth.molecule_batch(B, 90, 10, 13, seed=9)
    da, dx, _, _ = synth.molecule_batch(B, 90, 10, 13, seed=10)
    ee, en = synth.interpolation_eps(B, 9)
    t64 = lambda v: torch.from_numpy(v).double()
    t32 = lambda v: torch.from_numpy(v).cuda()
    _, _, d_loss = discriminator_loss(G, D, t32(da), t32(dx), t32(a), t32(x), B, "cuda", 10.0, eps=(t32(ee), t32(en)))
    d_loss.backward()
    _, _, od = orc.discriminator_loss(OG, OD, t64(da), t64(dx), t64(a), t64(x), 10.0, t64(ee), t64(en))
    od.backward()
    harness.compare_scalar(d_loss, float(od.detach()), TOL_OUT, "d_loss")
    want = dict(zip(OD.names, OD.flat))
    tot = torch.sqrt(sum((p.grad ** 2).sum() for p in OD.flat if p.grad is not None)).item()
    live = [k for k, p in want.items() if p.grad is not None]
    for k, p in D.named_parameters():
        if want[k].grad is None:
            assert p.grad is None, k
            continue
        err = (p.grad.double().cpu() - want[k].grad).norm().item()
        assert err <= TOL_GRAD * max(want[k].grad.norm().item(), tot / len(live) ** 0.5), k
