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


# Fixtures whose GENERATOR gradient sits on a ReLU threshold of the Discriminator: for the reference's generator output a
# pre-activation of D is zero to within float32 rounding, so the SIGN of a rounding error anywhere upstream decides which of two
# gradients the G step sees (two molecules: one unit of one row is 1e-3 of the gradient).  Measured on the two-launch float32
# feed-forward of round 5 (profiles/r06_c5_b2_threshold.txt, test_c5_b2_generator_gradient_sits_on_a_relu_threshold below): 45 % of
# two-ulp perturbations of the generator's logits move every G gradient tensor by 1.1e-3 - 1.8e-3 at once (uniformly, readouts
# included: the upstream gradient changed, not the generator's backward), the others stay at 2.2e-4 - 3.2e-4.  c2_b2: 0 of 40.
THRESHOLD_CASES = {"c5_b2": 2.5e-3}


def _perturbed_generator(G, seed, ulps=2e-7):
    """G with its logits multiplied by (1 + ulps * N(0, 1)): what another float32-class forward would hand to D."""
    orig = G.forward

    def forward(*a, **k):
        n, e, ns, es = orig(*a, **k)
        g = torch.Generator(device=ns.device).manual_seed(seed)
        return (n, e, ns * (1 + ulps * torch.randn(ns.shape, device=ns.device, generator=g)),
                es * (1 + ulps * torch.randn(es.shape, device=es.device, generator=g)))
    G.forward = forward
    return G


def _worst_g_grad(case, fx, seed=None):
    cfg, G, D = _build(case)
    if seed is not None:
        _perturbed_generator(G, seed)
    inp = harness.torch_inputs(case, torch.float32, "cuda")
    res = harness.run_step(G, D, _d_loss, _g_loss, inp, case["lambda_gp"])
    return harness.grad_table_errors(case, fx, "ref64", "G.grad", res["G.grad"])[0]


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
    g_tol = TOL_GRAD
    if name in THRESHOLD_CASES:
        # the D step (losses, every D gradient) is held to the bar as everywhere; the G gradient must be on ONE of the fixture's
        # two branches, and the reference's branch must be reachable within two ulps of the generator's logits
        worst_g = harness.grad_table_errors(case, fx, "ref64", "G.grad", res["G.grad"])[0]
        if worst_g[0] > TOL_GRAD:
            g_tol = THRESHOLD_CASES[name]
            near = [_worst_g_grad(case, fx, seed)[0] for seed in range(1, 9)]
            assert min(near) <= TOL_GRAD, f"{name}: no two-ulp neighbour of the generator output is on the reference's branch: {near}"
            print(name, f"G gradient on the other branch of the fixture's ReLU threshold ({worst_g[0]:.2e}); two-ulp neighbours: "
                  + " ".join(f"{e:.1e}" for e in near))
    harness.compare_scalar(res["d_loss"], fx["ref64/d_loss"], TOL_OUT, "d_loss")
    harness.compare_scalar(res["g_loss"], fx["ref64/g_loss"], TOL_OUT, "g_loss")
    assert res["G.grad_in_d_step"] == [], "generator received gradients in the D step"
    worst = {"D.grad": harness.compare_grad_table(case, fx, "ref64", "D.grad", res["D.grad"], TOL_GRAD),
             "G.grad": harness.compare_grad_table(case, fx, "ref64", "G.grad", res["G.grad"], g_tol)}
    if case["full"]:
        worst["D.delta"] = harness.compare_grad_table(case, fx, "ref64", "D.delta", res["D.delta"], 0.05)
        worst["G.delta"] = harness.compare_grad_table(case, fx, "ref64", "G.delta", res["G.delta"], 0.05)
    print(name, {k: (f"{v[0]:.2e}", v[1]) for k, v in worst.items()})


def test_c5_b2_generator_gradient_sits_on_a_relu_threshold():
    """The evidence behind THRESHOLD_CASES, on the TWO-LAUNCH float32 feed-forward (the round-5 path): two-ulp perturbations of the
    generator's logits put the G gradient of c5_b2 on either of two branches 1e-3 apart; c2_b2 has one branch."""
    from druggen_amd import functional as dgf
    dgf.set_fused_ffn_f32(False)
    try:
        case, fx = cases.CASES["c5_b2"], harness.load_fixture("c5_b2")
        errs = [_worst_g_grad(case, fx, seed)[0] for seed in range(1, 17)]
        lo, hi = [e for e in errs if e <= TOL_GRAD], [e for e in errs if e > TOL_GRAD]
        assert lo and hi, errs
        assert max(lo) < 5e-4 and max(hi) < THRESHOLD_CASES["c5_b2"], errs
        case, fx = cases.CASES["c2_b2"], harness.load_fixture("c2_b2")
        errs2 = [_worst_g_grad(case, fx, seed)[0] for seed in range(1, 9)]
        assert max(errs2) < TOL_GRAD, errs2
        print("c5_b2:", " ".join(f"{e:.1e}" for e in errs), "| c2_b2:", " ".join(f"{e:.1e}" for e in errs2))
    finally:
        dgf.set_fused_ffn_f32(True)


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


def _step_against_fp64_oracle(cfg, B, seed, with_g_step, scale=1.7):
    """One D step (and optionally the G step) of the HIP modules against the fp64 oracle on the CPU: losses at TOL_OUT,
    every parameter gradient at TOL_GRAD per tensor (floor: ||all grads|| / sqrt(n_tensors), SURVEY section 7 hard part 4),
    the None-gradient sets equal.  Returns the worst per-tensor error of each network."""
    from druggen_amd import synth
    from druggen_amd.model import Discriminator, Generator, discriminator_loss, generator_loss
    gp = synth.fill_parameters(orc.generator_schema(cfg), seed, scale)
    dp = synth.fill_parameters(orc.discriminator_schema(cfg), seed + 1, scale)
    args = (cfg.act, cfg.vertexes, cfg.edges, cfg.nodes, cfg.dropout)
    kw = dict(dim=cfg.dim, depth=cfg.depth, heads=cfg.heads, mlp_ratio=cfg.mlp_ratio)
    G, D = Generator(*args, **kw), Discriminator(*args, **kw)
    G.load_state_dict({k: torch.from_numpy(v) for k, v in gp.items()})
    D.load_state_dict({k: torch.from_numpy(v) for k, v in dp.items()})
    G, D = G.cuda(), D.cuda()
    OG = orc.OracleNet("G", cfg, {k: torch.from_numpy(v).double() for k, v in gp.items()})
    OD = orc.OracleNet("D", cfg, {k: torch.from_numpy(v).double() for k, v in dp.items()})
    a, x, _, _ = synth.molecule_batch(B, cfg.vertexes, cfg.edges, cfg.nodes, seed=seed + 2)
    da, dx, _, _ = synth.molecule_batch(B, cfg.vertexes, cfg.edges, cfg.nodes, seed=seed + 3)
    ee, en = synth.interpolation_eps(B, seed + 4)
    t64 = lambda v: torch.from_numpy(v).double()
    t32 = lambda v: torch.from_numpy(v).cuda()
    _, _, d_loss = discriminator_loss(G, D, t32(da), t32(dx), t32(a), t32(x), B, "cuda", 10.0, eps=(t32(ee), t32(en)))
    d_loss.backward()
    import os
    torch.set_num_threads(max(1, min(64, (os.cpu_count() or 2) // 2)))
    _, _, od = orc.discriminator_loss(OG, OD, t64(da), t64(dx), t64(a), t64(x), 10.0, t64(ee), t64(en))
    od.backward()
    harness.compare_scalar(d_loss, float(od.detach()), TOL_OUT, "d_loss")
    assert all(p.grad is None for p in G.parameters()), "generator received gradients in the D step"

    def table_errors(mod, onet):
        want = dict(zip(onet.names, onet.flat))
        live = [k for k, p in want.items() if p.grad is not None]
        tot = sum(float((want[k].grad ** 2).sum()) for k in live) ** 0.5
        worst = (0.0, None)
        for k, p in mod.named_parameters():
            if want[k].grad is None:
                assert p.grad is None, k
                continue
            assert p.grad is not None, k
            err = (p.grad.double().cpu() - want[k].grad).norm().item() / max(want[k].grad.norm().item(), tot / len(live) ** 0.5)
            assert err <= TOL_GRAD, (k, err)
            if err >= worst[0]:
                worst = (err, k)
        return worst
    worst = {"D": table_errors(D, OD)}
    if with_g_step:
        for net in (D, G):
            net.zero_grad(set_to_none=True)
        for p in list(OD.flat) + list(OG.flat):
            p.grad = None
        g_loss = generator_loss(G, D, t32(a), t32(x), B)[0]
        g_loss.backward()
        og = orc.generator_loss(OG, OD, t64(a), t64(x))[0]
        og.backward()
        harness.compare_scalar(g_loss, float(og.detach()), TOL_OUT, "g_loss")
        worst["G"] = table_errors(G, OG)
    return worst


def test_headline_batch_one_layer_against_fp64_oracle():
    """BASELINE configs[1]'s batch (B = 256, N = 45, E = 5, M = 13) with ONE encoder layer, so that the fp64 oracle on the host
    finishes in well under a minute: D-step loss and every D gradient, HIP float32 vs fp64, 1e-3 per tensor.  This is the
    size at which the fused float32 attention-half backward (B >= 128) and the riding launches run unforced."""
    cfg = orc.NetConfig(act="relu", vertexes=45, edges=5, nodes=13, dropout=0.0, dim=128, depth=1, heads=8, mlp_ratio=3)
    print("worst per-tensor error", _step_against_fp64_oracle(cfg, 256, 401, with_g_step=False))


def test_headline_depth_batch_32_against_fp64_oracle():
    """The headline model (L = 4) at B = 32: D step and G step, losses and all gradients of both networks vs fp64."""
    cfg = orc.NetConfig(act="relu", vertexes=45, edges=5, nodes=13, dropout=0.0, dim=128, depth=4, heads=8, mlp_ratio=3)
    print("worst per-tensor error", _step_against_fp64_oracle(cfg, 32, 411, with_g_step=True))


def test_headline_configuration_at_its_real_size_against_fp64_oracle():
    """BASELINE configs[1] at its REAL size -- B = 256, N = 45, L = 4 -- once: the D step (critic terms + gradient penalty with its
    double backward) of the HIP float32 path against the fp64 oracle on the host (cores / 2 threads, a few minutes): loss at
    1e-3, every Discriminator gradient tensor at 1e-3.  (B = 256 at L = 1 and L = 4 at B = 32 run above; this is the size
    bench.py times.)"""
    cfg = orc.NetConfig(act="relu", vertexes=45, edges=5, nodes=13, dropout=0.0, dim=128, depth=4, heads=8, mlp_ratio=3)
    worst = _step_against_fp64_oracle(cfg, 256, 421, with_g_step=False)
    print("worst per-tensor error", worst)
    assert worst["D"][0] < 6e-4


def test_parity_margin_tripwire():
    """The precision the default mode has spent (backward-only fp16 planes) must not grow unnoticed: the worst
    gradient tensor over ALL reference goldens stays below 6e-4 of the 1e-3 bar (4.8e-4 when the trade was made; the table is
    profiles/r06_parity_by_hidden_storage.txt).  For a fixture whose G gradient sits on a ReLU threshold of D (THRESHOLD_CASES) the
    branch of the reference counts: the best of the unperturbed step and its two-ulp neighbours."""
    from druggen_amd import functional as dgf
    assert dgf.hidden_storage() == "dh16"
    worst = (0.0, None, None)
    for name, case in cases.CASES.items():
        fx = harness.load_fixture(name)
        cfg, G, D = _build(case)
        inp = harness.torch_inputs(case, torch.float32, "cuda")
        res = harness.run_step(G, D, _d_loss, _g_loss, inp, case["lambda_gp"])
        d = harness.grad_table_errors(case, fx, "ref64", "D.grad", res["D.grad"])[0]
        g = harness.grad_table_errors(case, fx, "ref64", "G.grad", res["G.grad"])[0]
        if name in THRESHOLD_CASES and g[0] > 6e-4:
            g = min([g] + [_worst_g_grad(case, fx, seed) for seed in range(1, 9)])
        for e, k in (d, g):
            if e > worst[0]:
                worst = (e, name, k)
    print("worst golden tensor", worst)
    assert worst[0] <= 6e-4, worst


@pytest.mark.parametrize("name", ["c1_b4", "c2_b2", "c1_tanh_b4"])
def test_forced_fused_attention_half_backward_matches_golden_and_the_two_launch_path(name):
    """`dg_attn_half_f32_bwd1` is the default for float32 at B >= 128 but every golden case is smaller and took
    the two-launch path.  options.attn_half_f32_bwd = "force" routes every batch size through it: D and G gradients plus the
    penalty's second order (inside d_loss) must match the reference goldens at 1e-3 and the `off` path at 3e-4 (the two paths
    round differently, and the backward's fp16-plane tensors of DESIGN 3.16 carry those differences on)."""
    case = cases.CASES[name]
    fx = harness.load_fixture(name)
    inp = harness.torch_inputs(case, torch.float32, "cuda")
    res = {}
    for mode in ("off", "force"):
        from druggen_amd.options import options
        options.attn_half_f32_bwd = mode
        cfg, G, D = _build(case)
        res[mode] = harness.run_step(G, D, _d_loss, _g_loss, inp, case["lambda_gp"])
        harness.compare_step(case, fx, "ref64", res[mode], TOL_OUT, TOL_GRAD)
    for grp in ("D.grad", "G.grad"):
        num = den = 0.0
        for k, v in res["off"][grp].items():
            w = res["force"][grp][k]
            assert (v is None) == (w is None), k
            if v is not None:
                num += float(((v - w).double() ** 2).sum())
                den += float((v.double() ** 2).sum())
        assert num <= (3e-4 ** 2) * den, (grp, num, den)


def test_fp16_hidden_plane_mode_is_a_labelled_1e_2_method_at_batch_32(hidden_mode):
    """options.hidden = "f16" (DG_HIDDEN=f16 at import) (the FORWARD's hidden tensor h as one fp16 plane + row scales too) is an opt-in mode OUTSIDE the 1e-3
    bar: rounding h perturbs the forward and flips ReLU masks behind it.  At the headline model, B = 32, its losses stay
    within 1e-3 and every gradient tensor within 1e-2 of the fp64 oracle (measured worst 4.0e-3); the default (dh16: only the
    backward's hidden tensors) holds 1e-3 (the test above)."""
    from druggen_amd import functional as dgf
    hidden_mode("f16")
    assert dgf.hidden_storage() == "f16"
    cfg = orc.NetConfig(act="relu", vertexes=45, edges=5, nodes=13, dropout=0.0, dim=128, depth=4, heads=8, mlp_ratio=3)
    global TOL_GRAD
    keep = TOL_GRAD
    TOL_GRAD = 1e-2
    try:
        worst = _step_against_fp64_oracle(cfg, 32, 411, with_g_step=True)
    finally:
        TOL_GRAD = keep
    print("worst per-tensor error with fp16 hidden planes", worst)
