"""WGAN-GP losses with the reference's signatures (``src/model/loss.py``)."""
from __future__ import annotations

import torch

from .. import functional as dgf


def gradient_penalty(discriminator, real_node, real_edge, fake_node, fake_edge, batch_size, device, *, eps=None):
    """Reference loss.py:4-49.  ``eps=(eps_edge, eps_node)`` injects the two
    uniform draws (otherwise drawn like the reference: edge first, then node)."""
    if eps is None:
        eps_edge = torch.rand(batch_size, 1, 1, 1, device=device)
        eps_node = torch.rand(batch_size, 1, 1, device=device)
    else:
        eps_edge, eps_node = eps
    int_node = (eps_node * real_node + (1 - eps_node) * fake_node).requires_grad_(True)
    int_edge = (eps_edge * real_edge + (1 - eps_edge) * fake_edge).requires_grad_(True)
    with dgf.second_order_forward():      # this graph is differentiated twice (create_graph below)
        logits = discriminator(int_edge, int_node)
    with dgf.inputs_only_backward():      # parameter gradients of this pass are never used
        grad_node, grad_edge = torch.autograd.grad(
            outputs=logits, inputs=[int_node, int_edge], grad_outputs=torch.ones_like(logits),
            create_graph=True, retain_graph=True, only_inputs=True)
    grads = torch.cat([grad_node.reshape(batch_size, -1), grad_edge.reshape(batch_size, -1)], dim=1)
    return ((grads.norm(2, dim=1) - 1) ** 2).mean()


def _has_active_dropout(module) -> bool:
    return any(isinstance(m, torch.nn.Dropout) and m.p > 0 for m in module.modules())


def _accepts_edge_parts(discriminator) -> bool:
    """True for this package's Discriminator (possibly behind DistributedDataParallel's ``.module``): its trunk takes
    the edge batch as a tuple of parts.  nn.DataParallel is excluded on purpose."""
    from .models import _Trunk
    inner = discriminator
    if isinstance(inner, torch.nn.parallel.DistributedDataParallel):
        inner = inner.module
    return isinstance(inner, _Trunk)


def discriminator_loss(generator, discriminator, drug_adj, drug_annot, mol_adj, mol_annot, batch_size, device,
                       lambda_gp, *, eps=None, generator_outputs=None, return_terms=False):
    """Reference loss.py:52-72 -> (node, edge, d_loss); with ``return_terms`` additionally the two summands
    (prediction_fake + prediction_real, lambda_gp * gp) whose graphs share nothing but the parameters.

    The generator output only enters detached, so its forward runs without
    recording a graph (the reference records one it never uses).  ``generator_outputs``
    may carry the 4-tuple of an earlier ``generator(mol_adj, mol_annot)`` call to reuse."""
    if generator_outputs is None:
        with torch.no_grad():
            generator_outputs = generator(mol_adj, mol_annot)
    node, edge, node_sample, edge_sample = generator_outputs
    node_sample, edge_sample = node_sample.detach(), edge_sample.detach()
    if (drug_adj.shape == edge_sample.shape and drug_annot.shape == node_sample.shape
            and not (discriminator.training and _has_active_dropout(discriminator))):
        # D(real) and D(fake) as one pass over the concatenated batch: every molecule is processed
        # independently (no cross-sample op in D), so the logits are the reference's; half the launches,
        # and each weight gradient is accumulated once instead of twice.
        if _accepts_edge_parts(discriminator):
            # the halves stay separate tensors up to the edge embedding: a one-hot real batch takes the table path
            edges = (drug_adj, edge_sample)
        else:       # any other critic (user module, nn.DataParallel: scatter() would split the halves differently)
            edges = torch.cat([drug_adj, edge_sample])
        logits = discriminator(edges, torch.cat([drug_annot, node_sample]))
        n_real = drug_adj.shape[0]
        prediction_real = -torch.mean(logits[:n_real])
        prediction_fake = torch.mean(logits[n_real:])
    else:
        prediction_real = -torch.mean(discriminator(drug_adj, drug_annot))
        prediction_fake = torch.mean(discriminator(edge_sample, node_sample))
    gp = gradient_penalty(discriminator, drug_annot, drug_adj, node_sample, edge_sample, batch_size, device, eps=eps)
    main, pen = prediction_fake + prediction_real, lambda_gp * gp
    if return_terms:
        return node, edge, main + pen, main, pen
    return node, edge, main + pen


def generator_loss(generator, discriminator, mol_adj, mol_annot, batch_size, *, generator_outputs=None):
    """Reference loss.py:75-84 -> (g_loss, node, edge, node_sample, edge_sample)."""
    if generator_outputs is None:
        generator_outputs = generator(mol_adj, mol_annot)
    node, edge, node_sample, edge_sample = generator_outputs
    g_loss = -torch.mean(discriminator(edge_sample, node_sample))
    return g_loss, node, edge, node_sample, edge_sample
