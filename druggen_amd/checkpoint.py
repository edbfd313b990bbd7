"""Checkpoint I/O in the reference's format (``train.py:250-263``,
``inference.py:135-139``): bare ``state_dict()`` files named
``{epoch}-{iter}-G.ckpt`` / ``{epoch}-{iter}-D.ckpt`` (training) and
``{submodel}-G.ckpt`` (inference).  Files written from an ``nn.DataParallel``
wrapper carry a ``module.`` prefix (``train.py:262``); both spellings load."""
from __future__ import annotations

import os
from typing import Dict

import torch

__all__ = ["save_model", "restore_model", "load_generator", "strip_module_prefix"]


def strip_module_prefix(state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    if state and all(k.startswith("module.") for k in state):
        return {k[len("module."):]: v for k, v in state.items()}
    return dict(state)


def _load(path: str) -> Dict[str, torch.Tensor]:
    # same map_location idiom as the reference (train.py:256): always land on the CPU first
    return strip_module_prefix(torch.load(path, map_location=lambda storage, loc: storage))


def save_model(G, D, model_directory: str, idx: int, i: int) -> None:
    """train.py:259-263."""
    os.makedirs(model_directory, exist_ok=True)
    torch.save(G.state_dict(), os.path.join(model_directory, "{}-{}-G.ckpt".format(idx + 1, i + 1)))
    torch.save(D.state_dict(), os.path.join(model_directory, "{}-{}-D.ckpt".format(idx + 1, i + 1)))


def restore_model(G, D, model_directory: str, epoch, iteration) -> None:
    """train.py:250-257."""
    G.load_state_dict(_load(os.path.join(model_directory, "{}-{}-G.ckpt".format(epoch, iteration))))
    D.load_state_dict(_load(os.path.join(model_directory, "{}-{}-D.ckpt".format(epoch, iteration))))


def load_generator(G, model_directory: str, submodel: str) -> None:
    """inference.py:135-139."""
    G.load_state_dict(_load(os.path.join(model_directory, "{}-G.ckpt".format(submodel))))
