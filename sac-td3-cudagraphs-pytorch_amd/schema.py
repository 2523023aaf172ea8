"""Parameter schema of the reference's networks (agents/nets.py:66-82, 112-141, 178-204) as flat arrays.

The engine exchanges one float32 vector per network in state_dict order; these helpers convert between that
vector and a {key: array} mapping with the reference's key names, and produce the reference's initial values
(orthogonal weights, zero biases, LN ones/zeros - agents/nets.py:34-49) with the SAME torch RNG consumption
order as agents/agent.py:61-105, so `torch.manual_seed(seed)` gives the reference's initial parameters.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np

HID = 256  # agents/agent.py:56,101


def net_keys(in_dim: int, n_head: int, layer_norm: bool) -> List[Tuple[str, Tuple[int, ...]]]:
    keys: List[Tuple[str, Tuple[int, ...]]] = []
    for blk, n_in in (("fc_block_1", in_dim), ("fc_block_2", HID)):
        keys += [(f"fc_stack.{blk}.fc.weight", (HID, n_in)), (f"fc_stack.{blk}.fc.bias", (HID,))]
        if layer_norm:
            keys += [(f"fc_stack.{blk}.ln.weight", (HID,)), (f"fc_stack.{blk}.ln.bias", (HID,))]
    keys += [("head.weight", (n_head, HID)), ("head.bias", (n_head,))]
    return keys


def net_numel(in_dim: int, n_head: int, layer_norm: bool) -> int:
    return sum(int(np.prod(s)) for _, s in net_keys(in_dim, n_head, layer_norm))


def flat_to_dict(flat, in_dim: int, n_head: int, layer_norm: bool) -> "OrderedDict[str, np.ndarray]":
    flat = np.asarray(flat, np.float32).reshape(-1)
    assert flat.size == net_numel(in_dim, n_head, layer_norm), (flat.size, net_numel(in_dim, n_head, layer_norm))
    out, off = OrderedDict(), 0
    for k, shp in net_keys(in_dim, n_head, layer_norm):
        n = int(np.prod(shp))
        out[k] = flat[off:off + n].reshape(shp).copy()
        off += n
    return out


def dict_to_flat(sd: Dict[str, np.ndarray], in_dim: int, n_head: int, layer_norm: bool) -> np.ndarray:
    parts = []
    for k, shp in net_keys(in_dim, n_head, layer_norm):
        v = sd[k]
        v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        assert tuple(v.shape) == shp, (k, v.shape, shp)
        parts.append(np.asarray(v, np.float32).reshape(-1))
    return np.concatenate(parts)


def _init_net(in_dim: int, n_head: int, layer_norm: bool) -> np.ndarray:
    """One network's initial parameters; consumes the torch RNG exactly like building the reference module:
    nn.Linear default init for fc1, fc2, head (draws that are then overwritten), followed by orthogonal_
    on fc1, fc2, head (agents/nets.py:85-86 `apply(init())` order)."""
    import torch
    from torch import nn
    lins = [nn.Linear(in_dim, HID), nn.Linear(HID, HID), nn.Linear(HID, n_head)]
    for lin in lins:
        nn.init.orthogonal_(lin.weight)
        nn.init.zeros_(lin.bias)
    sd = {}
    for name, lin in zip(("fc_stack.fc_block_1", "fc_stack.fc_block_2"), lins[:2]):
        sd[f"{name}.fc.weight"], sd[f"{name}.fc.bias"] = lin.weight, lin.bias
        if layer_norm:
            sd[f"{name}.ln.weight"], sd[f"{name}.ln.bias"] = torch.ones(HID), torch.zeros(HID)
    sd["head.weight"], sd["head.bias"] = lins[2].weight, lins[2].bias
    return dict_to_flat(sd, in_dim, n_head, layer_norm)


def reference_initial_params(ob_dim: int, ac_dim: int, td3: bool, layer_norm: bool):
    """(actor_flat, critics_flat[2 nets back to back]) under the caller's torch seed.
    Draw order of agents/agent.py:61-105: actor, actor_detach (discarded), qnet1, qnet2."""
    nh = ac_dim if td3 else 2 * ac_dim
    actor = _init_net(ob_dim, nh, layer_norm)
    _init_net(ob_dim, nh, layer_norm)  # `actor_detach` is a second real initialisation in the reference
    q1 = _init_net(ob_dim + ac_dim, 1, layer_norm)
    q2 = _init_net(ob_dim + ac_dim, 1, layer_norm)
    return actor, np.concatenate([q1, q2])
