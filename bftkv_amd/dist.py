"""Multi-GPU plumbing of the verifier: shard-by-write partitioning and the one exchange step of the path,
the all-gather of per-write verdict bitmaps.  One process per GPU; ``torch.distributed``
backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.  No data-path collective besides this.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank: the first (n_items % world) ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def max_shard(n_items: int, world: int) -> int:
    return (n_items + world - 1) // world


_consts = {}


def _bit_consts(device):
    """(weights, shifts) uint8 [8] on `device`, created once per device (no per-step host-to-device copies)."""
    key = str(device)
    if key not in _consts:
        _consts[key] = (torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=device),
                        torch.arange(8, dtype=torch.uint8, device=device))
    return _consts[key]


def pack_verdicts(ok: torch.Tensor, n_slots: int) -> torch.Tensor:
    """ok: bool/uint8 [n_local] (1 = CollectiveSignature.Verify returned nil) -> uint8 bitmap of
    ceil(n_slots/8) bytes, bit i of byte i//8 = item i (LSB first), zero padded to n_slots."""
    nbytes = (n_slots + 7) // 8
    weights, _ = _bit_consts(ok.device)
    ok = ok.to(torch.uint8)
    if ok.numel() != nbytes * 8:
        pad = torch.zeros(nbytes * 8, dtype=torch.uint8, device=ok.device)
        pad[:ok.numel()] = ok
        ok = pad
    return (ok.view(nbytes, 8) * weights).sum(dim=1, dtype=torch.uint8)      # distinct bits: the byte sum cannot overflow


def unpack_verdicts(bits: torch.Tensor, n_slots: int) -> torch.Tensor:
    _, shifts = _bit_consts(bits.device)
    return ((bits.unsqueeze(1) >> shifts) & 1).reshape(-1)[:n_slots]


def allgather_verdicts(local_ok: torch.Tensor, n_items: int, group=None) -> torch.Tensor:
    """Every rank contributes the verdicts of its shard (shard_range) and receives the verdict of EVERY write,
    in global write order -- as every replica of the reference independently reaches every decision.
    Traffic: ceil(max_shard/8) bytes per rank (cfg 4: 1M writes over 8 GPUs = 15.6 KB per rank)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    slots = max_shard(n_items, world)
    bits = pack_verdicts(local_ok, slots)
    if world == 1:
        return unpack_verdicts(bits, n_items)
    gathered = torch.empty(world * bits.numel(), dtype=torch.uint8, device=bits.device)
    dist.all_gather_into_tensor(gathered, bits, group=group)
    if n_items % world == 0:
        # equal shards: rank r's bits are the first `slots` of its row
        return unpack_verdicts(gathered, gathered.numel() * 8).view(world, -1)[:, :slots].reshape(-1)
    out = torch.empty(n_items, dtype=torch.uint8, device=bits.device)
    for r in range(world):
        lo, hi = shard_range(n_items, r, world)
        out[lo:hi] = unpack_verdicts(gathered[r * bits.numel():(r + 1) * bits.numel()], slots)[:hi - lo]
    return out
