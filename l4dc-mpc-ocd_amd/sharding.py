"""Sharding of the (candidate x init x sample) episode batch across ranks.

The reference's only parallel axis is a multiprocessing.Pool over init groups
(experiments/run_mpc_ord.py:83-90); episodes are fully independent
(interact_drive/reward_design/mpc_ord.py:128-137).  Here rank g takes the
contiguous candidate block [g*P/G, (g+1)*P/G) x all inits x all samples, so a
candidate's reduction over inits stays on one rank and in init order; the data
path has no collective.  Per generation there is exactly one collective: an
all-gather of the fp32 per-episode returns (<= 128 KiB in total at the largest
BASELINE config -- latency-bound, so a single-shot all-gather, not a chunked
ring).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def candidate_block(P: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous candidate range of `rank`; sizes differ by at most one."""
    base, rem = divmod(P, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def episode_range(P: int, N: int, S: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Flat episode range e = (p*N + n)*S + s owned by `rank`."""
    lo, hi = candidate_block(P, world_size, rank)
    return lo * N * S, hi * N * S


_GATHER_BUFS = {}


def gather_returns(local: torch.Tensor, P: int, N: int, S: int, group: Optional[dist.ProcessGroup] = None,
                   out: Optional[torch.Tensor] = None, force: bool = False) -> torch.Tensor:
    """All-gather the per-episode returns of every rank into the full [P*N*S] vector (on every rank).
    The receive buffer is kept per (device, size) and reused across generations (a generation is ~1.7 ms:
    allocations are worth avoiding), so the result is valid until the next gather of the same size -- every caller
    copies it to the host at once.  `out`: a caller-owned [world_size * max_block] buffer instead.
    `force`: run the collective even in a one-rank group (bench.py --force-collective and the RCCL self-test:
    the same all_gather_into_tensor call on device memory that N ranks make)."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    if dist.get_world_size(group) == 1 and not force:
        return local
    ws = dist.get_world_size(group)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        local = local.cpu()                  # rehearsal backend: gloo gathers host tensors (RCCL takes device memory)
        out = None
    sizes = [(candidate_block(P, ws, r)[1] - candidate_block(P, ws, r)[0]) * N * S for r in range(ws)]
    m = max(sizes)
    if local.numel() != sizes[dist.get_rank(group)]:
        raise ValueError(f"rank holds {local.numel()} returns, expected {sizes[dist.get_rank(group)]}")
    padded = local if local.numel() == m else torch.cat([local, local.new_zeros(m - local.numel())])
    if out is None or out.numel() != ws * m or out.device != local.device or out.dtype != local.dtype:
        key = (str(local.device), ws * m, local.dtype)
        out = _GATHER_BUFS.get(key)
        if out is None:
            if len(_GATHER_BUFS) > 16:
                _GATHER_BUFS.clear()
            out = _GATHER_BUFS[key] = torch.empty(ws * m, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    if all(s == m for s in sizes):
        return out
    return torch.cat([out[r * m: r * m + sizes[r]] for r in range(ws)])


def fitness_from_returns(returns, P: int, N: int, S: int) -> np.ndarray:
    """[P*N*S] fp32 sample rewards -> [P] CMA-ES costs, with the reference's accumulation types.

    Per (candidate, init): samples are summed in fp32 (TensorFlow scalars, mpc_ord.py:102);
    across inits the Python sum promotes to float64 (np.float32 + int under NumPy 1.x value-based
    casting, mpc_ord.py:126,137); then / num_samples and negated (mpc_ord.py:139,151).
    """
    r = np.asarray(returns, dtype=np.float32).reshape(P, N, S)
    # add.accumulate is strictly sequential along the axis (np.sum is pairwise and would round differently)
    per_init = r[:, :, 0] if S == 1 else np.add.accumulate(r, axis=2, dtype=np.float32)[:, :, -1]
    total = np.add.accumulate(per_init.astype(np.float64), axis=1)[:, -1]
    total = total / S
    return -total
