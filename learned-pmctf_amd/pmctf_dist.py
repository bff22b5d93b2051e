"""Multi-GPU scheduling of the pMCTF encode path: one process per GPU, closed GOPs are independent units.

Rank r of `world` encodes GOPs r, r+world, r+2*world, ... with no data-path collective (GOP-level data
parallelism, weak scaling).  Only per-frame metrics (bits, PSNR) are gathered at the end, as small
float64 tensors, with torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests).
"""
import torch


def shard_gops(n_gops, rank, world):
    return list(range(rank, n_gops, world))


def gather_gop_metrics(local, n_gops, gop, dist=None, device="cpu"):
    """local: {gop_index: (bits[gop], psnr[gop])} coded by this rank -> on every rank two [n_gops, gop] float64
    tensors in GOP order."""
    bits = torch.zeros(n_gops, gop, dtype=torch.float64, device=device)
    psnr = torch.zeros(n_gops, gop, dtype=torch.float64, device=device)
    for g, (b, p) in local.items():
        bits[g] = torch.as_tensor(b, dtype=torch.float64)
        psnr[g] = torch.as_tensor(p, dtype=torch.float64)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        # each GOP is owned by exactly one rank and zero elsewhere: a sum reassembles the table
        dist.all_reduce(bits, op=dist.ReduceOp.SUM)
        dist.all_reduce(psnr, op=dist.ReduceOp.SUM)
    return bits, psnr
