"""Window sharding of one MultiExp over the GPUs of a node (one process per GPU, torch.distributed; backend "nccl"
is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference runs one goroutine per c-bit window and collects one g1JacExtended per window on a channel
(ecc/bn254/multiexp.go:148-209); here window w is owned by rank w % world, every rank holds all bases, and the only
exchange is one all-gather of ceil(nwin/world) window totals per rank (128 B each for BN254 G1) followed by the same
Horner fold (multiexp.go:302-315) on every rank.  EC points cannot be reduced by RCCL (no user-defined reduction), so
it is gather-then-fold, never all-reduce.
"""
import numpy as np


def owned_windows(nwin, rank, world):
    return list(range(rank, nwin, world))


def slots_per_rank(nwin, world):
    return (nwin + world - 1) // world


def pack_local(local_xyzz, nwin, world, xyzz_limbs):
    """Pad this rank's window totals to the common slot count (ranks may own one window fewer)."""
    per = slots_per_rank(nwin, world)
    buf = np.zeros((per, xyzz_limbs), dtype=np.uint64)
    buf[: local_xyzz.shape[0]] = local_xyzz
    return buf


def unpack_gathered(gathered, nwin, world, xyzz_limbs):
    """gathered[rank][slot] -> totals[w] with w = slot*world + rank."""
    gathered = np.asarray(gathered, dtype=np.uint64).reshape(world, -1, xyzz_limbs)
    totals = np.zeros((nwin, xyzz_limbs), dtype=np.uint64)
    for w in range(nwin):
        totals[w] = gathered[w % world, w // world]
    return totals


def sharded_multiexp(group, window_sums_fn, c, rank, world, all_gather_fn):
    """One window-sharded MultiExp.

    group            a gnark-crypto_amd.multiexp._Group (supplies num_windows / fold_windows)
    window_sums_fn   (c, win_first, win_stride) -> (nlocal, xyzz_limbs) uint64 window totals of this rank
    all_gather_fn    (per, xyzz_limbs) uint64 array -> (world, per, xyzz_limbs) uint64 array (same on every rank)
    Returns the Jacobian result (identical on every rank)."""
    nwin = group.num_windows(c)
    local = window_sums_fn(c, rank, world)
    assert local.shape[0] == len(owned_windows(nwin, rank, world))
    gathered = all_gather_fn(pack_local(local, nwin, world, group.xyzz_limbs))
    return group.fold_windows(unpack_gathered(gathered, nwin, world, group.xyzz_limbs), c)


def torch_all_gather(dist, device):
    """all_gather_fn over torch.distributed (RCCL when device is a GPU, gloo on CPU)."""
    import torch

    def fn(buf):
        world = dist.get_world_size()
        t_in = torch.from_numpy(np.ascontiguousarray(buf).view(np.int64)).to(device)
        # concatenation along dim 0 is the one output layout both RCCL and gloo accept
        t_out = torch.empty((world * t_in.shape[0],) + tuple(t_in.shape[1:]), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(t_out, t_in)
        return t_out.cpu().numpy().view(np.uint64).reshape((world,) + tuple(buf.shape))
    return fn
