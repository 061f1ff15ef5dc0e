"""Sharding of one MultiExp over the GPUs of a node (one process per GPU, torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" in the CPU tests).  Two decompositions, both ending in ONE all-gather of a few XYZZ points:

* windows: window w is owned by rank w % world, every rank holds all bases (below);
* points:  rank r owns points [r*n/world, (r+1)*n/world) and computes all windows of its slice; window w of the whole
  MultiExp is the sum of the ranks' totals for w (gmsm_fold_window_sets).  No replication of the bases, perfect balance;
  measured per-rank cost on MI355X (tools/shard_model.py, BN254 G1, profiles/r03_shard_model.log): 2^24 on 8 ranks 3.27 ms
  against 3.93 ms for the window decomposition; at 2^20 the window decomposition is ahead (0.53 / 0.60 ms) - see
  `choose_mode`.  (The same two decompositions exist INSIDE libgmsm.so for a single process that drives all devices:
  gmsm_multiexp_sharded, include/gmsm.h.)

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


# ---------------------------------------------------------------- device-resident exchange + point sharding
def point_slice(n, rank, world):
    """Points [lo, hi) owned by `rank` in the point decomposition."""
    return rank * n // world, (rank + 1) * n // world


def choose_mode(n, world):
    """Measured per-rank cost (tools/shard_model.py, BN254 G1, ms; windows / points; round 3):
         2^20:  N=2 1.10 / 1.11   N=4 0.72 / 0.76   N=8 0.53 / 0.60     (a rank with few windows runs a shorter
         2^24:  N=2 12.5 / 11.5   N=4 6.93 / 6.02   N=8 3.93 / 3.27      reduction; its fold needs no per-rank sums)
    points once a rank's slice reaches 2^20 points (no replicated base rewrite / decomposition), windows below."""
    return "points" if n // max(1, world) >= (1 << 20) else "windows"


def shard_plan(group, n, rank, world, mode="auto", c=None):
    """Everything a rank needs to know about its piece: returns a dict with mode, c, nwin, the point range [lo, hi), the
    window range (win_first, win_stride) and `rows` = window totals each rank contributes to the all-gather."""
    if mode == "auto":
        mode = choose_mode(n, world)
    if mode == "points":
        biggest = max(point_slice(n, r, world)[1] - point_slice(n, r, world)[0] for r in range(world))
        c = c or group.default_window_bits(max(1, biggest))  # one c for every slice: the totals must line up
        nwin = group.num_windows(c)
        lo, hi = point_slice(n, rank, world)
        return dict(mode=mode, c=c, nwin=nwin, lo=lo, hi=hi, win_first=0, win_stride=1, rows=nwin)
    if mode != "windows":
        raise ValueError("mode must be auto, windows or points")
    c = c or group.default_window_bits(n)
    nwin = group.num_windows(c)
    return dict(mode=mode, c=c, nwin=nwin, lo=0, hi=n, win_first=rank, win_stride=world, rows=slots_per_rank(nwin, world))


class Exchange:
    """The one collective of a sharded MultiExp, on buffers that live where the totals are produced: `local`
    (rows x xyzz_limbs int64; the engine writes this rank's window totals straight into it with
    gmsm_window_sums_enqueue) is all-gathered into `gathered`; only the gathered block crosses to the host."""

    def __init__(self, dist, device, rows, xyzz_limbs):
        import torch
        self.dist, self.world = dist, dist.get_world_size()
        self.rows, self.limbs = rows, xyzz_limbs
        self.local = torch.zeros((rows, xyzz_limbs), dtype=torch.int64, device=device)
        # A gloo process group (the CPU tests; bench.py --oversubscribe, where several ranks share one GPU and RCCL refuses
        # duplicate devices) gathers on the host: the totals cross to the host for the fold anyway.
        self.via_host = device.type != "cpu" and dist.get_backend() == "gloo"
        self.gathered = torch.zeros((self.world * rows, xyzz_limbs), dtype=torch.int64, device="cpu" if self.via_host else device)

    def gather(self):
        """-> (world, rows, xyzz_limbs) uint64 on the host, identical on every rank."""
        if self.via_host:
            self.dist.all_gather_into_tensor(self.gathered, self.local.cpu())  # .cpu() waits for the producer (current stream)
            return self.gathered.numpy().view(np.uint64).reshape(self.world, self.rows, self.limbs)
        self.dist.all_gather_into_tensor(self.gathered, self.local)  # ordered after the producer on the current stream
        return self.gathered.cpu().numpy().view(np.uint64).reshape(self.world, self.rows, self.limbs)


def sharded_multiexp_exchange(group, plan, enqueue_fn, exchange):
    """One sharded MultiExp with device-resident totals.

    plan         from shard_plan
    enqueue_fn   (plan, local_tensor) -> None: makes this rank's totals appear in local_tensor[:nlocal] in stream order
                 (GPU: gmsm_window_sums_enqueue into local_tensor.data_ptr(); the CPU tests fill it from the oracle)
    Returns the Jacobian result (identical on every rank)."""
    enqueue_fn(plan, exchange.local)
    gathered = exchange.gather()
    if plan["mode"] == "points":
        return group.fold_window_sets(gathered, plan["c"])
    return group.fold_windows(unpack_gathered(gathered, plan["nwin"], exchange.world, group.xyzz_limbs), plan["c"])


# ---------------------------------------------------------------- replicas: k MultiExp over the same bases on N ranks
# The callers that issue many MultiExp over one SRS (one kzg.Commit per polynomial, ecc/bn254/kzg/kzg.go:159-176;
# BatchOpenSinglePoint :246; the folded commitments of BatchVerifyMultiPoints :405-520) need no sharding of a single
# MSM at all: every rank registers the bases once, scalar vector j belongs to rank j % world, each rank runs its vectors
# through gmsm_multiexp_bases_batch (two in flight), and the only collective is ONE all-gather of the k Jacobian results.
def batch_owner(j, world):
    return j % world


def batch_slots(k, world):
    return (k + world - 1) // world


def owned_vectors(k, rank, world):
    return list(range(rank, k, world))


def replicated_batch(k, rank, world, jac_limbs, local_batch_fn, all_gather_fn):
    """k independent MultiExp, vector j on rank j % world.

    local_batch_fn   list of owned vector indices -> (len, jac_limbs) uint64 Jacobian results of those vectors
                     (GPU: ResidentBases.MultiExpBatch over this rank's copy of the bases; CPU tests: the oracle)
    all_gather_fn    (slots, jac_limbs) uint64 -> (world, slots, jac_limbs) uint64, identical on every rank
    Returns the (k, jac_limbs) results in vector order on every rank."""
    mine = owned_vectors(k, rank, world)
    slots = batch_slots(k, world)
    local = np.zeros((slots, jac_limbs), dtype=np.uint64)
    if mine:
        res = np.asarray(local_batch_fn(mine), dtype=np.uint64).reshape(len(mine), jac_limbs)
        local[: len(mine)] = res
    gathered = np.asarray(all_gather_fn(local), dtype=np.uint64).reshape(world, slots, jac_limbs)
    out = np.zeros((k, jac_limbs), dtype=np.uint64)
    for j in range(k):
        out[j] = gathered[batch_owner(j, world), j // world]
    return out
