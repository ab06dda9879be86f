"""Keyframe sharding for the multi-GPU path.

K1-K3 (SemiDenseRecon, PM.cc:137-256) of keyframe k read only immutable inputs of k and its
covisible neighbours; K4 (InterKeyFrameDepthChecking, PM.cc:628-799) needs the neighbours'
FINISHED {rho, sigma} maps (the reference gates on that at PM.cc:292-298).  So keyframes shard as
contiguous blocks, one block per GPU, with exactly one exchange step between K3 and K4: an
all-gather of the per-keyframe {rho, sigma} maps (RCCL over xGMI through torch.distributed).
Slot numbering is GLOBAL (slot == keyframe index) on every rank, so the gathered pool needs no
re-indexing and the depth pool is all-gathered in place.
"""
import torch
import torch.distributed as dist


def block_partition(n_total, world, rank):
    """Contiguous equal blocks; n_total must divide evenly (all-gather needs equal shards)."""
    if n_total % world:
        raise ValueError("n_total (%d) must be a multiple of world size (%d)" % (n_total, world))
    count = n_total // world
    return rank * count, count


def plan(n_total, world, rank, n_nbr, neighbours_fn):
    """Returns dict(first, count, own=[...], nbrs=[[...]], inputs=sorted slots whose IMAGES this
    rank must hold = own block + its neighbours (the input halo))."""
    first, count = block_partition(n_total, world, rank)
    own = list(range(first, first + count))
    nbrs = [list(neighbours_fn(k, n_total, n_nbr)) for k in own]
    need = set(own)
    for row in nbrs:
        need.update(row)
    return dict(first=first, count=count, own=own, nbrs=nbrs, inputs=sorted(need))


def allgather_depth(pool, first, count, group=None):
    """In-place all-gather of the depth pool: `pool` is the [n_total, H, W, 2] float32 tensor that
    backs the engine's depth pool (sdm_config.ext_depth_pool); this rank has just written rows
    [first, first+count).  After the call every rank holds every keyframe's {rho, sigma}."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    mine = pool[first:first + count]
    dist.all_gather_into_tensor(pool, mine, group=group)
