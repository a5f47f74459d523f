"""One-process-per-GPU plumbing for bench.py: rank environment, GPU pinning, gate sharding.

The gate path has no exchange step: ranks only meet in a barrier and in a MAX
reduction of the elapsed time.  No HIP call is made here."""
import os


def rank_env(env=None):
    env = os.environ if env is None else env
    return int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")), int(env.get("WORLD_SIZE", "1"))


def visible_device_for(local_rank, current=None):
    """HIP_VISIBLE_DEVICES value that makes this rank's GPU the only visible one."""
    if current:
        ids = [v for v in current.split(",") if v != ""]
        return ids[local_rank % len(ids)]
    return str(local_rank)


def pin_gpu(env=None):
    """Set HIP_VISIBLE_DEVICES for this rank (call before anything touches HIP).  Alternative to
    device_for_rank() for launchers that prefer each rank to see a single device."""
    env = os.environ if env is None else env
    rank, local_rank, world = rank_env(env)
    if world > 1:
        env["HIP_VISIBLE_DEVICES"] = visible_device_for(local_rank, env.get("HIP_VISIBLE_DEVICES"))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return rank, local_rank, world


def device_for_rank(local_rank, visible_count):
    """Physical device index for this rank when all of the node's GPUs are visible to every
    rank (what torch.distributed.run gives): local_rank, wrapped if fewer devices are visible."""
    if visible_count <= 0:
        raise RuntimeError("no GPU visible")
    return local_rank % visible_count


def shard(count, rank, world):
    """Contiguous split of `count` gates: gate i -> rank floor(i / ceil(count / world))."""
    per = -(-count // world)
    lo = min(count, rank * per)
    return lo, min(count, lo + per)


def max_over_ranks(value, dist=None):
    """MAX all-reduce of a python float over the default process group (gloo or nccl)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
