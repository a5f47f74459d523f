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


class SharedGpuError(RuntimeError):
    """Fewer distinct physical GPUs than ranks (and sharing was not asked for)."""


def gpu_key(identity):
    """What makes two ranks 'the same GPU': the UUID when the runtime reports one, else the PCI function."""
    if isinstance(identity, dict):
        u = identity.get("uuid")
        return u if u and u != "?" else identity.get("pci", "?")
    return str(identity)


def check_distinct_gpus(identities, allow_shared=False):
    """identities: one entry per rank (dicts from cufhe_amd.api.device_identity, or strings).  Returns
    (distinct_gpus, shared).  A run whose ranks share a GPU measures nothing about N GPUs: it is refused
    (SharedGpuError) unless allow_shared, and then the caller must label its line `shared_gpu` and report
    n_gpus = distinct_gpus."""
    keys = [gpu_key(i) for i in identities]
    distinct = len(set(keys))
    shared = distinct < len(keys)
    if shared and not allow_shared:
        dup = sorted({k for k in keys if keys.count(k) > 1})
        raise SharedGpuError("%d ranks on %d distinct GPU(s) (shared: %s): this would print a multi-GPU line measured on fewer "
                             "GPUs; pass --allow-shared-gpu to rehearse on this box" % (len(keys), distinct, ", ".join(dup)))
    return distinct, shared


def gather_objects(obj, dist=None):
    """all_gather_object over the default process group (gloo): [obj of rank 0, obj of rank 1, ...]"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def rank_summary(reports, allow_shared=False):
    """reports: per-rank dicts {rank, gpu: identity, value, ms_per_step, ...} as gathered on rank 0.  Returns the
    fields every multi-rank bench line carries: per_rank, distinct_gpus, value_min_rank / value_max_rank and, when
    ranks share a GPU (allowed explicitly), shared_gpu.  Raises SharedGpuError otherwise."""
    distinct, shared = check_distinct_gpus([r["gpu"] for r in reports], allow_shared)
    vals = [r["value"] for r in reports if r.get("value") is not None]
    out = {"per_rank": reports, "distinct_gpus": distinct,
           "value_min_rank": min(vals) if vals else None, "value_max_rank": max(vals) if vals else None}
    if shared:
        out["shared_gpu"] = True
    return out


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(child_cmd, world, env=None, timeout=None, poll=0.05):
    """Self-launcher for `bench.py --gpus N` run plainly (no torch.distributed.run around it): start
    `world` child processes of `child_cmd`, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment), wait for all of them, and return
    (exit_code, stdout_of_rank_0).  The caller must not have touched the GPU: children are fresh
    processes, nothing is exec'd over a process that initialised HIP.  A rank that fails (or the
    timeout) ends the others -- they would wait for it in the barrier forever -- by their exact
    process groups (each rank is started as the leader of its own session, so whatever a rank spawned goes with it and
    cannot keep rank 0's stdout pipe open), and the launcher reports a non-zero exit code.  If starting a rank fails
    (ENOMEM, EAGAIN) the ranks already running are ended before the error is re-raised.  Ranks other than 0 inherit stderr, their
    stdout is dropped (rank 0 prints the one JSON line)."""
    import subprocess
    import time
    base = dict(os.environ if env is None else env)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base["MASTER_ADDR"] = "127.0.0.1"
    base["MASTER_PORT"] = str(free_port())
    base["WORLD_SIZE"] = str(world)
    base["LOCAL_WORLD_SIZE"] = str(world)
    procs = []

    def end_all(which):
        """terminate, then kill, the given ranks AND what they started: every rank leads its own process group"""
        import signal
        for r in which:
            try:
                os.killpg(procs[r].pid, signal.SIGTERM)
            except (ProcessLookupError, PermissionError):
                pass
        deadline = time.monotonic() + 10
        for r in which:
            try:
                procs[r].wait(max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                pass
            try:
                os.killpg(procs[r].pid, signal.SIGKILL)       # whatever is left of the group (a grandchild holding the pipe)
            except (ProcessLookupError, PermissionError):
                pass
            procs[r].wait()

    # The ranks lead their own sessions, so a Ctrl-C or a SIGTERM aimed at the launcher (a harness timeout) does not reach them by
    # itself: from BEFORE the first rank is started until the last one has been reaped SIGTERM is turned into an exception, and ANY
    # exception ends the ranks that exist by then before it goes on.  While end_all runs SIGTERM is ignored: a second signal must
    # not abort the clean-up halfway.
    import signal

    class _Terminated(BaseException):
        pass

    def _on_term(signum, frame):
        raise _Terminated()
    old_term = None
    try:
        old_term = signal.signal(signal.SIGTERM, _on_term)
    except ValueError:          # not the main thread: signals go to the main thread, nothing to install
        old_term = None

    def end_all_quietly(which):
        prev = None
        try:
            prev = signal.signal(signal.SIGTERM, signal.SIG_IGN)
        except ValueError:
            prev = None
        try:
            end_all(which)
        finally:
            if prev is not None:
                signal.signal(signal.SIGTERM, prev)

    pending = set()
    try:
        for r in range(world):
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen(list(child_cmd), env=e, stdin=subprocess.DEVNULL, start_new_session=True,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
            pending.add(r)
        # rank 0's stdout is drained by a thread so that a long line can never block the child
        import threading
        chunks = []
        reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        t0 = time.monotonic()
        rc = 0
        while pending:
            for r in list(pending):
                code = procs[r].poll()
                if code is not None:
                    pending.discard(r)
                    if code != 0 and rc == 0:
                        rc = code if code > 0 else 128 - code
            if rc != 0 or (timeout is not None and time.monotonic() - t0 > timeout):
                if rc == 0:
                    rc = 124
                end_all_quietly(sorted(pending))        # kills each group BEFORE reaping its leader: the pgid cannot have been reused
                pending = set()
            if pending:
                time.sleep(poll)
    except BaseException:       # incl. an OSError while starting a rank (ENOMEM, EAGAIN) and a signal during the spawn loop
        end_all_quietly(sorted(pending))        # the ranks already started would wait in the barrier forever
        raise
    finally:
        if old_term is not None:
            signal.signal(signal.SIGTERM, old_term)
    # Every rank has been reaped (poll / wait above), so its pid -- and with it the pgid -- may already belong to somebody else: no
    # signal is sent by number any more.  A grandchild a rank left behind cannot hold the job up either: the reader below gives up
    # after 10 s and returns what rank 0 wrote.
    reader.join(10)
    out = chunks[0].decode(errors="replace") if chunks else ""
    return rc, out
