"""
Data-parallel helpers (new: the reference is single-process).

Training is plain data parallelism over windows: every rank holds a full replica (5.9 M fp32 parameters), computes
its own loss/gradients and the gradients are averaged with RCCL all-reduces over flat buckets after the backward
pass.  The model object is driven through `net(batch)` / `net.backward(batch, out)` like the reference's training
loop (scripts/train.py:146-150) rather than through a wrapped `forward`, so the averaging is an explicit call instead
of DistributedDataParallel hooks.  BatchNorm statistics stay per-rank, as in the reference (there is no SyncBN).
"""
import torch


LAUNCH_BACKEND_ENV = 'EMPOSE_DIST_BACKEND'   # 'gloo' = rendezvous without devices (the CPU test of the launcher)


def launched_by_a_launcher():
    """True inside a rank process (torch.distributed.run or `launch_ranks` set RANK and WORLD_SIZE)."""
    import os
    return 'RANK' in os.environ and 'WORLD_SIZE' in os.environ


def dist_backend(device=None):
    import os
    forced = os.environ.get(LAUNCH_BACKEND_ENV)
    if forced:
        return forced
    return 'nccl' if (device is not None and device.type == 'cuda') else 'gloo'


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(script, argv, n, poll_s=0.2):
    """
    Run `script argv` as `n` rank processes of ONE node, one per GPU -- what `python -m torch.distributed.run --nnodes=1
    --nproc-per-node n script argv` does, without needing the caller to type it: RANK / LOCAL_RANK / WORLD_SIZE /
    LOCAL_WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT (a free port) in each child's environment, stdout / stderr
    inherited (rank 0 is the one that prints results).  Fresh interpreters (no fork: the parent may already hold a HIP
    context).  If a rank exits non-zero the others are terminated (by PID) and its code is returned; otherwise 0.
    """
    import os
    import subprocess
    import sys
    import time
    port = _free_port()
    cores = os.cpu_count() or n
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # (the host driver only supports dmabuf IPC; a user's own setting wins)
        env.setdefault('OMP_NUM_THREADS', str(max(1, cores // n)))
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env))
    rc, clean = 0, False
    try:
        live = list(procs)
        while live and rc == 0:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0:
                    rc = code
                    break
            else:
                time.sleep(poll_s)
        clean = True
    finally:
        # a rank failed, or this process is being interrupted (KeyboardInterrupt, SystemExit from a signal handler): the
        # ranks still running are stopped -- by PID -- before the status (or the exception) goes up
        for p in procs:
            if p.poll() is None and (rc != 0 or not clean):
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20 if (rc != 0 or not clean) else None)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc


def maybe_self_launch(script, n, force=False, what='this script'):
    """
    `python script --gpus n` as a plain command: when no launcher set up the ranks, become the launcher.  Returns
    normally inside a rank process (or when one un-distributed process is all that is asked for); otherwise spawns the
    ranks, waits and exits with their status.  With fewer than `n` visible GPUs: exit non-zero, "needs n GPUs, found m"
    (unless EMPOSE_DIST_BACKEND=gloo asks for a device-less rendezvous, the launcher's CPU test).
    """
    import os
    import sys
    if launched_by_a_launcher() or (n <= 1 and not force):
        return
    if os.environ.get(LAUNCH_BACKEND_ENV) != 'gloo' and os.environ.get(SHARE_DEVICES_ENV) != '1':
        found = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if found < n:
            raise SystemExit('{} --gpus {} needs {} GPUs, found {}'.format(what, n, n, found))
    sys.stdout.flush()
    sys.exit(launch_ranks(script, sys.argv[1:], n))


SHARE_DEVICES_ENV = 'EMPOSE_SHARE_DEVICES'   # '1': several ranks may sit on one GPU (single-GPU self-test of the N > 1 paths)


def local_device_index(local_rank, what='this script'):
    """The rank's GPU.  One rank per GPU; with fewer GPUs than ranks: exit "needs N GPUs, found M" -- unless
    EMPOSE_SHARE_DEVICES=1 (together with EMPOSE_DIST_BACKEND=gloo: RCCL refuses two ranks on one device), which wraps
    the ranks around the devices there are: the multi-rank logic (sharding, reductions, the result line) then runs on
    real device code on a one-GPU box."""
    import os
    n = torch.cuda.device_count()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if os.environ.get(SHARE_DEVICES_ENV) == '1' and n > 0:
        if world > n:
            # Kernels whose workgroups wait for each other need the device to themselves: every workgroup of a launch must be
            # resident at once.  The one-launch training layers fill the chip (256 workgroups at width 512); two processes
            # launching them on ONE device starve each other until the polls give up (EMPOSE_ETIMEOUT, outputs NaN).  Ranks
            # that share a device therefore take the layer-by-layer path.
            import logging
            from em_pose_amd import _lib
            _lib.check(_lib.lib().empose_set_option(b'train_cols', 0))
            logging.getLogger('em_pose_amd.distributed').info(
                '%d ranks share %d device(s): option train_cols -> 0 for this process', world, n)
        return local_rank % n
    if n <= local_rank:
        raise SystemExit('{} --gpus {} needs {} GPUs, found {}'.format(what, world, world, n))
    return local_rank


def collective_device(device):
    """Where tensors must live for collectives of the active backend (gloo: the host)."""
    import torch.distributed as dist
    return device if (dist.is_initialized() and dist.get_backend() == 'nccl') else torch.device('cpu')


def results_stream():
    """In a rank process: a file object on the REAL stdout, after pointing file descriptor 1 at stderr.  Libraries that
    write banners to stdout from native code (RCCL prints its version block there when the first communicator comes up)
    then land on stderr, and the one JSON line rank 0 prints is the only thing on stdout."""
    import os
    import sys
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    return os.fdopen(real, 'w')


def init_process_group(device=None, log=None):
    """Rank-side: join the process group the launcher described in the environment (RCCL with the rank's device, or
    gloo).  Returns (dist module, rank, world)."""
    import os
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    backend = dist_backend(device)
    kwargs = {'device_id': device} if backend == 'nccl' else {}
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, **kwargs)
    rank, world = dist.get_rank(), dist.get_world_size()
    if log is not None:
        log('rank {}/{} joined the process group over {}'.format(rank, world, backend))
    return dist, rank, world


def init_from_env(device=None, backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (127.0.0.1 by default). Returns (rank, world)."""
    import os
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if (device is not None and device.type == 'cuda') else 'gloo'
        kwargs = {'device_id': device} if backend == 'nccl' else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world


def _bucket_plan(params, bucket_bytes):
    buckets, cur, cur_bytes = [], [], 0
    for p in params:
        cur.append(p)
        cur_bytes += p.numel() * p.element_size()
        if cur_bytes >= bucket_bytes:
            buckets.append(cur)
            cur, cur_bytes = [], 0
    if cur:
        buckets.append(cur)
    return buckets


class GradientBuckets(object):
    """
    Persistent flat gradient buckets with the all-reduce overlapped with the rest of the reverse sweep.

    * The parameters are laid out, in the order given (= the order in which the reverse sweep finishes their gradients:
      update networks first, then the initial estimate's heads, the LSTM last), into flat fp32 buffers of about
      `bucket_bytes` that live as long as this object.  `view_of(p)` is parameter p's slice of its bucket: the training
      engine writes dW / db straight into it and installs it as `p.grad` (nn/train_engine.py), so nothing is
      concatenated or copied, before or after the collective, and `.grad` has the same address every step.
    * `stage(p)` says "p's gradient is final".  When the last parameter of a bucket is staged, the bucket's all-reduce
      is enqueued on a side stream behind an event recorded on the compute stream -- it runs while the compute stream
      continues with back-propagation through time (RCCL ring all-reduce over xGMI is per-link bound: a few multi-MB
      messages, not one per tensor).  `finish()` makes the compute stream wait for the collectives (and stages whatever
      was not staged explicitly, so it is also the whole story for a caller without hooks).
    * A gradient that was produced elsewhere (autograd's own tensors on the fallback path) is copied into its slice
      when staged; `p.grad` is re-pointed to the slice.
    Averages over the ranks (sum, then 1 / world on the side stream).  `force=True` runs the collectives even with one
    rank (self-test of the RCCL path on one GPU).
    A parameter that has no gradient on this rank when it is staged gets a zeroed slice as `.grad` (every rank must issue
    the same collectives, and another rank may have a gradient for it): in a distributed run such parameters are stepped
    by the optimiser with the averaged (possibly zero) gradient -- `HipAdam`'s "skipped like torch" applies to `.grad is
    None`, which only a single-process run leaves in place.
    """

    def __init__(self, parameters, bucket_bytes=8 << 20, group=None, force=False):
        import torch.distributed as dist
        self.params = [p for p in parameters if p.requires_grad]
        if not self.params:
            raise ValueError('no parameters')
        self.group, self.force = group, force
        self.active = dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1)
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        dev = self.params[0].device
        plan = _bucket_plan(self.params, bucket_bytes)
        self.flat, self._slot, self._members = [], {}, []
        for members in plan:
            # every slice starts on a 256-byte boundary (the kernels store gradients with 16-byte accesses); the
            # padding stays zero and rides along in the collective
            pad = lambda n: (n + 63) & ~63
            flat = torch.zeros(sum(pad(p.numel()) for p in members), dtype=self.params[0].dtype, device=dev)
            off = 0
            for p in members:
                self._slot[id(p)] = (len(self.flat), flat[off:off + p.numel()].view_as(p))
                off += pad(p.numel())
            self.flat.append(flat)
            self._members.append(members)
        self.n_buckets = len(self.flat)
        self.side = torch.cuda.Stream(device=dev) if dev.type == 'cuda' else None
        self._staged = [set() for _ in self.flat]
        self._ready = [[] for _ in self.flat]   # per bucket: events recorded where its members' gradients became final
        self._launched = [False] * self.n_buckets
        self._pending = []
        self.hold = False   # True: stage() is a no-op (a step being captured into a HIP graph must not enqueue
                            # collectives; finish() after the replay stages everything)

    def view_of(self, p):
        slot = self._slot.get(id(p))
        return None if slot is None else slot[1]

    def _launch(self, b):
        import torch.distributed as dist
        self._launched[b] = True
        if not self.active:
            return
        flat = self.flat[b]
        if self.side is None:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(self.world)
            return
        # The bucket's gradients are complete where each member was staged -- not necessarily on ONE stream: the training
        # engine finishes the two update networks' weight gradients on two side streams, and a bucket boundary can fall
        # inside either network.  The collective waits for every member's event (stage() records them).
        ready = self._ready[b] + [torch.cuda.Event()]
        ready[-1].record()
        self._ready[b] = []
        with torch.cuda.stream(self.side):
            for ev in ready:
                self.side.wait_event(ev)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            if self.world > 1:
                flat.div_(self.world)
            done = torch.cuda.Event()
            done.record()
        self._pending.append(done)

    def stage(self, p):
        slot = self._slot.get(id(p))
        if slot is None or self.hold:
            return
        b, view = slot
        if p.grad is None:
            view.zero_()                          # every rank issues the same collectives
            p.grad = view
        elif p.grad.data_ptr() != view.data_ptr():
            view.copy_(p.grad)
            p.grad = view
        self._staged[b].add(id(p))
        if self.side is not None and self.active:
            ev = torch.cuda.Event()
            ev.record()                           # on the stream that produced (or just copied / zeroed) this gradient
            self._ready[b].append(ev)
        if not self._launched[b] and len(self._staged[b]) == len(self._members[b]):
            self._launch(b)

    def finish(self):
        """Stage what is left, wait (stream-side, not host-side) for every collective; ready for the next step."""
        self.hold = False
        for b, members in enumerate(self._members):
            for p in members:
                if id(p) not in self._staged[b]:
                    self.stage(p)
        for done in self._pending:
            torch.cuda.current_stream().wait_event(done)
        self._pending = []
        self._staged = [set() for _ in self.flat]
        self._ready = [[] for _ in self.flat]
        n, self._launched = sum(self._launched), [False] * self.n_buckets
        return n if self.active else 0


def attach_gradient_buckets(net, buckets):
    """The training engine of `net` writes gradients into `buckets` and stages each parameter group as the reverse sweep
    finishes it (None detaches).

    With buckets the collectives of finished groups run on a side stream BESIDE the rest of the sweep.  The one-launch
    training layers (csrc/train_cols.hip) need every workgroup of a launch resident at once; a collective kernel that holds
    compute units while it waits for a slower rank would leave such a launch half resident, spinning on its mailbox until
    the polls give up (EMPOSE_ETIMEOUT).  No deadlock -- the collectives do not depend on it -- but how long the wait lasts
    is the other ranks' business, so the overlapped sweep takes the layer-by-layer path (option "train_cols" = 0).  Options
    are process-wide: the value found is remembered on `net` and put back when the buckets are detached (None), and the
    change is logged.  (Eager launches of those layers are cooperative launches since round 6 -- the runtime guarantees
    residency -- but a captured step replays ordinary launches, and the engine captures whenever shapes repeat.)"""
    import logging
    from em_pose_amd import _lib
    lib = _lib.lib()
    log = logging.getLogger('em_pose_amd.distributed')
    if buckets is not None:
        if getattr(net, '_grad_sink', None) is None:
            net._train_cols_before = int(lib.empose_get_option(b'train_cols'))
        _lib.check(lib.empose_set_option(b'train_cols', 0))
        if net._train_cols_before != 0:
            log.info('gradient buckets attached: option train_cols %d -> 0 until they are detached', net._train_cols_before)
    elif getattr(net, '_grad_sink', None) is not None and hasattr(net, '_train_cols_before'):
        _lib.check(lib.empose_set_option(b'train_cols', net._train_cols_before))
        log.info('gradient buckets detached: option train_cols back to %d', net._train_cols_before)
        del net._train_cols_before
    net._grad_sink = buckets


_ONE_SHOT_BUCKETS = {}


def drop_one_shot_buckets():
    """Forget the buckets `allreduce_gradients` keeps for the last parameter list (they hold strong references to its
    parameters and flat gradient buffers): call it when that model is done with."""
    _ONE_SHOT_BUCKETS.clear()


def allreduce_gradients(parameters, bucket_bytes=8 << 20, group=None):
    """
    Average `.grad` of the given parameters over all ranks after the backward pass, in flat buckets of about
    `bucket_bytes`.  One-shot form of `GradientBuckets` (which overlaps the collectives with the reverse sweep); the
    buckets (and their side stream) are built once per parameter list and kept, so `.grad` has the same address from the
    second step on and an optimiser that caches gradient pointers (`HipAdam`) does not re-upload its table every step.
    Parameters without a gradient contribute zeros so that every rank issues the same collectives -- and therefore HAVE a
    (zero) gradient afterwards: in a distributed run the optimiser steps them (moments decay, step count advances), unlike
    a single-process run where a parameter whose `.grad` is None is skipped.  All ranks agree, which is what matters.
    """
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    params = [p for p in parameters if p.requires_grad]
    key = (tuple(id(p) for p in params), bucket_bytes, id(group))
    buckets = _ONE_SHOT_BUCKETS.get(key)
    if buckets is None or any(b.device != params[0].device for b in buckets.flat):
        _ONE_SHOT_BUCKETS.clear()     # (one parameter list at a time: a new model replaces the old buckets)
        buckets = _ONE_SHOT_BUCKETS[key] = GradientBuckets(params, bucket_bytes, group)
    return buckets.finish()


def shard_range(n_items, rank, world):
    """Contiguous, balanced [start, end) of `n_items` for `rank` (windows of a batch, BASELINE configs 2/3)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)
