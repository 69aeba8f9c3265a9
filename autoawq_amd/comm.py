"""One-shot small-message all-reduce for tensor-parallel decode (awq_allreduce_oneshot, csrc/allreduce.hip).

The reference has no distributed code; this is the collective SURVEY.md 8(e) plans for the [M, hidden] fp16
outputs of the row-parallel o / down projections: one launch per rank, no host involvement, hipGraph-capturable -- which
`torch.distributed.all_reduce` (RCCL) is not guaranteed to be -- so a whole decode step stays ONE graph under TP.

    ar = OneShotAllReduce.from_process_group(max_halfs=M * hidden)   # once, collectively: allocates + exchanges IPC handles
    y = ar(y)                                                        # in place, on the current stream

`OneShotAllReduce.local_group(P, ...)` builds P "ranks" inside one process on one GPU (each rank's buffers are ordinary
local allocations, the kernels run on P streams): what the single-GPU tests drive.
"""
import ctypes

import torch

from . import _lib


class _DeviceBytes:
    """`nbytes` of zeroed UNCACHED fine-grained device memory from awq_allreduce_alloc (flags and staging are polled by their
    owner while peer GPUs write them: ordinary coarse-grained memory gives a spinning kernel no guarantee to ever see the store),
    or this process's mapping of a peer's buffer (awq_allreduce_ipc_open).  Handed to torch through
    `__cuda_array_interface__`; freed / unmapped when the last tensor over it is gone."""

    def __init__(self, nbytes=None, handle=None):
        L = _lib.lib()
        p = ctypes.c_void_p()
        self._peer = handle is not None
        if self._peer:
            buf = ctypes.create_string_buffer(bytes(handle), len(handle))
            _lib.check(L.awq_allreduce_ipc_open(buf, ctypes.byref(p)), "awq_allreduce_ipc_open")
        else:
            _lib.check(L.awq_allreduce_alloc(ctypes.byref(p), nbytes), "awq_allreduce_alloc")
        self.ptr, self.nbytes = p.value, int(nbytes)
        self.__cuda_array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2}

    def tensor(self, device):
        return torch.as_tensor(self, device=device)

    def ipc_handle(self):
        buf = ctypes.create_string_buffer(64)
        _lib.check(_lib.lib().awq_allreduce_ipc_export(ctypes.c_void_p(self.ptr), buf), "awq_allreduce_ipc_export")
        return bytes(buf.raw)

    def __del__(self):
        try:
            if self.ptr:
                L = _lib.lib()
                (L.awq_allreduce_ipc_close if self._peer else L.awq_allreduce_free)(ctypes.c_void_p(self.ptr))
                self.ptr = None
        except Exception:  # interpreter shutdown
            pass


class OneShotSetupError(RuntimeError):
    """Raised by EVERY rank of the group when the setup of ANY rank failed (the ranks agree before anybody returns)."""


class OneShotAllReduce:
    CHECK_EVERY = 256  # calls between two looks at the sticky error word (asynchronous copy; `check()` forces one)

    def __init__(self, rank, world, staging, flags, state, max_halfs, keepalive=None):
        """staging / flags: lists of `world` uint8 tensors (entry `rank` is this rank's own, the others are peer mappings)."""
        assert 1 <= world <= 8 and 0 <= rank < world and len(staging) == world and len(flags) == world
        self.rank, self.world, self.max_halfs = rank, world, int(max_halfs)
        self.staging, self.flags, self.state = staging, flags, state
        self._keepalive = keepalive
        self._sp = (ctypes.c_void_p * world)(*[t.data_ptr() for t in staging])
        self._fp = (ctypes.c_void_p * world)(*[t.data_ptr() for t in flags])
        self.device = state.device
        self._calls = 0
        with torch.inference_mode(False):  # (a persistent buffer: never an inference tensor, whatever mode the constructor runs in)
            self._err_host = torch.zeros(4, dtype=torch.int32).pin_memory() if state.is_cuda else None
        self._err_event = None

    # ---- construction -----------------------------------------------------------------------------------------------
    @staticmethod
    def _alloc(max_halfs, device):
        """(staging, flags) as raw uncached allocations + their uint8 tensor views, and the private state tensor."""
        L = _lib.lib()
        device = torch.device(device)
        with torch.cuda.device(device):
            st = _DeviceBytes(L.awq_allreduce_staging_bytes(max_halfs))
            # the flag blocks of rounds 3-4 are unused since the protocol became a push of self-validating granules (round 5); the C ABI
            # keeps the argument ("pass any mapped buffer"): the staging buffer itself is passed -- no second allocation, no second IPC
            # mapping per peer
            fl = st
            state = torch.zeros(L.awq_allreduce_state_bytes(), dtype=torch.uint8, device=device)
        return st, fl, state

    @classmethod
    def local_group(cls, world, max_halfs, device="cuda"):
        """`world` ranks in ONE process on ONE device (tests, single-GPU graph-capture checks)."""
        device = torch.device(device if torch.device(device).index is not None else f"cuda:{torch.cuda.current_device()}")
        bufs = [cls._alloc(max_halfs, device) for _ in range(world)]
        torch.cuda.synchronize(device)
        st, fl = [b[0].tensor(device) for b in bufs], [b[1].tensor(device) for b in bufs]
        return [cls(r, world, st, fl, bufs[r][2], max_halfs) for r in range(world)]

    @classmethod
    def from_process_group(cls, max_halfs, group=None, device=None):
        """One process per GPU (torch.distributed initialised): allocate (uncached, zeroed), exchange the IPC handles of the
        staging and flag buffers through the process group, map the peers'.  COLLECTIVE: every rank of `group` must call it, and
        every rank gets the same outcome -- the ranks agree after each step, so a failure on ONE rank (a refused allocation, a
        hipIpcOpenMemHandle that fails) raises OneShotSetupError on ALL of them instead of leaving some inside a barrier while
        others fall back to another collective."""
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)

        def agree(ok, what, err):
            oks = [None] * world
            dist.all_gather_object(oks, (bool(ok), None if ok else f"rank {rank}: {what}: {err}"), group=group)
            bad = [m for o, m in oks if not o]
            if bad:
                raise OneShotSetupError("; ".join(bad))

        st = fl = state = None
        mine, err = None, None
        try:
            st, fl, state = cls._alloc(max_halfs, device)
            torch.cuda.synchronize(device)
            mine = (st.ipc_handle(), st.nbytes)
        except Exception as e:  # noqa: BLE001 -- reported to every rank below
            err = e
        agree(mine is not None, "allocation / IPC export", err)
        everyone = [None] * world
        dist.all_gather_object(everyone, mine, group=group)
        staging, flags, keep, err = [], [], [st], None

        def view(b):
            """the tensor view of a mapping must BE the mapping: torch infers the device of a raw pointer from the driver, and for a
            peer's IPC mapping a wrong answer would make as_tensor copy it silently -- flags and staging would then be private
            copies, the one-shot path would time out on every call; the mappings are kept alive explicitly"""
            t = b.tensor(device)
            if t.data_ptr() != b.ptr or t.device != device:
                raise RuntimeError(f"the view of a mapped buffer is a copy (pointer {t.data_ptr():#x} on {t.device}, mapping {b.ptr:#x} on {device})")
            keep.append(b)
            return t

        try:
            with torch.cuda.device(device):
                for r, (hs, ns) in enumerate(everyone):
                    staging.append(view(st) if r == rank else view(_DeviceBytes(ns, handle=hs)))
                flags = staging  # (unused by the kernel: see _alloc)
        except Exception as e:  # noqa: BLE001
            err = e
        agree(err is None, "mapping the peers' buffers", err)
        dist.barrier(group=group)  # nobody launches before everybody has mapped (and zeroed) everything
        return cls(rank, world, staging, flags, state, max_halfs, keepalive=keep)

    # ---- use --------------------------------------------------------------------------------------------------------
    def __call__(self, x, out=None):
        """All-reduce (sum) of a contiguous fp16 tensor on the current stream; in place unless `out` is given."""
        if x.dtype != torch.float16 or not x.is_contiguous() or x.device != self.device:
            raise _lib.AwqHipError("OneShotAllReduce: contiguous fp16 tensor on this rank's device expected")
        out = x if out is None else out
        n = x.numel()
        if n % 4 or n > self.max_halfs:
            raise _lib.AwqHipError(f"OneShotAllReduce: {n} elements (need a multiple of 4, at most {self.max_halfs})")
        with torch.cuda.device(self.device):
            rc = _lib.lib().awq_allreduce_oneshot(self._sp, self._fp, self.rank, self.world, x.data_ptr(), out.data_ptr(), n,
                                                  self.max_halfs, self.state.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "awq_allreduce_oneshot")
        self._calls += 1
        if self._calls % self.CHECK_EVERY == 0 and not torch.cuda.is_current_stream_capturing():
            self._poll_error()
        return out

    def _poll_error(self):
        """The sticky error word, looked at without stalling the stream: an asynchronous copy to pinned memory every
        CHECK_EVERY calls, read when the NEXT look finds it complete (same scheme as ops._Workspace)."""
        if self._err_event is not None and self._err_event.query():
            if int(self._err_host[1]) != 0:
                raise _lib.AwqHipError(f"OneShotAllReduce: rank {self.rank} gave up waiting for peer {int(self._err_host[1]) - 1} "
                                       f"(sticky error word); the affected outputs were written as NaN")
            self._err_event = None
        if self._err_event is None:
            self._err_host.copy_(self.state.view(torch.int32), non_blocking=True)
            self._err_event = torch.cuda.Event()
            self._err_event.record()

    def check(self):
        """Synchronising look at the sticky error word: call at the host sync points of a step loop (end of a token, after a
        graph replay).  Raises if a launch of this rank ever gave up on a peer."""
        epochs, err = self.status()
        if err:
            raise _lib.AwqHipError(f"OneShotAllReduce: rank {self.rank} gave up waiting for peer {err - 1} after {epochs} completed "
                                   f"all-reduces; the affected outputs were written as NaN")
        return epochs

    @staticmethod
    def group_call(ranks, xs, outs=None):
        """Every rank of a `local_group` in ONE launch on the current stream (awq_allreduce_oneshot_group): xs[r] is rank r's
        tensor; in place unless `outs` is given."""
        a = ranks[0]
        outs = xs if outs is None else outs
        n = xs[0].numel()
        for x, o in zip(xs, outs):
            if x.dtype != torch.float16 or not x.is_contiguous() or x.numel() != n or o.numel() != n or o.dtype != torch.float16:
                raise _lib.AwqHipError("OneShotAllReduce.group_call: equal-sized contiguous fp16 tensors expected")
        if n % 4 or n > a.max_halfs:
            raise _lib.AwqHipError(f"OneShotAllReduce: {n} elements (need a multiple of 4, at most {a.max_halfs})")
        P = a.world
        ins = (ctypes.c_void_p * P)(*[x.data_ptr() for x in xs])
        ous = (ctypes.c_void_p * P)(*[o.data_ptr() for o in outs])
        sts = (ctypes.c_void_p * P)(*[r.state.data_ptr() for r in ranks])
        with torch.cuda.device(a.device):
            rc = _lib.lib().awq_allreduce_oneshot_group(a._sp, a._fp, P, ins, ous, n, a.max_halfs, sts, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "awq_allreduce_oneshot_group")
        return outs

    def status(self):
        """(epochs completed, sticky error word) -- synchronises."""
        s = self.state.view(torch.int32).cpu()
        return int(s[0]), int(s[1])


def make_collective(max_halfs, device, group=None):
    """The all-reduce a tensor-parallel decode step should use, decided THE SAME WAY ON EVERY RANK: the one-shot kernel if its
    setup and a self-check (against `dist.all_reduce`) succeed on all ranks, else RCCL through torch.distributed.
    Returns (callable summing a contiguous fp16 tensor over the ranks in place, description, OneShotAllReduce | None)."""
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ar, why = None, None
    try:
        ar = OneShotAllReduce.from_process_group(max_halfs=max_halfs, group=group, device=device)  # same outcome on every rank
    except OneShotSetupError as e:
        why = f"setup failed: {str(e)[:160]}"
    if ar is not None:
        ok = 0
        try:
            probe = torch.full((max_halfs // 4 * 4,), float(rank + 1), dtype=torch.float16, device=device)
            ar(probe)
            torch.cuda.synchronize(device)
            ok = int(bool((probe == world * (world + 1) / 2).all()) and ar.status()[1] == 0)
        except Exception:  # noqa: BLE001 -- a local failure must not desynchronise the ranks: it becomes a vote
            ok = 0
        vote = torch.tensor([ok], device=device)
        dist.all_reduce(vote, op=dist.ReduceOp.MIN, group=group)
        if int(vote.item()) != 1:
            ar, why = None, "self-check against the expected sum failed on at least one rank"
    if ar is not None:
        return ar, "one-shot xGMI all-reduce (csrc/allreduce.hip; uncached fine-grained flags + staging), hipGraph-captured", ar
    return (lambda t: dist.all_reduce(t, group=group)), f"RCCL all_reduce via torch.distributed (one-shot {why})", None
