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


class OneShotAllReduce:
    def __init__(self, rank, world, staging, flags, state, max_halfs, keepalive=None):
        """staging / flags: lists of `world` uint8 tensors (entry `rank` is this rank's own, the others are peer mappings)."""
        assert 1 <= world <= 8 and 0 <= rank < world and len(staging) == world and len(flags) == world
        self.rank, self.world, self.max_halfs = rank, world, int(max_halfs)
        self.staging, self.flags, self.state = staging, flags, state
        self._keepalive = keepalive
        self._sp = (ctypes.c_void_p * world)(*[t.data_ptr() for t in staging])
        self._fp = (ctypes.c_void_p * world)(*[t.data_ptr() for t in flags])
        self.device = state.device

    # ---- construction -----------------------------------------------------------------------------------------------
    @staticmethod
    def _alloc(max_halfs, device):
        L = _lib.lib()
        st = torch.zeros(L.awq_allreduce_staging_bytes(max_halfs), dtype=torch.uint8, device=device)
        fl = torch.zeros(L.awq_allreduce_flag_bytes(), dtype=torch.uint8, device=device)
        state = torch.zeros(L.awq_allreduce_state_bytes(), dtype=torch.uint8, device=device)
        return st, fl, state

    @classmethod
    def local_group(cls, world, max_halfs, device="cuda"):
        """`world` ranks in ONE process on ONE device (tests, single-GPU graph-capture checks)."""
        bufs = [cls._alloc(max_halfs, device) for _ in range(world)]
        torch.cuda.synchronize(device)
        return [cls(r, world, [b[0] for b in bufs], [b[1] for b in bufs], bufs[r][2], max_halfs) for r in range(world)]

    @classmethod
    def from_process_group(cls, max_halfs, group=None, device=None):
        """One process per GPU (torch.distributed initialised): allocate, zero, exchange CUDA-IPC handles of the staging and
        flag buffers through the process group, map the peers'.  Collective: every rank of `group` must call it."""
        import torch.distributed as dist
        from torch.multiprocessing.reductions import reduce_tensor

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        st, fl, state = cls._alloc(max_halfs, device)
        torch.cuda.synchronize(device)
        mine = (reduce_tensor(st), reduce_tensor(fl))
        everyone = [None] * world
        dist.all_gather_object(everyone, mine, group=group)
        staging, flags, keep = [], [], []
        for r, ((fs, as_), (ff, af)) in enumerate(everyone):
            if r == rank:
                staging.append(st)
                flags.append(fl)
            else:
                ps, pf = fs(*as_), ff(*af)  # rebuild_cuda_tensor: hipIpcOpenMemHandle under the owner's device index
                staging.append(ps)
                flags.append(pf)
                keep += [ps, pf]
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
        return out

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
