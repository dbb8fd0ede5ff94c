"""Gradient-exchange transports of the data-parallel step (replaces the reduce_add of nn.DataParallel, reference
train.py:114-118).

Two transports behind one small interface (`all_reduce_async` / `wait` / `host_max`):

* `DirectRccl` - RCCL's C API through ctypes (the librccl.so torch itself links), on a communicator and a HIP stream of our own.
  An all-reduce forks that stream from the current one with an event, runs `ncclAllReduce` in place on the flat gradient slice,
  and `wait` joins it back with a second event.  Nothing of torch.distributed touches these collectives: no Work objects, no
  ProcessGroupNCCL watchdog thread polling their events - so the same calls are capturable into a hipGraph (fork / join become
  graph edges) with no process-wide state to quiesce first.  torch.distributed is used once, to hand rank 0's `ncclUniqueId` to
  the other ranks.
* `TorchGroup` - `torch.distributed.all_reduce(async_op=True)` on a process group: gloo in the CPU tests and for ranks that share
  one GPU, and the EAGER fallback over ProcessGroupNCCL should the direct communicator fail to come up.  It is NOT capturable:
  ProcessGroupNCCL's watchdog thread hipEventQuery()s the end events of eager works it has not reaped yet, an event's stream may
  not be capturing while it is queried, and nothing tells the caller when the watchdog has reaped a work - round 3 slept three
  watchdog periods before a capture, and a dedicated capture-only group (tried in round 4) still needs one eager collective to
  bring its communicator up, i.e. the same race: twenty captures in a row aborted.  Captured data-parallel steps therefore
  exist on the direct transport only (Trainer._capture refuses the other).

`make_transport()` picks one for a process group; every rank takes the same decision (the choice is agreed with a MIN
all-reduce on the bootstrap group).  With more than one rank the direct transport is rehearsed in child processes first
(`probe_direct`, `pesr_amd/comm_probe.py`): if it hangs or fails there, the children are killed and every rank uses
torch.distributed - the ctypes path has never run on more than one GPU, and a hang inside `ncclCommInitRank` cannot be undone.
"""
from __future__ import annotations

import ctypes
import os
import time
from typing import List, Optional

import torch
import torch.distributed as dist

NCCL_UNIQUE_ID_BYTES = 128           # rccl.h
NCCL_SUM, NCCL_MAX = 0, 2            # ncclRedOp_t
NCCL_FLOAT32, NCCL_FLOAT64 = 7, 8    # ncclDataType_t


class CommError(RuntimeError):
    pass


# ---- ONE wall-clock budget for everything a multi-rank run does before its first timed step (round 6) ---------------------------
# Probes in child processes (`probe_direct`, RCCL and peer memory), the transports' self-tests and Trainer.calibrate_dp_policy all
# draw on it: PESR_DP_BRINGUP_BUDGET seconds (default 300) from the first get_transport() of the process.  When it is spent the
# remaining probes / candidates are skipped, every rank runs `overlap` on the best transport that is up, and the reason is kept in
# the calibration record (`fallback_reason`).  Decisions are taken on the MAX over ranks of the time spent, so all ranks agree.
_BRINGUP = {"t0": None, "log": []}


def bringup_budget() -> float:
    return float(os.environ.get("PESR_DP_BRINGUP_BUDGET", "300"))


def bringup_start() -> None:
    if _BRINGUP["t0"] is None:
        _BRINGUP["t0"] = time.monotonic()


def bringup_spent() -> float:
    bringup_start()
    return time.monotonic() - _BRINGUP["t0"]


def bringup_left() -> float:
    return bringup_budget() - bringup_spent()


def bringup_note(what: str, seconds: float, ok=True, detail: str = "") -> None:
    _BRINGUP["log"].append({"what": what, "s": round(seconds, 2), "ok": bool(ok), **({"detail": detail[:300]} if detail else {})})


def bringup_log():
    return {"budget_s": bringup_budget(), "spent_s": round(bringup_spent(), 2), "steps": list(_BRINGUP["log"])}


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * NCCL_UNIQUE_ID_BYTES)]


_RCCL = None


def rccl_lib():
    """librccl.so - the copy torch links (one RCCL per process), else the system one."""
    global _RCCL
    if _RCCL is not None:
        return _RCCL
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so", "/opt/rocm/lib/librccl.so"]
    err = None
    for c in cands:
        try:
            L = ctypes.CDLL(c)
        except OSError as e:
            err = e
            continue
        L.ncclGetErrorString.restype = ctypes.c_char_p
        L.ncclGetErrorString.argtypes = [ctypes.c_int]
        L.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        L.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        L.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                    ctypes.c_void_p]
        L.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        L.ncclGetVersion.argtypes = [ctypes.POINTER(ctypes.c_int)]
        for f in (L.ncclGetUniqueId, L.ncclCommInitRank, L.ncclAllReduce, L.ncclCommDestroy, L.ncclGetVersion):
            f.restype = ctypes.c_int
        _RCCL = L
        return L
    raise CommError(f"librccl.so not found ({err})")


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise CommError(f"{what}: RCCL error {rc} ({rccl_lib().ncclGetErrorString(rc).decode()})")


class Transport:
    """all_reduce_async(t) -> handle: SUM over the ranks, in place, asynchronously to the current stream (which the collective
    waits for first); wait(handles): the current stream waits for those collectives; host_max(values): element-wise MAX of a
    few host floats over the ranks (policy decisions every rank must take alike)."""
    name = "?"
    world = 1
    capturable = False
    # time_collectives = True: transports that own their communication stream (DirectRccl, PeerCopy) bracket every all-reduce with
    # HIP timing events ON THAT STREAM; collective_times() -> [(bytes, ms)] (bench.py's per-bucket GB/s).  Not under capture.
    time_collectives = False

    def all_reduce_async(self, t: torch.Tensor):
        raise NotImplementedError

    def _timing_begin(self, stream, nbytes):
        if not self.time_collectives or torch.cuda.is_current_stream_capturing():
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        return (nbytes, e0)

    def _timing_end(self, br, stream):
        if br is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(stream)
            self.__dict__.setdefault("_timed", []).append((br[0], br[1], e1))

    def collective_times(self):
        """-> [(bytes, milliseconds on the communication stream)] of the all-reduces since the last call (synchronises)."""
        ev, self.__dict__["_timed"] = self.__dict__.get("_timed", []), []
        if not ev:
            return []
        torch.cuda.synchronize()
        return [(n, a.elapsed_time(b)) for n, a, b in ev]

    def wait(self, handles) -> None:
        raise NotImplementedError

    def host_max(self, values: List[float]) -> List[float]:
        raise NotImplementedError

    def begin_capture(self) -> None:
        """Called once before a hipGraph capture that will contain this transport's collectives."""

    def close(self) -> None:
        pass


class DirectRccl(Transport):
    name = "rccl-direct"
    capturable = True

    def __init__(self, device: torch.device, rank: int, world: int, bootstrap_group=None):
        L = rccl_lib()
        self.device, self.rank, self.world = device, rank, world
        uid = _UniqueId()
        if rank == 0:
            _check(L.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        if world > 1:      # the only use of torch.distributed: rank 0's id to everybody (a CPU tensor over gloo, a GPU one over nccl)
            on_gpu = dist.get_backend(bootstrap_group) == "nccl"
            t = torch.frombuffer(bytearray(bytes(uid.internal)) if rank == 0 else bytearray(NCCL_UNIQUE_ID_BYTES), dtype=torch.uint8).clone()
            if on_gpu:
                t = t.to(device)
            src = 0 if bootstrap_group is None else dist.get_global_rank(bootstrap_group, 0)
            dist.broadcast(t, src, group=bootstrap_group)
            ctypes.memmove(ctypes.byref(uid), bytes(t.cpu().numpy().tobytes()), NCCL_UNIQUE_ID_BYTES)
        self._comm = ctypes.c_void_p()
        with torch.cuda.device(device):
            _check(L.ncclCommInitRank(ctypes.byref(self._comm), world, uid, rank), "ncclCommInitRank")
            self.stream = torch.cuda.Stream(device=device)
        v = ctypes.c_int()
        L.ncclGetVersion(ctypes.byref(v))
        self.version = v.value
        # self-test: every rank contributes 1 -> world (also warms the communicator's first-use setup outside any timed region)
        one = torch.ones(4, dtype=torch.float32, device=device)
        self.wait([self.all_reduce_async(one)])
        torch.cuda.current_stream(device).synchronize()
        if [float(x) for x in one.cpu()] != [float(world)] * 4:
            raise CommError(f"ncclAllReduce self-test: expected {world}, got {one.cpu().tolist()}")

    def _launch(self, t: torch.Tensor, op: int):
        assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.float64)
        cur = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(cur)                      # fork: the collective reads what the current stream has written so far
        self.stream.wait_event(ready)
        dt = NCCL_FLOAT32 if t.dtype == torch.float32 else NCCL_FLOAT64
        br = self._timing_begin(self.stream, t.numel() * t.element_size()) if op == NCCL_SUM else None
        _check(rccl_lib().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), dt, op, self._comm, self.stream.cuda_stream), "ncclAllReduce")
        self._timing_end(br, self.stream)
        done = torch.cuda.Event()
        done.record(self.stream)
        return done

    def all_reduce_async(self, t: torch.Tensor):
        return self._launch(t, NCCL_SUM)

    def wait(self, handles) -> None:
        cur = torch.cuda.current_stream(self.device)
        for h in handles:
            cur.wait_event(h)                  # join (the host does not block)

    def host_max(self, values: List[float]) -> List[float]:
        t = torch.tensor(values, dtype=torch.float64, device=self.device)
        self.wait([self._launch(t, NCCL_MAX)])
        return [float(x) for x in t.cpu()]     # (.cpu() synchronises the current stream, which has joined the collective)

    def close(self) -> None:
        if self._comm:
            torch.cuda.synchronize(self.device)
            rccl_lib().ncclCommDestroy(self._comm)
            self._comm = ctypes.c_void_p()


class _PeerArgs(ctypes.Structure):        # include/pesr_hip.h PesrPeerArgs
    _fields_ = [("rank", ctypes.c_int), ("world", ctypes.c_int), ("epoch", ctypes.c_uint), ("pad_", ctypes.c_uint),
                ("mine", ctypes.c_void_p), ("peer", ctypes.c_void_p * 16), ("my_flags", ctypes.c_void_p),
                ("peer_flags", ctypes.c_void_p * 16), ("scratch", ctypes.c_void_p), ("numel", ctypes.c_size_t), ("ctx", ctypes.c_void_p)]


class PeerCopy(Transport):
    """Reduce-scatter + all-gather over peer memory (pesr_amd/csrc/peer_exchange.hip): every rank maps the other ranks' flat
    gradient buffers and flag words through IPC handles; an all-reduce is a sequence of stream wait / write-value operations and
    peer copies on this transport's stream plus ONE small kernel over 1/N of the bytes - nothing is resident on a compute unit
    while it waits or copies (RCCL's all-reduce is, and one held CU costs every 256-workgroup conv kernel a second round,
    profiles/r03_cu_contention.txt).  A slice's sum is formed by one rank in rank order and copied: bit-identical replicas.
    torch.distributed is used for the one-time exchange of the IPC handles and for host_max (gloo or nccl bootstrap group).
    Not capturable (stream memory operations under a hipGraph capture are not attempted).  Verified with two processes on ONE
    GPU (scripts/ipc_probe.py, tests/test_dp_gpu.py `ipc2-one-gpu`); between several GPUs it is rehearsed in child processes
    before use (`probe_direct(kind="peer")`)."""
    name = "peer-copy"
    capturable = False
    FLAG_BYTES = 4096

    def __init__(self, device: torch.device, rank: int, world: int, group=None):
        from . import _lib
        if world > 16:
            raise CommError("peer-copy transport: at most 16 ranks")
        self.L = _lib.lib()
        self.device, self.rank, self.world, self.group = device, rank, world, group
        self.epoch = 0
        # registered STORAGES (round 6): {(storage address, bytes): (the storage - kept alive, so the address cannot be
        # handed to another tensor while it is registered -, [every rank's mapping of ITS storage's first byte; own = the address])}
        self.regs = {}
        self._verified = set()           # (storage key, byte offset, numel) whose placement every rank has confirmed to be the same
        self._open_cache = {}            # (rank, IPC handle) -> base of that allocation as mapped here (one mapping per allocation)
        self._opened = []
        self.scratch, self.scratch_bytes = ctypes.c_void_p(), 0
        self.ctx = ctypes.c_void_p()
        with torch.cuda.device(device):
            self.stream = torch.cuda.Stream(device=device)
            # (a context or flag block that cannot be made HERE goes through the same agreed-error path as a failed mapping: every
            # rank raises together instead of one rank leaving the others in the collective)
            rc_ctx = self.L.pesr_peer_ctx_create(world, ctypes.byref(self.ctx))
            self.flags = ctypes.c_void_p()
            h = (ctypes.c_ubyte * 64)()
            rc = self.L.pesr_peer_alloc(self.FLAG_BYTES, ctypes.byref(self.flags), h) if rc_ctx == 0 else rc_ctx
            try:
                self.peer_flags = self._exchange(bytes(h) if rc == 0 else None, 0, self.flags.value or 0, "the context / flag block")
            except CommError:
                self.close(collective=False)
                raise
        # self-test: every rank contributes rank + 1 -> world (world + 1) / 2, twice (the epochs count on).  With a DEADLINE: an exchange
        # that does not complete (a flag write that never arrives) is abandoned - this rank's own flag words are forced, its streams
        # run out - and every rank raises together; the caller falls back to another transport.
        t = torch.full((8,), float(rank + 1), dtype=torch.float32, device=device)
        deadline = float(os.environ.get("PESR_PEER_SELFTEST_TIMEOUT", "30"))
        err = None
        try:
            for k in (1, 2):
                done = self.all_reduce_async(t)
                t0 = time.monotonic()
                while not done.query() and time.monotonic() - t0 < deadline:
                    time.sleep(0.001)
                if not done.query():
                    err = f"self-test {k} did not complete within {deadline:.0f} s"
                    with torch.cuda.device(device):
                        self.L.pesr_peer_release(self.flags, self.FLAG_BYTES)
                    torch.cuda.synchronize(device)
                    break
                self.wait([done])
                want = float(world * (world + 1) // 2) * world ** (k - 1)
                if [float(x) for x in t.cpu()] != [want] * 8:
                    err = f"self-test {k}: expected {want}, got {t.cpu().tolist()}"
                    break
        except CommError as e:              # (raised by every rank together: _lookup's agreed-error path)
            self.close(collective=False)
            raise
        if world > 1:
            errs = [None] * world
            dist.all_gather_object(errs, err, group=group)
            bad = [f"rank {r}: {e}" for r, e in enumerate(errs) if e]
            if bad and err is None:                          # (a peer gave up: its flag writes may never come - do not wait for them later)
                with torch.cuda.device(device):
                    self.L.pesr_peer_release(self.flags, self.FLAG_BYTES)
                torch.cuda.synchronize(device)
            err = "; ".join(bad) if bad else None
        if err:
            self.close(collective=False)
            raise CommError("peer-copy " + err)
        # the self-test tensor's registration is dropped: its small-pool block will hold other tensors later, and nothing may be
        # looked up through a placement that was agreed for THIS tensor
        torch.cuda.synchronize(device)
        self.regs.clear()
        self._verified.clear()

    def _exchange(self, handle, offset: int, own_ptr: int, what: str, extra_ok=None):
        """All ranks' (handle, offset) -> every rank's pointer as mapped here (own: own_ptr).  handle None = this rank could not
        make one: every rank then raises together (a rank that raised alone would leave the others waiting in the collective).
        extra_ok: a callable run after the mappings exist (scratch allocation); its exception counts as this rank's failure in the
        same agreed round."""
        mine = (self.rank, handle, int(offset))
        if self.world == 1:
            if handle is None:
                raise CommError(f"peer-copy: {what} failed")
            if extra_ok is not None:
                extra_ok()
            return [own_ptr]
        got = [None] * self.world
        dist.all_gather_object(got, mine, group=self.group)
        bad = [r for r, h, _ in got if h is None]
        out, err = [], None
        if not bad:
            for r, h, off in sorted(got):
                if r == self.rank:
                    out.append(own_ptr)
                    continue
                base = self._open_cache.get((r, h))
                if base is None:
                    bp = ctypes.c_void_p()
                    hb = (ctypes.c_ubyte * 64).from_buffer_copy(h)
                    with torch.cuda.device(self.device):
                        rc = self.L.pesr_peer_open(hb, ctypes.byref(bp))
                    if rc != 0:
                        err = f"pesr_peer_open(rank {r}) error {rc}"
                        break
                    self._opened.append(bp)
                    base = self._open_cache[(r, h)] = bp.value
                out.append(base + off)
            if err is None and extra_ok is not None:
                try:
                    extra_ok()
                except Exception as e:
                    err = f"{type(e).__name__}: {e}"
        ok = [None] * self.world
        dist.all_gather_object(ok, err is None, group=self.group)     # (second round: a mapping that failed on ONE rank stops all)
        if bad or not all(ok):
            raise CommError(f"peer-copy: {what} failed on rank(s) {bad or [r for r, v in enumerate(ok) if not v]}" + (f" ({err})" if err else ""))
        return out

    def _lookup(self, t: torch.Tensor):
        """-> every rank's mapping (as seen here) of the address that corresponds to t.data_ptr().  Registration is per STORAGE (a
        flat gradient buffer), not per allocator block: each rank exports the allocation that holds its storage together with the
        storage's offset in it, so two buffers the caching allocator happened to place in one block - or at different offsets on
        different ranks - each get their own, correct, mapping.  Every new (offset, length) inside a storage is confirmed once to be
        the same on all ranks (buckets are cut alike everywhere; a mismatch raises on every rank instead of summing the wrong bytes).
        All ranks get here in the same order: the collectives line up."""
        st = t.untyped_storage()
        key = (st.data_ptr(), st.nbytes())
        rel = t.data_ptr() - key[0]
        ent = self.regs.get(key)
        if ent is None:
            h = (ctypes.c_ubyte * 64)()
            off, size = ctypes.c_size_t(), ctypes.c_size_t()
            with torch.cuda.device(self.device):
                rc = self.L.pesr_peer_export(ctypes.c_void_p(key[0]), h, ctypes.byref(off), ctypes.byref(size))
            # scratch for the largest tensor this storage can hold, NOW (every rank is here, nothing of it is in flight): growing
            # it later would need a device synchronisation in the middle of a backward pass; a failure is agreed like a mapping's
            peers = self._exchange(bytes(h) if rc == 0 else None, off.value, key[0], "the export of a gradient buffer",
                                   extra_ok=lambda: self._ensure_scratch(key[1] // 4))
            ent = self.regs[key] = (st, peers)
        vkey = (key, rel, t.numel())
        if vkey not in self._verified:
            if self.world > 1:
                seen = [None] * self.world
                dist.all_gather_object(seen, (rel, t.numel(), key[1]), group=self.group)
                if any(x != seen[0] for x in seen):
                    raise CommError(f"peer-copy: the ranks disagree on a bucket's placement (offset, numel, buffer bytes): {seen}")
            self._verified.add(vkey)
        return [pb + rel for pb in ent[1]]

    def _ensure_scratch(self, numel: int) -> None:
        slice_elems = ((numel + self.world - 1) // self.world + 3) & ~3
        need = max(1, self.world - 1) * slice_elems * 4
        if need > self.scratch_bytes:
            with torch.cuda.device(self.device):
                torch.cuda.synchronize(self.device)
                if self.scratch:
                    self.L.pesr_peer_free(self.scratch)
                    self.scratch, self.scratch_bytes = ctypes.c_void_p(), 0
                h = (ctypes.c_ubyte * 64)()
                sc = ctypes.c_void_p()
                _check_hip(self.L.pesr_peer_alloc(need, ctypes.byref(sc), h), "pesr_peer_alloc(scratch)")
                self.scratch, self.scratch_bytes = sc, need

    def copy_engine_probe(self, t: torch.Tensor, nbytes: int = 64 << 20, hog_us: int = 20000):
        """Which engine moves this transport's peer copies?  A copy of `nbytes` out of the NEXT rank's mapping of tensor t's buffer into
        scratch, timed alone and started while a kernel holds every wave slot of this GPU for hog_us (pesr_peer_copy_probe): a blit
        kernel has to wait for the hog, a copy engine does not.  -> dict (bench.py's `peer_copy_engine`); all ranks call it together."""
        if self.world < 2:
            return None
        peers = self._lookup(t)
        nbytes = min(int(nbytes), t.numel() * 4, self.scratch_bytes)
        out = (ctypes.c_float * 3)()
        torch.cuda.synchronize(self.device)
        if self.world > 1:
            dist.barrier(group=self.group)           # nobody is writing the buffers that are read here
        with torch.cuda.device(self.device):
            rc = self.L.pesr_peer_copy_probe(ctypes.c_void_p(peers[(self.rank + 1) % self.world]), self.scratch, nbytes, int(hog_us), out)
        if self.world > 1:
            dist.barrier(group=self.group)
        if rc != 0:
            return {"error": f"pesr_peer_copy_probe: error {rc}"}
        alone, under, hog = float(out[0]), float(out[1]), float(out[2])
        return {"bytes": nbytes, "copy_alone_ms": round(alone, 3), "copy_under_cu_hog_ms": round(under, 3), "cu_hog_ms": round(hog, 3),
                "GB_per_s_alone": round(nbytes / alone / 1e6, 1) if alone > 0 else None,
                "engine": "copy engine (not held up by a kernel on every wave slot)" if under < alone + 0.5 * hog else
                          "blit kernel on the compute units (waited for the hog kernel)"}

    def all_reduce_async(self, t: torch.Tensor):
        assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.numel() % 4 == 0, "peer-copy: fp32, numel % 4 == 0"
        peers = self._lookup(t)                  # every rank's mapping of THIS tensor's first byte
        self._ensure_scratch(t.numel())          # (a no-op after the registration above)
        self.epoch += 1
        a = _PeerArgs()
        a.rank, a.world, a.epoch, a.numel = self.rank, self.world, self.epoch, t.numel()
        a.mine, a.my_flags, a.scratch, a.ctx = t.data_ptr(), self.flags.value, self.scratch.value, self.ctx.value
        for r in range(self.world):
            a.peer[r] = peers[r]
            a.peer_flags[r] = self.peer_flags[r]
        cur = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(cur)                      # fork: the exchange reads what the current stream has written so far
        self.stream.wait_event(ready)
        br = self._timing_begin(self.stream, t.numel() * 4)
        with torch.cuda.device(self.device):
            _check_hip(self.L.pesr_peer_allreduce(ctypes.byref(a), ctypes.c_void_p(self.stream.cuda_stream)), "pesr_peer_allreduce")
        self._timing_end(br, self.stream)
        done = torch.cuda.Event()
        done.record(self.stream)
        return done

    def wait(self, handles) -> None:
        cur = torch.cuda.current_stream(self.device)
        for h in handles:
            cur.wait_event(h)                  # join (the host does not block)

    def host_max(self, values: List[float]) -> List[float]:
        if self.world == 1:
            torch.cuda.synchronize(self.device)
            return list(values)
        on_gpu = dist.get_backend(self.group) == "nccl"
        t = torch.tensor(values, dtype=torch.float64, device=self.device if on_gpu else "cpu")
        torch.cuda.synchronize(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return [float(x) for x in t.cpu()]

    def begin_capture(self) -> None:
        raise CommError("the peer-copy transport's stream memory operations are not captured into a hipGraph: use the eager step")

    def close(self, collective: bool = True) -> None:
        if self.flags:
            torch.cuda.synchronize(self.device)
            if self.world > 1 and collective:
                dist.barrier(group=self.group)          # nobody unmaps a buffer a peer may still be reading
            for b in self._opened:
                self.L.pesr_peer_close(b)
            self._opened = []
            self._open_cache.clear()
            if self.scratch:
                self.L.pesr_peer_free(self.scratch)
            self.L.pesr_peer_free(self.flags)
            self.flags, self.scratch = ctypes.c_void_p(), ctypes.c_void_p()
            self.regs.clear()
        if self.ctx:
            self.L.pesr_peer_ctx_destroy(self.ctx)
            self.ctx = ctypes.c_void_p()


def _check_hip(rc: int, what: str) -> None:
    if rc != 0:
        raise CommError(f"{what}: error {rc}")


class TorchGroup(Transport):
    capturable = False

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.name = f"torch.distributed[{self.backend}]"

    def all_reduce_async(self, t: torch.Tensor):
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self, handles) -> None:
        for w in handles:
            w.wait()        # (nccl: the current stream waits for the communication stream; gloo: the host blocks)

    def host_max(self, values: List[float]) -> List[float]:
        dev = torch.device("cuda", torch.cuda.current_device()) if self.backend == "nccl" else torch.device("cpu")
        t = torch.tensor(values, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return [float(x) for x in t.cpu()]

    def begin_capture(self) -> None:
        raise CommError(f"a data-parallel step over {self.name} cannot be captured into a hipGraph (ProcessGroupNCCL's watchdog queries "
                        "events of the streams that join the capture): use the direct RCCL transport (PESR_DP_TRANSPORT=rccl or auto)")


def probe_direct(device: Optional[torch.device], group=None, timeout: Optional[float] = None, kind: str = "rccl"):
    """Can the direct RCCL transport (kind "rccl") / the peer-memory transport (kind "peer") come up among THESE ranks?  Answered in child processes (`pesr_amd.comm_probe`, one per rank,
    rendezvous among themselves over gloo on a port rank 0 picks): communicator, a 32 MB all-reduce with a known answer, a MAX
    all-reduce, teardown.  A child that has not exited after `timeout` seconds (env PESR_DP_PROBE_TIMEOUT, default 180; never more
    than what is left of the bring-up budget minus a reserve for the calibration) is killed.
    Returns (ok on THIS rank, reason); the caller agrees the answer over the ranks.  Why a child: `ncclCommInitRank` called through
    ctypes cannot be abandoned once it hangs (a partial failure - some ranks in, some out - leaves the others waiting in C),
    and no multi-GPU node was available to run that path before the first one the benchmark sees.
    The rendezvous port is picked by bind-and-close on rank 0 and re-bound by the children a moment later; should another process
    take it in between, the children fail to meet - that ONE kind of failure is retried once on a fresh port (agreed over the ranks:
    every rank takes part in the second round or none does)."""
    import socket
    import subprocess
    import sys
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    timeout = float(timeout if timeout is not None else os.environ.get("PESR_DP_PROBE_TIMEOUT", "180"))
    timeout = min(timeout, max(15.0, bringup_left() - 45.0))       # (never more than what the bring-up budget leaves, less a reserve for the calibration)
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    index = device.index if device is not None and device.type == "cuda" and device.index is not None else 0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # The children rendezvous among themselves: nothing of the launcher's environment may reach them - with torchrun's
    # TORCHELASTIC_USE_AGENT_STORE set, init_process_group would look for the launcher's store on the probe's port and wait for ever
    drop = {"RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE",
            "ROLE_NAME", "MASTER_PORT", "PESR_FORCE_DP"}
    env = {k: v for k, v in os.environ.items() if k not in drop and not k.startswith("TORCHELASTIC")}
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    t_all = time.monotonic()

    def attempt(tmo):
        port = [0]
        if rank == 0:
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("", 0))
                port[0] = sk.getsockname()[1]
        if world > 1:
            src = 0 if group is None else dist.get_global_rank(group, 0)
            dist.broadcast_object_list(port, src=src, group=group)
        cmd = [sys.executable, "-m", "pesr_amd.comm_probe", str(rank), str(world), str(index), addr, str(port[0]), str(max(10.0, tmo - 10.0)), kind]
        try:
            p = subprocess.Popen(cmd, cwd=root, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        except OSError as e:
            return False, f"probe child did not start: {e}", False
        try:
            _, err = p.communicate(timeout=tmo)
        except subprocess.TimeoutExpired:
            p.kill()
            _, err = p.communicate()
            marks = [l for l in err.decode(errors="replace").splitlines() if l.startswith("comm_probe:")]
            return False, f"probe child still running after {tmo:.0f} s (killed)" + (f"; last step: {marks[-1][12:]}" if marks else ""), False
        if p.returncode != 0:
            text = err.decode(errors="replace")
            marks = [l for l in text.splitlines() if l.startswith("comm_probe:")]
            tail = text.strip().splitlines()[-1:] or [""]
            # a failure before the communicator step = the children never met (port taken, store timeout): worth one more port
            at_rendezvous = not any("communicator" in m for m in marks)
            return False, f"probe child exited {p.returncode}: {tail[0][:300]}", at_rendezvous
        return True, "", False

    ok, why, retry = attempt(timeout)
    if world > 1:
        # retry only if EVERY failing rank failed at the rendezvous and at least one did; all ranks take the same decision
        flags = [None] * world
        dist.all_gather_object(flags, (ok, retry), group=group)
        again = any(not o for o, _ in flags) and all(o or r for o, r in flags) and bringup_left() > 90.0
    else:
        again = (not ok) and retry and bringup_left() > 90.0
    if again:
        ok2, why2, _ = attempt(max(15.0, min(timeout, bringup_left() - 45.0)))
        why = "" if ok2 else f"{why2} (second attempt; first: {why})"
        ok = ok2
    bringup_note(f"probe[{kind}]", time.monotonic() - t_all, ok, why)
    return ok, why


def make_transport(device: Optional[torch.device], group=None, prefer: Optional[str] = None) -> Transport:
    """The transport of this process group's gradient exchange.  prefer (or env PESR_DP_TRANSPORT): "rccl" = the direct
    communicator or an error, "torch" = torch.distributed, "auto" (default) = direct RCCL when the ranks own one GPU each and the
    bootstrap backend is nccl, else torch.distributed.  Under "auto" with more than one rank the direct transport is first
    brought up in CHILD processes (`probe_direct`; PESR_DP_PROBE=0 skips it) and used only if every rank's child succeeded: a
    hang or a partial failure of the ctypes path costs a killed child and an eager fallback to torch.distributed on all ranks,
    not a hung job.  Should the communicator then still fail in this process on ANY rank with an exception, all ranks fall
    back together (agreed by a MIN all-reduce); the reason is kept in `.fallback_reason`."""
    prefer = prefer or os.environ.get("PESR_DP_TRANSPORT", "auto")
    if prefer not in ("auto", "rccl", "torch", "peer"):
        raise ValueError(f"PESR_DP_TRANSPORT must be auto, rccl, torch or peer, got {prefer!r}")
    backend = dist.get_backend(group)
    if prefer == "peer" and (device is None or device.type != "cuda"):
        raise CommError("PESR_DP_TRANSPORT=peer needs one GPU per rank (IPC-mapped device memory); there is no host path")
    if prefer == "peer":       # explicit only: the CU-free exchange over peer memory (never chosen by "auto"; Trainer.calibrate_dp_policy
        return PeerCopy(device, dist.get_rank(group), dist.get_world_size(group), group)      # can time it as a candidate)
    want_direct = prefer == "rccl" or (prefer == "auto" and backend == "nccl" and device is not None and device.type == "cuda")
    if not want_direct:
        return TorchGroup(group)
    agree_dev = device if backend == "nccl" else "cpu"
    if prefer == "auto" and dist.get_world_size(group) > 1 and os.environ.get("PESR_DP_PROBE", "1") != "0":
        ok_here, why = probe_direct(device, group)
        ok = torch.tensor([1.0 if ok_here else 0.0], device=agree_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if float(ok.item()) != 1.0:
            fb = TorchGroup(group)
            fb.fallback_reason = "probe: " + (why or "another rank's probe failed")
            return fb
    tr, reason = None, ""
    t_init = time.monotonic()
    try:
        tr = DirectRccl(device, dist.get_rank(group), dist.get_world_size(group), group)
    except (CommError, OSError, RuntimeError) as e:      # (decided together below)
        reason = f"{type(e).__name__}: {e}"
    bringup_note("rccl-direct communicator + self-test", time.monotonic() - t_init, tr is not None, reason)
    ok = torch.tensor([1.0 if tr is not None else 0.0], device=agree_dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if float(ok.item()) == 1.0:
        return tr
    if prefer == "rccl":
        raise CommError("the direct RCCL communicator did not come up on every rank" + (f" (here: {reason})" if reason else ""))
    if tr is not None:
        tr.close()
    fb = TorchGroup(group)
    fb.fallback_reason = reason or "another rank failed to create its communicator"
    return fb


_TRANSPORTS = {}


def get_transport(device: Optional[torch.device], group=None) -> Transport:
    """One transport per (process group, device), shared by the optimizers of a process (ONE communicator for G and D)."""
    bringup_start()
    key = (None if group is None else id(group), None if device is None or device.type != "cuda" else device.index)
    tr = _TRANSPORTS.get(key)
    if tr is None:
        tr = _TRANSPORTS[key] = make_transport(device, group)
    return tr


def adopt_transport(tr: Transport) -> None:
    """A transport made outside get_transport() (Trainer.calibrate_dp_policy's peer-memory candidate, once it has won) is closed by
    close_transports() too - every rank adopts it at the same point, so the shutdown's collectives line up."""
    _TRANSPORTS[("adopted", len(_TRANSPORTS))] = tr


def close_transports() -> None:
    """Destroy the direct communicators (call before dist.destroy_process_group())."""
    for tr in list(_TRANSPORTS.values()):
        tr.close()
    _TRANSPORTS.clear()
