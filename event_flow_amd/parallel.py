"""Data parallelism over the GPUs of one MI355X node: one process per GPU,
batch slots (independent event sequences, reference dataloader/h5.py:51-68,184)
sharded across ranks, ONE all-reduce per optimizer step on the flat gradient
buffer over RCCL/xGMI (torch.distributed backend "nccl" is RCCL on ROCm).

The reference has no distributed code.  Its loss SUMS over
the batch (loss/flow.py:226,259,289), so the all-reduce is a SUM, not a mean:
sum over ranks of the per-shard gradient == the single-process gradient of the
global batch.  Clipping and Adam then run identically on every rank on the
reduced buffer (clip AFTER the reduce).  The scalar loss and the `new_seq`
reset flag (train_flow.py:100-105 resets all slots when any slot restarts)
ride in two extra floats at the tail of the same buffer, so a step is exactly
one collective: 299 KB for LIF-FireNet -- latency bound, never per-tensor.
"""

import os

import torch
import torch.distributed as dist

from . import _lib


class DataParallel:
    TAIL = 2  # [loss, new_seq flag]

    def __init__(self, backend=None, device=None, init=True, force_collectives=None):
        """force_collectives (or EVF_DP_FORCE=1): initialise the process group and ISSUE every collective also at world
        size 1 -- a one-rank RCCL communicator on the one GPU of a test box runs the same code path (init with device_id,
        all-reduce between the two step graphs, barrier(device_ids)) the 8-GPU run takes."""
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        if backend is None:
            backend = os.environ.get("EVF_DP_BACKEND")  # test hook: gloo with GPU tensors on a one-GPU box
        if backend is None:
            backend = "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo"
        self.backend = backend
        if force_collectives is None:
            force_collectives = os.environ.get("EVF_DP_FORCE", "0") == "1"
        # `active`: collectives are issued (several ranks, or forced at one rank)
        self.active = self.world > 1 or bool(force_collectives)
        if init and self.active and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = torch.device(device)
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, **kw)
        # RCCL collectives run on a stream of their own, never on the caller's: torch issues a blocking collective on the CURRENT
        # stream and hands its end event to the process group's watchdog thread, which polls it -- and HIP refuses a query of an
        # event whose stream has meanwhile entered a graph capture (hipErrorCapturedEvent: the watchdog throws, the process
        # aborts).  The step graphs are captured on the stream the warm-up's all-reduces ran on, so that race is real (found by
        # tests/test_gpu_training.py::test_rccl_code_path_on_one_rank_*).  Ordering against the caller's stream by events.
        self.stream = None
        if self.active and backend == "nccl" and device is not None:
            self.stream = torch.cuda.Stream(device=torch.device(device))
        # The step's ONE collective on a communicator of the library's own (include/evflow.h: evf_comm_*, evf_allreduce_sum):
        # a plain ncclAllReduce on the caller's stream -- no watchdog thread, no stream of its own -- and therefore CAPTURABLE:
        # the N-rank step replays as ONE hipGraph (bench.StepGraph).  Bootstrapped from torch's store (rank 0's 128-byte id).
        # OPT-IN (EVF_DP_NATIVE=1) since round 6: the captured multi-rank all-reduce has only ever executed at ONE rank (no
        # multi-GPU box was available to the builder), and a second communicator that misbehaves at start-up can hang a run
        # instead of failing it.  The default keeps every collective on torch.distributed: the step is then two graphs around
        # one eager all-reduce (~one launch gap per 3 ms step), the path torch's own RCCL binding has carried everywhere.
        self.native = None
        self.native_fallback = None
        self.native_version = None
        self.native_ranks = None  # ncclCommCount of the library's own communicator
        self.native_requested = os.environ.get("EVF_DP_NATIVE", "0") == "1"
        if self.active and backend == "nccl" and device is not None and self.native_requested:
            self._init_native()

    def _init_native(self):
        """Own RCCL communicator for the captured all-reduce.  Every step that can fail locally (library load, unique id,
        ncclCommInitRank, a first eager SUM of known values) is followed by a VOTE over torch's process group: either every
        rank ends up with a working communicator or every rank drops it and keeps the collectives on torch.distributed (the
        two-graph step) -- never a mix, which would hang the first step.  EVF_DP_NATIVE_STRICT=1 raises instead."""
        import ctypes

        from . import _lib

        strict = os.environ.get("EVF_DP_NATIVE_STRICT", "0") == "1"
        inject = os.environ.get("EVF_DP_NATIVE_INJECT", "")  # test hook: "load" | "init" | "check" | "capture" fails that stage
        self.native_fallback = None
        store = dist.distributed_c10d._get_default_store()
        key = "evf_rccl_unique_id/%d" % DataParallel._native_seq
        DataParallel._native_seq += 1
        err, L = None, None
        ident = (ctypes.c_char * 128)()
        # Everything a rank does BEFORE the first vote is local and wrapped: whatever raises (the library itself failing to
        # load included), rank 0 still publishes a key -- an empty one as the sentinel -- so that nobody blocks in store.get
        # until the store's timeout, and every rank reaches the vote.
        try:
            L = _lib.load()
            path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")  # the RCCL this process has loaded already
            rc = L.evf_comm_load(path.encode() if os.path.exists(path) else None)
            if rc != 0:
                err = f"evf_comm_load failed with status {rc}: {L.evf_comm_last_error().decode()}"
            if self.rank == 0 and err is None:
                rc = L.evf_comm_unique_id(ident)
                if rc != 0:
                    err = f"evf_comm_unique_id failed with status {rc}: {L.evf_comm_last_error().decode()}"
        except Exception as e:  # noqa: BLE001 -- the vote decides
            err = f"loading the library / RCCL raised: {e}"
        if self.rank == 0:
            store.set(key, bytes(ident.raw) if err is None else b"")  # (always published: nobody waits for ever)
        else:
            try:
                raw = bytes(store.get(key))  # (blocks until rank 0 has published it)
            except Exception as e:  # noqa: BLE001
                raw, err = b"", err or f"reading the unique id from the store raised: {e}"
            if len(raw) < 128:
                err = err or "rank 0 could not create the RCCL unique id"
            else:
                ident.raw = raw[:128]
        if inject == "load":
            err = err or "injected failure (load)"
        if not self._vote(err is None):
            return self._native_off(err or "another rank could not load RCCL / read the unique id", strict)
        # (every rank is past the vote with a bound RCCL and the id: the argument checks of evf_comm_init cannot fail on one
        # rank only, so every rank enters ncclCommInitRank -- which blocks until all of them have)
        comm = ctypes.c_void_p()
        with torch.cuda.device(torch.device(self.device)):
            rc = L.evf_comm_init(ident, self.rank, self.world, ctypes.byref(comm))
        if rc != 0:
            err = f"evf_comm_init failed with status {rc}: {L.evf_comm_last_error().decode()}"
            comm = None
        else:
            cnt = ctypes.c_int(-1)
            if L.evf_comm_count(comm, ctypes.byref(cnt)) == 0:
                self.native_ranks = cnt.value
                if cnt.value != self.world:
                    err = f"ncclCommCount says {cnt.value} ranks, the process group has {self.world}"
        if inject == "init":
            err = err or "injected failure (init)"
        if not self._vote(err is None):
            if comm is not None:
                L.evf_comm_destroy(comm)
            return self._native_off(err or "evf_comm_init failed on another rank", strict)
        # a first eager SUM of known values (rank r contributes r + 1): world * (world + 1) / 2 everywhere
        try:
            t = torch.full((1024,), float(self.rank + 1), dtype=torch.float32, device=torch.device(self.device))
            _lib.call("evf_allreduce_sum", comm, _lib.ptr(t), t.numel())
            torch.cuda.synchronize(torch.device(self.device))
            want = self.world * (self.world + 1) / 2.0
            if not bool((t == want).all().item()):
                err = f"evf_allreduce_sum self-check: got {float(t[0])}, expected {want}"
        except Exception as e:  # noqa: BLE001 -- whatever it was, the vote decides
            err = f"evf_allreduce_sum self-check raised: {e}"
        if inject == "check":
            err = err or "injected failure (check)"
        if not self._vote(err is None):
            L.evf_comm_destroy(comm)
            return self._native_off(err or "evf_allreduce_sum self-check failed on another rank", strict)
        # ... and the same SUM as a NODE of a hipGraph, replayed twice: what the one-graph step relies on.  (At one rank RCCL's
        # in-place all-reduce is trivial; only this check on the ranks of the run itself says that a multi-rank RCCL kernel
        # survives capture + replay on this stack.)  A failure keeps the step on two graphs around torch's eager all-reduce.
        if os.environ.get("EVF_DP_NATIVE_PREFLIGHT", "1") != "0":
            try:
                dev = torch.device(self.device)
                t = torch.zeros((1024,), dtype=torch.float32, device=dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    _lib.call("evf_allreduce_sum", comm, _lib.ptr(t), t.numel())
                for k in range(2):
                    t.fill_(float(self.rank + 1 + k))
                    g.replay()
                    torch.cuda.synchronize(dev)
                    want = self.world * (self.world + 1) / 2.0 + k * self.world
                    if not bool((t == want).all().item()):
                        err = f"captured evf_allreduce_sum: replay {k} gave {float(t[0])}, expected {want}"
                        break
                del g
            except Exception as e:  # noqa: BLE001
                err = f"captured evf_allreduce_sum raised: {e}"
            if inject == "capture":
                err = err or "injected failure (capture)"
            if not self._vote(err is None):
                # (the communicator is left alone: a capture that failed may have left it in a state destroy would wait on)
                return self._native_off(err or "captured evf_allreduce_sum failed on another rank", strict)
        self.native = comm
        ver = ctypes.c_int()
        self.native_version = ver.value if L.evf_comm_version(ctypes.byref(ver)) == 0 else None

    def _vote(self, ok):
        """True when `ok` holds on every rank (one MIN all-reduce on torch's process group)."""
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=torch.device(self.device))
        self._on_comm_stream(lambda: dist.all_reduce(t, op=dist.ReduceOp.MIN))
        return bool(t.item() > 0.5)

    def _native_off(self, why, strict):
        from . import _lib

        if strict:
            raise _lib.EvflowError(why + " (EVF_DP_NATIVE=0 keeps the collectives on torch.distributed)")
        self.native = None
        self.native_fallback = why
        if self.rank == 0:
            import sys

            print(f"[event_flow_amd.parallel] own RCCL communicator not used: {why}; the step's all-reduce stays on "
                  "torch.distributed (two-graph step)", file=sys.stderr)

    _native_seq = 0

    @property
    def capturable(self):
        """The step's collective may sit inside a hipGraph capture (own RCCL communicator, caller's stream)."""
        return self.native is not None

    # -- sharding ------------------------------------------------------------
    def shard(self, global_batch):
        """Contiguous slot range [start, stop) of this rank."""
        if global_batch % self.world:
            raise ValueError(f"global batch {global_batch} not divisible by {self.world} ranks")
        per = global_batch // self.world
        return self.rank * per, (self.rank + 1) * per

    # -- the one collective of a step ---------------------------------------
    def stage(self, comm, loss=None, new_seq=False):
        """Write this rank's loss / new-sequence flag into the tail of the flat buffer
        (device-side only; safe inside a hipGraph capture)."""
        n = comm.numel() - self.TAIL
        if loss is not None:
            # (a kernel, not copy_: a same-dtype device copy is a memcpy NODE in a captured step -- see _lib.zero_)
            torch.mul(loss.detach().reshape(1).to(comm.dtype), 1.0, out=comm[n : n + 1])
        else:
            _lib.zero_(comm[n : n + 1])
        comm[n + 1 : n + 2].fill_(1.0 if new_seq else 0.0)

    def _on_comm_stream(self, fn):
        """Run a collective on the communication stream, ordered after / before the caller's current stream."""
        if self.stream is None:
            return fn()
        cur = torch.cuda.current_stream(self.stream.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            out = fn()
        cur.wait_stream(self.stream)
        return out

    def reduce(self, comm):
        """THE collective of a step: in-place SUM all-reduce of gradient + tail."""
        if self.active and self.native is not None and comm.is_cuda:
            from . import _lib

            if comm.dtype != torch.float32 or not comm.is_contiguous():  # (the buffer goes to RCCL as n ncclFloat values)
                raise _lib.EvflowError(f"DataParallel.reduce needs a contiguous float32 buffer, got {comm.dtype}, strides {comm.stride()}")
            timed = self.__dict__.get("_timed") is not None and not torch.cuda.is_current_stream_capturing()
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            _lib.call("evf_allreduce_sum", self.native, _lib.ptr(comm), comm.numel())  # on the current stream
            if timed:
                e1.record()
                self._timed.append((e0, e1))
            return
        if self.active:
            if self.__dict__.get("_timed") is not None and comm.is_cuda:
                # (bench.py: the collective's own device time, HIP events on the stream it runs on)
                def timed():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    dist.all_reduce(comm, op=dist.ReduceOp.SUM)
                    e1.record()
                    self._timed.append((e0, e1))

                self._on_comm_stream(timed)
            else:
                self._on_comm_stream(lambda: dist.all_reduce(comm, op=dist.ReduceOp.SUM))

    def time_reduces(self, on=True):
        """Bracket every reduce() with HIP events on the communication stream (never inside a graph capture: the all-reduce of a
        step is an eager launch between the step's two graphs)."""
        self._timed = [] if on else None

    def reduce_times_ms(self):
        """Device time of the bracketed reduces so far (synchronises)."""
        ev = self.__dict__.get("_timed") or []
        if ev:
            torch.cuda.synchronize()
        return [e0.elapsed_time(e1) for e0, e1 in ev]

    def staged(self, comm):
        n = comm.numel() - self.TAIL
        return comm[n], comm[n + 1]

    def all_reduce_grads(self, comm, loss=None, new_seq=False):
        """comm: flat fp32 buffer [n + TAIL]; comm[:n] holds this rank's gradient.
        Writes loss / flag into the tail, SUM all-reduces the whole buffer in
        place, returns (global loss, any_new_seq)."""
        self.stage(comm, loss, new_seq)
        self.reduce(comm)
        return self.staged(comm)

    def broadcast(self, t, src=0):
        """In-place broadcast (parameter replicas start from rank `src`'s values)."""
        if self.active:
            self._on_comm_stream(lambda: dist.broadcast(t, src=src))

    def any_flags(self, flags):
        """Element-wise OR over the ranks of a few host booleans (one tiny MAX all-reduce): the loader events every
        rank has to act on together -- a slot started a new sequence (all slots are reset, train_flow.py:100-105),
        a rank finished its pass over the files (every rank ends the epoch)."""
        if not self.active:
            return [bool(f) for f in flags]
        t = torch.tensor([1.0 if f else 0.0 for f in flags], dtype=torch.float32,
                         device=self.device if self.backend == "nccl" else "cpu")
        self._on_comm_stream(lambda: dist.all_reduce(t, op=dist.ReduceOp.MAX))
        return [bool(v) for v in t.tolist()]

    def barrier(self):
        if self.active:
            if self.backend == "nccl":
                self._on_comm_stream(lambda: dist.barrier(device_ids=[torch.device(self.device).index]))
            else:
                dist.barrier()

    def max_over_ranks(self, value):
        if not self.active:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.backend == "nccl" else "cpu")
        self._on_comm_stream(lambda: dist.all_reduce(t, op=dist.ReduceOp.MAX))
        return float(t.item())

    def close(self):
        if self.native is not None:
            from . import _lib

            torch.cuda.synchronize()
            _lib.load().evf_comm_destroy(self.native)
            self.native = None
        if self.active and dist.is_initialized():
            dist.destroy_process_group()
