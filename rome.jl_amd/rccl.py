"""Direct RCCL binding for the separator exchange (ctypes on the librccl.so that torch already loaded).

Why: `torch.distributed.all_gather_into_tensor` costs ≈ 30 µs of host time per call (measured, scripts/host_step_cost.py) -- with a
39 µs sweep kernel the per-step host work of the N > 1 path (stream switch + launch + collective ≈ 47 µs) would bound the step, not the
GPU.  `ncclAllGather` through ctypes is a ≈ 3 µs call, is enqueued on the SAME stream as the sweep that produced the send buffer (no
cross-stream event join), and each pipeline slot gets its own communicator so that slots never serialise on one another.

The unique ids are created on rank 0 and distributed with the existing torch.distributed process group (which also remains the
fallback: `create_comms` returns None on any failure, agreed across ranks, and the caller keeps using torch.distributed)."""
import ctypes as C
import os

NCCL_FLOAT64 = 8
NCCL_UNIQUE_ID_BYTES = 128


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


_lib = None


def _load(torch):
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        lib = C.CDLL(path if os.path.exists(path) else "librccl.so")
        lib.ncclGetUniqueId.restype = C.c_int
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclCommInitRank.restype = C.c_int
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        lib.ncclCommDestroy.restype = C.c_int
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        lib.ncclCommAbort.restype = C.c_int
        lib.ncclCommAbort.argtypes = [C.c_void_p]
        lib.ncclAllGather.restype = C.c_int
        lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetErrorString.argtypes = [C.c_int]
        lib.ncclGroupStart.restype = C.c_int
        lib.ncclGroupStart.argtypes = []
        lib.ncclGroupEnd.restype = C.c_int
        lib.ncclGroupEnd.argtypes = []
        _lib = lib
    return _lib


class RcclComm:
    """One communicator; `all_gather_f64(send_ptr, recv_ptr, count, stream_ptr)` enqueues ncclAllGather on the given HIP stream."""

    def __init__(self, lib, handle):
        self._lib = lib
        self.comm = handle
        self._ag = lib.ncclAllGather

    def all_gather_f64(self, send_ptr, recv_ptr, count, stream_ptr):
        rc = self._ag(send_ptr, recv_ptr, count, NCCL_FLOAT64, self.comm, stream_ptr)
        if rc != 0:
            raise RuntimeError("ncclAllGather: %s" % self._lib.ncclGetErrorString(rc).decode())

    def close(self):
        if self.comm:
            self._lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()


class HostStagedComm:
    """RcclComm's interface over the default process group of ANY backend, through host memory: synchronous device -> host -> collective
    -> host -> device.  NOT a transport for production and never chosen implicitly -- it exists for the one situation RCCL refuses:
    several ranks sharing ONE device (a one-GPU test box).  With it the multi-rank drivers and `bench.py --gpus N` run as N real
    processes with the real kernels (tests/test_gpu_zz_ranks_one_device.py, `ROME_BENCH_SHARED_DEVICE=1`); only the wire is replaced."""

    def __init__(self, torch, dist, world, ctx):
        from . import _lib
        self.torch, self.dist, self.world, self.ctx, self._l, self._lib = torch, dist, world, ctx, _lib, _lib.load()
        self.calls = 0

    def all_gather_f64(self, send_ptr, recv_ptr, count, stream_ptr):
        import numpy as np
        torch = self.torch
        torch.cuda.synchronize()                      # the producing launches, whatever stream they are on
        send = np.empty(count, dtype=np.float64)
        self._l.check(self._lib.rome_dev_download(self.ctx.handle, send.ctypes.data, C.c_void_p(send_ptr), send.nbytes), self.ctx.handle)
        out = torch.empty(self.world * count, dtype=torch.float64)
        self.dist.all_gather_into_tensor(out, torch.from_numpy(send))
        host = out.numpy()
        self._l.check(self._lib.rome_dev_upload(self.ctx.handle, C.c_void_p(recv_ptr), host.ctypes.data, host.nbytes), self.ctx.handle)
        torch.cuda.synchronize()
        self.calls += 1

    def close(self):
        pass


def _init_group(torch, device, lib, world, rank, uids, out):
    """ALL n communicators inside ONE ncclGroupStart / ncclGroupEnd: every rank enters the same single collective initialisation
    (no rank can sit inside ncclCommInitRank of communicator k while another one has already given up on communicator k-1)."""
    if getattr(device, "type", "cpu") == "cuda":
        torch.cuda.set_device(device)   # (the current HIP device is per thread: the watchdog thread starts on device 0)
    handles = [C.c_void_p() for _ in uids]
    rc = lib.ncclGroupStart()
    if rc == 0:
        for h, uid in zip(handles, uids):
            r = lib.ncclCommInitRank(C.byref(h), int(world), uid, int(rank))
            rc = rc or r
        r = lib.ncclGroupEnd()
        rc = rc or r
    out["rc"], out["handles"] = rc, handles


_unusable = False   # set when an initialisation timed out in THIS process: a thread may still sit inside ncclGroupEnd


def create_comms(torch, dist, world, rank, device, n):
    """n independent communicators over the ranks of the default process group, or None -- ON EVERY RANK -- if anything failed
    anywhere.  Failure is agreed ONCE before the collective initialisation (unique ids) and ONCE after it; the initialisation
    itself is one ncclGroup over all n communicators, run under a watchdog (ROME_RCCL_INIT_TIMEOUT_S, default 120 s): a rank whose
    peers never arrive reports failure through torch.distributed instead of blocking its main thread forever.
    After a timeout the process is marked RCCL-unusable: the watchdog's thread may still be inside ncclGroupEnd on this device, so
    every later create_comms in this process reports failure up front (agreed across ranks like any other failure) and the caller
    stays on torch.distributed -- the documented fallback.  Communicators of a failed initialisation are ABORTED (ncclCommAbort:
    no handshake with a peer that may still be initialising), never destroyed."""
    import threading
    global _unusable
    comms, ok, lib = [], 1, None
    if _unusable:
        ok = 0
    try:
        lib = _load(torch)
        ids = torch.zeros((n, NCCL_UNIQUE_ID_BYTES), dtype=torch.uint8, device=device)
        if rank == 0:
            for k in range(n):
                uid = _UniqueId()
                rc = lib.ncclGetUniqueId(C.byref(uid))
                if rc != 0:
                    raise RuntimeError("ncclGetUniqueId: %s" % lib.ncclGetErrorString(rc).decode())
                ids[k] = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).to(device)
    except Exception:   # noqa: BLE001
        ok = 0
        ids = torch.zeros((n, NCCL_UNIQUE_ID_BYTES), dtype=torch.uint8, device=device)
    if world > 1:
        # agree BEFORE the (collective, blocking) initialisation: either every rank enters it or none does
        pre = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(pre, op=dist.ReduceOp.MIN)
        ok = int(pre.item())
        if ok:
            dist.broadcast(ids, src=0)
    handles = []
    if ok:
        try:
            host = ids.cpu().numpy()
            uids = [_UniqueId.from_buffer_copy(host[k].tobytes()) for k in range(n)]
            res = {}
            th = threading.Thread(target=_init_group, args=(torch, device, lib, world, rank, uids, res), daemon=True)
            th.start()
            th.join(float(os.environ.get("ROME_RCCL_INIT_TIMEOUT_S", "120")))
            if th.is_alive():
                ok = 0     # (a thread still inside ncclGroupEnd is left behind: its communicators are never used)
                _unusable = True
            elif res.get("rc", 1) != 0:
                ok = 0
                handles = [h for h in res.get("handles", []) if h]   # partially created: aborted below
            else:
                handles = res["handles"]
        except Exception:   # noqa: BLE001
            ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device=device)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        for h in handles:
            try:
                lib.ncclCommAbort(h)
            except Exception:   # noqa: BLE001
                pass
        return None
    return [RcclComm(lib, h) for h in handles]
