"""The gradient exchange through the C boundary (include/joeys2t_hip.h: js2t_comm_*): an RCCL communicator per process,
bootstrapped from an ncclUniqueId that rank 0 draws and this module broadcasts over whatever `torch.distributed` group is
up (gloo is enough - the id is 128 host bytes).  What torch's DistributedDataParallel does for the reference
(joeynmt/prediction.py:508-515, helpers_for_ddp.py:17-38): all-reduce(average) of gradient buckets, here issued by
helpers_for_ddp.FlatGradReducer when it is given a Communicator (TrainStep(comm="cabi") / JS2T_COMM=cabi).

The default exchange stays `torch.distributed.all_reduce` on the "nccl" backend (the same RCCL underneath): this module is the
drop-in for a host that has no torch.distributed - everything it needs from the host is the broadcast of 128 bytes.
"""
import ctypes as C
from typing import Callable, Optional

import torch

from joeys2t_amd._lib import Js2tError, check, lib
from joeys2t_amd.ops import _p, dt_code


def _bind(L):
    if getattr(L, "_comm_bound", False):
        return L
    L.js2t_comm_unique_id_bytes.restype = C.c_int64
    L.js2t_comm_unique_id.argtypes = [C.c_void_p, C.c_int64]
    L.js2t_comm_init.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32]
    L.js2t_comm_allreduce_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
    L.js2t_comm_wait.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.js2t_comm_stream.restype = C.c_void_p
    L.js2t_comm_stream.argtypes = [C.c_void_p]
    L.js2t_comm_destroy.argtypes = [C.c_void_p]
    L._comm_bound = True
    return L


def unique_id() -> bytes:
    """A fresh ncclUniqueId (call on ONE rank, hand the bytes to all)."""
    L = _bind(lib())
    n = int(L.js2t_comm_unique_id_bytes())
    buf = C.create_string_buffer(n)
    check(L.js2t_comm_unique_id(buf, n), "js2t_comm_unique_id")
    return buf.raw


class Communicator:
    """One rank's end of an RCCL communicator behind libjoeys2t_hip.so.

    `exchange_id(id_or_None) -> id`: called once with rank 0's fresh id on rank 0 and None elsewhere; returns the id every rank
    ends up with (default: a broadcast over the default torch.distributed group)."""

    def __init__(self, rank: int, world: int, device: torch.device, exchange_id: Optional[Callable[[Optional[bytes]], bytes]] = None):
        device = torch.device(device)
        if device.type != "cuda":
            raise Js2tError("Communicator: RCCL needs a GPU (the CPU rehearsal uses torch.distributed's gloo)")
        self._L = _bind(lib())
        self.rank, self.world, self.device = int(rank), int(world), device
        # rank 0 draws the id; if that fails there (librccl missing) the others must not be left waiting in the broadcast:
        # the failure travels instead of the id and every rank raises
        mine, failed = None, None
        if self.rank == 0:
            try:
                mine = unique_id()
            except Exception as exc:  # noqa: BLE001
                mine, failed = b"!" + repr(exc).encode()[:200], exc
        uid = (exchange_id or _broadcast_id)(mine)
        if failed is not None:
            raise failed
        if isinstance(uid, (bytes, bytearray)) and bytes(uid[:1]) == b"!" and len(uid) != int(self._L.js2t_comm_unique_id_bytes()):
            raise Js2tError("Communicator: rank 0 could not draw an id: " + bytes(uid[1:]).decode(errors="replace"))
        if not isinstance(uid, (bytes, bytearray)) or len(uid) != int(self._L.js2t_comm_unique_id_bytes()):
            raise Js2tError("Communicator: the exchanged id is not an ncclUniqueId")
        self._h = C.c_void_p()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        check(self._L.js2t_comm_init(C.byref(self._h), bytes(uid), len(uid), self.world, self.rank, int(idx)), "js2t_comm_init")
        self._stream = None

    @classmethod
    def from_process_group(cls, device) -> "Communicator":
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise Js2tError("Communicator.from_process_group: no torch.distributed group to broadcast the id over")
        return cls(dist.get_rank(), dist.get_world_size(), device)

    @property
    def stream(self) -> "torch.cuda.Stream":
        """The communicator's stream as a torch stream (casts into / out of a staging buffer go between the collectives)."""
        if self._stream is None:
            self._stream = torch.cuda.ExternalStream(int(self._L.js2t_comm_stream(self._h)), device=self.device)
        return self._stream

    def all_reduce_async(self, t: torch.Tensor, average: bool = True, producer: Optional["torch.cuda.Stream"] = None):
        """In place over all ranks, behind what `producer` (default: the current stream) has been given; returns at once."""
        if self._h is None:
            raise Js2tError("Communicator: closed")
        if not t.is_cuda or not t.is_contiguous() or t.dtype not in (torch.float32, torch.bfloat16):
            raise Js2tError("Communicator.all_reduce_async: contiguous float32 / bfloat16 GPU tensor")
        s = producer if producer is not None else torch.cuda.current_stream(self.device)
        check(self._L.js2t_comm_allreduce_async(self._h, _p(t), t.numel(), dt_code(t), int(bool(average)), C.c_void_p(s.cuda_stream)),
              "js2t_comm_allreduce_async")

    def wait(self, consumer: Optional["torch.cuda.Stream"] = None, host: bool = False):
        """`consumer` (default: the current stream) waits on the device for every collective issued so far; host=True: this thread."""
        if self._h is None:
            raise Js2tError("Communicator: closed")
        s = consumer if consumer is not None else torch.cuda.current_stream(self.device)
        check(self._L.js2t_comm_wait(self._h, C.c_void_p(s.cuda_stream), int(bool(host))), "js2t_comm_wait")

    def close(self):
        if getattr(self, "_h", None) is not None:
            h, self._h = self._h, None
            check(self._L.js2t_comm_destroy(h), "js2t_comm_destroy")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _broadcast_id(mine: Optional[bytes]) -> bytes:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        if mine is None:
            raise Js2tError("Communicator: no process group and no id (pass exchange_id=)")
        return mine  # a single process
    box = [mine]
    dist.broadcast_object_list(box, src=0)
    return box[0]
