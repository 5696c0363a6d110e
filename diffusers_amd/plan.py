"""Launch plans: one denoising step (or a decode) as a list of C-ABI launches that the library replays by itself.

``include/diffusers_amd.h`` ("launch plans") / ``csrc/plan.hip`` own the replay; this module is the recorder.  Run any piece of
engine code under :func:`record` and every launch entry point it calls is executed AND noted with its arguments; the resulting
:class:`Plan` replays those launches through ``da_plan_launch`` with no Python between them -- the same calls in the same order
on the same buffers, hence bit-identical results.  It is to the C ABI what the captured HIP graph is to the Python pipelines
(the reference's per-step work: ``pipeline_stable_diffusion_xl.py:1186-1250``), and what a host without Python uses to run a
whole step: :meth:`Plan.save` writes the ops, the device regions they touch and the expected outputs to one file that
``examples/abi_demo.cpp`` loads, relocates (``da_plan_relocate``), launches and checks.

Buffers: a plan holds raw device addresses.  Tensors allocated while recording come from a private ``torch.cuda.MemPool`` that
the Plan keeps alive (what graph capture does with its pool); tensors that existed before (weights, inputs, step counters) must
outlive the plan -- pass them as ``keep=``.

Work that is NOT a launch of this library (a stray ``torch.cat`` in the recorded code) cannot be replayed; the recorder lists
such torch operators in ``Plan.foreign_ops`` and ``record(strict=True)`` (the default) raises on them.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import struct
from pathlib import Path
from typing import Callable, List, Optional, Sequence

import torch

from . import _lib as L

MAGIC = b"DAPLAN01"

# aten operators that launch nothing: allocation, views, metadata
_NO_KERNEL = {
    "empty", "empty_strided", "empty_like", "new_empty", "new_empty_strided", "view", "_unsafe_view", "reshape", "_reshape_alias",
    "as_strided", "permute", "transpose", "t", "slice", "select", "expand", "squeeze", "unsqueeze", "detach", "alias", "unbind",
    "split", "split_with_sizes", "narrow", "unfold", "flatten", "unflatten", "view_as", "contiguous", "lift_fresh", "sym_size",
    "sym_stride", "sym_numel", "sym_storage_offset", "is_contiguous", "size", "stride", "numel", "dim", "chunk", "movedim",
    "swapaxes", "resize_",
}


def _float_bits(v: float) -> int:
    return struct.unpack("<I", struct.pack("<f", float(v)))[0]


def _addr(v) -> int:
    if v is None:
        return 0
    if isinstance(v, int):
        return v
    if isinstance(v, C.c_void_p):
        return v.value or 0
    if isinstance(v, (C.Array, C.Structure)):
        return C.addressof(v)
    if hasattr(v, "_obj"):                      # ctypes.byref(x)
        return C.addressof(v._obj)
    if hasattr(v, "contents"):                  # ctypes.pointer(x) / cast(...)
        return C.cast(v, C.c_void_p).value or 0
    raise TypeError(f"cannot take the address of {type(v).__name__}")


class _Proxy:
    """Stands in for the ctypes library while a recorder is active: launch entry points are executed and noted, everything else
    (queries, the tuner) passes through."""

    def __init__(self, lib, rec: "Recorder"):
        self._lib, self._rec = lib, rec

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        fid = L.FN_IDS.get(name)
        if fid is None:
            return fn
        rec = self._rec

        def call(*args):
            status = fn(*args)
            if status == L.DA_OK:               # a refused launch (DA_ERR_UNSUPPORTED -> the caller retries another tile) is not part of the step
                rec.note(name, fid, args)
            return status
        return call


class Recorder:
    def __init__(self):
        self.ops: List[L.PlanOp] = []
        self.names: List[str] = []
        self.keep: list = []                    # host objects the ops point at until da_plan_create has copied them
        self.foreign: List[str] = []
        self.streams = set()
        self.proxy = None

    def note(self, name: str, fid: int, args: Sequence) -> None:
        _, argtypes = L.SIGNATURES[name]
        if len(args) != len(argtypes):
            raise TypeError(f"{name}: {len(args)} arguments, the ABI has {len(argtypes)}")
        op = L.PlanOp()
        op.fn = fid
        for i, (a, t) in enumerate(zip(args[:-1], argtypes[:-1])):          # the trailing argument is the stream
            if t in (C.c_int, C.c_longlong):
                op.arg[i] = int(a) & 0xFFFFFFFFFFFFFFFF
            elif t is C.c_float:
                op.arg[i] = _float_bits(a)
            elif t is C.c_void_p:
                op.arg[i] = _addr(a)
            else:                                                            # POINTER(struct) / POINTER(c_int) / POINTER(c_void_p)
                obj = a._obj if hasattr(a, "_obj") else a
                if isinstance(obj, C.Structure):
                    obj = type(obj).from_buffer_copy(obj)                    # the caller may reuse / mutate its struct
                self.keep.append(obj)
                op.arg[i] = _addr(obj)
        self.streams.add(_addr(args[-1]))
        self.ops.append(op)
        self.names.append(name)


class _ForeignOps(torch.utils._python_dispatch.TorchDispatchMode):
    """Notes every aten operator on device tensors that is not pure allocation / view work while a plan is recorded."""

    def __init__(self, rec: Recorder):
        super().__init__()
        self.rec = rec

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
        if name not in _NO_KERNEL:
            flat = list(args) + list((kwargs or {}).values()) + ([out] if isinstance(out, torch.Tensor) else [])
            if any(isinstance(t, torch.Tensor) and t.is_cuda for t in flat):
                self.rec.foreign.append(name)
        return out


class Plan:
    """A created ``da_plan``.  ``launch()`` replays it on the current (or given) stream."""

    def __init__(self, rec: Recorder, keep=(), pool=None):
        lib = L._lib if L._lib is not None else L.load()
        self.names = list(rec.names)
        self.foreign_ops = list(rec.foreign)
        self.ops = (L.PlanOp * max(1, len(rec.ops)))(*rec.ops)   # kept: save() writes them; the structs they point at too
        self._host = rec.keep
        self._keep, self._pool = list(keep), pool
        self._lib = lib
        handle = C.c_void_p()
        L.check(lib.da_plan_create(self.ops, len(rec.ops), C.byref(handle)), "da_plan_create")
        self._h = handle

    def __len__(self) -> int:
        return self._lib.da_plan_op_count(self._h)

    def launch(self, stream: Optional[int] = None) -> None:
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        failed = C.c_int(-1)
        rc = self._lib.da_plan_launch(self._h, st, C.byref(failed))
        if rc != L.DA_OK:
            where = self.names[failed.value] if 0 <= failed.value < len(self.names) else "?"
            L.check(rc, f"da_plan_launch (op {failed.value}: {where})")

    replay = launch        # the name torch.cuda.CUDAGraph gives the same act (pipelines replay either)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.da_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown
            pass

    # ---- device addresses the plan touches ---------------------------------------------------------------------------
    def device_pointers(self) -> List[int]:
        """Every distinct non-NULL device address among the recorded arguments (struct fields and pointer arrays included)."""
        lib, out = self._lib, set()
        host = iter(self._host)
        for op in self.ops[:len(self.names)]:
            kinds = lib.da_plan_arg_kinds(op.fn).decode()
            parts = 0
            for i, k in enumerate(kinds):
                if k == "p":
                    out.add(op.arg[i])
                elif k == "n":
                    parts = op.arg[i]
                elif k in "GA":
                    obj = next(host)
                    out.update(getattr(obj, f) or 0 for f, t in obj._fields_ if t is C.c_void_p)
                elif k == "I":
                    next(host)
                elif k == "Q":
                    out.update((next(host)[j] or 0) for j in range(parts))
        out.discard(0)
        return sorted(out)

    def regions(self) -> List[tuple]:
        """(base, bytes) of the allocator SEGMENTS (the driver-level allocations) holding the plan's device addresses.  Segments,
        not blocks: an activation recorded inside the plan's private pool has been freed since, and the pool's free blocks are
        split and merged as it is reused -- the block around an address today is not the tensor it was recorded as."""
        import bisect
        segs = sorted((seg["address"], seg["total_size"]) for seg in torch.cuda.memory_snapshot())
        starts = [b for b, _ in segs]
        found = {}
        for p in self.device_pointers():
            i = bisect.bisect_right(starts, p) - 1
            if i < 0 or p >= segs[i][0] + segs[i][1]:
                raise RuntimeError(f"device address {p:#x} is not inside any segment of torch's allocator")
            found[segs[i][0]] = segs[i][1]
        return sorted(found.items())

    def save(self, path, outputs: Sequence[torch.Tensor], stream: Optional[int] = None, max_bytes: int = 1 << 30) -> dict:
        """Write the plan for a host without Python (``examples/abi_demo.cpp``): the ops with their parameter structs, the
        contents of every device region they touch AS THEY ARE NOW (= the state the replay starts from: restore the inputs
        before calling), and the bytes of ``outputs`` after one launch from that state.  The state is put back afterwards."""
        regs = self.regions()
        total = sum(n for _, n in regs)
        if total > max_bytes:
            raise ValueError(f"the plan's device regions hold {total / 2**30:.1f} GiB (whole allocator segments: weights included); "
                             f"raise max_bytes to write them, or ship the weights separately and relocate onto them")
        torch.cuda.synchronize()
        views = [_device_bytes(b, n) for b, n in regs]
        before = [v.cpu() for v in views]
        self.launch(stream)
        torch.cuda.synchronize()
        outs = []
        for t in outputs:
            if not t.is_contiguous():
                raise ValueError("outputs must be contiguous tensors")
            outs.append((t.data_ptr(), t.numel() * t.element_size(), t.view(torch.uint8).reshape(-1).cpu() if t.numel() else None))
        for v, b in zip(views, before):
            v.copy_(b)
        torch.cuda.synchronize()
        host_blobs = []
        host = iter(self._host)
        for op in self.ops[:len(self.names)]:
            kinds = self._lib.da_plan_arg_kinds(op.fn).decode()
            parts, blobs = 0, []
            for i, k in enumerate(kinds):
                if k == "n":
                    parts = op.arg[i]
                if k in "GAIQ":
                    obj = next(host)
                    blobs.append(bytes(obj) if k in "GA" else bytes(obj)[: parts * (4 if k == "I" else 8)])
            host_blobs.append(blobs)
        write_file(path, [(b, n, data.numpy().tobytes()) for (b, n), data in zip(regs, before)],
                   [(op.fn, list(op.arg), blobs) for op, blobs in zip(self.ops[:len(self.names)], host_blobs)],
                   [(ptr, data.numpy().tobytes()) for ptr, n, data in outs])
        return {"regions": len(regs), "region_bytes": sum(n for _, n in regs), "ops": len(self.names), "outputs": len(outs),
                "file_bytes": Path(path).stat().st_size}


def write_file(path, regions, ops, outputs) -> None:
    """The plan file ``examples/abi_demo.cpp`` reads.  Little endian:
    ``"DAPLAN01"``, u32 regions, u32 ops, u32 outputs;
    per region: u64 base address in the recording process, u64 bytes, the bytes;
    per op: i32 DA_FN_*, i32 argument count, 16 x u64 argument slots, then -- in argument order -- one (u32 bytes, bytes) blob per
    argument the slot passes by HOST address (the da_gemm_params / da_attention_params struct, the host arrays of
    da_rmsnorm_rope_bf16; the slot's recorded address is meaningless to the reader);
    per output: u64 device address (inside a region), u64 bytes, the expected bytes."""
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<III", len(regions), len(ops), len(outputs)))
        for base, n, data in regions:
            if len(data) != n:
                raise ValueError("region size and contents disagree")
            f.write(struct.pack("<QQ", base, n))
            f.write(data)
        lib = L._lib if L._lib is not None else L.load()
        for fn, args, blobs in ops:
            kinds = lib.da_plan_arg_kinds(fn)
            if kinds is None or len(args) != L.PLAN_MAX_ARGS or len(blobs) != sum(k in "GAIQ" for k in kinds.decode()):
                raise ValueError(f"op {fn}: not an entry point of this library, or its host blobs do not match its argument kinds")
            f.write(struct.pack("<ii", fn, len(kinds)))
            f.write(struct.pack(f"<{L.PLAN_MAX_ARGS}Q", *args))
            for blob in blobs:
                f.write(struct.pack("<I", len(blob)))
                f.write(blob)
        for ptr, data in outputs:
            f.write(struct.pack("<QQ", ptr, len(data)))
            f.write(data)


def read_file(path) -> dict:
    """Inverse of :func:`write_file`: ``{"regions": [(base, bytes, data)], "ops": [(fn, [16 slots], [blobs])], "outputs": [(ptr,
    data)]}``.  Raises ``ValueError`` on anything that is not a well-formed plan file of THIS ABI (struct sizes are checked the way
    abi_demo.cpp checks them)."""
    lib = L._lib if L._lib is not None else L.load()
    raw = Path(path).read_bytes()
    pos = 0

    def take(n):
        nonlocal pos
        if pos + n > len(raw):
            raise ValueError("truncated plan file")
        out = raw[pos:pos + n]
        pos += n
        return out
    if take(8) != MAGIC:
        raise ValueError("not a plan file")
    n_reg, n_ops, n_out = struct.unpack("<III", take(12))
    regions, ops, outputs = [], [], []
    for _ in range(n_reg):
        base, n = struct.unpack("<QQ", take(16))
        regions.append((base, n, take(n)))
    want = {"G": C.sizeof(L.GemmParams), "A": C.sizeof(L.AttentionParams)}
    for o in range(n_ops):
        fn, nargs = struct.unpack("<ii", take(8))
        kinds = lib.da_plan_arg_kinds(fn)
        if kinds is None or len(kinds) != nargs:
            raise ValueError(f"op {o}: entry point {fn} is not one this library replays")
        args = list(struct.unpack(f"<{L.PLAN_MAX_ARGS}Q", take(8 * L.PLAN_MAX_ARGS)))
        blobs = []
        for k in kinds.decode():
            if k in "GAIQ":
                (n,) = struct.unpack("<I", take(4))
                if k in want and n != want[k]:
                    raise ValueError(f"op {o}: a {n}-byte parameter struct where this ABI has {want[k]}")
                blobs.append(take(n))
        ops.append((fn, args, blobs))
    for _ in range(n_out):
        ptr, n = struct.unpack("<QQ", take(16))
        outputs.append((ptr, take(n)))
    if pos != len(raw):
        raise ValueError("trailing bytes after the last output")
    return {"regions": regions, "ops": ops, "outputs": outputs}


def create_from_file(path, new_bases=None):
    """``(da_plan handle, file contents)`` of a plan file: what abi_demo.cpp does, from Python -- the ops rebuilt with their host
    blobs, `da_plan_create`, and (``new_bases``: one device address per region, e.g. ``tensor.data_ptr()`` of buffers holding the
    regions' contents) `da_plan_relocate` onto the caller's memory.  The caller launches with ``da_plan_launch`` and destroys the
    handle with ``da_plan_destroy``."""
    lib = L._lib if L._lib is not None else L.load()
    doc = read_file(path)
    keep, arr = [], (L.PlanOp * max(1, len(doc["ops"])))()
    for i, (fn, args, blobs) in enumerate(doc["ops"]):
        arr[i].fn = fn
        it = iter(blobs)
        for j, k in enumerate(lib.da_plan_arg_kinds(fn).decode()):
            if k in "GAIQ":
                buf = C.create_string_buffer(next(it))
                keep.append(buf)
                arr[i].arg[j] = C.addressof(buf)
            else:
                arr[i].arg[j] = args[j]
    h = C.c_void_p()
    L.check(lib.da_plan_create(arr, len(doc["ops"]), C.byref(h)), "da_plan_create")
    if new_bases is not None:
        n = len(doc["regions"])
        if len(new_bases) != n:
            lib.da_plan_destroy(h)
            raise ValueError(f"{n} regions in the file, {len(new_bases)} new base addresses")
        old = (C.c_void_p * n)(*[r[0] for r in doc["regions"]])
        size = (C.c_ulonglong * n)(*[r[1] for r in doc["regions"]])
        new = (C.c_void_p * n)(*new_bases)
        miss = C.c_int(0)
        L.check(lib.da_plan_relocate(h, n, old, size, new, C.byref(miss)), "da_plan_relocate")
        if miss.value:
            lib.da_plan_destroy(h)
            raise ValueError(f"{miss.value} device addresses of the plan lie in no region of the file")
    return h, doc


class _RawDevice:
    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _device_bytes(ptr: int, nbytes: int) -> torch.Tensor:
    """uint8 view of raw device memory (no copy)."""
    return torch.as_tensor(_RawDevice(ptr, nbytes), device="cuda")


@contextlib.contextmanager
def recording(strict: bool = True, private_pool: bool = True):
    """``with recording() as rec: ...engine calls...`` then ``Plan(rec)``; prefer :func:`record`."""
    if getattr(L._tls, "recorder", None) is not None:
        raise RuntimeError("a plan is already being recorded on this thread")
    rec = Recorder()
    rec.proxy = _Proxy(L._lib if L._lib is not None else L.load(), rec)
    rec.pool = torch.cuda.MemPool() if private_pool and torch.cuda.is_available() else None
    pool_ctx = torch.cuda.use_mem_pool(rec.pool) if rec.pool is not None else contextlib.nullcontext()
    L._tls.recorder = rec
    try:
        with pool_ctx, _ForeignOps(rec):
            yield rec
    finally:
        L._tls.recorder = None
    if strict and rec.foreign:
        raise RuntimeError(f"recorded code ran torch operators a plan cannot replay: {sorted(set(rec.foreign))}")
    if len(rec.streams) > 1:
        raise RuntimeError("recorded launches went to more than one stream; a plan replays on one")


def record(fn: Callable[[], object], keep=(), strict: bool = True):
    """Run ``fn()`` once, eagerly, and return ``(Plan, fn's result)``.  Run it un-recorded first so that shapes are tuned and
    lazily built caches exist (a tuner pass inside the recording is harmless but its trial launches are not skipped)."""
    with recording(strict=strict) as rec:
        result = fn()
    return Plan(rec, keep=list(keep) + [result], pool=rec.pool), result
