"""Launch plans, host side (no GPU): the DA_FN_* table of include/diffusers_amd.h against the ctypes signatures, argument
encoding of the recorder, deep copies and address relocation inside da_plan (csrc/plan.hip is host code: it runs here)."""
import ctypes as C
import re
import struct
from pathlib import Path

import pytest

from diffusers_amd import _lib as L, plan as P

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "diffusers_amd.h").read_text()


def test_fn_ids_and_argument_layouts_match_the_header_and_the_signatures():
    lib = L.load()
    defs = {m.group(1): (int(m.group(2)), m.group(3)) for m in
            re.finditer(r"#define (DA_FN_\w+) (\d+)\s*/\* (da_\w+) \*/", HEADER)}
    assert len(defs) == len(L.FN_IDS) == L.FN_COUNT - 1
    assert int(re.search(r"#define DA_FN_COUNT (\d+)", HEADER).group(1)) == L.FN_COUNT
    assert int(re.search(r"#define DA_PLAN_MAX_ARGS (\d+)", HEADER).group(1)) == L.PLAN_MAX_ARGS
    assert {name: i for i, name in defs.values()} == L.FN_IDS          # same entry point behind the same number
    kind_of = {C.c_void_p: "p", C.c_int: "in", C.c_longlong: "l", C.c_float: "f"}
    for name, fid in L.FN_IDS.items():
        _, argtypes = L.SIGNATURES[name]
        kinds = lib.da_plan_arg_kinds(fid).decode()
        assert lib.da_plan_arg_count(fid) == len(kinds) == len(argtypes) - 1 <= L.PLAN_MAX_ARGS, name
        assert argtypes[-1] is C.c_void_p                               # the stream, not part of an op
        for k, t in zip(kinds, argtypes[:-1]):
            if t in kind_of:
                assert k in kind_of[t], (name, k, t)
            elif t == C.POINTER(L.GemmParams):
                assert k == "G", name
            elif t == C.POINTER(L.AttentionParams):
                assert k == "A", name
            else:
                assert k in "IQ", (name, k, t)
    assert lib.da_plan_arg_count(0) == -1 and lib.da_plan_arg_count(L.FN_COUNT) == -1 and lib.da_plan_arg_kinds(99) is None


def test_struct_sizes_in_the_plan_file_format_are_the_header_sizes(tmp_path):
    """examples/abi_demo.cpp refuses a plan file whose parameter blobs are not sizeof(da_gemm_params) / sizeof(da_attention_params):
    the ctypes mirrors must have exactly the C layout."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    src = tmp_path / "s.cpp"
    src.write_text('#include <cstdio>\n#include "diffusers_amd.h"\nint main() { std::printf("%zu %zu %zu\\n", '
                   'sizeof(da_gemm_params), sizeof(da_attention_params), sizeof(da_plan_op)); }\n')
    exe = tmp_path / "s"
    subprocess.run([gxx, f"-I{ROOT / 'include'}", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert [int(v) for v in out] == [C.sizeof(L.GemmParams), C.sizeof(L.AttentionParams), C.sizeof(L.PlanOp)]


def test_recorder_encodes_arguments_and_copies_structs():
    rec = P.Recorder()
    p = L.GemmParams()
    p.A, p.M = 0x1000, 64
    rec.note("da_gemm_bf16", L.FN_IDS["da_gemm_bf16"], (C.byref(p), 0x77))
    p.M = 1                                                             # the caller reuses its struct: the recording must not see it
    assert rec.keep[0].M == 64 and rec.keep[0].A == 0x1000 and rec.ops[0].arg[0] == C.addressof(rec.keep[0])
    # da_mul_scalar(x, out, s, rep, n, dtype, stream): pointer / None / float / negative int / long long
    rec.note("da_mul_scalar", L.FN_IDS["da_mul_scalar"], (0x2000, None, -1.5, -1, 1 << 40, 1, C.c_void_p(0x77)))
    a = rec.ops[1].arg
    assert a[0] == 0x2000 and a[1] == 0 and a[2] == struct.unpack("<I", struct.pack("<f", -1.5))[0]
    assert a[3] == 0xFFFFFFFFFFFFFFFF and a[4] == 1 << 40 and a[5] == 1
    assert rec.names == ["da_gemm_bf16", "da_mul_scalar"] and rec.streams == {0x77}
    with pytest.raises(TypeError):
        rec.note("da_mul_scalar", L.FN_IDS["da_mul_scalar"], (0, 0, 1.0))


def _plan(ops):
    lib = L.load()
    arr = (L.PlanOp * max(1, len(ops)))(*ops)
    h = C.c_void_p()
    rc = lib.da_plan_create(arr, len(ops), C.byref(h))
    return lib, rc, h


def test_plan_create_copies_what_it_is_given_and_relocates_device_addresses():
    lib = L.load()
    g = L.GemmParams()
    g.A, g.W, g.C, g.bias = 0x10000, 0x20000, 0x10800, 0
    op0 = L.PlanOp()
    op0.fn, op0.arg[0] = L.FN_IDS["da_gemm_bf16"], C.addressof(g)
    op1 = L.PlanOp()                                                    # da_transpose_bf16(in, out, R, C, ldi, ldo)
    op1.fn = L.FN_IDS["da_transpose_bf16"]
    op1.arg[0], op1.arg[1], op1.arg[2], op1.arg[3], op1.arg[4], op1.arg[5] = 0x10010, 0x30000, 8, 8, 8, 8
    # da_rmsnorm_rope_bf16: host arrays of `parts` entries are copied, the device pointers inside them relocated
    col = (C.c_int * 2)(0, 64)
    wts = (C.c_void_p * 2)(0x20040, None)
    op2 = L.PlanOp()
    op2.fn = L.FN_IDS["da_rmsnorm_rope_bf16"]
    vals = [0x10020, 128, 16, 16, 2, 64, 2, C.addressof(col), C.addressof(wts), 0, 0x20080, 0x200c0, 0, 1]
    for i, v in enumerate(vals):
        op2.arg[i] = v
    lib, rc, h = _plan([op0, op1, op2])
    assert rc == L.DA_OK and lib.da_plan_op_count(h) == 3
    g.A = 0xdead                                                        # the plan holds copies: later edits do not reach it
    col[1] = 7
    wts[0] = 0xbeef

    def relocate(regions):
        n = len(regions)
        old = (C.c_void_p * n)(*[r[0] for r in regions])
        size = (C.c_ulonglong * n)(*[r[1] for r in regions])
        new = (C.c_void_p * n)(*[r[2] for r in regions])
        miss = C.c_int(-1)
        assert lib.da_plan_relocate(h, n, old, size, new, C.byref(miss)) == L.DA_OK
        return miss.value
    # device addresses in the plan: gemm A W C (3) + transpose in / out (2) + rope x, weight[0], cos, sin (4); NULLs are skipped
    assert relocate([]) == 9
    assert relocate([(0x10000, 0x1000, 0x50000)]) == 9 - 4               # A, C, transpose in, rope x fall into the region
    assert relocate([(0x10000, 0x1000, 0x60000)]) == 9                   # ... and have moved out of it
    assert relocate([(0x50000, 0x1000, 0x10000), (0x20000, 0x100, 0x70000), (0x30000, 8, 0x80000)]) == 0
    assert relocate([(0x70000, 0x100, 0x20000)]) == 9 - 4                # W, weight[0], cos, sin: had the edits reached the plan,
    lib.da_plan_destroy(h)                                               # 0xbeef would not be among them


def test_plan_create_rejects_malformed_ops():
    bad = L.PlanOp()
    bad.fn = 0
    assert _plan([bad])[1] == 1
    bad.fn = L.FN_COUNT
    assert _plan([bad])[1] == 1
    bad.fn = L.FN_IDS["da_gemm_bf16"]                                    # NULL parameter struct
    assert _plan([bad])[1] == 1
    lib, rc, h = _plan([])
    assert rc == L.DA_OK and lib.da_plan_op_count(h) == 0
    failed = C.c_int(5)
    assert lib.da_plan_launch(h, None, C.byref(failed)) == L.DA_OK and failed.value == -1     # an empty plan launches nothing
    lib.da_plan_destroy(h)
    assert lib.da_plan_launch(None, None, None) == 1 and lib.da_plan_op_count(None) == -1
    assert lib.da_plan_create(None, 0, None) == 1


def test_recording_is_per_thread_and_exclusive():
    import threading
    with P.recording(private_pool=False) as rec:
        assert L.load() is rec.proxy and getattr(rec.proxy, "da_version")() == L.ABI_VERSION       # queries pass through
        seen = []
        t = threading.Thread(target=lambda: seen.append(L.load()))
        t.start(); t.join()
        assert seen[0] is L._lib
        with pytest.raises(RuntimeError, match="already being recorded"):
            with P.recording(private_pool=False):
                pass
    assert L.load() is L._lib and rec.ops == []


def test_plan_file_round_trip_and_python_side_loader(tmp_path):
    """The plan file format (plan.write_file / read_file, the one examples/abi_demo.cpp parses): a synthetic two-region, three-op
    file survives a round trip byte for byte, a loader rebuilds the da_plan from it and relocates it onto new base addresses, and
    malformed files are refused with a reason (wrong magic, truncation, a parameter struct of another ABI, an address outside the
    file's regions)."""
    lib = L.load()
    g = L.GemmParams()
    g.A, g.W, g.C, g.M, g.N, g.K = 0x10000, 0x20000, 0x10800, 64, 64, 64
    a = L.AttentionParams()
    a.q, a.k, a.vt, a.out = 0x10100, 0x10200, 0x10300, 0x10400
    col = (C.c_int * 2)(0, 64)
    wts = (C.c_void_p * 2)(0x20040, None)
    rope = [0x10020, 128, 16, 16, 2, 64, 2, 0xAAAA, 0xBBBB, 0, 0x20080, 0x200c0, 0, 1] + [0, 0]
    regions = [(0x10000, 0x1000, bytes(range(256)) * 16), (0x20000, 0x100, b"\x07" * 0x100)]
    ops = [(L.FN_IDS["da_gemm_bf16"], [0x1234] + [0] * 15, [bytes(g)]),
           (L.FN_IDS["da_attention_bf16"], [0x5678] + [0] * 15, [bytes(a)]),
           (L.FN_IDS["da_rmsnorm_rope_bf16"], rope, [bytes(col), bytes(wts)])]
    outs = [(0x10800, b"\x01\x02\x03\x04")]
    path = tmp_path / "p.daplan"
    P.write_file(path, regions, ops, outs)
    doc = P.read_file(path)
    assert doc["regions"] == regions and doc["outputs"] == outs
    assert [(fn, blobs) for fn, _, blobs in doc["ops"]] == [(fn, blobs) for fn, _, blobs in ops]
    assert doc["ops"][2][1] == rope
    h, _ = P.create_from_file(path, new_bases=[0x70000, 0x80000])
    assert lib.da_plan_op_count(h) == 3
    # every address moved with its region: nothing is left inside the old ranges, everything is inside the new ones
    n = 2
    miss = C.c_int(-1)
    old = (C.c_void_p * n)(0x10000, 0x20000)
    size = (C.c_ulonglong * n)(0x1000, 0x100)
    new = (C.c_void_p * n)(0x10000, 0x20000)
    assert lib.da_plan_relocate(h, n, old, size, new, C.byref(miss)) == L.DA_OK and miss.value == 3 + 4 + 4   # gemm 3, attention 4, rope 4
    old = (C.c_void_p * n)(0x70000, 0x80000)
    assert lib.da_plan_relocate(h, n, old, size, new, C.byref(miss)) == L.DA_OK and miss.value == 0
    lib.da_plan_destroy(h)
    with pytest.raises(ValueError, match="lie in no region"):
        P.write_file(path, regions[:1], ops, outs)
        P.create_from_file(path, new_bases=[0x70000])
    P.write_file(path, regions, ops, outs)
    raw = path.read_bytes()
    bad = tmp_path / "bad.daplan"
    bad.write_bytes(b"NOTAPLAN" + raw[8:])
    with pytest.raises(ValueError, match="not a plan file"):
        P.read_file(bad)
    bad.write_bytes(raw[:-2])
    with pytest.raises(ValueError, match="truncated"):
        P.read_file(bad)
    bad.write_bytes(raw + b"\x00")
    with pytest.raises(ValueError, match="trailing"):
        P.read_file(bad)
    with pytest.raises(ValueError, match="host blobs"):
        P.write_file(bad, regions, [(L.FN_IDS["da_gemm_bf16"], [0] * 16, [])], outs)
    P.write_file(bad, regions, [(L.FN_IDS["da_gemm_bf16"], [0] * 16, [bytes(g)[:-8]])], outs)
    with pytest.raises(ValueError, match="parameter struct"):
        P.read_file(bad)
