"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/mara_hip.h declares (no compute calls without a GPU), and its host-side integer work is
bit-exact with the reference-generated tables."""
import os
import re
import ctypes as C
import numpy as np
import pytest
from conftest import golden, ROOT


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    __graft_entry__.build()
    from mara3_amd import _lib
    return _lib.load_library()


def test_header_and_binding_agree(lib):
    from mara3_amd import _lib
    text = open(os.path.join(ROOT, "include", "mara_hip.h")).read()
    declared = set(re.findall(r"\b(mh_[a-z0-9_]+)\s*\(", text))
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared - bound, bound - declared)
    for name in declared:
        assert hasattr(lib, name), name


def test_no_gpu_is_reported_not_faked(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lib.mh_device_count() == 0
    ctx = C.c_void_p()
    assert lib.mh_create(C.byref(ctx), 0) != 0          # fails loudly: no CPU fallback
    assert b"" != lib.mh_last_error(None)


def test_partition_and_block_decomposition_bit_exact(lib):
    g = golden("decomposition")
    out = (C.c_ulong * 3)()
    for rank in (1, 2, 3):
        table = g["decomp_rank%d" % rank]
        for n in range(1, 33):
            assert lib.mh_propose_block_decomposition(rank, n, out) == 0
            assert tuple(out[i] for i in range(rank)) == tuple(table[n - 1])
    a, b = C.c_size_t(), C.c_size_t()
    for key in g.files:
        if key.startswith("blocks_"):
            _, count, nparts = key.split("_")
            for p in range(int(nparts)):
                lib.mh_partition_rows(int(count), int(nparts), p, C.byref(a), C.byref(b))
                assert (a.value, b.value) == tuple(g[key][p]), key
    # evaluate_on<8> on 4097 rows
    lib.mh_partition_rows(4097, 8, 7, C.byref(a), C.byref(b))
    assert (a.value, b.value) == (3584, 4097)
