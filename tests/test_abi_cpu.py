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


def test_block_layout_matches_reference_access_patterns(lib):
    """mh_block_layout = create_access_pattern_array over propose_block_decomposition<3> (src/app_parallel.hpp:119-179): blocks per axis
    and per-axis extents against the tables produced by the reference's own headers, for every rank; too many blocks is refused where the
    reference throws std::logic_error."""
    from mara3_amd.block import block_layout
    import mara3_amd
    g = golden("decomposition")
    shape = (4096, 1000, 640)
    for world in (1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16):
        B = tuple(int(x) for x in g["decomp_rank3"][world - 1])
        seen = set()
        for rank in range(world):
            Bg, c, s, k = block_layout(shape, world, rank)
            assert Bg == B and c == (rank // (B[1] * B[2]), (rank // B[2]) % B[1], rank % B[2])
            for a, N in enumerate(shape):
                table = g["blocks_%d_%d" % (N, B[a])]
                assert (s[a], s[a] + k[a]) == tuple(int(x) for x in table[c[a]]), (world, rank, a)
            seen.add(c)
        assert len(seen) == world
    with pytest.raises(mara3_amd.MaraHipError, match="too many blocks"):
        block_layout((7, 7, 7), 1024, 0)          # (8, 8, 16) blocks of a 7^3 grid
