"""INTEGRATION.md's C++ code blocks are cut out of a translation unit that compiles against the reference's headers
(oracle/ref_drivers/integration_ref.cpp, built by oracle/Makefile with -I/root/reference/src and linked with libmara_hip.so).
Here, without a GPU: the document and the file agree, and - where the reference tree is present - the file compiles and links
(the binding needs no GPU for that) and says so when there is no device to run on."""
import os
import re
import subprocess

import pytest
from conftest import ROOT

SRC = os.path.join(ROOT, "oracle", "ref_drivers", "integration_ref.cpp")
EXE = os.path.join(ROOT, "oracle", "_ref", "integration_ref")


def region(text, name):
    begin, end = "// [integration:%s begin]\n" % name, "// [integration:%s end]" % name
    return text[text.index(begin) + len(begin):text.index(end)].rstrip("\n")


def test_the_code_blocks_of_the_document_are_the_compiled_ones():
    src = open(SRC).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = [b.rstrip("\n") for b in re.findall(r"```cpp\n(.*?)```", doc, flags=re.S)]
    for name, first_line in (("euler", '#include "mara_hip.h"'), ("cloud", '#include "physics_srhd.hpp"')):
        block = [b for b in blocks if b.startswith(first_line)]
        assert len(block) == 1, name
        assert block[0] == region(src, name), "INTEGRATION.md's %s block and oracle/ref_drivers/integration_ref.cpp have drifted apart" % name


def test_the_binding_compiles_against_the_reference_headers_and_links_with_the_library():
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("the reference tree is present in the build container only (the prebuilt binary travels)")
    lib = os.path.join(ROOT, "mara3_amd", "libmara_hip.so")
    assert os.path.exists(lib), "build the library first (__graft_entry__.build())"
    subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "_ref/integration_ref"])
    needed = subprocess.run(["ldd", EXE], capture_output=True, text=True, check=True).stdout
    assert "libmara_hip.so => " + os.path.realpath(lib) in needed.replace("/oracle/_ref/../..", "")
    import mara3_amd
    if mara3_amd.load_library().mh_device_count() == 0:
        p = subprocess.run([EXE, "euler", "32", "48", "1"], capture_output=True, text=True)
        assert p.returncode == 77 and "compiled and linked" in p.stdout
