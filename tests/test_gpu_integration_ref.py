"""The drop-in boundary exercised through the reference's OWN TYPES (oracle/_ref/integration_ref: INTEGRATION.md's bindings compiled
against /root/reference/src where it lies, linked with libmara_hip.so; the binary travels to the GPU box): one nd::shared_array is
stepped by the reference's lazy-array composition of advance / next_solution (src/subprog_cloud.cpp:511-584, :676-697) and by the
binding over the C ABI, and the two results are compared bit for bit inside the program."""
import os
import subprocess

import pytest
from conftest import ROOT

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
EXE = os.path.join(ROOT, "oracle", "_ref", "integration_ref")


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/integration_ref is built where the reference tree is present")
@pytest.mark.parametrize("args,expect", [(["euler", "96", "200", "3"], 2), (["euler", "257", "130", "2"], 2), (["cloud", "32", "3"], 1), (["cloud", "70", "2"], 1)])
def test_reference_types_through_the_binding_equal_the_reference_composition(args, expect):
    p = subprocess.run([EXE] + args, capture_output=True, text=True, timeout=500)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count("(bit-identical)") == expect and " 0 values differ" in p.stdout
