"""CPU tier: the C-ABI library loads without a GPU and exports every symbol include/telomeri_hip.h declares;
compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import subprocess

import pytest

from telomeri_amd import capi


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    syms = capi.declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert L.tlm_abi_version() == 4


def test_walk_seed_matches_oracle():
    from oracle import pyoracle as po
    for metric in range(4):
        for node, r in ((0, 0), (1202, 1999), (2 ** 31 - 1, 7), (12345, 2 ** 31)):
            assert capi.walk_seed(metric, node, r) == po.walk_seed(metric, node, r)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present: covered by the gpu tier")
def test_no_cpu_fallback_without_gpu():
    with pytest.raises(capi.TlmError) as e:
        capi.Context(0)
    assert "hip" in str(e.value).lower() or "device" in str(e.value).lower()


def test_library_does_not_link_the_oracle():
    out = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    nm = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "tlo_" not in nm


def test_product_sources_do_not_reference_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for d in ("telomeri_amd", "include"):
        for dp, _, files in os.walk(os.path.join(root, d)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if "pyoracle" in txt or "telomeri_oracle" in txt or "import oracle" in txt or "from oracle" in txt:
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
