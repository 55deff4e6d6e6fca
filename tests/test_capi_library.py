"""CPU tier: the C-ABI library loads without a GPU and exports every symbol include/telomeri_hip.h declares;
compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import subprocess

import pytest

from telomeri_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    syms = capi.declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert L.tlm_abi_version() == 5


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


def test_cold_args_only_in_kernels():
    """csrc/walk_kernels.hip reads its "cold" kernel arguments through __builtin_amdgcn_kernarg_segment_ptr() (cold_args()).
    In a function that stays a REAL CALL that intrinsic yields a null pointer: the arguments would be read from address 0
    (round 3's kernel that "died on a null address", round 4's first far filter).  So every user of cold_args() must have been
    inlined into a kernel: the ISA of the shipped flags is read here -- the marker of cold_args()'s asm statement may occur in
    kernel bodies only, and every non-kernel function is one of the known leaf helpers."""
    import re
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "telomeri_amd", "csrc", "walk_kernels.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "wk.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                        "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"),
                        "--cuda-device-only", "-S", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    kernels = set(re.findall(r"\.amdhsa_kernel (\S+)", txt))
    assert len(kernels) == 12, kernels          # {MC, DET} x {plain, sqrt} x {full, lite, accounting}
    seen_kernels = 0
    for body in re.split(r"\n(?=[_A-Za-z0-9\.]+:\s+; @)", txt):
        m = re.match(r"([_A-Za-z0-9\.]+):\s+; @", body)
        if not m:
            continue
        name, marks = m.group(1), body.count("; cold_args")
        if name in kernels:
            seen_kernels += 1
            assert marks >= 3, name
            continue
        assert marks == 0, "%s is a real call and reads the kernarg segment (a null pointer there)" % name
        assert any(k in name for k in ("mt_seed", "mt_twist", "band_slide", "select_long", "select_serial", "det_select_long")), name
    assert seen_kernels == 12


def test_product_library_carries_no_test_hooks():
    """The fault-injection switch of the exchange lives in the checked build only; the product neither
    reads it nor can be talked onto another architecture."""
    product = open(os.path.join(ROOT, "telomeri_amd", "lib", "libtelomeri_hip.so"), "rb").read()
    assert b"TLM_EXCHANGE_FAULT" not in product and b"TLM_ALLOW_ANY_ARCH" not in product
    checked = os.path.join(ROOT, "telomeri_amd", "lib_checked", "libtelomeri_hip.so")
    if os.path.exists(checked):
        assert b"TLM_EXCHANGE_FAULT" in open(checked, "rb").read()


def test_sharded_load_runs_partition_the_text():
    """tlm_dist_text_run (ABI 5; plain arithmetic, no device): the ranks' runs of 1 GiB pieces are contiguous, disjoint, cover
    [0, bytes) in rank order, start on piece boundaries and differ by at most one piece -- what csrc/exchange.hip's swap and
    the hosts' uploads both rely on (the sends of rank a to b must meet the receives of b from a, length for length)."""
    L = capi.lib()
    piece = 1 << 30
    for world in (1, 2, 3, 4, 7, 8):
        for nbytes in (0, 1, piece - 1, piece, piece + 1, 3 * piece + 5, 15 * piece + 123, 16 * piece, 57 * piece + 1, 1000 * piece - 7):
            at, sizes = 0, []
            for r in range(world):
                b0, b1 = C.c_uint64(), C.c_uint64()
                assert L.tlm_dist_text_run(nbytes, r, world, C.byref(b0), C.byref(b1)) == 0
                assert b0.value == at and b1.value >= b0.value, (world, nbytes, r, b0.value, b1.value)
                assert b0.value % piece == 0 or b0.value == nbytes
                at = b1.value
                sizes.append(-(-(b1.value - b0.value) // piece))
            assert at == nbytes
            assert max(sizes) - min(sizes) <= 1, (world, nbytes, sizes)
    b0, b1 = C.c_uint64(), C.c_uint64()
    assert L.tlm_dist_text_run(10, 2, 2, C.byref(b0), C.byref(b1)) != 0        # rank outside the world
