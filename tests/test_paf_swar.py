"""telomeri_amd/csrc/paf_swar.h -- the arithmetic of the device tokeniser (csrc/paf_parse.hip): SWAR byte classes, the strict
integer read out of a word array, the word-wise name hash.  The header compiles for the host too: tests/paf_swar_check.cpp
runs every function against a byte-at-a-time restatement (every byte value in every position, every digit group, integer
tokens of every shape at every alignment and terminator, names of every length and alignment).  CPU tier: the kernels that
use the header are compared with the host tokeniser on the GPU tier (tests/test_host_cli.py)."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tokeniser_arithmetic_on_the_host(tmp_path):
    exe = tmp_path / "paf_swar_check"
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I", os.path.join(REPO, "telomeri_amd", "csrc"),
                    os.path.join(REPO, "tests", "paf_swar_check.cpp"), "-o", str(exe)], check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.startswith("ok ") and int(res.stdout.split()[1]) > 1_000_000
