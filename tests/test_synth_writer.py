"""Bench infrastructure: the all-cores text writer (telomeri_amd/csrc/synth_writer.c) prints exactly what the Python
writers of telomeri_amd.synth print -- the scaffold wall-clock leg of bench.py feeds the host program with its output."""
import os
import tempfile

import helpers as H
from telomeri_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe():
    from telomeri_amd import scaffold_bench
    return scaffold_bench


def test_fast_writers_match_python_writers():
    sb = _probe()
    for ds in (synth.make_config("tiny"), H.quirks_dataset()):
        with tempfile.TemporaryDirectory() as tmp:
            p = lambda n: os.path.join(tmp, n)
            for rec, name in ((ds.rr, "rr"), (ds.cr, "cr")):
                synth.write_paf(ds, rec, p(name + ".py.paf"))
                n = sb.write_paf(ds, rec, p(name + ".c.paf"), threads=3)
                a, b = open(p(name + ".py.paf"), "rb").read(), open(p(name + ".c.paf"), "rb").read()
                assert a == b and n == len(b)
            synth.write_fasta(ds, p("r.py.fa"), p("c.py.fa"), seed=5)
            n = sb.write_fasta(ds, p("r.c.fa"), p("c.c.fa"), seed=5)
            assert open(p("r.py.fa"), "rb").read() == open(p("r.c.fa"), "rb").read()
            assert open(p("c.py.fa"), "rb").read() == open(p("c.c.fa"), "rb").read()
            assert n == os.path.getsize(p("r.c.fa")) + os.path.getsize(p("c.c.fa"))
