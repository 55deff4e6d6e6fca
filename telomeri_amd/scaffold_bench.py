#!/usr/bin/env python3
"""The second half of BASELINE.json's metric -- scaffold wall-clock, PAF + FASTA text in, scaffold FASTA out -- for the
C++ host program `telomeri_hip` (reference argv, reference defaults: 2000 trials per anchor, 60 backtracks) on a synthetic
dataset of the given shape (reference surface: run.sh:27-47, Main.cpp:212-388).

  python -m telomeri_amd.scaffold_bench [config]          (default athaliana; run on the GPU box)

bench.py imports run() for its `scaffold_wall_s` block.  The input text (for A. thaliana: 150 M PAF lines = 15 GB, reads
FASTA 10 GB) is rendered by telomeri_amd/csrc/synth_writer.c on all cores into /dev/shm (or $TLM_SCRATCH) and deleted
afterwards; writing it is NOT part of the measured wall-clock, reading it is."""
import ctypes as C
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WRITER_SRC = os.path.join(HERE, "csrc", "synth_writer.c")
WRITER_LIB = os.path.join(HERE, "lib", "libsynth_writer.so")


def build_writer() -> str:
    if not os.path.exists(WRITER_LIB) or os.path.getmtime(WRITER_LIB) < os.path.getmtime(WRITER_SRC):
        os.makedirs(os.path.dirname(WRITER_LIB), exist_ok=True)
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-pthread", "-o", WRITER_LIB, WRITER_SRC], check=True)
    return WRITER_LIB


def _writer():
    L = C.CDLL(build_writer())
    L.synth_write_paf.restype = C.c_int64
    L.synth_write_paf.argtypes = [C.c_char_p, C.c_uint64] + [C.c_void_p] * 10 + [C.c_int64, C.c_int]
    L.synth_write_fasta.restype = C.c_int64
    L.synth_write_fasta.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
    return L


def write_paf(ds, rec, path: str, threads: int = 16) -> int:
    """Same text as telomeri_amd.synth.write_paf."""
    cols = [np.ascontiguousarray(getattr(rec, c).cpu().numpy(), np.int32) for c in
            ("q_id", "t_id", "q_len", "q_start", "q_end", "t_len", "t_start", "t_end", "matches", "strand")]
    n = _writer().synth_write_paf(path.encode(), len(rec), *[c.ctypes.data for c in cols], ds.n_reads, threads)
    if n < 0:
        raise OSError("synth_write_paf(%s) failed" % path)
    return int(n)


def write_fasta(ds, reads_path: str, contigs_path: str, seed: int = 0) -> int:
    """Same text as telomeri_amd.synth.write_fasta."""
    rng = np.random.default_rng(seed + 977)
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=ds.genome_len)]
    genome = np.ascontiguousarray(genome)
    L = _writer()
    total = 0
    for path, start, length, first in ((reads_path, ds.read_start, ds.read_len, 0),
                                       (contigs_path, ds.contig_start, ds.contig_len, ds.n_reads)):
        s = np.ascontiguousarray(start.cpu().numpy(), np.int64)
        ln = np.ascontiguousarray(length.cpu().numpy(), np.int64)
        n = L.synth_write_fasta(path.encode(), genome.ctypes.data, s.shape[0], s.ctypes.data, ln.ctypes.data, first, ds.n_reads)
        if n < 0:
            raise OSError("synth_write_fasta(%s) failed" % path)
        total += int(n)
    return total


def scratch_dir(need_bytes: int) -> str:
    for base in (os.environ.get("TLM_SCRATCH"), "/dev/shm", tempfile.gettempdir()):
        if base and os.path.isdir(base):
            try:
                if shutil.disk_usage(base).free > need_bytes * 1.2:
                    return tempfile.mkdtemp(prefix="tlm_scaffold_", dir=base)
            except OSError:
                pass
    raise OSError("no scratch directory with %.0f GB free" % (need_bytes / 1e9))


def prepare(config: str = "athaliana", ds=None, device: str = "cuda") -> dict:
    """Render `config` as text in a scratch directory.  -> {"tmp", "rr", "cr", "reads", "contigs", "fasta", "out": the first
    keys of the timing block}.  `ds`: an already generated telomeri_amd.synth.Dataset of that config."""
    from telomeri_amd import synth
    binary = os.path.join(ROOT, "telomeri_amd", "bin", "telomeri_hip")
    if not os.path.exists(binary):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "telomeri_amd", "host")], check=True)
    t0 = time.time()
    own = ds is None
    if ds is None:
        ds = synth.make_config(config, device=device)
    n_rec = len(ds.rr) + len(ds.cr)
    est = n_rec * 110 + int(ds.read_len.sum().item()) + int(ds.contig_len.sum().item()) + 32 * ds.n_reads
    tmp = scratch_dir(est)
    out = {"config": config, "records": n_rec, "reads": ds.n_reads, "contigs": ds.n_contigs,
           "program": "telomeri_amd/bin/telomeri_hip <rr.paf> <cr.paf> <reads.fasta> <contigs.fasta> <out.fasta> (reference "
                      "argv and defaults; lean graph + packed paths)"}
    prep = {"tmp": tmp, "binary": binary, "out": out}
    try:
        prep.update(rr=os.path.join(tmp, "rr_overlaps.paf"), cr=os.path.join(tmp, "cr_overlaps.paf"), reads=os.path.join(tmp, "reads.fasta"),
                    contigs=os.path.join(tmp, "contigs.fasta"), fasta=os.path.join(tmp, "HERA_result.fasta"))
        nb = write_paf(ds, ds.rr, prep["rr"]) + write_paf(ds, ds.cr, prep["cr"])
        nb += write_fasta(ds, prep["reads"], prep["contigs"])
        out["input_bytes"] = nb
        out["input_written_s"] = round(time.time() - t0, 2)
        # nothing of this process stays on the device while the program under test runs (the generator's tensors and the
        # allocator's cached blocks would otherwise sit scattered in the device memory the program allocates from)
        if own:
            del ds
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
        except ImportError:
            pass
    except BaseException:
        shutil.rmtree(tmp, ignore_errors=True)
        raise
    return prep


def execute(prep: dict, extra_args=(), ranks: int = 1, replicated: bool = False, rank_env: dict = None, timeout_s: float = None) -> dict:
    """Run telomeri_hip on a prepared input and return the timing block.
    ranks > 1: a REHEARSAL of the multi-rank job on ONE GPU -- `ranks` copies of the program share device 0 (and its PCIe
    link, and this box's CPU share) through tests/fake_rccl; the block is rank 0's, `rank_timing` holds the load lines of
    every rank.  replicated: TLM_LOAD_REPLICATED=1 (every rank uploads the whole text, round 3's load).
    rank_env: THIS call is one rank of a real multi-process job (bench.py at N > 1: one call per rank, each on its own GPU):
    the TLM_WORLD / TLM_RANK / TLM_DEVICE / TLM_RENDEZVOUS variables of that rank; rank 0 keeps the scaffold file."""
    out = dict(prep["out"])
    tmp, fasta = prep["tmp"], prep["fasta"]
    if True:
        # (the program's walk kernels take 780-790 ms in most runs and 1320-1360 ms in about one run of five, on the same
        # box, whatever this process keeps on the device and whether or not it waits here: docs/measurement_log.md)
        time.sleep(float(os.environ.get("TLM_SCAFFOLD_SETTLE_S", "0")))
        t0 = time.time()
        my_fasta = fasta
        if rank_env and int(rank_env.get("TLM_RANK", "0")) != 0:
            my_fasta = fasta + ".rank" + str(rank_env["TLM_RANK"])
        cmd = [prep["binary"]] + list(extra_args) + [prep["rr"], prep["cr"], prep["reads"], prep["contigs"], my_fasta]
        prof = os.environ.get("TLM_SCAFFOLD_ROCPROF")      # a directory: kernel trace + stats of the host program's run
        if prof:
            cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", prof, "--"] + cmd
        t_spawn = time.time()
        env = dict(os.environ, TLM_HOST_TIMING="1")
        if rank_env:
            env.update({k: str(v) for k, v in rank_env.items()})
            if replicated:
                env["TLM_LOAD_REPLICATED"] = "1"
            out["ranks"] = int(rank_env.get("TLM_WORLD", "1"))
            out["load"] = "replicated" if replicated else "sharded"
            # (output into files: the report is long, and nobody reads a pipe while the program runs)
            with open(os.path.join(tmp, "rank%s.out" % rank_env.get("TLM_RANK", "0")), "w+") as fo, \
                    open(os.path.join(tmp, "rank%s.err" % rank_env.get("TLM_RANK", "0")), "w+") as fe:
                p = subprocess.Popen(cmd, stdout=fo, stderr=fe, env=env)
                try:
                    rc = p.wait(timeout=timeout_s)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
                    rc = -9
                    out["timed_out_after_s"] = timeout_s
                fo.seek(0), fe.seek(0)
                res = subprocess.CompletedProcess(cmd, rc, fo.read(), fe.read())
            out["rank_timing"] = [ln[len("[host timing] "):] for ln in res.stderr.splitlines()
                                  if ln.startswith("[host timing]") and ("sharded load" in ln or "text upload" in ln or "tokenise" in ln)]
        elif ranks > 1:
            fake = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
            if not os.path.exists(fake):
                subprocess.run(["make", "-s", "-C", os.path.dirname(fake)], check=True)
            env.update(TLM_WORLD=str(ranks), TLM_DEVICE="0", TLM_RENDEZVOUS=os.path.join(tmp, "rendezvous"), TLM_RCCL_LIB=fake,
                       FAKE_RCCL_TIMEOUT_S="600")
            if replicated:
                env["TLM_LOAD_REPLICATED"] = "1"
            # (outputs into files: a rank whose pipe is full while nobody reads it would stop, and its peers with it)
            logs = [(open(os.path.join(tmp, "rank%d.out" % r), "w+"), open(os.path.join(tmp, "rank%d.err" % r), "w+")) for r in range(ranks)]
            procs = [subprocess.Popen(cmd[:-1] + [fasta if r == 0 else fasta + ".rank%d" % r], stdout=logs[r][0], stderr=logs[r][1],
                                      env=dict(env, TLM_RANK=str(r))) for r in range(ranks)]
            for p in procs:
                p.wait()
            outs = []
            for o, e in logs:
                o.seek(0), e.seek(0)
                outs.append((o.read(), e.read()))
                o.close(), e.close()
            res = subprocess.CompletedProcess(cmd, max(p.returncode for p in procs), outs[0][0], outs[0][1])
            out["ranks"] = ranks
            out["load"] = "replicated" if replicated else "sharded"
            out["rank_timing"] = [[ln[len("[host timing] "):] for ln in e.splitlines()
                                   if ln.startswith("[host timing]") and ("sharded load" in ln or "text upload" in ln or "tokenise" in ln)]
                                  for _, e in outs]
            out["rank_load_s"] = [float(x[0]) if x else None for x in (re.findall(r"Done \(([0-9.e+-]+)s\)", o) for o, _ in outs)]
        else:
            res = subprocess.run(cmd, capture_output=True, text=True, env=env)
        t_done = time.time()
        out["scaffold_wall_s"] = round(t_done - t0, 3)
        # the program does not wait for its upload helper (a second process that holds a device context): when THAT has
        # gone is when the device is really free again
        m = re.search(r"\[host timing\] upload helper process (\d+)", res.stderr)
        if m:
            hp = int(m.group(1))
            while os.path.exists("/proc/%d" % hp) and time.time() - t_done < 30:
                try:
                    if open("/proc/%d/stat" % hp).read().split(")")[-1].split()[0] == "Z":
                        break
                except OSError:
                    break
                time.sleep(0.002)
            out["helper_process_gone_s"] = round(time.time() - t0, 3)
        marks = {m.group(1): float(m.group(2)) for m in re.finditer(r"\[host timing\] ([a-z ]+) at ([0-9.]+)", res.stderr)}
        if "main entered" in marks:
            out["marks_s"] = {k: round(v - t_spawn, 3) for k, v in sorted(marks.items(), key=lambda kv: kv[1])}
        if "main entered" in marks and "main left" in marks:      # what the stages below do not cover
            out["outside_stages_s"] = {"exec_to_main": round(marks["main entered"] - t_spawn, 3),
                                       "main_to_loading": round(marks.get("loading starts", marks["main entered"]) - marks["main entered"], 3),
                                       "main_left_to_exit": round(t_done - marks["main left"], 3)}
        out["exit"] = res.returncode
        out["host_timing"] = [ln[len("[host timing] "):] for ln in res.stderr.splitlines() if ln.startswith("[host timing]")]
        done = [float(x) for x in re.findall(r"Done \(([0-9.e+-]+)s\)", res.stdout)]
        names = ["load_s", "path_build_s", "consensus_s", "fasta_load_s", "scaffold_write_s"]
        out["stages_s"] = {k: v for k, v in zip(names, done)}
        m = re.search(r"load breakdown: tokenise\+intern ([0-9.e+-]+) ms \((\w+)\), device graph build ([0-9.e+-]+) ms, "
                      r"download\+materialise ([0-9.e+-]+) ms", res.stdout)
        if m:
            out["load_breakdown_ms"] = {"tokenise": float(m.group(1)), "tokeniser": m.group(2), "device_graph_build": float(m.group(3)),
                                        "download_materialise": float(m.group(4))}
        m = re.search(r"edge-extension evals: (\d+) in ([0-9.e+-]+) ms of walk kernels", res.stdout)
        if m:
            out["evals"], out["walk_kernel_ms"] = int(m.group(1)), float(m.group(2))
        m = re.search(r"- total_num: (\d+)", res.stdout)
        if m:
            out["paths"] = int(m.group(1))
        if res.returncode == 0 and os.path.exists(my_fasta):    # (no file with --no-scaffold in extra_args, or on a rank other than 0)
            out["scaffold_bytes"] = os.path.getsize(my_fasta)
            chain = [ln for ln in res.stdout.splitlines() if ln.startswith("ctg") and "-" in ln]
            out["scaffold_nodes"] = chain[-1].count("-") + 1 if chain else 0
        elif res.returncode != 0 or not rank_env:
            out["stderr_tail"] = res.stderr[-400:]
    return out


def run(config: str = "athaliana", ds=None, device: str = "cuda", extra_args=(), keep: bool = False, ranks: int = 1,
        replicated: bool = False) -> dict:
    """prepare + execute + tidy: render `config` as text, run telomeri_hip on it, return the timing block."""
    prep = prepare(config, ds=ds, device=device)
    try:
        out = execute(prep, extra_args=extra_args, ranks=ranks, replicated=replicated)
        if keep:
            out["dir"] = prep["tmp"]
        return out
    finally:
        if not keep:
            shutil.rmtree(prep["tmp"], ignore_errors=True)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("config", nargs="?", default="athaliana")
    ap.add_argument("--ranks", type=int, default=1, help="rehearse a job of this many ranks on ONE GPU (tests/fake_rccl)")
    ap.add_argument("--replicated", action="store_true", help="with --ranks: every rank uploads the whole text (round 3's load)")
    ap.add_argument("--no-scaffold", action="store_true", help="stop after the path build")
    a = ap.parse_args()
    print(json.dumps(run(a.config, ranks=a.ranks, replicated=a.replicated, extra_args=("--no-scaffold",) if a.no_scaffold else ())))
