#!/usr/bin/env python3
"""bench.py -- Mreads/s of the BASAL seed-and-extend hot path on MI355X.

Workload (BASELINE.json config 2): synthetic 100 bp SE reads, -M C:T, -g 0, -S 1, on an
hg38-sized stand-in genome (tools/synth_gpu.py; hg38 itself is not on the GPU box), reference and
seed index resident in HBM.  A "step" is ONE call of basal_core_align_batch_device on one batch
of reads that is already in HBM (descriptors + bases in), plus the copy of that step's 32-byte results
to page-locked host memory on a second stream, overlapped with the next step's kernel (the timed
region ends when the last hit record is on the host);
the default batch is config 2's whole 10 M reads (a launch has a fixed cost of about 0.8 ms -- reads
from repeats take most of a millisecond and whichever starts last ends the launch -- so batches of millions
are how the path is meant to be fed; --batch 1000000 reproduces the 1 M-read launches of DESIGN.md's ladder).
Timing: W warm-up steps, then K steps between barrier + synchronize, max over ranks;
value = reads aligned by all ranks / that time.  N > 1 (torchrun): every rank holds the whole
reference + index, takes its own reads (weak scaling), and each step's per-read results are gathered
to rank 0 with one RCCL gather, issued behind the step's kernel so that it overlaps the next step
(inside the timed region; the last step's gather is exposed).

Extra objects on the JSON line:
  roofline      algorithmic bytes (4H+16S+4C+8W+L+16R per read, counters from the CPU
                oracle on a sample of the same reads) / mean kernel time from HIP events on the launch stream.
  cpu_baseline  the CPU oracle (C restatement, pthreads) timed on this box's host cores on a bounded sample
                of the same reads with the same index; the GPU results for that sample are checked against it.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def spawn_ranks(n):
    """Run this script as n ranks (one per GPU) under torch.distributed.run; returns the launcher's exit code.
    On a box with fewer than n GPUs nothing is launched: one JSON line with "skipped" says so and the exit code is 0."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n:
        print(json.dumps({"metric": "Mreads/s aligned (100 bp SE, -M C:T, hg38) at 1/2/4/8 GPUs; SAM bit-identical", "value": None, "unit": "Mreads/s",
                          "n_gpus": n, "skipped": "this box has %d GPU(s); %d ranks need one GPU each (set BASAL_DIST_BACKEND=gloo and launch under "
                                                   "torch.distributed.run to rehearse more ranks than GPUs)" % (have, n)}))
        return 0
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("launching %d ranks: %s" % (n, " ".join(cmd)))
    return subprocess.run(cmd).returncode


def time_reference(rule, gap, extra_flags, read_kw, read_len, n_reads, cores, dev, realistic=False):
    """The TRUE reference binary (oracle/_ref/basal, built from /root/reference by oracle/Makefile.ref and shipped with the snapshot)
    timed on this box's host cores.  It only reads files and rebuilds its index on every run, and an hg38-sized FASTA would take it
    many minutes, so it runs on a density-equivalent down-scaled genome: 50 Mbp with -s 12 gives the 3-letter seeds the ~40 index
    entries per seed that 3.09 Gbp gives them at -s 16.  Align time = wall clock minus
    the wall clock of the same command with -E 0 (no reads: load + index build only).  Returns a cpu_baseline dict or None."""
    import shutil
    import subprocess
    import tempfile
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "basal")
    if not os.path.exists(ref_bin) or n_reads <= 0:
        return None
    import basal_amd as B
    import synth_files
    import synth_gpu
    import torch
    d = tempfile.mkdtemp(prefix="basal_ref_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        p = B.Params(rule, ["-M", rule])
        G = synth_gpu.make_genome(p, dev, scale=50e6 / 3.088e9, seed=7, repeat_copies=650, realistic=realistic)
        fa, fq = os.path.join(d, "g.fa"), os.path.join(d, "r.fq")
        synth_files.write_fasta(fa, G)
        with open(fq, "wb") as f:
            for b0 in range(0, n_reads, 400_000):
                nb = min(400_000, n_reads - b0)
                bases, _, _, _ = synth_gpu.make_reads(G, nb, dev, read_len=read_len, seed=500 + b0, **read_kw)
                f.write(synth_files.fastq_bytes(bases.cpu().numpy().reshape(nb, read_len), np.full(nb, read_len), None, first=b0))
        del G
        torch.cuda.empty_cache()
        cmd = [ref_bin, "-a", fq, "-d", fa, "-M", rule, "-S", "1", "-s", "12", "-p", str(cores), "-o", os.path.join(d, "o.sam")] + (["-g", str(gap)] if gap else []) + list(extra_flags)

        def wall(extra):
            t = time.perf_counter()
            r = subprocess.run(cmd + extra, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            if r.returncode != 0:
                raise RuntimeError("reference binary failed")
            return time.perf_counter() - t
        t_idle = min(wall(["-E", "0"]) for _ in range(2))
        t_full = min(wall([]) for _ in range(2))
        secs = max(t_full - t_idle, 1e-3)
        return {"value": n_reads / secs / 1e6, "unit": "Mreads/s", "cores": cores, "kind": "reference",
                "sample": ("the unmodified reference binary (oracle/_ref/basal -p %d) on %d reads of the same kind on a density-equivalent down-scaled genome "
                           "(50 Mbp" + (" with the same repeat landscape" if realistic else "") + ", -s 12: ~40 index entries per seed as on 3.09 Gbp at -s 16); "
                           "align time = wall %.2f s minus %.2f s of the same command with -E 0 (load + index build); %.0f CPU-seconds of alignment")
                          % (cores, n_reads, t_full, t_idle, secs * cores)}
    except Exception as e:  # the baseline is a reported extra: never lose the bench line over it
        log("reference timing failed: %r" % (e,))
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def bench_pairs(args):
    """BASELINE.json config 3: synthetic 150 bp read pairs, -M A:G, on a transcriptome-sized stand-in (154 Mbp, 24 contigs): both mates aligned
    with every SnpAlign mode on the GPU, the pairing rounds on the GPU too (basal_core_align_pairs_batch), the records to print come back.
    A step = one batch of pairs from host buffers to host records (this entry point is host-to-host; the kernel-side time is the two kernels'
    HIP-event time).  This leg reports speed only: the path's parity is what tests/test_gpu_parity.py (6 paired-end golden SAMs, the host
    replay of the same rounds) and tests/test_gpu_cli_scale.py (200 k pairs against the reference binary) check.  One GPU."""
    import torch
    import basal_amd as B
    from basal_amd import core as bc
    import synth_gpu
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    flags = ["-M", "A:G", "-S", "1"]
    params = B.Params("A:G", flags)
    params.c.pairend = 1
    L = B.lib()
    # a transcriptome stand-in with <= 131 071 contigs (the 18-bit chr field): 100 000 sequences, ~0.17 Gbp.  Far more
    # than the 64 contigs the kernel keeps in LDS, so int2hit runs its 64-ary search of the anchor table in memory, as on a real transcriptome.
    G = synth_gpu.make_transcriptome(params, dev, n_contigs=args.contigs, seed=1) if args.contigs > 64 else synth_gpu.make_genome(params, dev, scale=0.05, seed=1, repeat_copies=2000)
    words = [w.cpu().numpy().view(np.uint64) for w in G.words]
    sizes = np.array(G.sizes, dtype=np.uint32)
    core = B.Core(params, 0)
    bc._check(L.basal_core_set_reference(core.h, words[0].ctypes.data, words[1].ctypes.data, len(words[0]), G.anchors.ctypes.data, sizes.ctypes.data,
                                         G.rc_offsets.ctypes.data, len(sizes)), "set_reference")
    mk = C.c_uint32()
    blocks = np.ascontiguousarray(G.blocks)
    bc._check(L.basal_core_build_index(core.h, blocks.ctypes.data, len(blocks), C.byref(mk)), "build_index")
    npairs, rl = min(args.batch, 2_000_000), 150
    PE_REC = np.dtype([("kind", "u1"), ("side", "u1"), ("chain_a", "u1"), ("chain_b", "u1"), ("ma", "<i4"), ("na", "<u4"), ("mb", "<i4"), ("insert", "<u4"),
                       ("ha", bc.HIT_DTYPE), ("hb", bc.HIT_DTYPE)])
    PE_PAIR = np.dtype([("first", "<u4"), ("n", "<u4"), ("status", "<u4")])
    seq = C.create_string_buffer(b"A" * rl, rl + 2)
    qual = C.create_string_buffer(b"I" * rl, rl + 2)
    ms = C.c_uint32()
    assert L.basal_host_filter_read(C.byref(params.c), seq, qual, C.byref(ms)) == 0
    batches = []
    for k in range(min(args.steps + args.warmup, 3)):
        b1, b2 = synth_gpu.make_pairs(G, npairs, dev, read_len=rl, seed=40 + k)
        # (page-locked host buffers, as a host feeding a GPU keeps them: the copies then run at the link's rate)
        hb_t = torch.empty(2 * npairs * rl, dtype=torch.uint8, pin_memory=True)
        hb_t.copy_(torch.stack([b1.view(npairs, rl), b2.view(npairs, rl)], dim=1).reshape(-1))  # a0 b0 a1 b1 ...
        hb = hb_t.numpy()
        d_t = torch.zeros(2 * npairs * bc.READ_DTYPE.itemsize, dtype=torch.uint8, pin_memory=True)
        d = d_t.numpy().view(bc.READ_DTYPE)
        d["seq_off"] = np.arange(2 * npairs, dtype=np.uint64) * rl
        d["index"] = np.repeat(np.arange(npairs, dtype=np.uint32) + k * npairs, 2)
        d["len"], d["max_snp"], d["stale_idx"] = rl, ms.value, B.STALE_NONE
        d["readset"] = np.tile(np.array([1 | B.READ_ALLMODES, 2 | B.READ_ALLMODES], np.uint8), npairs)
        batches.append((hb, d, hb_t, d_t))
    pairs_t = torch.zeros(npairs * PE_PAIR.itemsize, dtype=torch.uint8, pin_memory=True)
    recs_t = torch.zeros((2 * npairs + 4096) * PE_REC.itemsize, dtype=torch.uint8, pin_memory=True)
    pairs = pairs_t.numpy().view(PE_PAIR)
    recs = recs_t.numpy().view(PE_REC)
    core.set_timing(True)
    st = (C.c_uint32 * 9)()

    def step(i):
        hb, d = batches[i % len(batches)][:2]
        used = C.c_uint64()
        cy = np.zeros((2, 2), np.uint8)
        bc._check(L.basal_core_align_pairs_batch(core.h, hb.ctypes.data, len(hb), d.ctypes.data, npairs, None, 0, pairs.ctypes.data, recs.ctypes.data, len(recs),
                                                 C.byref(used), st, cy.ctypes.data), "align_pairs_batch")
        return used.value
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    for k in range(9):
        st[k] = 0
    kms, pms = [], []
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
        kms.append(core.kernel_ms())
        pms.append(float(L.basal_core_last_pair_ms(core.h)))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bc._check(L.basal_core_sync_check(core.h), "align kernels")  # the kernels' bounds ledger (and, in the diagnostic build, the phase clocks)
    paired = st[0] / (npairs * args.steps)
    out = {"metric": "Mpairs/s aligned and paired (150 bp PE, -M A:G, transcriptome stand-in), host buffers to host records", "value": npairs * args.steps / dt / 1e6, "unit": "Mpairs/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u64", "data": "synthetic",
           "config": {"workload": "config 3: %d k synthetic 150 bp read pairs per step, -M A:G -S 1, transcriptome stand-in (%d contigs, %.0f Mbp), mates aligned with every mode + "
                                  "paired on the GPU; host buffers in, records to print out (one synchronous batch at a time)" % (npairs // 1000, len(sizes), float(sizes.sum()) / 1e6),
                      "pairs_per_step": npairs, "mpairs_per_s_host_to_host": npairs * args.steps / dt / 1e6,
                      "mpairs_per_s_kernels": npairs / ((np.mean(kms) + np.mean(pms)) * 1e-3) / 1e6, "align_kernel_ms": float(np.mean(kms)), "pair_kernel_ms": float(np.mean(pms)),
                      "paired_frac": paired},
           "roofline": {"bound": "hbm", "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None, "kernel": "align_kernel<8,false,false,false,PE> + pair_kernel",
                        "kernel_ms": float(np.mean(kms) + np.mean(pms))},
           "cpu_baseline": None}
    # ---- parity + algorithmic bytes on a bounded sample, through files: the product's command line and the CPU oracle's on the same FASTA
    # and FASTQ pair (the oracle restates pairs.cpp; its counters are SURVEY 8d's, summed over both mates) -- then the TRUE reference binary,
    # timed on a 5 000-contig reference of the same kind (it clears one std::set per contig and read, align.cpp:437-444: 100 000 contigs
    # would take it hours).
    if args.cpu_sample > 0:
        import hashlib
        import re
        import shutil
        import subprocess
        import tempfile
        import synth_files
        import oracle as orc
        threads = args.cpu_threads or min(16, os.cpu_count() or 1)
        ns = max(2000, min(args.cpu_sample // 20, 20000))
        d = tempfile.mkdtemp(prefix="basal_c3_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)

        def digest(path):
            h, n = hashlib.md5(), 0
            for line in open(path, "rb"):
                if not line.startswith(b"@PG"):
                    h.update(line)
                    n += not line.startswith(b"@")
            return h.hexdigest(), n

        def write_pairs(G_, n_, seed_):
            b1, b2 = synth_gpu.make_pairs(G_, n_, dev, read_len=rl, seed=seed_)
            for nm, b in (("r1.fq", b1), ("r2.fq", b2)):
                open(os.path.join(d, nm), "wb").write(synth_files.fastq_bytes(b.cpu().numpy().reshape(n_, rl), np.full(n_, rl), None, name_prefix=b"p"))
        try:
            synth_files.write_fasta(os.path.join(d, "g.fa"), G)
            write_pairs(G, ns, 77)
            fl = ["-M", "A:G", "-S", "1", "-x", "700"]
            r1 = subprocess.run([os.path.join(ROOT, "basal_amd", "bin", "basal"), "-a", "r1.fq", "-b", "r2.fq", "-d", "g.fa"] + fl + ["-p", str(threads), "-o", "gpu.sam"],
                                capture_output=True, text=True, cwd=d)
            r2 = subprocess.run([orc.CLI, "-a", "r1.fq", "-b", "r2.fq", "-d", "g.fa"] + fl + ["-p", str(threads), "-o", "cpu.sam"], capture_output=True, text=True, cwd=d)
            if r1.returncode or r2.returncode:
                raise SystemExit("bench --config 3: sample run failed: %s %s" % (r1.stderr[-300:], r2.stderr[-300:]))
            # (-p N: the oracle's worker threads write whole batches in any order; the sample is below one batch of 50 000 pairs)
            if digest(os.path.join(d, "gpu.sam")) != digest(os.path.join(d, "cpu.sam")):
                raise SystemExit("bench --config 3: the SAM of %d sampled pairs differs between the GPU path and the CPU oracle -- number withheld" % ns)
            m = re.search(r"ORACLE_COUNTERS reads (\d+) H (\d+) S (\d+) C (\d+) W (\d+) L (\d+) R (\d+)", r2.stderr)
            secs = float(re.search(r"ORACLE_ALIGN_SECONDS ([0-9.]+)", r2.stderr).group(1))
            _, H_, S_, C_, W_, L_, R_ = (int(x) for x in m.groups())
            Bpair = (4 * H_ + 16 * S_ + 4 * C_ + 8 * W_ + L_ + 16 * R_) / ns
            kms_all = float(np.mean(kms) + np.mean(pms))
            ach = Bpair * npairs / (kms_all * 1e-3) / 1e9
            out["config"]["algorithmic_bytes_per_pair"] = round(Bpair, 1)
            out["config"]["counters_per_pair"] = {"H": H_ / ns, "S": S_ / ns, "C": C_ / ns, "W": W_ / ns, "L": L_ / ns, "R": R_ / ns}
            out["roofline"].update({"achieved": ach, "frac": ach / 8000.0, "bytes_per_launch": Bpair * npairs})
            port = {"value": ns / secs / 1e6, "unit": "Mpairs/s", "cores": threads, "kind": "port",
                    "sample": "%d pairs of the timed kind on the same reference, through files; the command line's SAM identical to the oracle's on all of them" % ns}
            out["cpu_baseline"] = port
            ref_bin = os.path.join(ROOT, "oracle", "_ref", "basal")
            if os.path.exists(ref_bin) and args.ref_sample > 0:
                G5 = synth_gpu.make_transcriptome(params, dev, n_contigs=5000, seed=1)
                synth_files.write_fasta(os.path.join(d, "g.fa"), G5)
                nref = 100_000
                write_pairs(G5, nref, 78)
                cmd = [ref_bin, "-a", "r1.fq", "-b", "r2.fq", "-d", "g.fa"] + fl + ["-p", str(threads), "-o", "ref.sam"]

                def wall(extra):
                    t = time.perf_counter()
                    if subprocess.run(cmd + extra, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d).returncode != 0:
                        raise RuntimeError("reference binary failed")
                    return time.perf_counter() - t
                t_idle = min(wall(["-E", "0"]) for _ in range(2))
                t_full = wall([])
                out["config"]["cpu_port"] = port
                out["cpu_baseline"] = {"value": nref / max(t_full - t_idle, 1e-3) / 1e6, "unit": "Mpairs/s", "cores": threads, "kind": "reference",
                                       "sample": "the unmodified reference binary (oracle/_ref/basal -p %d) on %d pairs of the same kind on a 5 000-contig transcriptome stand-in "
                                                 "(it clears one std::set per contig and read: 100 000 contigs are out of its reach); align time = wall %.2f s minus %.2f s of "
                                                 "the same command with -E 0" % (threads, nref, t_full, t_idle)}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):
        t = json.load(open(tj)).get("other_workloads", {}).get("config3")
        if t:
            out["roofline"]["traffic"] = t["hbm_bytes_per_pair"] * npairs
            out["roofline"]["traffic_source"] = t["source"]
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=10_000_000, help="reads per step and rank (default: config 2's 10 M reads as one batch)")
    ap.add_argument("--genome-scale", type=float, default=float(os.environ.get("BASAL_BENCH_SCALE", "1.0")), help="1.0 = hg38-sized (3.09 Gbp)")
    ap.add_argument("--cpu-sample", type=int, default=400_000, help="reads of the cpu_baseline / parity sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--ref-sample", type=int, default=1_600_000, help="reads the true reference binary is timed on (0 = skip; it is skipped when oracle/_ref/basal is missing)")
    ap.add_argument("--config", default="2", choices=["2", "3", "4", "5", "5p"],
                    help="BASELINE.json config whose reads and flags to use (SURVEY.md section 8d): 2 = C:T -g 0 (the bench line); 4 = A:CGT -g 2, each A to C/G/T "
                         "with p 0.3, 1 %% of the reads with a 1-2 base indel; 5 = T:- (30 %% of the reads with one T deleted), -g 0; 5p = the same reads with the "
                         "BID-seq pipeline flags -n 1 -g 3")
    ap.add_argument("--rule", default=None, help="ad hoc: another -M rule with config 2's kind of reads")
    ap.add_argument("--gap", type=int, default=None)
    ap.add_argument("--genome", default="realistic", choices=["uniform", "realistic"],
                    help="realistic (the bench line's stand-in for hg38): a repeat landscape shaped like hg38's (Alu / L1 / MIR / L2 / LTR / DNA families + "
                         "satellites, ~45 %% of the genome): the over-represented-k-mer cut-off then lands where seeds from repeats keep tens of thousands of "
                         "candidates, as on the real genome; uniform: random bases + one planted 300-base family (rounds 1-2's stand-in, 9 x easier; the "
                         "default run reports it as config.uniform_genome_mreads_per_s)")
    ap.add_argument("--fasta", default=os.environ.get("BASAL_HG38_FASTA"),
                    help="align against this FASTA (e.g. hg38) instead of a synthetic genome: loaded by the product's loader (basal_host_ref_load), reads are "
                         "sampled from it; default: $BASAL_HG38_FASTA if set")
    ap.add_argument("--contigs", type=int, default=100_000, help="config 3: contigs of the transcriptome stand-in (<= 64: the 24-contig genome-shaped stand-in instead)")
    ap.add_argument("--placement-draws", type=int, default=int(os.environ.get("BASAL_BENCH_PLACEMENT_DRAWS", "4")),
                    help="set-up: placements of the index in HBM tried at most (1 = take what hipMalloc gave; see DESIGN section 8)")
    ap.add_argument("--read-len", type=int, default=100, help="read length (the headline workload is 100; 150/300 exercise the 256/480-base kernels)")
    args = ap.parse_args()

    if args.config == "3":
        if args.batch == 10_000_000:
            args.batch = 1_000_000
        return bench_pairs(args)
    # `python bench.py --gpus N` launched plainly: start the N ranks ourselves (children of this process, decided before
    # anything here has touched a GPU -- device_count() does not initialise HIP), wait, and leave with their exit code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    import basal_amd as B
    from basal_amd import core as bc
    import synth_gpu

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the hot path")
    local = local % torch.cuda.device_count()  # (a gloo rehearsal may put several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = os.environ.get("BASAL_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
    # BASAL_BENCH_FORCE_DIST=1: run the N > 1 code path (process group, per-step gather) in a single process -- a rehearsal of
    # the RCCL calls on a one-GPU box
    dist_on = world > 1 or bool(os.environ.get("BASAL_BENCH_FORCE_DIST"))
    if dist_on and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if dist_on:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    presets = {"2": ("C:T", 0, []), "4": ("A:CGT", 2, []), "5": ("T:-", 0, []), "5p": ("T:-", 3, ["-n", "1"])}
    p_rule, p_gap, p_extra = presets[args.config]
    adhoc = args.rule is not None or args.gap is not None
    args.rule = args.rule or p_rule
    args.gap = p_gap if args.gap is None else args.gap
    flags = ["-M", args.rule, "-S", "1"] + (["-g", str(args.gap)] if args.gap else []) + p_extra
    params = B.Params(args.rule, flags)
    L = B.lib()

    # ---- reference + index, staged once -----------------------------------------------------
    t0 = time.time()
    if args.fasta:
        G = synth_gpu.genome_from_reference(params, B.Reference(params, fasta_path=args.fasta), dev)
        args.genome = "fasta"
    else:
        G = synth_gpu.make_genome(params, dev, scale=args.genome_scale, seed=1, repeat_copies=int(os.environ.get("BASAL_BENCH_REPEAT_COPIES", "40000")),
                                  realistic=args.genome == "realistic")
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    words = [w.cpu().numpy().view(np.uint64) for w in G.words]
    sizes = np.array(G.sizes, dtype=np.uint32)
    core = B.Core(params, local)
    t0 = time.time()
    bc._check(L.basal_core_set_reference(core.h, words[0].ctypes.data, words[1].ctypes.data, len(words[0]), G.anchors.ctypes.data,
                                         sizes.ctypes.data, G.rc_offsets.ctypes.data, len(sizes)), "set_reference")
    mk = C.c_uint32()
    blocks = np.ascontiguousarray(G.blocks)
    bc._check(L.basal_core_build_index(core.h, blocks.ctypes.data, len(blocks), C.byref(mk)), "build_index")
    t_index = time.time() - t0
    total_bp = int(sizes.sum())
    log(("genome %.3f Gbp in %d contigs " + ("loaded from " + args.fasta if args.fasta else "generated") + " in %.1f s; reference staged + seed index built on the GPU in %.1f s (cut-off %d)")
        % (total_bp / 1e9, len(sizes), t_gen, t_index, mk.value))

    # ---- reads resident in HBM -----------------------------------------------------------------
    n_steps = args.steps + args.warmup
    # Every step has its own reads, up to a pool of 8 batches (step i uses batch i % 8): a long run (--steps 100) then
    # cycles through 80 M distinct reads instead of allocating 150 GB; 1 GB of reads per batch leaves nothing in any cache.
    n_pool = min(n_steps, 8)
    n_reads = args.batch * n_pool
    read_len = args.read_len
    frm, to = "ACGT".index(args.rule[0].upper()), "ACGT".index(args.rule[2].upper()) if args.rule[2] in "ACGTacgt" else None
    read_kw = dict(conv_from=frm, conv_to=to if to is not None else frm, p_conv=0.95 if to is not None else 0.0)
    if not adhoc and args.config == "4":
        read_kw = dict(conv_from=0, conv_to=[1, 2, 3], p_conv=0.3, indel_frac=0.01, indel_max=2)
    elif not adhoc and args.config in ("5", "5p"):
        read_kw = dict(conv_from=3, conv_to=3, p_conv=0.0, del_base=3, del_frac=0.3, del_lo=20, del_hi=80)
    chunks = []
    per = 2_000_000
    for c0 in range(0, n_reads, per):
        nb = min(per, n_reads - c0)
        b, _, _, _ = synth_gpu.make_reads(G, nb, dev, read_len=read_len, seed=1000 * (rank + 1) + c0 // per, **read_kw)
        chunks.append(b)
    d_bases = torch.cat(chunks) if len(chunks) > 1 else chunks[0]
    del chunks
    # read_max_snp_num as FilterReads computes it for this length (product host code)
    seq = C.create_string_buffer(b"A" * read_len, read_len + 2)
    qual = C.create_string_buffer(b"I" * read_len, read_len + 2)
    ms = C.c_uint32()
    assert L.basal_host_filter_read(C.byref(params.c), seq, qual, C.byref(ms)) == 0
    descs = np.zeros(n_reads, bc.READ_DTYPE)
    if args.batch * read_len >= 2 ** 32:
        raise SystemExit("bench: --batch x --read-len must stay below 4 GiB (32-bit byte offsets within one batch)")
    descs["seq_off"] = (np.arange(n_reads, dtype=np.uint64) % args.batch) * read_len  # relative to the step's own byte buffer
    descs["index"] = np.arange(n_reads, dtype=np.uint32) + rank * n_reads  # global read numbers (myrand)
    descs["len"] = read_len
    descs["max_snp"] = ms.value
    descs["stale_idx"] = B.STALE_NONE
    d_reads = torch.from_numpy(descs.view(np.uint8).reshape(-1)).to(dev)
    d_results = torch.zeros(n_reads * 32, dtype=torch.uint8, device=dev)
    d_used = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    core.set_timing(True)
    # every step's records go to page-locked host memory (one buffer per pool slot), copied on a second stream behind the kernel
    h_results = torch.empty(n_reads * 32, dtype=torch.uint8, pin_memory=True)
    # (a stream of another priority: HIP multiplexes streams onto a few hardware queues, and two streams that land on the same queue
    # run in submission order -- the copies would then wait for kernels queued before them instead of overlapping them)
    copy_stream = torch.cuda.Stream(dev, priority=-1)
    copied = {}  # step -> event behind its copy (a pool slot is not rewritten before its last copy has left)

    def step(i):
        j = i % n_pool
        if i - n_pool in copied:
            torch.cuda.current_stream(dev).wait_event(copied.pop(i - n_pool))
        rc = L.basal_core_align_batch_device(core.h, d_bases.data_ptr() + j * args.batch * read_len, d_reads.data_ptr() + j * args.batch * 16, args.batch, None, 0,
                                             B.STREAM_NONE, d_results.data_ptr() + j * args.batch * 32, None, 0, d_used.data_ptr(), None,
                                             read_len, stream)
        bc._check(rc, "align_batch_device")
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(ev)
            h_results[j * args.batch * 32:(j + 1) * args.batch * 32].copy_(d_results[j * args.batch * 32:(j + 1) * args.batch * 32], non_blocking=True)
            copied[i] = torch.cuda.Event()
            copied[i].record(copy_stream)

    def sync_all():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- placement draws (set-up, not timed) ----------------------------------------------------
    # On an index that keeps long lists the kernel's time depends on where hipMalloc put the core's buffers in HBM: steady for a placement, up to 10 %
    # apart between placements (tools/probe_placement.py). The library can copy its buffers into fresh memory and keep the originals
    # aside (basal_core_placement_fork), go back (_swap) and free the set not wanted (_commit): a launch of batch 0 is timed on the placement hipMalloc
    # gave and on up to --placement-draws - 1 further ones, each time keeping the faster of the two sets. The steps run on the best placement seen.
    placement_ms = []
    if world > torch.cuda.device_count():
        args.placement_draws = 1  # (a rehearsal with several ranks on one GPU: no room for second sets, and another rank's index build may be under way)
    if args.placement_draws > 1 and mk.value >= 32768:
        def calibrate():  # mean launch time over the pool's batches (each has its own slice of the read and record buffers), after one launch not counted
            total = 0.0
            for k, j in enumerate([0] + list(range(n_pool))):
                bc._check(L.basal_core_align_batch_device(core.h, d_bases.data_ptr() + j * args.batch * read_len, d_reads.data_ptr() + j * args.batch * 16, args.batch, None, 0,
                                                          B.STREAM_NONE, d_results.data_ptr() + j * args.batch * 32, None, 0, d_used.data_ptr(), None, read_len, stream),
                          "align_batch_device")
                torch.cuda.synchronize()
                if k:
                    total += core.kernel_ms()
            return total / n_pool

        t_draws = time.time()
        best = calibrate()
        placement_ms.append(round(best, 3))
        for d in range(1, args.placement_draws):
            if d >= 2 and time.time() - t_draws > 10.0:
                break  # (the slow GAP workloads: two placements are what ten seconds of set-up buy)
            n_copied = L.basal_core_placement_fork(core.h)
            if n_copied < 0:
                raise SystemExit("bench: basal_core_placement_fork: " + L.basal_last_error().decode())
            if n_copied == 0:
                break  # (no room in HBM for a second set)
            kept = (d_bases, d_reads, d_results)  # the batch's own buffers are part of the placement: second copies of them too
            try:
                d_bases, d_reads, d_results = d_bases.clone(), d_reads.clone(), d_results.clone()
            except RuntimeError:  # (no room: the index's second set alone, then)
                d_bases, d_reads, d_results = kept
                torch.cuda.empty_cache()
            t = calibrate()
            placement_ms.append(round(t, 3))
            if t < best:
                best = t
            else:
                bc._check(L.basal_core_placement_swap(core.h), "placement_swap")  # back onto the faster set
                d_bases, d_reads, d_results = kept
            del kept
            bc._check(L.basal_core_placement_commit(core.h), "placement_commit")
        log("placement draws (mean launch of %d reads over the pool's %d batches, ms): %s -> %.3f" % (args.batch, n_pool, placement_ms, best))

    for i in range(args.warmup):
        step(i)
    sync_all()
    kernel_ms = []
    gdev = dev if backend == "nccl" else torch.device("cpu")
    step_bytes = args.batch * 32
    # rank 0 receives every rank's per-read records, step by step (what it would hand to the SAM writer)
    gathered = [[torch.empty(step_bytes, dtype=torch.uint8, device=gdev) for _ in range(world)] for _ in range(args.steps)] if (dist_on and rank == 0) else None
    works = []
    t0 = time.perf_counter()
    for i in range(args.warmup, n_steps):
        step(i)
        if not os.environ.get("BASAL_BENCH_NOSYNC"):
            kernel_ms.append(core.kernel_ms())  # waits for this step's stop event (HIP events on the launch stream)
        if dist_on:
            # The one collective of the path: this step's results to the rank that writes SAM. RCCL runs it on its own
            # stream behind this step's kernel (it orders itself after the work already queued on the current stream), so the
            # transfer over xGMI overlaps the next step's kernel; only the last step's gather is exposed.
            sl = d_results[(i % n_pool) * step_bytes:((i % n_pool) + 1) * step_bytes]
            if backend == "nccl":
                works.append(dist.gather(sl, gathered[i - args.warmup] if rank == 0 else None, dst=0, async_op=True))
            else:
                dist.gather(sl.cpu(), gathered[i - args.warmup] if rank == 0 else None, dst=0)
    bc._check(L.basal_core_sync_check(core.h), "align kernels")
    for w in works:
        w.wait()
    sync_all()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    reads_timed = args.batch * args.steps * world
    if dist_on and rank == 0:  # every rank's records arrived: count them (uint8 view of basal_result.best_level, byte 20 of 32)
        gathered_aligned = 0
        for per_step in gathered:
            for g in per_step:
                gathered_aligned += int((g.view(-1, 32)[:, 20] != 0xFF).sum().item())
        assert sum(g.numel() for per_step in gathered for g in per_step) == reads_timed * 32
    else:
        gathered_aligned = None

    # counts over all timed reads on the device (basal_result: n_hit u16 at byte 16, n_chit u16 at 18, best_level u8 at 20);
    # the first timed step's records come to the host for the oracle sample and the host-buffer cross-check
    first_slot = args.warmup % n_pool  # the first timed step's batch
    rv = (d_results if n_steps > n_pool else d_results[args.warmup * step_bytes:]).view(-1, 32)
    has = rv[:, 20] != 0xFF
    nh = rv[:, 16].to(torch.int32) + (rv[:, 17].to(torch.int32) << 8) + rv[:, 18].to(torch.int32) + (rv[:, 19].to(torch.int32) << 8)
    aligned = int(has.sum().item())
    unique = int((has & (nh == 1)).sum().item())
    n_timed = int(rv.shape[0])
    timed = np.frombuffer(d_results[first_slot * step_bytes:(first_slot + 1) * step_bytes].cpu().numpy().tobytes(), dtype=bc.RESULT_DTYPE)
    if n_steps <= n_pool:  # what arrived on the host is what the device holds (slots not overwritten by later steps)
        assert h_results[first_slot * step_bytes:(first_slot + 1) * step_bytes].numpy().tobytes() == timed.tobytes(), "host copy of the results differs"
    del rv, has, nh
    # aligned reads of the TIMED steps on every rank (the pool's slots hold the last step that used them: with more steps than slots a slot
    # was timed several times with the same reads and the same results), summed over the ranks: what rank 0's gather must have delivered
    aligned_timed = 0
    for i in range(args.warmup, n_steps):
        j = i % n_pool
        aligned_timed += int((d_results[j * step_bytes:(j + 1) * step_bytes].view(-1, 32)[:, 20] != 0xFF).sum().item())
    if dist_on:
        t = torch.tensor([aligned_timed], dtype=torch.int64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        aligned_timed = int(t.item())
    blocks_, threads_, lds_ = core.launch_info()

    out = {
        "metric": "Mreads/s aligned (100 bp SE, -M C:T, hg38) at 1/2/4/8 GPUs; SAM bit-identical",
        "value": reads_timed / dt / 1e6, "unit": "Mreads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": ("config %s: %d steps x %d synthetic %d bp SE reads per GPU (a pool of %d distinct batches, step i takes batch i mod %d), -M %s -g %d -S 1%s, " + ("the genome in " + os.path.basename(args.fasta) + " " if args.fasta else "hg38-sized synthetic genome ") +
                                "(%.2f Gbp, %d contigs, N gaps, " + ("hg38-like repeat landscape: ~45 %% repeats" if args.genome == "realistic" else "as loaded" if args.fasta else "uniform bases + one planted repeat family") + "), reference + seed index resident in HBM, reads resident in HBM, every step's hit "
                                "records copied to page-locked host memory inside the timed region")
                               % (args.config, args.steps, args.batch, args.read_len, n_pool, n_pool, args.rule, args.gap, (" " + " ".join(p_extra)) if p_extra else "",
                                  total_bp / 1e9, len(sizes)),
                   "reads_per_step_per_gpu": args.batch, "genome_bp": total_bp, "index_entries": None, "aligned_frac": aligned / max(1, n_timed),
                   "unique_frac": unique / max(1, n_timed), "gathered_aligned_reads": gathered_aligned, "aligned_reads_all_ranks": aligned_timed, "kernel_grid": [blocks_, threads_], "lds_bytes_per_block": lds_,
                   "index_build_s": round(t_index, 2)},
    }
    if placement_ms:
        out["config"]["placement_draws_ms"] = placement_ms
        out["config"]["placement_note"] = ("set-up, not timed: the index was placed in HBM %d time(s) (basal_core_placement_fork / _swap / _commit), the pool's batches of %d reads launched once each time; "
                                           "the steps ran on the fastest of them (DESIGN section 8: the kernel's time follows the placement)" % (len(placement_ms), args.batch))

    headline = args.config == "2" and not adhoc and args.read_len == 100 and args.genome_scale == 1.0 and args.genome == "realistic"
    headline_alg = args.config == "2" and not adhoc and args.read_len == 100 and args.genome_scale == 1.0 and args.genome in ("realistic", "uniform")
    kernel_name = "align_kernel<%d,%s,%s,%s>" % (4 if read_len <= 128 else 8 if read_len <= 256 else 16, "true" if params.c.new_rule else "false",
                                                 "true" if args.gap > 0 else "false", "HEAVY" if ((mk.value >= 32768 or os.environ.get("BASAL_HEAVY") == "1") and os.environ.get("BASAL_HEAVY", "1") != "0") else "false")
    # ---- cpu_baseline + parity on a bounded sample (rank 0, N=1 only) + roofline ------------------
    cpu = None
    roof = {"bound": "hbm", "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None}
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        import oracle_bridge
        ns = min(args.cpu_sample, args.batch)  # the head of the first timed step
        ob = oracle_bridge.OracleOnIndex(core, params, flags, G.names, sizes, words)
        out["config"]["index_entries"] = int(len(ob.locs))
        first = first_slot * args.batch
        sb = d_bases[first * read_len:(first + ns) * read_len].cpu().numpy()
        sd = descs[first:first + ns]
        threads = args.cpu_threads or min(16, os.cpu_count() or 1)
        best, cnt, secs_v = ob.align(sb, sd["seq_off"], sd["len"], sd["index"], sd["max_snp"], threads)
        bad = oracle_bridge.differing(timed[:ns], best)
        if len(bad):
            for i in bad[:8]:
                log("differs: read %d gpu %r oracle %r" % (i, timed[i], best[i]))
            raise SystemExit("bench: %d of %d sampled reads differ between the GPU path and the CPU oracle -- number withheld" % (len(bad), ns))

        class _S:
            value = secs_v
        secs = _S()
        Bread = (4 * cnt.hdr_lookups + 16 * cnt.seed_lookups + 4 * cnt.candidates + 8 * cnt.ref_words + cnt.read_bytes + 16 * cnt.hit_records) / ns
        cpu = {"value": ns / secs.value / 1e6, "unit": "Mreads/s", "cores": threads, "kind": "port",
               "sample": "%d reads of the timed workload, same index; GPU results identical to the oracle on all of them" % ns}
        out["config"]["algorithmic_bytes_per_read"] = round(Bread, 1)
        out["config"]["counters_per_read"] = {"H": cnt.hdr_lookups / ns, "S": cnt.seed_lookups / ns, "C": cnt.candidates / ns, "W": cnt.ref_words / ns,
                                              "L": cnt.read_bytes / ns, "R": cnt.hit_records / ns}
        mean_ms = float(np.mean(kernel_ms))
        ach = Bread * args.batch / (mean_ms * 1e-3) / 1e9
        roof.update({"achieved": ach, "frac": ach / 8000.0, "kernel": kernel_name, "kernel_ms": mean_ms,
                     "bytes_per_launch": Bread * args.batch})
    else:
        roof["kernel_ms"] = float(np.mean(kernel_ms)) if kernel_ms else None
        roof["kernel"] = kernel_name
        # no CPU sample in this run (N > 1, or --cpu-sample 0): the algorithmic bytes per read of the headline workload are a
        # property of the workload, measured by the oracle's counters in the N = 1 run and committed with the profiles
        aj = os.path.join(ROOT, "profiles", "algorithmic.json")
        if rank == 0 and headline_alg and os.path.exists(aj) and roof["kernel_ms"]:
            a = json.load(open(aj))[args.genome]  # (keyed by stand-in genome: the realistic landscape costs 7.5 x the uniform one's bytes per read)
            ach = a["bytes_per_read"] * args.batch / (roof["kernel_ms"] * 1e-3) / 1e9
            roof.update({"achieved": ach, "frac": ach / 8000.0, "bytes_per_launch": a["bytes_per_read"] * args.batch,
                         "bytes_source": a["source"] + "; kernel_ms = rank 0's launches"})
    # HBM traffic per read from the PMC passes of this same command (profiles/traffic.json, if committed)
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):  # a per-launch, per-GPU figure; the committed passes cover the headline workload, configs 4 and 5p and the realistic genome
        t = json.load(open(tj))
        plain = args.read_len == 100 and not adhoc and args.genome_scale == 1.0
        key = None if headline else ({"2": "uniform", "4": "config4", "5": "config5", "5p": "config5p"}.get(args.config) if (plain and args.genome == "uniform") else
                                     {"4": "config4_realistic", "5p": "config5p_realistic", "5": "config5_realistic"}.get(args.config, "-") if (plain and args.genome == "realistic") else "-")
        if args.read_len == 150 and args.genome_scale == 1.0:  # the 256-base kernels' committed passes
            if not adhoc and args.config == "2" and args.genome == "realistic":
                key = "150bp_realistic"
            elif args.config == "2" and args.rule == "C:T" and args.gap == 2 and args.genome == "uniform":
                key = "150bp_g2_uniform"
        t = t if key is None else t.get("other_workloads", {}).get(key)
        if t:
            roof["traffic"] = t["hbm_bytes_per_read"] * args.batch
            roof["traffic_source"] = t["source"]
            # the other roof of this integer kernel: vector-instruction issue (a wave64 instruction occupies its 16-lane SIMD for 4 cycles;
            # 1 024 SIMDs at 2.4 GHz). Reported beside the HBM roofline, from the same counter passes; informational.
            if t.get("valu_per_read") and roof.get("kernel_ms"):
                roof["vector_issue"] = {"valu_per_read": t["valu_per_read"],
                                        "busy_frac": t["valu_per_read"] * 4.0 * args.batch / (1024 * 2.4e9 * roof["kernel_ms"] * 1e-3),
                                        "note": "SQ_INSTS_VALU per read x 4 cycles / (1024 SIMDs x 2.4 GHz x kernel time)"}
    # Host-to-host rate: prepared reads in page-locked host buffers -> H2D -> kernel -> D2H of the hit records, three batches in
    # flight on their own streams (basal_pipe_*, BASAL_PIPE_OUT_RESULTS). Reported next to value, never as value.
    if rank == 0 and world == 1 and not os.environ.get("BASAL_BENCH_NO_H2H"):
        torch.cuda.synchronize()
        hb = d_bases[: args.batch * read_len].cpu().numpy()
        hd = descs[: args.batch].copy()
        # (one kernel stream here: the pipe would alternate consecutive batches' kernels between two streams so that one launch's tail overlaps the next
        # launch -- DESIGN 4.3 -- and overlapping launches would make this process's per-launch durations, which the roofline is priced with and
        # rocprofv3 --kernel-trace averages, something other than the time one launch takes)
        os.environ["BASAL_PIPE_ONE_STREAM"] = "1"
        pipe = B.Pipe(core, depth=3, max_reads=args.batch, max_bytes=args.batch * read_len + 4096, output=B.PIPE_OUT_RESULTS)
        for _ in range(3):  # fill the three slots' page-locked buffers once (the caller's reads; this also faults their pages in), untimed
            bl, rw = pipe.acquire()
            bl[: len(hb)] = hb
            rw[: hd.nbytes] = hd.view(np.uint8).reshape(-1)
            bc._check(L.basal_pipe_submit_prepared(pipe.h, len(hb), len(hd), read_len), "pipe_submit_prepared")
        for _ in range(3):
            rc, data, _st = pipe.collect(copy=False)
            assert rc == 0, data
            pipe.release()
        nb_h2h, inflight, submitted, collected = 6, 0, 0, 0
        got = None
        t1 = time.perf_counter()
        while collected < nb_h2h:
            while inflight < 3 and submitted < nb_h2h:
                pipe.acquire()
                bc._check(L.basal_pipe_submit_prepared(pipe.h, len(hb), len(hd), read_len), "pipe_submit_prepared")
                inflight += 1
                submitted += 1
            rc, data, _st = pipe.collect(copy=False)
            assert rc == 0, data
            if got is None:
                got = C.string_at(data[0], 32 * min(args.batch, 100000))
            pipe.release()
            inflight -= 1
            collected += 1
        dt_h2h = time.perf_counter() - t1
        out["config"]["host_to_host_mreads_per_s"] = round(nb_h2h * args.batch / dt_h2h / 1e6, 2)
        out["config"]["host_to_host_note"] = ("prepared reads (bases + 16-byte descriptors, %d B/read) in page-locked host buffers -> HBM -> kernel -> %d B/read of hit records in "
                                              "page-locked host memory; %d batches of %d reads, 3 in flight (basal_pipe_*)" % (read_len + 16, 32, nb_h2h, args.batch))
        assert got == d_results[: len(got)].cpu().numpy().tobytes(), "host-to-host results differ from the resident run"
        pipe.close()
        del pipe
    out["roofline"] = roof
    if rank == 0 and world == 1 and cpu is not None and args.ref_sample > 0:
        threads = args.cpu_threads or min(16, os.cpu_count() or 1)
        ref_cpu = time_reference(args.rule, args.gap, p_extra, read_kw, read_len, (args.ref_sample if args.gap == 0 else args.ref_sample // 4) // (8 if args.genome == "realistic" else 1),
                                 threads, dev, realistic=args.genome == "realistic")
        if ref_cpu is not None:  # the reference itself is the baseline; the port on the exact workload stays next to it
            out["config"]["cpu_port"] = cpu
            cpu = ref_cpu
            # the like-for-like figure -- the same binary on the bench genome itself at -s 16 -- takes a quarter of an hour (tools/ref_like_for_like.py)
            # and is a committed measurement, cited here
            lj = os.path.join(ROOT, "profiles", "r04_ref_like_for_like.json")
            if headline and os.path.exists(lj):
                try:
                    l4l = json.load(open(lj))
                    out["config"]["cpu_like_for_like"] = {"value": l4l["reference_mreads_per_s"], "unit": "Mreads/s", "cores": l4l["threads"], "kind": "reference",
                                                          "sample": l4l["workload"] + "; align time %.2f s = wall %.1f s minus %.1f s of load + index build; SAM identical to the GPU command line's on all %d reads "
                                                                    "(profiles/r04_ref_like_for_like.json)" % (l4l["reference_align_s"], l4l["reference_full_s"], l4l["reference_load_and_index_s"], l4l["sam_records"])}
                    cpu["sample"] += "; like for like (the bench genome itself, 3.09 Gbp at -s 16, tools/ref_like_for_like.py): %.4f Mreads/s on %d cores (config.cpu_like_for_like)" % (
                        l4l["reference_mreads_per_s"], l4l["threads"])
                except Exception as e:
                    log("like-for-like record unreadable: %r" % (e,))
    out["cpu_baseline"] = cpu
    # The default line also carries rounds 1-2's headline workload, the uniform stand-in genome, measured by the same script in a child process
    # (its own genome, index and oracle-checked sample; this process's core stays resident meanwhile: 288 GB hold both).
    if rank == 0 and world == 1 and headline and cpu is not None and not os.environ.get("BASAL_BENCH_NO_UNIFORM"):
        import subprocess
        env = dict(os.environ, BASAL_BENCH_NO_H2H="1", BASAL_BENCH_NO_UNIFORM="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--genome", "uniform", "--steps", "5", "--warmup", "1", "--cpu-sample", "200000", "--ref-sample", "0"],
                           capture_output=True, text=True, env=env)
        try:
            u = json.loads(r.stdout.strip().splitlines()[-1])
            out["config"]["uniform_genome_mreads_per_s"] = round(u["value"], 1)
            out["config"]["uniform_genome_roofline_frac"] = round(u["roofline"]["frac"], 3)
            out["config"]["uniform_genome_note"] = "config 2 on the uniform stand-in genome (rounds 1-2's bench line): 5 steps of 10 M reads, 200 000-read sample identical to the oracle"
        except Exception as e:
            log("uniform-genome leg failed: %r %s" % (e, r.stderr[-300:]))
    if rank == 0:
        print(json.dumps(out))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
