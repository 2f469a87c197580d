#!/usr/bin/env python3
"""Regenerate tests/golden/: seeded synthetic inputs + golden SAMs from the REAL reference.

Runs only in the build container (needs /root/reference to build oracle/_ref/basal through
oracle/Makefile.ref).  For every fixture in FIXTURES it
  1. generates <name>.fa.gz / <name>.fq.gz (and <name>_2.fq.gz for PE) with tools/gen_synth.py,
  2. runs the unmodified reference binary with `-p 1 -S <seed>` on them,
  3. stores the SAM (minus the @PG line, which embeds argv) as <name>.sam.gz,
and writes manifest.json with the exact flags.  Only data (inputs and expected outputs) is
committed; no reference source or binary enters the repo.
"""
import gzip
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "basal")
GEN = os.path.join(ROOT, "tools", "gen_synth.py")

# name -> (gen_synth args, basal flags).  -s 12 keeps the 3^k tables small so the CPU test
# suite stays fast; c1_s16 and acgt_s16 exercise the default seed size.
FIXTURES = {
    # config 1 of BASELINE.json, verbatim shape (1 k x 100 bp, 1 Mb, C:T)
    "c1_s16": (["--ref-bp", "1000000", "--contigs", "2", "--reads", "1000", "-M", "C:T"],
               ["-M", "C:T", "-S", "1"]),
    "ct_basic": (["--ref-bp", "300000", "--contigs", "3", "--reads", "500", "-M", "C:T", "--max-sub", "6"],
                 ["-M", "C:T", "-S", "1", "-s", "12"]),
    # both chains, N's in reads and reference, junk reads, lower-case reference, -u -R
    "ct_n1_dirty": (["--ref-bp", "300000", "--contigs", "3", "--reads", "500", "-M", "C:T", "--n-frac", "0.3",
                     "--junk-frac", "0.1", "--n-run-every", "20011", "--n-run-len", "37", "--lower-frac", "0.2",
                     "--max-sub", "8"],
                    ["-M", "C:T", "-S", "7", "-s", "12", "-n", "1", "-u", "-R"]),
    "ct_pbat": (["--ref-bp", "200000", "--contigs", "2", "--reads", "300", "-M", "C:T", "--pbat"],
                ["-M", "C:T", "-S", "1", "-s", "12", "-n", "2"]),
    "ag_se": (["--ref-bp", "300000", "--contigs", "2", "--reads", "500", "-M", "A:G", "--p-conv", "0.9"],
              ["-M", "A:G", "-S", "1", "-s", "12"]),
    # multi-way rule + gaps (config 4 family)
    "acgt_g2": (["--ref-bp", "300000", "--contigs", "2", "--reads", "500", "-M", "A:CGT", "--p-conv", "0.3",
                 "--indel-frac", "0.3", "--indel-max", "2"],
                ["-M", "A:CGT", "-S", "1", "-s", "12", "-g", "2"]),
    "acgt_s16": (["--ref-bp", "400000", "--contigs", "2", "--reads", "300", "-M", "A:CGT", "--p-conv", "0.3",
                  "--indel-frac", "0.2", "--indel-max", "2"],
                 ["-M", "A:CGT", "-S", "1", "-g", "2", "-n", "1"]),
    # deletion-induced (config 5 family): plain, and the BID-seq pipeline flags
    "tdel_g0": (["--ref-bp", "300000", "--contigs", "2", "--reads", "500", "-M", "T:-", "--p-conv", "0.02"],
                ["-M", "T:-", "-S", "1", "-s", "12"]),
    "tdel_pipeline": (["--ref-bp", "300000", "--contigs", "2", "--reads", "500", "-M", "T:-", "--p-conv", "0.02",
                       "--indel-frac", "0.2", "--indel-max", "3"],
                      ["-M", "T:-", "-S", "1", "-s", "12", "-n", "1", "-g", "3", "-R", "-u"]),
    "gact_del": (["--ref-bp", "300000", "--contigs", "2", "--reads", "400", "-M", "G:ACT-", "--p-conv", "0.2"],
                 ["-M", "G:ACT-", "-S", "3", "-s", "12", "-g", "1", "-n", "1"]),
    # C:T with gaps on both strands
    "ct_g3": (["--ref-bp", "300000", "--contigs", "2", "--reads", "500", "-M", "C:T", "--indel-frac", "0.5",
               "--indel-max", "3"],
              ["-M", "C:T", "-S", "1", "-s", "12", "-g", "3", "-n", "1"]),
    # repeats: 150-copy family; caps, threshold tightening, -r 0/1/2
    "rep_r1": (["--ref-bp", "400000", "--contigs", "2", "--reads", "300", "-M", "C:T", "--repeat-copies", "150",
                "--repeat-len", "400", "--max-sub", "4"],
               ["-M", "C:T", "-S", "1", "-s", "12", "-k", "1e-3"]),
    "rep_r2_w10": (["--ref-bp", "400000", "--contigs", "2", "--reads", "200", "-M", "C:T", "--repeat-copies", "150",
                    "--repeat-len", "400", "--max-sub", "4"],
                   ["-M", "C:T", "-S", "5", "-s", "12", "-k", "1e-3", "-r", "2", "-w", "10", "-n", "1"]),
    "rep_r0_u": (["--ref-bp", "400000", "--contigs", "2", "--reads", "200", "-M", "C:T", "--repeat-copies", "150",
                  "--repeat-len", "400", "--max-sub", "4"],
                 ["-M", "C:T", "-S", "1", "-s", "12", "-k", "1e-3", "-r", "0", "-u"]),
    "rep_g2": (["--ref-bp", "400000", "--contigs", "2", "--reads", "200", "-M", "C:T", "--repeat-copies", "150",
                "--repeat-len", "400", "--max-sub", "3", "--indel-frac", "0.3"],
               ["-M", "C:T", "-S", "2", "-s", "12", "-k", "1e-3", "-g", "2", "-w", "20", "-r", "2"]),
    # variable read lengths (incl. lengths where (len-I+1)%k==0 -> stale start offset), trimming
    "varlen_trim": (["--ref-bp", "300000", "--contigs", "2", "--reads", "600", "-M", "C:T", "--len", "131",
                     "--len-jitter", "100"],
                    ["-M", "C:T", "-S", "1", "-s", "12", "-u", "-L", "120"]),
    "varlen_s16": (["--ref-bp", "300000", "--contigs", "2", "--reads", "300", "-M", "C:T", "--len", "150",
                    "--len-jitter", "60"],
                   ["-M", "C:T", "-S", "1", "-n", "1"]),
    # -v forms, -I, -s 10
    "v_abs3": (["--ref-bp", "200000", "--contigs", "2", "--reads", "300", "-M", "C:T", "--max-sub", "6"],
               ["-M", "C:T", "-S", "1", "-s", "12", "-v", "3", "-u"]),
    "v_frac05_I2": (["--ref-bp", "200000", "--contigs", "2", "--reads", "300", "-M", "C:T", "--max-sub", "6"],
                    ["-M", "C:T", "-S", "1", "-s", "10", "-I", "2", "-v", "0.05", "-u"]),
    "v0": (["--ref-bp", "200000", "--contigs", "2", "--reads", "200", "-M", "C:T", "--max-sub", "1"],
           ["-M", "C:T", "-S", "1", "-s", "12", "-v", "0", "-u"]),
    # transcriptome-like: many short contigs, 150 bp reads, A:G
    "tx_ag_150": (["--ref-bp", "600000", "--contigs", "300", "--reads", "400", "-M", "A:G", "--p-conv", "0.9",
                   "--len", "150"],
                  ["-M", "A:G", "-S", "1", "-s", "12", "-n", "1"]),
    # long reads: the 256-bp and 480-bp kernel instantiations, truncation at -L / the 480-bp cap
    "long_300": (["--ref-bp", "300000", "--contigs", "2", "--reads", "200", "-M", "C:T", "--len", "300", "--len-jitter", "120",
                  "--max-sub", "12"],
                 ["-M", "C:T", "-S", "1", "-s", "12", "-n", "1"]),
    "long_490_g1": (["--ref-bp", "300000", "--contigs", "2", "--reads", "120", "-M", "A:G", "--p-conv", "0.8", "--len", "490",
                     "--len-jitter", "60", "--max-sub", "10", "--indel-frac", "0.3", "--indel-max", "1"],
                    ["-M", "A:G", "-S", "1", "-s", "12", "-g", "1", "-u"]),
    "long_400_multi": (["--ref-bp", "300000", "--contigs", "2", "--reads", "100", "-M", "A:CGT", "--p-conv", "0.2", "--len", "400",
                        "--max-sub", "8"],
                       ["-M", "A:CGT", "-S", "1", "-s", "14", "-v", "0.04"]),
    # a wide index interval: 2I = 24 (chain, phase) seeds per mode (the > 16-entry scan path), few seed segments
    "I12": (["--ref-bp", "300000", "--contigs", "2", "--reads", "300", "-M", "C:T", "--len", "120", "--len-jitter", "30", "--max-sub", "3"],
            ["-M", "C:T", "-S", "1", "-s", "12", "-I", "12", "-n", "1"]),
    "I3": (["--ref-bp", "300000", "--contigs", "2", "--reads", "300", "-M", "C:T", "--len", "100", "--len-jitter", "20", "--max-sub", "3"],
           ["-M", "C:T", "-S", "1", "-s", "12", "-I", "3"]),
    "I16_g1": (["--ref-bp", "300000", "--contigs", "2", "--reads", "200", "-M", "A:G", "--p-conv", "0.7", "--len", "150", "--max-sub", "3",
                "--indel-frac", "0.2", "--indel-max", "1"],
               ["-M", "A:G", "-S", "1", "-s", "12", "-I", "16", "-g", "1"]),
    # read lengths straddling the kernel instantiation bounds (128 / 256 bases), mixed in one batch
    "len_bound_128": (["--ref-bp", "300000", "--contigs", "2", "--reads", "240", "-M", "C:T", "--len", "136", "--len-jitter", "16",
                       "--max-sub", "4"],
                      ["-M", "C:T", "-S", "1", "-s", "12"]),
    "len_bound_256_g1": (["--ref-bp", "300000", "--contigs", "2", "--reads", "160", "-M", "T:-", "--p-conv", "0.02", "--len", "264", "--len-jitter", "16",
                          "--max-sub", "5", "--indel-frac", "0.3", "--indel-max", "1"],
                         ["-M", "T:-", "-S", "1", "-s", "12", "-g", "1", "-n", "1"]),
    # edge cases: reads around the minimum length, lower-case and IUPAC read bases, hidden -N (N counts as a mismatch)
    "edge_short": (["--ref-bp", "200000", "--contigs", "2", "--reads", "300", "-M", "C:T", "--len", "40", "--len-jitter", "30",
                    "--max-sub", "1", "--lower-reads-frac", "0.3", "--iupac-frac", "0.3"],
                   ["-M", "C:T", "-S", "1", "-s", "12", "-u", "-f", "2"]),
    "edge_Nmis": (["--ref-bp", "200000", "--contigs", "2", "--reads", "300", "-M", "C:T", "--n-frac", "0.6", "--max-sub", "3"],
                  ["-M", "C:T", "-S", "1", "-s", "12", "-u", "-N", "-n", "1"]),
    # FASTA reads input
    "fa_reads": (["--ref-bp", "200000", "--contigs", "2", "--reads", "200", "-M", "C:T", "--fasta-reads"],
                 ["-M", "C:T", "-S", "1", "-s", "12"]),
}

PE_FIXTURES = {
    "pe_ag_150": (["--ref-bp", "600000", "--contigs", "40", "--reads", "400", "-M", "A:G", "--p-conv", "0.9",
                   "--len", "150", "--rev-frac", "0.5"],
                  ["-M", "A:G", "-S", "1", "-s", "12"]),
    "pe_ct_100_u": (["--ref-bp", "400000", "--contigs", "2", "--reads", "400", "-M", "C:T", "--len", "100",
                     "--junk-frac", "0.1", "--max-sub", "8"],
                    ["-M", "C:T", "-S", "1", "-s", "12", "-u", "-R"]),
    "pe_rep_r2": (["--ref-bp", "400000", "--contigs", "2", "--reads", "200", "-M", "C:T", "--repeat-copies", "150",
                   "--repeat-len", "700", "--max-sub", "3"],
                  ["-M", "C:T", "-S", "1", "-s", "12", "-k", "1e-3", "-r", "2", "-w", "20", "-u"]),
    "pe_dirty_r1": (["--ref-bp", "400000", "--contigs", "3", "--reads", "400", "-M", "C:T", "--len", "100", "--len-jitter", "40",
                     "--n-frac", "0.4", "--junk-frac", "0.1", "--repeat-copies", "100", "--repeat-len", "500", "--max-sub", "6"],
                    ["-M", "C:T", "-S", "3", "-s", "12", "-k", "1e-3", "-u", "-w", "15", "-x", "700"]),
    "pe_rep_r0": (["--ref-bp", "400000", "--contigs", "2", "--reads", "200", "-M", "C:T", "--repeat-copies", "150",
                   "--repeat-len", "700", "--max-sub", "3"],
                  ["-M", "C:T", "-S", "1", "-s", "12", "-k", "1e-3", "-r", "0", "-u", "-w", "20"]),
    "pe_g2_n1": (["--ref-bp", "300000", "--contigs", "2", "--reads", "300", "-M", "A:CGT", "--p-conv", "0.3",
                  "--indel-frac", "0.3"],
                 ["-M", "A:CGT", "-S", "1", "-s", "12", "-g", "2", "-n", "1", "-u"]),
}


# contig-count edges (tools/gen_contigs.py): 5 000 contigs = three levels of the 64-ary contig search; 140 000 contigs = the
# reference's 18-bit contig field wraps (param.h:35-42). The 140 000-contig FASTA is regenerated, not committed (sha256 in the manifest).
CONTIG_FIXTURES = {
    "contigs_5k": ({"contigs": 5000, "seed": 3, "reads": 150, "len": 100, "rule": "A:G"}, ["-M", "A:G", "-S", "1", "-s", "12", "-n", "1"], True),
    "contigs_140k": ({"contigs": 140000, "seed": 1, "reads": 120, "len": 100, "rule": "A:G"}, ["-M", "A:G", "-S", "1", "-s", "12", "-n", "1"], False),
    "contigs_140k_g1": ({"contigs": 140000, "seed": 1, "reads": 120, "len": 100, "rule": "A:G"}, ["-M", "A:G", "-S", "2", "-s", "12", "-g", "1", "-R", "-u"], False),
}


def gz_write(path, data):
    with open(path, "wb") as raw:
        with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, compresslevel=9) as f:
            f.write(data)


BAM_CHECK = {"ct_basic", "varlen_trim", "ct_n1_dirty", "pe_ct_100_u", "pe_dirty_r1"}


def run(cmd, **kw):
    r = subprocess.run(cmd, **kw)
    if r.returncode != 0:
        raise SystemExit("command failed: %s" % " ".join(cmd))
    return r


def main():
    if not os.path.exists("/root/reference"):
        raise SystemExit("make_golden.py needs /root/reference (build container only)")
    run(["make", "-f", os.path.join(ROOT, "oracle", "Makefile.ref"), "-j8"], stdout=subprocess.DEVNULL)
    os.makedirs(GOLD, exist_ok=True)
    only = set(sys.argv[1:])
    manifest_path = os.path.join(GOLD, "manifest.json")
    manifest = json.load(open(manifest_path)) if os.path.exists(manifest_path) else {}
    for pe, table in ((False, FIXTURES), (True, PE_FIXTURES)):
        for name, (gen_args, flags) in table.items():
            if only and name not in only:
                continue
            with tempfile.TemporaryDirectory() as td:
                fa = os.path.join(td, name + ".fa")
                fasta_reads = "--fasta-reads" in gen_args
                fq = os.path.join(td, name + (".reads.fa" if fasta_reads else ".fq"))
                fq2 = os.path.join(td, name + "_2.fq")
                sam = os.path.join(td, name + ".sam")
                g = [sys.executable, GEN, "--ref-out", fa, "--reads-out", fq] + gen_args
                if pe:
                    g += ["--reads2-out", fq2]
                run(g)
                cmd = [REF_BIN, "-a", fq] + (["-b", fq2] if pe else []) + ["-d", fa] + flags + ["-p", "1", "-o", sam]
                run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                body = b"".join(l for l in open(sam, "rb") if not l.startswith(b"@PG"))
                gz_write(os.path.join(GOLD, name + ".fa.gz"), open(fa, "rb").read())
                gz_write(os.path.join(GOLD, name + (".reads.fa.gz" if fasta_reads else ".fq.gz")), open(fq, "rb").read())
                if pe:
                    gz_write(os.path.join(GOLD, name + "_2.fq.gz"), open(fq2, "rb").read())
                gz_write(os.path.join(GOLD, name + ".sam.gz"), body)
                nrec = sum(1 for l in body.splitlines() if not l.startswith(b"@"))
                manifest[name] = {"gen": gen_args, "flags": flags, "pe": pe, "records": nrec,
                                  "reads_file": name + (".reads.fa.gz" if fasta_reads else ".fq.gz")}
                if name in BAM_CHECK:
                    # the same reads as an unaligned BAM (mates interleaved for PE): the reference must print the same SAM, so the
                    # golden SAM also pins the CLI's BAM reader (tests convert the FASTQ with the same tool)
                    bam = os.path.join(td, name + ".bam")
                    run([sys.executable, os.path.join(ROOT, "tools", "fq2bam.py"), fq, bam] + ([fq2] if pe else []))
                    sam2 = os.path.join(td, name + ".bam.sam")
                    run([REF_BIN, "-a", bam] + (["-b", bam] if pe else []) + ["-d", fa] + flags + ["-p", "1", "-o", sam2],
                        stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                    if b"".join(l for l in open(sam2, "rb") if not l.startswith(b"@PG")) != body:
                        raise SystemExit("%s: the reference prints a different SAM for the BAM form of the reads" % name)
                    manifest[name]["bam_input_checked"] = True
                print("%-16s %5d records  flags: %s" % (name, nrec, " ".join(flags)))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import hashlib
    import gen_contigs as gc
    for name, (g, flags, commit_fa) in CONTIG_FIXTURES.items():
        if only and name not in only:
            continue
        with tempfile.TemporaryDirectory() as td:
            sizes, starts, seq = gc.make_reference(g["contigs"], g["seed"])
            fa_b = gc.fasta_bytes(sizes, starts, seq)
            to = g["rule"][2] if g["rule"][2] in "ACGT" else g["rule"][0]
            fq_b = gc.fastq_bytes(gc.make_reads(sizes, starts, seq, g["reads"], g["seed"], g["len"], g["rule"][0], to))
            fa, fq, sam = os.path.join(td, "r.fa"), os.path.join(td, "r.fq"), os.path.join(td, "r.sam")
            open(fa, "wb").write(fa_b)
            open(fq, "wb").write(fq_b)
            run([REF_BIN, "-a", fq, "-d", fa] + flags + ["-p", "1", "-o", sam], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            body = b"".join(l for l in open(sam, "rb") if not l.startswith(b"@PG"))
            if commit_fa:
                gz_write(os.path.join(GOLD, name + ".fa.gz"), fa_b)
            gz_write(os.path.join(GOLD, name + ".fq.gz"), fq_b)
            gz_write(os.path.join(GOLD, name + ".sam.gz"), body)
            nrec = sum(1 for l in body.splitlines() if not l.startswith(b"@"))
            manifest[name] = {"gen": ["gen_contigs"], "flags": flags, "pe": False, "records": nrec, "reads_file": name + ".fq.gz",
                              "fasta_gen": None if commit_fa else {"contigs": g["contigs"], "seed": g["seed"], "sha256": hashlib.sha256(fa_b).hexdigest()}}
            print("%-16s %5d records  flags: %s" % (name, nrec, " ".join(flags)))
    json.dump(manifest, open(manifest_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
