#!/usr/bin/env python3
"""Does the headline kernel's time depend on WHERE its index landed in HBM? One process: the genome and the reads are made once; the core (reference +
seed index, ~45 GB of hipMalloc) is created, timed on three launches of 10 M reads and destroyed, R times over; between rounds a block of `--hold` GB
may be kept allocated so that the next round's buffers cannot land where the last one's were.

  gpurun -- 'python3 tools/probe_placement.py --rounds 6 > gpurun_out/placement.log'
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--batch", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--hold", type=float, default=0.0, help="GB kept allocated from one round to the next (shifts the placement)")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--which", default="", help="comma list of buffer classes to move one at a time (basal_core_move_buffers: 0 locs, 1 flank words, 2 seed words, 3 k-mer tables, 4 reference, 5 hit logs)")
    ap.add_argument("--pre", type=float, default=0.0, help="GiB allocated (and kept) before the first core is created")
    ap.add_argument("--fork", action="store_true", help="every round: a new core timed as built, then on a second set made in one go")
    ap.add_argument("--pair", action="store_true", help="two cores resident at once, launches alternating")
    a = ap.parse_args()
    import torch
    import basal_amd as B
    from basal_amd import core as bc
    import synth_gpu
    dev = torch.device("cuda", 0)
    params = B.Params("C:T", ["-M", "C:T", "-S", "1"])
    L = B.lib()
    G = synth_gpu.make_genome(params, dev, scale=a.scale, seed=1, repeat_copies=40000, realistic=True)
    words = [w.cpu().numpy().view(np.uint64) for w in G.words]
    sizes = np.array(G.sizes, dtype=np.uint32)
    blocks = np.ascontiguousarray(G.blocks)
    anchors, rc_offsets = np.ascontiguousarray(G.anchors), np.ascontiguousarray(G.rc_offsets)
    chunks = []
    for c0 in range(0, a.batch, 2_000_000):
        nb = min(2_000_000, a.batch - c0)
        b, _, _, _ = synth_gpu.make_reads(G, nb, dev, read_len=100, seed=1000 + c0 // 2_000_000, conv_from=1, conv_to=3, p_conv=0.95)
        chunks.append(b)
    d_bases = torch.cat(chunks)
    del chunks, G
    torch.cuda.empty_cache()
    seq = C.create_string_buffer(b"A" * 100, 102)
    qual = C.create_string_buffer(b"I" * 100, 102)
    ms = C.c_uint32()
    assert L.basal_host_filter_read(C.byref(params.c), seq, qual, C.byref(ms)) == 0
    descs = np.zeros(a.batch, bc.READ_DTYPE)
    descs["seq_off"] = np.arange(a.batch, dtype=np.uint64) * 100
    descs["index"] = np.arange(a.batch, dtype=np.uint32)
    descs["len"] = 100
    descs["max_snp"] = ms.value
    descs["stale_idx"] = B.STALE_NONE
    d_reads = torch.from_numpy(descs.view(np.uint8).reshape(-1)).to(dev)
    d_results = torch.zeros(a.batch * 32, dtype=torch.uint8, device=dev)
    d_used = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    held = []
    if a.pre > 0:  # a blocker allocated before the first core: shifts where the core's buffers land in HBM
        held.append(torch.empty(int(a.pre * 2 ** 30), dtype=torch.uint8, device=dev))
        print("blocker of %.0f GiB at %#x" % (a.pre, held[-1].data_ptr()), flush=True)

    def make_core():
        core = B.Core(params, 0)
        bc._check(L.basal_core_set_reference(core.h, words[0].ctypes.data, words[1].ctypes.data, len(words[0]), anchors.ctypes.data, sizes.ctypes.data,
                                             rc_offsets.ctypes.data, len(sizes)), "set_reference")
        mk = C.c_uint32()
        bc._check(L.basal_core_build_index(core.h, blocks.ctypes.data, len(blocks), C.byref(mk)), "build_index")
        core.set_timing(True)
        return core

    def launch(core):
        bc._check(L.basal_core_align_batch_device(core.h, d_bases.data_ptr(), d_reads.data_ptr(), a.batch, None, 0, B.STREAM_NONE, d_results.data_ptr(), None, 0,
                                                  d_used.data_ptr(), None, 100, stream), "align_batch_device")
        torch.cuda.synchronize()
        return core.kernel_ms()

    if a.which:  # one core; one class of buffers moved at a time: whose placement is it?
        names = ["locs", "flank words", "seed words", "k-mer tables", "reference", "hit logs"]
        core = make_core()
        launch(core)
        print("start: %s" % " ".join("%.2f" % launch(core) for _ in range(a.steps)), flush=True)
        for r in range(a.rounds):
            for w in [int(x) for x in a.which.split(",")]:
                rc = L.basal_core_move_buffers(core.h, w)
                if rc < 0:
                    print("replace %s failed: %s" % (names[w], L.basal_last_error().decode()), flush=True)
                    continue
                print("round %d, %s moved: %s" % (r, names[w], " ".join("%.2f" % launch(core) for _ in range(a.steps))), flush=True)
        core.close()
        return
    if a.fork:  # every round: a new core timed as built, then on a second set made in one go (basal_core_placement_fork), the first freed
        for r in range(a.rounds):
            core = make_core()
            launch(core)
            t0 = [launch(core) for _ in range(a.steps)]
            n = L.basal_core_placement_fork(core.h)
            launch(core)
            t1 = [launch(core) for _ in range(a.steps)]
            L.basal_core_placement_commit(core.h)
            n2 = L.basal_core_placement_fork(core.h)
            launch(core)
            t2 = [launch(core) for _ in range(a.steps)]
            print("round %d: as built %s | forked (%d buffers) %s | first set freed, forked again (%d) %s" % (r, " ".join("%.2f" % m for m in t0), n, " ".join("%.2f" % m for m in t1), n2,
                                                                                                 " ".join("%.2f" % m for m in t2)), flush=True)
            core.close()
        return
    if a.pair:  # two cores resident at once, launches alternating: placement (A and B differ, each steady) or the clocks (both drift together)?
        for r in range(a.rounds):
            A, Bc = make_core(), make_core()
            launch(A), launch(Bc)
            ta, tb = [], []
            for i in range(a.steps):
                ta.append(launch(A))
                tb.append(launch(Bc))
            print("pair %d: A %s | B %s" % (r, " ".join("%.2f" % m for m in ta), " ".join("%.2f" % m for m in tb)), flush=True)
            A.close(), Bc.close()
        return
    for r in range(a.rounds):
        t0 = time.time()
        core = B.Core(params, 0)
        bc._check(L.basal_core_set_reference(core.h, words[0].ctypes.data, words[1].ctypes.data, len(words[0]), anchors.ctypes.data, sizes.ctypes.data,
                                             rc_offsets.ctypes.data, len(sizes)), "set_reference")
        mk = C.c_uint32()
        bc._check(L.basal_core_build_index(core.h, blocks.ctypes.data, len(blocks), C.byref(mk)), "build_index")
        core.set_timing(True)
        ms_list = []
        for i in range(a.steps + 1):
            bc._check(L.basal_core_align_batch_device(core.h, d_bases.data_ptr(), d_reads.data_ptr(), a.batch, None, 0, B.STREAM_NONE, d_results.data_ptr(), None, 0,
                                                      d_used.data_ptr(), None, 100, stream), "align_batch_device")
            torch.cuda.synchronize()
            ms_list.append(core.kernel_ms())
        free, total = torch.cuda.mem_get_info()
        print("round %d: kernel ms %s (set-up %.1f s, %.1f GB of HBM free)" % (r, " ".join("%.2f" % m for m in ms_list[1:]), time.time() - t0, free / 1e9), flush=True)
        core.close()
        del core
        if a.hold > 0:
            held.append(torch.empty(int(a.hold * 1e9), dtype=torch.uint8, device=dev))


if __name__ == "__main__":
    main()
