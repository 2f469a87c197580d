"""Hands a GPU-resident reference + seed index to the CPU oracle (checker side only: bench.py's
cpu_baseline leg and the at-scale GPU tests). Nothing here is used by the product path."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc  # noqa: E402

BEST_DTYPE = np.dtype([("best_level", "<u4"), ("n_hit", "<u4"), ("n_chit", "<u4"), ("chr", "<u4"), ("loc", "<u4"), ("gap_size", "<i4"),
                       ("gap_pos", "<u4"), ("chain", "<u4")])


class OracleOnIndex:
    """orc_ref built from the arrays the GPU core holds (get_index) + the packed reference words."""

    def __init__(self, core, params, flags, names, sizes, words):
        self.L = orc.lib()
        tk = 3 ** params.c.seed_size
        self.off, self.nfwd, self.locs, mk = core.get_index(tk)
        self.n_tot = np.diff(self.off.astype(np.uint64)).astype(np.uint32)
        self.off64 = self.off.astype(np.uint64)
        self.p = orc.make_param(flags)
        self.p.max_kmer_num = mk
        self.sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        self.words = words
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        self.ref = self.L.orc_ref_from_arrays(len(names), arr, self.sizes.ctypes.data, words[0].ctypes.data, words[1].ctypes.data, len(words[0]), tk,
                                              self.n_tot.ctypes.data, self.nfwd.ctypes.data, self.off64.ctypes.data, self.locs.ctypes.data, len(self.locs))
        self.L.orc_align_batch_mt.argtypes = [C.POINTER(orc.orc_param), C.POINTER(orc.orc_ref), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.POINTER(orc.orc_counters), C.POINTER(C.c_double)]

    def align(self, bases, seq_off, lens, index, max_snp, threads):
        n = len(lens)
        best = np.zeros(n, BEST_DTYPE)
        cnt = orc.orc_counters()
        secs = C.c_double()
        bases = np.ascontiguousarray(bases, np.uint8)
        seq_off = np.ascontiguousarray(seq_off, np.uint32)
        lens = np.ascontiguousarray(lens, np.uint16)
        index = np.ascontiguousarray(index, np.uint32)
        max_snp = np.ascontiguousarray(max_snp, np.uint8)
        self.L.orc_align_batch_mt(C.byref(self.p), self.ref, bases.ctypes.data, seq_off.ctypes.data, lens.ctypes.data, index.ctypes.data,
                                  max_snp.ctypes.data, n, threads, best.ctypes.data, C.byref(cnt), C.byref(secs))
        return best, cnt, secs.value


def differing(gpu_results, best):
    """Indices where the GPU's per-read summary differs from the oracle's."""
    same = ((gpu_results["best_level"].astype(np.uint32) == (best["best_level"] & 0xFF)) & (gpu_results["n_hit"] == best["n_hit"]) &
            (gpu_results["n_chit"] == best["n_chit"]))
    hit = best["best_level"] != 0xFF
    b = gpu_results["best"]
    same &= ~hit | ((b["loc"] == best["loc"]) & (b["chr"] == best["chr"]) & (b["chain"] == best["chain"]) &
                    (b["gap_size"].astype(np.int32) == best["gap_size"]) & (b["gap_pos"].astype(np.uint32) == best["gap_pos"]))
    return np.nonzero(~same)[0]
