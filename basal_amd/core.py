"""ctypes mirror of include/basal_core.h.  Loads basal_amd/lib/libbasal_amd.so and fails loudly if
it is missing (there is no Python or CPU implementation of the hot path behind this module)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
STREAM_NONE, STREAM_BEST, STREAM_ALL = 0, 1, 2
STALE_NONE, STALE_CARRY = 0xFFFFFFFF, 0xFFFFFFFE


class BasalError(RuntimeError):
    pass


def lib_path():
    return os.environ.get("BASAL_LIB") or os.path.join(_HERE, "lib", "libbasal_amd.so")


def build(verbose=False):
    """Compile the HIP core and the C++ host for gfx950 (hipcc cross-compiles without a GPU)."""
    # `chk`: the diagnostic twin of the library that tests/test_gpu_parity.py runs the -g fixtures through (it travels prebuilt to the GPU box)
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j4", "all", "chk"], capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise BasalError("building libbasal_amd.so failed:\n%s\n%s" % (r.stdout, r.stderr))


class basal_params(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "seed_size", "index_interval", "max_snp_num", "max_num_hits", "chains", "randseed", "gap", "gap_edge",
        "n_mis", "new_rule", "report_repeat_hits", "pairend", "max_ns", "min_read_size", "max_readlen",
        "trim_qual_threshold", "n_adapter", "out_ref", "out_unmap", "min_insert", "max_insert")] + [
        ("max_kmer_ratio", C.c_float), ("zero_qual", C.c_uint8), ("default_qual", C.c_uint8), ("pad0", C.c_uint8),
        ("pad1", C.c_uint8), ("adapter", (C.c_char * 128) * 10), ("refnt", C.c_char), ("readnts", C.c_char * 5),
        ("readnt_cnt", C.c_uint8), ("pad2", C.c_uint8), ("alphabet", C.c_uint8 * 256), ("rev_alphabet", C.c_uint8 * 256),
        ("reg_alphabet", C.c_uint8 * 256), ("alphabet_mread", C.c_uint8 * 256), ("rev_alphabet_mread", C.c_uint8 * 256),
        ("useful_nt", C.c_char * 12)]


class basal_hit(C.Structure):
    _fields_ = [("loc", C.c_uint32), ("chr", C.c_uint32), ("gap_size", C.c_int8), ("strand", C.c_uint8),
                ("gap_pos", C.c_uint16), ("level", C.c_uint8), ("chain", C.c_uint8), ("mode", C.c_uint8), ("pad", C.c_uint8)]


class basal_result(C.Structure):
    _fields_ = [("best", basal_hit), ("n_hit", C.c_uint16), ("n_chit", C.c_uint16), ("best_level", C.c_uint8),
                ("start_off", C.c_uint8 * 2), ("status", C.c_uint8), ("stream_first", C.c_uint32), ("stream_n", C.c_uint32)]


class basal_read(C.Structure):
    _fields_ = [("seq_off", C.c_uint32), ("index", C.c_uint32), ("len", C.c_uint16), ("readset", C.c_uint8),
                ("max_snp", C.c_uint8), ("stale_idx", C.c_uint32)]


class basal_stale(C.Structure):
    _fields_ = [("src", C.c_uint32), ("overlay", (C.c_uint32 * 15) * 2)]


class basal_mate(C.Structure):
    _fields_ = [("name", C.c_char_p), ("seq", C.c_char_p), ("qual", C.c_char_p), ("readset", C.c_uint32), ("index", C.c_uint32),
                ("max_snp", C.c_uint32), ("qc_failed", C.c_int), ("res", C.POINTER(basal_result))]


class basal_rawread(C.Structure):
    _fields_ = [("name_off", C.c_uint32), ("seq_off", C.c_uint32), ("qual_off", C.c_uint32), ("name_len", C.c_uint16),
                ("seq_len", C.c_uint16), ("qual_len", C.c_uint16), ("readset", C.c_uint8), ("pad", C.c_uint8), ("index", C.c_uint32)]


class basal_pipe_opts(C.Structure):
    _fields_ = [("depth", C.c_uint32), ("max_reads", C.c_uint32), ("max_bytes", C.c_uint64), ("output", C.c_uint32), ("flags", C.c_uint32)]


class basal_batch_stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_reads", "n_aligned", "n_unique", "n_multiple", "n_filtered")] + [
        (n, C.c_float) for n in ("ms_h2d", "ms_prep", "ms_align", "ms_format", "ms_d2h")] + [("pe", C.c_uint32 * 9), ("pad", C.c_uint32)]


PIPE_OUT_SAM, PIPE_OUT_RESULTS = 0, 1
FMT_FASTQ, FMT_FASTA = 0, 1
RAWREAD_DTYPE = np.dtype([("name_off", "<u4"), ("seq_off", "<u4"), ("qual_off", "<u4"), ("name_len", "<u2"), ("seq_len", "<u2"),
                          ("qual_len", "<u2"), ("readset", "u1"), ("pad", "u1"), ("index", "<u4")])
assert RAWREAD_DTYPE.itemsize == 24 and C.sizeof(basal_rawread) == 24
READ_ALLMODES = 0x80
HIT_DTYPE = np.dtype([("loc", "<u4"), ("chr", "<u4"), ("gap_size", "i1"), ("strand", "u1"), ("gap_pos", "<u2"),
                      ("level", "u1"), ("chain", "u1"), ("mode", "u1"), ("pad", "u1")])
RESULT_DTYPE = np.dtype([("best", HIT_DTYPE), ("n_hit", "<u2"), ("n_chit", "<u2"), ("best_level", "u1"),
                         ("start_off", "u1", (2,)), ("status", "u1"), ("stream_first", "<u4"), ("stream_n", "<u4")])
READ_DTYPE = np.dtype([("seq_off", "<u4"), ("index", "<u4"), ("len", "<u2"), ("readset", "u1"), ("max_snp", "u1"),
                       ("stale_idx", "<u4")])
STALE_DTYPE = np.dtype([("src", "<u4"), ("overlay", "<u4", (2, 15))])
assert HIT_DTYPE.itemsize == 16 and RESULT_DTYPE.itemsize == 32 and READ_DTYPE.itemsize == 16
assert C.sizeof(basal_hit) == 16 and C.sizeof(basal_result) == 32 and C.sizeof(basal_read) == 16
assert C.sizeof(basal_stale) == 124 and STALE_DTYPE.itemsize == 124

_lib = None

# every symbol include/basal_core.h declares: (name, restype, argtypes)
_P = C.POINTER
_vp, _u32, _u64, _i = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
SYMBOLS = [
    ("basal_core_create", _i, [_P(basal_params), _i, _P(_vp)]),
    ("basal_core_destroy", None, [_vp]),
    ("basal_core_set_reference", _i, [_vp, _vp, _vp, _u64, _vp, _vp, _vp, _u32]),
    ("basal_core_set_index", _i, [_vp, _vp, _vp, _vp, _u64, _u32]),
    ("basal_core_build_index", _i, [_vp, _vp, _u64, _P(_u32)]),
    ("basal_core_get_index", _i, [_vp, _vp, _vp, _vp, _P(_u64), _P(_u32)]),
    ("basal_core_align_batch", _i, [_vp, _vp, _u64, _vp, _u32, _vp, _u32, _i, _vp, _vp, _u64, _P(_u64), _vp]),
    ("basal_core_align_batch_device", _i, [_vp, _vp, _vp, _u32, _vp, _u32, _i, _vp, _vp, _u64, _vp, _vp, _u32, _vp]),
    ("basal_core_sync_check", _i, [_vp]),
    ("basal_core_occupancy_report", _i, [C.c_char_p, C.c_size_t]),
    ("basal_core_placement_fork", _i, [_vp]),
    ("basal_core_placement_swap", _i, [_vp]),
    ("basal_core_placement_commit", _i, [_vp]),
    ("basal_core_move_buffers", _i, [_vp, _i]),
    ("basal_core_set_timing", _i, [_vp, _i]),
    ("basal_core_last_kernel_ms", C.c_float, [_vp]),
    ("basal_core_last_pair_ms", C.c_float, [_vp]),
    ("basal_core_launch_info", _i, [_vp, _P(_u32), _P(_u32), _P(_u32)]),
    ("basal_last_error", C.c_char_p, []),
    ("basal_core_set_contig_names", _i, [_vp, _P(C.c_char_p), _u32]),
    ("basal_pipe_create", _i, [_vp, _P(basal_pipe_opts), _P(_vp)]),
    ("basal_pipe_submit_text_pairs", _i, [_vp, _u64, _u64, _u32, _i, _u32]),
    ("basal_pipe_create_multi", _i, [_P(_vp), _i, _P(basal_pipe_opts), _P(_vp)]),
    ("basal_pipe_destroy", None, [_vp]),
    ("basal_pipe_acquire", _i, [_vp, _P(_vp), _P(_vp)]),
    ("basal_pipe_submit_text", _i, [_vp, _u64, _i, _u32, _u32]),
    ("basal_pipe_submit_records", _i, [_vp, _u64, _u32]),
    ("basal_pipe_submit_prepared", _i, [_vp, _u64, _u32, _u32]),
    ("basal_pipe_collect", _i, [_vp, _P(_vp), _P(_u64), _P(basal_batch_stats)]),
    ("basal_pipe_release", _i, [_vp]),
    ("basal_pipe_cancel", _i, [_vp]),
    ("basal_pipe_stop", _i, [_vp]),
    ("basal_pipe_rewind", _i, [_vp]),
    ("basal_core_align_pairs_batch", _i, [_vp, _vp, _u64, _vp, _u32, _vp, _u32, _vp, _vp, _u64, _P(_u64), _vp, _vp]),
    ("basal_host_format_pe_records", C.c_int64, [_P(basal_params), _vp, _P(basal_mate), _P(basal_mate), _vp, _u32, C.c_char_p, C.c_size_t]),
    ("basal_shard_range", None, [_u64, _u32, _u32, _P(_u64), _P(_u64)]),
    ("basal_multi_create", _i, [_P(basal_params), _P(_i), _i, _P(_vp)]),
    ("basal_multi_destroy", None, [_vp]),
    ("basal_multi_ndev", _i, [_vp]),
    ("basal_multi_core", _vp, [_vp, _i]),
    ("basal_multi_last_h2d_bytes", _u64, [_vp, _i]),
    ("basal_multi_upload", _i, [_vp, _vp, _i, _P(_u32)]),
    ("basal_multi_align_batch", _i, [_vp, _vp, _u64, _vp, _u32, _vp, _u32, _i, _vp, _vp, _u64, _P(_u64), _vp]),
    ("basal_pipe_set_read_range", _i, [_vp, _u32, _u32]),
    ("basal_host_params_defaults", None, [_P(basal_params)]),
    ("basal_host_params_set_seed_size", _i, [_P(basal_params), _i]),
    ("basal_host_params_set_align", _i, [_P(basal_params), C.c_char_p]),
    ("basal_host_params_set_v", None, [_P(basal_params), C.c_double]),
    ("basal_host_ref_load", _i, [_P(basal_params), C.c_char_p, _P(_vp)]),
    ("basal_host_ref_load_mem", _i, [_P(basal_params), C.c_char_p, C.c_size_t, _P(_vp)]),
    ("basal_host_ref_free", None, [_vp]),
    ("basal_host_ref_ncontig", _u32, [_vp]),
    ("basal_host_ref_name", C.c_char_p, [_vp, _u32]),
    ("basal_host_ref_sizes", _P(_u32), [_vp]),
    ("basal_host_ref_rc_offsets", _P(_u32), [_vp]),
    ("basal_host_ref_anchors", _P(_u32), [_vp]),
    ("basal_host_ref_nwords", _u64, [_vp]),
    ("basal_host_ref_words", _P(_u64), [_vp, _i]),
    ("basal_host_ref_nblocks", _u64, [_vp]),
    ("basal_host_ref_blocks", _P(_u32), [_vp]),
    ("basal_host_ref_build_index", _i, [_vp, _P(basal_params), _i]),
    ("basal_host_ref_total_kmers", _u32, [_vp]),
    ("basal_host_ref_kmer_off", _P(_u32), [_vp]),
    ("basal_host_ref_kmer_nfwd", _P(_u32), [_vp]),
    ("basal_host_ref_locs", _P(_u32), [_vp]),
    ("basal_host_ref_nlocs", _u64, [_vp]),
    ("basal_host_ref_max_kmer_num", _u32, [_vp]),
    ("basal_host_ref_upload", _i, [_vp, _vp, _i, _P(_u32)]),
    ("basal_host_stale_new", _vp, [_P(basal_params)]),
    ("basal_host_stale_free", None, [_vp]),
    ("basal_host_stale_begin_batch", None, [_vp]),
    ("basal_host_stale_visit", _i, [_vp, C.c_char_p, _u32, _u32, _i, _u32, _P(basal_stale)]),
    ("basal_host_filter_read", _i, [_P(basal_params), C.c_char_p, C.c_char_p, _P(_u32)]),
    ("basal_host_format_se", C.c_int64, [_P(basal_params), _vp, C.c_char_p, C.c_char_p, C.c_char_p, _u32, _i,
                                         _P(basal_result), _vp, C.c_char_p, C.c_size_t]),
    ("basal_host_format_pe", C.c_int64, [_P(basal_params), _vp, _P(basal_mate), _P(basal_mate), _vp, C.c_char_p, C.c_size_t, _P(_u32)]),
    ("basal_host_fix_pair_names", _i, [C.c_char_p, C.c_char_p]),
    ("basal_host_sam_header", C.c_int64, [_vp, C.c_char_p, C.c_char_p, C.c_size_t]),
]


def lib():
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise BasalError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no fallback implementation)" % path)
        _lib = C.CDLL(path)
        for name, res, args in SYMBOLS:
            if os.environ.get("BASAL_LIB") and not hasattr(_lib, name) and name in ("basal_core_occupancy_report", "basal_multi_last_h2d_bytes", "basal_core_placement_fork", "basal_core_placement_swap", "basal_core_placement_commit", "basal_core_move_buffers"):
                continue  # (A/B runs against an older build of the library, tools/run_ab_*.sh: the instrumentation entry points are newer than it)
            f = getattr(_lib, name)
            f.restype = res
            f.argtypes = args
    return _lib


def _check(rc, what):
    if rc != 0:
        raise BasalError("%s failed (%d): %s" % (what, rc, lib().basal_last_error().decode()))


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype)


class Params:
    """basal_params filled the way the reference's command line would (main.cpp:272-364)."""

    def __init__(self, rule, flags=()):
        self.c = basal_params()
        L = lib()
        L.basal_host_params_defaults(C.byref(self.c))
        self.threads = 1
        it = iter(flags)
        for f in it:
            if f == "-s":
                _check(L.basal_host_params_set_seed_size(C.byref(self.c), int(next(it))), "-s")
            elif f == "-I":
                self.c.index_interval = int(next(it))
            elif f == "-v":
                L.basal_host_params_set_v(C.byref(self.c), float(next(it)))
            elif f == "-g":
                self.c.gap = min(3, int(next(it)))
            elif f == "-w":
                self.c.max_num_hits = int(next(it))
            elif f == "-n":
                self.c.chains = int(next(it))
            elif f == "-S":
                self.c.randseed = int(next(it))
            elif f == "-r":
                self.c.report_repeat_hits = int(next(it))
            elif f == "-k":
                self.c.max_kmer_ratio = float(next(it))
            elif f == "-f":
                self.c.max_ns = int(next(it))
            elif f == "-q":
                self.c.trim_qual_threshold = int(next(it))
            elif f == "-z":
                self.c.zero_qual = int(next(it))
            elif f == "-L":
                self.c.max_readlen = int(next(it))
            elif f == "-m":
                self.c.min_insert = int(next(it))
            elif f == "-x":
                self.c.max_insert = int(next(it))
            elif f == "-p":
                self.threads = int(next(it))
            elif f == "-A":
                a = next(it).encode()
                C.memmove(self.c.adapter[self.c.n_adapter], a, min(len(a), 127))
                self.c.n_adapter += 1
            elif f == "-R":
                self.c.out_ref = 1
            elif f == "-u":
                self.c.out_unmap = 1
            elif f == "-N":
                self.c.n_mis = 1
            elif f in ("-M",):
                next(it)
            elif f in ("-H",):
                pass
            else:
                raise BasalError("unknown flag %s" % f)
        _check(L.basal_host_params_set_align(C.byref(self.c), rule.encode()), "-M")


class Reference:
    def __init__(self, params, fasta_path=None, fasta_bytes=None):
        self.params = params
        self.h = C.c_void_p()
        L = lib()
        if fasta_path is not None:
            _check(L.basal_host_ref_load(C.byref(params.c), fasta_path.encode(), C.byref(self.h)), "ref_load")
        else:
            _check(L.basal_host_ref_load_mem(C.byref(params.c), fasta_bytes, len(fasta_bytes), C.byref(self.h)), "ref_load")

    def close(self):
        if self.h:
            lib().basal_host_ref_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ncontig(self):
        return lib().basal_host_ref_ncontig(self.h)

    def names(self):
        return [lib().basal_host_ref_name(self.h, i).decode() for i in range(self.ncontig)]

    def sizes(self):
        return _arr(lib().basal_host_ref_sizes(self.h), self.ncontig, np.uint32)

    def rc_offsets(self):
        return _arr(lib().basal_host_ref_rc_offsets(self.h), self.ncontig, np.uint32)

    def anchors(self):
        return _arr(lib().basal_host_ref_anchors(self.h), self.ncontig + 1, np.uint32)

    def words(self, strand):
        return _arr(lib().basal_host_ref_words(self.h, strand), lib().basal_host_ref_nwords(self.h), np.uint64)

    def blocks(self):
        return _arr(lib().basal_host_ref_blocks(self.h), 3 * lib().basal_host_ref_nblocks(self.h), np.uint32).reshape(-1, 3)

    def build_index(self, threads=4):
        _check(lib().basal_host_ref_build_index(self.h, C.byref(self.params.c), threads), "build_index")

    def index(self):
        L = lib()
        tk = L.basal_host_ref_total_kmers(self.h)
        return (_arr(L.basal_host_ref_kmer_off(self.h), tk + 1, np.uint32), _arr(L.basal_host_ref_kmer_nfwd(self.h), tk, np.uint32),
                _arr(L.basal_host_ref_locs(self.h), L.basal_host_ref_nlocs(self.h), np.uint32), L.basal_host_ref_max_kmer_num(self.h))


class Core:
    """One GPU core object (basal_core_t)."""

    def __init__(self, params, device=0):
        self.params = params
        self.h = C.c_void_p()
        _check(lib().basal_core_create(C.byref(params.c), device, C.byref(self.h)), "core_create")

    def close(self):
        if self.h:
            lib().basal_core_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, ref, build_on_gpu=False):
        mk = C.c_uint32()
        _check(lib().basal_host_ref_upload(ref.h, self.h, int(build_on_gpu), C.byref(mk)), "ref_upload")
        return mk.value

    def get_index(self, total_kmers):
        L = lib()
        n = C.c_uint64()
        mk = C.c_uint32()
        _check(L.basal_core_get_index(self.h, None, None, None, C.byref(n), C.byref(mk)), "get_index")
        off = np.zeros(total_kmers + 1, np.uint32)
        nf = np.zeros(total_kmers, np.uint32)
        locs = np.zeros(n.value, np.uint32)
        _check(L.basal_core_get_index(self.h, off.ctypes.data, nf.ctypes.data, locs.ctypes.data, C.byref(n), C.byref(mk)), "get_index")
        return off, nf, locs, mk.value

    def align_batch(self, bases, reads, stream_mode=STREAM_NONE, stream_cap=0, carry=None, stales=None):
        """bases: uint8 array; reads: READ_DTYPE array; stales: STALE_DTYPE array. Returns (results, stream, carry)."""
        n = len(reads)
        res = np.zeros(n, RESULT_DTYPE)
        stream = np.zeros(max(stream_cap, 1), HIT_DTYPE)
        used = C.c_uint64()
        cy = np.zeros((2, 2), np.uint8) if carry is None else np.array(carry, np.uint8).reshape(2, 2).copy()
        bases = np.ascontiguousarray(bases, np.uint8)
        reads = np.ascontiguousarray(reads)
        stales = np.zeros(0, STALE_DTYPE) if stales is None else np.ascontiguousarray(stales)
        rc = lib().basal_core_align_batch(self.h, bases.ctypes.data, len(bases), reads.ctypes.data, n,
                                          stales.ctypes.data if len(stales) else None, len(stales), stream_mode,
                                          res.ctypes.data, stream.ctypes.data if stream_mode else None, stream_cap,
                                          C.byref(used), cy.ctypes.data)
        _check(rc, "align_batch")
        return res, stream[: used.value], cy

    def kernel_ms(self):
        return float(lib().basal_core_last_kernel_ms(self.h))

    def set_timing(self, on=True):
        lib().basal_core_set_timing(self.h, int(on))

    def launch_info(self):
        b, t, l = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(lib().basal_core_launch_info(self.h, C.byref(b), C.byref(t), C.byref(l)), "launch_info")
        return b.value, t.value, l.value


    def set_contig_names(self, names):
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        _check(lib().basal_core_set_contig_names(self.h, arr, len(names)), "set_contig_names")


class Pipe:
    """basal_pipe_t: batches of raw read text / raw read tables / prepared reads in, SAM text / basal_result records out."""

    def __init__(self, core, depth=3, max_reads=1 << 16, max_bytes=32 << 20, output=PIPE_OUT_SAM, pairs=False):
        """core: one Core, or a list of Cores (one per GPU, all staged alike): batches then fan out over them (basal_pipe_create_multi)."""
        self.core = core
        self.h = C.c_void_p()
        self.max_reads = (max_reads + 4095) & ~4095
        self.max_bytes = (max_bytes + 4095) & ~4095
        o = basal_pipe_opts(depth, max_reads, max_bytes, output, 1 if pairs else 0)
        if isinstance(core, (list, tuple)):
            arr = (C.c_void_p * len(core))(*[c.h for c in core])
            _check(lib().basal_pipe_create_multi(arr, len(core), C.byref(o), C.byref(self.h)), "pipe_create_multi")
        else:
            _check(lib().basal_pipe_create(core.h, C.byref(o), C.byref(self.h)), "pipe_create")

    def close(self):
        if self.h:
            lib().basal_pipe_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def acquire(self):
        """(blob, raw): numpy views of the next slot's page-locked input buffers."""
        blob, raw = C.c_void_p(), C.c_void_p()
        _check(lib().basal_pipe_acquire(self.h, C.byref(blob), C.byref(raw)), "pipe_acquire")
        b = np.ctypeslib.as_array(C.cast(blob, C.POINTER(C.c_uint8)), shape=(self.max_bytes,))
        r = np.ctypeslib.as_array(C.cast(raw, C.POINTER(C.c_uint8)), shape=(self.max_reads * 24,))
        return b, r

    def submit_text(self, text, fmt=FMT_FASTQ, first_index=0xFFFFFFFF, readset=0):
        b, _ = self.acquire()
        b[: len(text)] = np.frombuffer(text, np.uint8)
        _check(lib().basal_pipe_submit_text(self.h, len(text), fmt, first_index, readset), "pipe_submit_text")

    def submit_records(self, blob, raw):
        b, r = self.acquire()
        b[: len(blob)] = np.frombuffer(blob, np.uint8)
        rb = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
        r[: len(rb)] = rb
        _check(lib().basal_pipe_submit_records(self.h, len(blob), len(raw)), "pipe_submit_records")

    def submit_prepared(self, bases, descs, max_len):
        b, r = self.acquire()
        b[: len(bases)] = bases
        db = np.ascontiguousarray(descs).view(np.uint8).reshape(-1)
        r[: len(db)] = db
        _check(lib().basal_pipe_submit_prepared(self.h, len(bases), len(descs), max_len), "pipe_submit_prepared")

    def collect(self, copy=True):
        """(rc, bytes, stats): the oldest batch's output. rc != 0 (e.g. BASAL_EIO) is returned, not raised."""
        out, n, st = C.c_void_p(), C.c_uint64(), basal_batch_stats()
        rc = lib().basal_pipe_collect(self.h, C.byref(out), C.byref(n), C.byref(st))
        if rc:
            return rc, lib().basal_last_error().decode(), st
        data = C.string_at(out, n.value) if copy else (out.value, n.value)
        if copy:
            lib().basal_pipe_release(self.h)
        return 0, data, st

    def release(self):
        lib().basal_pipe_release(self.h)

    def rewind(self):
        _check(lib().basal_pipe_rewind(self.h), "pipe_rewind")

    def set_read_range(self, next_index, read_end=0xFFFFFFFF):
        _check(lib().basal_pipe_set_read_range(self.h, next_index, read_end), "pipe_set_read_range")


class Multi:
    """basal_multi_t: one core per listed GPU, reads sharded by read number, records gathered with RCCL."""

    def __init__(self, params, devices):
        self.params = params
        self.h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        _check(lib().basal_multi_create(C.byref(params.c), arr, len(devices), C.byref(self.h)), "multi_create")

    def close(self):
        if self.h:
            lib().basal_multi_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, ref, build_on_gpu=False):
        mk = C.c_uint32()
        _check(lib().basal_multi_upload(self.h, ref.h, int(build_on_gpu), C.byref(mk)), "multi_upload")
        return mk.value

    def align_batch(self, bases, reads, stream_mode=STREAM_NONE, stream_cap=0, carry=None, stales=None):
        n = len(reads)
        res = np.zeros(n, RESULT_DTYPE)
        stream = np.zeros(max(stream_cap, 1), HIT_DTYPE)
        used = C.c_uint64()
        cy = np.zeros((2, 2), np.uint8) if carry is None else np.array(carry, np.uint8).reshape(2, 2).copy()
        bases = np.ascontiguousarray(bases, np.uint8)
        reads = np.ascontiguousarray(reads)
        stales = np.zeros(0, STALE_DTYPE) if stales is None else np.ascontiguousarray(stales)
        rc = lib().basal_multi_align_batch(self.h, bases.ctypes.data, len(bases), reads.ctypes.data, n, stales.ctypes.data if len(stales) else None, len(stales),
                                           stream_mode, res.ctypes.data, stream.ctypes.data if stream_mode else None, stream_cap, C.byref(used), cy.ctypes.data)
        _check(rc, "multi_align_batch")
        return res, stream[: used.value], cy


def shard_range(n, rank, world):
    """basal_shard_range (the native sharding rule; no GPU needed)."""
    b, e = C.c_uint64(), C.c_uint64()
    lib().basal_shard_range(n, rank, world, C.byref(b), C.byref(e))
    return b.value, e.value
