"""ctypes binding of libhast.so (include/hast.h).  Loads the in-tree build; fails loudly if absent."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p, u32p, u64p, vp = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_void_p

STATUS = {0: "OK", 1: "INVALID", 2: "NO_DEVICE", 3: "HIP", 4: "OOM", 5: "TABLE_FULL", 6: "FORMAT", 7: "RCCL", 8: "IO", 9: "UNSUPPORTED"}


class SynthParams(C.Structure):
    _fields_ = [("seed_k", C.c_uint64), ("seed_r", C.c_uint64), ("seed_b", C.c_uint64),
                ("n_keys_per_hap", C.c_uint64), ("n_barcodes", C.c_uint32), ("read_len", C.c_uint32),
                ("k", C.c_uint32), ("reserved", C.c_uint32)]


class KcSynth(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("genome_len", C.c_uint64), ("read_len", C.c_uint32), ("snp_per_1024", C.c_uint32),
                ("err_per_4096", C.c_uint32), ("n_per_4096", C.c_uint32)]


class FqBlock(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_bases", C.c_uint64), ("max_read_len", C.c_uint32), ("short_read", C.c_uint32),
                ("bytes", C.POINTER(C.c_uint8)), ("bc_pos", C.POINTER(C.c_uint32)), ("bc_len", C.POINTER(C.c_uint32)),
                ("bc_text", C.POINTER(C.c_uint8)), ("ids", C.POINTER(C.c_uint32)), ("unknown", C.POINTER(C.c_uint32)),
                ("n_unknown", C.c_uint64), ("dict_ids", C.c_uint64)]


class FqRouted(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("count", C.c_uint64 * 4), ("run", C.POINTER(C.c_uint8) * 4), ("run_bytes", C.c_uint64 * 4),
                ("host_block", C.c_int), ("bytes", C.POINTER(C.c_uint8)), ("rec_start", C.POINTER(C.c_uint32)), ("rec_len", C.POINTER(C.c_uint32)),
                ("rec_class", C.POINTER(C.c_uint8)), ("n_slots", C.c_uint64), ("tail", C.POINTER(C.c_uint8)), ("tail_bytes", C.c_uint64)]


class GzStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("compressed_bytes", "out_bytes", "chunks", "accepted", "followup_jobs", "followup_rounds", "followup_accepted", "members")] + \
               [(n, C.c_double) for n in ("decode_s", "windows_crc_s", "wait_upload_s", "wait_consumer_s", "wait_decode_s", "open_s")] + \
               [(n, C.c_uint64) for n in ("ring_bytes", "upload_waited_for_ring")] + [("chain_walk_s", C.c_double), ("ring_laps", C.c_uint64)]


KC_HISTO_HIGH = 10000

# every symbol include/hast.h declares: name -> (restype, argtypes)
ABI_SYMBOLS = {
    "hast_version": (C.c_char_p, []),
    "hast_last_error": (C.c_char_p, []),
    "hast_ctx_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(vp)]),
    "hast_ctx_destroy": (None, [vp]),
    "hast_release_parked": (None, []),
    "hast_ctx_k": (C.c_int, [vp]),
    "hast_ctx_minimizer": (C.c_int, [vp]),
    "hast_ctx_set_minimizer": (C.c_int, [vp, C.c_int]),
    "hast_ctx_device": (C.c_int, [vp]),
    "hast_ctx_set_option": (C.c_int, [vp, C.c_char_p, C.c_long]),
    "hast_ctx_options": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    "hast_ctx_stream": (vp, [vp]),
    "hast_stream_sync": (C.c_int, [vp, vp]),
    "hast_dev_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "hast_dev_free": (C.c_int, [vp, vp]),
    "hast_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "hast_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "hast_memset_d": (C.c_int, [vp, vp, C.c_int, C.c_size_t, vp]),
    "hast_table_reserve": (C.c_int, [vp, C.c_uint64, C.c_double]),
    "hast_table_insert_text": (C.c_int, [vp, C.c_int, C.c_char_p, C.c_size_t, u64p]),
    "hast_table_insert_text_file": (C.c_int, [vp, C.c_int, C.c_char_p, u64p]),
    "hast_ctx_set_text_check": (C.c_int, [vp, C.c_int]),
    "hast_table_insert_keys": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
    "hast_table_insert_keys_device": (C.c_int, [vp, C.c_int, vp, C.c_size_t, vp]),
    "hast_table_erase": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "hast_table_sizes": (C.c_int, [vp, u64p, u64p]),
    "hast_table_lookup": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "hast_table_save": (C.c_int, [vp, C.c_char_p]),
    "hast_table_load": (C.c_int, [vp, C.c_char_p, C.c_double]),
    "hast_table_file_info": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), u64p]),
    "hast_table_info": (C.c_int, [vp, u64p, u64p]),
    "hast_table_clone": (C.c_int, [vp, vp]),
    "hast_ctx_set_filter": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "hast_filter_build": (C.c_int, [vp]),
    "hast_filter_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), u64p]),
    "hast_filter_request_ceiling": (C.c_int, [vp, C.POINTER(C.c_double)]),
    "hast_counts_resize": (C.c_int, [vp, C.c_size_t]),
    "hast_counts_bind": (C.c_int, [vp, vp, C.c_size_t]),
    "hast_counts_zero": (C.c_int, [vp, vp]),
    "hast_counts_read": (C.c_int, [vp, vp, vp, vp, C.c_size_t]),
    "hast_counts_pack": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "hast_counts_unpack": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "hast_counts_add_votes": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_uint32, vp]),
    "hast_counts_allreduce": (C.c_int, [C.POINTER(vp), C.c_int]),
    "hast_classify_timing": (C.c_int, [vp, C.c_int]),
    "hast_classify_times": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]),
    "hast_classify_device": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_uint32, vp, vp, C.c_size_t, vp]),
    "hast_classify_batch": (C.c_int, [vp, vp, vp, vp, C.c_size_t, C.c_uint32]),
    "hast_classify_perread_device": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t, vp, vp]),
    "hast_classify_perread": (C.c_int, [vp, vp, vp, C.c_size_t, vp]),
    "hast_batch_begin": (C.c_int, [vp, C.c_size_t, C.c_size_t, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
    "hast_batch_submit": (C.c_int, [vp, C.c_size_t, C.c_uint32]),
    "hast_names_create": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "hast_names_destroy": (None, [vp]),
    "hast_fq_create": (C.c_int, [vp, C.c_size_t, C.c_int, vp, C.POINTER(vp)]),
    "hast_fq_create_ex": (C.c_int, [vp, C.c_size_t, C.c_int, vp, C.c_int, C.POINTER(vp)]),
    "hast_fq_create_striped": (C.c_int, [C.POINTER(vp), C.c_int, C.c_size_t, C.c_int, C.POINTER(vp), C.POINTER(vp)]),
    "hast_fq_create_striped_ex": (C.c_int, [C.POINTER(vp), C.c_int, C.c_size_t, C.c_int, C.POINTER(vp), C.c_int, C.POINTER(vp)]),
    "hast_fq_lanes": (C.c_int, [vp]),
    "hast_fq_lane_records": (C.c_uint64, [vp, C.c_int]),
    "hast_fq_destroy": (None, [vp]),
    "hast_fq_block_bytes": (C.c_size_t, [vp]),
    "hast_fq_acquire": (C.c_int, [vp, C.POINTER(C.POINTER(C.c_uint8))]),
    "hast_fq_submit": (C.c_int, [vp, C.c_size_t, C.c_int]),
    "hast_fq_device_block": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp)]),
    "hast_fq_submit_device": (C.c_int, [vp, C.c_size_t, C.c_int]),
    "hast_fq_block_host_bytes": (C.c_int, [vp, C.POINTER(C.POINTER(C.c_uint8))]),
    "hast_fq_poll": (C.c_int, [vp]),
    "hast_fq_next": (C.c_int, [vp, C.POINTER(FqBlock)]),
    "hast_fq_commit": (C.c_int, [vp]),
    "hast_dev_mem_info": (C.c_int, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "hast_names_create_dict": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "hast_names_limit": (C.c_size_t, [vp]),
    "hast_names_count": (C.c_int, [vp, C.POINTER(C.c_size_t)]),
    "hast_names_texts": (C.c_int, [vp, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint8)]),
    "hast_counts_read_range": (C.c_int, [vp, C.c_size_t, C.c_size_t, u64p, u64p, u64p]),
    "hast_names_merge": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint32)]),
    "hast_counts_permute": (C.c_int, [vp, C.POINTER(C.c_uint32), C.c_size_t, C.c_size_t]),
    "hast_names_insert": (C.c_int, [vp, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.c_size_t]),
    "hast_fq_set_route": (C.c_int, [vp, C.POINTER(vp), C.c_int]),
    "hast_fq_next_routed": (C.c_int, [vp, C.POINTER(FqRouted)]),
    "hast_gz_open": (C.c_int, [vp, C.c_char_p, C.POINTER(vp)]),
    "hast_gz_open_ex": (C.c_int, [vp, C.c_char_p, C.c_size_t, C.c_size_t, C.c_double, C.POINTER(vp)]),
    "hast_gz_open_multi": (C.c_int, [C.POINTER(vp), C.c_int, C.c_char_p, C.POINTER(vp)]),
    "hast_gz_open_multi_ex": (C.c_int, [C.POINTER(vp), C.c_int, C.c_char_p, C.c_size_t, C.c_size_t, C.c_double, C.POINTER(vp)]),
    "hast_gz_units": (C.c_int, [vp]),
    "hast_gz_read_device": (C.c_int, [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), vp]),
    "hast_gz_get_stats": (C.c_int, [vp, C.POINTER(GzStats)]),
    "hast_gz_close": (None, [vp]),
    "hast_parse_barcode": (None, [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "hast_get_hap": (C.c_int, [C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.c_double]),
    "hast_canon_kmer": (C.c_uint64, [C.c_char_p, C.c_int]),
    "hast_chop_read": (C.c_size_t, [C.c_char_p, C.c_size_t, C.c_int, u64p]),
    "hast_kmer_to_str": (None, [C.c_uint64, C.c_int, C.c_char_p]),
    "hast_synth_keys_host": (C.c_int, [C.POINTER(SynthParams), C.c_int, C.c_uint64, C.c_size_t, vp]),
    "hast_synth_reads_host": (C.c_int, [C.POINTER(SynthParams), C.c_uint64, C.c_size_t, vp, vp]),
    "hast_synth_keys_device": (C.c_int, [vp, C.POINTER(SynthParams), C.c_int, C.c_uint64, C.c_size_t, vp, vp]),
    "hast_synth_reads_device": (C.c_int, [vp, C.POINTER(SynthParams), C.c_uint64, C.c_size_t, vp, vp, vp]),
    "hast_synth_table_build": (C.c_int, [vp, C.POINTER(SynthParams)]),
    # stage 00: parent-unique k-mer sets
    "hast_kc_create": (C.c_int, [C.c_int, C.c_int, C.c_size_t, C.POINTER(vp)]),
    "hast_kc_create_ex": (C.c_int, [C.c_int, C.c_int, C.c_size_t, C.c_uint64, C.POINTER(vp)]),
    "hast_kc_destroy": (None, [vp]),
    "hast_kc_stream": (vp, [vp]),
    "hast_kc_set_slice": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
    "hast_kc_count_device": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
    "hast_kc_count": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
    "hast_kc_sync": (C.c_int, [vp]),
    "hast_kc_partition_info": (C.c_int, [vp, u64p]),
    "hast_kc_stats": (C.c_int, [vp, u64p]),
    "hast_kc_histo": (C.c_int, [vp, C.c_int, vp]),
    "hast_kc_find_bounds": (None, [vp, C.POINTER(C.c_long)]),
    "hast_kc_select": (C.c_int, [vp, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_size_t)]),
    "hast_kc_selection_clear": (C.c_int, [vp]),
    "hast_kc_selection_adopt": (C.c_int, [vp, vp]),
    "hast_kc_release_table": (C.c_int, [vp]),
    "hast_kc_selection_sort": (C.c_int, [vp, C.c_int, C.POINTER(C.c_size_t)]),
    "hast_kc_selection_text": (C.c_int, [vp, C.c_int, C.c_size_t, C.c_size_t, vp]),
    "hast_kc_selection_keys": (C.c_int, [vp, C.c_int, C.c_size_t, C.c_size_t, vp]),
    "hast_kc_synth_host": (C.c_int, [C.POINTER(KcSynth), C.c_int, C.c_uint64, C.c_size_t, vp]),
    "hast_kc_synth_device": (C.c_int, [vp, C.POINTER(KcSynth), C.c_int, C.c_uint64, C.c_size_t, vp]),
}


def B_ALG_PER_READ(read_len: int, k: int) -> int:
    """Algorithmic HBM bytes per read: every base once + one 64-B line per window."""
    return read_len + max(0, read_len - k + 1) * 64


class HastError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("hast: %s: %s" % (STATUS.get(status, status), msg))
        self.status = status


def lib_path():
    return os.environ.get("HAST_LIB") or os.path.join(_HERE, "libhast.so")


def classify_exe():
    return os.path.join(_HERE, "classify")


def classify_read_exe():
    return os.path.join(_HERE, "classify_read")


def unshared_kmers_exe():
    return os.path.join(_HERE, "unshared_kmers")


def build(verbose=False):
    """Compile libhast.so + classify for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    res = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if verbose or res.returncode:
        print(res.stdout.decode())
    if res.returncode:
        raise RuntimeError("building libhast.so failed")


def lib():
    """The loaded library.  No fallback: a missing libhast.so is an error."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(or make -C hast_amd/csrc); there is no Python/CPU fallback" % p)
        l = C.CDLL(p)
        for name, (res, args) in ABI_SYMBOLS.items():
            f = getattr(l, name)
            f.restype, f.argtypes = res, args
        _LIB = l
    return _LIB


def _ck(st):
    if st != 0:
        raise HastError(st, lib().hast_last_error().decode())


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return C.c_void_p(a.ctypes.data)
    return a


# ---- host-side pieces --------------------------------------------------------------------------
def parse_barcode(head: bytes) -> bytes:
    s, n = C.c_size_t(), C.c_size_t()
    lib().hast_parse_barcode(head, len(head), C.byref(s), C.byref(n))
    return head[s.value:s.value + n.value]


def get_hap(barcode: bytes, c0, c1, n0, n1, w0=1.0, w1=1.0) -> int:
    return lib().hast_get_hap(barcode, len(barcode), c0, c1, n0, n1, w0, w1)


def canon_kmer(s: bytes) -> int:
    return lib().hast_canon_kmer(s, len(s))


def chop_read(seq: bytes, k: int):
    n = max(0, len(seq) - k + 1)
    out = (C.c_uint64 * max(1, n))()
    got = lib().hast_chop_read(seq, len(seq), k, out)
    return [out[i] for i in range(got)]


def make_params(k, read_len=150, n_keys_per_hap=0, n_barcodes=1, seed_k=0, seed_r=0, seed_b=0, clustered=False, no_plants=False):
    return SynthParams(seed_k, seed_r, seed_b, n_keys_per_hap, n_barcodes, read_len, k, (1 if clustered else 0) | (2 if no_plants else 0))


def synth_keys_host(p: SynthParams, hap, first, n):
    out = np.empty(n, dtype=np.uint64)
    _ck(lib().hast_synth_keys_host(C.byref(p), hap, first, n, _ptr(out)))
    return out


def synth_reads_host(p: SynthParams, first, n):
    bases = np.empty(n * p.read_len, dtype=np.uint8)
    ids = np.empty(n, dtype=np.uint32)
    _ck(lib().hast_synth_reads_host(C.byref(p), first, n, _ptr(bases), _ptr(ids)))
    return bases, ids


# ---- device context ----------------------------------------------------------------------------
class Context:
    """One GPU context (hast_ctx).  Raises HastError(NO_DEVICE) when there is no GPU."""

    def __init__(self, k, device=0, minimizer=None):
        self._h = C.c_void_p()
        self._lib = lib()
        _ck(self._lib.hast_ctx_create(device, k, C.byref(self._h)))
        self.k = k
        self.device = device
        if minimizer is not None:
            _ck(self._lib.hast_ctx_set_minimizer(self._h, minimizer))

    @property
    def minimizer(self):
        return self._lib.hast_ctx_minimizer(self._h)

    def set_filter(self, enable=True, m=0, t=0, kp=0):
        """enable: False/0 = probe the table directly, True/1 = filter (exact entries where they fit), 2 = filter with prints always"""
        _ck(self._lib.hast_ctx_set_filter(self._h, int(enable), m, t, kp))

    def filter_build(self):
        _ck(self._lib.hast_filter_build(self._h))

    def filter_request_ceiling(self):
        """requests/s at which this GPU serves random 128-B blocks of this context's filter (measurement entry)"""
        r = C.c_double()
        _ck(self._lib.hast_filter_request_ceiling(self._h, C.byref(r)))
        return r.value

    def filter_info(self):
        """(enabled, m, t, kp, bytes); m = t = kp = bytes = 0 until the filter has been built for the current table"""
        en, m, t, kp, b = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_uint64()
        _ck(self._lib.hast_filter_info(self._h, C.byref(en), C.byref(m), C.byref(t), C.byref(kp), C.byref(b)))
        return bool(en.value), m.value, t.value, kp.value, b.value

    def filter_mode(self):
        """0 = off, 1 = 16-bit prints, 2 = exact entries (hast_common.h); known once the filter has been built"""
        en = C.c_int()
        _ck(self._lib.hast_filter_info(self._h, C.byref(en), None, None, None, None))
        return en.value

    def close(self):
        if self._h:
            self._lib.hast_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return self._lib.hast_ctx_stream(self._h)

    def sync(self, stream=None):
        _ck(self._lib.hast_stream_sync(self._h, stream))

    # raw device memory
    def alloc(self, nbytes) -> int:
        p = C.c_void_p()
        _ck(self._lib.hast_dev_alloc(self._h, nbytes, C.byref(p)))
        return p.value

    def free(self, dptr):
        _ck(self._lib.hast_dev_free(self._h, C.c_void_p(dptr)))

    def to_device(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        d = self.alloc(max(arr.nbytes, 1))
        _ck(self._lib.hast_memcpy_h2d(self._h, C.c_void_p(d), _ptr(arr), arr.nbytes))
        return d

    def to_host(self, dptr, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        _ck(self._lib.hast_memcpy_d2h(self._h, _ptr(out), C.c_void_p(dptr), out.nbytes))
        return out

    def memset(self, dptr, byte, nbytes, stream=None):
        _ck(self._lib.hast_memset_d(self._h, C.c_void_p(dptr), byte, nbytes, stream))

    # table
    def table_reserve(self, max_keys, load_factor=0.0):
        _ck(self._lib.hast_table_reserve(self._h, max_keys, load_factor))

    def table_insert_text(self, hap, text: bytes) -> int:
        lines = C.c_uint64()
        _ck(self._lib.hast_table_insert_text(self._h, hap, text, len(text), C.byref(lines)))
        return lines.value

    def set_text_check(self, acgt_only=True):
        _ck(self._lib.hast_ctx_set_text_check(self._h, 1 if acgt_only else 0))

    def table_insert_text_file(self, hap, path) -> int:
        lines = C.c_uint64()
        _ck(self._lib.hast_table_insert_text_file(self._h, hap, os.fsencode(path), C.byref(lines)))
        return lines.value

    def table_insert_keys(self, hap, keys: np.ndarray):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        _ck(self._lib.hast_table_insert_keys(self._h, hap, _ptr(keys), keys.size))

    def table_insert_keys_device(self, hap, d_keys, n, stream=None):
        _ck(self._lib.hast_table_insert_keys_device(self._h, hap, C.c_void_p(d_keys), n, stream))

    def table_erase(self, keys) -> np.ndarray:
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        hit = np.zeros(keys.size, dtype=np.uint8)
        _ck(self._lib.hast_table_erase(self._h, _ptr(keys), keys.size, _ptr(hit)))
        return hit

    def table_sizes(self):
        a, b = C.c_uint64(), C.c_uint64()
        _ck(self._lib.hast_table_sizes(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def table_lookup(self, keys) -> np.ndarray:
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        tags = np.zeros(keys.size, dtype=np.uint8)
        _ck(self._lib.hast_table_lookup(self._h, _ptr(keys), keys.size, _ptr(tags)))
        return tags

    def table_save(self, path):
        _ck(self._lib.hast_table_save(self._h, os.fsencode(path)))

    def table_load(self, path, load_factor=0.0):
        _ck(self._lib.hast_table_load(self._h, os.fsencode(path), load_factor))

    def table_clone_from(self, src):
        _ck(self._lib.hast_table_clone(self._h, src._h))

    def table_info(self):
        a, b = C.c_uint64(), C.c_uint64()
        _ck(self._lib.hast_table_info(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # counters
    def counts_resize(self, n):
        _ck(self._lib.hast_counts_resize(self._h, n))

    def counts_bind(self, d_counts, n):
        _ck(self._lib.hast_counts_bind(self._h, C.c_void_p(d_counts), n))

    def counts_zero(self, stream=None):
        _ck(self._lib.hast_counts_zero(self._h, stream))

    def counts_read(self, n):
        """(c0, c1, neg) as uint64 arrays: the device counts in 64-bit words (include/hast.h)."""
        c0 = np.zeros(n, dtype=np.uint64)
        c1 = np.zeros(n, dtype=np.uint64)
        neg = np.zeros(n, dtype=np.uint64)
        _ck(self._lib.hast_counts_read(self._h, _ptr(c0), _ptr(c1), _ptr(neg), n))
        return c0, c1, neg

    def counts_pack(self, d_packed, n, stream=None):
        """counts[n][4] -> d_packed = c0[n] | c1[n] | neg[n] (what a caller's own collective moves)"""
        _ck(self._lib.hast_counts_pack(self._h, C.c_void_p(d_packed), n, stream))

    def counts_unpack(self, d_packed, n, stream=None):
        _ck(self._lib.hast_counts_unpack(self._h, C.c_void_p(d_packed), n, stream))

    def counts_add_votes(self, d_votes, d_barcode_ids, n_reads, max_votes, stream=None):
        _ck(self._lib.hast_counts_add_votes(self._h, C.c_void_p(d_votes), C.c_void_p(d_barcode_ids), n_reads, max_votes, stream))

    def set_option(self, name, value):
        _ck(self._lib.hast_ctx_set_option(self._h, name.encode(), int(value)))

    def options(self):
        buf = C.create_string_buffer(512)
        _ck(self._lib.hast_ctx_options(self._h, buf, len(buf)))
        return buf.value.decode()

    # classify
    def classify_device(self, d_bases, bases_bytes, n_reads, read_len, d_offsets=None, d_barcode_ids=None,
                        d_votes=None, stream=None):
        _ck(self._lib.hast_classify_device(self._h, C.c_void_p(d_bases), bases_bytes,
                                           C.c_void_p(d_offsets) if d_offsets else None, read_len,
                                           C.c_void_p(d_barcode_ids) if d_barcode_ids else None,
                                           C.c_void_p(d_votes) if d_votes else None, n_reads, stream))

    def classify_timing(self, n_slots):
        _ck(self._lib.hast_classify_timing(self._h, n_slots))

    def classify_times(self, max_n=4096):
        a, b, n = (C.c_float * max_n)(), (C.c_float * max_n)(), C.c_int()
        _ck(self._lib.hast_classify_times(self._h, a, b, max_n, C.byref(n)))
        return list(a[:n.value]), list(b[:n.value])

    def classify_batch(self, bases: np.ndarray, offsets: np.ndarray, ids: np.ndarray, max_read_len):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        _ck(self._lib.hast_classify_batch(self._h, _ptr(bases), _ptr(offsets), _ptr(ids), ids.size, max_read_len))

    def classify_perread(self, bases: np.ndarray, offsets: np.ndarray) -> np.ndarray:
        """per-read (hits0, hits1), stage-03 string semantics, any read length"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        votes = np.zeros((offsets.size - 1, 2), dtype=np.uint32)
        _ck(self._lib.hast_classify_perread(self._h, _ptr(bases), _ptr(offsets), offsets.size - 1, _ptr(votes)))
        return votes

    def classify_perread_device(self, d_bases, bases_bytes, d_offsets, n_reads, d_votes, stream=None):
        _ck(self._lib.hast_classify_perread_device(self._h, C.c_void_p(d_bases), bases_bytes, C.c_void_p(d_offsets),
                                                   n_reads, C.c_void_p(d_votes), stream))

    # synthetic
    def synth_keys_device(self, p, hap, first, n, d_out, stream=None):
        _ck(self._lib.hast_synth_keys_device(self._h, C.byref(p), hap, first, n, C.c_void_p(d_out), stream))

    def synth_reads_device(self, p, first, n, d_bases, d_ids, stream=None):
        _ck(self._lib.hast_synth_reads_device(self._h, C.byref(p), first, n, C.c_void_p(d_bases),
                                              C.c_void_p(d_ids) if d_ids else None, stream))

    def synth_table_build(self, p):
        _ck(self._lib.hast_synth_table_build(self._h, C.byref(p)))


# ---- stage 00: parent-unique k-mer sets ---------------------------------------------------------------------
def kc_find_bounds(histo: np.ndarray):
    """(MIN_INDEX, MAX_INDEX, LOWER_INDEX, UPPER_INDEX) of find_bounds.awk; host arithmetic only"""
    histo = np.ascontiguousarray(histo, dtype=np.uint64)
    assert histo.size == KC_HISTO_HIGH + 2
    out = (C.c_long * 4)()
    lib().hast_kc_find_bounds(_ptr(histo), out)
    return tuple(out)


def kc_synth_host(p: KcSynth, parent, first, n_reads):
    out = np.empty(n_reads * (p.read_len + 1), dtype=np.uint8)
    _ck(lib().hast_kc_synth_host(C.byref(p), parent, first, n_reads, _ptr(out)))
    return out


class GzReader:
    """hast_gz: one .gz file inflated on the GPU (include/hast.h); read() returns the next bytes as a numpy array (test view:
    the product hands the bytes to the FASTQ framer on the device)."""

    def __init__(self, ctx, path, chunk_bytes=0, chunks_per_pass=0, room=0.0, ctxs=None):
        """ctxs: several contexts -> hast_gz_open_multi_ex (the passes of the one stream go to their GPUs in turn); reads land on ctx"""
        self._lib = lib()
        self._ctx = ctx
        h = C.c_void_p()
        if ctxs:
            arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
            _ck(self._lib.hast_gz_open_multi_ex(arr, len(ctxs), os.fsencode(path), chunk_bytes, chunks_per_pass, room, C.byref(h)))
        else:
            _ck(self._lib.hast_gz_open_ex(ctx._h, os.fsencode(path), chunk_bytes, chunks_per_pass, room, C.byref(h)))
        self._h = h
        self._d = None
        self._cap = 0

    def read_device(self, d_dst, cap, stream=None):
        n = C.c_size_t()
        _ck(self._lib.hast_gz_read_device(self._h, C.c_void_p(d_dst), cap, C.byref(n), stream))
        return n.value

    def read(self, cap):
        if self._cap < cap:
            if self._d:
                self._ctx.free(self._d)
            self._d = self._ctx.alloc(cap + 16)
            self._cap = cap
        n = self.read_device(self._d, cap)
        self._ctx.sync()
        return self._ctx.to_host(self._d, (n,), np.uint8) if n else np.zeros(0, np.uint8)

    def read_all(self, piece=1 << 22):
        parts = []
        while True:
            a = self.read(piece)
            if a.size == 0:
                break
            parts.append(a)
        return np.concatenate(parts).tobytes() if parts else b""

    def units(self):
        return self._lib.hast_gz_units(self._h)

    def stats(self):
        st = GzStats()
        _ck(self._lib.hast_gz_get_stats(self._h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in GzStats._fields_}

    def close(self):
        if self._h:
            self._lib.hast_gz_close(self._h)
            self._h = None
        if self._d:
            self._ctx.free(self._d)
            self._d = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class KmerCounter:
    """One count table of both parents' canonical k-mers in HBM (hast_kc_*)."""

    def __init__(self, k, table_bytes=0, device=0):
        self._lib = lib()
        h = C.c_void_p()
        _ck(self._lib.hast_kc_create(device, k, table_bytes, C.byref(h)))
        self._h = h
        self.k = k

    def close(self):
        if self._h:
            self._lib.hast_kc_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_slice(self, slice_, n_slices):
        _ck(self._lib.hast_kc_set_slice(self._h, slice_, n_slices))

    def count(self, parent, data: np.ndarray):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        _ck(self._lib.hast_kc_count(self._h, parent, _ptr(data), data.size))

    def count_device(self, parent, d_bytes, n_bytes):
        _ck(self._lib.hast_kc_count_device(self._h, parent, C.c_void_p(d_bytes), n_bytes))

    def sync(self):
        _ck(self._lib.hast_kc_sync(self._h))

    def partition_info(self):
        out = np.zeros(5, dtype=np.uint64)
        _ck(self._lib.hast_kc_partition_info(self._h, out.ctypes.data_as(u64p)))
        return dict(partitioned=bool(out[0]), flushes=int(out[1]), records=int(out[2]), spilled_windows=int(out[3]), record_capacity=int(out[4]))

    def stats(self):
        out = np.zeros(6, dtype=np.uint64)
        _ck(self._lib.hast_kc_stats(self._h, out.ctypes.data_as(u64p)))
        return dict(distinct=(int(out[0]), int(out[1])), keys=int(out[2]), capacity=int(out[3]), total=(int(out[4]), int(out[5])))

    def histo(self, parent, into=None):
        h = np.zeros(KC_HISTO_HIGH + 2, dtype=np.uint64) if into is None else into
        _ck(self._lib.hast_kc_histo(self._h, parent, _ptr(h)))
        return h

    def select(self, parent, lower, upper):
        n = C.c_size_t()
        _ck(self._lib.hast_kc_select(self._h, parent, lower, upper, C.byref(n)))
        return n.value

    def selection_clear(self):
        _ck(self._lib.hast_kc_selection_clear(self._h))

    def release_table(self):
        _ck(self._lib.hast_kc_release_table(self._h))

    def selection_sort(self, parent):
        n = C.c_size_t()
        _ck(self._lib.hast_kc_selection_sort(self._h, parent, C.byref(n)))
        return n.value

    def selection_text(self, parent, first, count) -> bytes:
        out = np.empty(count * (self.k + 1), dtype=np.uint8)
        _ck(self._lib.hast_kc_selection_text(self._h, parent, first, count, _ptr(out)))
        return out.tobytes()

    def selection_keys(self, parent, first, count):
        out = np.empty(count, dtype=np.uint64)
        _ck(self._lib.hast_kc_selection_keys(self._h, parent, first, count, _ptr(out)))
        return out

    def synth_device(self, p: KcSynth, parent, first, n_reads, d_out):
        _ck(self._lib.hast_kc_synth_device(self._h, C.byref(p), parent, first, n_reads, C.c_void_p(d_out)))
