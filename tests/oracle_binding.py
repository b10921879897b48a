"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE: the CPU restatement of the
reference).  Only tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C

u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)


def load(path):
    lib = C.CDLL(path)
    sig = {
        "ho_base2int": (C.c_uint8, [C.c_uint8]),
        "ho_mask": (C.c_uint64, [C.c_int]),
        "ho_pack": (C.c_uint64, [C.c_char_p, C.c_int]),
        "ho_revcomp": (C.c_uint64, [C.c_uint64, C.c_int]),
        "ho_canon_str": (C.c_uint64, [C.c_char_p, C.c_int]),
        "ho_chop_read": (C.c_size_t, [C.c_char_p, C.c_size_t, C.c_int, u64p]),
        "ho_kmer_to_str": (None, [C.c_uint64, C.c_int, C.c_char_p]),
        "ho_parse_name": (None, [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
        "ho_get_hap": (C.c_int, [C.c_char_p, C.c_size_t, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64,
                                 C.c_double, C.c_double]),
        "ho_new": (C.c_void_p, []),
        "ho_free": (None, [C.c_void_p]),
        "ho_set_weights": (None, [C.c_void_p, C.c_double, C.c_double]),
        "ho_k": (C.c_int, [C.c_void_p]),
        "ho_set_size": (C.c_uint64, [C.c_void_p, C.c_int]),
        "ho_lines_loaded": (C.c_uint64, [C.c_void_p, C.c_int]),
        "ho_load_kmers_file": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
        "ho_load_kmers_text": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int]),
        "ho_load_keys": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]),
        "ho_load_keys_mt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int]),
        "ho_contains": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64]),
        "ho_init_adaptor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p]),
        "ho_process_read": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
        "ho_process_fastq": (C.c_int, [C.c_void_p, C.c_char_p]),
        "ho_n_barcodes": (C.c_size_t, [C.c_void_p]),
        "ho_read_votes": (None, [C.c_void_p, C.c_char_p, C.c_size_t, u32p, u32p, C.POINTER(C.c_int)]),
        "ho_classify_ids": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
        "ho_classify_ids_votes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    }
    sig.update({
        "ho_s03_new": (C.c_void_p, []),
        "ho_s03_free": (None, [C.c_void_p]),
        "ho_s03_load_text": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int]),
        "ho_s03_load_file": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
        "ho_s03_k": (C.c_int, [C.c_void_p]),
        "ho_s03_lines": (C.c_uint64, [C.c_void_p, C.c_int]),
        "ho_s03_read_hits": (None, [C.c_void_p, C.c_char_p, C.c_size_t, u32p, u32p]),
        "ho_s03_format_row": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_char_p]),
    })
    lp = C.POINTER(C.c_long)
    sig.update({          # stage 00 (oracle/s00_oracle.h)
        "ho_s00_new": (C.c_void_p, [C.c_int]),
        "ho_s00_free": (None, [C.c_void_p]),
        "ho_s00_k": (C.c_int, [C.c_void_p]),
        "ho_s00_canon_str": (C.c_uint64, [C.c_char_p, C.c_int]),
        "ho_s00_key_to_str": (None, [C.c_uint64, C.c_int, C.c_char_p]),
        "ho_s00_add_seq": (None, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]),
        "ho_s00_add_stream": (None, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
        "ho_s00_add_files": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int]),
        "ho_s00_distinct": (C.c_uint64, [C.c_void_p, C.c_int]),
        "ho_s00_total": (C.c_uint64, [C.c_void_p, C.c_int]),
        "ho_s00_count": (C.c_uint32, [C.c_void_p, C.c_int, C.c_uint64]),
        "ho_s00_histo": (None, [C.c_void_p, C.c_int, u64p]),
        "ho_s00_find_bounds": (None, [u64p, lp, lp, lp, lp]),
        "ho_s00_select": (C.c_size_t, [C.c_void_p, C.c_int, C.c_long, C.c_long, u64p]),
    })
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    return lib


def parse_name(lib, head: bytes) -> bytes:
    s, n = C.c_size_t(), C.c_size_t()
    lib.ho_parse_name(head, len(head), C.byref(s), C.byref(n))
    return head[s.value:s.value + n.value]


def chop(lib, seq: bytes, k: int):
    n = max(0, len(seq) - k + 1)
    out = (C.c_uint64 * max(n, 1))()
    got = lib.ho_chop_read(seq, len(seq), k, out)
    return [out[i] for i in range(got)]
