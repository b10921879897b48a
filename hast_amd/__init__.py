"""hast_amd -- Python-side handle on the MI355X-native HAST stage-01 classifier.

The product is native: `libhast.so` (HIP kernels for gfx950 + C ABI, include/hast.h) and the drop-in
`classify` executable, both built in-tree by hast_amd/csrc/Makefile.  This package is only the
ctypes view of that C ABI used by bench.py and the tests; there is no Python or CPU compute path --
every call needs the shared library, and every compute call needs a GPU.
"""
from .binding import (HastError, Context, SynthParams, lib, lib_path, classify_exe, classify_read_exe, build,  # noqa: F401
                      parse_barcode, get_hap, canon_kmer, chop_read, synth_keys_host, synth_reads_host,
                      ABI_SYMBOLS, B_ALG_PER_READ, KmerCounter, KcSynth, KC_HISTO_HIGH, kc_find_bounds, kc_synth_host,
                      unshared_kmers_exe, GzReader, GzStats)
