"""Loader for the in-tree C-ABI library `libbamqc_gpu.so` (hand-written HIP kernels for gfx950).

There is deliberately no fallback: if the library is missing the import of the product API
fails with instructions to build it (`python -c "import __graft_entry__ as g; g.build()"` or
`make -C bamqc_amd/csrc`).
"""
import ctypes as C
import os

from . import _abi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BQC_LIB_PATH") or os.path.join(HERE, "libbamqc_gpu.so")  # (BQC_LIB_PATH: experiment builds, tools/build_inflate_variant.sh)

# every symbol include/bamqc.h declares (checked by tests/test_abi.py against the header)
_SIGNATURES = {
    "bqc_abi_version": (C.c_int, []),
    "bqc_create": (C.c_int, [C.POINTER(_abi.Options), C.POINTER(C.c_void_p)]),
    "bqc_destroy": (None, [C.c_void_p]),
    "bqc_warmup": (C.c_int, [C.c_int32]),
    "bqc_set_fasta_index": (C.c_int, [C.c_void_p, _abi.i32p]),
    "bqc_last_error": (C.c_char_p, [C.c_void_p]),
    "bqc_set_reference": (C.c_int, [C.c_void_p, C.c_int32, _abi.u8p, C.c_uint64]),
    "bqc_reserve_references": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32]),
    "bqc_submit": (C.c_int, [C.c_void_p, C.POINTER(_abi.Batch)]),
    "bqc_submit_async": (C.c_int, [C.c_void_p, C.POINTER(_abi.Batch), C.POINTER(C.c_uint64)]),
    "bqc_batch_uploaded": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int]),
    "bqc_anchor_enqueue": (C.c_int, [C.c_void_p, C.POINTER(_abi.Batch), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bqc_anchor_complete": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "bqc_submit_anchored": (C.c_int, [C.c_void_p, C.POINTER(_abi.Batch), C.c_void_p, C.POINTER(C.c_uint64)]),
    "bqc_anchor_discard": (None, [C.c_void_p, C.c_void_p]),
    "bqc_anchor_error": (C.c_char_p, [C.c_void_p]),
    "bqc_host_register": (C.c_int, [C.c_void_p, C.c_uint64]),
    "bqc_host_unregister": (C.c_int, [C.c_void_p]),
    "bqc_upload": (C.c_int, [C.c_void_p, C.POINTER(_abi.Batch), C.POINTER(C.c_void_p)]),
    "bqc_process": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bqc_dbatch_free": (None, [C.c_void_p, C.c_void_p]),
    "bqc_dbatch_bytes": (C.c_uint64, [C.c_void_p]),
    "bqc_sync": (C.c_int, [C.c_void_p]),
    "bqc_reset": (C.c_int, [C.c_void_p]),
    "bqc_flush": (C.c_int, [C.c_void_p]),
    "bqc_shard_fasta_span": (C.c_int, [C.c_void_p, _abi.i32p]),
    "bqc_shard_state_bytes": (C.c_uint64, [C.c_void_p]),
    "bqc_shard_resolve": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bqc_shard_export": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bqc_state_words": (C.c_uint64, [C.c_void_p]),
    "bqc_state_export": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bqc_state_import": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bqc_state_export_host": (C.c_int, [C.c_void_p, _abi.u64p]),
    "bqc_state_import_host": (C.c_int, [C.c_void_p, _abi.u64p]),
    "bqc_finalize": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(_abi.Counts))]),
    "bqc_last_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_char_p)),
                                  C.POINTER(C.POINTER(C.c_float))]),
    "bqc_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "bqc_write_bamqc": (C.c_int, [C.POINTER(_abi.Counts), C.POINTER(_abi.HeaderInfo), C.c_char_p]),
    # include/bamqc_host.h
    "bqc_synth_reference": (C.c_int, [C.c_uint64, C.c_int32, C.c_uint64, _abi.u8p]),
    "bqc_synth_batch": (C.c_int, [C.POINTER(_abi.SynthParams), C.POINTER(_abi.u8p), C.POINTER(C.POINTER(_abi.Batch))]),
    "bqc_synth_batch_free": (None, [C.POINTER(_abi.Batch)]),
    "bqc_synth_write": (C.c_int, [C.POINTER(_abi.SynthParams), C.POINTER(C.c_char_p), C.c_char_p, C.c_char_p, C.c_uint32]),
    "bqc_synth_slice": (C.c_int, [C.POINTER(_abi.SynthParams), C.c_uint64, C.c_uint32, C.POINTER(_abi.u8p), C.POINTER(C.POINTER(_abi.Batch))]),
    "bqc_synth_stream": (C.c_int, [C.POINTER(_abi.SynthParams), C.POINTER(C.c_char_p), C.c_char_p, C.c_char_p, C.c_uint32, C.c_int]),
    "bqc_bam_write": (C.c_int, [C.c_char_p, C.POINTER(_abi.Batch), C.c_uint32, C.POINTER(C.c_char_p), _abi.u32p, C.c_uint32, C.c_uint64, C.c_int]),
    "bqc_bam_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "bqc_bam_open_range": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]),
    "bqc_bam_open_gpu": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "bqc_bam_batches_handed_over": (C.c_uint64, [C.c_void_p]),
    "bqc_bam_open_gpu_range": (C.c_int, [C.c_char_p, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]),
    "bqc_bam_range_begin_block": (C.c_uint64, [C.c_void_p]),
    "bqc_bam_range_end_block": (C.c_uint64, [C.c_void_p]),
    "bqc_bam_range_first": (C.c_uint64, [C.c_void_p]),
    "bqc_bam_range_over": (C.c_uint64, [C.c_void_p]),
    "bqc_file_size": (C.c_uint64, [C.c_char_p]),
    "bqc_gpu_inflate_device": (None, [C.c_int]),
    "bqc_gpu_inflated_blocks": (C.c_uint64, []),
    "bqc_bam_close": (None, [C.c_void_p]),
    "bqc_bam_error": (C.c_char_p, [C.c_void_p]),
    "bqc_bam_n_refs": (C.c_uint32, [C.c_void_p]),
    "bqc_bam_ref_name": (C.c_char_p, [C.c_void_p, C.c_uint32]),
    "bqc_bam_ref_len": (C.c_uint32, [C.c_void_p, C.c_uint32]),
    "bqc_bam_sample_id": (C.c_char_p, [C.c_void_p]),
    "bqc_bam_lane_count": (C.c_uint32, [C.c_void_p]),
    "bqc_bam_n_lane_names": (C.c_uint32, [C.c_void_p]),
    "bqc_bam_lane_name": (C.c_char_p, [C.c_void_p, C.c_uint32]),
    "bqc_bam_lane_index": (C.c_uint32, [C.c_void_p, C.c_uint32]),
    "bqc_bam_set_main_chrom": (C.c_int, [C.c_void_p, _abi.u8p]),
    "bqc_bam_set_rid_filter": (C.c_int, [C.c_void_p, _abi.u8p, C.c_int]),
    "bqc_bam_next": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.POINTER(_abi.Batch))]),
    "bqc_fasta_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_char_p)),
                                 C.POINTER(C.POINTER(_abi.u8p)), C.POINTER(_abi.u64p)]),
    "bqc_fasta_free": (None, [C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(_abi.u8p), _abi.u64p]),
    "bqc_main": (C.c_int, [C.c_int, C.POINTER(C.c_char_p)]),
    "bqc_main_shard": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.c_uint32, C.c_uint32, _abi.SHARD_HOOK, C.c_void_p]),
    "bqc_main_multi": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.c_int]),
    "bqc_program_args": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_uint64]),
    "bqc_calib_read4": (C.c_int, [C.c_uint64, C.c_int]),
    "bqc_inflate_raw": (C.c_int, [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]),
    "bqc_crc32": (C.c_uint32, [C.c_char_p, C.c_uint64]),
}

_LIB = None


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "bamqc_amd: %s is missing. Build the HIP extension first: `make -C bamqc_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        f = getattr(lib, name)  # AttributeError here = the library does not export the ABI
        f.restype = res
        f.argtypes = args
    _LIB = lib
    return lib
