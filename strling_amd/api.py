"""Python host mirror of the C ABI (include/strling_amd.h) via ctypes.

This is plumbing for tests, bench.py and scripting: every compute call goes through
libstrling_amd.so into the HIP kernels.  There is no CPU fallback -- if the library is missing or
no HIP device is present the calls raise.
"""
import ctypes as C
import os
import numpy as np

from . import build as _build
from .records import RecordBatch, GenomeStr

_LIB = None


class StrlingError(RuntimeError):
    pass


class Opts(C.Structure):
    _fields_ = [("median_fragment_length", C.c_int32), ("proportion_repeat", C.c_double), ("min_mapq", C.c_uint8)]


class CGenomeStr(C.Structure):
    _fields_ = [("n_tid", C.c_int32), ("has_chrom", C.c_void_p), ("iv_off", C.c_void_p), ("iv_start", C.c_void_p),
                ("iv_stop", C.c_void_p)]


class CRecords(C.Structure):
    _fields_ = [("n", C.c_int64), ("tid", C.c_void_p), ("pos", C.c_void_p), ("mtid", C.c_void_p), ("mpos", C.c_void_p),
                ("flag", C.c_void_p), ("mapq", C.c_void_p), ("cigar_off", C.c_void_p), ("cigar", C.c_void_p),
                ("seq_off", C.c_void_p), ("l_seq", C.c_void_p), ("seq4", C.c_void_p), ("qname_off", C.c_void_p),
                ("qnames", C.c_void_p)]


class CReadSoa(C.Structure):
    _fields_ = [("n", C.c_uint64), ("tid", C.c_void_p), ("pos", C.c_void_p), ("end", C.c_void_p), ("seq_off", C.c_void_p),
                ("l_seq", C.c_void_p), ("clip_l", C.c_void_p), ("clip_r", C.c_void_p), ("mapq", C.c_void_p),
                ("cig", C.c_void_p), ("seq4", C.c_void_p), ("seq4_bytes", C.c_uint64), ("max_l_seq", C.c_uint32),
                ("mem", C.c_int32), ("meta", C.c_void_p)]


class CPairSoa(C.Structure):
    _fields_ = [("rec", C.c_void_p), ("qhash", C.c_void_p)]


PAIR_REC_DTYPE = np.dtype([("tid", "<i4"), ("pos", "<i4"), ("mtid", "<i4"), ("mpos", "<i4"), ("end", "<i4"), ("flag", "<u2"), ("l_seq", "<u2"),
                           ("clip_l", "<u2"), ("clip_r", "<u2"), ("mapq", "u1"), ("cig", "u1"), ("pad", "<u2")])
assert PAIR_REC_DTYPE.itemsize == 32


class FrontChunk(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_primary", C.c_uint64), ("last_placed", C.c_int64), ("tail_primary", C.c_uint64),
                ("max_l_seq", C.c_uint32), ("scan_slow_segments", C.c_uint32)]


def _front_chunk(c):
    return {k: int(getattr(c, k)) for k, _ in FrontChunk._fields_}


class ScoreStats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_skipped", C.c_uint64), ("n_scored", C.c_uint64), ("n_soft_items", C.c_uint64),
                ("ms_classify", C.c_float), ("ms_score", C.c_float), ("ms_soft", C.c_float), ("n_stage_b_whole", C.c_uint32),
                ("n_stage_b_soft", C.c_uint32)]


class ClusterStats(C.Structure):
    _fields_ = [("n_treads", C.c_uint64), ("n_groups", C.c_uint64), ("n_clusters", C.c_uint64), ("n_bounds", C.c_uint64),
                ("n_tie_fixups", C.c_uint64), ("ms_sort", C.c_float), ("ms_sweep", C.c_float), ("ms_bounds", C.c_float)]


class BinInfo(C.Structure):
    _fields_ = [("proportion_repeat", C.c_float), ("min_mapq", C.c_uint8), ("frag", C.c_uint32 * 4096),
                ("header_len", C.c_int32), ("n_reads", C.c_int32), ("qnames_bytes", C.c_uint64)]


SOFT_DTYPE = np.dtype([("read_side", "<u4"), ("res_first", "<u4"), ("res_after", "<u4"), ("seg_len", "<u4")])
TREAD_DTYPE = np.dtype([("tid", "<i4"), ("position", "<u4"), ("repeat", "S6"), ("flag", "<u2"), ("split", "u1"),
                        ("mapping_quality", "u1"), ("repeat_count", "u1"), ("align_length", "u1"), ("qname_id", "<i8")],
                       align=True)
REGION_REQ_DTYPE = np.dtype([("first_block", "<u4"), ("n_blocks", "<u4"), ("in_block", "<u4"), ("tid", "<i4"), ("beg", "<i4"), ("end", "<i4")])
BOUNDS_DTYPE = np.dtype([("tid", "<i4"), ("left", "<u4"), ("left_most", "<u4"), ("right", "<u4"), ("right_most", "<u4"),
                         ("center_mass", "<u4"), ("n_left", "<u2"), ("n_right", "<u2"), ("n_total", "<u2"),
                         ("repeat", "S7")], align=True)
UNPLACED_DTYPE = np.dtype([("repeat", "S7"), ("count", "<i8")], align=True)
assert TREAD_DTYPE.itemsize == 32 and BOUNDS_DTYPE.itemsize == 40 and UNPLACED_DTYPE.itemsize == 16

SUPPORT_DTYPE = np.dtype([("type", "u1"), ("repeat_count", "u1"), ("cigar_ins", "u1"), ("cigar_del", "u1"), ("fragment_length", "<u4"),
                          ("fragment_percentile", "<f8"), ("rec", "<i8")], align=True)
CALL_DTYPE = np.dtype([("tid", "<i4"), ("start", "<u4"), ("stop", "<u4"), ("repeat", "S7"), ("allele1", "<f8"), ("allele2", "<f8"),
                       ("overlapping_reads", "<u4"), ("anchored_reads", "<u4"), ("spanning_reads", "<u4"), ("spanning_pairs", "<u4"),
                       ("left_clips", "<u4"), ("right_clips", "<u4"), ("sum_str_counts", "<u4"), ("expected_spanning_fragments", "<f4"),
                       ("spanning_fragments_oe_percentile", "<f4"), ("unplaced_reads", "<i4"), ("depth", "<f8"), ("is_large", "<i4")],
                      align=True)
assert SUPPORT_DTYPE.itemsize == 24 and CALL_DTYPE.itemsize == 96, (SUPPORT_DTYPE.itemsize, CALL_DTYPE.itemsize)
SUPPORT_TYPES = ["SpanningFragment", "SpanningRead", "OverlappingRead"]


class SpanSummary(C.Structure):
    _fields_ = [("median_depth", C.c_int32), ("expected_spanners", C.c_float), ("n_support", C.c_uint64)]


class CallOpts(C.Structure):
    _fields_ = [("median_fragment_length", C.c_int32), ("min_support", C.c_int32), ("min_clip", C.c_uint16), ("min_clip_total", C.c_uint16)]


LOCUS_DTYPE = np.dtype([("b", BOUNDS_DTYPE), ("name", "S128")], align=True)
assert LOCUS_DTYPE.itemsize == 168, LOCUS_DTYPE.itemsize
SOFT_TAKEN = 255
MEM_HOST, MEM_DEVICE = 0, 1
MODE_MERGE, MODE_CALL = 0, 1

# every symbol include/strling_amd.h declares
EXPORTS = ["strl_version", "strl_last_error", "strl_device_count", "strl_ctx_create", "strl_ctx_destroy", "strl_ctx_stream",
           "strl_ctx_sync", "strl_ctx_set_opts", "strl_ctx_set_genome", "strl_soa_from_records", "strl_score_reads", "strl_index_chrom", "strl_index_regions",
           "strl_ctx_enable_timing", "strl_ctx_kernel_times", "strl_ctx_kernel_times_detail", "strl_pair_reads", "strl_pairer_create", "strl_pairer_destroy", "strl_pairer_add",
           "strl_pairer_result", "strl_qname_hash", "strl_extract", "strl_cluster", "strl_cluster_replay", "strl_frag_median",
           "strl_bin_write", "strl_bin_read", "strl_bounds_row", "strl_cluster_members", "strl_spanners", "strl_genotype",
           "strl_calls_finish", "strl_unplaced_order", "strl_call_row", "strl_canonical_repeat", "strl_assign_reads_loci", "strl_group_order",
           "strl_extract_device", "strl_treads_fetch", "strl_ctx_pair_times", "strl_sort_pairs", "strl_cluster_resident", "strl_ctx_cluster_times", "strl_pair_rows", "strl_extract_begin", "strl_extract_add", "strl_extract_finish", "strl_pair_rule", "strl_bounds_bare", "strl_ctx_treads_device", "strl_cluster_gathered", "strl_inflate_blocks", "strl_ctx_inflate_ms", "strl_regions_fetch", "strl_front_begin", "strl_front_push", "strl_front_push_after", "strl_front_reserve", "strl_front_stage", "strl_front_enqueue_after", "strl_front_collect", "strl_ctxs_extract_gather", "strl_front_finish", "strl_front_fragwords", "strl_front_fragwords_async", "strl_event_wait", "strl_front_records", "strl_front_tids", "strl_front_qnames", "strl_front_treads_named", "strl_pinned_alloc", "strl_pinned_free", "strl_comm_unique_id", "strl_ctx_comm_init", "strl_ctxs_comm_init", "strl_ctx_comm_info", "strl_cluster_exchange", "strl_ctxs_cluster_exchange", "strl_exchange_treads", "strl_ctx_set_treads", "strl_cluster_collect", "strl_ctx_tail_stream", "strl_ctx_mem_info", "strl_bin_peek", "strl_front_trim_next", "strl_front_tail_bytes", "strl_ctx_blocking_waits", "strl_score_read_host", "strl_front_end"]


def lib_path():
    return _build.LIB


def load(build_if_missing=True):
    """dlopen the in-tree library (building it with hipcc when missing). Raises when impossible."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get("STRL_LIB", _build.LIB)   # STRL_LIB: A/B a differently built copy of the same library
    if not os.path.exists(path):
        if not build_if_missing:
            raise StrlingError(f"{path} is missing: build it with `python -m strling_amd.build` (no CPU fallback)")
        _build.build()
    L = C.CDLL(path)
    L.strl_last_error.restype = C.c_char_p
    L.strl_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.strl_ctx_destroy.argtypes = [C.c_void_p]
    L.strl_ctx_stream.argtypes = [C.c_void_p]
    L.strl_ctx_stream.restype = C.c_void_p
    L.strl_ctx_sync.argtypes = [C.c_void_p]
    L.strl_ctx_set_opts.argtypes = [C.c_void_p, C.POINTER(Opts)]
    L.strl_ctx_set_genome.argtypes = [C.c_void_p, C.POINTER(CGenomeStr)]
    L.strl_ctx_enable_timing.argtypes = [C.c_void_p, C.c_int]
    L.strl_ctx_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_double * 3), C.POINTER(C.c_uint64)]
    L.strl_ctx_kernel_times_detail.argtypes = [C.c_void_p, C.POINTER(C.c_double * 8), C.POINTER(C.c_uint64)]
    L.strl_soa_from_records.argtypes = [C.POINTER(CRecords)] + [C.c_void_p] * 6 + [C.POINTER(C.c_uint32)]
    L.strl_score_reads.argtypes = [C.c_void_p, C.POINTER(CReadSoa), C.c_void_p, C.c_void_p, C.c_uint64,
                                   C.POINTER(C.c_uint64), C.POINTER(ScoreStats)]
    L.strl_index_chrom.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64)]
    L.strl_index_regions.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64,
                                     C.POINTER(C.c_uint64)]
    L.strl_pair_reads.argtypes = [C.POINTER(CRecords), C.POINTER(Opts), C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64,
                                  C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.strl_qname_hash.argtypes = [C.POINTER(CRecords), C.c_void_p]
    L.strl_extract.argtypes = [C.c_void_p, C.POINTER(CRecords), C.c_int64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                               C.POINTER(ScoreStats)]
    L.strl_cluster.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_int32, C.c_uint16, C.c_uint16,
                               C.c_uint16, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.c_uint64,
                               C.POINTER(C.c_uint64), C.POINTER(ClusterStats)]
    L.strl_cluster_replay.argtypes = [C.c_void_p]
    L.strl_frag_median.argtypes = [C.c_void_p, C.c_double]
    L.strl_bin_write.argtypes = [C.c_char_p, C.c_float, C.c_uint8, C.c_void_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_uint64,
                                 C.c_void_p, C.c_char_p]
    L.strl_bin_read.argtypes = [C.c_char_p, C.POINTER(BinInfo), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.strl_bounds_row.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_char_p]
    L.strl_cluster_members.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.strl_spanners.argtypes = [C.POINTER(CRecords), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint8, C.c_void_p, C.c_uint64,
                                C.POINTER(SpanSummary)]
    L.strl_genotype.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(CallOpts),
                                C.c_double, C.c_void_p]
    L.strl_calls_finish.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    L.strl_unplaced_order.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    L.strl_call_row.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_char_p]
    L.strl_score_read_host.argtypes = [C.c_char_p, C.c_int32, C.c_double, C.POINTER(C.c_uint32)]
    L.strl_canonical_repeat.argtypes = [C.c_char_p, C.c_char_p]
    L.strl_canonical_repeat.restype = None
    L.strl_group_order.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.strl_assign_reads_loci.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
    L.strl_extract_device.argtypes = [C.c_void_p, C.POINTER(CReadSoa), C.POINTER(CPairSoa), C.c_int64, C.c_uint64, C.c_uint64]
    L.strl_treads_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(ScoreStats)]
    L.strl_ctx_pair_times.argtypes = [C.c_void_p, C.POINTER(C.c_double * 5)]
    L.strl_cluster_resident.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int, C.c_uint32, C.c_int32, C.c_uint16, C.c_uint16, C.c_uint16,
                                        C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                        C.POINTER(ClusterStats)]
    L.strl_ctx_tail_stream.argtypes = [C.c_void_p]
    L.strl_ctx_tail_stream.restype = C.c_void_p
    L.strl_cluster_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                       C.POINTER(ClusterStats)]
    L.strl_extract_begin.argtypes = [C.c_void_p, C.c_uint64]
    L.strl_extract_add.argtypes = [C.c_void_p, C.POINTER(CReadSoa), C.POINTER(CPairSoa)]
    L.strl_extract_finish.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64]
    L.strl_pair_rule.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(Opts), C.c_uint32, C.POINTER(C.c_int)]
    L.strl_bounds_bare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint16, C.c_uint16, C.c_uint16, C.c_void_p, C.POINTER(C.c_int)]
    L.strl_ctx_treads_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p)]
    L.strl_cluster_gathered.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int32, C.c_int, C.c_uint32,
                                        C.c_int32, C.c_uint16, C.c_uint16, C.c_uint16, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p,
                                        C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(ClusterStats)]
    L.strl_inflate_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
    L.strl_ctx_inflate_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.strl_regions_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                     C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.strl_front_begin.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64]
    L.strl_front_push.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_int)]
    L.strl_front_finish.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    L.strl_front_trim_next.argtypes = [C.c_void_p, C.c_uint32]
    L.strl_front_tail_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.strl_ctx_blocking_waits.argtypes = [C.c_void_p, C.c_int]
    L.strl_ctxs_extract_gather.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
    L.strl_front_fragwords.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    L.strl_front_tids.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.strl_front_qnames.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.strl_pinned_alloc.argtypes = [C.c_uint64]
    L.strl_pinned_alloc.restype = C.c_void_p
    L.strl_pinned_free.argtypes = [C.c_void_p]
    L.strl_comm_unique_id.argtypes = [C.c_void_p]
    L.strl_ctx_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.strl_ctxs_comm_init.argtypes = [C.c_void_p, C.c_int]
    L.strl_ctx_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.strl_cluster_exchange.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int32, C.c_int, C.c_uint32, C.c_int32, C.c_uint16, C.c_uint16, C.c_uint16,
                                        C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(ClusterStats)]
    L.strl_ctxs_cluster_exchange.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_int32, C.c_int, C.c_uint32, C.c_int32, C.c_uint16, C.c_uint16, C.c_uint16]
    L.strl_exchange_treads.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.strl_pair_rows.argtypes = [C.POINTER(CRecords)] + [C.c_void_p] * 5
    L.strl_ctx_cluster_times.argtypes = [C.c_void_p, C.POINTER(C.c_double * 3)]
    L.strl_sort_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
    _LIB = L
    return L


REGION_DTYPE = np.dtype([("start", "<u8"), ("stop", "<u8"), ("unit", "S8")])


def index_regions(seq, words, window=100, step=60):
    """host half of strling index: merge + trim the scored windows (strl_index_regions)"""
    L = load()
    words = np.ascontiguousarray(words, np.uint32)
    cap = len(words) + 1
    out = np.zeros(cap, REGION_DTYPE)
    n = C.c_uint64(0)
    _check(L.strl_index_regions(seq, len(seq), words.ctypes.data, len(words), window, step, out.ctypes.data, cap, C.byref(n)))
    return [(int(r["start"]), int(r["stop"]), r["unit"].decode()) for r in out[:n.value]]


def _check(rc):
    if rc != 0:
        raise StrlingError(f"strling_amd error {rc}: {load().strl_last_error().decode()}")


def _ptr(a):
    return a.ctypes.data if a is not None and a.size else None


class _RecView:
    def __init__(self, rec: RecordBatch):
        k = dict(tid=np.ascontiguousarray(rec.tid, np.int32), pos=np.ascontiguousarray(rec.pos, np.int32),
                 mtid=np.ascontiguousarray(rec.mtid, np.int32), mpos=np.ascontiguousarray(rec.mpos, np.int32),
                 flag=np.ascontiguousarray(rec.flag, np.uint16), mapq=np.ascontiguousarray(rec.mapq, np.uint8),
                 cigar_off=np.ascontiguousarray(rec.cigar_off, np.uint32),
                 cigar=np.ascontiguousarray(np.append(rec.cigar, 0), np.uint32),
                 seq_off=np.ascontiguousarray(rec.seq_off, np.uint64), l_seq=np.ascontiguousarray(rec.l_seq, np.int32),
                 seq4=np.ascontiguousarray(rec.seq4, np.uint8), qname_off=np.ascontiguousarray(rec.qname_off, np.uint64),
                 qnames=np.frombuffer(bytes(rec.qnames) + b"\0", dtype=np.uint8))
        self.keep = k
        self.n = int(k["tid"].size)
        self.c = CRecords(self.n, *[k[f].ctypes.data for f in ("tid", "pos", "mtid", "mpos", "flag", "mapq", "cigar_off",
                                                               "cigar", "seq_off", "l_seq", "seq4", "qname_off", "qnames")])


class Soa:
    """Host numpy SoA batch (the layout the kernels consume)."""

    def __init__(self, rec: RecordBatch):
        L = load()
        rv = _RecView(rec)
        n = rv.n
        self.rv = rv
        self.n = n
        self.end = np.zeros(n, np.int32)
        self.seq_off = np.zeros(n, np.uint32)
        self.l_seq = np.zeros(n, np.uint16)
        self.clip_l = np.zeros(n, np.uint16)
        self.clip_r = np.zeros(n, np.uint16)
        self.cig = np.zeros(n, np.uint8)
        mx = C.c_uint32(0)
        _check(L.strl_soa_from_records(C.byref(rv.c), self.end.ctypes.data, self.seq_off.ctypes.data, self.l_seq.ctypes.data,
                                       self.clip_l.ctypes.data, self.clip_r.ctypes.data, self.cig.ctypes.data, C.byref(mx)))
        self.max_l_seq = mx.value
        self.tid, self.pos, self.mapq, self.seq4 = rv.keep["tid"], rv.keep["pos"], rv.keep["mapq"], rv.keep["seq4"]

    def pair_rows(self):
        """the 32-byte rows + qname hashes the pair logic reads (strl_pair_rows, strl_qname_hash) -> (rows, qhash)"""
        rows = np.zeros(max(self.n, 1), PAIR_REC_DTYPE)
        _check(load().strl_pair_rows(C.byref(self.rv.c), _ptr(self.end), _ptr(self.clip_l), _ptr(self.clip_r), _ptr(self.cig), rows.ctypes.data))
        qh = np.zeros(max(self.n, 1), np.uint64)
        _check(load().strl_qname_hash(C.byref(self.rv.c), qh.ctypes.data))
        return rows[:self.n], qh[:self.n]

    def meta_rows(self):
        """strl_read_meta rows (16 B per read: seq_off | l_seq, clip_l | clip_r, cig, mapq | pad) for strl_read_soa.meta"""
        m = np.zeros((max(self.n, 1), 4), np.uint32)
        m[:self.n, 0] = self.seq_off
        m[:self.n, 1] = self.l_seq.astype(np.uint32) | (self.clip_l.astype(np.uint32) << 16)
        m[:self.n, 2] = self.clip_r.astype(np.uint32) | (self.cig.astype(np.uint32) << 16) | (self.mapq.astype(np.uint32) << 24)
        return m[:self.n]

    def c_struct(self):
        return CReadSoa(self.n, _ptr(self.tid), _ptr(self.pos), _ptr(self.end), _ptr(self.seq_off), _ptr(self.l_seq),
                        _ptr(self.clip_l), _ptr(self.clip_r), _ptr(self.mapq), _ptr(self.cig), self.seq4.ctypes.data,
                        self.seq4.size, self.max_l_seq, MEM_HOST)


class Context:
    """One context per GPU (strl_ctx)."""

    def __init__(self, device=0):
        self.L = load()
        h = C.c_void_p()
        _check(self.L.strl_ctx_create(device, C.byref(h)))
        self.h = h
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.L.strl_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return self.L.strl_ctx_stream(self.h)

    def sync(self):
        _check(self.L.strl_ctx_sync(self.h))

    def enable_timing(self, on=True):
        _check(self.L.strl_ctx_enable_timing(self.h, int(on)))

    def kernel_times(self):
        """-> ((ms_classify, ms_score, ms_soft) summed, n_launches) since enable_timing(True)"""
        ms = (C.c_double * 3)()
        n = C.c_uint64(0)
        _check(self.L.strl_ctx_kernel_times(self.h, C.byref(ms), C.byref(n)))
        return tuple(ms), n.value

    KERNEL_NAMES = ["classify_kernel", "score_kernel<whole,A>", "compact_kernel<whole>", "score_kernel<whole,B>", "soft_compact_kernel",
                    "score_kernel<segment,A>", "compact_kernel<segment>", "score_kernel<segment,B>"]

    def kernel_times_detail(self):
        """-> ({launch name: ms summed}, n) since enable_timing(True): one entry per kernel launch of a scoring pass"""
        ms = (C.c_double * 8)()
        n = C.c_uint64(0)
        _check(self.L.strl_ctx_kernel_times_detail(self.h, C.byref(ms), C.byref(n)))
        return dict(zip(self.KERNEL_NAMES, list(ms))), n.value

    def set_opts(self, proportion_repeat=0.8, min_mapq=40, median_fragment_length=0):
        self.opts = Opts(median_fragment_length, proportion_repeat, min_mapq)
        _check(self.L.strl_ctx_set_opts(self.h, C.byref(self.opts)))

    def set_genome(self, g: GenomeStr):
        if g is None:
            _check(self.L.strl_ctx_set_genome(self.h, None))
            return
        has = np.ascontiguousarray(g.has_chrom, np.uint8)
        off = np.ascontiguousarray(g.iv_off, np.int64)
        st = np.ascontiguousarray(g.iv_start, np.int32)
        en = np.ascontiguousarray(g.iv_stop, np.int32)
        cg = CGenomeStr(int(g.n_tid), has.ctypes.data, off.ctypes.data, _ptr(st), _ptr(en))
        _check(self.L.strl_ctx_set_genome(self.h, C.byref(cg)))

    # ---- scorer --------------------------------------------------------------------------------
    def score_reads(self, rec_or_soa):
        """-> (whole uint32[n], soft SOFT_DTYPE[m] sorted by read_side, ScoreStats)"""
        soa = rec_or_soa if isinstance(rec_or_soa, Soa) else Soa(rec_or_soa)
        n = soa.n
        whole = np.zeros(max(n, 1), np.uint32)
        soft = np.zeros(2 * n + 1, SOFT_DTYPE)
        ns = C.c_uint64(0)
        st = ScoreStats()
        cs = soa.c_struct()
        _check(self.L.strl_score_reads(self.h, C.byref(cs), whole.ctypes.data, soft.ctypes.data, 2 * n, C.byref(ns), C.byref(st)))
        return whole[:n], soft[:ns.value].copy(), st

    def score_device(self, cs: CReadSoa, whole_ptr, soft_ptr, soft_cap, sync=False):
        """Device-resident batch (pointers from torch tensors). Asynchronous unless sync."""
        if sync:
            ns = C.c_uint64(0)
            st = ScoreStats()
            _check(self.L.strl_score_reads(self.h, C.byref(cs), whole_ptr, soft_ptr, soft_cap, C.byref(ns), C.byref(st)))
            return ns.value, st
        _check(self.L.strl_score_reads(self.h, C.byref(cs), whole_ptr, soft_ptr, soft_cap, None, None))
        return None

    def index_chrom(self, seq, window=100, step=60):
        """packed get_repeat word of every window of a chromosome (strling index, genome_strs.nim:61-92)"""
        if isinstance(seq, str):
            seq = seq.encode()
        nw = C.c_uint64(0)
        _check(self.L.strl_index_chrom(self.h, seq, len(seq), window, step, None, C.byref(nw)))
        words = np.zeros(max(1, nw.value), np.uint32)
        _check(self.L.strl_index_chrom(self.h, seq, len(seq), window, step, words.ctypes.data, C.byref(nw)))
        return words[:nw.value]

    def index_regions(self, seq, window=100, step=60):
        """the (start, stop, unit) rows `strling index` writes for one chromosome (genome_strs.nim:61-92, :137)"""
        if isinstance(seq, str):
            seq = seq.encode()
        return index_regions(seq, self.index_chrom(seq, window, step), window, step)

    def sort_pairs(self, keys, vals, bit_lo=0, bits=64, n_max=None):
        """stable LSD radix sort of (uint64 key, uint32 value) pairs on the device (strl_sort_pairs) -> (keys, vals)"""
        k = np.ascontiguousarray(keys, np.uint64).copy()
        v = np.ascontiguousarray(vals, np.uint32).copy()
        _check(self.L.strl_sort_pairs(self.h, _ptr(k), _ptr(v), k.size, n_max or k.size, bit_lo, bits))
        return k, v

    def extract_device(self, cs: CReadSoa, cp: CPairSoa, n_tail, item_cap=0, tread_cap=0):
        """scoring + pair logic of one batch, asynchronous for device-resident batches (strl_extract_device)"""
        _check(self.L.strl_extract_device(self.h, C.byref(cs), C.byref(cp), n_tail, item_cap, tread_cap))

    def extract_chunks(self, chunks, n_tail, hint=0):
        """chunked form: chunks = iterable of (CReadSoa, CPairSoa) in file order (strl_extract_begin / _add / _finish)"""
        _check(self.L.strl_extract_begin(self.h, hint))
        for cs, cp in chunks:
            _check(self.L.strl_extract_add(self.h, C.byref(cs), C.byref(cp)))
        _check(self.L.strl_extract_finish(self.h, n_tail, 0, 0))

    def treads_fetch(self, want=True):
        """-> (treads of the last extract_device call in .bin order, ScoreStats)"""
        no = C.c_uint64(0)
        st = ScoreStats()
        _check(self.L.strl_treads_fetch(self.h, None, 0, C.byref(no), C.byref(st)))
        out = np.zeros(max(1, no.value), TREAD_DTYPE)
        if want and no.value:
            _check(self.L.strl_treads_fetch(self.h, out.ctypes.data, out.size, C.byref(no), None))
        return out[:no.value], st

    def pair_times(self):
        ms = (C.c_double * 5)()
        _check(self.L.strl_ctx_pair_times(self.h, C.byref(ms)))
        return dict(zip(["pair_soft_items_kernel", "pair_probe_kernel", "pair_join_sort", "pair_groups_kernel", "pair_order"], list(ms)))

    # ---- extract (score + pair) -----------------------------------------------------------------
    def extract(self, rec: RecordBatch, n_tail=-1):
        rv = _RecView(rec)
        cap = max(1024, rv.n // 4)
        while True:
            out = np.zeros(cap, TREAD_DTYPE)
            no = C.c_uint64(0)
            st = ScoreStats()
            rc = self.L.strl_extract(self.h, C.byref(rv.c), n_tail, out.ctypes.data, cap, C.byref(no), C.byref(st))
            if rc == -4 and no.value > cap:
                cap = int(no.value)
                continue
            _check(rc)
            return out[:no.value].copy(), st

    def pair_reads(self, rec: RecordBatch, whole, soft, n_tail=-1):
        rv = _RecView(rec)
        whole = np.ascontiguousarray(whole, np.uint32)
        soft = np.ascontiguousarray(soft, SOFT_DTYPE)
        cap = max(1024, rv.n // 4)
        while True:
            out = np.zeros(cap, TREAD_DTYPE)
            no = C.c_uint64(0)
            rc = self.L.strl_pair_reads(C.byref(rv.c), C.byref(self.opts), whole.ctypes.data, _ptr(soft), soft.size, n_tail,
                                        out.ctypes.data, cap, C.byref(no))
            if rc == -4 and no.value > cap:
                cap = int(no.value)
                continue
            _check(rc)
            return out[:no.value].copy()

    # ---- cluster --------------------------------------------------------------------------------
    def cluster(self, treads, mode, window, min_support=5, min_clip=0, min_clip_total=0, max_clip_dist=200):
        t = np.ascontiguousarray(treads, TREAD_DTYPE)
        cap = max(64, t.size)
        out = np.zeros(cap, BOUNDS_DTYPE)
        unpl = np.zeros(8192, UNPLACED_DTYPE)
        no, nu = C.c_uint64(0), C.c_uint64(0)
        st = ClusterStats()
        _check(self.L.strl_cluster(self.h, _ptr(t), t.size, mode, window, min_support, min_clip, min_clip_total, max_clip_dist,
                                   out.ctypes.data, cap, C.byref(no), unpl.ctypes.data, unpl.size, C.byref(nu), C.byref(st)))
        return out[:no.value].copy(), unpl[:nu.value].copy(), st

    def cluster_resident(self, n_tid, window, min_support=5, min_clip=0, min_clip_total=0, max_clip_dist=200, pos_bits=0, fetch=True,
                         cap=None):
        """cluster the treads the last extract_device call left on the device (strl_cluster_resident, call mode).
        fetch=False only enqueues the kernels."""
        if not fetch:
            _check(self.L.strl_cluster_resident(self.h, MODE_CALL, n_tid, pos_bits, window, min_support, min_clip, min_clip_total,
                                                max_clip_dist, None, 0, None, None, 0, None, None))
            return None
        cap = cap or 1 << 16
        while True:
            out = np.zeros(cap, BOUNDS_DTYPE)
            unpl = np.zeros(8192, UNPLACED_DTYPE)
            no, nu = C.c_uint64(0), C.c_uint64(0)
            st = ClusterStats()
            rc = self.L.strl_cluster_resident(self.h, MODE_CALL, n_tid, pos_bits, window, min_support, min_clip, min_clip_total,
                                              max_clip_dist, out.ctypes.data, cap, C.byref(no), unpl.ctypes.data, unpl.size,
                                              C.byref(nu), C.byref(st))
            if rc == -4 and no.value > cap:
                cap = int(no.value)
                continue
            _check(rc)
            return out[:no.value].copy(), unpl[:nu.value].copy(), st

    def cluster_collect(self, cap=None):
        """results of the last cluster_resident(fetch=False) -- which ran on the context's side stream, overlapping whatever
        was enqueued after it (strl_cluster_collect) -> (bounds, unplaced, stats)"""
        cap = cap or 1 << 16
        while True:
            out = np.zeros(cap, BOUNDS_DTYPE)
            unpl = np.zeros(8192, UNPLACED_DTYPE)
            no, nu = C.c_uint64(0), C.c_uint64(0)
            st = ClusterStats()
            rc = self.L.strl_cluster_collect(self.h, out.ctypes.data, cap, C.byref(no), unpl.ctypes.data, unpl.size, C.byref(nu), C.byref(st))
            if rc == -4 and no.value > cap:
                cap = int(no.value)
                continue
            _check(rc)
            return out[:no.value].copy(), unpl[:nu.value].copy(), st

    def bounds_bare(self, positions, splits, max_clip_dist, min_clip=0, min_clip_total=0):
        """bounds() + gate of one bare cluster (strl_bounds_bare) -> (bounds record, good)"""
        p = np.ascontiguousarray(positions, np.uint32)
        sp = np.ascontiguousarray(splits, np.uint8)
        out = np.zeros(1, BOUNDS_DTYPE)
        good = C.c_int(0)
        _check(self.L.strl_bounds_bare(self.h, p.ctypes.data, sp.ctypes.data, p.size, min_clip, min_clip_total, max_clip_dist, out.ctypes.data, C.byref(good)))
        return out[0], bool(good.value)

    def regions_fetch(self, streams, sizes, regions, crcs=None):
        """strl_regions_fetch: `streams` / `sizes` = raw DEFLATE payloads + ISIZE of BGZF blocks; regions = list of
        (first_block, n_blocks, in_block, tid, beg, end) -> list of (record bytes, status)"""
        comp = np.frombuffer(b"".join(streams) + b"\0" * 8, np.uint8)
        clen = np.array([len(x) for x in streams], np.uint32)
        coff = np.zeros(len(streams), np.uint64)
        coff[1:] = np.cumsum(clen[:-1], dtype=np.uint64)
        isz = np.asarray(sizes, np.uint32)
        req = np.zeros(len(regions), REGION_REQ_DTYPE)
        for k, r in enumerate(regions):
            req[k] = tuple(r)
        cap = sum(int(isz[r[0]:r[0] + r[1]].sum()) + 32 for r in regions) + 64
        out = np.zeros(cap, np.uint8)
        off = np.zeros(len(regions), np.uint64)
        ln = np.zeros(len(regions), np.uint64)
        st = np.zeros(len(regions), np.uint8)
        crc = None if crcs is None else np.asarray(crcs, np.uint32)
        _check(self.L.strl_regions_fetch(self.h, comp.ctypes.data, int(clen.sum()), _ptr(coff), _ptr(clen), _ptr(isz), None if crc is None else _ptr(crc), len(streams),
                                         req.ctypes.data, len(regions), out.ctypes.data, cap, _ptr(off), _ptr(ln), _ptr(st)))
        return [(out[int(o):int(o) + int(n)].tobytes(), int(x)) for o, n, x in zip(off, ln, st)]

    def inflate_blocks(self, streams, sizes):
        """raw DEFLATE streams (list of bytes) with their inflated sizes -> list of inflated bytes (strl_inflate_blocks)"""
        comp = np.frombuffer(b"".join(streams) + b"\0" * 8, np.uint8)
        clen = np.array([len(s) for s in streams], np.uint32)
        coff = np.zeros(len(streams), np.uint64)
        coff[1:] = np.cumsum(clen[:-1], dtype=np.uint64)
        isz = np.asarray(sizes, np.uint32)
        out = np.zeros(int(isz.sum()) + 16, np.uint8)
        _check(self.L.strl_inflate_blocks(self.h, comp.ctypes.data, int(clen.sum()), _ptr(coff), _ptr(clen), _ptr(isz), len(streams), out.ctypes.data, out.size))
        res, o = [], 0
        for n in isz:
            res.append(out[o:o + int(n)].tobytes())
            o += int(n)
        return res

    # ---- extract with the BAM front end on the device ---------------------------------------------
    @staticmethod
    def _bam_blocks(path):
        """the host side of the device front end in Python: BGZF block table (payload offset, payload length, isize, crc), the
        header (inflated here) -> dict(data, blocks, n_ref, targets, text, b0 = the block the first record starts in, first_off)"""
        import struct, zlib
        data = np.fromfile(path, np.uint8)
        raw = data.tobytes()
        blocks, o = [], 0
        while o < len(raw):
            xlen = struct.unpack_from("<H", raw, o + 10)[0]
            bsize = struct.unpack_from("<H", raw, o + 16)[0] + 1
            isz = struct.unpack_from("<I", raw, o + bsize - 4)[0]
            if isz:
                blocks.append((o + 12 + xlen, bsize - 12 - xlen - 8, isz, struct.unpack_from("<I", raw, o + bsize - 8)[0]))
            o += bsize
        hdr, k = b"", 0

        def need(nbytes):
            nonlocal hdr, k
            while len(hdr) < nbytes and k < len(blocks):
                po, pl = blocks[k][:2]
                hdr += zlib.decompress(raw[po:po + pl], -15)
                k += 1
            if len(hdr) < nbytes:
                raise StrlingError("truncated BAM header")
        need(12)
        assert hdr[:4] == b"BAM\1"
        l_text = struct.unpack_from("<i", hdr, 4)[0]
        need(12 + l_text)
        text = hdr[8:8 + l_text].decode()
        n_ref = struct.unpack_from("<i", hdr, 8 + l_text)[0]
        at, targets = 12 + l_text, []
        for _ in range(n_ref):
            need(at + 4)
            ln = struct.unpack_from("<i", hdr, at)[0]
            need(at + 8 + ln)
            targets.append((hdr[at + 4:at + 4 + ln - 1].decode(), struct.unpack_from("<i", hdr, at + 4 + ln)[0]))
            at += 8 + ln
        cum, b0 = 0, 0                                        # the block the first record starts in
        while b0 < len(blocks) and cum + blocks[b0][2] <= at:
            cum += blocks[b0][2]
            b0 += 1
        return dict(data=data, raw=raw, blocks=blocks, n_ref=n_ref, targets=targets, text=text, b0=b0, first_off=at - cum)

    def _front_push_blocks(self, B, c0, c1, chunk_blocks, check_crc, trim=0):
        """blocks [c0, c1) of the table through strl_front_push in chunks; trim = inflated bytes at the end of the LAST block that
        are not this context's (strl_front_trim_next before the last chunk) -> the chunk summaries that came back"""
        done = (FrontChunk * 2)()
        nd = C.c_int(0)
        chunks, keep = [], []
        data, blocks = B["data"], B["blocks"]
        for s0 in range(c0, c1, chunk_blocks):
            cb = blocks[s0:min(c1, s0 + chunk_blocks)]
            lo, hi = cb[0][0], cb[-1][0] + cb[-1][1]
            comp = np.ascontiguousarray(data[lo:hi])
            coff = np.array([b[0] - lo for b in cb], np.uint64)
            clen = np.array([b[1] for b in cb], np.uint32)
            isz = np.array([b[2] for b in cb], np.uint32)
            crc = np.array([b[3] for b in cb], np.uint32)
            keep.append((comp, coff, clen, isz, crc))           # pageable memory: the copy is staged by the runtime
            if trim and s0 + chunk_blocks >= c1:
                _check(self.L.strl_front_trim_next(self.h, trim))
            _check(self.L.strl_front_push(self.h, comp.ctypes.data, comp.size, _ptr(coff), _ptr(clen), _ptr(isz), _ptr(crc) if check_crc else None, len(cb), done, C.byref(nd)))
            chunks += [_front_chunk(done[i]) for i in range(nd.value)]
            keep = keep[-3:]
        _check(self.L.strl_front_finish(self.h, done, C.byref(nd)))
        chunks += [_front_chunk(done[i]) for i in range(nd.value)]
        return chunks

    def _front_results(self, B, chunks):
        """pair logic over everything this context holds + the treads, their names, the fragment words"""
        n_ref = B["n_ref"]
        n_rec = sum(c["n_records"] for c in chunks)
        n_tail = 0
        for c in chunks:                                      # trailing run of unplaced records over the whole file
            n_tail = c["n_records"] - 1 - c["last_placed"] if c["last_placed"] >= 0 else n_tail + c["n_records"]
        _check(self.L.strl_extract_finish(self.h, n_tail, 3 * n_rec + 16, 8 * n_rec + 16))
        no = C.c_uint64(0)
        _check(self.L.strl_treads_fetch(self.h, None, 0, C.byref(no), None))
        out = np.zeros(max(1, no.value), TREAD_DTYPE)
        _check(self.L.strl_treads_fetch(self.h, out.ctypes.data, out.size, C.byref(no), None))
        out = out[:no.value]
        ids = np.ascontiguousarray(out["qname_id"], np.int64)
        qoff = np.zeros(ids.size + 1, np.uint64)
        needb = C.c_uint64(0)
        buf = np.zeros(max(16, 256 * ids.size), np.uint8)
        _check(self.L.strl_front_qnames(self.h, _ptr(ids), ids.size, _ptr(qoff), buf.ctypes.data, buf.size, C.byref(needb)))
        names = [buf[int(qoff[i]):int(qoff[i + 1])].tobytes() for i in range(ids.size)]
        fw = np.zeros(max(1, n_rec), np.uint32)
        _check(self.L.strl_front_fragwords(self.h, 0, n_rec, fw.ctypes.data))
        seen = np.zeros(max(1, n_ref), np.uint8)
        _check(self.L.strl_front_tids(self.h, seen.ctypes.data, n_ref))
        return dict(treads=out, qnames=names, fragwords=fw[:n_rec], chunks=chunks, n_records=n_rec, n_tail=n_tail, targets=B["targets"], header=B["text"],
                    tids_seen=seen[:n_ref])

    def extract_bam_device(self, path, chunk_blocks=16384, n_reads_hint=0, check_crc=True):
        """`strling extract`'s read loop over a BAM FILE with inflate, record scan and parse on the device
        (strl_front_begin / _push / _finish + strl_extract_finish).  The host side here only walks BGZF block headers.
        -> dict(treads, qnames, fragwords, chunks, n_records, n_tail, targets, header)"""
        B = self._bam_blocks(path)
        _check(self.L.strl_front_begin(self.h, B["n_ref"], B["first_off"], n_reads_hint))
        chunks = self._front_push_blocks(B, B["b0"], len(B["blocks"]), chunk_blocks, check_crc)
        return self._front_results(B, chunks)

    def inflate_ms(self):
        """kernel time (ms) of the last inflate_blocks call"""
        ms = C.c_double(0)
        _check(self.L.strl_ctx_inflate_ms(self.h, C.byref(ms)))
        return ms.value

    def cluster_times(self):
        ms = (C.c_double * 3)()
        _check(self.L.strl_ctx_cluster_times(self.h, C.byref(ms)))
        return dict(zip(["cluster_keys_sort_groups", "cluster_sweep", "cluster_bounds"], list(ms)))

    def tail_stream(self):
        """hipStream_t (as an int) the tail of the last extract_device call runs on (strl_ctx_tail_stream)"""
        return int(self.L.strl_ctx_tail_stream(self.h) or 0)

    def treads_device(self):
        """(device pointer of the resident treads, capacity in treads, device pointer of their uint32 count)"""
        p, cap, cnt = C.c_void_p(), C.c_uint64(0), C.c_void_p()
        _check(self.L.strl_ctx_treads_device(self.h, C.byref(p), C.byref(cap), C.byref(cnt)))
        return p.value, cap.value, cnt.value

    def cluster_gathered(self, gathered_ptr, counts_ptr, world, pad, rank, n_tid, window, min_support=5, min_clip=0, min_clip_total=0,
                         max_clip_dist=200, pos_bits=0, mode=MODE_CALL, fetch=True):
        """this rank's share of an all-gathered tread set (strl_cluster_gathered); fetch=False only enqueues the kernels"""
        if not fetch:
            _check(self.L.strl_cluster_gathered(self.h, gathered_ptr, counts_ptr, world, pad, rank, mode, n_tid, pos_bits, window, min_support,
                                                min_clip, min_clip_total, max_clip_dist, None, 0, None, None, 0, None, None))
            return None
        cap = 1 << 16
        while True:
            out = np.zeros(cap, BOUNDS_DTYPE)
            unpl = np.zeros(8192, UNPLACED_DTYPE)
            no, nu = C.c_uint64(0), C.c_uint64(0)
            st = ClusterStats()
            rc = self.L.strl_cluster_gathered(self.h, gathered_ptr, counts_ptr, world, pad, rank, mode, n_tid, pos_bits, window, min_support,
                                              min_clip, min_clip_total, max_clip_dist, out.ctypes.data, cap, C.byref(no), unpl.ctypes.data,
                                              unpl.size, C.byref(nu), C.byref(st))
            if rc == -4 and no.value > cap:
                cap = int(no.value)
                continue
            _check(rc)
            return out[:no.value].copy(), unpl[:nu.value].copy(), st

    # ---- the exchange step inside the library (comm.hip: RCCL over xGMI / device copies) ---------
    def comm_init(self, world, rank, unique_id=None):
        """one process per GPU: join the RCCL communicator `unique_id` (bytes from comm_unique_id() on rank 0) names"""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id) if unique_id is not None else None
        _check(self.L.strl_ctx_comm_init(self.h, world, rank, buf))

    def comm_info(self):
        w, r, u = C.c_int(0), C.c_int(0), C.c_int(0)
        _check(self.L.strl_ctx_comm_info(self.h, C.byref(w), C.byref(r), C.byref(u)))
        return w.value, r.value, bool(u.value)

    def cluster_exchange(self, pad, n_tid, window, min_support=5, min_clip=0, min_clip_total=0, max_clip_dist=200, pos_bits=0, mode=MODE_CALL, fetch=True):
        """all-gather of the resident treads over the context's communicator + clustering of my (tid, unit) groups
        (strl_cluster_exchange); fetch=False only enqueues"""
        if not fetch:
            _check(self.L.strl_cluster_exchange(self.h, pad, mode, n_tid, pos_bits, window, min_support, min_clip, min_clip_total, max_clip_dist,
                                                None, 0, None, None, 0, None, None))
            return None
        cap = 1 << 16
        while True:
            out = np.zeros(cap, BOUNDS_DTYPE)
            unpl = np.zeros(8192, UNPLACED_DTYPE)
            no, nu = C.c_uint64(0), C.c_uint64(0)
            st = ClusterStats()
            rc = self.L.strl_cluster_exchange(self.h, pad, mode, n_tid, pos_bits, window, min_support, min_clip, min_clip_total, max_clip_dist,
                                              out.ctypes.data, cap, C.byref(no), unpl.ctypes.data, unpl.size, C.byref(nu), C.byref(st))
            if rc == -4 and no.value > cap:
                cap = int(no.value)
                continue
            _check(rc)
            return out[:no.value].copy(), unpl[:nu.value].copy(), st

    def exchange_treads(self):
        """all ranks' treads of the last exchange, (rank, .bin) order"""
        n = C.c_uint64(0)
        _check(self.L.strl_exchange_treads(self.h, None, 0, C.byref(n)))
        out = np.zeros(max(1, n.value), TREAD_DTYPE)
        _check(self.L.strl_exchange_treads(self.h, out.ctypes.data, out.size, C.byref(n)))
        return out[:n.value]

    def cluster_collect(self, cap=1 << 16):
        """rows of the last (asynchronous) clustering pass of this context (strl_cluster_collect)"""
        while True:
            out = np.zeros(cap, BOUNDS_DTYPE)
            unpl = np.zeros(8192, UNPLACED_DTYPE)
            no, nu = C.c_uint64(0), C.c_uint64(0)
            st = ClusterStats()
            rc = self.L.strl_cluster_collect(self.h, out.ctypes.data, cap, C.byref(no), unpl.ctypes.data, unpl.size, C.byref(nu), C.byref(st))
            if rc == -4 and no.value > cap:
                cap = int(no.value)
                continue
            _check(rc)
            return out[:no.value].copy(), unpl[:nu.value].copy(), st

    def cluster_members(self, n_bounds):
        """indices (into the tread array of the last cluster() call) of every returned bound's reads, cluster order"""
        off = np.zeros(n_bounds + 1, np.uint64)
        nm = C.c_uint64(0)
        _check(self.L.strl_cluster_members(self.h, off.ctypes.data, None, 0, C.byref(nm)))
        mem = np.zeros(max(1, nm.value), np.uint32)
        _check(self.L.strl_cluster_members(self.h, off.ctypes.data, mem.ctypes.data, mem.size, C.byref(nm)))
        return off, mem[:nm.value]

    def cluster_replay(self):
        """device side of the last cluster() call again, asynchronously (bench)"""
        _check(self.L.strl_cluster_replay(self.h))


def extract_bam_shares(ctxs, path, chunk_blocks=16384, check_crc=True, cut_shift=0):
    """the file in len(ctxs) contiguous shares, one per context (the CLI's `extract --gpus N`): every share begins at a record
    start -- found here by walking the inflated file; the CLI takes them from the .bai --, ends where the next begins
    (strl_front_trim_next on its last chunk, strl_front_tail_bytes == 0 afterwards), the per-read state is gathered on the
    first context (strl_ctxs_extract_gather), which then runs the pair logic.  cut_shift != 0 moves every cut that many bytes
    off its record start: what a stale index does.  -> (result of the first context as extract_bam_device, tail bytes per share)"""
    import struct, zlib
    c0 = ctxs[0]
    B = c0._bam_blocks(path)
    blocks, raw = B["blocks"], B["raw"]
    infl = b"".join(zlib.decompress(raw[po:po + pl], -15) for po, pl, _, _ in blocks)
    ustart = np.concatenate([[0], np.cumsum([b[2] for b in blocks])]).astype(np.int64)
    q = int(ustart[B["b0"]]) + B["first_off"]
    starts = []
    while q + 4 <= len(infl):
        starts.append(q)
        q += 4 + struct.unpack_from("<i", infl, q)[0]
    G = len(ctxs)
    cuts = [starts[0]] + [starts[len(starts) * g // G] + cut_shift for g in range(1, G)] + [None]
    owners, counts, tails = [], [], []
    for g, ctx in enumerate(ctxs):
        a = cuts[g]
        ba = int(np.searchsorted(ustart, a, side="right")) - 1
        _check(ctx.L.strl_ctx_blocking_waits(ctx.h, 1))
        _check(ctx.L.strl_front_begin(ctx.h, B["n_ref"], a - int(ustart[ba]), 0))
        if cuts[g + 1] is None:
            be, trim = len(blocks), 0
        else:
            e = cuts[g + 1]
            bl = int(np.searchsorted(ustart, e, side="right")) - 1
            uo = e - int(ustart[bl])
            be, trim = (bl + 1, blocks[bl][2] - uo) if uo else (bl, 0)
        chunks = ctx._front_push_blocks(B, ba, be, chunk_blocks, check_crc, trim=trim)
        t = C.c_uint32(0)
        _check(ctx.L.strl_front_tail_bytes(ctx.h, C.byref(t)))
        tails.append(int(t.value))
        owners += [g] * len(chunks)
        counts += [c["n_records"] for c in chunks]
        if g == 0:
            all_chunks = list(chunks)
        else:
            all_chunks += chunks
    if any(tails[:-1]):
        return None, tails
    arr = (C.c_void_p * G)(*[c.h for c in ctxs])
    ow, cn = np.array(owners, np.uint32), np.array(counts, np.uint64)
    _check(c0.L.strl_ctxs_extract_gather(arr, G, _ptr(ow), _ptr(cn), len(owners)))
    return c0._front_results(B, all_chunks), tails


def comm_unique_id():
    """128 bytes that name a new RCCL communicator (rank 0 makes them, every rank passes them to Context.comm_init)"""
    L = load()
    buf = (C.c_uint8 * 128)()
    _check(L.strl_comm_unique_id(buf))
    return bytes(buf)


def group_comm_init(ctxs):
    """one process, several contexts: RCCL when they sit on different devices, ordered device copies when they share one"""
    L = load()
    arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    _check(L.strl_ctxs_comm_init(arr, len(ctxs)))


def group_cluster_exchange(ctxs, n_tid, window, min_support=5, min_clip=0, min_clip_total=0, max_clip_dist=200, pos_bits=0, mode=MODE_CALL, pad=0):
    """the exchange step for all contexts of a one-process group (strl_ctxs_cluster_exchange); rows: Context.cluster_collect"""
    L = load()
    arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    _check(L.strl_ctxs_cluster_exchange(arr, len(ctxs), pad, mode, n_tid, pos_bits, window, min_support, min_clip, min_clip_total, max_clip_dist))


def qname_hash(rec):
    """uint64 hash of every record's qname (strl_qname_hash)"""
    rv = _RecView(rec)
    out = np.zeros(max(rv.n, 1), np.uint64)
    _check(load().strl_qname_hash(C.byref(rv.c), out.ctypes.data))
    return out[:rv.n]


def pair_reads(rec, opts, whole, soft, n_tail=-1):
    """strl_pair_reads without a device context (host state machine only). opts = (p, min_mapq, median_fragment_length)"""
    L = load()
    rv = _RecView(rec)
    o = Opts(int(opts[2]), float(opts[0]), int(opts[1]))
    whole = np.ascontiguousarray(whole, np.uint32)
    soft = np.ascontiguousarray(soft, SOFT_DTYPE)
    cap = max(1024, rv.n // 4)
    while True:
        out = np.zeros(cap, TREAD_DTYPE)
        no = C.c_uint64(0)
        rc = L.strl_pair_reads(C.byref(rv.c), C.byref(o), _ptr(whole), _ptr(soft), soft.size, n_tail, out.ctypes.data, cap, C.byref(no))
        if rc == -4 and no.value > cap:
            cap = int(no.value)
            continue
        _check(rc)
        return out[:no.value].copy()


def pair_rule(ctx, op, A, B, opts, B_position=0):
    """one pair rule on single treads (strl_pair_rule): ctx = a Context (device code) or None (host twin).
    op 0 adjust_by, 1 unplaced_pair, 2 canonical_repeat; opts = (p, min_mapq, median_fragment_length) -> (result, A after)"""
    a = np.ascontiguousarray(A, TREAD_DTYPE).reshape(1).copy()
    b = np.ascontiguousarray(B, TREAD_DTYPE).reshape(1).copy()
    o = Opts(int(opts[2]), float(opts[0]), int(opts[1]))
    res = C.c_int(0)
    _check(load().strl_pair_rule(ctx.h if ctx is not None else None, op, a.ctypes.data, b.ctypes.data, C.byref(o), int(B_position), C.byref(res)))
    return res.value, a[0]


def _noop():
    pass


def frag_median(frag, pct=0.5):
    frag = np.ascontiguousarray(frag, np.uint32)
    return load().strl_frag_median(frag.ctypes.data, pct)


GROUP_KEY_DTYPE = np.dtype([("tid", "<i4"), ("repeat", "S8")])


def group_order(treads, mode):
    """[(tid, unit bytes)] of the (tid, unit) groups in the reference's Table iteration order (strl_group_order)"""
    L = load()
    t = np.ascontiguousarray(treads, TREAD_DTYPE)
    ng = C.c_uint64(0)
    _check(L.strl_group_order(_ptr(t), t.size, mode, None, 0, C.byref(ng)))
    out = np.zeros(max(1, ng.value), GROUP_KEY_DTYPE)
    _check(L.strl_group_order(_ptr(t), t.size, mode, out.ctypes.data, out.size, C.byref(ng)))
    return [(int(k["tid"]), bytes(k["repeat"])) for k in out[:ng.value]]


def assign_reads_loci(treads, loci, mode):
    """assign_reads_locus (callclusters.nim:14-50) for every locus in order.  Returns (treads with taken ones marked,
    loci with recounted reads, [index arrays of the reads each locus received])."""
    L = load()
    t = np.ascontiguousarray(treads, TREAD_DTYPE).copy()
    lo = np.ascontiguousarray(loci, LOCUS_DTYPE).copy()
    off = np.zeros(lo.size + 1, np.uint64)
    idx = np.zeros(max(1, t.size), np.uint32)
    _check(L.strl_assign_reads_loci(t.ctypes.data, t.size, mode, lo.ctypes.data, lo.size, off.ctypes.data, idx.ctypes.data, idx.size))
    return t, lo, [idx[int(off[j]):int(off[j + 1])].copy() for j in range(lo.size)]


def spanners(rec: RecordBatch, bound, window, frag, min_mapq=20):
    """collect.nim:132-182 over the records of `rec` -> (support array, median_depth, expected_spanners)"""
    L = load()
    rv = _RecView(rec)
    isz = np.ascontiguousarray(rec.isize if rec.isize is not None else np.zeros(rec.n), np.int32)
    frag = np.ascontiguousarray(frag, np.uint32)
    bb = np.ascontiguousarray(bound, BOUNDS_DTYPE).reshape(1)
    cap = 2 * rv.n + 16
    out = np.zeros(cap, SUPPORT_DTYPE)
    sm = SpanSummary()
    _check(L.strl_spanners(C.byref(rv.c), isz.ctypes.data, bb.ctypes.data, window, frag.ctypes.data, min_mapq, out.ctypes.data, cap, C.byref(sm)))
    return out[:sm.n_support].copy(), sm.median_depth, sm.expected_spanners


def genotype(bound, members, qname_off, qnames, supports, depth, median_fragment_length, min_support=5, min_clip=0, min_clip_total=0):
    L = load()
    bb = np.ascontiguousarray(bound, BOUNDS_DTYPE).reshape(1)
    m = np.ascontiguousarray(members, TREAD_DTYPE)
    sp = np.ascontiguousarray(supports, SUPPORT_DTYPE)
    qo = np.ascontiguousarray(qname_off, np.uint64)
    qn = np.frombuffer(bytes(qnames) + b"\0", dtype=np.uint8)
    out = np.zeros(1, CALL_DTYPE)
    o = CallOpts(median_fragment_length, min_support, min_clip, min_clip_total)
    _check(L.strl_genotype(bb.ctypes.data, m.ctypes.data, m.size, qo.ctypes.data, qn.ctypes.data, sp.ctypes.data, sp.size, C.byref(o), float(depth),
                           out.ctypes.data))
    return out[0]


def calls_finish(calls, unplaced):
    """percentiles + output order (call.nim:38-48,264-276) -> (calls, order)"""
    L = load()
    c = np.ascontiguousarray(calls, CALL_DTYPE).copy()
    u = np.ascontiguousarray(unplaced, UNPLACED_DTYPE)
    order = np.zeros(max(1, c.size), np.uint64)
    _check(L.strl_calls_finish(c.ctypes.data, c.size, u.ctypes.data, u.size, order.ctypes.data))
    return c, order[:c.size]


def call_row(call, chrom):
    buf = C.create_string_buffer(1024)
    cc = np.ascontiguousarray(call, CALL_DTYPE).reshape(1)
    load().strl_call_row(buf, 1024, cc.ctypes.data, chrom.encode())
    return buf.value.decode()


def bounds_row(b, chrom):
    buf = C.create_string_buffer(512)
    bb = np.ascontiguousarray(b, BOUNDS_DTYPE).reshape(1)
    load().strl_bounds_row(buf, 512, bb.ctypes.data, chrom.encode())
    return buf.value.decode()


def bin_write(path, proportion_repeat, min_mapq, frag, sam_header, treads, qname_off, qnames):
    t = np.ascontiguousarray(treads, TREAD_DTYPE)
    frag = np.ascontiguousarray(frag, np.uint32)
    qo = np.ascontiguousarray(qname_off, np.uint64)
    hdr = sam_header.encode() if isinstance(sam_header, str) else sam_header
    _check(load().strl_bin_write(path.encode(), proportion_repeat, min_mapq, frag.ctypes.data, hdr, len(hdr), _ptr(t), t.size,
                                 qo.ctypes.data, bytes(qnames) + b"\0"))


def bin_read(path):
    L = load()
    info = BinInfo()
    _check(L.strl_bin_read(path.encode(), C.byref(info), None, None, None, None))
    hdr = C.create_string_buffer(max(1, info.header_len))
    t = np.zeros(max(1, info.n_reads), TREAD_DTYPE)
    qo = np.zeros(info.n_reads + 1, np.uint64)
    qn = C.create_string_buffer(max(1, int(info.qnames_bytes)))
    _check(L.strl_bin_read(path.encode(), C.byref(info), hdr, t.ctypes.data, qo.ctypes.data, qn))
    return dict(proportion_repeat=info.proportion_repeat, min_mapq=info.min_mapq, frag=np.array(info.frag, np.uint32),
                header=hdr.raw[:info.header_len].decode(), treads=t[:info.n_reads].copy(), qname_off=qo,
                qnames=qn.raw[:int(info.qnames_bytes)])
