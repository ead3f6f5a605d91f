"""ctypes loader for csrc/libpwr.so (the C ABI of include/pwr.h).  There is no fallback: if the
HIP library is missing the import of the product path fails loudly."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpwr.so")


class PwrStats(ctypes.Structure):
    _fields_ = [("cells_reference", ctypes.c_uint64), ("cells_computed", ctypes.c_uint64),
                ("fill_launches", ctypes.c_uint64), ("fill_ms", ctypes.c_double),
                ("rows_committed", ctypes.c_uint64), ("rows_recomputed", ctypes.c_uint64),
                ("batches", ctypes.c_uint64), ("rows_changed", ctypes.c_uint64),
                ("reject_reason", ctypes.c_uint64 * 4), ("fill_launches_timed", ctypes.c_uint64),
                ("stalls", ctypes.c_uint64), ("rows_ahead", ctypes.c_uint64), ("rows_wide", ctypes.c_uint64),
                ("seg_jobs", ctypes.c_uint64), ("segs", ctypes.c_uint64), ("seg_fails", ctypes.c_uint64),
                ("rows_jumped", ctypes.c_uint64)]


# every symbol include/pwr.h declares
EXPORTS = ["pwr_create", "pwr_destroy", "pwr_trim_ends", "pwr_realign_row", "pwr_realign_round", "pwr_realign_rows",
           "pwr_total_score", "pwr_dims", "pwr_export_rows", "pwr_set_option", "pwr_get_option", "pwr_get_stats",
           "pwr_reset_stats", "pwr_strerror", "pwr_device_count", "pwr_read_msa_file",
           "pwr_write_msa_file", "pwr_run_file", "pwr_split_begin", "pwr_split_slot_bytes", "pwr_split_stage", "pwr_split_commit",
           "pwr_snapshot_begin", "pwr_snapshot_wait", "pwr_snapshot_free"]

# every symbol include/pia.h declares (the InitialAligner)
PIA_EXPORTS = ["pia_create", "pia_destroy", "pia_align", "pia_get_stats", "pia_set_option", "pia_get_timing", "pia_read_template", "pia_read_fasta",
               "pia_build_msa", "pia_run_files"]

# every symbol include/pmc.h declares (MaxCorrelation)
PMC_EXPORTS = ["pmc_maxcorrs", "pmc_last_timing", "pmc_read_msa", "pmc_write", "pmc_run_file"]

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C repeatresolver_amd/csrc`.  There is no CPU fallback for the product path.")
    # Load order matters in a process that also uses torch: torch brings its own copy of the HIP runtime, libpwr.so is linked
    # against the system's, and the dynamic loader gives the process whichever copy came first under that name.  With the
    # system's first, torch.cuda later fails to initialise ("No HIP GPUs are available" -- found when a test module that touches
    # torch ran after modules that had loaded libpwr.so); with torch's first both work.  So torch goes first when it is there.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.pwr_create.restype = ci
    lib.pwr_create.argtypes = [ctypes.POINTER(vp), ci, ci, ctypes.c_char_p, ci, ci]
    lib.pwr_destroy.restype = None
    lib.pwr_destroy.argtypes = [vp]
    for n in ("pwr_trim_ends", "pwr_realign_round", "pwr_reset_stats"):
        getattr(lib, n).restype = ci
        getattr(lib, n).argtypes = [vp]
    lib.pwr_realign_row.restype = ci
    lib.pwr_realign_row.argtypes = [vp, ci]
    lib.pwr_realign_rows.restype = ci
    lib.pwr_realign_rows.argtypes = [vp, ci, ci]
    lib.pwr_total_score.restype = ci
    lib.pwr_total_score.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    lib.pwr_dims.restype = ci
    lib.pwr_dims.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci)]
    lib.pwr_export_rows.restype = ci
    lib.pwr_export_rows.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t]
    lib.pwr_set_option.restype = ci
    lib.pwr_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_long]
    lib.pwr_get_option.restype = ci
    lib.pwr_get_option.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_long)]
    lib.pwr_get_stats.restype = ci
    lib.pwr_get_stats.argtypes = [vp, ctypes.POINTER(PwrStats)]
    lib.pwr_strerror.restype = ctypes.c_char_p
    lib.pwr_strerror.argtypes = [ci]
    lib.pwr_device_count.restype = ci
    lib.pwr_device_count.argtypes = []
    lib.pwr_split_begin.restype = ci
    lib.pwr_split_begin.argtypes = [vp, ci, ci, ci, ci]
    lib.pwr_split_slot_bytes.restype = ci
    lib.pwr_split_slot_bytes.argtypes = [vp, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ci)]
    lib.pwr_split_stage.restype = ci
    lib.pwr_split_stage.argtypes = [vp, vp]
    lib.pwr_split_commit.restype = ci
    lib.pwr_split_commit.argtypes = [vp, vp, ctypes.POINTER(ci)]
    if hasattr(lib, "pwr_snapshot_begin"):
        lib.pwr_snapshot_begin.restype = ci
        lib.pwr_snapshot_begin.argtypes = [vp, ctypes.POINTER(vp)]
        lib.pwr_snapshot_wait.restype = ci
        lib.pwr_snapshot_wait.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ci), ctypes.POINTER(ci)]
        lib.pwr_snapshot_free.restype = None
        lib.pwr_snapshot_free.argtypes = [vp]
    lib.pwr_debug_last_job.restype = ci
    lib.pwr_debug_last_job.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci),
                                       ctypes.POINTER(ci), ctypes.POINTER(ci), ci]
    lib.pwr_debug_fill_clock.restype = ci
    lib.pwr_debug_fill_clock.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    lib.pwr_debug_row_columns.restype = ci
    lib.pwr_debug_row_columns.argtypes = [vp, ci, ctypes.POINTER(ci), ci]
    lib.pwr_debug_rounds.restype = ci
    lib.pwr_debug_rounds.argtypes = [vp]
    ll = ctypes.c_longlong
    lib.pia_create.restype = ci
    lib.pia_create.argtypes = [ctypes.POINTER(vp), ctypes.c_char_p, ci, ci]
    lib.pia_destroy.restype = None
    lib.pia_destroy.argtypes = [vp]
    lib.pia_align.restype = ci
    lib.pia_align.argtypes = [vp, ci, ctypes.c_char_p, ctypes.POINTER(ll), ctypes.POINTER(ci), ctypes.POINTER(ci)]
    lib.pia_get_stats.restype = ci
    lib.pia_get_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double)]
    lib.pia_set_option.restype = ci
    lib.pia_set_option.argtypes = [vp, ctypes.c_char_p, ll]
    lib.pia_get_timing.restype = ci
    lib.pia_get_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
    lib.pia_read_template.restype = ci
    lib.pia_read_template.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ci)]
    lib.pia_read_fasta.restype = ci
    lib.pia_read_fasta.argtypes = [ctypes.c_char_p, ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]
    lib.pia_build_msa.restype = ci
    lib.pia_build_msa.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ci, ctypes.c_char_p, ctypes.POINTER(ll), ctypes.POINTER(ci),
                                  ctypes.POINTER(ci), ctypes.c_double, ci]
    lib.pmc_maxcorrs.restype = ci
    lib.pmc_maxcorrs.argtypes = [ci, ci, ctypes.c_char_p, ci, ci, ctypes.POINTER(ctypes.c_double)]
    lib.pmc_last_timing.restype = ci
    lib.pmc_last_timing.argtypes = [ctypes.POINTER(ctypes.c_double)]
    lib.pmc_read_msa.restype = ci
    lib.pmc_read_msa.argtypes = [ctypes.c_char_p, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_void_p)]
    lib.pmc_write.restype = ci
    lib.pmc_write.argtypes = [ctypes.c_char_p, ci, ctypes.POINTER(ctypes.c_double)]
    _lib = lib
    return lib
