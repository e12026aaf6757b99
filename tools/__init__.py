"""Development tools (GPU box).  The shipped library reads nothing from the environment; the tools translate their
CMR_<OPTION> environment variables into cmr_index_set_option calls (`options=env_options()`).  `wide_abl` exists only in a
development build: CMR_EXTRA_HIPCC_FLAGS=-DCMR_DEV_KNOBS CMR_BUILD_LIB=/path/libdev.so python -m comorag_amd.build, then
COMORAG_HIP_LIB=/path/libdev.so."""
import os

OPTION_NAMES = ("scan_ring", "scan_asm_ring", "scan_grid", "scan_no_sample", "scan_no_wide", "scan_no_tiny", "scan_no_small", "small_max_panels",
                "tiny_multi", "zero_copy", "sample_single", "sample_tau_in_scan", "sample_div", "sample_maxmul", "pipe_reserve_cus", "pipe_slots", "wide_waves",
                "pipe_dual_scan", "pipe_cu_mask", "wide_abl", "wide_mode", "stream_nt", "sample_single_max", "scan_fin", "scan_fin_queries", "scan_fin_dense", "scan_fin_spin", "sync_poll", "scan_fin_suppliers", "scan_fin_cap")


def env_options() -> dict:
    return {n: int(os.environ["CMR_" + n.upper()]) for n in OPTION_NAMES if os.environ.get("CMR_" + n.upper(), "") != ""}
