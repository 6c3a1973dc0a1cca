#!/usr/bin/env python3
"""Compose profiles/rNN_pmc_roi_align.json from the per-kernel counter tables tools/gpu_profiles.sh wrote.
    python tools/make_pmc_json.py TAG fwd.json bwd.json [nhwc.json] > profiles/TAG_pmc_roi_align.json
HBM bytes per call = sum over the call's kernels of (2 * FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE is doubled as
MI355X_MICROARCH.md prescribes for gfx950 (it reports half the bytes of wide coalesced reads); WRITE_SIZE as reported."""
import json
import sys

tag = sys.argv[1]
KEEP = ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum", "TA_BUSY_avr", "TA_BUSY_max",
        "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
        "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE")


def section(path):
    table = json.load(open(path))
    kernels, total_kb = {}, 0.0
    for name, counters in sorted(table.items()):
        if not name.startswith("roi_align"):
            continue
        kernels[name] = {k: counters[k] for k in KEEP if k in counters}
        total_kb += 2 * counters.get("FETCH_SIZE", 0.0) + counters.get("WRITE_SIZE", 0.0)
    return {"kernels": kernels, "hbm_bytes_per_call": int(total_kb * 1024)}


doc = {
    "source": "rocprofv3 --kernel-trace --pmc <group> -- python tools/run_one_kernel.py roi_align_fwd|roi_align_bwd 5 "
              "(tools/gpu_profiles.sh pmc, %s; one pass per counter group; per-kernel averages over the 5 launches)" % tag,
    "corrections": "FETCH_SIZE doubled (gfx950: reports 1/2 of the bytes of wide coalesced reads, MI355X_MICROARCH.md "
                   "section HBM); WRITE_SIZE as reported (calibrated on a 68.8 MB torch fill: 67200 KB); units KB = 1024 B",
    "shape": "R=512 C=256 7x7 sr=2 on 200x336 (config 2)",
    "forward": section(sys.argv[2]),
    "backward": section(sys.argv[3]),
}
if len(sys.argv) > 4:
    doc["forward_channels_last"] = section(sys.argv[4])
json.dump(doc, sys.stdout, indent=1)
print()
