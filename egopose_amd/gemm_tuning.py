"""Pre-tuned library GEMMs for the PPO update (rocBLAS / hipBLASLt picks through torch's TunableOp).

The update's GEMMs are plain library calls (nets.py): `(N, 243) x (243, 300)` and friends with N = all steps of the
batch. hipBLASLt's default heuristic leaves 35 % on the table at these skinny shapes (tools/gemm_probe.py: 2.39 ms ->
1.52 ms for the three layers of one MLP, forward + both gradients, at N = 134 k). TunableOp picks per exact shape,
and N changes with every rollout, so the update pads its row counts to multiples of `ROW_BUCKET`
(`nets.bucket_rows`; pad rows are zeros in, sliced off before the loss, exact zeros in every gradient) and
`assets/tunableop/gfx950.csv` holds the tuned picks of the buckets the shipped configs reach
(`tools/tune_update.py` regenerates it on an MI355X). Shapes the file does not know run the default pick.

`EGP_TUNED_GEMMS=0` switches the whole mechanism off (no padding, no TunableOp).
"""
from __future__ import annotations

import os
import tempfile

ROW_BUCKET = 8192
EPISODE_BUCKET = 64
_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "tunableop", "gfx950.csv")
_state = {"on": False}


def tuned_file():
    return _FILE


def enabled():
    return _state["on"]


def enable(tune=False, out_file=None):
    """Pad update batches to shape buckets and route torch's GEMMs through TunableOp with the shipped picks.
    `tune=True` (tools/tune_update.py) times every candidate for new shapes and records the winner in `out_file`."""
    import torch
    if os.environ.get("EGP_TUNED_GEMMS", "1") == "0" or not torch.cuda.is_available():
        return False
    import torch.cuda.tunable as tn
    path = tuned_file()
    tn.enable(True)
    tn.record_untuned_enable(False)
    if tune:
        tn.set_filename(out_file or path)
        tn.tuning_enable(True)
        tn.set_max_tuning_duration(40)
        tn.set_max_tuning_iterations(30)
        if os.path.exists(path):
            tn.read_file(path)
    else:
        tn.tuning_enable(False)
        tn.set_filename(os.path.join(tempfile.gettempdir(), "egp_tunableop_%d.csv" % os.getpid()))   # nothing new to write
        if not os.path.exists(path) or not tn.read_file(path):
            tn.enable(False)           # no picks for this build of the libraries: default heuristics, no padding
            return False
    _state["on"] = True
    return True


def disable():
    import torch.cuda.tunable as tn
    tn.enable(False)
    _state["on"] = False


def pad_to(n, bucket):
    return (-int(n)) % bucket
