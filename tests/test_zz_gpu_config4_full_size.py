"""The longest test of the -m gpu suite, in a file of its own so that it runs LAST (pytest takes the files in name order): BASELINE config 4 at FULL size — 10 000 nodes x 110 000
pods, allocate + consolidation + reclaim on one session with queueDepthPerAction 8 for the victim actions — hashed against the oracle's end-to-end run of the same cycle
(profiles/full_size_pins.json C4_100pct_depth8; the oracle took 8 197 s).  About 270 s on one MI355X: the victim search of the sequential engine.  A slower box
that runs into the driver's limit does so here, after every other test has reported."""
import pytest

from test_gpu_parity import gpu, test_gpu_config4_cycle_hashes_to_the_oracles as _hash_test  # noqa: F401  (the fixture; the 10 % / 30 % cases run in their own file)

pytestmark = pytest.mark.gpu


def test_gpu_config4_at_full_size_hashes_to_the_oracles(gpu):
    _hash_test(gpu, "C4_100pct_depth8", 1.0, 8)
