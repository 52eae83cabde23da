"""A short run of the differential fuzzer (tests/fuzz_gpu.py) in the GPU tier: random parameter set, LWE dimension, batch size
around the dispatch boundaries, entry point (host / device pointers, gates / MUX / programmable bootstraps / blind rotate + key
switch) and dispatch options, against the C oracle -- every word identical at the exact sets, decryption + phase at the Uint
sets.  Long runs are under profiles/ (r05_n_fuzz_*.txt)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_differential_fuzz_against_the_oracle(seed):
    import fuzz_gpu
    lines = []
    cases, stats = fuzz_gpu.run(12.0, seed, lines.append)
    assert cases >= 20, lines[-3:]
    assert len(stats) >= 4, stats            # several entry points were drawn
