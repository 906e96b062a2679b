"""A short randomised parity sweep in the GPU suite (tests/fuzz_blend.py; longer sweeps: python tests/fuzz_blend.py 1500 7)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_random_blend_and_assessment_cases():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fuzz_blend
    assert fuzz_blend.run(120, 3) == 0


def test_random_colour_correction_cases():
    """tests/fuzz_adjust.py: LUT map and both guided-filter branches on random sizes / channels / tables, in place or not."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fuzz_adjust
    assert fuzz_adjust.run(40, 5) == 0
