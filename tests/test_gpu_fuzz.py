"""A slice of the randomised differential test (scripts/fuzz_parity.py: random domain x simulator x belief x
sizes x modes, engine against oracle on the same Philox streams, every trace field bit-equal)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_random_configurations_match_the_oracle():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", path)
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    bad, refused = fuzz.run(250, seed=7, verbose=False)
    assert bad == 0
    assert refused < 60          # (model / domain pairs neither side supports)
