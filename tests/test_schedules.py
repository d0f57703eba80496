"""Schedules: the reference's own unit vectors (rl_coach/tests/test_schedules.py) and, when
/root/reference is present (build container only), a step-by-step bit-exact comparison against the
reference classes themselves."""
import numpy as np
import pytest

from coach_amd.schedules import ConstantSchedule, ExponentialSchedule, LinearSchedule


def test_constant_and_linear_reference_vectors():
    # rl_coach/tests/test_schedules.py: constant stays; linear reaches its end and stays there
    s = ConstantSchedule(0.3)
    for _ in range(10):
        s.step()
    assert s.current_value == 0.3
    s = LinearSchedule(1, 3, 10)                          # increasing
    vals = []
    for _ in range(15):
        s.step()
        vals.append(float(s.current_value))
    np.testing.assert_allclose(vals[:10], np.linspace(1.2, 3.0, 10))
    assert vals[10:] == [3.0] * 5
    s = LinearSchedule(1.0, 0.1, 4)                       # decreasing
    for _ in range(6):
        s.step()
    assert abs(float(s.current_value) - 0.1) < 1e-15
    s = LinearSchedule(0.1, 0.1, 50000)                   # AdditiveNoiseParameters default: constant
    s.step()
    assert s.current_value == 0.1


def test_exponential_reference_behaviour():
    s = ExponentialSchedule(10, 3, 0.99)
    for _ in range(1000):
        s.step()
    assert s.current_value == 3 and s.current_step == 1000
    with pytest.raises(ValueError):
        ExponentialSchedule(1, 2, 0.9)
    with pytest.raises(ValueError):
        ExponentialSchedule(2, 1, 1.1)


@pytest.mark.reference
def test_bit_exact_against_reference_classes():
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import _refstub
    _refstub.install()
    from rl_coach import schedules as R
    for args in [(1.0, 0.1, 1000), (0.4, 1.0, 333), (1, 0.01, 10000), (1.0, 0, 1000000)]:
        x, y = LinearSchedule(*args), R.LinearSchedule(*args)
        for i in range(12000):
            x.step(); y.step()
            assert float(x.current_value) == float(y.current_value), (args, i)
    x, y = ExponentialSchedule(1.0, 0.05, 0.995), R.ExponentialSchedule(1.0, 0.05, 0.995)
    for i in range(2000):
        x.step(); y.step()
        assert float(x.current_value) == float(y.current_value)
