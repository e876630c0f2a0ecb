"""GPU: exactness of the arithmetic building blocks the ray-state path relies on."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('b', [53.33319111122489, 3333342.222238815, 40., 3000.,
                               1.0000000000000002, 1.9999999999999998, 3., 7e-5,
                               123456.789e20])
def test_constant_divisor_division_is_ieee(b):
    """a/b via (a*y, fma remainder, fma correction) with y = RN(1/b) must equal the
    IEEE quotient - it decides ray states (toroid x/r and y^2/2/R)."""
    from xrt_amd import hipcalls
    rng = np.random.default_rng(int(b * 1000) % 2**31)
    a = np.concatenate([rng.normal(0, 1, 2_000_000) * 10.0 ** rng.integers(-8, 8, 2_000_000),
                        rng.uniform(-300, 300, 1_000_000), np.array([0., -0., b, -b, 1.])])
    q = hipcalls.debug_divconst(torch.from_numpy(a).cuda(), b).cpu().numpy()
    assert np.array_equal(q, a / b)
