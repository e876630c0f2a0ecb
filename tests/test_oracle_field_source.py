"""CPU: oracle/undulator_np.py:trajectory (the restatement of the reference's
SourceFromField._build_trajectory_conv) against the Runge-Kutta tables the reference itself
produced (golden G13, oracle/gen_fixtures_field_source.py)."""
import os

import numpy as np
import pytest

from oracle import undulator_np as un


@pytest.mark.parametrize('tag', ['plain', 'filament'])
def test_trajectory_restatement_is_the_references(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, 'g13_trajectory_%s.npz' % tag))
    gamma = float(g['gamma']) if int(g['filament']) else None
    betax, betay, betam, trajx, trajy, trajz = un.trajectory(
        g['wtGrid'], g['Bx'], g['By'], g['Bz'], gamma)
    for mine, name in ((betax, 'betax'), (betay, 'betay'), (trajx, 'trajx'),
                       (trajy, 'trajy'), (trajz, 'trajz')):
        assert np.array_equal(mine, g[name]), name
    assert betam == float(g['betam'])
    # a wiggling electron that comes back: zero mean velocity and position
    assert abs(np.trapezoid(g['betax'], g['wtGrid'])) < 1e-9 * np.abs(g['betax']).max() * 400
