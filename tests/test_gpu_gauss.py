"""GaussianBeam / LaguerreGaussianBeam / HermiteGaussianBeam .shine(wave=...) against the
reference's fields on the same screen meshes (golden G16, oracle/gen_fixtures_gauss.py).
The carrier phase k y is ~2e11 rad at 5 m: matching the complex field to 1e-9 means the
device rounds the phase terms in numpy's order."""
import os

import numpy as np
import pytest

import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.screens as rsc
import xrt_amd.backends.raycing.sources as rs
from oracle.gen_fixtures_gauss import CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_gaussian_modes_match_reference(golden_dir, case):
    tag, cls, kw, dist = case
    g = np.load(os.path.join(golden_dir, 'g16_gaussian_beams.npz'))
    bl = raycing.BeamLine(azimuth=float(g['azimuth']))
    src = getattr(rs, cls)(bl, tag, **kw)
    scr = rsc.Screen(bl, 'fsm', [np.sin(0.01)*dist, np.cos(0.01)*dist, 0])
    wave = scr.prepare_wave(src, g[tag + '_x'], g[tag + '_z'])
    np.random.seed(int(g[tag + '_seed']))
    bo = src.shine(wave=wave)
    scale = np.abs(g[tag + '_wave_Es']).max() + np.abs(g[tag + '_wave_Ep']).max()
    for mine, ref in ((wave.Es, g[tag + '_wave_Es']), (wave.Ep, g[tag + '_wave_Ep'])):
        assert np.abs(mine - ref).max() <= 1e-9 * scale, tag
    for mine, ref in zip((wave.Jss, wave.Jpp), g[tag + '_wave_J']):
        assert np.abs(mine - ref).max() <= 1e-12 * g[tag + '_wave_J'].max(), tag
    for mine, ref in zip((wave.a, wave.b, wave.c), g[tag + '_wave_abc']):
        assert np.abs(mine - ref).max() <= 1e-14, tag
    for f in ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp', 'Jsp', 'Es', 'Ep'):
        ref = g['%s_bo_%s' % (tag, f)]
        assert np.abs(getattr(bo, f) - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-300), \
            (tag, f)
    assert np.array_equal(bo.state, g[tag + '_bo_state'])
    if tag + '_sourceWeight' in g.files:
        assert abs(wave.sourceWeight / float(g[tag + '_sourceWeight']) - 1) < 1e-12
    if tag in ('plain', 'waist', 'lg11', 'hg21'):
        assert abs((wave.Jss + wave.Jpp).sum() - 1) < 1e-3      # modes of unit flux


def test_gaussian_beam_needs_a_wave():
    bl = raycing.BeamLine()
    src = rs.GaussianBeam(bl, 'g', w0=0.02)
    with pytest.raises(ValueError):
        src.shine()
    with pytest.raises(ValueError):
        rs.LaguerreGaussianBeam(bl, 'lg', w0=(0.01, 0.02), vortex=(1, 0))
    assert abs(src.w(src.rayleigh_range(9000.), E=9000.) / 0.02 - 2**0.5) < 1e-15
