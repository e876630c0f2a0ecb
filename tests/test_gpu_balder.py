"""End to end on the reference's own example beamline, Balder (its run_process is
the caller of the ray path): the same rays from the wiggler (numpy seeded alike), every
element of the chain bit-exact in ray states, positions at 1e-12, flux at 1e-9 -- against the
beams the reference produced for its example (golden g17_balder_chain,
oracle/gen_fixtures_balder.py)."""
import os

import numpy as np
import pytest

import balder_case
from oracle.gen_fixtures_balder import BEAMS

pytestmark = pytest.mark.gpu


def test_balder_example_chain_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g17_balder_chain.npz'))
    bl, seed = balder_case.build(g)
    np.random.seed(seed)
    beams = balder_case.trace(bl)
    for name in BEAMS:
        b = beams[name]
        assert np.array_equal(b.state, g[name + '_state']), name
        seen = g[name + '_state'] > 0 if name != 'beamFSM0' else np.ones(len(b.x), bool)
        for f in ('x', 'y', 'z', 'a', 'c', 'E'):
            ref = g['%s_%s' % (name, f)]
            scale = max(np.abs(ref[seen]).max(), 1e-300)
            assert np.abs(getattr(b, f) - ref)[seen].max() <= 1e-12 * scale, (name, f)
        scale = (g[name + '_Jss'] + g[name + '_Jpp']).max()
        for f in ('Jss', 'Jpp'):
            assert np.abs(getattr(b, f) - g['%s_%s' % (name, f)]).max() <= 1e-9 * scale, \
                (name, f)
    good = g['beamFSMSample_state'] == 1
    flux = (beams['beamFSMSample'].Jss + beams['beamFSMSample'].Jpp)[good].sum()
    assert abs(flux / (g['beamFSMSample_Jss'] + g['beamFSMSample_Jpp'])[good].sum() - 1) < 1e-9
    # the focusing mirror does focus: the image at the sample is smaller than at the slit
    assert beams['beamFSMSample'].z[good].std() < beams['beamSlitEHLocal'].z[good].std() * 1.5
