"""CPU: the numpy restatement of the undulator field sums
(oracle/undulator_np.py) against golden vectors produced by the imported
reference (oracle/gen_fixtures_undulator.py -> tests/golden/g9_undulator_*.npz:
outputs of Undulator._sp_sum and Undulator._build_I_map_conv)."""
import os

import numpy as np
import pytest

from oracle import undulator_np as un

CASES = ['far_planar', 'far_helical', 'taper', 'nf']


def load(golden_dir, tag):
    return np.load(os.path.join(golden_dir, 'g9_undulator_%s.npz' % tag))


def tables_of(g):
    return {k: g[k] for k in ('tg', 'ag', 'sintg', 'costg', 'sintgph', 'costgph')} | \
        {'dstep': float(g['dstep'])}


def nan_to_none(v):
    v = float(v)
    return None if np.isnan(v) else v


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.mark.parametrize('tag', CASES)
def test_node_tables(golden_dir, tag):
    g = load(golden_dir, tag)
    tab = un.node_tables(int(g['quadm']), int(g['gIntervals']), float(g['phase']))
    for k in ('tg', 'ag', 'sintg', 'costg', 'sintgph', 'costgph'):
        assert np.allclose(tab[k], g[k], rtol=0, atol=4e-16), k
    assert tab['dstep'] == float(g['dstep'])


@pytest.mark.parametrize('tag', CASES)
def test_raw_sums_match_reference(golden_dir, tag):
    g = load(golden_dir, tag)
    Is, Ip = un.sp_sum(int(g['mode']), float(g['Kx']), float(g['Ky']), int(g['Np']),
                       tables_of(g), g['ww1'], g['w'], g['wu'], g['gamma'],
                       g['ddphi'], g['ddpsi'], nan_to_none(g['taperVal']),
                       float(g['r0z']))
    assert rel(Is, g['Is']) < 1e-13
    assert rel(Ip, g['Ip']) < 1e-13


@pytest.mark.parametrize('tag', CASES)
def test_scaled_map_matches_reference(golden_dir, tag):
    g = load(golden_dir, tag)
    I, Es, Ep = un.intensity_map(
        int(g['mode']), float(g['Kx']), float(g['Ky']), int(g['Np']), float(g['L0']),
        float(g['gamma0']), float(g['eI']), True, tables_of(g), g['w'], g['ddphi'],
        g['ddpsi'], nan_to_none(g['taperVal']), nan_to_none(g['R0']))
    assert rel(I, g['I']) < 1e-13
    assert rel(Es, g['Es']) < 1e-13
    assert rel(Ep, g['Ep']) < 1e-13


def test_planar_on_axis_has_no_vertical_field(golden_dir):
    g = load(golden_dir, 'far_planar')
    on_axis = (g['ddphi'] == 0) & (g['ddpsi'] == 0)
    assert on_axis.sum() >= 2
    assert np.all(np.abs(g['Ip'][on_axis]) <= 1e-12 * np.abs(g['Is'][on_axis]))


@pytest.mark.parametrize('tag', ['far', 'filament', 'nf'])
def test_custom_field_sums_match_reference(golden_dir, tag):
    """SourceFromField._sp_sum (tabulated field) restated in
    undulator_np.custom_sp_sum."""
    g = np.load(os.path.join(golden_dir, 'g12_custom_field_%s.npz' % tag))
    tab = {k: g[k] for k in ('tg', 'ag', 'Bx', 'By', 'Bz', 'betax', 'betay', 'trajx',
                             'trajy', 'trajz')}
    Is, Ip = un.custom_sp_sum(bool(g['filament']), tab, g['emcg'], g['w'], g['gamma'],
                              g['ddphi'], g['ddpsi'], float(g['betam']),
                              nan_to_none(g['R0']))
    assert rel(Is, g['Is']) < 1e-13 and rel(Ip, g['Ip']) < 1e-13
