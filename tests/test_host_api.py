"""CPU: host-side behaviour the reference's users rely on (no GPU work)."""
import os

import numpy as np

import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.apertures as ra
import xrt_amd.backends.raycing.sources as rs


def test_aperture_opening_and_kind_can_be_assigned():
    """`slit.opening = [...]` in a scan (reference apertures.py:90-131): the blades and the
    optical limits prepare_wave samples from follow."""
    bl = raycing.BeamLine()
    slit = ra.RectangularAperture(bl, 'slit', [0, 100., 0], ('left', 'right', 'bottom', 'top'),
                                  [-1., 1., -2., 2.])
    assert slit.limOptX == [-1., 1.] and slit.limOptY == [-2., 2.]
    slit.opening = [-0.5, 0.25, -0.1, 0.3]
    assert slit.opening == [-0.5, 0.25, -0.1, 0.3]
    assert slit.limOptX == [-0.5, 0.25] and slit.limOptY == [-0.1, 0.3]
    assert slit._record().blade[0] == -0.5
    slit.kind = ('right', 'left', 'top', 'bottom')       # same edges, other blade names
    assert dict(zip(slit.kind, slit.opening)) == {'left': 0.25, 'right': -0.5, 'bottom': 0.3,
                                                  'top': -0.1}
    only_x = ra.RectangularAperture(bl, 'sx', [0, 100., 0], ('left', 'right'), [-1., 1.])
    only_x.opening = [-3., 3.]
    assert only_x.limOptX == [-3., 3.] and only_x.limOptY[1] == raycing.maxHalfSizeOfOE


def test_beam_round_trip_through_a_mat_file(tmp_path):
    """Beam(copyFrom='x.mat', bl=...): scipy's loadmat hands names and scalars back as
    arrays; the reference warns and goes on, here they become scalars / names again."""
    bl = raycing.BeamLine()
    src = rs.GeometricSource(bl, 'src', nrays=16)
    np.random.seed(1)
    beam = src.shine()
    path = os.path.join(str(tmp_path), 'beam.mat')
    beam.export_beam(path, 'mat')
    back = rs.Beam(copyFrom=path, bl=bl)
    for f in ('x', 'z', 'a', 'c', 'E', 'Jss', 'state'):
        assert np.array_equal(getattr(back, f), getattr(beam, f)), f
    other = rs.Beam(copyFrom=path)          # without a beamline: names stay names
    assert len(other.x) == 16
