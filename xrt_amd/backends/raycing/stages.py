"""Mechanical supports of optical elements (reference xrt/backends/raycing/stages.py): a
tripod of three vertical jacks under the element (height, pitch, roll) and one or two
horizontal translation stages (x shift, yaw). Host arithmetic; mixed into the element
classes, which provide ``center``, ``pitch``, ``roll``, ``yaw``, ``positionRoll``, ``bl``."""
import math

from .. import raycing


class Tripod(object):
    """*jack1*, *jack2*, *jack3*: [x, y, z] of the jack balls in the global frame with the
    element horizontal (all three at one height)."""

    def __init__(self, jack1, jack2, jack3):
        self.jack1, self.jack2, self.jack3 = jack1, jack2, jack3
        above_floor = self.center[2] - self.bl.height
        self.jack1Offset, self.jack2Offset, self.jack3Offset = (
            above_floor - j[2] for j in self._jacks())
        for step in (self.init_jacks_local, self.set_jacks):
            step()

    def _jacks(self):
        return self.jack1, self.jack2, self.jack3

    def pop_kwargs(self, **kwargs):
        jacks = tuple(kwargs.pop('jack%d' % k) for k in (1, 2, 3))
        return kwargs, jacks

    def init_jacks_local(self):
        """The jacks relative to the element's centre, y along the beamline; the distance
        from the plane of the balls to the optical surface never changes."""
        if not (self.jack1[2] == self.jack2[2] == self.jack3[2]):
            raise ValueError('the three jacks must start at one height (element horizontal)')
        self.jackToMirrorInvariant = self.center[2] - self.jack1[2]
        local = []
        for jack in self._jacks():
            rel = [j - c for j, c in zip(jack, self.center)]
            rel[0], rel[1] = raycing.rotate_z(rel[0], rel[1], self.bl.cosAzimuth,
                                              self.bl.sinAzimuth)
            local.append(rel)
        self.jack1local, self.jack2local, self.jack3local = local

    def set_jacks(self):
        """Jack heights that realise the element's pitch and roll: the balls lie in the
        plane n . r = -invariant, n = the element's normal."""
        nx, ny, nz = 0.0, 0.0, 1.0
        tilt = self.pitch * math.cos(self.positionRoll)
        if self.roll != 0:
            nx, nz = raycing.rotate_y(nx, nz, math.cos(self.roll), math.sin(self.roll))
        if tilt != 0:
            ny, nz = raycing.rotate_x(ny, nz, math.cos(tilt), math.sin(tilt))
        level = 0
        level -= self.jackToMirrorInvariant
        for rel, jack in zip((self.jack1local, self.jack2local, self.jack3local),
                             self._jacks()):
            rel[2] = (level - nx*rel[0] - ny*rel[1]) / nz
            jack[2] = rel[2] + self.center[2]
        for k, jack in enumerate(self._jacks(), 1):      # what the jack encoders read
            setattr(self, 'jack%dCalib' % k, jack[2] + getattr(self, 'jack%dOffset' % k))

    def get_orientation(self):
        """Height of the centre, pitch and roll from the three jack positions: the normal
        of the plane through the balls."""
        j1, j2, j3 = self._jacks()
        nx = (j2[1]-j1[1]) * (j3[2]-j1[2]) - (j3[1]-j1[1]) * (j2[2]-j1[2])
        ny = (j3[0]-j1[0]) * (j2[2]-j1[2]) - (j2[0]-j1[0]) * (j3[2]-j1[2])
        nz = (j2[0]-j1[0]) * (j3[1]-j1[1]) - (j3[0]-j1[0]) * (j2[1]-j1[1])
        length = (nx**2 + ny**2 + nz**2)**0.5
        if nz < 0:
            length *= -1          # the normal looks up
        nx /= length
        ny /= length
        nz /= length
        plane = nx*j1[0] + ny*j1[1] + nz*j1[2]
        plane += self.jackToMirrorInvariant
        self.center[2] = (plane - nx*self.center[0] - ny*self.center[1]) / nz
        along_x, along_y = raycing.rotate_z(nx, ny, self.bl.cosAzimuth, self.bl.sinAzimuth)
        self.roll = math.atan(along_x / nz)
        slope = -along_y / (along_x*math.sin(self.roll) + nz*math.cos(self.roll))
        self.pitch = math.atan(slope) * math.cos(self.positionRoll)


class OneXStage(object):
    """Horizontal translation of the element across the beam: *dx* = shift of the centre in
    the local frame. An element with several stripes (``surface`` = their names, limits
    given per stripe) is moved to a stripe by ``select_surface``."""

    def __init__(self, dx=0):
        self.dx = dx
        names = self.surface
        if names is None:
            return
        if not raycing.is_sequence(names):
            raise ValueError('"surface": the names of the stripes, a sequence')
        stripes = len(names)
        for optical in (self.limOptX, self.limOptY):
            if optical is None:
                continue
            if not (raycing.is_sequence(optical[0]) and raycing.is_sequence(optical[1])):
                raise ValueError('optical limits of a multi-stripe element: (lows, highs), '
                                 'each a sequence per stripe')
            if not (len(optical[0]) == len(optical[1]) == stripes):
                raise ValueError('one optical limit per stripe, please')
        for edge in (self.limPhysX[0], self.limPhysX[1], self.limPhysY[0], self.limPhysY[1]):
            if raycing.is_sequence(edge) and len(edge) != stripes:
                raise ValueError('one physical limit per stripe, please')

    def pop_kwargs(self, **kwargs):
        return kwargs, (kwargs.pop('dx', 0),)

    def select_surface(self, surfaceName):
        if self.surface is None:
            return
        self.curSurface = stripe = self.surface.index(surfaceName)
        across = self.limPhysX if self.limOptX is None else self.limOptX
        self.dx = -(across[0][stripe] + across[1][stripe]) * 0.5
        self.get_surface_limits()


class TwoXStages(OneXStage):
    """Two translation stages at different y: x shift and yaw. *tx1*, *tx2*: [x, y] of the
    stages in the local frame (lists: their x is kept up to date)."""

    def __init__(self, tx1, tx2, dx=0):
        self.tx1, self.tx2 = tx1, tx2
        if tx2[1] == tx1[1]:
            raise ValueError('the two x stages must sit at different y')
        OneXStage.__init__(self)      # (its dx argument is not handed on, as in the reference)
        self.set_x_stages()

    def pop_kwargs(self, **kwargs):
        stages = kwargs.pop('tx1'), kwargs.pop('tx2')
        return kwargs, stages + (kwargs.pop('dx', 0),)

    def set_x_stages(self):
        slope = math.tan(self.yaw)
        turned = self.positionRoll != 0
        for stage in (self.tx1, self.tx2):
            stage[0] = (-slope*stage[1] + self.dx)
            if turned:                   # the element hangs rolled: the stage sees the projection
                stage[0] *= math.cos(self.positionRoll)

    def select_surface(self, surfaceName):
        """The stripe of that name into the beam; both stages follow."""
        OneXStage.select_surface(self, surfaceName)
        self.set_x_stages()

    def get_orientation(self):
        x1, x2 = self.tx1[0], self.tx2[0]
        if self.positionRoll != 0:
            x1, x2 = x1 * math.cos(self.positionRoll), x2 * math.cos(self.positionRoll)
        span = self.tx2[1] - self.tx1[1]
        self.dx = x1 - (x2-x1) * self.tx1[1] / span
        self.yaw = -math.atan((x2-x1) / span)
