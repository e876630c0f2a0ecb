"""The Balder beamline of the reference's example (examples/withRaycing/02_Balder_BL) set up
with xrt_amd from the numbers stored in golden g17_balder_chain (the example's parameters
after its own alignment), and its ray path run element by element the way the example's
run_process does."""
import math

import numpy as np

import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.apertures as ra
import xrt_amd.backends.raycing.materials as rm
import xrt_amd.backends.raycing.oes as roe
import xrt_amd.backends.raycing.screens as rsc
import xrt_amd.backends.raycing.sources as rs

SIDES = ('left', 'right', 'bottom', 'top')


def build(g):
    par = {k[4:]: g[k] for k in g.files if k.startswith('par_')}
    bl = raycing.BeamLine(azimuth=0, height=0)
    rs.Wiggler(bl, name='SoleilW50', center=(0, 0, 0), nrays=int(par['nrays']), period=50.,
               K=8.446, n=39, eE=3., eI=0.5, eSigmaX=48.66, eSigmaZ=6.197, eEpsilonX=0.263,
               eEpsilonZ=0.008, eMin=float(par['src_eMin']), eMax=float(par['src_eMax']),
               xPrimeMax=0.22, zPrimeMax=0.06)
    bl.fsm0 = rsc.Screen(bl, 'FSM0', (0, 15000, 0))
    bl.feFixedMask = ra.RectangularAperture(bl, 'FEFixedMask', (0, 15750, 0),
                                            blades=dict(zip(SIDES, par['mask'])))
    bl.fsmFE = rsc.Screen(bl, 'FSM-FE', (0, 16000, 0))
    diamond = rm.Material('C', rho=3.52, kind='plate')
    bl.filter1 = roe.Plate(bl, 'Filter1', (0, 23620, 0), pitch=math.pi/2, limPhysX=(-9., 9.),
                           limPhysY=(-4., 4.), material=diamond, t=0.06)
    bl.vcm = roe.SimpleVCM(bl, 'VCM', [0, 25290, 0], surface=('Si',),
                           material=(rm.Material('Si', rho=2.33),), limPhysX=(-15., 15.),
                           limPhysY=(-680., 680.), limOptX=(-6, 6), limOptY=(-670., 670.),
                           R=float(par['vcm_R']), pitch=float(par['vcm_pitch']))
    bl.fsmVCM = rsc.Screen(bl, 'FSM-VCM', (0, 26300, 0))
    bl.dcm = roe.DCM(bl, 'DCM', [0, 27060, float(par['dcm_z'])], surface=('Si111',),
                     material=(rm.CrystalSi(hkl=(1, 1, 1), tK=-171+273.15),),
                     material2=(rm.CrystalSi(hkl=(1, 1, 1), tK=-140+273.15),),
                     alpha=np.radians(0), limPhysX=(-10, 10), limPhysY=(-30, 30),
                     cryst2perpTransl=float(par['dcm_perp']), cryst2longTransl=65,
                     limPhysX2=(-10, 10), limPhysY2=(-90, 90), bragg=float(par['dcm_bragg']))
    bl.BSBlock = ra.RectangularAperture(bl, 'BSBlock', (0, 29100, 0), blades={'bottom': 22})
    bl.slitAfterDCM = ra.RectangularAperture(bl, 'SlitAfterDCM', (0, 29200, 0),
                                             blades=dict(zip(SIDES, par['slitDCM'])))
    bl.fsmDCM = rsc.Screen(bl, 'FSM-DCM', (0, 29400, 0))
    bl.vfm = roe.SimpleVFM(bl, 'VFM', [0, 30575, float(par['vfm_z'])], surface=('SiO2',),
                           material=(rm.Material(('Si', 'O'), quantities=(1, 2), rho=2.2),),
                           limPhysX=(-20., 20.), limPhysY=(-700., 700.), limOptX=(-10, 10),
                           limOptY=(-700, 700), positionRoll=math.pi, R=float(par['vfm_R']),
                           r=40.77, pitch=float(par['vfm_pitch']))
    bl.slitAfterVFM = ra.RectangularAperture(bl, 'SlitAfterVFM', (0, 31720, 0),
                                             blades=dict(zip(SIDES, par['slitVFM'])))
    bl.fsmVFM = rsc.Screen(bl, 'FSM-VFM', (0, 32000, 0))
    bl.ohPS = ra.RectangularAperture(bl, 'OH-PS', (0, 32070, 0),
                                     blades={'left': -20, 'right': 20, 'bottom': 25, 'top': 55})
    bl.slitEH = ra.RectangularAperture(bl, 'slitEH', (0, 43000, 0),
                                       blades=dict(zip(SIDES, par['slitEH'])))
    bl.fsmSample = rsc.Screen(bl, 'FSM-Sample', (0, 45863, 0))
    return bl, int(par['seed'])


def trace(bl):
    out = {}
    src = out['beamSource'] = bl.sources[0].shine()
    out['beamFSM0'] = bl.fsm0.expose(src)
    bl.feFixedMask.propagate(src)
    out['beamFSMFE'] = bl.fsmFE.expose(src)
    f1g, f1l1, f1l2 = bl.filter1.double_refract(src)
    out['beamFilter1global'] = f1g
    lost = out['beamFilter1local2A'] = rs.Beam(copyFrom=f1l2)
    lost.absorb_intensity(src)
    vg, vl = bl.vcm.reflect(f1g)
    vl.absorb_intensity(f1g)
    out['beamVCMglobal'], out['beamVCMlocal'] = vg, vl
    out['beamFSMVCM'] = bl.fsmVCM.expose(vg)
    dg, dl1, dl2 = bl.dcm.double_reflect(vg)
    dl1.absorb_intensity(vg)
    out['beamDCMglobal'], out['beamDCMlocal1'], out['beamDCMlocal2'] = dg, dl1, dl2
    bl.BSBlock.propagate(dg)
    out['beamSlitAfterDCMlocal'] = bl.slitAfterDCM.propagate(dg)
    out['beamFSMDCM'] = bl.fsmDCM.expose(dg)
    fg, fl = bl.vfm.reflect(dg)
    out['beamVFMglobal'], out['beamVFMlocal'] = fg, fl
    bl.slitAfterVFM.propagate(fg)
    out['beamFSMVFM'] = bl.fsmVFM.expose(fg)
    bl.ohPS.propagate(fg)
    out['beamSlitEHLocal'] = bl.slitEH.propagate(fg)
    out['beamFSMSample'] = bl.fsmSample.expose(fg)
    return out
