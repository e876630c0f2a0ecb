"""``run_process`` hook, as xrt/backends/raycing/run.py:2-9: user scripts assign
their own ``run_process(beamLine) -> dict(name=Beam)`` here."""


def run_process(beamLine):
    raise NotImplementedError(
        'assign your own run_process(beamLine) to '
        'xrt_amd.backends.raycing.run.run_process')
