"""CPU: the host side of xrt_amd/graphs.py outside any recording -- bookkeeping runs at once,
refusals do not fire, and the runner's keyword exists with the reference's default behaviour."""
import inspect

from xrt_amd import graphs, runner


def test_bookkeeping_runs_at_once_when_nothing_is_recorded():
    seen = []
    assert graphs.capturing() is None
    graphs.per_iteration(lambda: seen.append(1))
    graphs.refuse('anything')          # no recording: not an error
    assert seen == [1]


def test_refusal_names_the_reason_while_recording():
    class Fake(object):
        after_replay = []
    graphs._tls.recording = Fake()
    try:
        graphs.per_iteration(lambda: None)
        assert len(Fake.after_replay) == 1
        try:
            graphs.refuse('numpy random numbers')
        except graphs.CaptureError as e:
            assert 'numpy random numbers' in str(e) and 'graph=False' in str(e)
        else:
            raise AssertionError('no CaptureError')
    finally:
        graphs._tls.recording = None


def test_run_ray_tracing_is_eager_by_default():
    assert inspect.signature(runner.run_ray_tracing).parameters['graph'].default is False
