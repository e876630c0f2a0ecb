"""GPU: the multi-rank route of bench.py rehearsed on ONE GPU -- two ranks share the device
(XRT_BENCH_SHARE_GPU=1) and gloo carries the collectives (RCCL refuses two ranks on one GPU):
rendezvous, replicas of the ray pass, pixel tiles of the Kirchhoff integral, the packed gather,
rank statistics, the in-process multi-device leg (multigpu.kirchhoff_devices on devices [0, 0])
and exactly ONE JSON line on stdout. The driver's SCALE run is the first contact of this path
with more than one GPU; this keeps everything but RCCL itself exercised."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu():
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(XRT_BENCH_SHARE_GPU='1', XRT_BENCH_BACKEND='gloo')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps',
                        '3', '--warmup', '1', '--rays', '1e6', '--kirchhoff-steps', '1'],
                       capture_output=True, text=True, timeout=580, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['value'] > 1e9
    k = d['kirchhoff']
    assert k['n_gpus'] == 2 and k['rccl_ranks'] == 2 and len(k['kernel_ms_by_rank']) == 2
    assert all(ms > 0 for ms in k['kernel_ms_by_rank']) and k['value'] > 1e11
    assert 'all_gather_into_tensor' in k['gather']
    assert k['in_process'] and 'error' not in k['in_process'], k['in_process']
    assert k['in_process']['devices'] == [0, 0] and k['in_process']['value'] > 1e11
    # the self-check that runs before the timed steps when N > 1 (tiles + gather against one GPU)
    chk = d['multi_gpu_self_check']
    assert chk['ok'] and chk['gather_vs_one_gpu'] <= 1e-12, chk
    assert d['roofline']['multi_gpu_self_check'] == 1
    for leg in ('e2e', 'softimax', 'hist'):          # single-GPU legs stay out of a multi-rank line
        assert leg not in d
