"""CPU: bench.py's own launcher.  ``--gpus N`` without a torch.distributed
launcher must start N ranks itself (fresh children, before any GPU call), the
line's n_gpus must be the ranks the process group connected, and a mismatch
between --gpus and the launcher's WORLD_SIZE, or too few GPUs for RCCL, must
fail loudly instead of measuring one GPU."""
import json
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, 'bench.py')


def _run(args, env=None):
    e = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, timeout=240,
                          capture_output=True, text=True)


def test_self_spawn_two_ranks_gloo():
    r = _run(['--gpus', '2', '--backend', 'gloo', '--rendezvous-only'])
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                      # rank 0 only
    assert json.loads(lines[0])['n_gpus'] == 2


def test_gpus_must_match_world_size():
    r = _run(['--gpus', '4', '--rendezvous-only'],
             {'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode == 2 and 'WORLD_SIZE' in r.stderr
    r = _run(['--gpus', '1', '--rendezvous-only'],
             {'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode == 2


def test_rccl_needs_one_gpu_per_rank():
    import torch
    have = torch.cuda.device_count()
    r = _run(['--gpus', str(have + 2), '--rendezvous-only'])
    assert r.returncode == 2 and 'GPU(s) visible' in r.stderr
    assert not r.stdout.strip()


def test_single_rank_is_unchanged():
    r = _run(['--gpus', '1', '--rendezvous-only'])
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1])['n_gpus'] == 1
