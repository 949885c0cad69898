"""bench.py --gpus N must launch itself (the driver calls `python bench.py --gpus N` without a launcher): the N ranks are
spawned as children of a fresh torch.distributed.run before anything touches a GPU.  Exercised here over gloo with the
GPU work switched off (--rehearse-cpu): launch, rendezvous on 127.0.0.1, all-reduce, barrier, one JSON line, exit code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, env=e, timeout=timeout)


def test_bench_launches_its_own_ranks():
    r = _run(['--gpus', '2', '--steps', '2', '--warmup', '1', '--rehearse-cpu'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['sum_of_ranks'] == 3.0 and out['global_batch'] == 32768


def test_bench_config5_rehearsal_names_its_workload():
    """--config 5 = BASELINE configs[4] (case300, 8192 grids per GPU, K=10): the launcher path a future 8-GPU run of it takes."""
    r = _run(['--config', '5', '--gpus', '2', '--steps', '2', '--warmup', '1', '--rehearse-cpu'])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert out['n_gpus'] == 2 and out['config'] == 5 and out['global_batch'] == 2 * 8192 and 'case300' in out['workload']


def test_bench_byte_and_flop_models_match_the_survey():
    """compulsory bytes and nominal forward MFLOP per grid of the three bench configurations."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.bytes_per_grid(118) == 10504 and b.bytes_per_grid(300) == 23048 and b.bytes_per_grid(30) == 2284
    assert abs(b.mflop_fwd_per_grid(118, 4) - 4.19216) < 1e-5 and abs(b.mflop_fwd_per_grid(300, 10) - 24.603) < 1e-3
    assert abs(b.mflop_fwd_per_grid(30, 4) - 0.9828) < 1e-4


def test_bench_rejects_a_world_size_that_does_not_match():
    r = _run(['--gpus', '4', '--rehearse-cpu'], env={'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'does not match' in (r.stderr + r.stdout)
