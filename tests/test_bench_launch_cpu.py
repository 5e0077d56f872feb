"""`python bench.py --gpus N` started directly (no WORLD_SIZE): the command is its own launcher.  It must start N
fresh ranks before touching any GPU, relay rank 0's ONE JSON line and hand back a failing rank's status.  Here on
the CPU with `--dry-run` (process group over gloo, barrier, an agreed number; no GPU call at all); the same launch
with the training step on it runs in tests/test_parallel_gpu.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def run(args, env_drop=('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=300, cwd=ROOT)


def test_direct_start_with_two_ranks_prints_one_json_line():
    p = run(['--gpus', '2', '--dry-run'])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec == {'dry_run': True, 'n_gpus': 2, 'backend': 'gloo', 'max_rank_seen': 1.0}


def test_a_failing_rank_ends_the_launch_with_its_status():
    # every rank refuses the argument (two ranks, one device named) and exits 1: the launcher must not hang and
    # must not print a JSON line
    p = run(['--gpus', '2', '--dry-run', '--devices', '0'])
    assert p.returncode == 1
    assert p.stdout.strip() == ''
    assert '--devices names 1 GPUs for 2 ranks' in p.stderr


def test_the_launcher_imports_no_torch_and_makes_no_gpu_call():
    """The parent of `--gpus N` must not have initialised the GPU when it starts its children (and never execs):
    checked structurally - launch_ranks runs before `import torch` in main(), and names no exec."""
    src = open(BENCH).read()
    main = src[src.index('def main():'):src.index('def launch_ranks')]
    assert main.index('launch_ranks(args.gpus)') < main.index('import torch')
    body = src[src.index('def launch_ranks'):src.index('def source_sha')]
    assert 'os.exec' not in body and 'execv' not in body and 'import torch' not in body


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--dry-run'], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=120, cwd=ROOT)
    assert p.returncode != 0 and 'WORLD_SIZE=1' in p.stderr
