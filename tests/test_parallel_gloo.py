"""CPU rehearsal of the collectives wrapper (gloo, world size 2): sums, max-over-ranks timing,
barrier.  The sharded step itself needs the HIP kernels and is covered by test_parallel_gpu.py."""
import os
import socket
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))

WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from dp_cgans_amd.parallel import Dist
d = Dist(backend='gloo')
assert d.active and d.world == 2 and d.backend == 'gloo'
t = torch.arange(6, dtype=torch.float32) * (d.rank + 1)
d.allreduce(t)
assert torch.equal(t, torch.arange(6, dtype=torch.float32) * 3)
p = torch.full((5,), float(d.rank), dtype=torch.float64)
d.allreduce(p[1:3])                    # views (BatchNorm partials are slices of a workspace)
assert p.tolist() == [d.rank, 1.0, 1.0, d.rank, d.rank]
assert d.max_float(1.0 + d.rank, 'cpu') == 2.0
d.barrier(); d.shutdown()
print('ok')
'''


def test_dist_wrapper_world_size_2(tmp_path):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / 'w.py'
    script.write_text(WORKER)
    root = os.path.dirname(HERE)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), root], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0 and 'ok' in out, out[-2000:]


def test_single_process_is_a_no_op():
    import torch
    from dp_cgans_amd.parallel import Dist
    env = {k: os.environ.pop(k, None) for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    try:
        d = Dist()
        t = torch.ones(3)
        assert not d.active and d.allreduce(t) is t and d.max_float(2.5, 'cpu') == 2.5
        d.barrier()
    finally:
        for k, v in env.items():
            if v is not None:
                os.environ[k] = v
