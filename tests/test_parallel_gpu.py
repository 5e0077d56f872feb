"""Data-parallel step, 2 ranks sharing the one GPU of the test box (collectives over gloo, staged
through the host; RCCL needs one GPU per rank, which only the driver's 8-GPU run has).  What is
checked is the product's sharding algebra: BatchNorm from all-reduced column sums, loss means and
penalty scaled by the GLOBAL batch, flat-gradient all-reduce, identical replicas, and the HIP-graph
segmentation at the collectives."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(args, world=2, timeout=240, extra_env=None):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK='0', WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_worker.py')] + args, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank %d failed:\n%s' % (r, out[-4000:])


def test_two_ranks_replay_the_reference_trace():
    launch(['replay'])


@pytest.mark.parametrize('clip', ['none', 'median', 'all'])
def test_two_ranks_dpsgd_update_equals_the_single_process_oracle(clip):
    """Per-rank clip factors + all-reduced clipped sums + one shared noise add == oracle/dpsgd.py on the global batch
    (clip bound above every norm, at the median norm, below every norm)."""
    launch(['replay_dpsgd', clip])


@pytest.mark.parametrize('mode', ['reference', 'dpsgd'])
def test_two_ranks_graph_segments_equal_eager(mode):
    launch(['graph', mode])


RCCL1 = dict(DP_WORKER_BACKEND='nccl', DPCGAN_FORCE_DIST='1')


def test_rccl_single_rank_replays_the_reference_trace():
    """The real RCCL process group on the one GPU here (world size 1, collectives forced on): device
    all-reduces of fp32 / fp64 buffers on the step's streams, against the reference trace."""
    launch(['replay'], world=1, extra_env=RCCL1)


@pytest.mark.parametrize('collectives', ['in-graph', 'between-segments'])
@pytest.mark.parametrize('mode', ['reference', 'dpsgd'])
def test_rccl_single_rank_graph_replay_equals_eager(mode, collectives):
    """GraphStep over RCCL: by default the all-reduces are nodes of the one step graph; DPCGAN_GRAPH_COLLECTIVES=0
    keeps them eager between graph segments (the fallback).  Both bit-equal to the eager step."""
    env = dict(RCCL1, DPCGAN_GRAPH_COLLECTIVES='1' if collectives == 'in-graph' else '0')
    launch(['graph', mode], world=1, extra_env=env)


def test_rccl_capture_failure_falls_back_to_graph_segments():
    """The first collective of the in-graph capture raises (test hook): GraphStep must warn, end that capture cleanly
    although branches were open, re-capture in segments, reproduce the eager step bit for bit - and a later in-graph
    capture on the same streams must still work."""
    launch(['fallback', 'dpsgd'], world=1, extra_env=RCCL1)


@pytest.mark.parametrize('mode', ['reference', 'dpsgd'])
def test_rccl_single_rank_default_step_with_fused_statistics(mode):
    """The default data-parallel step (statistics from the GEMM epilogues, folded slots, paired forwards) over RCCL."""
    launch(['fused', mode], world=1, extra_env=RCCL1)


@pytest.mark.parametrize('mode', ['dpsgd'])
def test_two_ranks_default_step_with_fused_statistics(mode):
    launch(['fused', mode])


def test_two_ranks_per_rank_batchnorm_equals_the_mean_of_the_shard_oracles():
    """bn_sync=False (what DistributedDataParallel computes): each rank normalises over its own rows; the all-reduced
    gradients equal the mean over ranks of the single-process oracle on each shard, the running statistics are the
    rank's own, the parameters stay replicated."""
    launch(['local_bn'])


@pytest.mark.parametrize('mode', ['reference', 'dpsgd'])
def test_two_ranks_per_rank_batchnorm_graph_segments_equal_eager(mode):
    """... and its graph form: two eager all-reduces (the gradients) between three segments, bit-equal to eager."""
    launch(['graph_local_bn', mode])


@pytest.mark.parametrize('collectives', ['in-graph', 'between-segments'])
def test_rccl_single_rank_per_rank_batchnorm_graph_replay_equals_eager(collectives):
    """The single-GPU schedule (generator forward on its own lane beside the discriminator update's tail) with the two
    gradient all-reduces as graph nodes / as graph cuts that join the lane first."""
    env = dict(RCCL1, DPCGAN_GRAPH_COLLECTIVES='1' if collectives == 'in-graph' else '0')
    launch(['graph_local_bn', 'dpsgd'], world=1, extra_env=env)


@pytest.mark.parametrize('bn_sync', [0, 1])
def test_bench_starts_its_own_ranks(bn_sync):
    """`python bench.py --gpus 2` started directly: the launcher's two children share cuda:0 here (over gloo), run the
    timed graph-replayed step and rank 0's JSON line comes back through the launcher.  Per-rank BatchNorm statistics
    (the weak-scaling default) leave the two gradient all-reduces, --bn-sync 1 adds the four BatchNorm ones."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--devices', '0,0', '--rows', '20000',
                        '--batch', '512', '--dims', '64,64', '--steps', '3', '--warmup', '2', '--no-roofline',
                        '--no-cpu-baseline'] + (['--bn-sync', '1'] if bn_sync else []), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=400, cwd=root)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['steps'] == 3 and rec['value'] > 0 and rec['scaling'] == 'weak'
    assert rec['config']['global_batch'] == 1024 and rec['config']['parallelism'] == 'dp2'
    # gloo: graph segments cut at the collectives
    assert ('%d eager all-reduces between them' % (6 if bn_sync else 2)) in rec['config']['launch'], rec['config']['launch']
    assert rec['config']['batchnorm'] == ('statistics over the global batch' if bn_sync else 'per-rank statistics')
