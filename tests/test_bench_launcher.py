"""`python bench.py --gpus N` starts its own ranks (bench.spawn_ranks): the parent stays off the
GPU, the children get the environment torch.distributed.run would give them, rank 0's ONE JSON
line is relayed, and a failing rank makes the whole call fail.  CPU only: the children here are
small scripts, not the benchmark."""
import io
import json
import os
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _script(tmp_path, body):
    path = tmp_path / 'child.py'
    path.write_text(textwrap.dedent(body))
    return str(path)


def test_children_rendezvous_over_gloo_and_rank0_line_is_relayed(tmp_path):
    script = _script(tmp_path, '''
        import json, os, sys
        import torch
        import torch.distributed as dist
        dist.init_process_group('gloo')          # env:// -- RANK, WORLD_SIZE, MASTER_* from the parent
        t = torch.tensor([float(dist.get_rank() + 1)])
        dist.all_reduce(t)
        assert os.environ['LOCAL_RANK'] == os.environ['RANK']
        print(json.dumps({'n_gpus': dist.get_world_size(), 'sum': float(t), 'argv': sys.argv[1:]}))
        dist.barrier()
        dist.destroy_process_group()
    ''')
    out, err = io.StringIO(), io.StringIO()
    rc = bench.spawn_ranks(2, ['--gpus', '2', '--steps', '3'], script=script, out=out, err=err,
                           timeout=120)
    assert rc == 0, err.getvalue()
    lines = out.getvalue().splitlines()
    assert len(lines) == 1                      # rank 1's line is not relayed
    rec = json.loads(lines[0])
    assert rec == {'n_gpus': 2, 'sum': 3.0, 'argv': ['--gpus', '2', '--steps', '3']}


def test_a_failing_rank_fails_the_call_and_stops_the_others(tmp_path):
    script = _script(tmp_path, '''
        import os, sys, time
        if os.environ['RANK'] == '1':
            sys.exit(3)
        time.sleep(600)                          # rank 0 would wait for its peer forever
    ''')
    out, err = io.StringIO(), io.StringIO()
    rc = bench.spawn_ranks(2, [], script=script, out=out, err=err, timeout=60)
    assert rc != 0 and out.getvalue() == ''


def test_more_or_less_than_one_line_from_rank0_is_an_error(tmp_path):
    script = _script(tmp_path, '''
        import os
        if os.environ['RANK'] == '0':
            print('{"a": 1}')
            print('{"a": 2}')
    ''')
    out, err = io.StringIO(), io.StringIO()
    assert bench.spawn_ranks(2, [], script=script, out=out, err=err, timeout=60) != 0
    assert out.getvalue() == ''


def test_the_parent_needs_no_torch_before_spawning():
    """The launcher branch of main() runs before `import torch` (a process that has initialised
    the GPU must not start others by exec; children are plain subprocesses)."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    main = src[src.index('def main():'):]
    assert main.index('spawn_ranks(') < main.index('import torch')
    head = src[:src.index('def main():')]
    assert 'import torch' not in head.replace('import torch\n        import torch.distributed', '')
