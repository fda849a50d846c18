"""CPU: the host logic of bench.py that does not need a GPU -- the self-launch of `--gpus G > 1` (a CHILD
torch.distributed.run process, started before anything touches the GPU) and the committed profile digest that the
JSON line's `roofline.kernels` is read from."""
import json
import os
import sys
import types

import pytest

from conftest import REPO


def _bench():
    sys.path.insert(0, REPO)
    import bench
    return bench


def test_self_launch_spawns_torch_distributed_run(monkeypatch):
    bench = _bench()
    calls = []
    monkeypatch.setattr(bench.subprocess, 'call', lambda cmd: calls.append(cmd) or 0)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '7', '--warmup', '2'])
    assert bench.self_launch(types.SimpleNamespace(gpus=4)) == 0
    cmd = calls[0]
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and '--nproc-per-node=4' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    assert cmd[-7:] == [os.path.join(REPO, 'bench.py'), '--gpus', '4', '--steps', '7', '--warmup', '2']


def test_main_self_launches_without_a_launcher(monkeypatch):
    bench = _bench()
    seen = []
    monkeypatch.setattr(bench, 'self_launch', lambda args: seen.append(args.gpus) or 0)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8'])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0 and seen == [8]


def test_committed_step_counters_are_physical():
    """profiles/r02_step_counters.json (tools/make_step_counters.py): every fraction <= 1, the kernels bench.py
    lists are present."""
    path = os.path.join(REPO, 'profiles', 'r02_step_counters.json')
    j = json.load(open(path))
    assert j['config'] == {'agents_total': 4096, 'obstacle_points': 2000}
    names = [k['name'] for k in j['all_step_kernels']]
    for need in ('relfeat_fwd_kernel', 'enc_fwd_x3_kernel', 'enc_bwd_dx_x3_kernel', 'enc_bwd_dw_x3_kernel', 'dec_fwd_head_kernel',
                 'pinnsf_reduce_kernel', 'pinnsf_pack_kernel'):
        assert need in names
    for k in j['all_step_kernels']:
        assert k['us'] > 0 and (k.get('frac') is None or 0 <= k['frac'] <= 1), k
    assert 0 < j['relfeat_fwd_kernel']['valu_busy_frac'] <= 1
    assert j['step_hbm_bytes'] > 1e8


def test_packed_weights_is_a_noop_without_a_gpu():
    """`model.packed_weights()` on a CPU model (and on configurations the fused network does not cover) must just run
    the block: no HIP call, no state left behind."""
    import types
    import torch
    import piml_amd.models.model as MODEL
    args = types.SimpleNamespace(
        ped_feature_dim=6, obs_feature_dim=6, self_feature_dim=7, encoder_hidden_size=32,
        processor_hidden_size=32, decoder_hidden_size=16, encoder_hidden_layers=2, processor_hidden_layers=2,
        decoder_hidden_layers=2, dropout=0.0, activation='relu', dataset_name='gc1560')
    for name in ('PINNSF_multitask', 'PINNSF_bottleneck_multitask'):
        net = getattr(MODEL, name)(args).eval()
        x = [torch.randn(5, 6, 6), torch.randn(5, 10, 6), torch.randn(5, 7)]
        with net.packed_weights():
            a = net(*x)[0]
            with net.packed_weights():          # re-entrant
                b = net(*x)[0]
        assert torch.equal(a, b) and (net._packs is None or not net._packs.active)
