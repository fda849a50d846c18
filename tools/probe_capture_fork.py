"""Which fork structure does hipStreamEndCapture survive?  Each variant in its own process.
Finding (ROCm 7.2, MI355X): forks inside piml_pinnsf_fwd / bwd and the prepack on ops' side stream capture and replay
fine; a pack that forked AGAIN from inside that side stream (library stream -> join -> join) segfaulted in
hipStreamEndCapture, so piml_pinnsf_pack is three launches in a row."""
import os
import subprocess
import sys

VARIANTS = {
    'noprepack_forked': dict(prepack='0', PIML_FORK_NETWORK='1'),
    'noprepack_serial': dict(prepack='0'),
    'prepack_forked': dict(prepack='1', PIML_FORK_NETWORK='1'),
    'prepack_serial': dict(prepack='1'),
}


def child(prepack):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import contextlib
    from tests.test_encoder_gpu import _multitask_net, _net_pass
    net = _multitask_net(3)
    g = torch.Generator().manual_seed(6)
    base = [torch.randn(512, 6, 6, generator=g).cuda(), torch.randn(512, 10, 6, generator=g).cuda(),
            torch.randn(512, 7, generator=g).cuda()]
    ctx = net.packed_weights if prepack else contextlib.nullcontext
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            with ctx():
                _net_pass(net, base)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    net.zero_grad(set_to_none=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph), ctx():
        out = net(*base)
        (out[0].square().sum() + out[1].sum() * 1e-2).backward()
    print('captured', flush=True)
    graph.replay()
    torch.cuda.synchronize()
    print('replayed OK', float(out[0].abs().sum()), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        child(sys.argv[1] == '1')
    else:
        for name, env in VARIANTS.items():
            e = dict(os.environ)
            prepack = env.pop('prepack')
            e.update(env)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), prepack], env=e, capture_output=True, text=True)
            print(f'{name}: rc={r.returncode} {r.stdout.strip().splitlines()[-1:] }', flush=True)
            if r.returncode:
                print('   ', '\n    '.join(r.stderr.strip().splitlines()[-6:]), flush=True)
